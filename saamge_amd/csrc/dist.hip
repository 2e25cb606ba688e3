// Row-partitioned solve phase (one process per GPU): ownership ranges, halo-exchange lists and
// the exchange itself.  Reference counterpart: hypre's ParCSR communication package behind
// every HypreParMatrix::Mult of the V-cycle (amg/src/tg.cpp:91-132) and the MPI_Allreduce of
// the PCG inner products (amg/src/mfem_addons.cpp:106-248).
//
// Layout: every rank keeps GLOBAL-length vectors; rank r owns rows [row_off[r], row_off[r+1])
// (boundaries are multiples of 256 so that neither SELL-64 slices nor the 256-row tiles of the staged kernel are split) and applies only those
// rows of A_l, reading x by global column index.  Entries of x outside the own range are valid
// only at the halo positions (the columns the own rows reference), which are refreshed from
// their owners before each SpMV: pack kernel -> alltoallv (RCCL send/recv groups through
// torch.distributed) -> unpack kernel.  No column renumbering, no second matrix copy.
#include "dist.h"

#include <algorithm>
#include <cstdlib>

namespace saamge_amd {


__global__ __launch_bounds__(256) void halo_mark_kernel(int row0, int nloc, const roff_t *__restrict__ rowptr,
                                                        const int *__restrict__ col, int *__restrict__ flag) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 7;
    const long r = gtid >> 3;
    if (r >= nloc) return;
    const int row = row0 + (int)r;
    for (roff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 8) {
        const int c = col[k];
        if (c < row0 || c >= row0 + nloc) flag[c] = 1;
    }
}

// slice s of the own rows reads a column outside the own range
__global__ __launch_bounds__(256) void halo_slice_kernel(int row0, int nloc, const roff_t *__restrict__ rowptr,
                                                         const int *__restrict__ col, int *__restrict__ sflag) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 7;
    const long r = gtid >> 3;
    if (r >= nloc) return;
    const int row = row0 + (int)r;
    bool out = false;
    for (roff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 8) {
        const int c = col[k];
        out |= (c < row0 || c >= row0 + nloc);
    }
    if (out) sflag[r >> 6] = 1;
}

__global__ __launch_bounds__(256) void halo_compact_kernel(int n, const int *__restrict__ flag,
                                                           const int *__restrict__ pos, int *__restrict__ idx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) idx[pos[i]] = (int)i;
}

__global__ __launch_bounds__(256) void halo_pack_kernel(int n, const int *__restrict__ idx,
                                                        const double *__restrict__ x, double *__restrict__ buf) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) buf[i] = x[idx[i]];
}

__global__ __launch_bounds__(256) void halo_unpack_kernel(int n, const int *__restrict__ idx,
                                                          const double *__restrict__ buf, double *__restrict__ x) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[idx[i]] = buf[i];
}

static void comm_fence(Hierarchy &H) {
    if (!H.params.comm_stream_ordered) SA_HIP_CHECK(hipStreamSynchronize(H.stream));
}

void dist_allreduce(Hierarchy &H, double *buf, long long count) {
    if (H.params.world <= 1 || count == 0) return;
    comm_fence(H);
    SA_REQUIRE(H.params.allreduce_sum(H.params.allgather_ctx, buf, count) == 0, "allreduce callback failed");
}

void dist_allgather_rows(Hierarchy &H, Level::Dist &D, double *x) {
    comm_fence(H);
    SA_REQUIRE(H.params.allgather(H.params.allgather_ctx, x, D.own_off.data()) == 0,
               "allgather callback failed");
}

// Reduce-scatter of a vector of partial sums over the ownership ranges of D (the restricted residual on its way to a
// row-partitioned coarser level, which reads it on its own rows only): every rank sends each owner its slab of partial
// sums (one all-to-all: half the bytes of the full-vector all-reduce it replaces) and adds the world contributions of its
// own slab in RANK ORDER -- the sum does not depend on the collective's algorithm.  The other rows of buf keep this
// rank's partial sums (nobody reads them).
// (the all-to-all implementations skip a rank's piece for itself: its own partial sums are read where they lie)
__global__ __launch_bounds__(256) void slab_sum_kernel(int world, int rank, int nloc, const double *__restrict__ parts, double *out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nloc) return;
    double s = 0.0;
    for (int p = 0; p < world; ++p) {
        const double v = (p == rank) ? out[i] : parts[(size_t)p * nloc + i];
        s = (p == 0) ? v : s + v;
    }
    out[i] = s;
}
void dist_reduce_scatter_rows(Hierarchy &H, Level::Dist &D, double *buf) {
    const int world = H.params.world;
    if (world <= 1) return;
    if (D.rs_buf.n < (size_t)world * (size_t)D.nloc) D.rs_buf.alloc((size_t)world * (size_t)D.nloc + 1);
    if (D.rs_off.empty()) {
        D.rs_off.resize((size_t)world + 1);
        for (int p = 0; p <= world; ++p) D.rs_off[(size_t)p] = 8ll * p * D.nloc;
    }
    comm_fence(H);
    SA_REQUIRE(H.params.alltoallv(H.params.allgather_ctx, buf, D.own_off.data(), D.rs_buf.p, D.rs_off.data()) == 0,
               "alltoallv callback failed");
    if (D.nloc) {
        hipLaunchKernelGGL(slab_sum_kernel, dim3(div_up(D.nloc, 256)), dim3(256), 0, H.stream, world, H.params.rank, D.nloc, D.rs_buf.p, buf + D.row0);
        SA_HIP_CHECK(hipGetLastError());
    }
}

void halo_exchange(Hierarchy &H, Level::Dist &D, double *x) {
    hipStream_t s = H.stream;
    if (D.nsend) {
        profiler().begin(s);
        hipLaunchKernelGGL(halo_pack_kernel, dim3(div_up(D.nsend, 256)), dim3(256), 0, s, D.nsend,
                           D.send_idx.p, x, D.send_buf.p);
        SA_HIP_CHECK(hipGetLastError());
        profiler().end(s, "halo_pack", 20.0 * D.nsend, 0.0);
    }
    comm_fence(H);
    SA_REQUIRE(H.params.alltoallv(H.params.allgather_ctx, D.send_buf.p, D.send_off.data(), D.recv_buf.p,
                                  D.recv_off.data()) == 0,
               "alltoallv callback failed");
    if (D.nrecv) {
        profiler().begin(s);
        hipLaunchKernelGGL(halo_unpack_kernel, dim3(div_up(D.nrecv, 256)), dim3(256), 0, s, D.nrecv,
                           D.recv_idx.p, D.recv_buf.p, x);
        SA_HIP_CHECK(hipGetLastError());
        profiler().end(s, "halo_unpack", 20.0 * D.nrecv, 0.0);
    }
}

void halo_then(Hierarchy &H, Level::Dist *D, double *x, const std::function<void(hipStream_t, RowRange)> &op) {
    hipStream_t s = H.stream;
    if (!D) {
        op(s, RowRange());
        return;
    }
    RowRange own;
    own.row0 = D->row0;
    own.nrows = D->nloc;
    if (D->int_nrows <= 0) {
        halo_exchange(H, *D, x);
        op(s, own);
        return;
    }
    if (!H.ev_fork) {
        SA_HIP_CHECK(hipEventCreateWithFlags(&H.ev_fork, hipEventDisableTiming));
        SA_HIP_CHECK(hipEventCreateWithFlags(&H.ev_join, hipEventDisableTiming));
    }
    hipStream_t side = side_stream(3);
    SA_HIP_CHECK(hipEventRecord(H.ev_fork, s));              // x is complete on the own rows
    SA_HIP_CHECK(hipStreamWaitEvent(side, H.ev_fork, 0));
    RowRange in;
    in.row0 = D->int_row0;
    in.nrows = D->int_nrows;
    {
        ThreadStreamScope on_side(side);                      // (temporaries of op() are freed in the side stream's order)
        op(side, in);                                         // rows without halo entries: beside the exchange
    }
    SA_HIP_CHECK(hipEventRecord(H.ev_join, side));
    halo_exchange(H, *D, x);
    RowRange lo, hi;
    lo.row0 = D->row0;
    lo.nrows = D->int_row0 - D->row0;
    hi.row0 = D->int_row0 + D->int_nrows;
    hi.nrows = D->row0 + D->nloc - hi.row0;
    if (lo.nrows > 0) op(s, lo);
    if (hi.nrows > 0) op(s, hi);
    SA_HIP_CHECK(hipStreamWaitEvent(s, H.ev_join, 0));
}

bool dist_setup_level(Hierarchy &H, int lev) {
    const Params &p = H.params;
    Level &L = *H.levels[lev];
    Level::Dist &D = L.dist;
    D.on = false;
    const int world = p.world, rank = p.rank, n = L.A.nrows;
    if (world <= 1 || !p.allreduce_sum || !p.alltoallv || !p.allgather) return false;
    if ((long long)n < p.dist_min_local_rows * (long long)world) return false;
    if (lev > 0 && !H.levels[lev - 1]->dist.on) return false;
    hipStream_t s = H.stream;
    D.row_off.assign((size_t)world + 1, 0);
    for (int r = 1; r < world; ++r) {
        long long b = ((long long)n * r / world + 255) / 256 * 256;
        D.row_off[r] = (int)std::min<long long>(std::max<long long>(b, D.row_off[r - 1]), n);
    }
    D.row_off[world] = n;
    D.row0 = D.row_off[rank];
    D.nloc = D.row_off[rank + 1] - D.row0;
    D.own_off.resize((size_t)world + 1);
    for (int r = 0; r <= world; ++r) D.own_off[r] = 8ll * D.row_off[r];
    // columns referenced by the own rows outside the own range, ascending == grouped by owner
    DBuf<int> flag((size_t)n + 1), pos((size_t)n + 2);
    flag.zero(s);
    if (D.nloc)
        hipLaunchKernelGGL(halo_mark_kernel, dim3(div_up((long)D.nloc * 8, 256)), dim3(256), 0, s, D.row0,
                           D.nloc, L.A.rowptr.p, L.A.col.p, flag.p);
    SA_HIP_CHECK(hipGetLastError());
    exclusive_scan_int(s, n, flag.p, pos.p);
    int nrecv = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&nrecv, pos.p + n, sizeof(int), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    D.nrecv = nrecv;
    D.recv_idx.alloc((size_t)nrecv + 1);
    hipLaunchKernelGGL(halo_compact_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, flag.p, pos.p,
                       D.recv_idx.p);
    SA_HIP_CHECK(hipGetLastError());
    std::vector<long long> rcnt((size_t)world, 0);
    {
        DBuf<int> view;
        view.view(D.recv_idx.p, (size_t)nrecv);
        auto h_idx = view.to_host(s);
        int r = 0;
        for (int k = 0; k < nrecv; ++k) {
            while (h_idx[k] >= D.row_off[r + 1]) ++r;
            ++rcnt[r];
        }
    }
    SA_REQUIRE(rcnt[rank] == 0, "halo list contains own rows");
    // every rank learns how much every other rank needs from it
    std::vector<long long> M((size_t)world * world, 0), moff((size_t)world + 1);
    for (int r = 0; r < world; ++r) M[(size_t)rank * world + r] = rcnt[r];
    for (int r = 0; r <= world; ++r) moff[r] = 8ll * world * r;
    DBuf<long long> dM;
    dM.from_host(M, s);
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_REQUIRE(p.allgather(p.allgather_ctx, dM.p, moff.data()) == 0, "allgather callback failed");
    auto hM = dM.to_host(s);
    D.send_off.assign((size_t)world + 1, 0);
    D.recv_off.assign((size_t)world + 1, 0);
    std::vector<long long> ioff_s((size_t)world + 1, 0), ioff_r((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) {
        const long long sc = hM[(size_t)r * world + rank];
        D.send_off[r + 1] = D.send_off[r] + 8 * sc;
        D.recv_off[r + 1] = D.recv_off[r] + 8 * rcnt[r];
        ioff_s[r + 1] = ioff_s[r] + 4 * sc;
        ioff_r[r + 1] = ioff_r[r] + 4 * rcnt[r];
    }
    D.nsend = (int)(D.send_off[world] / 8);
    D.send_idx.alloc((size_t)D.nsend + 1);
    // tell each owner which of its entries are needed here
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_REQUIRE(p.alltoallv(p.allgather_ctx, D.recv_idx.p, ioff_r.data(), D.send_idx.p, ioff_s.data()) == 0,
               "alltoallv callback failed");
    {
        DBuf<int> view;
        view.view(D.send_idx.p, (size_t)D.nsend);
        auto h_idx = view.to_host(s);
        for (int k = 0; k < D.nsend; ++k)
            SA_REQUIRE(h_idx[k] >= D.row0 && h_idx[k] < D.row0 + D.nloc, "halo request outside the own rows");
    }
    D.send_buf.alloc((size_t)D.nsend + 1);
    D.recv_buf.alloc((size_t)D.nrecv + 1);
    // interior rows: the longest run of own slices without a halo column (a slab of a banded operator: everything
    // but the two ends); overlapped with the exchange when it is at least a quarter of the own rows
    // (saamge_amd_options.overlap bit 1 cleared: never)
    D.int_row0 = D.row0;
    D.int_nrows = 0;
    const bool no_overlap = !(options().overlap & 2);
    if (D.nloc > 0 && !no_overlap) {
        const int nsl = div_up(D.nloc, 64);
        DBuf<int> sflag((size_t)nsl);
        sflag.zero(s);
        hipLaunchKernelGGL(halo_slice_kernel, dim3(div_up((long)D.nloc * 8, 256)), dim3(256), 0, s, D.row0, D.nloc,
                           L.A.rowptr.p, L.A.col.p, sflag.p);
        SA_HIP_CHECK(hipGetLastError());
        auto hf = sflag.to_host(s);
        int best0 = 0, best = 0, run0 = 0;
        for (int i = 0; i <= nsl; ++i)
            if (i == nsl || hf[i]) {
                if (i - run0 > best) { best = i - run0; best0 = run0; }
                run0 = i + 1;
            }
        // (whole 256-row tiles: the three row ranges of an application then all take the staged kernel)
        const int end0 = best0 + best;
        const int a0 = (best0 + 3) / 4 * 4;
        const int a1 = (end0 == nsl) ? nsl : end0 / 4 * 4;
        if (a1 - a0 >= 1 && 4 * (a1 - a0) >= nsl) {
            D.int_row0 = D.row0 + 64 * a0;
            D.int_nrows = std::min(64 * (a1 - a0), D.row0 + D.nloc - D.int_row0);
        }
    }
    L.r.zero(s);  // the restriction R r sums over ranks: r must vanish outside the own rows
    SA_HIP_CHECK(hipStreamSynchronize(s));
    D.on = true;
    return true;
}

}  // namespace saamge_amd
