// General sparse products for the smoothed prolongator (reference: interp_smooth,
// amg/src/interp.cpp:172-229 -> hypre ParMult chains; tg_coarse_matr = mfem::RAP,
// amg/inc/tg.hpp:696-709):
//     C = beta E + alpha diag(d) A B        (row-wise Gustavson product with an LDS hash table)
//     R = P^T                               (count / scan / fill, rows sorted by column)
// One WAVEFRONT owns one output row and walks the entries of A's row one after the other; the
// lanes spread over the (distinct) columns of the matching row of B.  Contributions to one output
// entry therefore arrive in the fixed order of A's row and are added with plain LDS read-modify-
// writes: no floating-point atomics, results are run-to-run and rank-to-rank reproducible.
// The row is then sorted by column (wave-level bitonic in LDS) and written out.
#include "spgemm.h"

namespace saamge_amd {


constexpr int SPG_EMPTY = 0x7fffffff;

__device__ inline void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// slot of `key` in the open-addressing table (inserted if absent); -1 when the table is full
template <int T>
__device__ inline int hash_slot(int *keys, int key) {
    unsigned h = hash_home((unsigned)key, (unsigned)T);
    for (int probe = 0; probe < T; ++probe) {
        const int prev = atomicCAS(&keys[h], SPG_EMPTY, key);
        if (prev == SPG_EMPTY || prev == key) return (int)h;
        h = (h + 1) & (unsigned)(T - 1);
    }
    return -1;
}

// ascending wave-level bitonic sort of keys[0..T) (+ payload), T a power of two
template <int T>
__device__ inline void wave_sort(int *keys, double *vals, int lane) {
    for (int k = 2; k <= T; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < T; i += 64) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = keys[i], c = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) {
                        keys[i] = c; keys[ixj] = a;
                        if (vals) { const double t = vals[i]; vals[i] = vals[ixj]; vals[ixj] = t; }
                    }
                }
            }
            wave_lds_sync();
        }
}

// mode 0: rowcnt[i] = number of distinct columns of row i (or -1 on table overflow);
// mode 1: write row i (sorted) at Crow[i]
template <int T, int WPB>
__global__ __launch_bounds__(64 * WPB) void spgemm_kernel(
    int mode, int nrows, const roff_t *__restrict__ Arow, const int *__restrict__ Acol,
    const double *__restrict__ Aval, const roff_t *__restrict__ Brow, const int *__restrict__ Bcol,
    const double *__restrict__ Bval, const roff_t *__restrict__ Erow, const int *__restrict__ Ecol,
    const double *__restrict__ Eval, const double *__restrict__ d, double alpha, double beta,
    int *__restrict__ rowcnt, const roff_t *__restrict__ Crow, int *__restrict__ Ccol,
    double *__restrict__ Cval) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * WPB + w;
    if (row >= nrows) return;
    double *vals = (double *)smem + (size_t)w * T;
    int *keys = (int *)((double *)smem + (size_t)WPB * T) + (size_t)w * T;
    for (int i = lane; i < T; i += 64) { keys[i] = SPG_EMPTY; vals[i] = 0.0; }
    wave_lds_sync();
    int full = 0;
    if (Erow) {
        for (roff_t q = Erow[row] + lane; q < Erow[row + 1]; q += 64) {
            const int sl = hash_slot<T>(keys, Ecol[q]);
            if (sl < 0) full = 1;
            else vals[sl] += beta * Eval[q];
        }
        wave_lds_sync();
    }
    const double sc = alpha * (d ? d[row] : 1.0);
    for (roff_t p = Arow[row]; p < Arow[row + 1]; ++p) {
        const int k = Acol[p];
        const double a = sc * Aval[p];
        for (roff_t q = Brow[k] + lane; q < Brow[k + 1]; q += 64) {
            const int sl = hash_slot<T>(keys, Bcol[q]);
            if (sl < 0) full = 1;
            else vals[sl] = fma(a, Bval[q], vals[sl]);
        }
        wave_lds_sync();   // the next entry of A's row may hit the same slots
    }
    if (__ballot(full) != 0ull) {
        if (mode == 0 && lane == 0) rowcnt[row] = -1;
        return;
    }
    if (mode == 0) {
        int c = 0;
        for (int i = lane; i < T; i += 64) c += keys[i] != SPG_EMPTY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) rowcnt[row] = c;
        return;
    }
    wave_sort<T>(keys, vals, lane);
    const roff_t base = Crow[row];
    const int len = (int)(Crow[row + 1] - base);
    for (int i = lane; i < len; i += 64) {
        Ccol[base + i] = keys[i];
        Cval[base + i] = vals[i];
    }
}

template <int T, int WPB>
static void launch_spgemm(hipStream_t s, int mode, const DCsr &A, const DCsr &B, const DCsr *E, const double *d,
                          double alpha, double beta, int *rowcnt, const DCsr &C) {
    const size_t lds = (size_t)WPB * T * (sizeof(double) + sizeof(int));
    static bool attr = false;
    if (!attr && lds > 48 * 1024) {
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)spgemm_kernel<T, WPB>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((spgemm_kernel<T, WPB>), dim3(div_up(A.nrows, WPB)), dim3(64 * WPB), lds, s, mode, A.nrows,
                       A.rowptr.p, A.col.p, A.val.p, B.rowptr.p, B.col.p, B.val.p, E ? E->rowptr.p : nullptr,
                       E ? E->col.p : nullptr, E ? E->val.p : nullptr, d, alpha, beta, rowcnt, C.rowptr.p,
                       C.col.p, C.val.p);
    SA_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void min_int_kernel(int n, const int *__restrict__ v, int *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && v[i] < 0) atomicMin(out, -1);
}
__global__ __launch_bounds__(256) void clamp_nonneg_kernel(int n, int *__restrict__ v) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && v[i] < 0) v[i] = 0;
}

// ---- few, very long rows times a matrix with few columns (R (A P) on the coarse levels) --------
// B is expanded to a dense nB x ncols image (+ a byte mask of its structure); one workgroup owns a
// row of A, thread t owns columns t, t + 256, ... and walks the row of A serially: coalesced reads
// of B's rows, fixed summation order, output already sorted by column.
constexpr int SPD_MAXC = 2048;   // columns handled (8 per thread)

__global__ __launch_bounds__(256) void densify_kernel(int nrows, int ncols, const roff_t *__restrict__ rowptr,
                                                      const int *__restrict__ col, const double *__restrict__ val,
                                                      double *__restrict__ Bd, unsigned char *__restrict__ Bm) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(gt >> 3), lane = (int)(gt & 7);
    if (row >= nrows) return;
    for (roff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 8) {
        Bd[(size_t)row * ncols + col[k]] = val[k];
        Bm[(size_t)row * ncols + col[k]] = 1;
    }
}

__global__ __launch_bounds__(256) void spgemm_dense_b_kernel(int mode, int ncols, const roff_t *__restrict__ Arow,
                                                             const int *__restrict__ Acol,
                                                             const double *__restrict__ Aval,
                                                             const double *__restrict__ Bd,
                                                             const unsigned char *__restrict__ Bm,
                                                             int *__restrict__ rowcnt, const roff_t *__restrict__ Crow,
                                                             int *__restrict__ Ccol, double *__restrict__ Cval) {
    __shared__ int wsum[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    double acc[SPD_MAXC / 256];
    int hit[SPD_MAXC / 256];
#pragma unroll
    for (int u = 0; u < SPD_MAXC / 256; ++u) { acc[u] = 0.0; hit[u] = 0; }
    for (roff_t p = Arow[row]; p < Arow[row + 1]; ++p) {
        const size_t base = (size_t)Acol[p] * ncols;
        const double a = Aval[p];
#pragma unroll
        for (int u = 0; u < SPD_MAXC / 256; ++u) {
            const int j = tid + 256 * u;
            if (j < ncols) {
                acc[u] = fma(a, Bd[base + j], acc[u]);
                hit[u] |= Bm[base + j];
            }
        }
    }
    // positions of the structurally present columns, in column order: columns j = tid + 256 u are
    // ordered by (u, tid), so scan per u over the threads
    int before = 0;
    for (int u = 0; u < SPD_MAXC / 256; ++u) {
        int incl = hit[u];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += v;
        }
        __syncthreads();
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < (tid >> 6)) off += wsum[w]; tot += wsum[w]; }
        if (mode == 1 && hit[u]) {
            const roff_t pos = Crow[row] + before + off + incl - 1;
            Ccol[pos] = tid + 256 * u;
            Cval[pos] = acc[u];
        }
        before += tot;
    }
    if (mode == 0 && tid == 0) rowcnt[row] = before;
}

static bool spgemm_dense_b(hipStream_t s, const DCsr &A, const DCsr &B, DCsr &C) {
    const int n = A.nrows;
    if (B.ncols > SPD_MAXC || n == 0) return false;
    const size_t cells = (size_t)B.nrows * B.ncols;
    if (cells > ((size_t)1 << 27)) return false;                 // 1 GiB of doubles
    if ((double)A.nnz < 256.0 * n) return false;                  // only worth it for long rows
    DBuf<double> Bd(cells);
    DBuf<unsigned char> Bm(cells);
    Bd.zero(s);
    Bm.zero(s);
    hipLaunchKernelGGL(densify_kernel, dim3(div_up((long)B.nrows * 8, 256)), dim3(256), 0, s, B.nrows, B.ncols,
                       B.rowptr.p, B.col.p, B.val.p, Bd.p, Bm.p);
    DBuf<int> rowcnt((size_t)n);
    hipLaunchKernelGGL(spgemm_dense_b_kernel, dim3(n), dim3(256), 0, s, 0, B.ncols, A.rowptr.p, A.col.p, A.val.p,
                       Bd.p, Bm.p, rowcnt.p, nullptr, nullptr, nullptr);
    exclusive_scan_off(s, n, rowcnt.p, C.rowptr.p);
    roff_t nnz = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&nnz, C.rowptr.p + n, sizeof(roff_t), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    C.nnz = nnz;
    C.col.alloc((size_t)nnz + 1);
    C.val.alloc((size_t)nnz + 1);
    hipLaunchKernelGGL(spgemm_dense_b_kernel, dim3(n), dim3(256), 0, s, 1, B.ncols, A.rowptr.p, A.col.p, A.val.p,
                       Bd.p, Bm.p, rowcnt.p, C.rowptr.p, C.col.p, C.val.p);
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));
    C.lanes_per_row = pick_lanes_per_row(C.nnz, n);
    return true;
}

void spgemm(hipStream_t s, const DCsr &A, const DCsr &B, const DCsr *E, const double *d, double alpha,
            double beta, DCsr &C) {
    SA_REQUIRE(A.ncols == B.nrows, "spgemm: inner dimensions differ");
    const int n = A.nrows;
    C.nrows = n;
    C.ncols = B.ncols;
    C.nnz = 0;
    C.has_sell = false;
    C.max_row = -1;
    C.rowptr.alloc((size_t)n + 1);
    if (n == 0) return;
    if (!E && !d && alpha == 1.0 && beta == 0.0) {
        profiler().begin(s);
        const bool done = spgemm_dense_b(s, A, B, C);
        profiler().end(s, "spgemm_dense_b", 0.0, 0.0);
        if (done) return;
    }
    DBuf<int> rowcnt((size_t)n), flag(1);
    profiler().begin(s);
    // small tables first (4 rows per workgroup); rows that overflow them are redone with big ones
    int tier = 0;
    for (; tier < 3; ++tier) {
        if (tier == 0) launch_spgemm<256, 4>(s, 0, A, B, E, d, alpha, beta, rowcnt.p, C);
        else if (tier == 1) launch_spgemm<2048, 2>(s, 0, A, B, E, d, alpha, beta, rowcnt.p, C);
        else launch_spgemm<8192, 1>(s, 0, A, B, E, d, alpha, beta, rowcnt.p, C);
        flag.zero(s);
        hipLaunchKernelGGL(min_int_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, rowcnt.p, flag.p);
        if (flag.to_host(s)[0] == 0) break;
    }
    SA_REQUIRE(tier < 3, "spgemm: a product row has more than ~8000 entries");
    exclusive_scan_off(s, n, rowcnt.p, C.rowptr.p);
    roff_t nnz = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&nnz, C.rowptr.p + n, sizeof(roff_t), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    C.nnz = nnz;
    C.col.alloc((size_t)nnz + 1);
    C.val.alloc((size_t)nnz + 1);
    if (tier == 0) launch_spgemm<256, 4>(s, 1, A, B, E, d, alpha, beta, rowcnt.p, C);
    else if (tier == 1) launch_spgemm<2048, 2>(s, 1, A, B, E, d, alpha, beta, rowcnt.p, C);
    else launch_spgemm<8192, 1>(s, 1, A, B, E, d, alpha, beta, rowcnt.p, C);
    SA_HIP_CHECK(hipStreamSynchronize(s));   // rowcnt is freed on return
    profiler().end(s, "spgemm", 12.0 * ((double)A.nnz + B.nnz + nnz), 0.0);
    C.lanes_per_row = pick_lanes_per_row(C.nnz, n);
}

// ---- transpose --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tr_count_kernel(long nnz, const int *__restrict__ col, int *__restrict__ cnt) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k < nnz) atomicAdd(&cnt[col[k]], 1);
}
__global__ __launch_bounds__(256) void tr_fill_kernel(int nrows, const roff_t *__restrict__ rowptr,
                                                      const int *__restrict__ col, const double *__restrict__ val,
                                                      const roff_t *__restrict__ Trow, int *__restrict__ cursor,
                                                      int *__restrict__ Tcol, double *__restrict__ Tval) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(gt >> 2), lane = (int)(gt & 3);
    if (row >= nrows) return;
    for (roff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 4) {
        const int c = col[k];
        const roff_t pos = Trow[c] + atomicAdd(&cursor[c], 1);
        Tcol[pos] = row;
        Tval[pos] = val[k];
    }
}
// order every row by column: (col_in, val_in) hold the rows in arbitrary order (atomic fill),
// (col, val) receive them sorted.  One wavefront per row; rows up to T entries are sorted in LDS
// (bitonic), longer ones (a coarse dof supported on a large part of a small level) by rank
// counting: the position of an entry is the number of smaller keys in its row.
template <int T>
__global__ __launch_bounds__(64) void row_order_kernel(int nrows, const roff_t *__restrict__ rowptr,
                                                       const int *__restrict__ col_in,
                                                       const double *__restrict__ val_in, int *__restrict__ col,
                                                       double *__restrict__ val) {
    __shared__ int keys[T];
    __shared__ double vals[T];
    const int row = blockIdx.x, lane = threadIdx.x;
    const roff_t b = rowptr[row];
    const int len = (int)(rowptr[row + 1] - b);
    if (len == 0) return;
    if (len > T) {
        for (int i = lane; i < len; i += 64) {
            const int key = col_in[b + i];
            int rank = 0;
            for (int j = 0; j < len; ++j) rank += col_in[b + j] < key;
            col[b + rank] = key;
            val[b + rank] = val_in[b + i];
        }
        return;
    }
    for (int i = lane; i < T; i += 64) {
        keys[i] = (i < len) ? col_in[b + i] : SPG_EMPTY;
        vals[i] = (i < len) ? val_in[b + i] : 0.0;
    }
    wave_lds_sync();
    wave_sort<T>(keys, vals, lane);
    for (int i = lane; i < len; i += 64) {
        col[b + i] = keys[i];
        val[b + i] = vals[i];
    }
}

void csr_transpose(hipStream_t s, const DCsr &P, DCsr &R) {
    R.nrows = P.ncols;
    R.ncols = P.nrows;
    R.nnz = P.nnz;
    R.has_sell = false;
    R.max_row = -1;
    R.rowptr.alloc((size_t)R.nrows + 1);
    R.col.alloc((size_t)P.nnz + 1);
    R.val.alloc((size_t)P.nnz + 1);
    if (R.nrows == 0) return;
    DBuf<int> cnt((size_t)R.nrows), tcol((size_t)P.nnz + 1);
    DBuf<double> tval((size_t)P.nnz + 1);
    cnt.zero(s);
    hipLaunchKernelGGL(tr_count_kernel, dim3(div_up(P.nnz, 256)), dim3(256), 0, s, (long)P.nnz, P.col.p, cnt.p);
    exclusive_scan_off(s, R.nrows, cnt.p, R.rowptr.p);
    cnt.zero(s);
    hipLaunchKernelGGL(tr_fill_kernel, dim3(div_up((long)P.nrows * 4, 256)), dim3(256), 0, s, P.nrows, P.rowptr.p,
                       P.col.p, P.val.p, R.rowptr.p, cnt.p, tcol.p, tval.p);
    hipLaunchKernelGGL((row_order_kernel<2048>), dim3(R.nrows), dim3(64), 0, s, R.nrows, R.rowptr.p, tcol.p,
                       tval.p, R.col.p, R.val.p);
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));   // cnt is freed on return
    R.lanes_per_row = pick_lanes_per_row(R.nnz, R.nrows);
}

// ---------------------------------------------------------------------------------------
// thresholding (AltThreshold, amg/src/interp.cpp:89-170): keep entries with |v| > tol
// ---------------------------------------------------------------------------------------
// One wavefront per row; the kept entries stay in their order (ballot + prefix popcount).
template <bool FILL>
__global__ __launch_bounds__(256) void threshold_kernel(int nrows, double tol, const roff_t *__restrict__ rowptr,
                                                        const int *__restrict__ col, const double *__restrict__ val,
                                                        int *__restrict__ cnt, const roff_t *__restrict__ orow,
                                                        int *__restrict__ ocol, double *__restrict__ oval) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const roff_t b = rowptr[row], e = rowptr[row + 1];
    int kept = 0;
    for (roff_t p0 = b; p0 < e; p0 += 64) {
        const roff_t p = p0 + lane;
        const double v = (p < e) ? val[p] : 0.0;
        const bool keep = (p < e) && (fabs(v) > tol);
        const unsigned long long m = __ballot(keep);
        if (FILL && keep) {
            const roff_t dst = orow[row] + kept + __popcll(m & ((1ull << lane) - 1ull));
            ocol[dst] = col[p];
            oval[dst] = v;
        }
        kept += __popcll(m);
    }
    if (!FILL && lane == 0) cnt[row] = kept;
}

void csr_threshold(hipStream_t s, const DCsr &A, double tol, DCsr &C) {
    C.nrows = A.nrows;
    C.ncols = A.ncols;
    C.has_sell = false;
    C.max_row = -1;
    C.rowptr.alloc((size_t)A.nrows + 1);
    DBuf<int> cnt((size_t)A.nrows + 1);
    const dim3 grid(div_up(A.nrows, 4));
    if (A.nrows > 0)
        hipLaunchKernelGGL((threshold_kernel<false>), grid, dim3(256), 0, s, A.nrows, tol, A.rowptr.p, A.col.p,
                           A.val.p, cnt.p, nullptr, nullptr, nullptr);
    exclusive_scan_off(s, A.nrows, cnt.p, C.rowptr.p);
    roff_t nnz = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&nnz, C.rowptr.p + A.nrows, sizeof(roff_t), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    C.nnz = nnz;
    C.col.alloc((size_t)nnz + 1);
    C.val.alloc((size_t)nnz + 1);
    if (A.nrows > 0)
        hipLaunchKernelGGL((threshold_kernel<true>), grid, dim3(256), 0, s, A.nrows, tol, A.rowptr.p, A.col.p,
                           A.val.p, nullptr, C.rowptr.p, C.col.p, C.val.p);
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));
    C.lanes_per_row = pick_lanes_per_row(C.nnz, C.nrows);
}

}  // namespace saamge_amd
