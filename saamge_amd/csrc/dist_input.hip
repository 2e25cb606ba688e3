// Per-rank inputs (one process per GPU): what the reference's multi-rank drivers hand to ml_produce_data -- a
// HypreParMatrix, i.e. hypre's ParCSR split of the rank's row block into `diag` (the columns of the rank's own range,
// local indices) and `offd` (the others, compressed through col_map_offd), the rank's OWN elements with their matrices
// and the rank's own agglomerate partition (agg_create_partitioning_fine, amg/src/aggregates.cpp:1340-1443; pmltest,
// amg/CMakeLists.txt:198-203).
//
// What happens to them here (round 4; DESIGN.md section 6 lists what is still replicated): the integer topology
// (elem_to_dof, partitions, boundary flags) and the operator's rows are all-gathered into the replicated form the setup
// works on -- rows merged and sorted by global column --; the ELEMENT MATRICES, the largest input (8 nde^2 bytes per
// element against 12 bytes per stored entry of A), stay on the rank that passed them and are never exchanged: a rank
// assembles, factors and coarsens exactly the agglomerates made of its own elements (the reference's invariant: no
// agglomerate straddles ranks), so the agglomerate ownership ranges of every level follow the inputs.
#include "hierarchy.h"

namespace saamge_amd {

namespace {

__global__ __launch_bounds__(256) void parcsr_rowlen_kernel(int nloc, const int *__restrict__ di, const int *__restrict__ oi,
                                                            int *__restrict__ len) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < nloc) len[r] = (di[r + 1] - di[r]) + (oi ? oi[r + 1] - oi[r] : 0);
}

// one thread per own row: the diag entries (local column + first own column) and the offd entries (col_map_offd) written
// behind each other at the row's place in the global arrays, then sorted by global column (insertion sort: hypre keeps the
// diagonal entry first and the rest in assembly order, the setup's kernels expect ascending columns)
__global__ __launch_bounds__(256) void parcsr_merge_kernel(int nloc, int col0, const int *__restrict__ di,
                                                           const int *__restrict__ dj, const double *__restrict__ da,
                                                           const int *__restrict__ oi, const int *__restrict__ oj,
                                                           const double *__restrict__ oa, const long long *__restrict__ cmap,
                                                           const roff_t *__restrict__ rowptr, int *__restrict__ col,
                                                           double *__restrict__ val) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nloc) return;
    const roff_t base = rowptr[r];
    int len = 0;
    for (int k = di[r]; k < di[r + 1]; ++k, ++len) { col[base + len] = col0 + dj[k]; val[base + len] = da[k]; }
    if (oi)
        for (int k = oi[r]; k < oi[r + 1]; ++k, ++len) { col[base + len] = (int)cmap[oj[k]]; val[base + len] = oa[k]; }
    for (int i = 1; i < len; ++i) {
        const int c = col[base + i];
        const double v = val[base + i];
        int j = i - 1;
        while (j >= 0 && col[base + j] > c) {
            col[base + j + 1] = col[base + j];
            val[base + j + 1] = val[base + j];
            --j;
        }
        col[base + j + 1] = c;
        val[base + j + 1] = v;
    }
}

__global__ __launch_bounds__(256) void add_offset_kernel(long n, int off, const int *__restrict__ src, int *__restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] + off;
}

void gather(const Params &p, hipStream_t s, void *buf, const std::vector<long long> &byte_off) {
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_REQUIRE(p.allgather(p.allgather_ctx, buf, byte_off.data()) == 0, "all-gather callback failed (per-rank inputs)");
}

}  // namespace

Hierarchy *hierarchy_create_dist(const ParCsrIn &A, int NE_loc, int nde, const int *elem_to_dof, const double *elmat,
                                 const signed char *bdr_own, const int *const *partitions, const int *nparts_loc,
                                 const Params &p, hipStream_t s) {
    const int world = p.world > 1 ? p.world : 1, rank = world > 1 ? p.rank : 0;
    SA_REQUIRE(world == 1 || p.allgather != nullptr, "per-rank inputs on several ranks need the all-gather primitive");
    SA_REQUIRE(!p.algebraic, "per-rank inputs: the element-free mode is single-rank only");
    SA_REQUIRE(A.nrows >= 0 && NE_loc >= 0 && nde > 0, "bad per-rank sizes");
    SA_REQUIRE(p.num_coarsenings >= 1 && p.num_coarsenings < MAX_LEVELS, "bad number of coarsenings");
    const int nc = p.num_coarsenings;
    std::unique_ptr<DistIn> din(new DistIn);
    // ---- sizes of every rank ----
    const int K = 3 + MAX_LEVELS;
    std::vector<long long> cnt((size_t)world * K, 0), coff((size_t)world + 1);
    {
        DBuf<int> di_v, oi_v;
        import_array(di_v, A.diag_i, (size_t)A.nrows + 1, s);
        int dn = 0, on = 0;
        SA_HIP_CHECK(hipMemcpyAsync(&dn, di_v.p + A.nrows, sizeof(int), hipMemcpyDeviceToHost, s));
        if (A.offd_i) {
            import_array(oi_v, A.offd_i, (size_t)A.nrows + 1, s);
            SA_HIP_CHECK(hipMemcpyAsync(&on, oi_v.p + A.nrows, sizeof(int), hipMemcpyDeviceToHost, s));
        }
        SA_HIP_CHECK(hipStreamSynchronize(s));
        long long *mine = cnt.data() + (size_t)rank * K;
        mine[0] = A.nrows;
        mine[1] = (long long)dn + on;
        mine[2] = NE_loc;
        for (int l = 0; l < nc; ++l) mine[3 + l] = nparts_loc[l];
    }
    if (world > 1) {
        DBuf<long long> d;
        d.from_host(cnt, s);
        for (int r = 0; r <= world; ++r) coff[r] = 8ll * K * r;
        gather(p, s, d.p, coff);
        auto h = d.to_host(s);
        cnt.assign(h.begin(), h.end());
    }
    std::vector<long long> row_off((size_t)world + 1, 0), nnz_off((size_t)world + 1, 0), el_off((size_t)world + 1, 0);
    std::vector<std::vector<long long>> ae_off((size_t)nc, std::vector<long long>((size_t)world + 1, 0));
    for (int r = 0; r < world; ++r) {
        row_off[r + 1] = row_off[r] + cnt[(size_t)r * K];
        nnz_off[r + 1] = nnz_off[r] + cnt[(size_t)r * K + 1];
        el_off[r + 1] = el_off[r] + cnt[(size_t)r * K + 2];
        for (int l = 0; l < nc; ++l) ae_off[l][r + 1] = ae_off[l][r] + cnt[(size_t)r * K + 3 + l];
    }
    const long long n = row_off[world], NE = el_off[world];
    SA_REQUIRE(n > 0 && n < (1ll << 31) && NE > 0 && NE < (1ll << 31), "per-rank inputs: global sizes out of range");
    if (A.row_starts)
        for (int r = 0; r <= world; ++r)
            SA_REQUIRE(A.row_starts[r] == row_off[r], "row_starts does not match the ranks' row counts (contiguous row blocks in rank order)");
    SA_REQUIRE(A.global_rows <= 0 || A.global_rows == n, "global_rows does not match the sum of the ranks' rows");
    const int row0 = (int)row_off[rank];
    std::vector<long long> boff((size_t)world + 1);
    // ---- the operator: row lengths -> global row offsets -> merged, sorted rows of the own block -> all-gather ----
    {
        DBuf<int> di, dj, oi, oj;
        DBuf<double> da, oa;
        DBuf<long long> cmap;
        const long long dn = cnt[(size_t)rank * K + 1];
        import_array(di, A.diag_i, (size_t)A.nrows + 1, s);
        int dnnz = 0;
        SA_HIP_CHECK(hipMemcpyAsync(&dnnz, di.p + A.nrows, sizeof(int), hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
        import_array(dj, A.diag_j, (size_t)dnnz, s);
        import_array(da, A.diag_a, (size_t)dnnz, s);
        const long long onnz = dn - dnnz;
        if (A.offd_i) {
            import_array(oi, A.offd_i, (size_t)A.nrows + 1, s);
            import_array(oj, A.offd_j, (size_t)onnz, s);
            import_array(oa, A.offd_a, (size_t)onnz, s);
            import_array(cmap, A.col_map_offd, (size_t)A.num_cols_offd, s);
        }
        DBuf<int> len((size_t)n + 1);
        len.zero(s);
        if (A.nrows)
            hipLaunchKernelGGL(parcsr_rowlen_kernel, dim3(div_up(A.nrows, 256)), dim3(256), 0, s, A.nrows, di.p,
                               A.offd_i ? oi.p : (const int *)nullptr, len.p + row0);
        SA_HIP_CHECK(hipGetLastError());
        if (world > 1) {
            for (int r = 0; r <= world; ++r) boff[r] = 4ll * row_off[r];
            gather(p, s, len.p, boff);
        }
        din->rowptr.alloc((size_t)n + 1);
        exclusive_scan_off(s, (int)n, len.p, din->rowptr.p);
        const long long nnz = nnz_off[world];
        din->col.alloc((size_t)nnz);
        din->val.alloc((size_t)nnz);
        if (A.nrows)
            hipLaunchKernelGGL(parcsr_merge_kernel, dim3(div_up(A.nrows, 256)), dim3(256), 0, s, A.nrows, row0, di.p, dj.p, da.p,
                               A.offd_i ? oi.p : (const int *)nullptr, oj.p, oa.p, cmap.p, din->rowptr.p + row0, din->col.p,
                               din->val.p);
        SA_HIP_CHECK(hipGetLastError());
        if (world > 1) {
            for (int r = 0; r <= world; ++r) boff[r] = 4ll * nnz_off[r];
            gather(p, s, din->col.p, boff);
            for (int r = 0; r <= world; ++r) boff[r] = 8ll * nnz_off[r];
            gather(p, s, din->val.p, boff);
        }
        SA_HIP_CHECK(hipStreamSynchronize(s));      // (the views of the caller's arrays end here)
    }
    // ---- integer topology: elem_to_dof (global dof ids), boundary flags of the own rows, partitions ----
    {
        din->e2d.alloc((size_t)NE * nde);
        if (NE_loc)
            SA_HIP_CHECK(hipMemcpyAsync(din->e2d.p + (size_t)el_off[rank] * nde, elem_to_dof, sizeof(int) * (size_t)NE_loc * nde,
                                        hipMemcpyDefault, s));
        if (world > 1) {
            for (int r = 0; r <= world; ++r) boff[r] = 4ll * nde * el_off[r];
            gather(p, s, din->e2d.p, boff);
        }
        din->bdr.alloc((size_t)n);
        din->bdr.zero(s);
        if (bdr_own && A.nrows)
            SA_HIP_CHECK(hipMemcpyAsync(din->bdr.p + row0, bdr_own, (size_t)A.nrows, hipMemcpyDefault, s));
        if (world > 1) {
            for (int r = 0; r <= world; ++r) boff[r] = row_off[r];
            gather(p, s, din->bdr.p, boff);
        }
        din->parts.resize((size_t)nc);
        din->nparts.resize((size_t)nc);
        din->ae_begin.resize((size_t)nc);
        for (int l = 0; l < nc; ++l) {
            const std::vector<long long> &src_off = l == 0 ? el_off : ae_off[l - 1];     // level-l elements = level-(l-1) agglomerates
            const long long mine = src_off[rank + 1] - src_off[rank], all = src_off[world];
            SA_REQUIRE(ae_off[l][world] > 0 && ae_off[l][world] < (1ll << 31), "per-rank inputs: no agglomerates on a level");
            din->parts[l].alloc((size_t)all);
            if (mine) {
                DBuf<int> loc;
                import_array(loc, partitions[l], (size_t)mine, s);
                hipLaunchKernelGGL(add_offset_kernel, dim3(div_up(mine, 256)), dim3(256), 0, s, (long)mine, (int)ae_off[l][rank],
                                   loc.p, din->parts[l].p + src_off[rank]);
                SA_HIP_CHECK(hipGetLastError());
                SA_HIP_CHECK(hipStreamSynchronize(s));
            }
            if (world > 1) {
                for (int r = 0; r <= world; ++r) boff[r] = 4ll * src_off[r];
                gather(p, s, din->parts[l].p, boff);
            }
            din->nparts[l] = (int)ae_off[l][world];
            din->ae_begin[l].assign(ae_off[l].begin(), ae_off[l].end());
        }
    }
    din->elem0 = (int)el_off[rank];
    din->NE_loc = NE_loc;
    std::vector<const int *> parts((size_t)nc);
    for (int l = 0; l < nc; ++l) parts[l] = din->parts[l].p;
    const int *e2d_p = din->e2d.p;
    const signed char *bdr_p = din->bdr.p;
    const void *rowptr_p = din->rowptr.p;
    const int *col_p = din->col.p;
    const double *val_p = din->val.p;
    const int *nparts_p = din->nparts.data();
    return hierarchy_create((int)n, rowptr_p, 64, col_p, val_p, (int)NE, nde, e2d_p, elmat, bdr_p, parts.data(), nparts_p, p, s,
                            std::move(din));
}

}  // namespace saamge_amd
