// Direct coarsest solver beyond the reach of ONE explicit dense inverse (dense.hip: 8 n^2 bytes, n^3 flops): the
// reference's serial `--coarse-direct` (UMFPACK through MFEM, amg/src/tg.cpp:979-1014 -> HypreDirect) on a coarsest
// operator of tens of thousands of rows (BASELINE config 2 as stated: two levels, 67 975 rows).
//
// A level structure of the operator's graph (breadth-first search: an entry only ever couples a level with itself and
// its two neighbours) makes the operator BLOCK TRIDIAGONAL; block Gaussian elimination then needs one dense Schur
// complement per level,
//     S_0 = A_00,     S_k = A_kk - A_{k,k-1} S_{k-1}^-1 A_{k-1,k},
// and each of them is inverted explicitly by the block Gauss-Jordan kernels of dense.hip (matrix cores).  A solve is a
// forward and a backward sweep over the levels, per level one sparse coupling product and one dense symmetric
// matrix-vector product -- bandwidth work (8 sum n_k^2 bytes per sweep), no triangular chains:
//     z_k = S_k^-1 (b_k - A_{k,k-1} z_{k-1}),        x_K = z_K,   x_k = z_k - S_k^-1 A_{k,k+1} x_{k+1},
// followed by one step of iterative refinement on the original operator (the explicit inverses carry an error of
// ~cond(S_k) eps).  The level structure starts from a pseudo-peripheral vertex or from the far end SET of such a search
// (a box-shaped coarse grid: whole planes instead of shells around a corner), whichever needs fewer flops.
// Everything that touches numbers runs on the device; the graph search (integers, once per setup) runs on the host.
#include "blocktri.h"
#include "dense.h"
#include "sparse.h"

#include <algorithm>
#include <cmath>

namespace saamge_amd {

namespace {

constexpr int BT_MAX_BLOCK = 12288;     // rows of one level: its Schur row must fit LDS (96 KB), its inverse 1.2 GB
constexpr int BT_MIN_BLOCK = 256;       // smaller neighbouring levels are merged (the first shells around a start vertex)

// levels of a breadth-first search from the set `start`; returns the number of levels, -1 for an empty graph.  Vertices
// the search does not reach (another connected component) continue the numbering: no entry couples them to the rest.
int bfs_levels(int n, const roff_t *I, const int *J, const std::vector<int> &start, std::vector<int> &lev,
               std::vector<int> &order) {
    lev.assign((size_t)n, -1);
    order.clear();
    order.reserve((size_t)n);
    int top = -1;
    size_t head = 0;
    for (int v : start)
        if (lev[v] < 0) { lev[v] = 0; order.push_back(v); }
    int seed = 0;
    for (;;) {
        while (head < order.size()) {
            const int u = order[head++];
            top = std::max(top, lev[u]);
            for (roff_t k = I[u]; k < I[u + 1]; ++k) {
                const int v = J[k];
                if (lev[v] < 0) { lev[v] = lev[u] + 1; order.push_back(v); }
            }
        }
        while (seed < n && lev[seed] >= 0) ++seed;
        if (seed >= n) break;
        lev[seed] = top + 1;
        order.push_back(seed);
    }
    return top + 1;
}

double structure_cost(int nlev, const std::vector<int> &lev, std::vector<int> &sizes) {
    sizes.assign((size_t)nlev, 0);
    for (int l : lev) ++sizes[l];
    double c = 0.0;
    for (int sz : sizes) c += (double)sz * sz * sz;
    return c;
}

// one workgroup per row i of block k: T = A_{k,k-1}[i, :] S_{k-1}^-1 into LDS (threads over the columns: the rows of the
// symmetric inverse are contiguous), then S[i, i'] -= T . A_{k,k-1}[i', :] for every row i' of the block (threads over i').
__global__ __launch_bounds__(256) void bt_schur_kernel(int nk, int m, int row0, int col0, const roff_t *__restrict__ lo_ptr,
                                                       const int *__restrict__ lo_col, const double *__restrict__ lo_val,
                                                       const double *__restrict__ Sprev, double *__restrict__ S) {
    extern __shared__ double T[];
    const int i = blockIdx.x;
    const roff_t e0 = lo_ptr[row0 + i], e1 = lo_ptr[row0 + i + 1];
    for (int c = threadIdx.x; c < m; c += 256) {
        double t = 0.0;
        for (roff_t e = e0; e < e1; ++e) t = fma(lo_val[e], Sprev[(size_t)(lo_col[e] - col0) * m + c], t);
        T[c] = t;
    }
    __syncthreads();
    if (e0 == e1) return;
    for (int ip = threadIdx.x; ip < nk; ip += 256) {
        double acc = 0.0;
        for (roff_t e = lo_ptr[row0 + ip]; e < lo_ptr[row0 + ip + 1]; ++e) acc = fma(T[lo_col[e] - col0], lo_val[e], acc);
        S[(size_t)i * nk + ip] -= acc;
    }
}

__global__ __launch_bounds__(256) void bt_gather_kernel(int n, const int *__restrict__ perm, const double *__restrict__ src,
                                                        double *__restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// x[perm[i]] = (add ? x[perm[i]] : 0) + src[i]
__global__ __launch_bounds__(256) void bt_scatter_kernel(int n, const int *__restrict__ perm, const double *__restrict__ src,
                                                         double *__restrict__ x, int add) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[perm[i]] = (add ? x[perm[i]] : 0.0) + src[i];
}
// rows [row0, row0 + nk): out[i] = (base ? base[i] : 0) - sum_e val_e v[col_e]; eight lanes per row
__global__ __launch_bounds__(256) void bt_couple_kernel(int nk, int row0, const roff_t *__restrict__ ptr, const int *__restrict__ col,
                                                        const double *__restrict__ val, const double *__restrict__ v,
                                                        const double *__restrict__ base, double *__restrict__ out) {
    const int gt = blockIdx.x * 256 + threadIdx.x, r = gt >> 3, l = gt & 7;
    if (r >= nk) return;
    const int row = row0 + r;
    double s = 0.0;
    for (roff_t e = ptr[row] + l; e < ptr[row + 1]; e += 8) s = fma(val[e], v[col[e]], s);
    s += __shfl_xor(s, 1, 8);
    s += __shfl_xor(s, 2, 8);
    s += __shfl_xor(s, 4, 8);
    if (l == 0) out[row] = (base ? base[row] : 0.0) - s;
}
// out = (base ? base : 0) + sign X v, X symmetric n x n: one wavefront per entry walks its column
__global__ __launch_bounds__(256) void bt_symv_kernel(int n, const double *__restrict__ X, const double *__restrict__ v,
                                                      const double *__restrict__ base, double sign, double *__restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const double *colp = X + (size_t)row * n;
    double s0 = 0.0, s1 = 0.0;
    int i = lane;
    for (; i + 64 < n; i += 128) {
        s0 = fma(colp[i], v[i], s0);
        s1 = fma(colp[i + 64], v[i + 64], s1);
    }
    if (i < n) s0 = fma(colp[i], v[i], s0);
    double sm = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if (lane == 0) out[row] = (base ? base[row] : 0.0) + sign * sm;
}

}  // namespace

bool blocktri_factor(hipStream_t s, const DCsr &A, BlockTri &B) {
    B = BlockTri();
    const int n = A.nrows;
    if (n <= 0) return false;
    auto I = A.rowptr.to_host(s);
    auto J = A.col.to_host(s);
    auto V = A.val.to_host(s);
    // ---- level structure (host, integers only) ----
    std::vector<int> lev, order, lev2, order2, sizes, sizes2;
    int nlev = bfs_levels(n, I.data(), J.data(), std::vector<int>(1, 0), lev, order);
    {   // a pseudo-peripheral start: a vertex of smallest degree in the last level of a search from vertex 0
        int best = order.back();
        for (size_t q = order.size(); q-- > 0 && lev[order[q]] == nlev - 1;)
            if (I[order[q] + 1] - I[order[q]] < I[best + 1] - I[best]) best = order[q];
        nlev = bfs_levels(n, I.data(), J.data(), std::vector<int>(1, best), lev, order);
    }
    double cost = structure_cost(nlev, lev, sizes);
    {   // ... or the whole far end of that search as the start set (planes instead of shells around a corner)
        std::vector<int> far;
        for (int v = 0; v < n; ++v)
            if (lev[v] == nlev - 1) far.push_back(v);
        const int nlev2 = bfs_levels(n, I.data(), J.data(), far, lev2, order2);
        const double cost2 = structure_cost(nlev2, lev2, sizes2);
        if (cost2 < cost) { lev.swap(lev2); sizes.swap(sizes2); nlev = nlev2; cost = cost2; }
    }
    // blocks: consecutive levels, merged while they stay small
    std::vector<int> blk_of_lev((size_t)nlev, 0);
    int nblk = 0, acc = 0;
    for (int l = 0; l < nlev; ++l) {
        blk_of_lev[l] = nblk;
        acc += sizes[l];
        if (acc >= BT_MIN_BLOCK) { ++nblk; acc = 0; }
    }
    if (acc > 0) {
        if (nblk > 0) { for (int l = 0; l < nlev; ++l) if (blk_of_lev[l] == nblk) blk_of_lev[l] = nblk - 1; }
        else nblk = 1;
    }
    std::vector<int> off((size_t)nblk + 1, 0);
    for (int l = 0; l < nlev; ++l) off[(size_t)blk_of_lev[l] + 1] += sizes[l];
    int maxb = 0;
    for (int k = 0; k < nblk; ++k) { maxb = std::max(maxb, off[k + 1]); off[(size_t)k + 1] += off[k]; }
    if (maxb > BT_MAX_BLOCK) return false;
    std::vector<int> perm((size_t)n), iperm((size_t)n), fill(off.begin(), off.end() - 1);
    for (int v = 0; v < n; ++v) {       // stable inside a block: ascending original index
        const int p = fill[blk_of_lev[lev[v]]]++;
        perm[p] = v;
        iperm[v] = p;
    }
    // ---- the three parts of every permuted row: previous block, own block (local columns), next block ----
    hvec<roff_t> lo_ptr((size_t)n + 1, 0), up_ptr((size_t)n + 1, 0), dg_ptr((size_t)n + 1, 0);
    for (int p = 0; p < n; ++p) {
        const int v = perm[p], k = blk_of_lev[lev[v]];
        for (roff_t e = I[v]; e < I[v + 1]; ++e) {
            const int kc = blk_of_lev[lev[J[e]]];
            if (kc == k) ++dg_ptr[(size_t)p + 1];
            else if (kc == k - 1) ++lo_ptr[(size_t)p + 1];
            else if (kc == k + 1) ++up_ptr[(size_t)p + 1];
            else return false;      // (cannot happen for a level structure)
        }
    }
    for (int p = 0; p < n; ++p) { lo_ptr[p + 1] += lo_ptr[p]; up_ptr[p + 1] += up_ptr[p]; dg_ptr[p + 1] += dg_ptr[p]; }
    hvec<int> lo_col((size_t)lo_ptr[n]), up_col((size_t)up_ptr[n]), dg_col((size_t)dg_ptr[n]);
    hvec<double> lo_val((size_t)lo_ptr[n]), up_val((size_t)up_ptr[n]), dg_val((size_t)dg_ptr[n]);
    for (int p = 0; p < n; ++p) {
        const int v = perm[p], k = blk_of_lev[lev[v]];
        roff_t a = lo_ptr[p], b = up_ptr[p], c = dg_ptr[p];
        for (roff_t e = I[v]; e < I[v + 1]; ++e) {
            const int pc = iperm[J[e]], kc = blk_of_lev[lev[J[e]]];
            if (kc == k) { dg_col[c] = pc - off[k]; dg_val[c++] = V[e]; }
            else if (kc == k - 1) { lo_col[a] = pc; lo_val[a++] = V[e]; }
            else { up_col[b] = pc; up_val[b++] = V[e]; }
        }
    }
    B.n = n;
    B.nblk = nblk;
    B.off = off;
    B.max_block = maxb;
    B.perm.from_host(perm, s);
    B.lo_ptr.from_host(lo_ptr, s); B.lo_col.from_host(lo_col, s); B.lo_val.from_host(lo_val, s);
    B.up_ptr.from_host(up_ptr, s); B.up_col.from_host(up_col, s); B.up_val.from_host(up_val, s);
    DBuf<roff_t> d_dg_ptr; DBuf<int> d_dg_col; DBuf<double> d_dg_val;
    d_dg_ptr.from_host(dg_ptr, s); d_dg_col.from_host(dg_col, s); d_dg_val.from_host(dg_val, s);
    B.soff.assign((size_t)nblk + 1, 0);
    for (int k = 0; k < nblk; ++k) {
        const size_t nk = (size_t)(off[k + 1] - off[k]);
        B.soff[(size_t)k + 1] = B.soff[k] + nk * nk;
    }
    B.Sinv.alloc(B.soff[nblk]);
    for (DBuf<double> *w : {&B.bp, &B.z, &B.t, &B.xp, &B.r, &B.dx}) w->alloc((size_t)n);
    // ---- block elimination (device) ----
    GjWork gw;
    gw.reserve(maxb, s);
    SA_HIP_CHECK(hipFuncSetAttribute((const void *)bt_schur_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8 * BT_MAX_BLOCK)));
    profiler().begin(s);
    double flops = 0.0;
    for (int k = 0; k < nblk; ++k) {
        const int nk = off[k + 1] - off[k];
        double *S = B.Sinv.p + B.soff[k];
        dense_zero(s, (size_t)nk * nk, S);
        dense_scatter(s, nk, d_dg_ptr.p + off[k], d_dg_col.p, d_dg_val.p, S);
        if (k > 0) {
            const int m = off[k] - off[k - 1];
            hipLaunchKernelGGL(bt_schur_kernel, dim3(nk), dim3(256), (size_t)8 * m, s, nk, m, off[k], off[k - 1], B.lo_ptr.p,
                               B.lo_col.p, B.lo_val.p, B.Sinv.p + B.soff[k - 1], S);
        }
        dense_inverse_inplace(s, nk, S, gw);
        flops += 2.0 * (double)nk * nk * nk;
    }
    SA_HIP_CHECK(hipGetLastError());
    const int bad = gw.info.to_host(s)[0];
    profiler().end(s, "coarse_blocktri_factor", 16.0 * (double)B.soff[nblk], flops);
    if (bad) { B = BlockTri(); return false; }
    return true;
}

static void blocktri_sweeps(hipStream_t s, const BlockTri &B, const double *b, double *x, bool add) {
    const int n = B.n, K = B.nblk;
    profiler().begin(s);
    hipLaunchKernelGGL(bt_gather_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, B.perm.p, b, B.bp.p);
    for (int k = 0; k < K; ++k) {
        const int r0 = B.off[k], nk = B.off[k + 1] - r0;
        const double *rhs = B.bp.p;
        if (k > 0) {
            hipLaunchKernelGGL(bt_couple_kernel, dim3(div_up((long)nk * 8, 256)), dim3(256), 0, s, nk, r0, B.lo_ptr.p, B.lo_col.p,
                               B.lo_val.p, B.z.p, B.bp.p, B.t.p);
            rhs = B.t.p;
        }
        hipLaunchKernelGGL(bt_symv_kernel, dim3(div_up(nk, 4)), dim3(256), 0, s, nk, B.Sinv.p + B.soff[k], rhs + r0,
                           (const double *)nullptr, 1.0, B.z.p + r0);
    }
    {
        const int r0 = B.off[K - 1], nk = B.off[K] - r0;
        SA_HIP_CHECK(hipMemcpyAsync(B.xp.p + r0, B.z.p + r0, 8 * (size_t)nk, hipMemcpyDeviceToDevice, s));
    }
    for (int k = K - 2; k >= 0; --k) {
        const int r0 = B.off[k], nk = B.off[k + 1] - r0;
        // t = -A_{k,k+1} x_{k+1};  x_k = z_k + S_k^-1 t
        hipLaunchKernelGGL(bt_couple_kernel, dim3(div_up((long)nk * 8, 256)), dim3(256), 0, s, nk, r0, B.up_ptr.p, B.up_col.p,
                           B.up_val.p, B.xp.p, (const double *)nullptr, B.t.p);
        hipLaunchKernelGGL(bt_symv_kernel, dim3(div_up(nk, 4)), dim3(256), 0, s, nk, B.Sinv.p + B.soff[k], B.t.p + r0,
                           B.z.p + r0, 1.0, B.xp.p + r0);
    }
    hipLaunchKernelGGL(bt_scatter_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, B.perm.p, B.xp.p, x, add ? 1 : 0);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "coarse_solve_blocktri", 16.0 * (double)B.soff[B.nblk], 4.0 * (double)B.soff[B.nblk]);
}

void blocktri_solve(hipStream_t s, const DCsr &A, const BlockTri &B, const double *b, double *x) {
    blocktri_sweeps(s, B, b, x, false);
    spmv_residual(s, A, x, b, B.r.p);
    blocktri_sweeps(s, B, B.r.p, x, true);
}

}  // namespace saamge_amd
