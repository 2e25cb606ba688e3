// Agglomerate matrix assembly on gfx950: one workgroup per AE, the dense n x n matrix is
// written straight into the eigensolver's workspace (column-major).  Each AE-local row is
// owned by one thread and its element contributions are added in ascending element id, so
// the sums are bit-reproducible and match the reference's accumulation order.
#include "assemble.h"

#include <cstdlib>

namespace saamge_amd {

constexpr int ASM_NT = 256;

__global__ __launch_bounds__(ASM_NT) void ae_assemble_kernel(
    int ae0, const int *__restrict__ ns, const int64_t *__restrict__ moff, double *__restrict__ W,
    const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J, const int *__restrict__ d2ae_I,
    const int *__restrict__ d2ae_J, const int *__restrict__ dof_id_inAE,
    const signed char *__restrict__ flags, const int *__restrict__ d2e_I,
    const int *__restrict__ d2e_J, const int *__restrict__ part, const int *__restrict__ e2d_I,
    const int *__restrict__ e2d_J, const int *__restrict__ elem_ldof,
    const int64_t *__restrict__ eloff, const double *__restrict__ elval, int has_A,
    const roff_t *__restrict__ Arow, const int *__restrict__ Acol, const double *__restrict__ Aval,
    const int64_t *__restrict__ voff, const short *__restrict__ perm, int do_zero, const int *__restrict__ ae_ids = nullptr) {
    // (ae_ids: the agglomerate behind matrix b of the batch when the batch is not a contiguous range -- the representatives
    // of the classes of identical agglomerates)
    const int b = blockIdx.x, p = ae_ids ? ae_ids[b] : ae0 + b, n = ns[b];
    double *Wm = W + moff[b];
    const int tid = threadIdx.x;
    const size_t nn = (size_t)n * n;
    if (do_zero) {      // (0: ae_zero_band_kernel has cleared the band the eigensolver reads)
        for (size_t idx = tid; idx < nn; idx += ASM_NT) Wm[idx] = 0.0;
        __syncthreads();
    }
    const int *aedofs = ae2d_J + ae2d_I[p];
    const short *pm = perm ? perm + voff[b] : nullptr;    // position of an agglomerate row in the matrix
    // (gridDim.y > 1 only without the clearing pass above: the rows are independent, large agglomerates are
    // spread over several workgroups)
    for (int lr0 = blockIdx.y * ASM_NT + tid; lr0 < n; lr0 += ASM_NT * gridDim.y) {
        const int g = aedofs[lr0];
        const int lr = pm ? pm[lr0] : lr0;
        const int fg = has_A ? flags[g] : 0;
        if (has_A) {
            // entries copied from the global matrix (aggregates.cpp:930-934)
            for (roff_t k = Arow[g]; k < Arow[g + 1]; ++k) {
                const int c = Acol[k];
                int idx = -1;
                for (int q = d2ae_I[c]; q < d2ae_I[c + 1]; ++q)
                    if (d2ae_J[q] == p) { idx = q; break; }
                if (idx < 0) continue;  // neighbour not in this AE
                const int fc = flags[c];
                const bool assembled = (fg & 1) && (fc & 1) && (!((fg | fc) & 2) || c == g);
                if (!assembled) {
                    const double v = Aval[k];
                    if (v != 0.0) Wm[(size_t)(pm ? pm[dof_id_inAE[idx]] : dof_id_inAE[idx]) * n + lr] = v;
                }
            }
        }
        // locally assembled entries (agg_assemble_value, aggregates.cpp:68-184)
        for (int q = d2e_I[g]; q < d2e_I[g + 1]; ++q) {
            const int e = d2e_J[q];
            if (part[e] != p) continue;
            const int eb = e2d_I[e], nd = e2d_I[e + 1] - eb;
            int kk = 0;
            while (kk < nd && e2d_J[eb + kk] != g) ++kk;
            const double *M = elval + eloff[e] + (size_t)kk * nd;
            for (int jj = 0; jj < nd; ++jj) {
                bool assembled = true;
                if (has_A) {
                    const int c = e2d_J[eb + jj];
                    const int fc = flags[c];
                    assembled = (fg & 1) && (fc & 1) && (!((fg | fc) & 2) || c == g);
                }
                if (assembled) Wm[(size_t)(pm ? pm[elem_ldof[eb + jj]] : elem_ldof[eb + jj]) * n + lr] += M[jj];
            }
        }
    }
}

// Few large agglomerates on their way to the few-eigenpairs path (coarse levels: 256 matrices of 2 600 rows, half
// bandwidth 300 .. 500 in the permuted order): the half bandwidth is known before a single entry is assembled --
// max |pos(i) - pos(j)| over the non-zero entries of the element matrices -- so only the band (+ AE_BAND_PAD, what the
// blocked factorisation and the 64-row blocks of the row sums read beyond it) is cleared, summed and scaled:
// 7.8 instead of 13.9 GB per pass on the second level of the 256^3 problem.  The entries outside are never read.
constexpr int AE_BAND_PAD = 64;
__global__ __launch_bounds__(ASM_NT) void ae_band_topo_kernel(
    int ae0, const int *__restrict__ ns, const int64_t *__restrict__ voff, const short *__restrict__ perm,
    const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J, const int *__restrict__ d2e_I,
    const int *__restrict__ d2e_J, const int *__restrict__ part, const int *__restrict__ e2d_I,
    const int *__restrict__ e2d_J, const int *__restrict__ elem_ldof, const int64_t *__restrict__ eloff,
    const double *__restrict__ elval, int *__restrict__ bws, const int *__restrict__ ae_ids = nullptr) {
    // (exact zeros of the element matrices do not count: dofs on opposite faces of a coarse element share the
    // element but no entry -- the band of the numbers is half the band of the lists)
    __shared__ int wmax[ASM_NT / 64];
    const int b = blockIdx.x, p = ae_ids ? ae_ids[b] : ae0 + b, n = ns[b], tid = threadIdx.x;
    const int *aedofs = ae2d_J + ae2d_I[p];
    const short *pm = perm + voff[b];
    int bw = 0;
    for (int lr0 = blockIdx.y * ASM_NT + tid; lr0 < n; lr0 += ASM_NT * gridDim.y) {
        const int g = aedofs[lr0], me = pm[lr0];
        for (int q = d2e_I[g]; q < d2e_I[g + 1]; ++q) {
            const int e = d2e_J[q];
            if (part[e] != p) continue;
            const int eb = e2d_I[e], nd = e2d_I[e + 1] - eb;
            int kk = 0;
            while (kk < nd && e2d_J[eb + kk] != g) ++kk;
            const double *M = elval + eloff[e] + (size_t)kk * nd;
            for (int jj = 0; jj < nd; ++jj)
                if (M[jj] != 0.0) bw = max(bw, abs((int)pm[elem_ldof[eb + jj]] - me));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bw = max(bw, __shfl_xor(bw, o, 64));
    if ((tid & 63) == 0) wmax[tid >> 6] = bw;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < ASM_NT / 64; ++w) bw = max(bw, wmax[w]);
        atomicMax(bws + b, bw);         // (zeroed by the host; gridDim.y workgroups share an agglomerate)
    }
}
__global__ __launch_bounds__(256) void ae_zero_band_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                           double *__restrict__ W, const int *__restrict__ bws) {
    const int b = blockIdx.x, n = ns[b], zw = bws[b] + AE_BAND_PAD;
    double *A = W + moff[b];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = blockIdx.y * 4 + wv; j < n; j += 4 * gridDim.y) {
        const int lo = max(0, j - zw), hi = min(n - 1, j + zw);
        for (int i = lo + lane; i <= hi; i += 64) A[(size_t)j * n + i] = 0.0;
    }
}

void ae_assemble(hipStream_t s, const DevRelations &rel, const DCsr *A, const DevElmats &el,
                 int ae0, EigBatch &batch, bool banded, const int *ae_ids) {
    if (!batch.count) return;
    double bytes = 0.0;
    for (int n : batch.h_n) bytes += 8.0 * (double)n * n;
    profiler().begin(s);
    if (banded) {
        if (batch.bw.n < (size_t)batch.count) batch.bw.alloc((size_t)batch.count);
        SA_HIP_CHECK(hipMemsetAsync(batch.bw.p, 0, sizeof(int) * (size_t)batch.count, s));
        const int ny_t = std::max(1, std::min(div_up(batch.max_n, ASM_NT), 4096 / std::max(1, batch.count)));
        hipLaunchKernelGGL(ae_band_topo_kernel, dim3(batch.count, ny_t), dim3(ASM_NT), 0, s, ae0, batch.n.p, batch.voff.p,
                           batch.perm.p, rel.ae2d_I.p, rel.ae2d_J.p, rel.d2e_I.p, rel.d2e_J.p, rel.part.p, rel.e2d_I.p,
                           rel.e2d_J.p, rel.elem_ldof.p, el.off.p, el.val.p, batch.bw.p, ae_ids);
        const int ny = std::max(1, std::min(256, 65536 / std::max(1, batch.count)));
        hipLaunchKernelGGL(ae_zero_band_kernel, dim3(batch.count, ny), dim3(256), 0, s, batch.n.p, batch.moff.p, batch.W.p,
                           batch.bw.p);
        batch.has_bw = true;
    }
    // (not banded: the whole images are cleared by a memset -- one workgroup per agglomerate clearing 38 MB of a
    // 2 187-row Q2 elasticity agglomerate with 256 threads was most of this kernel: 1.1 s of config 5's 23 s)
    if (!banded) SA_HIP_CHECK(hipMemsetAsync(batch.W.p, 0, sizeof(double) * (size_t)batch.h_moff[batch.count], s));
    const int ny_rows = std::max(1, std::min(div_up(batch.max_n, ASM_NT), 4096 / std::max(1, batch.count)));
    hipLaunchKernelGGL(ae_assemble_kernel, dim3(batch.count, ny_rows), dim3(ASM_NT), 0, s, ae0, batch.n.p,
                       batch.moff.p, batch.W.p, rel.ae2d_I.p, rel.ae2d_J.p, rel.d2ae_I.p,
                       rel.d2ae_J.p, rel.dof_id_inAE.p, rel.flags.p, rel.d2e_I.p, rel.d2e_J.p,
                       rel.part.p, rel.e2d_I.p, rel.e2d_J.p, rel.elem_ldof.p, el.off.p, el.val.p,
                       A ? 1 : 0, A ? A->rowptr.p : nullptr, A ? A->col.p : nullptr,
                       A ? A->val.p : nullptr, batch.voff.p, batch.has_perm ? batch.perm.p : nullptr, 0, ae_ids);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "ae_assemble", bytes, 0.0);
}

// D and the symmetric scaling.  Row sums are split over 4 column groups per 64-row block
// (coalesced down the columns) so large agglomerates keep all lanes busy.
__global__ __launch_bounds__(ASM_NT) void ae_scale_kernel(const int *__restrict__ ns,
                                                          const int64_t *__restrict__ moff,
                                                          const int64_t *__restrict__ voff,
                                                          double *__restrict__ W,
                                                          double *__restrict__ dis_out,
                                                          double *__restrict__ D_out) {
    extern __shared__ __align__(16) double lds[];
    const int b = blockIdx.x, n = ns[b];
    double *Wm = W + moff[b];
    const int64_t vo = voff[b];
    // sqrt(a_rr / a_jj) = sqrt(a_rr) * (1 / sqrt(a_jj)): the two factors once per row instead of a
    // square root and a division per matrix entry (which made this pass compute-bound)
    double *dg = lds, *dis = lds + n, *part = lds + 2 * n;  // dg[i] = sqrt(a_ii); part[4][64]
    double *idg = part + 256;                                // 1 / sqrt(a_ii)
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += ASM_NT) {
        const double sq = sqrt(Wm[(size_t)i * n + i]);
        dg[i] = sq;
        idg[i] = 1.0 / sq;
    }
    __syncthreads();
    const int rr = tid & 63, g = tid >> 6;
    for (int r0 = 0; r0 < n; r0 += 64) {
        const int r = r0 + rr;
        double sum = 0.0;
        if (r < n) {
            const double dr = dg[r];
            const int cb = (int)(((long)n * g) / 4), ce = (int)(((long)n * (g + 1)) / 4);
            int j = cb;
            for (; j + 8 <= ce; j += 8) {   // 8 independent loads in flight per lane
                double a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = Wm[(size_t)(j + u) * n + r];
#pragma unroll
                for (int u = 0; u < 8; ++u) sum = fma(fabs(a[u]), dr * idg[j + u], sum);
            }
            for (; j < ce; ++j) sum = fma(fabs(Wm[(size_t)j * n + r]), dr * idg[j], sum);
        }
        part[g * 64 + rr] = sum;
        __syncthreads();
        if (g == 0 && r < n) {
            const double s4 = (part[rr] + part[64 + rr]) + (part[128 + rr] + part[192 + rr]);
            const double di = 1.0 / sqrt(s4);
            dis[r] = di;
            dis_out[vo + r] = di;
            if (D_out) D_out[vo + r] = s4;
        }
        __syncthreads();
    }
    const size_t nn = (size_t)n * n;
    for (size_t idx = tid; idx < nn; idx += ASM_NT) {
        const int r = (int)(idx % n), j = (int)(idx / n);
        Wm[idx] = dis[r] * Wm[idx] * dis[j];
    }
}

// The same for FEW LARGE agglomerates (coarse levels: 256 matrices of 2 600 rows): one workgroup
// per matrix cannot keep HBM busy, so the row sums and the scaling are spread over
// (matrix, 64-row block) and (matrix, column) grids.
__global__ __launch_bounds__(256) void ae_rowsum_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                        const int64_t *__restrict__ voff, const double *__restrict__ W,
                                                        double *__restrict__ dis_out, double *__restrict__ D_out,
                                                        const short *__restrict__ iperm,
                                                        const int *__restrict__ bws = nullptr) {
    __shared__ double part[4 * 64];
    const int b = blockIdx.y, n = ns[b];
    const int r0 = blockIdx.x * 64;
    if (r0 >= n) return;
    // (bws: the columns that can hold an entry of these 64 rows; cleared up to AE_BAND_PAD >= 63 beyond each row's band)
    const int c_lo = bws ? max(0, r0 - bws[b]) : 0, c_hi = bws ? min(n, r0 + 64 + bws[b]) : n;
    const double *Wm = W + moff[b];
    const int64_t vo = voff[b];
    const int rr = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int r = r0 + rr;
    double sum = 0.0;
    if (r < n) {
        const double dr = sqrt(Wm[(size_t)r * n + r]);
        const int cb = c_lo + (int)(((long)(c_hi - c_lo) * g) / 4), ce = c_lo + (int)(((long)(c_hi - c_lo) * (g + 1)) / 4);
        int j = cb;
        for (; j + 4 <= ce; j += 4) {
            double a[4], dj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = Wm[(size_t)(j + u) * n + r]; dj[u] = Wm[(size_t)(j + u) * n + (j + u)]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) sum = fma(fabs(a[u]), dr / sqrt(dj[u]), sum);
        }
        for (; j < ce; ++j) sum = fma(fabs(Wm[(size_t)j * n + r]), dr / sqrt(Wm[(size_t)j * n + j]), sum);
    }
    part[g * 64 + rr] = sum;
    __syncthreads();
    if (g == 0 && r < n) {
        const double s4 = (part[rr] + part[64 + rr]) + (part[128 + rr] + part[192 + rr]);
        const int ro = iperm ? iperm[vo + r] : r;       // (the scalings are stored in agglomerate order)
        dis_out[vo + ro] = 1.0 / sqrt(s4);
        if (D_out) D_out[vo + ro] = s4;
    }
}
__global__ __launch_bounds__(256) void ae_apply_scale_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                             const int64_t *__restrict__ voff, double *__restrict__ W,
                                                             const double *__restrict__ dis,
                                                             const short *__restrict__ iperm,
                                                             const int *__restrict__ bws = nullptr) {
    const int b = blockIdx.y, n = ns[b];
    const int j = blockIdx.x;
    if (j >= n) return;
    double *col = W + moff[b] + (size_t)j * n;
    const double *d = dis + voff[b];
    const short *ip = iperm ? iperm + voff[b] : nullptr;
    const double dj = d[ip ? ip[j] : j];
    const int lo = bws ? max(0, j - bws[b]) : 0, hi = bws ? min(n, j + bws[b] + 1) : n;
    for (int r = lo + threadIdx.x; r < hi; r += 256) col[r] = d[ip ? ip[r] : r] * col[r] * dj;
}

void ae_scale(hipStream_t s, EigBatch &batch, double *Dout, int split) {
    if (!batch.count) return;
    double bytes = 0.0;
    for (int n : batch.h_n) bytes += 24.0 * (double)n * n;
    if (split < 0 ? (batch.count <= 2048 && batch.max_n >= 1024) : split != 0) {
        profiler().begin(s);
        const short *ip = batch.has_perm ? batch.iperm.p : nullptr;
        const int *bws = batch.has_bw ? batch.bw.p : nullptr;      // (band-limited assembly: ae_assemble)
        hipLaunchKernelGGL(ae_rowsum_kernel, dim3(div_up(batch.max_n, 64), batch.count), dim3(256), 0, s, batch.n.p,
                           batch.moff.p, batch.voff.p, batch.W.p, batch.dis.p, Dout, ip, bws);
        hipLaunchKernelGGL(ae_apply_scale_kernel, dim3(batch.max_n, batch.count), dim3(256), 0, s, batch.n.p,
                           batch.moff.p, batch.voff.p, batch.W.p, batch.dis.p, ip, bws);
        SA_HIP_CHECK(hipGetLastError());
        profiler().end(s, "ae_scale", bytes, 0.0);
        return;
    }
    const size_t lds = (3 * (size_t)batch.max_n + 256) * sizeof(double);
    SA_REQUIRE(lds <= 160 * 1024, "agglomerate too large for the scaling kernel");
    static bool attr = false;
    if (!attr) {
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)ae_scale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    profiler().begin(s);
    hipLaunchKernelGGL(ae_scale_kernel, dim3(batch.count), dim3(ASM_NT), lds, s, batch.n.p, batch.moff.p,
                       batch.voff.p, batch.W.p, batch.dis.p, Dout);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "ae_scale", bytes, 0.0);
}

// ---------------------------------------------------------------------------------------
// Fused fine-level path: sparse assembly in LDS -> D -> ONE coalesced write of the dense
// (optionally scaled) matrix.  The AE matrix has the sparsity of A's rows (<= RW entries per
// row), so the n x RW value/column tables of one AE fit in LDS; the dense n x n image that the
// eigensolver wants is then written exactly once (8 n^2 bytes, whole cache lines) instead of
// zero-fill + scatter + read-modify-write scaling (32 n^2 bytes and a latency-bound scatter).
// Entry (row lr, slot k) is produced by ONE thread, contributions added in the element order of
// the row's dof: bit-identical to the scatter kernel above and run-to-run deterministic.
// The dense image is written by sparse ROWS into column-major columns, i.e. as the transpose;
// the matrix is symmetric up to the round-off of the element-matrix sums.
// ---------------------------------------------------------------------------------------
constexpr int AB_NT = 512;

__global__ __launch_bounds__(256) void max_row_kernel(int n, const roff_t *__restrict__ rowptr, int *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    int len = (i < n) ? (int)(rowptr[i + 1] - rowptr[i]) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o, 64));
    // (one contended address: only the few wavefronts that raise the maximum touch it)
    if ((threadIdx.x & 63) == 0 && len > *(volatile int *)out) atomicMax(out, len);
}

// Sparse rows of all AE matrices of a chunk, one thread per (AE, local row, slot of A's row):
// full occupancy for the dependent gathers (dof -> AE-local index, flags, element lists) that
// bound this step.  8-dof elements (dense elem_to_dof): the dof's <= 8 elements are examined with
// independent vector loads and added in ascending element id (the reference's order).
// Output: vals / cols at ((voff[b] + lr) * RW + k).
__global__ __launch_bounds__(256) void ae_rows8_kernel(
    int ae0, int RW, const int *__restrict__ ns, const int64_t *__restrict__ voff,
    const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J, const int *__restrict__ d2ae_I,
    const int *__restrict__ d2ae_J, const int *__restrict__ dof_id_inAE,
    const signed char *__restrict__ flags, const int *__restrict__ d2e_I, const int *__restrict__ d2e_J,
    const int *__restrict__ part, const int *__restrict__ e2d_J, const double *__restrict__ elval,
    const roff_t *__restrict__ Arow, const int *__restrict__ Acol, const double *__restrict__ Aval,
    double *__restrict__ rvals, short *__restrict__ rcols) {
    const int b = blockIdx.y, p = ae0 + b, n = ns[b];
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= n * RW) return;
    const int lr = it / RW, k = it - lr * RW;
    const int g = ae2d_J[ae2d_I[p] + lr];
    const roff_t a0 = Arow[g];
    int lc = -1;
    double v = 0.0;
    if (k < Arow[g + 1] - a0) {
        const int c = Acol[a0 + k];
        for (int q = d2ae_I[c]; q < d2ae_I[c + 1]; ++q)
            if (d2ae_J[q] == p) { lc = dof_id_inAE[q]; break; }
        if (lc >= 0) {
            const int fg = flags[g], fc = flags[c];
            const bool assembled = (fg & 1) && (fc & 1) && (!((fg | fc) & 2) || c == g);
            if (!assembled) {
                v = Aval[a0 + k];               // copied from the global matrix (aggregates.cpp:930-934)
            } else {                            // agg_assemble_value, aggregates.cpp:68-184
                const int qb = d2e_I[g], cnt = d2e_I[g + 1] - qb;
                int es[8], kk[8], jj[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = d2e_J[qb + min(q, cnt - 1)];
                    const int4 lo = *(const int4 *)(e2d_J + (size_t)e * 8), hi = *(const int4 *)(e2d_J + (size_t)e * 8 + 4);
                    es[q] = (q < cnt && part[e] == p) ? e : -1;
                    kk[q] = (lo.x == g) ? 0 : (lo.y == g) ? 1 : (lo.z == g) ? 2 : (lo.w == g) ? 3 :
                            (hi.x == g) ? 4 : (hi.y == g) ? 5 : (hi.z == g) ? 6 : 7;
                    jj[q] = (lo.x == c) ? 0 : (lo.y == c) ? 1 : (lo.z == c) ? 2 : (lo.w == c) ? 3 :
                            (hi.x == c) ? 4 : (hi.y == c) ? 5 : (hi.z == c) ? 6 : (hi.w == c) ? 7 : -1;
                }
                double m[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool on = es[q] >= 0 && jj[q] >= 0;
                    m[q] = on ? elval[((size_t)max(es[q], 0) * 8 + kk[q]) * 8 + max(jj[q], 0)] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (es[q] >= 0 && jj[q] >= 0) v += m[q];
                for (int q = 8; q < cnt; ++q) {   // (more than 8 elements at a dof: plain loop)
                    const int e = d2e_J[qb + q];
                    if (part[e] != p) continue;
                    int k2 = -1, j2 = -1;
                    for (int t = 0; t < 8; ++t) {
                        const int dd = e2d_J[(size_t)e * 8 + t];
                        if (dd == g && k2 < 0) k2 = t;
                        if (dd == c && j2 < 0) j2 = t;
                    }
                    if (j2 >= 0) v += elval[((size_t)e * 8 + k2) * 8 + j2];
                }
            }
        }
    }
    const size_t o = ((size_t)voff[b] + lr) * RW + k;
    rvals[o] = v;
    rcols[o] = (short)lc;
}

// The same with one workgroup per agglomerate and the agglomerate's own tables in LDS (round 3).  The kernel above
// starts every (row, entry) thread from scratch: the row's dof and row offsets, then, for the entry's column, the
// dof -> AE list scanned for this agglomerate, the local index, two flags -- a dozen loads per thread, six of them
// scattered (one cache line per lane) and dependent.  Here the agglomerate's dof list goes into an LDS hash table
// (dof -> local index) together with the flags and row offsets of its dofs; an entry then costs its column and value
// (coalesced: the entries of a row are contiguous) and LDS look-ups.  Interface entries (both dofs between
// agglomerates) still gather their element matrices as above.  hsize: a power of two >= 2 n.
__global__ __launch_bounds__(256) void ae_rows8_lds_kernel(
    int ae0, int RW, int hsize, const int *__restrict__ ns, const int64_t *__restrict__ voff,
    const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J, const signed char *__restrict__ flags,
    const int *__restrict__ d2e_I, const int *__restrict__ d2e_J, const int *__restrict__ part,
    const int *__restrict__ e2d_J, const double *__restrict__ elval, const roff_t *__restrict__ Arow,
    const int *__restrict__ Acol, const double *__restrict__ Aval, double *__restrict__ rvals,
    short *__restrict__ rcols) {
    extern __shared__ __align__(16) unsigned char ar_lds[];
    const int b = blockIdx.x, p = ae0 + b, n = ns[b];
    roff_t *la0 = (roff_t *)ar_lds;                 // [n] first entry of the row in A
    int *lg = (int *)(la0 + n);                     // [n] global dof
    int *llen = lg + n;                             // [n] entries of the row
    int *hkey = llen + n;                           // [hsize] dof or -1
    short *hval = (short *)(hkey + hsize);          // [hsize] local index
    signed char *lflag = (signed char *)(hval + hsize);   // [n]
    // per dof of the agglomerate: its elements INSIDE the agglomerate, in the order of the dof -> element list, and the dof's
    // slot in each (round 4).  An assembled entry (row g, column c) used to walk g's element list in global memory -- per
    // element its partition and two 16-byte loads of its dofs -- for every one of the row's ~27 entries; now the lists are
    // made once per dof, and an entry matches the row's list against the column's in LDS: one global load (the element-matrix
    // entry) per common element, the same products in the same order.
    int *del = (int *)(((uintptr_t)(lflag + n) + 3) & ~(uintptr_t)3);      // [n][8] element or -1
    unsigned char *dkk = (unsigned char *)(del + 8 * (size_t)n);          // [n][8] slot of the dof in it
    const int tid = threadIdx.x;
    for (int i = tid; i < hsize; i += 256) hkey[i] = -1;
    __syncthreads();
    const int *dofs = ae2d_J + ae2d_I[p];
    for (int i = tid; i < n; i += 256) {
        const int g = dofs[i];
        const roff_t a0 = Arow[g];
        lg[i] = g;
        la0[i] = a0;
        llen[i] = (int)(Arow[g + 1] - a0);
        lflag[i] = flags[g];
        unsigned hpos = hash_home((unsigned)g, (unsigned)hsize);
        while (atomicCAS(&hkey[hpos], -1, g) != -1) hpos = (hpos + 1) & (unsigned)(hsize - 1);
        hval[hpos] = (short)i;
    }
    bool many = false;      // (a dof with more than 8 elements: the agglomerate takes the walk through global memory)
    for (int it = tid; it < n * 8; it += 256) {
        const int i = it >> 3, q = it & 7;
        const int g = dofs[i];
        const int qb = d2e_I[g], cnt = d2e_I[g + 1] - qb;
        many = many || cnt > 8;
        int e = -1, kk = 0;
        if (q < cnt) {
            e = d2e_J[qb + q];
            if (part[e] != p) e = -1;
            else {
                const int4 lo = *(const int4 *)(e2d_J + (size_t)e * 8), hi = *(const int4 *)(e2d_J + (size_t)e * 8 + 4);
                kk = (lo.x == g) ? 0 : (lo.y == g) ? 1 : (lo.z == g) ? 2 : (lo.w == g) ? 3 :
                     (hi.x == g) ? 4 : (hi.y == g) ? 5 : (hi.z == g) ? 6 : 7;
            }
        }
        del[it] = e;
        dkk[it] = (unsigned char)kk;
    }
    const bool walk = __syncthreads_or(many ? 1 : 0) != 0;
    const size_t obase = (size_t)voff[b] * RW;
    // four entries per thread and trip: their columns and values are requested together (one entry at a time, a thread
    // walked its ~43 entries through 43 x two dependent global latencies)
    constexpr int UN = 4;
    for (int it0 = tid; it0 < n * RW; it0 += 256 * UN) {
        int lrs[UN], cs[UN], lcs[UN];
        double vs[UN];
        roff_t as[UN];
        bool on[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int it = it0 + 256 * u;
            const int lr = it < n * RW ? it / RW : 0, k = it - lr * RW;
            lrs[u] = lr;
            on[u] = it < n * RW && k < llen[lr];
            as[u] = la0[lr] + k;
            cs[u] = on[u] ? Acol[as[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) vs[u] = on[u] ? Aval[as[u]] : 0.0;      // (used unless the entry is assembled from elements)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            lcs[u] = -1;
            if (on[u]) {
                const int c = cs[u];
                unsigned hpos = hash_home((unsigned)c, (unsigned)hsize);
                for (;;) {
                    const int key = hkey[hpos];
                    if (key == c) { lcs[u] = hval[hpos]; break; }
                    if (key == -1) break;
                    hpos = (hpos + 1) & (unsigned)(hsize - 1);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int it = it0 + 256 * u;
            if (it >= n * RW) continue;
            const int lr = lrs[u], lc = lcs[u], c = cs[u];
            double v = 0.0;
            if (lc >= 0) {
                const int g = lg[lr];
                const int fg = lflag[lr], fc = lflag[lc];
                const bool assembled = (fg & 1) && (fc & 1) && (!((fg | fc) & 2) || c == g);
                if (!assembled) {
                    v = vs[u];                      // copied from the global matrix (aggregates.cpp:930-934)
                } else if (!walk) {                 // agg_assemble_value, aggregates.cpp:68-184: the elements of g inside the agglomerate
                    // that also hold c, ascending in g's list (the order of the walk below)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = del[lr * 8 + q];
                        if (e < 0) continue;
                        int jj = -1;
#pragma unroll
                        for (int q2 = 0; q2 < 8; ++q2)
                            if (del[lc * 8 + q2] == e) jj = dkk[lc * 8 + q2];
                        if (jj >= 0) v += elval[((size_t)e * 8 + dkk[lr * 8 + q]) * 8 + jj];
                    }
                } else {
                    const int qb = d2e_I[g], cnt = d2e_I[g + 1] - qb;
                    for (int q = 0; q < cnt; ++q) {      // ascending element id: the order of the kernel above
                        const int e = d2e_J[qb + q];
                        if (part[e] != p) continue;
                        const int4 lo = *(const int4 *)(e2d_J + (size_t)e * 8), hi = *(const int4 *)(e2d_J + (size_t)e * 8 + 4);
                        const int kk = (lo.x == g) ? 0 : (lo.y == g) ? 1 : (lo.z == g) ? 2 : (lo.w == g) ? 3 :
                                       (hi.x == g) ? 4 : (hi.y == g) ? 5 : (hi.z == g) ? 6 : 7;
                        const int jj = (lo.x == c) ? 0 : (lo.y == c) ? 1 : (lo.z == c) ? 2 : (lo.w == c) ? 3 :
                                       (hi.x == c) ? 4 : (hi.y == c) ? 5 : (hi.z == c) ? 6 : (hi.w == c) ? 7 : -1;
                        if (jj >= 0) v += elval[((size_t)e * 8 + kk) * 8 + jj];
                    }
                }
            }
            rvals[obase + it] = v;
            rcols[obase + it] = (short)lc;
        }
    }
}

constexpr int AB_MAXE = 8;   // elements per dof kept in the LDS row tables (hexes: <= 8)

// NDE > 0: every element has exactly NDE dofs (level 0: elem_to_dof is a dense NE x NDE array),
// which turns the per-element searches into a few independent vector loads.
template <bool SCALE, int NDE, bool PRE>
__global__ __launch_bounds__(AB_NT) void ae_build_kernel(
    int ae0, int RW, const int *__restrict__ ns, const int64_t *__restrict__ moff,
    const int64_t *__restrict__ voff, double *__restrict__ W, double *__restrict__ dis_out,
    double *__restrict__ D_out, const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J,
    const int *__restrict__ d2ae_I, const int *__restrict__ d2ae_J,
    const int *__restrict__ dof_id_inAE, const signed char *__restrict__ flags,
    const int *__restrict__ d2e_I, const int *__restrict__ d2e_J, const int *__restrict__ part,
    const int *__restrict__ e2d_I, const int *__restrict__ e2d_J, const int64_t *__restrict__ eloff,
    const double *__restrict__ elval, const roff_t *__restrict__ Arow, const int *__restrict__ Acol,
    const double *__restrict__ Aval, const double *__restrict__ rvals, const short *__restrict__ rcols,
    const short *__restrict__ perm, int *__restrict__ bw_out, int band_only, const int *__restrict__ only = nullptr) {
    extern __shared__ __align__(16) double lds[];
    __shared__ int anybig, sbw;
    // (only: the matrices of the batch to build -- the representatives of its classes of identical agglomerates)
    const int b = only ? only[blockIdx.x] : (int)blockIdx.x, p = ae0 + b, n = ns[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = AB_NT / 64;
    double *vals = lds;                                   // [n * RW]
    double *dg = vals + (size_t)n * RW;                   // [n]
    double *dis = dg + n;                                 // [n]
    double *colbuf = dis + n;                             // [NW][n]; phases 0-1: the row tables
    int *rowel = (int *)colbuf;                           // [n * AB_MAXE] elements of the row's dof in this AE
    short *rowkk = (short *)(rowel + (size_t)n * AB_MAXE);  // [n * AB_MAXE] position of the dof in them
    short *cols = (short *)(colbuf + (size_t)NW * n);     // [n * RW] AE-local column or -1
    int *gdof = (int *)(cols + (((size_t)n * RW + 3) & ~(size_t)3));  // [n] global dof of each row
    const int *aedofs = ae2d_J + ae2d_I[p];
    if (PRE) {   // sparse rows precomputed by ae_rows8_kernel: coalesced copy into LDS
        const size_t base = (size_t)voff[b] * RW;
        for (int it = tid; it < n * RW; it += AB_NT) {
            vals[it] = rvals[base + it];
            cols[it] = rcols[base + it];
        }
        __syncthreads();
    } else {
    if (tid == 0) anybig = 0;
    for (int lr = tid; lr < n; lr += AB_NT) gdof[lr] = aedofs[lr];
    __syncthreads();
    // ---- 0. per row: the elements of its dof that lie in this AE, and the dof's slot in them ----
    for (int it = tid; it < n * AB_MAXE; it += AB_NT) {
        const int lr = it / AB_MAXE, q = it - lr * AB_MAXE;
        const int g = gdof[lr];
        const int qb = d2e_I[g], cnt = d2e_I[g + 1] - qb;
        int e = -1, kk = 0;
        if (q == 0 && cnt > AB_MAXE) anybig = 1;
        if (q < cnt) {
            e = d2e_J[qb + q];
            if (part[e] != p) {
                e = -1;
            } else if (NDE == 8) {
                const int4 lo = *(const int4 *)(e2d_J + (size_t)e * 8), hi = *(const int4 *)(e2d_J + (size_t)e * 8 + 4);
                kk = (lo.x == g) ? 0 : (lo.y == g) ? 1 : (lo.z == g) ? 2 : (lo.w == g) ? 3 :
                     (hi.x == g) ? 4 : (hi.y == g) ? 5 : (hi.z == g) ? 6 : 7;
            } else {
                const int eb = e2d_I[e], nd = e2d_I[e + 1] - eb;
                while (kk < nd && e2d_J[eb + kk] != g) ++kk;
            }
        }
        rowel[it] = e;
        rowkk[it] = (short)kk;
    }
    __syncthreads();
    const bool big = anybig != 0;
    // ---- 1. sparse rows ----
    for (int it = tid; it < n * RW; it += AB_NT) {
        const int lr = it / RW, k = it - lr * RW;
        const int g = gdof[lr];
        const roff_t a0 = Arow[g];
        int lc = -1;
        double v = 0.0;
        if (k < Arow[g + 1] - a0) {
            const int c = Acol[a0 + k];
            for (int q = d2ae_I[c]; q < d2ae_I[c + 1]; ++q)
                if (d2ae_J[q] == p) { lc = dof_id_inAE[q]; break; }
            if (lc >= 0) {
                const int fg = flags[g], fc = flags[c];
                const bool assembled = (fg & 1) && (fc & 1) && (!((fg | fc) & 2) || c == g);
                if (!assembled) {
                    v = Aval[a0 + k];           // copied from the global matrix (aggregates.cpp:930-934)
                } else if (NDE == 8 && !big) {  // agg_assemble_value, aggregates.cpp:68-184
                    int es[AB_MAXE], jj[AB_MAXE];
#pragma unroll
                    for (int q = 0; q < AB_MAXE; ++q) {
                        es[q] = rowel[lr * AB_MAXE + q];
                        const int ec = max(es[q], 0);
                        const int4 lo = *(const int4 *)(e2d_J + (size_t)ec * 8), hi = *(const int4 *)(e2d_J + (size_t)ec * 8 + 4);
                        jj[q] = (lo.x == c) ? 0 : (lo.y == c) ? 1 : (lo.z == c) ? 2 : (lo.w == c) ? 3 :
                                (hi.x == c) ? 4 : (hi.y == c) ? 5 : (hi.z == c) ? 6 : (hi.w == c) ? 7 : -1;
                    }
                    double m[AB_MAXE];
#pragma unroll
                    for (int q = 0; q < AB_MAXE; ++q) {
                        const bool on = es[q] >= 0 && jj[q] >= 0;
                        const int kk = rowkk[lr * AB_MAXE + q];
                        m[q] = on ? elval[((size_t)es[q] * 8 + kk) * 8 + jj[q]] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < AB_MAXE; ++q)
                        if (es[q] >= 0 && jj[q] >= 0) v += m[q];
                } else {
                    for (int q = d2e_I[g]; q < d2e_I[g + 1]; ++q) {
                        const int e = d2e_J[q];
                        if (part[e] != p) continue;
                        const int eb = e2d_I[e], nd = e2d_I[e + 1] - eb;
                        int kk = -1, j2 = -1;
                        for (int t = 0; t < nd; ++t) {
                            const int dd = e2d_J[eb + t];
                            if (dd == g && kk < 0) kk = t;
                            if (dd == c && j2 < 0) j2 = t;
                        }
                        if (j2 >= 0) v += elval[eloff[e] + (size_t)kk * nd + j2];
                    }
                }
            }
        }
        vals[it] = v;
        cols[it] = (short)lc;
    }
    __syncthreads();
    }
    // ---- 2. diagonal, 3. D and D^-1/2 ----
    for (int lr = tid; lr < n; lr += AB_NT) {
        double d = 0.0;
        for (int k = 0; k < RW; ++k)
            if (cols[lr * RW + k] == lr) d = vals[lr * RW + k];
        dg[lr] = d;
    }
    __syncthreads();
    if (SCALE) {
        for (int lr = tid; lr < n; lr += AB_NT) {
            const double dr = dg[lr];
            double sum = 0.0;
            for (int k = 0; k < RW; ++k) {
                const int lc = cols[lr * RW + k];
                const double a = vals[lr * RW + k];
                if (lc >= 0 && a != 0.0) sum += fabs(a) * sqrt(dr / dg[lc]);
            }
            const double di = 1.0 / sqrt(sum);
            dis[lr] = di;
            dis_out[voff[b] + lr] = di;
            if (D_out) D_out[voff[b] + lr] = sum;
        }
        __syncthreads();
    }
    // ---- 4. dense image: wavefront w writes columns w, w + NW, ... ----
    // (perm: row / column r of the agglomerate goes to position perm[r] of the dense matrix)
    double *Wm = W + moff[b];
    double *cb = colbuf + (size_t)wave * n;
    const short *pm = perm ? perm + voff[b] : nullptr;
    for (int r = lane; r < n; r += 64) cb[r] = 0.0;
    // band_only: the consumer is the banded factorisation, which never reads further than bw + 2 SB
    // from the diagonal (bw = half bandwidth of the stored entries) -- only that part of every column is
    // written (a fifth of the 87 GB of dense images on the headline problem's fine level)
    int wb = n;
    if (band_only) {
        if (tid == 0) sbw = 0;
        __syncthreads();
        int m = 0;
        for (int it = tid; it < n * RW; it += AB_NT) {
            const int lc = cols[it];
            if (lc >= 0 && vals[it] != 0.0) m = max(m, abs((pm ? pm[lc] : lc) - (pm ? pm[it / RW] : it / RW)));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if (lane == 0) atomicMax(&sbw, m);
        __syncthreads();
        wb = sbw + 32;
    }
    int mybw = 0;          // half bandwidth of the matrix as written (for the banded factorisation)
    for (int j = wave; j < n; j += NW) {
        const double dj = SCALE ? dis[j] : 1.0;
        const int pj = pm ? pm[j] : j;
        for (int k = lane; k < RW; k += 64) {
            const int lc = cols[j * RW + k];
            if (lc >= 0) {
                const int pc = pm ? pm[lc] : lc;
                const double v = SCALE ? dj * vals[j * RW + k] * dis[lc] : vals[j * RW + k];
                cb[pc] = v;
                if (v != 0.0) mybw = max(mybw, abs(pc - pj));
            }
        }
        // (wave-private LDS: program order within the wavefront is enough)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        double *col = Wm + (size_t)pj * n;
        for (int r = max(0, pj - wb) + lane; r < min(n, pj + wb + 1); r += 64) col[r] = cb[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int k = lane; k < RW; k += 64) {
            const int lc = cols[j * RW + k];
            if (lc >= 0) cb[pm ? pm[lc] : lc] = 0.0;
        }
    }
    if (bw_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mybw = max(mybw, __shfl_xor(mybw, o, 64));
        if (lane == 0 && mybw > 0) atomicMax(bw_out + b, mybw);
    }
}

// Element-free mode, ExtractSubMatrices (amg/src/tg.cpp:579-672): the principal submatrix of A
// on the AE's dofs (non-zero entries only); rows with more than one stored entry get their row
// sum subtracted from the diagonal, a non-positive diagonal becomes 1, a single-dof AE is [1].
// One thread per local row (the AEs do not overlap in this mode: every dof is in exactly one AE).
__global__ __launch_bounds__(ASM_NT) void ae_extract_kernel(
    int ae0, const int *__restrict__ ns, const int64_t *__restrict__ moff, double *__restrict__ W,
    const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J, const int *__restrict__ d2ae_I,
    const int *__restrict__ d2ae_J, const int *__restrict__ dof_id_inAE, const roff_t *__restrict__ Arow,
    const int *__restrict__ Acol, const double *__restrict__ Aval) {
    const int b = blockIdx.x, p = ae0 + b, n = ns[b];
    double *Wm = W + moff[b];
    const int tid = threadIdx.x;
    const size_t nn = (size_t)n * n;
    for (size_t idx = tid; idx < nn; idx += ASM_NT) Wm[idx] = 0.0;
    __syncthreads();
    if (n == 1) {
        if (tid == 0) Wm[0] = 1.0;
        return;
    }
    const int *aedofs = ae2d_J + ae2d_I[p];
    for (int lr = tid; lr < n; lr += ASM_NT) {
        const int g = aedofs[lr];
        double rowsum = 0.0, diag = 0.0;
        int stored = 0;
        for (roff_t k = Arow[g]; k < Arow[g + 1]; ++k) {
            const int c = Acol[k];
            const double v = Aval[k];
            if (v == 0.0) continue;
            int lc = -1;
            for (int q = d2ae_I[c]; q < d2ae_I[c + 1]; ++q)
                if (d2ae_J[q] == p) { lc = dof_id_inAE[q]; break; }
            if (lc < 0) continue;
            ++stored;
            rowsum += v;
            if (lc == lr) diag = v;
            else Wm[(size_t)lc * n + lr] = v;     // entry (row lr, column lc), column-major
        }
        if (stored > 1) diag -= rowsum;
        if (!(diag > 0.0)) diag = 1.0;
        Wm[(size_t)lr * n + lr] = diag;
    }
}

// Element-free mode, window variant (WindowSubMatrices, amg/src/tg.cpp:741-858): W = A_TT + A_TX E
// with E[x, l] = a_lx / sum_{k in T} a_xk for the outside neighbours x of the AE's dof set T, i.e.
// W[i, l] = a_il + sum_x a_ix a_xl / denom_x (A symmetric).  One thread per local row i: it alone
// writes row i of the dense image, walking the rows of its outside neighbours twice (denominator,
// then update) -- no atomics, fixed summation order.
__global__ __launch_bounds__(ASM_NT) void ae_window_kernel(
    int ae0, const int *__restrict__ ns, const int64_t *__restrict__ moff, double *__restrict__ W,
    const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J, const int *__restrict__ d2ae_I,
    const int *__restrict__ d2ae_J, const int *__restrict__ dof_id_inAE, const roff_t *__restrict__ Arow,
    const int *__restrict__ Acol, const double *__restrict__ Aval) {
    const int b = blockIdx.x, p = ae0 + b, n = ns[b];
    double *Wm = W + moff[b];
    const int tid = threadIdx.x;
    const size_t nn = (size_t)n * n;
    for (size_t idx = tid; idx < nn; idx += ASM_NT) Wm[idx] = 0.0;
    __syncthreads();
    if (n == 1) {
        if (tid == 0) Wm[0] = 1.0;
        return;
    }
    auto local_id = [&](int c) {
        for (int q = d2ae_I[c]; q < d2ae_I[c + 1]; ++q)
            if (d2ae_J[q] == p) return dof_id_inAE[q];
        return -1;
    };
    const int *aedofs = ae2d_J + ae2d_I[p];
    for (int lr = tid; lr < n; lr += ASM_NT) {
        const int g = aedofs[lr];
        for (roff_t k = Arow[g]; k < Arow[g + 1]; ++k) {
            const int c = Acol[k];
            const double v = Aval[k];
            const int lc = local_id(c);
            if (lc >= 0) {
                Wm[(size_t)lc * n + lr] += v;
                continue;
            }
            double denom = 0.0;
            for (roff_t kk = Arow[c]; kk < Arow[c + 1]; ++kk)
                if (local_id(Acol[kk]) >= 0) denom += Aval[kk];
            for (roff_t kk = Arow[c]; kk < Arow[c + 1]; ++kk) {
                const int ll = local_id(Acol[kk]);
                if (ll >= 0) Wm[(size_t)ll * n + lr] += v * (Aval[kk] / denom);
            }
        }
    }
}

int csr_max_row(hipStream_t s, const DCsr &A) {
    DBuf<int> m(1);
    m.zero(s);
    if (A.nrows)
        hipLaunchKernelGGL(max_row_kernel, dim3(div_up(A.nrows, 256)), dim3(256), 0, s, A.nrows, A.rowptr.p, m.p);
    SA_HIP_CHECK(hipGetLastError());
    return m.to_host(s)[0];
}

// Sparse rows of the AE matrices of one chunk (RW slots per row).  With `rows` (position of the
// chunk's first row among the rows of ALL agglomerates of the level, and their total) the rows
// are kept for the whole hierarchy build in a grow-only buffer: the eigenproblem pass computes
// them once and the pass that builds the coarse element matrices reads them again.
struct RowsCache {
    int gen = -1;                 // hierarchy build the contents belong to
    const void *key = nullptr;    // the level matrix
    int RW = 0;
    int lo = 0, hi = 0;           // AEs [lo, hi) are present
    DBuf<double> vals;
    DBuf<short> cols;
};
static RowsCache &g_rows = *new RowsCache;       // never destroyed: no HIP calls from static destructors (see eig.hip arena())
static int g_rows_gen = 0;
void ae_rows_new_build() { ++g_rows_gen; }

static void launch_rows8(hipStream_t s, const DevRelations &rel, const DCsr &A, const DevElmats &el, int ae0,
                         const EigBatch &batch, int RW, const double *&rv, const short *&rc,
                         const RowsSpan *rows = nullptr) {
    static DBuf<double> &g_rvals = *new DBuf<double>;
    static DBuf<short> &g_rcols = *new DBuf<short>;
    double *dv;
    short *dc;
    if (rows) {
        RowsCache &c = g_rows;
        const size_t need = (size_t)rows->total * RW + 64;
        if (c.gen != g_rows_gen || c.key != (const void *)A.val.p || c.RW != RW || c.vals.n < need) {
            if (c.vals.n < need) { c.vals.alloc(need); c.cols.alloc(need); }
            c.gen = g_rows_gen; c.key = A.val.p; c.RW = RW; c.lo = c.hi = 0;
        }
        dv = c.vals.p + (size_t)rows->first * RW;
        dc = c.cols.p + (size_t)rows->first * RW;
        rv = dv;
        rc = dc;
        if (ae0 >= c.lo && ae0 + batch.count <= c.hi) return;          // computed earlier in this build
        if (c.lo == c.hi) { c.lo = ae0; c.hi = ae0 + batch.count; }
        else if (ae0 == c.hi) c.hi += batch.count;                      // the passes walk the AEs in order
    } else {
        const size_t need = (size_t)batch.h_voff[batch.count] * RW + 64;
        if (g_rvals.n < need) { g_rvals.alloc(need + need / 8); g_rcols.alloc(need + need / 8); }
        dv = g_rvals.p;
        dc = g_rcols.p;
    }
    profiler().begin(s);
    constexpr bool old_rows = false;      // (the kernel without the LDS hash of the agglomerate's dofs: larger agglomerates only)
    int hsize = 64;
    while (hsize < 2 * batch.max_n) hsize <<= 1;
    const size_t lds = (size_t)batch.max_n * (8 + 4 + 4 + 1) + (size_t)hsize * 6 + 16 + (size_t)batch.max_n * 40 + 8;      // (+ the per-dof element lists)
    if (!old_rows && lds <= 64 * 1024)
        hipLaunchKernelGGL(ae_rows8_lds_kernel, dim3(batch.count), dim3(256), lds, s, ae0, RW, hsize, batch.n.p, batch.voff.p,
                           rel.ae2d_I.p, rel.ae2d_J.p, rel.flags.p, rel.d2e_I.p, rel.d2e_J.p, rel.part.p, rel.e2d_J.p,
                           el.dense(), A.rowptr.p, A.col.p, A.val.p, dv, dc);
    else
    hipLaunchKernelGGL(ae_rows8_kernel, dim3(div_up((long)batch.max_n * RW, 256), batch.count), dim3(256), 0, s,
                       ae0, RW, batch.n.p, batch.voff.p, rel.ae2d_I.p, rel.ae2d_J.p, rel.d2ae_I.p,
                       rel.d2ae_J.p, rel.dof_id_inAE.p, rel.flags.p, rel.d2e_I.p, rel.d2e_J.p, rel.part.p,
                       rel.e2d_J.p, el.dense(), A.rowptr.p, A.col.p, A.val.p, dv, dc);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "ae_rows", 0.0, 0.0);
    rv = dv;
    rc = dc;
}

bool ae_sparse_rows(hipStream_t s, const DevRelations &rel, const DCsr &A, const DevElmats &el, int ae0,
                    const EigBatch &batch, int &RW, const double *&rv, const short *&rc, const RowsSpan *rows) {
    if (!batch.count || el.algebraic || el.nde != 8 || batch.count > 65535 || batch.max_n > 32767) return false;
    if (A.max_row < 0) A.max_row = csr_max_row(s, A);
    RW = A.max_row;
    launch_rows8(s, rel, A, el, ae0, batch, RW, rv, rc, rows);
    return true;
}

// perm[r] = position of local row r in the dense matrix: the rank of its global dof among the
// agglomerate's dofs -- unless the sorted dofs form a lexicographic box
//   g = g0 + i + s2 j + s3 k,  0 <= i < a, 0 <= j < b, 0 <= k < c   (a structured-mesh agglomerate),
// in which case the box is renumbered with its SHORTEST extent running fastest: half bandwidth
// e1 e2 + e1 + 1 with e1 <= e2 the two smaller extents instead of a b + a + 1 (the 9 x 9 x 5 boxes
// of the headline problem: 51 instead of 91).  Everything is verified entry by entry; any other
// agglomerate keeps the rank order.  (box_order = 0: rank order only.)
__global__ __launch_bounds__(256) void ae_perm_kernel(int ae0, const int *__restrict__ ns, const int64_t *__restrict__ voff,
                                                      const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J,
                                                      short *__restrict__ perm, short *__restrict__ iperm, int box_order) {
    extern __shared__ int gd[];          // [n] dofs in table order, [n] sorted
    __shared__ int box[6];               // a, b, c, s2, s3, ok
    const int b = blockIdx.x, n = ns[b];
    int *sid = gd + n;
    const int *aedofs = ae2d_J + ae2d_I[ae0 + b];
    for (int r = threadIdx.x; r < n; r += 256) gd[r] = aedofs[r];
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256) {
        const int g = gd[r];
        int rank = 0;
        for (int k = 0; k < n; ++k) rank += gd[k] < g;
        sid[rank] = g;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 1, bb = 1, c = 1, s2 = 0, s3 = 0, ok = 0;
        while (a < n && sid[a] == sid[0] + a) ++a;
        if (box_order && a < n && n % a == 0) {
            s2 = sid[a] - sid[0];
            while (bb * a < n && sid[bb * a] == sid[0] + bb * s2) ++bb;
            if (n % (a * bb) == 0) {
                c = n / (a * bb);
                s3 = (c > 1) ? sid[a * bb] - sid[0] : 0;
                ok = s2 >= a && (c == 1 || s3 >= bb * s2);
            }
        }
        box[0] = a; box[1] = bb; box[2] = c; box[3] = s2; box[4] = s3; box[5] = ok;
    }
    __syncthreads();
    const int a = box[0], bb = box[1], c = box[2], s2 = box[3], s3 = box[4];
    if (box[5]) {
        int bad = 0;
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int i = idx % a, j = (idx / a) % bb, k = idx / (a * bb);
            bad |= sid[idx] != sid[0] + i + s2 * j + s3 * k;
        }
        if (__syncthreads_or(bad) && threadIdx.x == 0) box[5] = 0;
        __syncthreads();
    }
    // extents sorted ascending (e[0] fastest); the rank order is (a, b, c) = (fastest .. slowest)
    int ext[3] = {a, bb, c}, dim[3] = {0, 1, 2};
    if (box[5]) {
        for (int u = 0; u < 2; ++u)
            for (int v = 0; v < 2 - u; ++v)
                if (ext[v] > ext[v + 1]) {
                    const int t = ext[v]; ext[v] = ext[v + 1]; ext[v + 1] = t;
                    const int d = dim[v]; dim[v] = dim[v + 1]; dim[v + 1] = d;
                }
    }
    for (int r = threadIdx.x; r < n; r += 256) {
        const int g = gd[r];
        int lo = 0, hi = n - 1;                      // rank = position in the sorted list
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sid[mid] < g) lo = mid + 1; else hi = mid; }
        int pos = lo;
        if (box[5]) {
            const int crd[3] = {lo % a, (lo / a) % bb, lo / (a * bb)};
            const int c0 = dim[0] == 0 ? crd[0] : dim[0] == 1 ? crd[1] : crd[2];
            const int c1 = dim[1] == 0 ? crd[0] : dim[1] == 1 ? crd[1] : crd[2];
            const int c2 = dim[2] == 0 ? crd[0] : dim[2] == 1 ? crd[1] : crd[2];
            pos = c0 + ext[0] * (c1 + ext[1] * c2);
        }
        perm[voff[b] + r] = (short)pos;
        iperm[voff[b] + pos] = (short)r;
    }
}

// The same for the GENERIC assembly (ae_assemble_kernel + the scaling: coarse levels, elements that are not 8-dof hexes): the
// matrix, its scaling and its band are functions of what that kernel reads -- per row of the agglomerate (in agglomerate
// order): its position in the matrix, its flag, the coarse start vector; per entry of its row of the global matrix whose column
// lies in the agglomerate: the column's local number, its flag, the value; per element of the row's dof inside the agglomerate:
// the dof's slot in the element, and per dof of the element its local number, its flag, the entry of the element matrix.
// ai_row_walk visits exactly those words for one row, a wavefront per row; asm_hash_kernel sums mixed (word, position, row) triples
// into a 128-bit hash per agglomerate, asm_verify_kernel walks an agglomerate and the first member of its class in lockstep and
// compares them position by position.
struct AeInputs {
    const int *ns;
    const int64_t *voff;
    const short *perm;          // or null
    const double *x0c;          // or null
    const int *ae2d_I, *ae2d_J, *d2ae_I, *d2ae_J, *dof_id_inAE;
    const signed char *flags;
    const int *d2e_I, *d2e_J, *part, *e2d_I, *e2d_J, *elem_ldof;
    const int64_t *eloff;
    const double *elval;
    int has_A;
    const roff_t *Arow;
    const int *Acol;
    const double *Aval;
    int ae0;
};
// local number of a global dof in an agglomerate, or -1: an LDS hash table of the agglomerate's dofs (open addressing, hsize a
// power of two >= 2 n; the list search through dof -> AE of the assembly kernel costs six dependent global loads per entry of a
// 375-entry row)
struct AiTable {
    int *key;
    short *val;
    unsigned mask;
    __device__ inline void build(const AeInputs &v, int p, int n, int tid, int nt) {
        for (int i = tid; i <= (int)mask; i += nt) key[i] = -1;
        __syncthreads();
        const int *dofs = v.ae2d_J + v.ae2d_I[p];
        for (int i = tid; i < n; i += nt) {
            const int g = dofs[i];
            unsigned h = hash_home((unsigned)g, mask + 1u);
            while (atomicCAS(&key[h], -1, g) != -1) h = (h + 1) & mask;
            val[h] = (short)i;
        }
        __syncthreads();
    }
    __device__ inline int operator()(int c) const {
        unsigned h = hash_home((unsigned)c, mask + 1u);
        for (;;) {
            const int k = key[h];
            if (k == c) return val[h];
            if (k == -1) return -1;
            h = (h + 1) & mask;
        }
    }
};
// The words of row lr0 of agglomerate p (matrix b), visited by a WAVEFRONT: lanes take the entries of the row of the global
// matrix, then, element by element of the row's dof, the dofs of the element (coalesced reads of the element-matrix row).
// F(word, position): every word carries a position that does not depend on the lane that visits it (entry index in the row of
// A / element ordinal and dof slot), so sums over (word, position) are order-independent and two agglomerates can be compared
// position by position.  `other` (verification): the same walk over a second agglomerate in lockstep; returns false at the
// first difference.
__device__ inline unsigned long long ai_mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct AiRow {      // what identifies one side of a walk
    int b, p;
    const AiTable *loc;
};
template <bool PAIR, class F>
__device__ inline bool ai_row_walk(const AeInputs &v, const AiRow &x, const AiRow &y, int lr0, int lane, F &&f) {
    const int gx = v.ae2d_J[v.ae2d_I[x.p] + lr0], gy = PAIR ? v.ae2d_J[v.ae2d_I[y.p] + lr0] : 0;
    bool ok = true;
    if (lane == 0) {
        const int64_t vx = v.voff[x.b];
        const unsigned long long w0 = (unsigned long long)(unsigned short)(v.perm ? v.perm[vx + lr0] : (short)lr0);
        const unsigned long long w1 = v.x0c ? (unsigned long long)__double_as_longlong(v.x0c[vx + lr0]) : 0ull;
        const unsigned long long w2 = v.has_A ? (unsigned long long)(unsigned char)v.flags[gx] : 0ull;
        if (PAIR) {
            const int64_t vy = v.voff[y.b];
            ok = w0 == (unsigned long long)(unsigned short)(v.perm ? v.perm[vy + lr0] : (short)lr0) &&
                 w1 == (v.x0c ? (unsigned long long)__double_as_longlong(v.x0c[vy + lr0]) : 0ull) &&
                 w2 == (v.has_A ? (unsigned long long)(unsigned char)v.flags[gy] : 0ull);
        } else {
            f(w0, 1ull); f(w1, 2ull); f(w2, 3ull);
        }
    }
    if (v.has_A) {
        const roff_t ax = v.Arow[gx], nx = v.Arow[gx + 1] - ax;
        const roff_t ay = PAIR ? v.Arow[gy] : 0, ny = PAIR ? v.Arow[gy + 1] - ay : 0;
        if (PAIR && nx != ny) return false;
        for (roff_t k = lane; k < nx; k += 64) {
            const int c = v.Acol[ax + k];
            const int lc = (*x.loc)(c);
            // (an entry whose column is outside the agglomerate is skipped by the assembly: only its being outside counts)
            const unsigned long long w = lc < 0 ? ~0ull : (((unsigned long long)(unsigned)lc << 8) | (unsigned long long)(unsigned char)v.flags[c] | (c == gx ? 1ull << 40 : 0ull));
            const unsigned long long a = lc < 0 ? 0ull : (unsigned long long)__double_as_longlong(v.Aval[ax + k]);
            if (PAIR) {
                const int c2 = v.Acol[ay + k];
                const int l2 = (*y.loc)(c2);
                const unsigned long long w2 = l2 < 0 ? ~0ull : (((unsigned long long)(unsigned)l2 << 8) | (unsigned long long)(unsigned char)v.flags[c2] | (c2 == gy ? 1ull << 40 : 0ull));
                const unsigned long long a2 = l2 < 0 ? 0ull : (unsigned long long)__double_as_longlong(v.Aval[ay + k]);
                ok = ok && w == w2 && a == a2;
            } else {
                f(w, (1ull << 40) + 2 * (unsigned long long)k);
                f(a, (1ull << 40) + 2 * (unsigned long long)k + 1);
            }
        }
    }
    const int qx = v.d2e_I[gx], cx = v.d2e_I[gx + 1] - qx;
    const int qy = PAIR ? v.d2e_I[gy] : 0;
    if (PAIR && cx != v.d2e_I[gy + 1] - qy) return false;
    for (int q = 0; q < cx; ++q) {      // (wave-uniform)
        const int e = v.d2e_J[qx + q];
        const bool in = v.part[e] == x.p;
        int e2 = 0;
        if (PAIR) {
            e2 = v.d2e_J[qy + q];
            if (in != (v.part[e2] == y.p)) return false;
        }
        if (!in) continue;
        const int eb = v.e2d_I[e], nd = v.e2d_I[e + 1] - eb;
        const int eb2 = PAIR ? v.e2d_I[e2] : 0;
        if (PAIR && nd != v.e2d_I[e2 + 1] - eb2) return false;
        // the dof's slot in the element: found by the lanes
        int kk = nd, kk2 = nd;
        for (int j0 = 0; j0 < nd && kk == nd; j0 += 64) {
            const unsigned long long m = __ballot(j0 + lane < nd && v.e2d_J[eb + j0 + lane] == gx);
            if (m) kk = j0 + __builtin_ctzll(m);
        }
        if (PAIR) {
            for (int j0 = 0; j0 < nd && kk2 == nd; j0 += 64) {
                const unsigned long long m = __ballot(j0 + lane < nd && v.e2d_J[eb2 + j0 + lane] == gy);
                if (m) kk2 = j0 + __builtin_ctzll(m);
            }
            if (kk != kk2) return false;
        } else if (lane == 0) {
            f(((unsigned long long)(unsigned)nd << 32) | (unsigned)kk, (2ull << 40) + ((unsigned long long)q << 20));
        }
        const double *M = v.elval + v.eloff[e] + (size_t)kk * nd;
        const double *M2 = PAIR ? v.elval + v.eloff[e2] + (size_t)kk * nd : nullptr;
        for (int jj = lane; jj < nd; jj += 64) {
            const int d = v.e2d_J[eb + jj];
            const unsigned long long w = ((unsigned long long)(unsigned)v.elem_ldof[eb + jj] << 8) |
                                         (v.has_A ? (unsigned long long)(unsigned char)v.flags[d] | (d == gx ? 1ull << 40 : 0ull) : 0ull);
            const unsigned long long a = (unsigned long long)__double_as_longlong(M[jj]);
            if (PAIR) {
                const int d2 = v.e2d_J[eb2 + jj];
                const unsigned long long w2 = ((unsigned long long)(unsigned)v.elem_ldof[eb2 + jj] << 8) |
                                              (v.has_A ? (unsigned long long)(unsigned char)v.flags[d2] | (d2 == gy ? 1ull << 40 : 0ull) : 0ull);
                ok = ok && w == w2 && a == (unsigned long long)__double_as_longlong(M2[jj]);
            } else {
                const unsigned long long pos = (3ull << 40) + ((unsigned long long)q << 20) + 2 * (unsigned long long)jj;
                f(w, pos);
                f(a, pos + 1);
            }
        }
    }
    return ok;
}
constexpr int AI_NT = 256;
__global__ __launch_bounds__(AI_NT) void asm_hash_kernel(AeInputs v, int hsize, unsigned long long *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char ai_lds[];
    __shared__ unsigned long long red[2][AI_NT / 64];
    const int b = blockIdx.x, p = v.ae0 + b, n = v.ns[b], tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    AiTable loc{(int *)ai_lds, (short *)(ai_lds + 4 * (size_t)hsize), (unsigned)(hsize - 1)};
    if (v.has_A) loc.build(v, p, n, tid, AI_NT);
    const AiRow x{b, p, &loc};
    unsigned long long h1 = 0, h2 = 0;
    for (int lr0 = blockIdx.y * (AI_NT / 64) + wv; lr0 < n; lr0 += (AI_NT / 64) * gridDim.y) {
        const unsigned long long rowtag = (unsigned long long)(lr0 + 1) * 0xC2B2AE3D27D4EB4Full;
        ai_row_walk<false>(v, x, x, lr0, lane, [&](unsigned long long w, unsigned long long pos) {
            const unsigned long long k = ai_mix(w + 0x9E3779B97F4A7C15ull * (pos + 1) + rowtag);
            h1 += k;
            h2 += (k >> 32) * (k & 0xffffffffull);      // (second sum: the product of the halves of the mixed word; a full second mix was half of the kernel)
        });
    }
    if (tid == 0 && blockIdx.y == 0) { h1 += ai_mix((unsigned long long)n + 0x4444444444444444ull); h2 += ai_mix((unsigned long long)n ^ 0x7777777777777777ull); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); }
    if (lane == 0) { red[0][wv] = h1; red[1][wv] = h2; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long s1 = 0, s2 = 0;
        for (int q = 0; q < AI_NT / 64; ++q) { s1 += red[0][q]; s2 += red[1][q]; }
        atomicAdd(out + 2 * (size_t)b, s1);
        atomicAdd(out + 2 * (size_t)b + 1, s2);
    }
}
__global__ void scatter_int_kernel(int n, const int *__restrict__ idx, const int *__restrict__ src, int *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}
__global__ __launch_bounds__(AI_NT) void asm_verify_kernel(AeInputs v, int hsize, const int *__restrict__ rep, int *__restrict__ differ) {
    extern __shared__ __align__(16) unsigned char ai_lds[];
    const int b = blockIdx.x, r0 = rep[b], tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (r0 == b) return;
    const int n = v.ns[b];
    if (n != v.ns[r0]) { if (tid == 0) differ[b] = 1; return; }
    const size_t tb = 6 * (size_t)hsize;      // bytes of one table (keys + values), a multiple of 8
    AiTable locb{(int *)ai_lds, (short *)(ai_lds + 4 * (size_t)hsize), (unsigned)(hsize - 1)};
    AiTable locr{(int *)(ai_lds + tb), (short *)(ai_lds + tb + 4 * (size_t)hsize), (unsigned)(hsize - 1)};
    if (v.has_A) { locb.build(v, v.ae0 + b, n, tid, AI_NT); locr.build(v, v.ae0 + r0, n, tid, AI_NT); }
    const AiRow x{b, v.ae0 + b, &locb}, y{r0, v.ae0 + r0, &locr};
    bool ok = true;
    for (int lr0 = blockIdx.y * (AI_NT / 64) + wv; lr0 < n; lr0 += (AI_NT / 64) * gridDim.y)
        ok = ai_row_walk<true>(v, x, y, lr0, lane, [](unsigned long long, unsigned long long) {}) && ok;
    if (!ok) differ[b] = 1;
}

// Classes of identical agglomerates BEFORE their matrices are built (eig.hip, "Duplicate agglomerate matrices"): the fused
// kernel below makes the scaled matrix, its scaling and its band from the agglomerate's sparse rows (RW slots of column +
// value per row) and its row order alone, so agglomerates whose rows and order agree bit for bit get identical matrices --
// only one member of a class is built (and factored, and iterated): DdSource kind 0.
void ae_build(hipStream_t s, const DevRelations &rel, const DCsr *A, const DevElmats &el, int ae0,
              EigBatch &batch, bool scale, double *Dout, const RowsSpan *rows, AeClasses *classes) {
    if (!batch.count) return;
    batch.has_perm = false;
    batch.has_bw = false;
    if (A && el.algebraic) {
        double bytes = 0.0;
        for (int n : batch.h_n) bytes += 8.0 * (double)n * n;
        profiler().begin(s);
        hipLaunchKernelGGL(el.algebraic == 2 ? ae_window_kernel : ae_extract_kernel, dim3(batch.count),
                           dim3(ASM_NT), 0, s, ae0, batch.n.p, batch.moff.p, batch.W.p, rel.ae2d_I.p, rel.ae2d_J.p,
                           rel.d2ae_I.p, rel.d2ae_J.p, rel.dof_id_inAE.p, A->rowptr.p, A->col.p, A->val.p);
        SA_HIP_CHECK(hipGetLastError());
        profiler().end(s, "ae_extract", bytes, 0.0);
        if (scale) ae_scale(s, batch, Dout);
        return;
    }
    constexpr bool no_fused = false;
    size_t lds = 0;
    int RW = 0;
    // rows ordered by global dof number for the banded factorisation of the few-eigenpairs path (the
    // fused kernel below, or the plain assembly followed by the two-kernel scaling)
    constexpr bool use_perm = true;
    const bool split_scale = batch.count <= 2048 && batch.max_n >= 1024;      // (what ae_scale will pick)
    if (use_perm && scale && eig_batch_takes_subspace(batch) && batch.max_n <= 16384 && ((A && !no_fused) || split_scale)) {
        const size_t rows_total = (size_t)batch.h_voff[batch.count];
        if (batch.perm.n < rows_total) { batch.perm.alloc(rows_total); batch.iperm.alloc(rows_total); }
        constexpr int box_order = 1;
        hipLaunchKernelGGL(ae_perm_kernel, dim3(batch.count), dim3(256), 2 * sizeof(int) * (size_t)batch.max_n, s, ae0,
                           batch.n.p, batch.voff.p, rel.ae2d_I.p, rel.ae2d_J.p, batch.perm.p, batch.iperm.p, box_order);
        batch.has_perm = true;
    }
    if (A && !no_fused) {
        if (A->max_row < 0) A->max_row = csr_max_row(s, *A);
        RW = A->max_row;
        const size_t n = (size_t)batch.max_n;
        lds = 8 * (n * RW + 2 * n + (AB_NT / 64) * n) + 2 * n * RW + 4 * n + 64;
    }
    if (!A || no_fused || lds > 160 * 1024 - 256 || batch.max_n > 32767) {
        if (!split_scale) batch.has_perm = false;       // (the one-kernel scaling works in agglomerate order)
        const bool band_asm = options().band_assembly != 0;
        // (also with the global matrix at hand -- level 0 of Q2 elasticity, whose agglomerates do not fit the fused
        // kernel's LDS: an entry copied from A couples two dofs of an element of this agglomerate, so the band of the
        // element matrices holds it)
        const bool banded = band_asm && batch.has_perm && split_scale && scale && eig_ss_band_enabled();
        // classes of identical agglomerates on the INPUTS of the assembly (AeInputs above): only their first members are assembled
        // and scaled -- as the batch of the representatives over the same workspace, through the same kernels the whole batch
        // would take (the choice between the one-kernel and the spread scaling is the whole batch's)
        if (classes && scale && banded && !Dout && batch.count >= 16) {
            classes->searched = true;
            AeInputs v{batch.n.p, batch.voff.p, batch.has_perm ? batch.perm.p : nullptr, batch.has_x0c ? batch.x0c.p : nullptr,
                       rel.ae2d_I.p, rel.ae2d_J.p, rel.d2ae_I.p, rel.d2ae_J.p, rel.dof_id_inAE.p, rel.flags.p, rel.d2e_I.p, rel.d2e_J.p,
                       rel.part.p, rel.e2d_I.p, rel.e2d_J.p, rel.elem_ldof.p, el.off.p, el.val.p, A ? 1 : 0,
                       A ? A->rowptr.p : nullptr, A ? A->col.p : nullptr, A ? A->val.p : nullptr, ae0};
            profiler().begin(s);
            const int ny = std::max(1, std::min(div_up(batch.max_n, 256), 4096 / std::max(1, batch.count)));
            DBuf<unsigned long long> hash(2 * (size_t)batch.count);
            hash.zero(s);
            int hsize = 64;
            while (hsize < batch.max_n + batch.max_n / 2) hsize <<= 1;
            const size_t tbytes = A ? 6 * (size_t)hsize : 0;      // (the tables are needed for the rows of the global matrix only)
            SA_REQUIRE(2 * tbytes <= 150 * 1024, "agglomerate too large for the class search");
            static bool attr_ai = false;
            if (!attr_ai) {
                SA_HIP_CHECK(hipFuncSetAttribute((const void *)asm_hash_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                SA_HIP_CHECK(hipFuncSetAttribute((const void *)asm_verify_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                attr_ai = true;
            }
            hipLaunchKernelGGL(asm_hash_kernel, dim3(batch.count, ny), dim3(256), tbytes, s, v, hsize, hash.p);
            SA_HIP_CHECK(hipGetLastError());
            auto hh = hash.to_host(s);
            std::vector<int> rep;
            const int nuniq = eig_dedupe_group(hh.data(), batch.count, rep);
            bool found = (long)nuniq * 4 <= (long)batch.count * 3;
            if (found) {
                DBuf<int> d_rep, differ((size_t)batch.count);
                d_rep.from_host(rep, s);
                differ.zero(s);
                hipLaunchKernelGGL(asm_verify_kernel, dim3(batch.count, ny), dim3(256), 2 * tbytes, s, v, hsize, d_rep.p, differ.p);
                SA_HIP_CHECK(hipGetLastError());
                auto hd = differ.to_host(s);
                for (int i = 0; i < batch.count; ++i)
                    if (hd[i]) rep[i] = i;
            }
            profiler().end(s, "eig_dedupe", 0.0, 0.0);
            if (found) {
                DdClasses &cl = classes->cls;
                cl.reps.clear();
                std::vector<int> pos((size_t)batch.count, -1);
                for (int i = 0; i < batch.count; ++i)
                    if (rep[i] == i) { pos[i] = (int)cl.reps.size(); cl.reps.push_back(i); }
                cl.rep_of.resize((size_t)batch.count);
                for (int i = 0; i < batch.count; ++i) cl.rep_of[i] = pos[rep[i]];
                if ((options().debug & 1)) std::fprintf(stderr, "duplicate agglomerates (assembly inputs): %d distinct of %d\n", (int)cl.reps.size(), batch.count);
                EigBatch cb;
                eig_batch_compact(s, cb, batch, cl.reps);
                std::vector<int> ids(cl.reps.size());
                for (size_t q = 0; q < ids.size(); ++q) ids[q] = ae0 + cl.reps[q];
                DBuf<int> d_ids, d_reps;
                d_ids.from_host(ids, s);
                d_reps.from_host(cl.reps, s);
                ae_assemble(s, rel, A, el, ae0, cb, banded, d_ids.p);
                ae_scale(s, cb, nullptr, split_scale ? 1 : 0);
                // the representatives' half bandwidths at their places in the batch; what the matrices consist of from here on --
                // for the comparison with the classes of other chunks -- is their assembled form (DdSource kind 1)
                if (batch.bw.n < (size_t)batch.count) batch.bw.alloc((size_t)batch.count);
                SA_HIP_CHECK(hipMemsetAsync(batch.bw.p, 0, sizeof(int) * (size_t)batch.count, s));
                hipLaunchKernelGGL(scatter_int_kernel, dim3(div_up(cb.count, 256)), dim3(256), 0, s, cb.count, d_reps.p, cb.bw.p, batch.bw.p);
                SA_HIP_CHECK(hipGetLastError());
                batch.has_bw = cb.has_bw;
                batch.has_perm = cb.has_perm;
                classes->src = eig_dedupe_source(batch);
                cl.rep_hash = eig_dedupe_hash_list(s, classes->src, batch.max_n, cl.reps);
                classes->early = true;
                SA_HIP_CHECK(hipStreamSynchronize(s));      // (the lists are freed here)
                return;
            }
        }
        ae_assemble(s, rel, A, el, ae0, batch, banded);
        if (scale) ae_scale(s, batch, Dout);
        return;
    }
    const short *pm = batch.has_perm ? batch.perm.p : nullptr;
    int *bwp = nullptr;           // half bandwidths straight from the sparse rows (scaled matrices of the few-eigenpairs path)
    if (scale && eig_batch_takes_subspace(batch)) {
        if (batch.bw.n < (size_t)batch.count) batch.bw.alloc((size_t)batch.count);
        SA_HIP_CHECK(hipMemsetAsync(batch.bw.p, 0, sizeof(int) * (size_t)batch.count, s));
        bwp = batch.bw.p;
        batch.has_bw = true;
    }
    constexpr bool band_write = true;
    const int band_only = (bwp && band_write && eig_ss_band_enabled()) ? 1 : 0;
    const bool nde8 = el.nde == 8 && batch.count <= 65535;   // (grid.y of the rows kernel)
    const double *rv = nullptr;
    const short *rc = nullptr;
    if (nde8) launch_rows8(s, rel, *A, el, ae0, batch, RW, rv, rc, rows);
    // classes of identical agglomerates: only their first members are built (the caller runs the eigensolvers on those);
    // not when D is wanted for every agglomerate (keep_debug)
    DBuf<int> only;
    int nbuild = batch.count;
    if (classes && nde8 && bwp && !Dout && !batch.has_x0c) {
        DdSource &src = classes->src;
        src = DdSource();
        src.kind = 0;
        src.ns = batch.n.p; src.voff = batch.voff.p; src.moff = batch.moff.p;
        src.perm = pm; src.rvals = rv; src.rcols = rc; src.RW = RW;
        classes->searched = true;
        classes->early = eig_dedupe_find(s, src, batch.count, batch.max_n, classes->cls);
        if (classes->early) {
            only.from_host(classes->cls.reps, s);
            nbuild = (int)classes->cls.reps.size();
        }
    }
    auto launch = [&](auto kern) {
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        hipLaunchKernelGGL(kern, dim3(nbuild), dim3(AB_NT), lds, s, ae0, RW, batch.n.p, batch.moff.p,
                           batch.voff.p, batch.W.p, batch.dis.p, Dout, rel.ae2d_I.p, rel.ae2d_J.p,
                           rel.d2ae_I.p, rel.d2ae_J.p, rel.dof_id_inAE.p, rel.flags.p, rel.d2e_I.p,
                           rel.d2e_J.p, rel.part.p, rel.e2d_I.p, rel.e2d_J.p, el.off.p, el.val.p,
                           A->rowptr.p, A->col.p, A->val.p, rv, rc, pm, bwp, band_only, only.p);
    };
    double bytes = 0.0;
    for (int n : batch.h_n) bytes += 8.0 * (double)n * n;
    profiler().begin(s);
    if (scale) {
        if (nde8) launch(ae_build_kernel<true, 8, true>); else launch(ae_build_kernel<true, 0, false>);
    } else {
        if (nde8) launch(ae_build_kernel<false, 8, true>); else launch(ae_build_kernel<false, 0, false>);
    }
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "ae_build", bytes * (double)nbuild / (double)batch.count, 0.0);
    if (only.n) SA_HIP_CHECK(hipStreamSynchronize(s));      // (the list is freed here)
}

// E_e = P_loc^T A_e P_loc from the DENSE image of A_e (coarse levels, elements that are not 8-dof hexes), in two
// launches; T = A_e P_loc goes through a global scratch block (n x k_e).
// coarse_elmat_T_kernel, grid (agglomerates, row chunks): a thread takes a row i and ALL the columns of a MIS at once
// (eight accumulators at a time), so that a column of A_e is read once per MIS -- one thread per (row, column) re-read
// A_e k times (38 MB per 2 187-row agglomerate of config 5, far beyond the L2: 2.5 TB/s of re-reads, 1.25 s per step)
// -- and one workgroup per agglomerate left most of the card idle.  Per (row, column) the products are added in the
// same order as before (q ascending).
constexpr int CE_KC = 8;
__global__ __launch_bounds__(ASM_NT) void coarse_elmat_T_kernel(
    int ae0, const int *__restrict__ ns, const int64_t *__restrict__ moff,
    const double *__restrict__ W, const int *__restrict__ ae2mis_I,
    const int *__restrict__ ae2mis_J, const int *__restrict__ ae_pair,
    const int64_t *__restrict__ pair_loc_off, const int *__restrict__ pair_loc,
    const int *__restrict__ mis2d_I, const int *__restrict__ mis_k,
    const int64_t *__restrict__ mis_u_off, const double *__restrict__ mis_u,
    const int *__restrict__ colpos_ptr, const int *__restrict__ colpos,
    double *__restrict__ scratch, const int64_t *__restrict__ scratch_off) {
    const int b = blockIdx.x, e = ae0 + b, n = ns[b];
    const double *Am = W + moff[b];
    double *T = scratch + scratch_off[b];
    const int i = blockIdx.y * ASM_NT + threadIdx.x;
    if (blockIdx.y * ASM_NT >= n) return;
    const int ic = min(i, n - 1);      // (threads past the last row repeat it and store nothing)
    const int mb = ae2mis_I[e], me = ae2mis_I[e + 1];
    for (int t = mb; t < me; ++t) {
        const int mis = ae2mis_J[t], k = mis_k[mis];
        if (k == 0) continue;
        const int r = mis2d_I[mis + 1] - mis2d_I[mis];
        const int *loc = pair_loc + pair_loc_off[ae_pair[t]];
        const double *U = mis_u + mis_u_off[mis];
        const int *cp = colpos + colpos_ptr[t];
        for (int v0 = 0; v0 < k; v0 += CE_KC) {
            const int kc = min(CE_KC, k - v0);
            double acc[CE_KC];
#pragma unroll
            for (int v = 0; v < CE_KC; ++v) acc[v] = 0.0;
            int q = 0;
            for (; q + 4 <= r; q += 4) {      // four columns of A_e in flight (U and loc are uniform: scalar loads)
                double a4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a4[u] = Am[(size_t)loc[q + u] * n + ic];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < CE_KC; ++v)
                        if (v < kc) acc[v] = fma(a4[u], U[(size_t)(v0 + v) * r + q + u], acc[v]);
            }
            for (; q < r; ++q) {
                const double a = Am[(size_t)loc[q] * n + ic];
#pragma unroll
                for (int v = 0; v < CE_KC; ++v)
                    if (v < kc) acc[v] = fma(a, U[(size_t)(v0 + v) * r + q], acc[v]);
            }
            if (i < n) {
#pragma unroll
                for (int v = 0; v < CE_KC; ++v)
                    if (v < kc) T[(size_t)cp[v0 + v] * n + i] = acc[v];
            }
        }
    }
}
// E[(cb + v), :] = U[:, v]^T T[loc, :]
__global__ __launch_bounds__(ASM_NT) void coarse_elmat_E_kernel(
    int ae0, const int *__restrict__ ns, const int *__restrict__ ae2mis_I,
    const int *__restrict__ ae2mis_J, const int *__restrict__ ae_pair,
    const int64_t *__restrict__ pair_loc_off, const int *__restrict__ pair_loc,
    const int *__restrict__ mis2d_I, const int *__restrict__ mis_k,
    const int64_t *__restrict__ mis_u_off, const double *__restrict__ mis_u,
    const int *__restrict__ colpos_ptr, const int *__restrict__ colpos,
    const int64_t *__restrict__ out_off, double *__restrict__ out,
    const double *__restrict__ scratch, const int64_t *__restrict__ scratch_off) {
    const int b = blockIdx.x, e = ae0 + b, n = ns[b];
    const double *T = scratch + scratch_off[b];
    double *E = out + out_off[e];
    const int ke = (int)(sqrt((double)(out_off[e + 1] - out_off[e])) + 0.5);
    const int tid = threadIdx.x;
    const int mb = ae2mis_I[e], me = ae2mis_I[e + 1];
    for (int t = mb + blockIdx.y; t < me; t += gridDim.y) {
        const int mis = ae2mis_J[t], k = mis_k[mis];
        if (k == 0) continue;
        const int r = mis2d_I[mis + 1] - mis2d_I[mis];
        const int *loc = pair_loc + pair_loc_off[ae_pair[t]];
        const double *U = mis_u + mis_u_off[mis];
        const int *cp = colpos + colpos_ptr[t];
        for (int idx = tid; idx < k * ke; idx += ASM_NT) {
            const int col = idx % ke, v = idx / ke;
            double sum = 0.0;
            for (int q = 0; q < r; ++q) sum = fma(U[(size_t)v * r + q], T[(size_t)col * n + loc[q]], sum);
            E[(size_t)cp[v] * ke + col] = sum;
        }
    }
}

// The same from the SPARSE rows of A_e (fine level: ~27 entries per row instead of the dense
// n x n image -- neither written nor read): T = A_e P_loc row by row (thread = row, entries
// scattered into the k columns of the entry's MIS; no other thread touches the row), then
// E = P_loc^T T as above.
__global__ __launch_bounds__(ASM_NT) void coarse_elmat_sparse_kernel(
    int ae0, int RW, const int *__restrict__ ns, const int64_t *__restrict__ voff,
    const double *__restrict__ rvals, const short *__restrict__ rcols, const int *__restrict__ ae2mis_I,
    const int *__restrict__ ae2mis_J, const int *__restrict__ ae_pair,
    const int64_t *__restrict__ pair_loc_off, const int *__restrict__ pair_loc,
    const int *__restrict__ mis2d_I, const int *__restrict__ mis_k,
    const int64_t *__restrict__ mis_u_off, const double *__restrict__ mis_u,
    const int *__restrict__ colpos_ptr, const int *__restrict__ colpos,
    const int64_t *__restrict__ out_off, double *__restrict__ out,
    double *__restrict__ scratch, const int64_t *__restrict__ scratch_off, const int *__restrict__ only_flagged) {
    extern __shared__ __align__(16) short dof_tq[];   // [2 n]: MIS slot of the AE and row in that MIS, per local dof
    if (only_flagged && !only_flagged[blockIdx.x]) return;      // (the fallback pass behind coarse_elmat_rows_kernel)
    constexpr int MAXT = 64;                          // MISes of one AE kept in LDS (27 for a box of hexes)
    __shared__ int tk[MAXT], tr[MAXT], tcp[MAXT];
    __shared__ long long tu[MAXT];
    const int b = blockIdx.x, e = ae0 + b, n = ns[b];
    double *T = scratch + scratch_off[b];
    double *E = out + out_off[e];
    const int ke = (int)(sqrt((double)(out_off[e + 1] - out_off[e])) + 0.5);
    const int tid = threadIdx.x;
    const int mb = ae2mis_I[e], me = ae2mis_I[e + 1];
    short *dof_t = dof_tq, *dof_q = dof_tq + n;
    for (int t = mb + tid; t < me && t - mb < MAXT; t += ASM_NT) {
        const int mis = ae2mis_J[t];
        tk[t - mb] = mis_k[mis];
        tr[t - mb] = mis2d_I[mis + 1] - mis2d_I[mis];
        tu[t - mb] = mis_u_off[mis];
        tcp[t - mb] = colpos_ptr[t];
    }
    for (int t = mb; t < me; ++t) {
        const int mis = ae2mis_J[t];
        const int r = mis2d_I[mis + 1] - mis2d_I[mis];
        const int *loc = pair_loc + pair_loc_off[ae_pair[t]];
        for (int q = tid; q < r; q += ASM_NT) {
            dof_t[loc[q]] = (short)(t - mb);
            dof_q[loc[q]] = (short)q;
        }
    }
    for (size_t idx = tid; idx < (size_t)n * ke; idx += ASM_NT) T[idx] = 0.0;
    __syncthreads();
    const size_t rbase = (size_t)voff[b] * RW;
    const bool cached = me - mb <= MAXT;
    for (int i = tid; i < n; i += ASM_NT) {
        const short *rc_i = rcols + rbase + (size_t)i * RW;
        const double *rv_i = rvals + rbase + (size_t)i * RW;
        for (int k = 0; k < RW; ++k) {
            const int c = rc_i[k];
            if (c < 0) continue;
            const double v = rv_i[k];
            const int tl = dof_t[c], q = dof_q[c];
            int km, r, cpo;
            long long uo;
            if (cached) {
                km = tk[tl]; r = tr[tl]; uo = tu[tl]; cpo = tcp[tl];
            } else {
                const int mis = ae2mis_J[mb + tl];
                km = mis_k[mis]; r = mis2d_I[mis + 1] - mis2d_I[mis]; uo = mis_u_off[mis]; cpo = colpos_ptr[mb + tl];
            }
            const double *U = mis_u + uo + q;
            const int *cp = colpos + cpo;
            for (int w = 0; w < km; ++w) {
                double *dst = T + (size_t)cp[w] * n + i;
                *dst = fma(v, U[(size_t)w * r], *dst);
            }
        }
    }
    __syncthreads();
    for (int t = mb; t < me; ++t) {
        const int mis = ae2mis_J[t], k = mis_k[mis];
        if (k == 0) continue;
        const int r = mis2d_I[mis + 1] - mis2d_I[mis];
        const int *loc = pair_loc + pair_loc_off[ae_pair[t]];
        const double *U = mis_u + mis_u_off[mis];
        const int *cp = colpos + colpos_ptr[t];
        for (int idx = tid; idx < k * ke; idx += ASM_NT) {
            const int col = idx % ke, v = idx / ke;
            double sum = 0.0;
            for (int q = 0; q < r; ++q) sum = fma(U[(size_t)v * r + q], T[(size_t)col * n + loc[q]], sum);
            E[(size_t)cp[v] * ke + col] = sum;
        }
    }
}


// The same around the sparsity of T = A_e P_loc (round 3; the kernel above kept T as a dense n x k_e image in global
// memory, zero-filled and read-modify-written one 8-byte entry at a time: 135 GB fetched + 15 GB written per setup of
// the 256^3 problem for 7 GB of input, 22 ms).  A row of a fine-level agglomerate matrix reaches at most NS = 8
// MISes (the 2 x 2 x 2 cells around a vertex of the MIS decomposition), so its row of T has few non-zero entries
// (the coarse dofs of those MISes: 1 - 2 for most rows).  They live in LDS in a packed pool, row after row: per row a
// key word (byte s = the MIS of slot s), a word of starting columns per slot and the row's offset into the pool.
//   count   one thread per row over the cached column indices: distinct MISes -> slots, columns per row; block scan
//   fill    the same walk with the values: products into the pool (the thread owns its row: no conflicts)
//   reduce  one thread per entry of E: the rows of its row-MIS in their fixed order (a deterministic sum), the
//           column-MIS's slot found in the row's key word
// The MISes' row lists and bases are copied to LDS first, in one round of loads.  An agglomerate with a row that
// reaches more than NS MISes, or whose pool / bases exceed the LDS set aside (pool_cap / u_cap doubles), is flagged and
// redone by the kernel above (only_flagged).
constexpr int CE_NS = 8;
__global__ __launch_bounds__(ASM_NT) void coarse_elmat_rows_kernel(
    int ae0, int RW, int pool_cap, int u_cap, const int *__restrict__ ns, const int64_t *__restrict__ voff,
    const double *__restrict__ rvals, const short *__restrict__ rcols, const int *__restrict__ ae2mis_I,
    const int *__restrict__ ae2mis_J, const int *__restrict__ ae_pair,
    const int64_t *__restrict__ pair_loc_off, const int *__restrict__ pair_loc,
    const int *__restrict__ mis2d_I, const int *__restrict__ mis_k,
    const int64_t *__restrict__ mis_u_off, const double *__restrict__ mis_u,
    const int *__restrict__ colpos_ptr, const int *__restrict__ colpos,
    const int64_t *__restrict__ out_off, double *__restrict__ out, int *__restrict__ flagged,
    const int *__restrict__ list = nullptr) {
    extern __shared__ __align__(16) unsigned char ce_lds[];
    constexpr int MAXT = 64, MAXKE = 256;
    __shared__ int tk[MAXT], tr[MAXT];
    __shared__ long long tu[MAXT];
    __shared__ unsigned char col_t[MAXKE], col_w[MAXKE];
    __shared__ int bad, wsum[ASM_NT / 64], total_s;
    // (list: the agglomerates of the batch to compute -- the first members of its classes of identical inputs)
    const int b = list ? list[blockIdx.x] : (int)blockIdx.x, e = ae0 + b, n = ns[b];
    double *E = out + out_off[e];
    const int ke = (int)(sqrt((double)(out_off[e + 1] - out_off[e])) + 0.5);
    const int tid = threadIdx.x;
    const int mb = ae2mis_I[e], me = ae2mis_I[e + 1], nt = me - mb;
    // LDS: pool [pool_cap] doubles, the MIS bases U [u_cap] doubles, keys [n] + column starts [n] 64-bit, row offsets [n]
    // and MIS row lists [n] ints, dof_t / dof_q [n] shorts
    double *pool = (double *)ce_lds;
    double *lU = pool + pool_cap;
    unsigned long long *lkeys = (unsigned long long *)(lU + u_cap);
    unsigned long long *lcols = lkeys + n;
    int *rowoff = (int *)(lcols + n);
    int *lloc = rowoff + n;
    short *dof_t = (short *)(lloc + n), *dof_q = dof_t + n;
    __shared__ int troff[MAXT + 1], tuoff[MAXT + 1];      // prefix sums of the MIS sizes r_t and of r_t k_t
    __shared__ long long tloc[MAXT];                      // where the MIS's row list starts in pair_loc
    if (tid == 0) bad = (nt > MAXT || ke > MAXKE) ? 1 : 0;
    for (int t = tid; t < nt && t < MAXT; t += ASM_NT) {
        const int mis = ae2mis_J[mb + t];
        tk[t] = mis_k[mis];
        tr[t] = mis2d_I[mis + 1] - mis2d_I[mis];
        tu[t] = mis_u_off[mis];
        tloc[t] = pair_loc_off[ae_pair[mb + t]];
        const int *cp = colpos + colpos_ptr[mb + t];
        for (int w = 0; w < tk[t]; ++w)
            if (cp[w] < MAXKE) { col_t[cp[w]] = (unsigned char)t; col_w[cp[w]] = (unsigned char)w; }
    }
    __syncthreads();
    if (bad) {
        if (tid == 0) flagged[b] = 1;
        return;
    }
    if (tid == 0) {      // (at most 64 MISes: a serial scan in LDS)
        int ar = 0, au = 0;
        for (int t = 0; t < nt; ++t) {
            troff[t] = ar;
            tuoff[t] = au;
            ar += tr[t];
            au += tr[t] * tk[t];
        }
        troff[nt] = ar;
        tuoff[nt] = au;
        if (au > u_cap || ar != n) bad = 1;
    }
    __syncthreads();
    if (bad) {
        if (tid == 0) flagged[b] = 1;
        return;
    }
    // the row lists and the bases of all MISes in ONE round of loads (flat index -> MIS by bisection over the prefix sums;
    // the loops below then run out of LDS: with the lists and bases in global memory every step of the reduction was a
    // dependent L2 access, 147 of them in a row for the interior MIS)
    for (int idx = tid; idx < n; idx += ASM_NT) {
        int lo = 0, hi = nt;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (troff[mid] <= idx) lo = mid; else hi = mid; }
        const int q = idx - troff[lo];
        const int d = pair_loc[tloc[lo] + q];
        lloc[idx] = d;
        dof_t[d] = (short)lo;
        dof_q[d] = (short)q;
    }
    for (int idx = tid; idx < tuoff[nt]; idx += ASM_NT) {
        int lo = 0, hi = nt;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tuoff[mid] <= idx) lo = mid; else hi = mid; }
        lU[idx] = mis_u[tu[lo] + (idx - tuoff[lo])];      // (U of a MIS: r x k column-major, copied as it is)
    }
    __syncthreads();
    // count: a contiguous run of rows per thread (the scan below is over threads)
    const size_t rbase = (size_t)voff[b] * RW;
    const int per = (n + ASM_NT - 1) / ASM_NT;
    const int i0 = min(tid * per, n), i1 = min(i0 + per, n);
    bool over = false;
    int mine = 0;
    for (int i = i0; i < i1; ++i) {
        const short *rc_i = rcols + rbase + (size_t)i * RW;
        unsigned long long keys = ~0ull, cols = 0ull;      // byte s: MIS (AE-local) of slot s (0xff free) / first column of slot s
        int cnt = 0, ncol = 0;
        for (int k = 0; k < RW; ++k) {
            const int c = rc_i[k];
            if (c < 0) continue;
            const int tl = dof_t[c];
            const int km = tk[tl];
            if (km == 0) continue;
            bool found = false;
#pragma unroll
            for (int sl = 0; sl < CE_NS; ++sl) found = found || (int)((keys >> (8 * sl)) & 0xffull) == tl;
            if (found) continue;
            if (cnt == CE_NS || ncol + km > 255) { over = true; continue; }
            keys = (keys & ~(0xffull << (8 * cnt))) | ((unsigned long long)tl << (8 * cnt));
            cols |= (unsigned long long)ncol << (8 * cnt);
            ncol += km;
            ++cnt;
        }
        lkeys[i] = keys;
        lcols[i] = cols;
        rowoff[i] = ncol;      // (the count; turned into the offset below)
        mine += ncol;
    }
    // exclusive scan of `mine` over the threads
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += up;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    if (over) bad = 1;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    if (tid == ASM_NT - 1) total_s = base + mine;
    for (int i = i0; i < i1; ++i) {
        const int c = rowoff[i];
        rowoff[i] = base;
        base += c;
    }
    __syncthreads();
    const int total = total_s;
    if (bad || total > pool_cap) {
        if (tid == 0) flagged[b] = 1;
        return;
    }
    for (int idx = tid; idx < total; idx += ASM_NT) pool[idx] = 0.0;
    __syncthreads();
    // fill: row i of T, slot by slot
    for (int i = i0; i < i1; ++i) {
        const short *rc_i = rcols + rbase + (size_t)i * RW;
        const double *rv_i = rvals + rbase + (size_t)i * RW;
        const unsigned long long keys = lkeys[i], cols = lcols[i];
        double *row = pool + rowoff[i];
        for (int k = 0; k < RW; ++k) {
            const int c = rc_i[k];
            if (c < 0) continue;
            const int tl = dof_t[c];
            const int km = tk[tl];
            if (km == 0) continue;
            int col0 = 0;
#pragma unroll
            for (int sl = 0; sl < CE_NS; ++sl)
                if ((int)((keys >> (8 * sl)) & 0xffull) == tl) col0 = (int)((cols >> (8 * sl)) & 0xffull);
            const double v = rv_i[k];
            const double *U = lU + tuoff[tl] + dof_q[c];
            const int r = tr[tl];
            for (int w = 0; w < km; ++w) row[col0 + w] = fma(v, U[w * r], row[col0 + w]);
        }
    }
    __syncthreads();
    // reduce: E[a][c] = sum over the rows i of a's MIS (in the MIS's own order) of U[q_i][w_a] T_i[column of c]
    for (int idx = tid; idx < ke * ke; idx += ASM_NT) {
        const int a = idx / ke, c = idx - a * ke;
        const int t = col_t[a], w = col_w[a], t2 = col_t[c], w2 = col_w[c];
        const int r = tr[t];
        const int *loc = lloc + troff[t];
        const double *U = lU + tuoff[t] + w * r;
        double sum = 0.0;
        for (int q = 0; q < r; ++q) {
            const int i = loc[q];
            const unsigned long long keys = lkeys[i];
            int pos = -1;
#pragma unroll
            for (int sl = 0; sl < CE_NS; ++sl)
                if ((int)((keys >> (8 * sl)) & 0xffull) == t2) pos = sl;
            if (pos >= 0) sum = fma(U[q], pool[rowoff[i] + (int)((lcols[i] >> (8 * pos)) & 0xffull) + w2], sum);
        }
        E[(size_t)a * ke + c] = sum;
    }
}

// Classes of agglomerates with identical coarse element matrices (eig.hip, "Duplicate agglomerate matrices", one stage further
// down the setup): E_e = P_loc^T A_e P_loc is a function of the agglomerate's sparse rows -- whose class the eigenproblem stage
// has established, ae_class -- and of its MISes in their order: per MIS the number of basis vectors, the row list in the
// agglomerate's numbering, the basis itself and the positions of its coarse dofs in the element.  CeIn names those arrays;
// ce_walk visits the words of one agglomerate (a workgroup per agglomerate, threads over the entries of each MIS) either summing
// mixed (word, position) pairs (the 128-bit hash) or comparing them with the first member of the class in lockstep.
struct CeIn {
    const int *ae_class, *ae2mis_I, *ae2mis_J, *ae_pair;
    const int64_t *pair_loc_off;
    const int *pair_loc, *mis2d_I, *mis_k;
    const int64_t *mis_u_off;
    const double *mis_u;
    const int *colpos_ptr, *colpos;
    const int64_t *out_off;
    int ae0;
};
template <bool PAIR, class F>
__device__ inline bool ce_walk(const CeIn &v, int e, int e2, int tid, int nt_, F &&f) {
    const int mb = v.ae2mis_I[e], nm = v.ae2mis_I[e + 1] - mb;
    const int mb2 = PAIR ? v.ae2mis_I[e2] : 0;
    if (PAIR && (nm != v.ae2mis_I[e2 + 1] - mb2 || v.ae_class[e] != v.ae_class[e2] ||
                 v.out_off[e + 1] - v.out_off[e] != v.out_off[e2 + 1] - v.out_off[e2])) return false;
    bool ok = true;
    if (!PAIR && tid == 0) { f((unsigned long long)(unsigned)v.ae_class[e], 1ull); f((unsigned long long)nm, 2ull); f((unsigned long long)(v.out_off[e + 1] - v.out_off[e]), 3ull); }
    for (int t = 0; t < nm; ++t) {      // (workgroup-uniform)
        const int mis = v.ae2mis_J[mb + t], k = v.mis_k[mis], r = v.mis2d_I[mis + 1] - v.mis2d_I[mis];
        const int *loc = v.pair_loc + v.pair_loc_off[v.ae_pair[mb + t]];
        const int *cp = v.colpos + v.colpos_ptr[mb + t];
        const double *U = v.mis_u + v.mis_u_off[mis];
        const unsigned long long tag = (unsigned long long)(t + 1) << 40;
        if (PAIR) {
            const int mis2 = v.ae2mis_J[mb2 + t];
            if (k != v.mis_k[mis2] || r != v.mis2d_I[mis2 + 1] - v.mis2d_I[mis2]) return false;
            const int *loc2 = v.pair_loc + v.pair_loc_off[v.ae_pair[mb2 + t]];
            const int *cp2 = v.colpos + v.colpos_ptr[mb2 + t];
            const long long *U1 = (const long long *)U, *U2 = (const long long *)(v.mis_u + v.mis_u_off[mis2]);
            for (int i = tid; i < k; i += nt_) ok = ok && cp[i] == cp2[i];
            for (int i = tid; i < r; i += nt_) ok = ok && loc[i] == loc2[i];
            for (int i = tid; i < r * k; i += nt_) ok = ok && U1[i] == U2[i];
        } else {
            if (tid == 0) f(((unsigned long long)(unsigned)k << 32) | (unsigned)r, tag);
            for (int i = tid; i < k; i += nt_) f((unsigned long long)(unsigned)cp[i], tag + (1ull << 36) + (unsigned long long)i);
            for (int i = tid; i < r; i += nt_) f((unsigned long long)(unsigned)loc[i], tag + (2ull << 36) + (unsigned long long)i);
            for (int i = tid; i < r * k; i += nt_) f((unsigned long long)__double_as_longlong(U[i]), tag + (3ull << 36) + (unsigned long long)i);
        }
    }
    return ok;
}
__global__ __launch_bounds__(256) void ce_hash_kernel(CeIn v, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned long long h1 = 0, h2 = 0;
    ce_walk<false>(v, v.ae0 + b, 0, tid, 256, [&](unsigned long long w, unsigned long long pos) {
        const unsigned long long k = ai_mix(w + 0x9E3779B97F4A7C15ull * (pos + 1));
        h1 += k;
        h2 += (k >> 32) * (k & 0xffffffffull);      // (second sum: the product of the halves of the mixed word; a full second mix was half of the kernel)
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = h1; red[1][tid >> 6] = h2; }
    __syncthreads();
    if (tid == 0) {
        out[2 * (size_t)b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        out[2 * (size_t)b + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
__global__ __launch_bounds__(256) void ce_verify_kernel(CeIn v, const int *__restrict__ rep, int *__restrict__ differ) {
    const int b = blockIdx.x, r0 = rep[b];
    if (r0 == b) return;
    if (!ce_walk<true>(v, v.ae0 + b, v.ae0 + r0, threadIdx.x, 256, [](unsigned long long, unsigned long long) {})) differ[b] = 1;
}
// E of every member of a class from the class's first member
__global__ __launch_bounds__(256) void ce_copy_kernel(int ae0, const int *__restrict__ rep, const int64_t *__restrict__ out_off,
                                                      double *__restrict__ out) {
    const int b = blockIdx.x, r0 = rep[b];
    if (r0 == b) return;
    const int64_t n = out_off[ae0 + b + 1] - out_off[ae0 + b];
    const double *src = out + out_off[ae0 + r0];
    double *dst = out + out_off[ae0 + b];
    for (int64_t i = blockIdx.y * 256 + threadIdx.x; i < n; i += 256 * (int64_t)gridDim.y) dst[i] = src[i];
}

void coarse_elmats_sparse(hipStream_t s, const DevRelations &rel, int ae0, const EigBatch &batch, int RW,
                          const double *rv, const short *rc, const int *mis_k, const int64_t *mis_u_off,
                          const double *mis_u, const int *colpos_ptr, const int *colpos, const int64_t *out_off,
                          double *out, double *scratch, const int64_t *scratch_off, int kmax, const int *ae_class) {
    if (!batch.count) return;
    // classes of identical inputs: only their first members are computed, the others copied (ae_class: the classes of the
    // agglomerates' sparse rows from the eigenproblem stage, per agglomerate of the level; null: every agglomerate on its own)
    std::vector<int> rep;
    DBuf<int> d_rep, d_list;
    int ncompute = batch.count;
    if (ae_class && batch.count >= 16) {
        profiler().begin(s);
        CeIn v{ae_class, rel.ae2mis_I.p, rel.ae2mis_J.p, rel.ae_pair.p, rel.pair_loc_off.p, rel.pair_loc.p, rel.mis2d_I.p, mis_k,
               mis_u_off, mis_u, colpos_ptr, colpos, out_off, ae0};
        DBuf<unsigned long long> hash(2 * (size_t)batch.count);
        hipLaunchKernelGGL(ce_hash_kernel, dim3(batch.count), dim3(256), 0, s, v, hash.p);
        SA_HIP_CHECK(hipGetLastError());
        auto hh = hash.to_host(s);
        const int nuniq = eig_dedupe_group(hh.data(), batch.count, rep);
        if ((long)nuniq * 4 <= (long)batch.count * 3) {
            DBuf<int> differ((size_t)batch.count);
            d_rep.from_host(rep, s);
            differ.zero(s);
            hipLaunchKernelGGL(ce_verify_kernel, dim3(batch.count), dim3(256), 0, s, v, d_rep.p, differ.p);
            SA_HIP_CHECK(hipGetLastError());
            auto hd = differ.to_host(s);
            std::vector<int> list;
            for (int i = 0; i < batch.count; ++i) {
                if (hd[i]) rep[i] = i;
                if (rep[i] == i) list.push_back(i);
            }
            d_rep.from_host(rep, s);
            d_list.from_host(list, s);
            ncompute = (int)list.size();
            if (options().debug & 1) std::fprintf(stderr, "coarse element matrices: %d distinct of %d\n", ncompute, batch.count);
        } else {
            rep.clear();
        }
        profiler().end(s, "eig_dedupe", 0.0, 0.0);
    }
    profiler().begin(s);
    // the packed rows of T of one agglomerate in LDS: a pool of 6 doubles per row on average (+ keys, column starts, offsets
    // and dof maps: 24 bytes per row); an agglomerate that needs more is redone by the dense-T kernel
    constexpr bool old_only = false;
    const int pool_cap = 6 * batch.max_n, u_cap = 3 * batch.max_n;
    const size_t lds = 8 * (size_t)(pool_cap + u_cap) + 28 * (size_t)batch.max_n + 16;
    if (options().debug & 2)
        std::fprintf(stderr, "coarse_elmats_sparse: %d agglomerates, max n %d, kmax %d, RW %d, LDS %zu\n", batch.count, batch.max_n, kmax, RW, lds);
    const int *only = nullptr;
    DBuf<int> flagged;
    if (!old_only && kmax >= 1 && kmax < 256 && lds <= 64 * 1024) {
        flagged.alloc((size_t)batch.count);
        flagged.zero(s);
        hipLaunchKernelGGL(coarse_elmat_rows_kernel, dim3(ncompute), dim3(ASM_NT), lds, s, ae0, RW, pool_cap, u_cap, batch.n.p,
                           batch.voff.p, rv, rc, rel.ae2mis_I.p, rel.ae2mis_J.p, rel.ae_pair.p, rel.pair_loc_off.p,
                           rel.pair_loc.p, rel.mis2d_I.p, mis_k, mis_u_off, mis_u, colpos_ptr, colpos, out_off, out,
                           flagged.p, rep.empty() ? (const int *)nullptr : d_list.p);
        SA_HIP_CHECK(hipGetLastError());
        only = flagged.p;
        profiler().end(s, "coarse_elmats_rows", 0.0, 0.0);
        profiler().begin(s);
    }
    hipLaunchKernelGGL(coarse_elmat_sparse_kernel, dim3(batch.count), dim3(ASM_NT), 4 * (size_t)batch.max_n + 16, s,
                       ae0, RW, batch.n.p, batch.voff.p, rv, rc, rel.ae2mis_I.p, rel.ae2mis_J.p, rel.ae_pair.p,
                       rel.pair_loc_off.p, rel.pair_loc.p, rel.mis2d_I.p, mis_k, mis_u_off, mis_u, colpos_ptr,
                       colpos, out_off, out, scratch, scratch_off, only);
    SA_HIP_CHECK(hipGetLastError());
    if (!rep.empty() && only) {      // (classes: the fallback kernel above ran for the flagged first members only)
        const int ny = std::max(1, std::min(16, 65536 / std::max(1, batch.count)));
        hipLaunchKernelGGL(ce_copy_kernel, dim3(batch.count, ny), dim3(256), 0, s, ae0, d_rep.p, out_off, out);
        SA_HIP_CHECK(hipGetLastError());
        SA_HIP_CHECK(hipStreamSynchronize(s));      // (the lists are freed here)
    }
    profiler().end(s, "coarse_elmats", 0.0, 0.0);
}

void coarse_elmats(hipStream_t s, const DevRelations &rel, int ae0, const EigBatch &batch,
                   const int *mis_k, const int64_t *mis_u_off, const double *mis_u,
                   const int *colpos_ptr, const int *colpos, const int64_t *out_off, double *out,
                   double *scratch, const int64_t *scratch_off) {
    if (!batch.count) return;
    profiler().begin(s);
    hipLaunchKernelGGL(coarse_elmat_T_kernel, dim3(batch.count, div_up(batch.max_n, ASM_NT)), dim3(ASM_NT), 0, s, ae0, batch.n.p,
                       batch.moff.p, batch.W.p, rel.ae2mis_I.p, rel.ae2mis_J.p, rel.ae_pair.p,
                       rel.pair_loc_off.p, rel.pair_loc.p, rel.mis2d_I.p, mis_k, mis_u_off, mis_u,
                       colpos_ptr, colpos, scratch, scratch_off);
    hipLaunchKernelGGL(coarse_elmat_E_kernel, dim3(batch.count, 4), dim3(ASM_NT), 0, s, ae0, batch.n.p,
                       rel.ae2mis_I.p, rel.ae2mis_J.p, rel.ae_pair.p,
                       rel.pair_loc_off.p, rel.pair_loc.p, rel.mis2d_I.p, mis_k, mis_u_off, mis_u,
                       colpos_ptr, colpos, out_off, out, scratch, scratch_off);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "coarse_elmats", 0.0, 0.0);
}

}  // namespace saamge_amd
