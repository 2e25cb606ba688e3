// Agglomerate matrix assembly on gfx950: one workgroup per AE, the dense n x n matrix is
// written straight into the eigensolver's workspace (column-major).  Each AE-local row is
// owned by one thread and its element contributions are added in ascending element id, so
// the sums are bit-reproducible and match the reference's accumulation order.
#include "assemble.h"

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
    const int *__restrict__ Arow, const int *__restrict__ Acol, const double *__restrict__ Aval) {
    const int b = blockIdx.x, p = ae0 + b, n = ns[b];
    double *Wm = W + moff[b];
    const int tid = threadIdx.x;
    const size_t nn = (size_t)n * n;
    for (size_t idx = tid; idx < nn; idx += ASM_NT) Wm[idx] = 0.0;
    __syncthreads();
    const int *aedofs = ae2d_J + ae2d_I[p];
    for (int lr = tid; lr < n; lr += ASM_NT) {
        const int g = aedofs[lr];
        const int fg = has_A ? flags[g] : 0;
        if (has_A) {
            // entries copied from the global matrix (aggregates.cpp:930-934)
            for (int k = Arow[g]; k < Arow[g + 1]; ++k) {
                const int c = Acol[k];
                int idx = -1;
                for (int q = d2ae_I[c]; q < d2ae_I[c + 1]; ++q)
                    if (d2ae_J[q] == p) { idx = q; break; }
                if (idx < 0) continue;  // neighbour not in this AE
                const int fc = flags[c];
                const bool assembled = (fg & 1) && (fc & 1) && (!((fg | fc) & 2) || c == g);
                if (!assembled) {
                    const double v = Aval[k];
                    if (v != 0.0) Wm[(size_t)dof_id_inAE[idx] * n + lr] = v;
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
                if (assembled) Wm[(size_t)elem_ldof[eb + jj] * n + lr] += M[jj];
            }
        }
    }
}

void ae_assemble(hipStream_t s, const DevRelations &rel, const DCsr *A, const DevElmats &el,
                 int ae0, EigBatch &batch) {
    if (!batch.count) return;
    double bytes = 0.0;
    for (int n : batch.h_n) bytes += 8.0 * (double)n * n;
    profiler().begin(s);
    hipLaunchKernelGGL(ae_assemble_kernel, dim3(batch.count), dim3(ASM_NT), 0, s, ae0, batch.n.p,
                       batch.moff.p, batch.W.p, rel.ae2d_I.p, rel.ae2d_J.p, rel.d2ae_I.p,
                       rel.d2ae_J.p, rel.dof_id_inAE.p, rel.flags.p, rel.d2e_I.p, rel.d2e_J.p,
                       rel.part.p, rel.e2d_I.p, rel.e2d_J.p, rel.elem_ldof.p, el.off.p, el.val.p,
                       A ? 1 : 0, A ? A->rowptr.p : nullptr, A ? A->col.p : nullptr,
                       A ? A->val.p : nullptr);
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
    double *dg = lds, *dis = lds + n, *part = lds + 2 * n;  // part[4][64]
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += ASM_NT) dg[i] = Wm[(size_t)i * n + i];
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
                for (int u = 0; u < 8; ++u)
                    if (a[u] != 0.0) sum += fabs(a[u]) * sqrt(dr / dg[j + u]);
            }
            for (; j < ce; ++j) {
                const double a = Wm[(size_t)j * n + r];
                if (a != 0.0) sum += fabs(a) * sqrt(dr / dg[j]);
            }
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

void ae_scale(hipStream_t s, EigBatch &batch, double *Dout) {
    if (!batch.count) return;
    double bytes = 0.0;
    for (int n : batch.h_n) bytes += 24.0 * (double)n * n;
    profiler().begin(s);
    hipLaunchKernelGGL(ae_scale_kernel, dim3(batch.count), dim3(ASM_NT),
                       (2 * (size_t)batch.max_n + 256) * sizeof(double), s, batch.n.p, batch.moff.p,
                       batch.voff.p, batch.W.p, batch.dis.p, Dout);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "ae_scale", bytes, 0.0);
}

// E_e = P_loc^T A_e P_loc.  T = A_e P_loc goes through a global scratch block (n x k_e).
__global__ __launch_bounds__(ASM_NT) void coarse_elmat_kernel(
    int ae0, const int *__restrict__ ns, const int64_t *__restrict__ moff,
    const double *__restrict__ W, const int *__restrict__ ae2mis_I,
    const int *__restrict__ ae2mis_J, const int *__restrict__ ae_pair,
    const int64_t *__restrict__ pair_loc_off, const int *__restrict__ pair_loc,
    const int *__restrict__ mis2d_I, const int *__restrict__ mis_k,
    const int64_t *__restrict__ mis_u_off, const double *__restrict__ mis_u,
    const int *__restrict__ colpos_ptr, const int *__restrict__ colpos,
    const int64_t *__restrict__ out_off, double *__restrict__ out,
    double *__restrict__ scratch, const int64_t *__restrict__ scratch_off) {
    const int b = blockIdx.x, e = ae0 + b, n = ns[b];
    const double *Am = W + moff[b];
    double *T = scratch + scratch_off[b];
    double *E = out + out_off[e];
    const int ke = (int)(sqrt((double)(out_off[e + 1] - out_off[e])) + 0.5);
    const int tid = threadIdx.x;
    const int mb = ae2mis_I[e], me = ae2mis_I[e + 1];
    // T[:, cb+v] = A_e[:, loc] U[:, v]
    for (int t = mb; t < me; ++t) {
        const int mis = ae2mis_J[t], k = mis_k[mis];
        if (k == 0) continue;
        const int r = mis2d_I[mis + 1] - mis2d_I[mis];
        const int *loc = pair_loc + pair_loc_off[ae_pair[t]];
        const double *U = mis_u + mis_u_off[mis];
        const int *cp = colpos + colpos_ptr[t];
        for (int idx = tid; idx < n * k; idx += ASM_NT) {
            const int i = idx % n, v = idx / n;
            double sum = 0.0;
            for (int q = 0; q < r; ++q) sum = fma(Am[(size_t)loc[q] * n + i], U[(size_t)v * r + q], sum);
            T[(size_t)cp[v] * n + i] = sum;
        }
    }
    __syncthreads();
    // E[(cb+v), :] = U[:, v]^T T[loc, :]
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

void coarse_elmats(hipStream_t s, const DevRelations &rel, int ae0, const EigBatch &batch,
                   const int *mis_k, const int64_t *mis_u_off, const double *mis_u,
                   const int *colpos_ptr, const int *colpos, const int64_t *out_off, double *out,
                   double *scratch, const int64_t *scratch_off) {
    if (!batch.count) return;
    profiler().begin(s);
    hipLaunchKernelGGL(coarse_elmat_kernel, dim3(batch.count), dim3(ASM_NT), 0, s, ae0, batch.n.p,
                       batch.moff.p, batch.W.p, rel.ae2mis_I.p, rel.ae2mis_J.p, rel.ae_pair.p,
                       rel.pair_loc_off.p, rel.pair_loc.p, rel.mis2d_I.p, mis_k, mis_u_off, mis_u,
                       colpos_ptr, colpos, out_off, out, scratch, scratch_off);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "coarse_elmats", 0.0, 0.0);
}

}  // namespace saamge_amd
