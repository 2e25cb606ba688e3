// Dense direct coarsest solver: the explicit inverse X = Ac^-1 (fp64, column-major), x = X b.
// Reference counterpart: the serial `--coarse-direct` path (UMFPACK through MFEM,
// amg/src/tg.cpp:979-1014 -> HypreDirect / UMFPackSolver) -- third-party there, hand-written
// here.  The coarsest operator has a few thousand rows and is applied once per V-cycle, so the
// solve must be ONE bandwidth-bound pass (a GEMV over 8 n^2 bytes: ~20 us at n = 3 300) and not a
// chain of triangular block steps (2.4 ms) or an inner Krylov loop (~170 tiny SpMV launches).
// The inverse is built in place by block Gauss-Jordan elimination without pivoting (Ac is SPD, so
// is every Schur complement), DNB = 64 columns per step, two launches per step (gj_panel_kernel, gj_apply_kernel
// below), on the LOWER block triangle only (the partially inverted matrix is symmetric up to the sign of its
// swept x unswept blocks): n^3 flops on the matrix cores, half the matrix read + written per step (n / 64 steps).  A non-positive pivot
// (semi-definite operator) is reported so that the caller falls back to the inner PCG.
#include "dense.h"

namespace saamge_amd {

constexpr int DNB = 64;
namespace gj {
constexpr int SB = 16;
#include "chol16.h"      // chol16_inverse_wave: Cholesky factor + inverse of a 16 x 16 block by one wavefront
}  // namespace gj

__global__ __launch_bounds__(256) void dense_zero_kernel(size_t nn, double *__restrict__ L) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nn) L[i] = 0.0;
}
__global__ __launch_bounds__(256) void dense_scatter_kernel(int n, const roff_t *__restrict__ rowptr,
                                                            const int *__restrict__ col,
                                                            const double *__restrict__ val, double *__restrict__ M) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(gt >> 3), lane = (int)(gt & 7);
    if (row >= n) return;
    for (roff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 8) M[(size_t)col[k] * n + row] = val[k];
}

// ---- step k of the block Gauss-Jordan elimination, two launches --------------------------------------
// gj_panel_kernel, one workgroup per 64-wide block b of the matrix:
//     Pinv = (pivot block)^-1 by Gauss-Jordan elimination in LDS (a short last block is padded with the identity);
//            EVERY workgroup repeats this 64-step chain -- it is latency, not work, and nothing has to wait for a
//            single workgroup's result;
//     Rp[b] = Pinv M[k, b]                (64 x 64, stored [t][c]: row t of the panel, column c of block b)
//     Col[b] = M[b, k]                    (64 x 64 copy of the OLD column panel, stored [t][r])
// gj_apply_kernel, one workgroup per 64 x 64 tile (bi, bj), reads only the copies, so the tiles are independent:
//     pivot block <- Pinv,   pivot row M[k, bj] <- Rp[bj],   pivot column M[bi, k] <- -Col[bi] Pinv,
//     elsewhere   M[bi, bj] -= Col[bi] Rp[bj]
// The 64 x 64 x 64 products run on the matrix cores (v_mfma_f64_16x16x4_f64: A operand row = lane & 15,
// k = lane >> 4; B operand col = lane & 15, k = lane >> 4; C/D register r: col = lane & 15, row = (lane >> 4) + 4 r).
// A wavefront owns a 32 x 32 quadrant and holds it TRANSPOSED, D(j, i) = sum_t X(t, j) C(i, t): the D columns are
// then consecutive rows i of the column-major matrix, so a load / store of register r is four 128-B segments.
typedef double gj_v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gj_panel_kernel(int n, int k0, int nb, const double *__restrict__ M,
                                                       double *__restrict__ Pinv, double *__restrict__ Rp,
                                                       double *__restrict__ Col, int *__restrict__ info) {
    __shared__ double Dbuf[2][DNB][DNB + 1];      // ping-pong: one barrier per elimination step
    __shared__ double Mk[DNB][DNB + 1];           // Mk[s][c] = M[k0 + s, j0 + c]
    const int b = blockIdx.x, tid = threadIdx.x, kb = k0 / DNB, j0 = b * DNB;
    const int r = tid & (DNB - 1), c0 = tid >> 6;          // this thread owns D[r][c0 + 4 u], u = 0..15
    // Only the LOWER block triangle of M is kept up to date (gj_apply_kernel).  With S the swept and U the unswept
    // blocks, the partially inverted matrix is symmetric inside S x S and U x U and M(S, U) = -M(U, S)^T, so the
    // blocks above the diagonal come from their mirror images: the row panel right of the pivot block (U x U) as
    // the transpose, the column panel above it (S x U) as the negated transpose -- read along the contiguous
    // direction of the mirror image.
    for (int u = 0; u < DNB / 4; ++u) {
        const int c = c0 + 4 * u;
        Dbuf[0][r][c] = (r < nb && c < nb) ? M[(size_t)(k0 + c) * n + k0 + r] : ((r == c) ? 1.0 : 0.0);
        if (b <= kb) Mk[r][c] = (r < nb && j0 + c < n) ? M[(size_t)(j0 + c) * n + k0 + r] : 0.0;
        else Mk[c][r] = (c < nb && j0 + r < n) ? M[(size_t)(k0 + c) * n + j0 + r] : 0.0;          // M[k0 + c, j0 + r] = M[j0 + r, k0 + c]
        if (b >= kb) Col[((size_t)b * DNB + c) * DNB + r] = (c < nb && j0 + r < n) ? M[(size_t)(k0 + c) * n + j0 + r] : 0.0;
        else Col[((size_t)b * DNB + r) * DNB + c] = (r < nb && j0 + c < n) ? -M[(size_t)(j0 + c) * n + k0 + r] : 0.0;   // M[j0 + c, k0 + r] = -M[k0 + r, j0 + c]
    }
    __syncthreads();
    // Pinv = (pivot block)^-1.  The block is symmetric positive definite (a Schur complement of an SPD operator; a short
    // last block is padded with the identity), so instead of a 64-step Gauss-Jordan chain with a barrier per step (115 us
    // of latency per block step, round 3) it is inverted through its Cholesky factor in 16 x 16 blocks:
    //   A = L L^T by four block steps -- diagonal block factor AND its inverse by one wavefront in registers
    //   (chol16_inverse_wave, the kernel of the banded eigen path), panel L_ij = A_ij L_jj^-T, trailing update --,
    //   W = L^-1 by block forward substitution, Pinv = W^T W.
    double (*Am)[DNB + 1] = Dbuf[0], (*W)[DNB + 1] = Dbuf[1];
    __shared__ double Ld[gj::SB][gj::SB + 1], Li[gj::SB][gj::SB + 1];
    __shared__ int sh_bad;
    if (tid == 0) sh_bad = 0;
    for (int u = 0; u < DNB / 4; ++u) W[r][c0 + 4 * u] = 0.0;
    for (int jb = 0; jb < DNB / 16; ++jb) {
        const int o = 16 * jb;
        { const int i = tid >> 4, j = tid & 15; Ld[i][j] = (j <= i) ? Am[o + i][o + j] : 0.0; }
        __syncthreads();
        if (tid < 64) {
            const int badb = gj::chol16_inverse_wave<false>(Ld, Li, tid);
            if (tid == 0 && badb) sh_bad = 1;
        }
        __syncthreads();
        if (sh_bad) {                   // not positive definite (uniform): every workgroup leaves
            if (tid == 0) atomicMax(info, k0 + o + 1);
            return;
        }
        { const int i = tid >> 4, j = tid & 15; W[o + i][o + j] = (j <= i) ? Li[i][j] : 0.0; }
        const int below = DNB - o - 16;        // rows under the diagonal block
        // panel: L(rr, o + cc) = sum_{q <= cc} A(rr, o + q) Linv_jj(cc, q); values first, then the writes (in place)
        double pv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + 256 * u, rr = o + 16 + (idx >> 4), cc = idx & 15;
            double t = 0.0;
            if ((idx >> 4) < below)
                for (int q = 0; q <= cc; ++q) t = fma(Am[rr][o + q], Li[cc][q], t);
            pv[u] = t;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + 256 * u, rr = o + 16 + (idx >> 4), cc = idx & 15;
            if ((idx >> 4) < below) Am[rr][o + cc] = pv[u];
        }
        __syncthreads();
        // trailing update of the lower triangle: A(i, k) -= sum_c L(i, o + c) L(k, o + c), o + 16 <= k <= i
        for (int idx = tid; idx < below * below; idx += 256) {
            const int i = o + 16 + idx / below, k = o + 16 + idx % below;
            if (k > i) continue;
            double t = 0.0;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) t = fma(Am[i][o + cc], Am[k][o + cc], t);
            Am[i][k] -= t;
        }
        __syncthreads();
    }
    // W = L^-1: W_ij = -W_ii sum_{k = j}^{i - 1} L_ik W_kj by block distance d = i - j (blocks of smaller distance are done)
    for (int d = 1; d < DNB / 16; ++d) {
        const int nblk = DNB / 16 - d;
        double sv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + 256 * u, bq = idx >> 8, e = idx & 255, ib = bq + d, jb2 = bq, rr = e >> 4, cc = e & 15;
            double t = 0.0;
            if (bq < nblk)
                for (int kb = jb2; kb < ib; ++kb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) t = fma(Am[16 * ib + rr][16 * kb + q], W[16 * kb + q][16 * jb2 + cc], t);
            sv[u] = t;
        }
        __syncthreads();
        // (the product with -W_ii needs the whole block of sums: through the still unused upper triangle of Am)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + 256 * u, bq = idx >> 8, e = idx & 255, ib = bq + d, jb2 = bq, rr = e >> 4, cc = e & 15;
            if (bq < nblk) Am[16 * jb2 + cc][16 * ib + rr] = sv[u];      // stored transposed above the diagonal
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int idx = tid + 256 * u, bq = idx >> 8, e = idx & 255, ib = bq + d, jb2 = bq, rr = e >> 4, cc = e & 15;
            if (bq < nblk) {
                double t = 0.0;
                for (int q = 0; q <= rr; ++q) t = fma(W[16 * ib + rr][16 * ib + q], Am[16 * jb2 + cc][16 * ib + q], t);
                W[16 * ib + rr][16 * jb2 + cc] = -t;
            }
        }
        __syncthreads();
    }
    // Pinv = W^T W (symmetric): Pinv(r, c) = sum_{t >= max(r, c)} W(t, r) W(t, c)
    for (int u = 0; u < DNB / 4; ++u) {
        const int c = c0 + 4 * u;
        double t = 0.0;
        for (int q = max(r, c); q < DNB; ++q) t = fma(W[q][r], W[q][c], t);
        Am[r][c] = t;          // (each thread writes entries nobody reads any more: the L blocks are done with)
    }
    __syncthreads();
    const int cur = 0;
    double (*P)[DNB + 1] = Dbuf[cur];
    if (b == kb)
        for (int u = 0; u < DNB / 4; ++u) {
            const int c = c0 + 4 * u;
            Pinv[(size_t)c * DNB + r] = P[r][c];
        }
    // Rp[b][t][c] = sum_s Pinv[t][s] Mk[s][c]: column c = lane, rows t = 16 wave .. 16 wave + 15 (P reads are broadcasts)
    {
        const int c = tid & 63, t0 = (tid >> 6) * 16;
        double acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0;
#pragma unroll 4
        for (int s = 0; s < DNB; ++s) {
            const double m = Mk[s][c];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = fma(P[t0 + u][s], m, acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) Rp[((size_t)b * DNB + t0 + u) * DNB + c] = acc[u];
    }
}

// One workgroup per 128 x 128 tile, one wavefront per 64 x 64 quadrant = block pair (bi, bj): the panel copies are
// fetched once for four block pairs (a 64 x 64 tile per workgroup re-read 128 KB of panels for 64 KB of matrix and
// was bound by the L2).  As[t][r] = Col[.][t][r] (two blocks side by side), Xs[t][c] = Rp[.][t][c], or Pinv[t][c] for
// the pivot block column.
constexpr int GJP2 = 2 * DNB + 4;
constexpr size_t GJ_APPLY_LDS = 2 * sizeof(double) * DNB * GJP2;
__global__ __launch_bounds__(256) void gj_apply_kernel(int n, int k0, int nb, int nt, double *__restrict__ M,
                                                       const double *__restrict__ Pinv, const double *__restrict__ Rp,
                                                       const double *__restrict__ Col) {
    extern __shared__ double gj_lds[];
    double (*As)[GJP2] = (double (*)[GJP2])gj_lds;
    double (*Xs)[GJP2] = (double (*)[GJP2])(gj_lds + (size_t)DNB * GJP2);
    const int kb = k0 / DNB, tid = threadIdx.x;
    if (blockIdx.y > blockIdx.x) return;        // (tiles above the diagonal are not kept: half the products, half the traffic)
    for (int h = 0; h < 2; ++h) {
        const int bi = 2 * blockIdx.x + h, bj = 2 * blockIdx.y + h;
        for (int idx = tid; idx < DNB * DNB; idx += 256) {
            const int lo = idx & (DNB - 1), hi = idx >> 6;
            As[hi][64 * h + lo] = (bi < nt) ? Col[(size_t)bi * DNB * DNB + idx] : 0.0;
            if (bj == kb) Xs[lo][64 * h + hi] = Pinv[idx];                       // Pinv is column-major: idx = c * 64 + t
            else Xs[hi][64 * h + lo] = (bj < nt) ? Rp[(size_t)bj * DNB * DNB + idx] : 0.0;
        }
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6, qi = w & 1, qj = w >> 1;
    const int bi = 2 * blockIdx.x + qi, bj = 2 * blockIdx.y + qj;
    if (bi >= nt || bj >= nt || bi < bj) return;
    const int i0 = bi * DNB, j0 = bj * DNB;
    if (bi == kb) {                         // pivot row <- Rp, pivot block <- Pinv: both sit in Xs
        if (lane < nb)
            for (int c = 0; c < DNB && j0 + c < n; ++c) {
                if (bj == kb && c >= nb) break;
                M[(size_t)(j0 + c) * n + k0 + lane] = Xs[lane][64 * qj + c];
            }
        return;
    }
    const bool pcol = bj == kb;             // pivot column <- -Col Pinv
    const int l15 = lane & 15, l4 = lane >> 4;
    // the accumulators start as the old tile and the X operand enters negated: D = M - Col X in one chain, every
    // load of the tile issued before the products and every store after them (a read-modify-write per element
    // after the loop is 64 dependent round trips per wavefront: the stores may alias the next load)
    gj_v4d acc[4][4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = i0 + 16 * ib + l15, gj = j0 + 16 * jb + l4 + 4 * reg;
                acc[jb][ib][reg] = (!pcol && gi < n && gj < n) ? M[(size_t)gj * n + gi] : 0.0;
            }
#pragma unroll 2
    for (int ks = 0; ks < DNB / 4; ++ks) {
        const int t = 4 * ks + l4;
        double a[4], bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = -Xs[t][64 * qj + 16 * q + l15];
            bb[q] = As[t][64 * qi + 16 * q + l15];
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
                acc[jb][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jb], bb[ib], acc[jb][ib], 0, 0, 0);
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int jl = 16 * jb + l4 + 4 * reg, gi = i0 + 16 * ib + l15, gj = j0 + jl;
                if (gi >= n || gj >= n || (pcol && jl >= nb)) continue;
                M[(size_t)gj * n + gi] = acc[jb][ib][reg];
            }
}

// X <- (X + X^T) / 2 (the elimination is symmetric only up to round-off; the coarse solve of a
// symmetric V-cycle should be exactly symmetric)
__global__ __launch_bounds__(256) void dense_symmetrize_kernel(int n, double *__restrict__ M) {
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= n || j >= n || i <= j) return;
    // (off the diagonal blocks only the lower triangle was kept: it is mirrored; inside a diagonal block both
    // triangles were, and are averaged)
    const double v = (i / DNB == j / DNB) ? 0.5 * (M[(size_t)j * n + i] + M[(size_t)i * n + j]) : M[(size_t)j * n + i];
    M[(size_t)j * n + i] = v;
    M[(size_t)i * n + j] = v;
}

// y = X b (+ y when `add`), X symmetric: one wavefront per output entry walks its column
__global__ __launch_bounds__(256) void dense_symv_kernel(int n, const double *__restrict__ X, const double *__restrict__ b,
                                                         double *__restrict__ y, int add) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const double *col = X + (size_t)row * n;
    double s0 = 0.0, s1 = 0.0;
    int i = lane;
    for (; i + 64 < n; i += 128) {
        s0 = fma(col[i], b[i], s0);
        s1 = fma(col[i + 64], b[i + 64], s1);
    }
    if (i < n) s0 = fma(col[i], b[i], s0);
    double s = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) y[row] = add ? y[row] + s : s;
}

void GjWork::reserve(int nmax, hipStream_t s) {
    const size_t nt = (size_t)div_up(nmax, DNB);
    if (Pinv.n < (size_t)DNB * DNB) Pinv.alloc((size_t)DNB * DNB);
    if (Rp.n < (size_t)DNB * DNB * nt) { Rp.alloc((size_t)DNB * DNB * nt); Col.alloc((size_t)DNB * DNB * nt); }
    if (!info.n) { info.alloc(1); info.zero(s); }
}

void dense_zero(hipStream_t s, size_t nn, double *X) {
    hipLaunchKernelGGL(dense_zero_kernel, dim3(div_up((long)nn, 256)), dim3(256), 0, s, nn, X);
}
void dense_scatter(hipStream_t s, int n, const roff_t *rowptr, const int *col, const double *val, double *X) {
    hipLaunchKernelGGL(dense_scatter_kernel, dim3(div_up((long)n * 8, 256)), dim3(256), 0, s, n, rowptr, col, val, X);
}

void dense_inverse_inplace(hipStream_t s, int n, double *X, GjWork &w) {
    w.reserve(n, s);
    const int nt = div_up(n, DNB), nt2 = div_up(nt, 2);
    static bool attr_set = false;
    if (!attr_set) {
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)gj_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GJ_APPLY_LDS));
        attr_set = true;
    }
    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int nb = std::min(DNB, n - k0);
        hipLaunchKernelGGL(gj_panel_kernel, dim3(nt), dim3(256), 0, s, n, k0, nb, X, w.Pinv.p, w.Rp.p, w.Col.p, w.info.p);
        hipLaunchKernelGGL(gj_apply_kernel, dim3(nt2, nt2), dim3(256), GJ_APPLY_LDS, s, n, k0, nb, nt, X, w.Pinv.p, w.Rp.p, w.Col.p);
    }
    hipLaunchKernelGGL(dense_symmetrize_kernel, dim3(div_up(n, 16), div_up(n, 16)), dim3(256), 0, s, n, X);
    SA_HIP_CHECK(hipGetLastError());
}

bool dense_inverse_spd(hipStream_t s, const DCsr &A, DBuf<double> &X) {
    const int n = A.nrows;
    const size_t nn = (size_t)n * n;
    X.alloc(nn);
    GjWork w;
    profiler().begin(s);
    dense_zero(s, nn, X.p);
    dense_scatter(s, n, A.rowptr.p, A.col.p, A.val.p, X.p);
    dense_inverse_inplace(s, n, X.p, w);
    const int bad = w.info.to_host(s)[0];       // (synchronises: the work buffers may go)
    profiler().end(s, "coarse_inverse", 16.0 * (double)nn * div_up(n, DNB), 2.0 * (double)n * n * n);
    return bad == 0;
}

void dense_symv(hipStream_t s, int n, const double *X, const double *b, double *y, bool add) {
    profiler().begin(s);
    hipLaunchKernelGGL(dense_symv_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, n, X, b, y, add ? 1 : 0);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "coarse_solve_dense", 8.0 * (double)n * n, 2.0 * (double)n * n);
}

}  // namespace saamge_amd
