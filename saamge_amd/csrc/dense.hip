// Dense direct coarsest solver: the explicit inverse X = Ac^-1 (fp64, column-major), x = X b.
// Reference counterpart: the serial `--coarse-direct` path (UMFPACK through MFEM,
// amg/src/tg.cpp:979-1014 -> HypreDirect / UMFPackSolver) -- third-party there, hand-written
// here.  The coarsest operator has a few thousand rows and is applied once per V-cycle, so the
// solve must be ONE bandwidth-bound pass (a GEMV over 8 n^2 bytes: ~20 us at n = 3 300) and not a
// chain of triangular block steps (2.4 ms) or an inner Krylov loop (~170 tiny SpMV launches).
// The inverse is built in place by block Gauss-Jordan elimination without pivoting (Ac is SPD, so
// is every Schur complement), DNB = 64 columns per step, four launches per step:
//     gj_pivot_kernel     Pinv = (pivot block)^-1 in LDS, one workgroup
//     gj_rowpanel_kernel  Rp = Pinv M[k, :]                        (64 x n, separate buffer)
//     gj_update_kernel    M[i, j] -= M[i, k] Rp[:, j], i, j outside the pivot block
//                         (64 x 64 output tiles, operands through LDS, 4 x 4 register tiles)
//     gj_finish_kernel    M[k, j] = Rp,  M[i, k] = -M[i, k] Pinv,  M[k, k] = Pinv
// 2 n^3 flops, the matrix read + written once per step (n / 64 steps).  A non-positive pivot
// (semi-definite operator) is reported so that the caller falls back to the inner PCG.
#include "dense.h"

namespace saamge_amd {

constexpr int DNB = 64;

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

// inverse of the nb x nb pivot block at k0 (SPD) by Gauss-Jordan elimination in LDS; a short last
// block is padded with the identity.  Pinv: DNB x DNB column-major.
__global__ __launch_bounds__(256) void gj_pivot_kernel(int n, int k0, int nb, const double *__restrict__ M,
                                                       double *__restrict__ Pinv, int *__restrict__ info) {
    __shared__ double Dbuf[2][DNB][DNB + 1];      // ping-pong: one barrier per elimination step
    const int tid = threadIdx.x;
    const int r = tid & (DNB - 1), c0 = tid >> 6;          // this thread owns D[r][c0 + 4 u], u = 0..15
    for (int u = 0; u < DNB / 4; ++u) {
        const int c = c0 + 4 * u;
        Dbuf[0][r][c] = (r < nb && c < nb) ? M[(size_t)(k0 + c) * n + k0 + r] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    int cur = 0;
    for (int j = 0; j < nb; ++j, cur ^= 1) {
        double (*D)[DNB + 1] = Dbuf[cur], (*E)[DNB + 1] = Dbuf[cur ^ 1];
        const double d = D[j][j];
        if (!(d > 0.0)) {               // not positive definite (also catches NaN): uniform exit
            if (tid == 0) atomicMax(info, k0 + j + 1);
            return;
        }
        const double inv = 1.0 / d;
        const double rj = D[r][j];
#pragma unroll
        for (int u = 0; u < DNB / 4; ++u) {
            const int c = c0 + 4 * u;
            const double pjc = (c == j) ? 1.0 : D[j][c];
            double v;
            if (r == j) v = pjc * inv;
            else if (c == j) v = -rj * inv;
            else v = fma(-rj * inv, pjc, D[r][c]);
            E[r][c] = v;
        }
        __syncthreads();
    }
    for (int u = 0; u < DNB / 4; ++u) {
        const int c = c0 + 4 * u;
        Pinv[(size_t)c * DNB + r] = Dbuf[cur][r][c];
    }
}

// Rp[t, j] = sum_s Pinv[t, s] M[k0 + s, j] for every column j (the pivot block's own columns are
// skipped by the consumers); one thread per column j, the column segment in registers
__global__ __launch_bounds__(256) void gj_rowpanel_kernel(int n, int k0, int nb, const double *__restrict__ M,
                                                          const double *__restrict__ Pinv, double *__restrict__ Rp) {
    __shared__ double P[DNB][DNB + 1];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < DNB * DNB; idx += 256) P[idx & (DNB - 1)][idx >> 6] = Pinv[idx];
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j >= n) return;
    double x[DNB];
#pragma unroll
    for (int s = 0; s < DNB; ++s) x[s] = (s < nb) ? M[(size_t)j * n + k0 + min(s, nb - 1)] : 0.0;
#pragma unroll 4
    for (int t = 0; t < DNB; ++t) {
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < DNB; ++s) acc = fma(P[t][s], x[s], acc);
        Rp[(size_t)j * DNB + t] = acc;
    }
}

// M[i, j] -= sum_t M[i, k0 + t] Rp[t, j] on the 64 x 64 tile (blockIdx.x, blockIdx.y); tiles of the
// pivot block's rows / columns are left alone (k0 is a multiple of 64)
__global__ __launch_bounds__(256) void gj_update_kernel(int n, int k0, int nb, double *__restrict__ M,
                                                        const double *__restrict__ Rp) {
    const int ti = blockIdx.x, tj = blockIdx.y, kt = k0 / DNB;
    if (ti == kt || tj == kt) return;
    __shared__ double As[DNB][DNB + 1], Bs[DNB][DNB + 1];     // As[t][r] = M[i0 + r, k0 + t], Bs[t][c] = Rp[t, j0 + c]
    const int i0 = ti * DNB, j0 = tj * DNB, tid = threadIdx.x;
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int r = idx & (DNB - 1), t = idx >> 6;
        As[t][r] = (i0 + r < n && t < nb) ? M[(size_t)(k0 + t) * n + i0 + r] : 0.0;
        Bs[r][t] = (j0 + t < n) ? Rp[(size_t)(j0 + t) * DNB + r] : 0.0;     // (r plays t here: contiguous reads of Rp)
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 8
    for (int t = 0; t < DNB; ++t) {
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = As[t][tx + 16 * a];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = Bs[t][ty + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int gi = i0 + tx + 16 * a, gj = j0 + ty + 16 * b;
            if (gi < n && gj < n) M[(size_t)gj * n + gi] -= acc[a][b];
        }
}

// row panel <- Rp, column panel <- -M[:, k] Pinv, pivot block <- Pinv; one thread per index q:
// it owns column q of the row panel and row q of the column panel
__global__ __launch_bounds__(256) void gj_finish_kernel(int n, int k0, int nb, double *__restrict__ M,
                                                        const double *__restrict__ Pinv, const double *__restrict__ Rp) {
    __shared__ double P[DNB][DNB + 1];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < DNB * DNB; idx += 256) P[idx & (DNB - 1)][idx >> 6] = Pinv[idx];
    __syncthreads();
    const int q = blockIdx.x * 256 + tid;
    if (q >= n) return;
    if (q >= k0 && q < k0 + nb) {            // inside the pivot block: column q - k0 of Pinv
        for (int t = 0; t < nb; ++t) M[(size_t)q * n + k0 + t] = P[t][q - k0];
        return;
    }
    double x[DNB];
#pragma unroll
    for (int s = 0; s < DNB; ++s) x[s] = (s < nb) ? M[(size_t)(k0 + min(s, nb - 1)) * n + q] : 0.0;     // row q of the column panel
    for (int t = 0; t < nb; ++t) M[(size_t)q * n + k0 + t] = Rp[(size_t)q * DNB + t];                  // column q of the row panel
#pragma unroll 4
    for (int c = 0; c < DNB; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < DNB; ++s) acc = fma(x[s], P[s][c], acc);
        if (c < nb) M[(size_t)(k0 + c) * n + q] = -acc;
    }
}

// X <- (X + X^T) / 2 (the elimination is symmetric only up to round-off; the coarse solve of a
// symmetric V-cycle should be exactly symmetric)
__global__ __launch_bounds__(256) void dense_symmetrize_kernel(int n, double *__restrict__ M) {
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= n || j >= n || i <= j) return;
    const double v = 0.5 * (M[(size_t)j * n + i] + M[(size_t)i * n + j]);
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

bool dense_inverse_spd(hipStream_t s, const DCsr &A, DBuf<double> &X) {
    const int n = A.nrows;
    const size_t nn = (size_t)n * n;
    X.alloc(nn);
    DBuf<double> Pinv((size_t)DNB * DNB), Rp((size_t)DNB * n);
    DBuf<int> info(1);
    info.zero(s);
    profiler().begin(s);
    hipLaunchKernelGGL(dense_zero_kernel, dim3(div_up((long)nn, 256)), dim3(256), 0, s, nn, X.p);
    hipLaunchKernelGGL(dense_scatter_kernel, dim3(div_up((long)n * 8, 256)), dim3(256), 0, s, n, A.rowptr.p,
                       A.col.p, A.val.p, X.p);
    const int nt = div_up(n, DNB);
    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int nb = std::min(DNB, n - k0);
        hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(256), 0, s, n, k0, nb, X.p, Pinv.p, info.p);
        hipLaunchKernelGGL(gj_rowpanel_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, k0, nb, X.p, Pinv.p, Rp.p);
        if (nt > 1) hipLaunchKernelGGL(gj_update_kernel, dim3(nt, nt), dim3(256), 0, s, n, k0, nb, X.p, Rp.p);
        hipLaunchKernelGGL(gj_finish_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, k0, nb, X.p, Pinv.p, Rp.p);
    }
    hipLaunchKernelGGL(dense_symmetrize_kernel, dim3(div_up(n, 16), div_up(n, 16)), dim3(256), 0, s, n, X.p);
    SA_HIP_CHECK(hipGetLastError());
    const int bad = info.to_host(s)[0];       // (synchronises: Pinv / Rp may go)
    profiler().end(s, "coarse_inverse", 16.0 * (double)nn * nt, 2.0 * (double)n * n * n);
    return bad == 0;
}

void dense_symv(hipStream_t s, int n, const double *X, const double *b, double *y, bool add) {
    profiler().begin(s);
    hipLaunchKernelGGL(dense_symv_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, n, X, b, y, add ? 1 : 0);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "coarse_solve_dense", 8.0 * (double)n * n, 2.0 * (double)n * n);
}

}  // namespace saamge_amd
