// Dense direct coarsest solver: Ac = L L^T (fp64, lower, column-major), x = L^-T L^-1 b.
// Reference counterpart: the serial `--coarse-direct` path (UMFPACK through MFEM,
// amg/src/tg.cpp:979-1014 -> HypreDirect / UMFPackSolver) -- third-party there, hand-written
// here.  The coarsest operator is small (a few thousand rows), so the kernels are simple:
//   factor, left-looking, one block column (64) per step, three launches:
//     chol_update_kernel   A[k:, k] -= L[k:, 0:k] L[k, 0:k]^T         (all history, tiled)
//     chol_diag_kernel     the 64x64 diagonal block, one workgroup, in LDS
//     chol_panel_kernel    L21 = A21 L11^-T, one row per thread held in registers
//   solve, one launch per block column and direction:
//     tri_solve_kernel     every workgroup re-solves the diagonal block system in LDS and
//                          eliminates it from its own rows of the right-hand side
// A non-positive pivot (semi-definite operator) is reported so that the caller falls back to the
// inner PCG.
#include "dense.h"

namespace saamge_amd {

constexpr int DNB = 64;

__global__ __launch_bounds__(256) void dense_zero_kernel(size_t nn, double *__restrict__ L) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nn) L[i] = 0.0;
}
__global__ __launch_bounds__(256) void dense_scatter_kernel(int n, const int *__restrict__ rowptr,
                                                            const int *__restrict__ col,
                                                            const double *__restrict__ val, double *__restrict__ L) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(gt >> 3), lane = (int)(gt & 7);
    if (row >= n) return;
    for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += 8) {
        const int c = col[k];
        if (c <= row) L[(size_t)c * n + row] = val[k];   // lower triangle, column-major
    }
}

// A[i, k0 + j] -= sum_{t < k0} L[i, t] L[k0 + j, t]  for i >= k0, j < nb: 64 x 64 output tiles,
// history walked in chunks of 16 columns through LDS, 4 x 4 register tiles
__global__ __launch_bounds__(256) void chol_update_kernel(int n, int k0, int nb, double *__restrict__ L) {
    __shared__ double As[16][DNB + 1], Bs[16][DNB + 1];
    const int i0 = k0 + blockIdx.x * DNB;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int t0 = 0; t0 < k0; t0 += 16) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 16 * DNB; idx += 256) {
            const int r = idx & (DNB - 1), t = idx >> 6;
            const int gi = i0 + r, gj = k0 + r;
            As[t][r] = (gi < n && t0 + t < k0) ? L[(size_t)(t0 + t) * n + gi] : 0.0;
            Bs[t][r] = (r < nb && gj < n && t0 + t < k0) ? L[(size_t)(t0 + t) * n + gj] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
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
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gi = i0 + tx + 16 * a, j = ty + 16 * b;
            if (gi < n && j < nb && gi >= k0 + j) L[(size_t)(k0 + j) * n + gi] -= acc[a][b];
        }
}

// factor the nb x nb diagonal block at k0 in LDS (one workgroup)
__global__ __launch_bounds__(256) void chol_diag_kernel(int n, int k0, int nb, double *__restrict__ L,
                                                        int *__restrict__ info) {
    __shared__ double D[DNB][DNB + 1];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int r = idx & (DNB - 1), c = idx >> 6;
        D[r][c] = (r < nb && c < nb && r >= c) ? L[(size_t)(k0 + c) * n + k0 + r] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double d = D[j][j];
        if (!(d > 0.0)) {               // not positive definite (also catches NaN): uniform exit
            if (tid == 0) atomicMax(info, k0 + j + 1);
            return;
        }
        const double s = sqrt(d);
        __syncthreads();
        if (tid < nb && tid > j) D[tid][j] /= s;
        if (tid == j) D[j][j] = s;
        __syncthreads();
        // trailing update of the block: D[r][c] -= D[r][j] D[c][j], j < c <= r
        for (int idx = tid; idx < nb * nb; idx += 256) {
            const int r = idx % nb, c = idx / nb;
            if (c > j && r >= c) D[r][c] -= D[r][j] * D[c][j];
        }
        __syncthreads();
    }
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int r = idx % nb, c = idx / nb;
        if (r >= c) L[(size_t)(k0 + c) * n + k0 + r] = D[r][c];
    }
}

// rows i0 .. i0+255 of the panel below the (factored) diagonal block: L21 = A21 L11^-T, one row
// per thread, the whole row in registers
__global__ __launch_bounds__(256) void chol_panel_kernel(int n, int k0, int nb, double *__restrict__ L) {
    __shared__ double D[DNB][DNB + 1];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int r = idx & (DNB - 1), c = idx >> 6;
        // (a short last block is padded with the identity so that the substitution below can run
        // fully unrolled on registers)
        D[r][c] = (r < nb && c < nb && r >= c) ? L[(size_t)(k0 + c) * n + k0 + r] : ((r == c && r >= nb) ? 1.0 : 0.0);
    }
    __syncthreads();
    const int i = k0 + nb + blockIdx.x * 256 + tid;
    if (i >= n) return;
    double x[DNB];
#pragma unroll
    for (int c = 0; c < DNB; ++c) x[c] = (c < nb) ? L[(size_t)(k0 + min(c, nb - 1)) * n + i] : 0.0;
#pragma unroll
    for (int c = 0; c < DNB; ++c) {
        const double xc = x[c] / D[c][c];
        x[c] = xc;
#pragma unroll
        for (int t = c + 1; t < DNB; ++t) x[t] = fma(-xc, D[t][c], x[t]);
    }
#pragma unroll
    for (int c = 0; c < DNB; ++c)
        if (c < nb) L[(size_t)(k0 + c) * n + i] = x[c];
}

// One block column of a triangular solve, in place on b.
//   forward  (lower):  x_k = L_kk^-1 b_k ;  b_i -= L_ik x_k  for the rows below
//   backward (upper = L^T):  x_k = L_kk^-T b_k ;  b_i -= L_ki^T x_k  for the rows above
// `b` is the right-hand side being eliminated (rows outside the block are updated in place, the
// block's own rows are only read); the block of the solution goes to `out` (a different array:
// workgroups start at different times and all need the unmodified b_k).
__global__ __launch_bounds__(256) void tri_solve_kernel(int n, int k0, int nb, int backward,
                                                        const double *__restrict__ L, double *__restrict__ b,
                                                        double *__restrict__ out) {
    __shared__ double D[DNB][DNB + 1];
    __shared__ double xk[DNB];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int r = idx & (DNB - 1), c = idx >> 6;
        D[r][c] = (r < nb && c < nb && r >= c) ? L[(size_t)(k0 + c) * n + k0 + r] : 0.0;
    }
    if (tid < DNB) xk[tid] = (tid < nb) ? b[k0 + tid] : 0.0;
    __syncthreads();
    if (tid < 64) {   // one wavefront solves the diagonal block: lane r owns x_r
        double xr = xk[tid];
        if (!backward) {
            for (int c = 0; c < nb; ++c) {
                const double xc = __shfl(tid == c ? xr / D[c][c] : 0.0, c, 64);
                if (tid == c) xr = xc;
                else if (tid > c && tid < nb) xr = fma(-D[tid][c], xc, xr);
            }
        } else {
            for (int c = nb - 1; c >= 0; --c) {
                const double xc = __shfl(tid == c ? xr / D[c][c] : 0.0, c, 64);
                if (tid == c) xr = xc;
                else if (tid < c) xr = fma(-D[c][tid], xc, xr);
            }
        }
        xk[tid] = xr;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < nb) out[k0 + tid] = xk[tid];
    if (!backward) {
        const int i = k0 + nb + blockIdx.x * 256 + tid;
        if (i >= n) return;
        double s = 0.0;
        for (int c = 0; c < nb; ++c) s = fma(L[(size_t)(k0 + c) * n + i], xk[c], s);
        b[i] -= s;
    } else {
        const int i = blockIdx.x * 256 + tid;      // rows above: L[k0 + c, i], i < k0
        if (i >= k0) return;
        double s = 0.0;
        for (int c = 0; c < nb; ++c) s = fma(L[(size_t)i * n + k0 + c], xk[c], s);
        b[i] -= s;
    }
}

bool dense_cholesky_factor(hipStream_t s, const DCsr &A, DBuf<double> &L) {
    const int n = A.nrows;
    const size_t nn = (size_t)n * n;
    L.alloc(nn);
    profiler().begin(s);
    hipLaunchKernelGGL(dense_zero_kernel, dim3(div_up((long)nn, 256)), dim3(256), 0, s, nn, L.p);
    hipLaunchKernelGGL(dense_scatter_kernel, dim3(div_up((long)n * 8, 256)), dim3(256), 0, s, n, A.rowptr.p,
                       A.col.p, A.val.p, L.p);
    DBuf<int> info(1);
    info.zero(s);
    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int nb = std::min(DNB, n - k0);
        if (k0 > 0)
            hipLaunchKernelGGL(chol_update_kernel, dim3(div_up(n - k0, DNB)), dim3(256), 0, s, n, k0, nb, L.p);
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s, n, k0, nb, L.p, info.p);
        const int below = n - k0 - nb;
        if (below > 0)
            hipLaunchKernelGGL(chol_panel_kernel, dim3(div_up(below, 256)), dim3(256), 0, s, n, k0, nb, L.p);
    }
    SA_HIP_CHECK(hipGetLastError());
    const int bad = info.to_host(s)[0];
    profiler().end(s, "coarse_cholesky", 8.0 * (double)nn, (double)n * n * n / 3.0);
    return bad == 0;
}

void dense_cholesky_solve(hipStream_t s, int n, const double *L, const double *b, double *x, double *work) {
    // work: 2 n doubles.  forward: w0 = b -> z in w1 ; backward: w1 -> x
    double *w0 = work, *w1 = work + n;
    profiler().begin(s);
    SA_HIP_CHECK(hipMemcpyAsync(w0, b, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int nb = std::min(DNB, n - k0);
        const int below = n - k0 - nb;
        hipLaunchKernelGGL(tri_solve_kernel, dim3(std::max(1, div_up(below, 256))), dim3(256), 0, s, n, k0, nb, 0,
                           L, w0, w1);
    }
    for (int k0 = ((n - 1) / DNB) * DNB; k0 >= 0; k0 -= DNB) {
        const int nb = std::min(DNB, n - k0);
        hipLaunchKernelGGL(tri_solve_kernel, dim3(std::max(1, div_up(k0, 256))), dim3(256), 0, s, n, k0, nb, 1,
                           L, w1, x);
    }
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "coarse_solve_dense", 8.0 * (double)n * n, 2.0 * (double)n * n);
}

}  // namespace saamge_amd
