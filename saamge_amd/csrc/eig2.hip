// Two-stage batched symmetric tridiagonalisation for gfx950.
//
//   stage 1  dense -> band (bandwidth SB):  per panel of SB columns, batched over all matrices
//            of the chunk (grid.y = matrix):
//              sbr_qr_*          Householder QR of the block below the band, compact WY (V, T)
//              sbr_symm_kernel   X = A22 V        (first panel only) tall-skinny product
//              sbr_z_kernel      Y = X T, S = T^T V^T Y, Z = Y - V S / 2
//              sbr_panel_update / sbr_fused_kernel   A22 -= Z V^T + V Z^T with the NEXT panel's
//                                product X' = A22' V' riding on the same pass (look-ahead)
//            HBM traffic 16 n'^2 B per panel  ->  16 n^3 / (3 SB) bytes per matrix instead of the
//            one-stage 8 n^3 / 3.
//   stage 2  band -> tridiagonal by bulge chasing, one workgroup per matrix, band resident in
//            LDS (n*2*SB doubles) when it fits; the 16 wavefronts run 16 sweeps at once in a
//            lock-step software pipeline (sweep s does its q-th chase step at time start[s] + q).
//   back-transformation  y = Q1 Q2 z : the chase reflectors (length <= SB) are applied per
//            sweep by 16-lane groups, the panel reflectors through their compact WY form.
//
// Same mathematics as LAPACK's dsytrd + dstebz + dstein + dormtr used by the reference through
// dsygvx (amg/src/xpacks.cpp:222-314); only the reduction is organised in two stages.
#include <cfloat>

#include "eig.h"

#include <cstdlib>

namespace saamge_amd {

constexpr int SB = EIG_SB;      // band width
constexpr int LDB = 2 * SB;     // band storage: distances 0 .. 2*SB-1 (band + bulge)

__device__ inline double unit_rand_ss(unsigned a, unsigned b) {  // deterministic uniform(-1,1)
    unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u + (a << 6) + (a >> 2));
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return ((double)h + 0.5) * (2.0 / 4294967296.0) - 1.0;
}

__device__ inline double gsum16(double v) {   // sum inside each 16-lane group
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// Reciprocal and square root to ~1 ulp from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-24)
// plus Newton steps: the IEEE-exact sequences cost 25-30 instructions each, and the reflector
// construction sits on the serial critical path of the panel QR and of every chase step.
// Arguments are O(1) entries of scaled matrices (no denormals, no overflow).
__device__ inline double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ inline double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}

__device__ inline double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}

__device__ inline double wsum64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 16-lane all-reduce with DPP row rotations (no LDS crossbar): {8,4,2,1} rotations cover the row
template <int CTRL>
__device__ inline double dpp_rot(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (row rotations read a valid lane everywhere: no "old" value, so no register to initialise)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double rsum16(double v) {
    v += dpp_rot<0x128>(v);  // row_ror:8
    v += dpp_rot<0x124>(v);  // row_ror:4
    v += dpp_rot<0x122>(v);  // row_ror:2
    v += dpp_rot<0x121>(v);  // row_ror:1
    return v;
}
// v + v(lane ^ 16) and v + v(lane ^ 32) with the gfx950 row / half swaps (VALU) instead of
// ds_bpermute round trips through the LDS crossbar
__device__ inline double xsum16(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ inline double xsum32(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
template <int NT>
__device__ inline double bsum(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += red[i];
    return r;
}

// XCD-aware work decode: workgroups are dealt round-robin over the 8 XCDs, so linear id
// 8 (g T + t) + x  ->  matrix 8 g + x, tile t : all tiles of one matrix run back to back on ONE
// XCD and share its L2 (panel V/Z, T factors).  grid = 8 * ceil(count / 8) * T.
__device__ inline void xcd_decode(int T, int &matrix, int &tile) {
    const int id = blockIdx.x;
    const int x = id & 7, rest = id >> 3;
    tile = rest % T;
    matrix = 8 * (rest / T) + x;
}

// V(r, c) of the current panel, read in place from A (unit diagonal implicit, zero above)
// Branch-free: the load is always in bounds (r < np, c < SB), so unrolled staging loops can
// keep many of them in flight.
__device__ inline double vmask(const double *A, int n, int k0, int r, int c) {
    const double v = A[(size_t)(k0 + c) * n + (k0 + SB + r)];
    return (r > c) ? v : ((r == c) ? 1.0 : 0.0);
}

// ---------------------------------------------------------------------------------------
// stage 1
// ---------------------------------------------------------------------------------------
template <int NT, bool LDSP>
__global__ __launch_bounds__(NT) void sbr_qr_kernel(int k0, const int *__restrict__ ns,
                                                    const int64_t *__restrict__ moff,
                                                    const int64_t *__restrict__ voff,
                                                    double *__restrict__ Wm,
                                                    double *__restrict__ Tfac,
                                                    double *__restrict__ Vpk) {
    extern __shared__ __align__(16) double plds[];
    __shared__ double red[NT / 64];
    __shared__ double zs[SB];
    __shared__ double Ts[SB * SB];
    const int b = blockIdx.x;
    const int n = ns[b];
    const int np = n - k0 - SB;  // rows below the band
    if (np < 2) return;
    double *A = Wm + moff[b];
    double *T = Tfac + voff[b] * SB + (size_t)(k0 / SB) * SB * SB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    // panel P(i, c) = A[k0+SB+i, k0+c], i < np, c < SB
    double *Pg = A + (size_t)k0 * n + (k0 + SB);
    double *P = LDSP ? plds : Pg;
    const int ldp = LDSP ? np : n;
    if (LDSP) {
        for (int idx = tid; idx < np * SB; idx += NT) {
            const int i = idx % np, c = idx / np;
            P[(size_t)c * ldp + i] = Pg[(size_t)c * n + i];
        }
    }
    for (int i = tid; i < SB * SB; i += NT) Ts[i] = 0.0;
    __syncthreads();
    const int nref = min(SB, np - 1);
    for (int c = 0; c < nref; ++c) {
        double *pc = P + (size_t)c * ldp;
        const double alpha = pc[c];
        double ss = 0.0;
        for (int i = c + 1 + tid; i < np; i += NT) ss = fma(pc[i], pc[i], ss);
        ss = bsum<NT>(ss, red);
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (ss != 0.0) {
            beta = -copysign(fast_sqrt(fma(alpha, alpha, ss)), alpha);
            tau = (beta - alpha) * fast_rcp(beta);
            scale = fast_rcp(alpha - beta);
        }
        for (int i = c + 1 + tid; i < np; i += NT) pc[i] *= scale;
        if (tid == 0) pc[c] = beta;
        __syncthreads();
        if (tau != 0.0) {
            // apply H_c to the remaining columns of the panel, one wavefront per column
            for (int j = c + 1 + wave; j < SB; j += NW) {
                double *pj = P + (size_t)j * ldp;
                double w = 0.0;
                for (int i = c + 1 + lane; i < np; i += 64) w = fma(pc[i], pj[i], w);
                w = wsum64(w) + pj[c];
                const double tw = tau * w;
                for (int i = c + 1 + lane; i < np; i += 64) pj[i] = fma(-tw, pc[i], pj[i]);
                if (lane == 0) pj[c] -= tw;
            }
            // z_j = V(:, j)^T v_c  for j < c
            for (int j = wave; j < c; j += NW) {
                const double *pj = P + (size_t)j * ldp;
                double z = 0.0;
                for (int i = c + 1 + lane; i < np; i += 64) z = fma(pj[i], pc[i], z);
                z = wsum64(z) + pj[c];  // row c of V(:, j) times v_c[c] = 1
                if (lane == 0) zs[j] = z;
            }
        }
        __syncthreads();
        if (tid <= c) {  // T(0:c, c) = -tau T(0:c,0:c) z ; T(c,c) = tau   (dlarft, forward columnwise)
            double t = 0.0;
            if (tid == c) {
                t = tau;
            } else if (tau != 0.0) {
                for (int j = tid; j < c; ++j) t = fma(Ts[j * SB + tid], zs[j], t);
                t = -tau * t;
            }
            Ts[c * SB + tid] = t;
        }
        __syncthreads();
    }
    for (int i = tid; i < SB * SB; i += NT) T[i] = Ts[i];
    if (LDSP) {
        for (int idx = tid; idx < np * SB; idx += NT) {
            const int i = idx % np, c = idx / np;
            Pg[(size_t)c * n + i] = P[(size_t)c * ldp + i];
        }
    }
    // row-major copy of V with the implicit unit diagonal / zeros made explicit: the product and
    // update kernels read whole rows of it through the scalar cache
    double *Vp = Vpk + voff[b] * SB;
    for (int idx = tid; idx < np * SB; idx += NT) {
        const int i = idx / SB, c = idx % SB;
        Vp[idx] = (i > c) ? P[(size_t)c * ldp + i] : ((i == c) ? 1.0 : 0.0);
    }
}

// Register-resident panel QR for np <= 512: thread t owns panel rows t and t + 256 (2 x SB
// doubles in VGPRs).  Per column c ONE reduction round gives everything the column needs: the 16
// raw products G_j = sum_{i > c} P(i, c) P(i, j) -- for j = c the squared norm below the diagonal,
// for j > c the dots of the updates, for j < c (those columns hold V) the z of the T factor --
// because the reflector is v = e_c + scale P(c+1:, c) with scale known only after the norm:
// v . P(:, j) = P(c, j) + scale G_j.  Two barriers per column (partials -> G), the T column is a
// 16 x 16 product spread over the workgroup instead of a serial chain on 16 threads.
__global__ __launch_bounds__(256) void sbr_qr_reg_kernel(int k0, const int *__restrict__ ns,
                                                         const int64_t *__restrict__ moff,
                                                         const int64_t *__restrict__ voff,
                                                         double *__restrict__ Wm,
                                                         double *__restrict__ Tfac,
                                                         double *__restrict__ Vpk) {
    __shared__ double part[SB * 128];     // [j][pair of lanes]: partial products
    __shared__ double Gs[2][SB];          // G_j, double-buffered by column parity
    __shared__ double rowc[2][SB];        // row c of the panel, same
    __shared__ double Ts[SB * SB];
    const int b = blockIdx.x;
    const int n = ns[b];
    const int np = n - k0 - SB;
    if (np < 2) return;
    double *A = Wm + moff[b];
    double *T = Tfac + voff[b] * SB + (size_t)(k0 / SB) * SB * SB;
    const int tid = threadIdx.x;
    double *Pg = A + (size_t)k0 * n + (k0 + SB);
    const int r0 = tid, r1 = tid + 256;
    double p0[SB], p1[SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        p0[c] = (r0 < np) ? Pg[(size_t)c * n + r0] : 0.0;
        p1[c] = (r1 < np) ? Pg[(size_t)c * n + r1] : 0.0;
    }
    for (int i = tid; i < SB * SB; i += 256) Ts[i] = 0.0;
    const int nref = min(SB, np - 1);
    const int gj = tid >> 4, gk = tid & 15;      // reduction role: column gj, sixteenth gk
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        if (c < nref) {   // uniform: the loop stays fully unrolled, the panel stays in registers
        const int pb = c & 1;
        // ---- A: raw products of column c with every column over the rows below c ----
        const double a0 = (r0 > c) ? p0[c] : 0.0;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            double sj = fma(p1[c], p1[j], a0 * p0[j]);
            sj += dpp_rot<0xB1>(sj);                 // quad_perm [1,0,3,2]: neighbouring lane
            if (!(tid & 1)) part[j * 128 + (tid >> 1)] = sj;
        }
        if (r0 == c) {
#pragma unroll
            for (int j = 0; j < SB; ++j) rowc[pb][j] = p0[j];
        }
        __syncthreads();
        // ---- B: G_j ----
        {
            const double *pp = part + gj * 128 + gk * 8;
            double g = ((pp[0] + pp[1]) + (pp[2] + pp[3])) + ((pp[4] + pp[5]) + (pp[6] + pp[7]));
            g = rsum16(g);
            if (gk == 0) Gs[pb][gj] = g;
        }
        __syncthreads();
        // ---- C: reflector, update of the columns to the right, column c of T ----
        const double ss = Gs[pb][c];
        const double alpha = rowc[pb][c];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (ss != 0.0) {
            beta = -copysign(fast_sqrt(fma(alpha, alpha, ss)), alpha);
            tau = (beta - alpha) * fast_rcp(beta);
            scale = fast_rcp(alpha - beta);
        }
        // v entries of my rows (v_c = 1 on row c, 0 above)
        const double v0 = (r0 > c) ? p0[c] * scale : ((r0 == c) ? 1.0 : 0.0);
        const double v1 = p1[c] * scale;
        if (r0 > c) p0[c] = v0;
        if (r0 == c) p0[c] = beta;
        p1[c] = v1;
        if (tau != 0.0) {
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                if (j <= c) continue;
                const double tw = tau * fma(scale, Gs[pb][j], rowc[pb][j]);   // tau (v . P(:, j))
                p0[j] = fma(-tw, v0, p0[j]);
                p1[j] = fma(-tw, v1, p1[j]);
            }
        }
        // T(0:c, c) = -tau T(0:c, 0:c) z, T(c, c) = tau (dlarft, forward columnwise); z_j = V(:, j) . v
        {
            const int i = gj, j = gk;
            double t = 0.0;
            if (tau != 0.0 && j >= i && j < c) t = Ts[j * SB + i] * fma(scale, Gs[pb][j], rowc[pb][j]);
            t = rsum16(t);
            if (j == 0 && i <= c) Ts[c * SB + i] = (i == c) ? tau : -tau * t;
        }
        }
    }
    __syncthreads();
    for (int i = tid; i < SB * SB; i += 256) T[i] = Ts[i];
    double *Vp = Vpk + voff[b] * SB;
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        if (r0 < np) {
            Pg[(size_t)c * n + r0] = p0[c];
            Vp[(size_t)r0 * SB + c] = (r0 > c) ? p0[c] : ((r0 == c) ? 1.0 : 0.0);
        }
        if (r1 < np) {
            Pg[(size_t)c * n + r1] = p1[c];
            Vp[(size_t)r1 * SB + c] = (r1 > c) ? p1[c] : ((r1 == c) ? 1.0 : 0.0);
        }
    }
}

// X(r, :) = sum_c A22(r, c) V(c, :).  One workgroup per (64-row block, matrix); lane = row, the
// four wavefronts split the columns.  The column index is wave-uniform, so V(c, 0..15) comes in
// through the scalar cache (s_load from the packed row-major V) and every FMA takes it as an
// SGPR operand: no LDS traffic in the main loop, A22 streamed once with coalesced 512-B loads.
constexpr int SY_NT = 256;

__global__ __launch_bounds__(SY_NT) void sbr_symm_kernel(int k0, const int *__restrict__ ns,
                                                         const int64_t *__restrict__ moff,
                                                         const int64_t *__restrict__ voff,
                                                         const double *__restrict__ Wm,
                                                         const double *__restrict__ Vpk,
                                                         double *__restrict__ Xbuf,
                                                         const int64_t *__restrict__ goff,
                                                         double *__restrict__ Gbuf, int count,
                                                         int tiles) {
    constexpr int SBP = SB + 1;
    __shared__ double red[4 * SB * 64];              // K-split reduction, [g][j][r]
    __shared__ double xs[64 * SBP], vs2[64 * SBP];   // padded [rr][j]: conflict-free reads below
    int b, blk;
    xcd_decode(tiles, b, blk);
    if (b >= count) return;
    const int n = ns[b];
    const int np = n - k0 - SB;
    if (np < 2) return;
    const int r0 = blk * 64;
    if (r0 >= np) return;
    const double *A = Wm + moff[b];
    const double *A22 = A + (size_t)(k0 + SB) * n + (k0 + SB);
    const double *__restrict__ Vp = Vpk + voff[b] * SB;
    double *X = Xbuf + voff[b] * SB;
    const int tid = threadIdx.x;
    const int r = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowc = min(r0 + r, np - 1);            // clamped: every load stays in bounds
    double acc[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) acc[j] = 0.0;
    {
        const int q = (np + 3) >> 2;
        const int cb = g * q, ce = min(cb + q, np);
        const double *Ar = A22 + rowc;
        int cc = cb;
        for (; cc + 4 <= ce; cc += 4) {   // four 512-B loads in flight per wavefront
            const double a0 = Ar[(size_t)cc * n], a1 = Ar[(size_t)(cc + 1) * n];
            const double a2 = Ar[(size_t)(cc + 2) * n], a3 = Ar[(size_t)(cc + 3) * n];
            const double *vr = Vp + (size_t)cc * SB;   // wave-uniform
#pragma unroll
            for (int j = 0; j < SB; ++j)
                acc[j] = fma(a3, vr[3 * SB + j], fma(a2, vr[2 * SB + j], fma(a1, vr[SB + j], fma(a0, vr[j], acc[j]))));
        }
        for (; cc < ce; ++cc) {
            const double a = Ar[(size_t)cc * n];
            const double *vr = Vp + (size_t)cc * SB;
#pragma unroll
            for (int j = 0; j < SB; ++j) acc[j] = fma(a, vr[j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) red[(g * SB + j) * 64 + r] = acc[j];
    __syncthreads();
    for (int idx = tid; idx < 64 * SB; idx += SY_NT) {
        const int rr = idx & 63, j = idx >> 6;
        double s = 0.0;
        if (r0 + rr < np) {
            s = (red[(0 * SB + j) * 64 + rr] + red[(1 * SB + j) * 64 + rr]) +
                (red[(2 * SB + j) * 64 + rr] + red[(3 * SB + j) * 64 + rr]);
            X[(size_t)j * n + r0 + rr] = s;
        }
        xs[rr * SBP + j] = s;
        vs2[rr * SBP + j] = (r0 + rr < np) ? Vp[(size_t)(r0 + rr) * SB + j] : 0.0;
    }
    __syncthreads();
    {   // partial G(a, c) = sum_rr V(r0+rr, a) X(r0+rr, c), reduced over row blocks by sbr_z_kernel
        const int a = tid >> 4, c = tid & 15;
        double s = 0.0;
        for (int rr = 0; rr < 64; ++rr) s = fma(vs2[rr * SBP + a], xs[rr * SBP + c], s);
        Gbuf[goff[b] + (size_t)blk * (SB * SB) + tid] = s;
    }
}

// Z = X T - V S / 2 with S = T^T (G T), G = V^T X summed over the row-block partials.
// Z is written row-major (Z(r, c) at r * SB + c) for the scalar-operand reads of the update.
constexpr int SM_NT = 256;
__global__ __launch_bounds__(SM_NT) void sbr_z_kernel(int k0, const int *__restrict__ ns,
                                                      const int64_t *__restrict__ voff,
                                                      const double *__restrict__ Vpk,
                                                      const double *__restrict__ Tfac,
                                                      const double *__restrict__ Xbuf,
                                                      const int64_t *__restrict__ goff,
                                                      const double *__restrict__ Gbuf,
                                                      double *__restrict__ Zbuf, int count, int tiles) {
    __shared__ double Ts[SB * SB], Gs[SB * SB], GT[SB * SB], Ss[SB * SB];
    int b, blk;
    xcd_decode(tiles, b, blk);
    if (b >= count) return;
    const int n = ns[b];
    const int np = n - k0 - SB;
    if (np < 2) return;
    const int r0 = blk * SM_NT;
    if (r0 >= np) return;
    const double *T = Tfac + voff[b] * SB + (size_t)(k0 / SB) * SB * SB;
    const double *X = Xbuf + voff[b] * SB;
    const double *Vp = Vpk + voff[b] * SB;
    double *Z = Zbuf + voff[b] * SB;
    const int tid = threadIdx.x;
    {
        Ts[tid] = T[tid];
        const int nblk = (np + 63) / 64;
        const double *Gp = Gbuf + goff[b] + tid;
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += Gp[(size_t)k * (SB * SB)];
        Gs[tid] = s;   // G(a, c) at a * SB + c
    }
    __syncthreads();
    {   // GT(a, c) = sum_{j <= c} G(a, j) T(j, c)      (T column-major: T(j, c) = Ts[c * SB + j])
        const int a = tid >> 4, c = tid & 15;
        double s = 0.0;
        for (int j = 0; j <= c; ++j) s = fma(Gs[a * SB + j], Ts[c * SB + j], s);
        GT[a * SB + c] = s;
    }
    __syncthreads();
    {   // S(a, c) = sum_{j <= a} T(j, a) GT(j, c)
        const int a = tid >> 4, c = tid & 15;
        double s = 0.0;
        for (int j = 0; j <= a; ++j) s = fma(Ts[a * SB + j], GT[j * SB + c], s);
        Ss[a * SB + c] = s;
    }
    __syncthreads();
    // V rows in / Z rows out are row-major (16 contiguous doubles per row): staged through LDS so that
    // the global accesses of the workgroup are one contiguous block instead of 128-B strided per lane
    __shared__ double stage[SM_NT * (SB + 1)];
    const int nr = min(SM_NT, np - r0);
    for (int idx = tid; idx < nr * SB; idx += SM_NT) stage[(idx >> 4) * (SB + 1) + (idx & 15)] = Vp[(size_t)r0 * SB + idx];
    __syncthreads();
    const int r = r0 + tid;
    double x[SB], v[SB], z[SB];
    if (r < np) {
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            x[j] = X[(size_t)j * n + r];
            v[j] = stage[tid * (SB + 1) + j];
        }
#pragma unroll
        for (int c = 0; c < SB; ++c) {
            double y = 0.0, s = 0.0;
#pragma unroll
            for (int j = 0; j <= c; ++j) y = fma(x[j], Ts[c * SB + j], y);
#pragma unroll
            for (int a = 0; a < SB; ++a) s = fma(v[a], Ss[a * SB + c], s);
            z[c] = fma(-0.5, s, y);
        }
    }
    __syncthreads();
    if (r < np) {
#pragma unroll
        for (int c = 0; c < SB; ++c) stage[tid * (SB + 1) + c] = z[c];
    }
    __syncthreads();
    for (int idx = tid; idx < nr * SB; idx += SM_NT) Z[(size_t)r0 * SB + idx] = stage[(idx >> 4) * (SB + 1) + (idx & 15)];
}


constexpr int S2_NT = 256;
constexpr int S2_KC = 8;   // columns per wavefront per step (bounded by the SGPR budget)

// ---- look-ahead variant: the product of the NEXT panel is fused into the update -------------
// Per panel p (V_p, T_p, Z_p known):
//   sbr_panel_update_kernel   A22(:, 0:SB) -= Z V^T + V Z^T      (the next panel's columns only)
//   sbr_qr_*                  QR of the next panel -> V_{p+1}, T_{p+1}
//   sbr_fused_kernel          A22' -= Z V^T + V Z^T on the rest (A22' = A22(SB:, SB:)) and, in the
//                             same pass over A22', X_{p+1} = A22' V_{p+1} (+ the partial V^T X)
//   sbr_z_kernel              Z_{p+1}
// The trailing matrix is read and written ONCE per panel (16 n'^2 bytes instead of 24 n'^2).

// rows [r0, r0 + 256) of the first SB columns of A22
__global__ __launch_bounds__(256) void sbr_panel_update_kernel(int k0, const int *__restrict__ ns,
                                                               const int64_t *__restrict__ moff,
                                                               const int64_t *__restrict__ voff,
                                                               double *__restrict__ Wm,
                                                               const double *__restrict__ Vpk,
                                                               const double *__restrict__ Zbuf, int count,
                                                               int tiles, int min_np = 2,
                                                               const int *__restrict__ bws = nullptr,
                                                               const int *__restrict__ skip = nullptr) {
    int b, blk;
    xcd_decode(tiles, b, blk);
    if (b >= count) return;
    if (skip && skip[b]) return;
    const int n = ns[b];
    const int np = n - k0 - SB;
    if (np < min_np) return;       // (band reduction: a trailing matrix of order 1 has no reflector; Cholesky: 1)
    const int r = blk * 256 + threadIdx.x;
    if (r >= (bws ? min(np, bws[b]) : np)) return;      // (banded Cholesky: the panel is zero below the band)
    double *A22 = Wm + moff[b] + (size_t)(k0 + SB) * n + (k0 + SB);
    const double *__restrict__ Z = Zbuf + voff[b] * SB;
    const double *__restrict__ Vp = Vpk + voff[b] * SB;
    double zr[SB], vr[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        zr[k] = Z[(size_t)r * SB + k];
        vr[k] = Vp[(size_t)r * SB + k];
    }
    const int nc = min(SB, np);
    for (int c = 0; c < nc; ++c) {
        const double *zc = Z + (size_t)c * SB;   // uniform
        const double *vc = Vp + (size_t)c * SB;
        double t = A22[(size_t)c * n + r];
#pragma unroll
        for (int k = 0; k < SB; ++k) t = fma(-zr[k], vc[k], fma(-vr[k], zc[k], t));
        A22[(size_t)c * n + r] = t;
    }
}


constexpr int SF_ROWS = 64;   // rows per lane-row: 2 x 16 row operands + 16 accumulators in VGPRs
// PROD = false: the update alone (no accumulators: fewer VGPRs, one more wavefront per SIMD).
// RPL = rows per lane (strip of 64 RPL rows): every column operand fetched through the scalar cache
// feeds RPL FMAs.  The kernel is bound by that operand delivery (~17 TFLOP/s with one FMA per
// fetched scalar, both on the 405-row and on the 2 600-row agglomerates), not by HBM.
// TERMS = 1: A22' -= V V^T only (the trailing update of the Cholesky factorisation of the
// few-eigenpairs path: half the FMAs and scalar operands, Z is not read).  TERMS = 3: A22' -= V V^T +
// Z Z^T with V = the (row-shifted) factor panel k and Z = panel k + 1 (not shifted): two panels in one
// pass over the trailing matrix.  Both only on the tiles on and below the diagonal.
template <bool PROD, int RPL, int TERMS = 2>
__global__ __launch_bounds__(S2_NT) void sbr_fused_kernel(int k0, const int *__restrict__ ns,
                                                          const int64_t *__restrict__ moff,
                                                          const int64_t *__restrict__ voff,
                                                          double *__restrict__ Wm,
                                                          const double *__restrict__ Vcur,
                                                          const double *__restrict__ Zbuf,
                                                          const double *__restrict__ Vnext,
                                                          double *__restrict__ Xbuf,
                                                          const int64_t *__restrict__ goff,
                                                          double *__restrict__ Gbuf,
                                                          double *__restrict__ trashbuf, int count,
                                                          int tiles, int shift,
                                                          const int *__restrict__ bws = nullptr,
                                                          const double *__restrict__ VrowCur = nullptr,
                                                          const double *__restrict__ ZrowBuf = nullptr,
                                                          const int *__restrict__ skip = nullptr) {
    // (VrowCur / ZrowBuf, TERMS = 3 only: the ROW operands come from these panels instead of Vcur / Zbuf --
    // the signed factorisation C - theta I = L S L^T updates with (L S) L^T)
    constexpr int SBP = SB + 1;
    constexpr int KC = (RPL == 1) ? S2_KC : 4;   // columns per step (VGPR budget: 2 RPL KC tile values)
    __shared__ double red[4 * SB * SF_ROWS];   // 32 KiB: K-split reduction
    __shared__ double xs[SF_ROWS * SBP], vs2[SF_ROWS * SBP];
    int b, blk;
    xcd_decode(tiles, b, blk);
    if (b >= count) return;
    if (skip && skip[b]) return;
    const int n = ns[b];
    if (n - k0 - SB < 2) return;               // this matrix has no panel k0
    // shift = SB: look-ahead pipeline, the first SB columns / rows were handled by
    // sbr_panel_update_kernel; shift = 0: the whole trailing matrix of panel k0
    int np = n - k0 - SB - shift;              // order of A22' (may be < 2: update only)
    if (TERMS != 2 && bws) np = min(np, bws[b]);   // banded Cholesky: the panels are zero below the band
    if (np < 1) return;
    const int i0 = blk * (SF_ROWS * RPL);
    if (i0 >= np) return;
    // TERMS == 1 (Cholesky): only the columns up to the end of this strip's diagonal tile -- the
    // factorisation never reads the upper triangle of the trailing matrix
    const int ncols = (TERMS != 2) ? min(np, i0 + SF_ROWS * RPL) : np;
    const bool prod = PROD && np >= 2;         // the next panel has reflectors
    double *A22 = Wm + moff[b] + (size_t)(k0 + SB + shift) * n + (k0 + SB + shift);
    const double *__restrict__ Z = Zbuf + voff[b] * SB + (TERMS == 3 ? 0 : shift * SB);   // rows shifted
    const double *__restrict__ Vc = Vcur + voff[b] * SB + shift * SB;
    const double *__restrict__ Vn = Vnext + voff[b] * SB;
    const double *__restrict__ Zr = (TERMS == 3 && ZrowBuf) ? ZrowBuf + voff[b] * SB : Z;
    const double *__restrict__ Vr = (TERMS == 3 && VrowCur) ? VrowCur + voff[b] * SB + shift * SB : Vc;
    double *X = Xbuf + voff[b] * SB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *pa[RPL];
    size_t sa[RPL];
    double za[RPL][SB], va[RPL][SB];   // negated row operands
    double xa[RPL][SB];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int ia = i0 + SF_ROWS * r + lane;
        const bool oka = ia < np;
        // rows past the end of the matrix: the lane keeps computing (a predicated store would make
        // the compiler sink the arithmetic into the branch and spill) but reads and writes a
        // private trash slot with stride 0
        pa[r] = oka ? A22 + ia : trashbuf + lane;
        sa[r] = oka ? (size_t)n : 0;
        const int iac = min(ia, np - 1);
#pragma unroll
        for (int c = 0; c < SB; ++c) {
            za[r][c] = (TERMS != 1) ? -Zr[(size_t)iac * SB + c] : 0.0;
            va[r][c] = -Vr[(size_t)iac * SB + c];
            xa[r][c] = 0.0;
        }
    }
    // Column groups of KC; the A-tile of the NEXT group is requested before the current one is
    // worked on (double-buffered in registers), so its HBM round trip is covered by KC x 48 RPL FMAs.
    int l0 = KC * w;
    double ta[RPL][KC], tn[RPL][KC];
    if (l0 + KC <= ncols) {
#pragma unroll
        for (int r = 0; r < RPL; ++r)
#pragma unroll
            for (int k = 0; k < KC; ++k) ta[r][k] = pa[r][(size_t)(l0 + k) * sa[r]];
    }
    for (; l0 + KC <= ncols; l0 += 4 * KC) {
        const int l1 = l0 + 4 * KC;
        const bool more = l1 + KC <= ncols;    // wave-uniform
        if (more) {
#pragma unroll
            for (int r = 0; r < RPL; ++r)
#pragma unroll
                for (int k = 0; k < KC; ++k) tn[r][k] = pa[r][(size_t)(l1 + k) * sa[r]];
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const double *zl = Z + (size_t)(l0 + k) * SB;    // wave-uniform: scalar loads
            const double *vl = Vc + (size_t)(l0 + k) * SB;
#pragma unroll
            for (int c = 0; c < SB; ++c)
#pragma unroll
                for (int r = 0; r < RPL; ++r)
                    ta[r][k] = (TERMS == 2) ? fma(za[r][c], vl[c], fma(va[r][c], zl[c], ta[r][k]))
                             : (TERMS == 3) ? fma(za[r][c], zl[c], fma(va[r][c], vl[c], ta[r][k]))
                                            : fma(va[r][c], vl[c], ta[r][k]);
            if (prod) {
                const double *vn = Vn + (size_t)(l0 + k) * SB;
#pragma unroll
                for (int j = 0; j < SB; ++j)
#pragma unroll
                    for (int r = 0; r < RPL; ++r) xa[r][j] = fma(ta[r][k], vn[j], xa[r][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < RPL; ++r)
#pragma unroll
            for (int k = 0; k < KC; ++k) pa[r][(size_t)(l0 + k) * sa[r]] = ta[r][k];
        if (more) {
#pragma unroll
            for (int r = 0; r < RPL; ++r)
#pragma unroll
                for (int k = 0; k < KC; ++k) ta[r][k] = tn[r][k];
        }
    }
    for (; l0 < ncols; ++l0) {                // the last, partial group
        double t0[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) t0[r] = pa[r][(size_t)l0 * sa[r]];
        const double *zl = Z + (size_t)l0 * SB;
        const double *vl = Vc + (size_t)l0 * SB;
#pragma unroll
        for (int c = 0; c < SB; ++c)
#pragma unroll
            for (int r = 0; r < RPL; ++r)
                t0[r] = (TERMS == 2) ? fma(za[r][c], vl[c], fma(va[r][c], zl[c], t0[r]))
                      : (TERMS == 3) ? fma(za[r][c], zl[c], fma(va[r][c], vl[c], t0[r]))
                                     : fma(va[r][c], vl[c], t0[r]);
        if (prod) {
            const double *vn = Vn + (size_t)l0 * SB;
#pragma unroll
            for (int j = 0; j < SB; ++j)
#pragma unroll
                for (int r = 0; r < RPL; ++r) xa[r][j] = fma(t0[r], vn[j], xa[r][j]);
        }
#pragma unroll
        for (int r = 0; r < RPL; ++r) pa[r][(size_t)l0 * sa[r]] = t0[r];
    }
    if (!prod) return;   // block-uniform
    // ---- X = sum of the four column splits; partial G = V^T X of each 64-row block ----
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int ib = i0 + SF_ROWS * r;           // first row of this 64-row block
        if (ib >= np) break;                       // block-uniform
        if (r) __syncthreads();
#pragma unroll
        for (int j = 0; j < SB; ++j) red[(w * SB + j) * SF_ROWS + lane] = xa[r][j];
        __syncthreads();
        for (int idx = tid; idx < SF_ROWS * SB; idx += S2_NT) {
            const int rr = idx & (SF_ROWS - 1), j = idx >> 6;
            double s = 0.0;
            if (ib + rr < np) {
                s = (red[(0 * SB + j) * SF_ROWS + rr] + red[(1 * SB + j) * SF_ROWS + rr]) +
                    (red[(2 * SB + j) * SF_ROWS + rr] + red[(3 * SB + j) * SF_ROWS + rr]);
                X[(size_t)j * n + ib + rr] = s;
            }
            xs[rr * SBP + j] = s;
            vs2[rr * SBP + j] = (ib + rr < np) ? Vn[(size_t)(ib + rr) * SB + j] : 0.0;
        }
        __syncthreads();
        {
            const int a = tid >> 4, c = tid & 15;
            double s = 0.0;
            for (int rr = 0; rr < SF_ROWS; ++rr) s = fma(vs2[rr * SBP + a], xs[rr * SBP + c], s);
            Gbuf[goff[b] + (size_t)(blk * RPL + r) * (SB * SB) + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------
// stage 2: bulge chasing
// ---------------------------------------------------------------------------------------

// number of chase steps of sweep s
__device__ __host__ inline int chase_steps(int n, int s) { return (n - 1 - s + SB - 1) / SB; }

struct BandRef {
    double *p;
    // (int index: j * LDB + i - j < 2^31 for any admissible n)
    __device__ inline double &operator()(int i, int j) const { return p[j * (LDB - 1) + i]; }
    __device__ inline double sym(int i, int j) const { return i >= j ? p[j * (LDB - 1) + i] : p[i * (LDB - 1) + j]; }
};

__device__ inline void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

constexpr int HAND = 4 * SB + 2;  // per-wave LDS scratch: vprev[SB], tau, pad, sv[SB], sw[SB], sx[SB]

// One chase step of sweep s (q-th block), executed by 16 NCQ lanes (a "slot": NCQ = 4: a whole
// wavefront; NCQ = 2: half of one, the two halves work on different sweeps).  Lane l of the slot:
// r = l & 15 is the row inside the SB x SB block, cq = l >> 4 selects SB / NCQ of its 16 columns.
// Vectors that every lane needs (v, w, first column) go through a slot-private LDS scratch
// instead of cross-lane permutes; the 16-lane sums use DPP row rotations.  s, q (and everything
// derived from them) are per-slot values; a slot is active or idle as a whole, so the DPP rows and
// the permlane swaps below never mix active and idle lanes.
template <int NCQ>
__device__ inline void chase_step(const BandRef &B, int n, int s, int q, double *hand,
                                  double *refl_v, double *refl_tau, int lane) {
    constexpr int CPL = SB / NCQ;               // columns per lane
    const int sl = lane & (16 * NCQ - 1);       // lane inside the slot
    const int r = sl & 15, cq = sl >> 4;
    auto rowsum = [](double v) {                // sum over the NCQ column groups of a row
        if (NCQ == 4) return xsum32(xsum16(v));
        if (NCQ == 2) return xsum16(v);
        return v;
    };
    double *vprev = hand, *sv = hand + SB + 2, *sw = sv + SB, *sx = sw + SB;
    const int i0 = s + 1 + q * SB;              // first row of I_q
    const int L = min(SB, n - i0);              // |I_q| >= 1
    double vr = 0.0, tau = 0.0;                 // new reflector (entry r), built below
    // The diagonal block D = A[I_q, I_q] is not touched before its own update at the end of the
    // step: request it now, so that its LDS round trip overlaps the work on C (the step is a
    // chain of dependent operations; every exposed round trip counts).
    double dv[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = CPL * cq + k;
        dv[k] = (r < L && c < L) ? B.sym(i0 + r, i0 + c) : 0.0;
    }
    if (q == 0) {
        // eliminate column s below its first sub-diagonal entry
        const double x = (r < L) ? B(i0 + r, s) : 0.0;
        const double ss = rsum16((r >= 1) ? x * x : 0.0);   // every 16-lane group holds the column
        if (sl == 0) sx[0] = x;
        wave_lds_fence();
        const double alpha = sx[0];
        double beta = alpha, scale = 0.0;
        if (ss != 0.0 && L >= 2) {
            beta = -copysign(fast_sqrt(fma(alpha, alpha, ss)), alpha);
            tau = (beta - alpha) * fast_rcp(beta);
            scale = fast_rcp(alpha - beta);
        }
        vr = (r == 0) ? 1.0 : x * scale;
        if (r >= L) vr = 0.0;
        if (tau != 0.0 && cq == 0 && r < L) B(i0 + r, s) = (r == 0) ? beta : 0.0;
    } else {
        const int j0 = i0 - SB;                  // I_{q-1} = [j0, j0 + SB)
        const double tprev = vprev[SB];
        // C(r, c) = A[i0 + r, j0 + c], CPL columns per lane
        double cv[CPL], vp[CPL];
        double w = 0.0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = CPL * cq + k;
            vp[k] = vprev[c];
            cv[k] = (r < L) ? B(i0 + r, j0 + c) : 0.0;
            w = fma(cv[k], vp[k], w);
        }
        w = rowsum(w);
        const double tw = tprev * w;
#pragma unroll
        for (int k = 0; k < CPL; ++k) cv[k] = fma(-tw, vp[k], cv[k]);   // C <- C H_prev
        // new reflector from the first column of C (held by the cq == 0 lanes)
        if (cq == 0) sx[r] = cv[0];
        wave_lds_fence();
        const double x = sx[r];
        const double alpha = sx[0];
        const double ss = rsum16((r >= 1 && r < L) ? x * x : 0.0);
        double beta = alpha, scale = 0.0;
        if (ss != 0.0 && L >= 2) {
            beta = -copysign(fast_sqrt(fma(alpha, alpha, ss)), alpha);
            tau = (beta - alpha) * fast_rcp(beta);
            scale = fast_rcp(alpha - beta);
        }
        vr = (r == 0) ? 1.0 : x * scale;
        if (r >= L) vr = 0.0;
        if (tau != 0.0) {
            // C <- H C : z_c = sum_r v_r C(r, c)
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const double z = rsum16(vr * cv[k]);
                cv[k] = fma(-tau * z, vr, cv[k]);
            }
            if (cq == 0) cv[0] = (r == 0) ? beta : 0.0;   // exact zeros in the annihilated column
        }
        if (r < L) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) B(i0 + r, j0 + CPL * cq + k) = cv[k];
        }
    }
    // publish v: next step of this sweep, the D update below, and the back-transformation
    if (cq == 0) {
        sv[r] = vr;
        refl_v[r] = vr;
    }
    if (sl == 0) *refl_tau = tau;
    wave_lds_fence();
    // two-sided update of the diagonal block D = A[I_q, I_q] (lower part stored)
    if (tau != 0.0) {
        double vc[CPL];
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            vc[k] = sv[CPL * cq + k];
            p = fma(dv[k], vc[k], p);
        }
        p = rowsum(p);
        p *= tau;                                   // p_r
        const double pv = rsum16(p * vr);           // p and v are replicated in the column groups
        const double wr = fma(-0.5 * tau * pv, vr, p);   // w_r = p_r - tau/2 (p.v) v_r
        if (cq == 0) sw[r] = wr;
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = CPL * cq + k;
            const double wc = sw[c];
            if (r < L && c <= r) B(i0 + r, i0 + c) = dv[k] - vr * wc - wr * vc[k];
        }
    }
    // hand the reflector to the next step (same slot)
    if (cq == 0) vprev[r] = vr;
    if (sl == 0) vprev[SB] = tau;
}

// IN_LDS: the band lives in LDS.  A compile-time switch, so that every band access is a 32-bit
// ds_read / ds_write; with a run-time choice of the base pointer the accesses become FLAT
// instructions with 64-bit address arithmetic.  NT threads = NT / (16 NCQ) slots, one sweep each.
template <bool IN_LDS, int NCQ, int NT>
__global__ __launch_bounds__(NT) void band_chase_kernel(
    const int *__restrict__ ns, const int64_t *__restrict__ moff, const int64_t *__restrict__ voff,
    const int64_t *__restrict__ roff, const double *__restrict__ Wm, double *__restrict__ bandg,
    int band_in_lds, double *__restrict__ dd, double *__restrict__ ee, double *__restrict__ rv,
    double *__restrict__ rtau) {
    extern __shared__ __align__(16) double lds[];
    constexpr int NSLOT = NT / (16 * NCQ);
    const int b = blockIdx.x;
    const int n = ns[b];
    const double *A = Wm + moff[b];
    const int64_t vo = voff[b];
    double *d = dd + vo, *e = ee + vo;
    double *RV = rv + roff[b] * SB;
    double *RT = rtau + roff[b];
    const int tid = threadIdx.x, lane = tid & 63;
    const int slot = tid / (16 * NCQ);
    // LDS: per-slot reflector hand-off [NSLOT][HAND], cum[n+1] and start[n] ints, then
    // (optionally) the band
    double *hand = lds;
    int *cum = (int *)(hand + NSLOT * HAND);
    int *start = cum + n + 1;
    double *bandl = (double *)(cum + 2 * n + 2);
    BandRef B;
    B.p = IN_LDS ? bandl : (bandg + vo * LDB);
    // load the band (zero beyond it) -- A holds it in its lower triangle
    for (int idx = tid; idx < n * LDB; idx += NT) {
        const int j = idx / LDB, t = idx % LDB;
        const int i = j + t;
        B.p[idx] = (t <= SB && i < n) ? A[(size_t)j * n + i] : 0.0;
    }
    // Schedule: step q of sweep s runs at time start[s] + q on slot s mod NSLOT.  Sweep s + 1 must
    // stay two steps behind sweep s (its block q overlaps blocks q, q + 1 of s), and a slot must
    // have finished sweep s - NSLOT.  The sweeps get shorter, so the spacing drops to the minimum
    // of 2 once NSLOT sweeps in flight cover a whole sweep.
    if (tid == 0) {
        int run = 0;
        for (int s = 0; s < n; ++s) {
            cum[s] = run;
            if (s <= n - 3) run += chase_steps(n, s);
            int st = (s == 0) ? 0 : start[s - 1] + 2;
            if (s >= NSLOT) st = max(st, start[s - NSLOT] + chase_steps(n, s - NSLOT));
            start[s] = st;
        }
        cum[n] = run;
    }
    __syncthreads();
    if (n >= 3) {
        const int nsweeps = n - 2;
        const int tend = start[nsweeps - 1] + chase_steps(n, nsweeps - 1);
        double *myhand = hand + slot * HAND;
        int s = slot;                               // the sweep this slot works on
        int s_begin = 0, s_end = 0;
        if (s < nsweeps) { s_begin = start[s]; s_end = s_begin + chase_steps(n, s); }
        for (int t = 0; t < tend; ++t) {
            if (s < nsweeps && t >= s_end) {
                s += NSLOT;
                if (s < nsweeps) { s_begin = start[s]; s_end = s_begin + chase_steps(n, s); }
            }
            if (s < nsweeps && t >= s_begin) {
                const int q = t - s_begin;
                const int rid = cum[s] + q;
                chase_step<NCQ>(B, n, s, q, myhand, RV + (size_t)rid * SB, RT + rid, lane);
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += NT) {
        d[i] = B(i, i);
        e[i] = (i + 1 < n) ? B(i + 1, i) : 0.0;
    }
}

// ---------------------------------------------------------------------------------------
// back-transformation of the eigenvectors:  y = Q1 Q2 z , then x = D^-1/2 y
// ---------------------------------------------------------------------------------------
// One workgroup per matrix, vectors one after the other with y resident in LDS.  Q2: per sweep
// every 16-lane group applies one chase reflector (they touch disjoint rows), one barrier per
// sweep.  Q1: compact-WY panels in reverse order.
template <int NT>
__global__ __launch_bounds__(NT) void backtransform2_kernel(
    const int *__restrict__ ns, const int64_t *__restrict__ moff, const int64_t *__restrict__ voff,
    const int64_t *__restrict__ roff, const double *__restrict__ Wm, const double *__restrict__ Tfac,
    const double *__restrict__ rv, const double *__restrict__ rtau, const double *__restrict__ dis,
    const int *__restrict__ ms, const int64_t *__restrict__ xoff, double *__restrict__ evecs) {
    extern __shared__ __align__(16) double lds[];
    const int b = blockIdx.x;
    const int n = ns[b], m = ms[b];
    const double *A = Wm + moff[b];
    const int64_t vo = voff[b];
    const double *RV = rv + roff[b] * SB;
    const double *RT = rtau + roff[b];
    double *Y = evecs + xoff[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    constexpr int NGRP = NT / 16;
    double *y = lds;                 // [n]
    double *g = y + n;               // [SB]
    double *hv = g + SB;             // [SB]
    int *cum = (int *)(hv + SB);     // [n + 1]
    if (tid == 0) {
        int run = 0;
        for (int s = 0; s < n; ++s) {
            cum[s] = run;
            if (s <= n - 3) run += chase_steps(n, s);
        }
        cum[n] = run;
    }
    const int grp = tid >> 4, r = tid & 15;
    int kmax = -1;
    if (n - SB >= 2) {
        kmax = 0;
        while (n - (kmax + SB) - SB >= 2) kmax += SB;
    }
    for (int jj = 0; jj < m; ++jj) {
        double *Yj = Y + (size_t)jj * n;
        __syncthreads();
        for (int i = tid; i < n; i += NT) y[i] = Yj[i];
        __syncthreads();
        // Q2: sweeps in reverse order.  The reflector of the NEXT sweep (global memory) is requested
        // before the current one is applied: otherwise every sweep pays a full memory round trip
        // between two barriers (403 of them at n = 405).
        double tau_n = 0.0, vr_n = 0.0;
        if (n >= 3) {
            const int s = n - 3;
            if (grp < chase_steps(n, s)) {
                tau_n = RT[cum[s] + grp];
                vr_n = RV[(size_t)(cum[s] + grp) * SB + r];
            }
        }
        for (int s = n - 3; s >= 0; --s) {
            const int nst = chase_steps(n, s);
            const int base = cum[s];
            const double tau_c = tau_n, vr_c = vr_n;
            if (s > 0 && grp < chase_steps(n, s - 1)) {
                tau_n = RT[cum[s - 1] + grp];
                vr_n = RV[(size_t)(cum[s - 1] + grp) * SB + r];
            }
            for (int q = grp; q < nst; q += NGRP) {
                const int i0 = s + 1 + q * SB;
                const bool act = (i0 + r < n);
                const double tau = (q == grp) ? tau_c : RT[base + q];
                const double vr = act ? ((q == grp) ? vr_c : RV[(size_t)(base + q) * SB + r]) : 0.0;
                const double yr = act ? y[i0 + r] : 0.0;
                const double dot = rsum16(vr * yr);
                if (act && tau != 0.0) y[i0 + r] = fma(-tau * dot, vr, yr);
            }
            __syncthreads();
        }
        // Q1: panels in reverse order, y[k0+SB:] -= V (T (V^T y))
        for (int k0 = kmax; k0 >= 0; k0 -= SB) {
            const int np = n - k0 - SB;
            const double *T = Tfac + vo * SB + (size_t)(k0 / SB) * SB * SB;
            double *yp = y + k0 + SB;
            for (int c = wave; c < SB; c += NW) {
                double s2 = 0.0;
                for (int i = lane; i < np; i += 64) s2 = fma(vmask(A, n, k0, i, c), yp[i], s2);
                s2 = wsum64(s2);
                if (lane == 0) g[c] = s2;
            }
            __syncthreads();
            if (tid < SB) {  // h = T g  (T upper triangular)
                double s2 = 0.0;
                for (int c = tid; c < SB; ++c) s2 = fma(T[c * SB + tid], g[c], s2);
                hv[tid] = s2;
            }
            __syncthreads();
            for (int i = tid; i < np; i += NT) {
                double s2 = 0.0;
#pragma unroll
                for (int c = 0; c < SB; ++c) s2 = fma(vmask(A, n, k0, i, c), hv[c], s2);
                yp[i] -= s2;
            }
            __syncthreads();
        }
        for (int i = tid; i < n; i += NT) Yj[i] = y[i] * dis[vo + i];
    }
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
int64_t chase_reflector_count(int n) {
    int64_t r = 0;
    for (int s = 0; s + 3 <= n; ++s) r += chase_steps(n, s);
    return r;
}

// Lanes per chase step and threads per matrix (SAAMGE_AMD_CHASE="ncq,nt" overrides; ncq = 4 | 2
// column groups of 16 lanes per step, nt / (16 ncq) sweeps in flight).  Measured at 128^3
// (n = 405, 8 192 matrices): 64 lanes per step 42.9 ms, 32 lanes 70.2 ms, 16 lanes 96 ms, and
// 32 slots instead of 16 do not help the 2 600-row level-1 matrices either (80 vs 60 ms): a step
// is a chain of ~180 dependent operations (LDS round trips, 64-bit DPP sums, the reflector's
// rsq / rcp), and spreading a step over more lanes shortens that chain -- fewer lanes per step
// save instructions but the kernel is bound by the chain, not by issue slots.
struct ChaseConfig {
    int ncq, nt;
    int slots() const { return nt / (16 * ncq); }
};
// a whole wavefront per chase step (NCQ = 4), 16 slots per workgroup: a sweep of n rows keeps ~n / 32 sweeps in
// flight at the minimum spacing of two steps (half-wavefront slots, NCQ = 2, measured slower in round 1)
static ChaseConfig chase_config(int) { return ChaseConfig{4, 1024}; }

void eig_tridiagonalize_two_stage(hipStream_t s, EigBatch &b, int phases) {
    if (!b.count) return;
    static bool attr = false;
    if (!attr) {
#define SA_CHASE_ATTR(L, Q, T)                                                              \
    SA_HIP_CHECK(hipFuncSetAttribute((const void *)band_chase_kernel<L, Q, T>,                 \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        SA_CHASE_ATTR(true, 4, 1024); SA_CHASE_ATTR(false, 4, 1024);
#undef SA_CHASE_ATTR
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)backtransform2_kernel<256>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)backtransform2_kernel<1024>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)sbr_qr_kernel<256, true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr = true;
    }
    const size_t rows = (size_t)b.h_voff[b.count];
    (void)rows;
    b.h_roff.assign((size_t)b.count + 1, 0);
    for (int i = 0; i < b.count; ++i) b.h_roff[i + 1] = b.h_roff[i] + chase_reflector_count(b.h_n[i]);
    const size_t nrefl = (size_t)b.h_roff[b.count];
    const int nmax0 = b.max_n;
    const size_t fixed0 = sizeof(double) * (chase_config(nmax0).slots() * HAND) + sizeof(int) * (2 * (size_t)nmax0 + 4);
    const bool in_lds0 = fixed0 + sizeof(double) * (size_t)nmax0 * LDB + 64 <= 160 * 1024;
    b.h_xpoff.clear();
    b.h_goff.assign((size_t)b.count + 1, 0);
    for (int i = 0; i < b.count; ++i)
        b.h_goff[i + 1] = b.h_goff[i] + (int64_t)((b.h_n[i] + 63) / 64) * SB * SB;
    if (phases & 1) eig_batch_two_stage_buffers(b, nrefl, !in_lds0, s);
    double flops = 0.0, bytes = 0.0;
    for (int n : b.h_n) {
        flops += 4.0 / 3.0 * (double)n * n * n;
        bytes += 8.0 * (double)n * n;
    }
    // ---- stage 1 ----
    const int nmax = b.max_n;
    const bool prof = profiler().enabled;
    // The product of panel p + 1 rides on the update of panel p (one fused kernel, vector FMAs with scalar-cache
    // operands).  Measured alternatives (round 1, profiles/r01_*): separate product / update kernels, the same
    // update on the matrix cores (fp64 MFMA sustains 46 TFLOP/s against 69 for v_fma_f64: tools/fma64_bench.hip)
    // and a lower-triangle-only variant were all slower or equal -- none is bound by FMA issue or HBM bytes, they
    // are bound by operand delivery (scalar-load latency).
    constexpr int rpl = 2;      // rows per lane of the fused kernel (one row per lane: measured slower on the dense reduction, round 1)
    if (phases & 1) {
    if (!prof) profiler().begin(s);
    const int cnt8 = 8 * div_up(b.count, 8);
    auto launch_qr = [&](int k0, double *Vp) {
        const int npmax = nmax - k0 - SB;
        if (npmax <= 512)
            hipLaunchKernelGGL(sbr_qr_reg_kernel, dim3(b.count), dim3(256), 0, s, k0, b.n.p, b.moff.p,
                               b.voff.p, b.W.p, b.Tfac.p, Vp);
        else if ((size_t)npmax * SB * sizeof(double) <= 96 * 1024)
            hipLaunchKernelGGL((sbr_qr_kernel<256, true>), dim3(b.count), dim3(256),
                               (size_t)npmax * SB * sizeof(double), s, k0, b.n.p, b.moff.p, b.voff.p,
                               b.W.p, b.Tfac.p, Vp);
        else
            hipLaunchKernelGGL((sbr_qr_kernel<1024, false>), dim3(b.count), dim3(1024), 0, s, k0,
                               b.n.p, b.moff.p, b.voff.p, b.W.p, b.Tfac.p, Vp);
    };
    auto launch_symm = [&](int k0, double *Vp) {
        const int npmax = nmax - k0 - SB;
        hipLaunchKernelGGL(sbr_symm_kernel, dim3(cnt8 * div_up(npmax, 64)), dim3(SY_NT), 0, s, k0,
                           b.n.p, b.moff.p, b.voff.p, b.W.p, Vp, b.Xbuf.p, b.goff.p, b.Gbuf.p, b.count,
                           div_up(npmax, 64));
    };
    auto launch_z = [&](int k0, double *Vp) {
        const int npmax = nmax - k0 - SB;
        hipLaunchKernelGGL(sbr_z_kernel, dim3(cnt8 * div_up(npmax, SM_NT)), dim3(SM_NT), 0, s, k0,
                           b.n.p, b.voff.p, Vp, b.Tfac.p, b.Xbuf.p, b.goff.p, b.Gbuf.p,
                           b.Zbuf.p, b.count, div_up(npmax, SM_NT));
    };
    if (nmax - SB >= 2) {
        // look-ahead pipeline: the product of panel p+1 rides on the update of panel p
        double *Vcur = b.Vpk.p, *Vnext = b.Vpk2.p;
        if (prof) profiler().begin(s);
        launch_qr(0, Vcur);
        if (prof) { profiler().end(s, "eig_sbr_qr", 0.0, 0.0); profiler().begin(s); }
        launch_symm(0, Vcur);
        if (prof) { profiler().end(s, "eig_sbr_symm", 0.0, 0.0); profiler().begin(s); }
        launch_z(0, Vcur);
        if (prof) profiler().end(s, "eig_sbr_z", 0.0, 0.0);
        for (int k0 = 0; nmax - k0 - SB >= 2; k0 += SB) {
            const int npmax = nmax - k0 - SB;          // order of A22(k0)
            const int npn = npmax - SB;                // order of A22' = A22(k0 + SB)
            const bool has_next = npn >= 2;
            if (prof) profiler().begin(s);
            hipLaunchKernelGGL(sbr_panel_update_kernel, dim3(cnt8 * div_up(npmax, 256)), dim3(256), 0, s,
                               k0, b.n.p, b.moff.p, b.voff.p, b.W.p, Vcur, b.Zbuf.p, b.count,
                               div_up(npmax, 256));
            if (prof) { profiler().end(s, "eig_sbr_panel", 0.0, 0.0); profiler().begin(s); }
            if (has_next) launch_qr(k0 + SB, Vnext);
            if (prof) { profiler().end(s, "eig_sbr_qr", 0.0, 0.0); profiler().begin(s); }
            // two rows per lane are ~10 % faster per row slot but pad the strip count to 128 rows
            const bool two_rows = rpl == 2 &&
                (double)div_up(npn, 2 * SF_ROWS) * (2 * SF_ROWS) < 1.10 * (double)div_up(npn, SF_ROWS) * SF_ROWS;
            if (npn >= 1 && two_rows)
                hipLaunchKernelGGL((sbr_fused_kernel<true, 2>), dim3(cnt8 * div_up(npn, 2 * SF_ROWS)), dim3(S2_NT), 0, s,
                                   k0, b.n.p, b.moff.p, b.voff.p, b.W.p, Vcur, b.Zbuf.p, Vnext, b.Xbuf.p,
                                   b.goff.p, b.Gbuf.p, b.trash.p, b.count, div_up(npn, 2 * SF_ROWS), SB);
            else if (npn >= 1)
                hipLaunchKernelGGL((sbr_fused_kernel<true, 1>), dim3(cnt8 * div_up(npn, SF_ROWS)), dim3(S2_NT), 0, s,
                                   k0, b.n.p, b.moff.p, b.voff.p, b.W.p, Vcur, b.Zbuf.p, Vnext, b.Xbuf.p,
                                   b.goff.p, b.Gbuf.p, b.trash.p, b.count, div_up(npn, SF_ROWS), SB);
            if (prof) {
                // (one label per kernel symbol, bytes of THIS launch: 16 np'^2 per matrix)
                double lb = 0.0;
                for (int n : b.h_n) {
                    const double q = (double)n - k0 - 2 * SB;
                    if (n - k0 - SB >= 2 && q >= 1.0) lb += 16.0 * q * q;
                }
                profiler().end(s, two_rows ? "eig_sbr_fused" : "eig_sbr_fused1", lb, 0.0);
                profiler().begin(s);
            }
            if (has_next) launch_z(k0 + SB, Vnext);
            if (prof) profiler().end(s, "eig_sbr_z", 0.0, 0.0);
            std::swap(Vcur, Vnext);
        }
    }
    SA_HIP_CHECK(hipGetLastError());
    if (!prof) profiler().end(s, "eig_band_reduce", bytes, flops);
    }
    if (!(phases & 2)) return;
    // ---- stage 2 ----
    const ChaseConfig cc = chase_config(nmax);
    const size_t fixed = sizeof(double) * (cc.slots() * HAND) + sizeof(int) * (2 * (size_t)nmax + 4);
    const size_t band_bytes = sizeof(double) * (size_t)nmax * LDB;
    const int in_lds = (fixed + band_bytes + 64 <= 160 * 1024) ? 1 : 0;
    double cflops = 0.0;
    for (int n : b.h_n) cflops += 6.0 * (double)n * n * SB;
    profiler().begin(s);
#define SA_CHASE(Q, T)                                                                                          \
    if (cc.ncq == Q && cc.nt == T) {                                                                            \
        if (in_lds)                                                                                             \
            hipLaunchKernelGGL((band_chase_kernel<true, Q, T>), dim3(b.count), dim3(T), fixed + band_bytes + 64, s, \
                               b.n.p, b.moff.p, b.voff.p, b.roff.p, b.W.p, b.bandg.p, in_lds, b.d.p, b.e.p,     \
                               b.rv.p, b.rtau.p);                                                               \
        else                                                                                                    \
            hipLaunchKernelGGL((band_chase_kernel<false, Q, T>), dim3(b.count), dim3(T), fixed + 64, s, b.n.p,  \
                               b.moff.p, b.voff.p, b.roff.p, b.W.p, b.bandg.p, in_lds, b.d.p, b.e.p, b.rv.p,    \
                               b.rtau.p);                                                                       \
    }
    SA_CHASE(4, 1024)
#undef SA_CHASE
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "eig_band_chase", 0.0, cflops);
}

void eig_backtransform_two_stage(hipStream_t s, EigBatch &b, const int64_t *xoff, double *evecs) {
    if (!b.count) return;
    const size_t lds = sizeof(double) * ((size_t)b.max_n + 2 * SB) + sizeof(int) * ((size_t)b.max_n + 4) + 64;
    SA_REQUIRE(lds <= 160 * 1024, "agglomerate too large for the LDS-resident back-transformation");
    double flops = 0.0;
    for (int i = 0; i < b.count; ++i) flops += 4.0 * (double)b.h_n[i] * b.h_n[i] * b.h_m[i];
    profiler().begin(s);
    if (b.max_n > 1024)
        hipLaunchKernelGGL((backtransform2_kernel<1024>), dim3(b.count), dim3(1024), lds, s, b.n.p,
                           b.moff.p, b.voff.p, b.roff.p, b.W.p, b.Tfac.p, b.rv.p, b.rtau.p, b.dis.p,
                           b.m.p, xoff, evecs);
    else
        hipLaunchKernelGGL((backtransform2_kernel<256>), dim3(b.count), dim3(256), lds, s, b.n.p,
                           b.moff.p, b.voff.p, b.roff.p, b.W.p, b.Tfac.p, b.rv.p, b.rtau.p, b.dis.p,
                           b.m.p, xoff, evecs);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "eig_backtransform", 0.0, flops);
}


// =========================================================================================
// Few-eigenpairs path (SAAMGE_AMD_EIG=subspace, opt-in): the agglomerates want one or two
// eigenpairs and the dense reduction pays 2 n^3 flops for them.  Here: C - sigma I = L L^T
// (sigma < 0, C is positive semidefinite; n^3/3 FMAs through the SAME rank-16 update kernel as
// the band reduction, with Z = L21 / 2, V = L21), then shift-invert subspace iteration on a block
// of SS_B vectors with a Rayleigh-Ritz step per iteration that needs only the solves:
//   Z = (C - sigma)^-1 X,  M = Z^T X (= Z^T (C - sigma) Z),  G = Z^T Z,  M c = mu G c,
//   X <- Z c,  lambda = sigma + mu.
// A pair is accepted when its inverse residual || Z_j - X_j / mu_j || bounds the residual of C
// below the tolerance (EigBatch::ss_tol); the count is certified by the first unwanted pair having converged above vu.
// Anything that does not fit (more than SS_B - 2 wanted pairs, no convergence, a non-positive
// pivot) makes the caller fall back to the dense path on a re-assembled matrix.
// =========================================================================================
constexpr int SS_B = 8;
constexpr double SS_SIGMA = -1e-3;
constexpr int SS_MAX_ITER = 80;
// Matrices with more wanted pairs than the block's six: the first six converged pairs are LOCKED (copied out, the block
// started again and kept orthogonal to them: ss_lock_kernel, ss_deflate_kernel) and the iteration goes on for the
// next ones, on the same factor.  One lock of six: up to twelve wanted pairs.
constexpr int SS_LOCK = 6, SS_LOCK_PITCH = 8, SS_WANT_MAX = 2 * SS_LOCK;

__global__ __launch_bounds__(256) void ss_shift_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                       double *__restrict__ W, double sigma_all,
                                                       const double *__restrict__ sigmas = nullptr) {
    const int b = blockIdx.x, n = ns[b];
    const double sigma = sigmas ? sigmas[b] : sigma_all;
    double *A = W + moff[b];
    for (int i = threadIdx.x; i < n; i += 256) A[(size_t)i * n + i] -= sigma;
}

// Half bandwidth of every matrix: bw = max (i - j) over the non-zero entries below the diagonal.
// The agglomerate-local numbering of a mesh keeps it far below n (405-row box agglomerates of the
// 256^3 workload: 91); Cholesky fill stays inside the band, so the factorisation and the solves
// skip everything outside it -- exactly the entries a dense factorisation would carry as zeros.
__global__ __launch_bounds__(256) void ss_band_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                      const double *__restrict__ W, int *__restrict__ bws) {
    __shared__ int wmax[4];
    const int b = blockIdx.x, n = ns[b];
    const double *A = W + moff[b];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bw = 0;
    // one wavefront per column, bottom up, until the first non-zero; gridDim.y workgroups share a matrix
    for (int j = blockIdx.y * 4 + w; j < n; j += 4 * gridDim.y) {
        const double *col = A + (size_t)j * n;
        for (int r1 = n; r1 > j + 1 + bw; r1 -= 64) {
            const int r = r1 - 64 + lane;
            const bool nz = r > j + bw && col[r] != 0.0;
            const unsigned long long m = __ballot(nz);
            if (m) { bw = max(bw, r1 - 64 + 63 - __builtin_clzll(m) - j); break; }
        }
    }
    if (lane == 0) wmax[w] = bw;
    __syncthreads();
    if (tid == 0) atomicMax(bws + b, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));    // (zeroed by the host)
}

#include "chol16.h"

// Cholesky of the SB x SB diagonal block at k0 and L21 = A21 L11^-T below it.  L goes to the lower
// triangle, its transpose to the upper one (the solves then stream columns both ways), and the
// packed row-major copies V = L21, Z = L21 / 2 feed the trailing update.
// SIGNED: the same walk for the indefinite C - theta I = L S L^T (inertia count of the wide-band
// matrices): the packed copies are V = L21, Z = L21 S / 2 (so that Z V^T + V Z^T = L21 S L21^T in
// the next-columns update) and Sout = L21 S (row operands of the trailing update); the matrix keeps
// only what later panels read (nothing of the factor is written back); neg[b] accumulates the
// negative pivots, info[b] = 1 marks a pivot too small to trust the count.
template <int NT, bool SIGNED = false>
__global__ __launch_bounds__(NT) void chol_panel_kernel(int k0, const int *__restrict__ ns,
                                                         const int64_t *__restrict__ moff,
                                                         const int64_t *__restrict__ voff, double *__restrict__ W,
                                                         double *__restrict__ Vpk, double *__restrict__ Zbuf,
                                                         int *__restrict__ info, const int *__restrict__ bws,
                                                         int rext, double *__restrict__ Sout = nullptr,
                                                         int *__restrict__ neg = nullptr, int keep = 0,
                                                         const int *__restrict__ skip = nullptr) {
    // keep (SIGNED): the factor IS written back (L below, L^T above the diagonal, inverted diagonal blocks), as
    // the unsigned walk does -- where every pivot turns out positive, L S L^T is the Cholesky factorisation
    // of C - theta I and the few-eigenpairs path iterates with it instead of factoring again.
    // skip: matrices this launch leaves alone (second factorisation of a batch whose other members kept theirs)
    __shared__ double Ld[SB][SB + 1];
    const int b = blockIdx.x, n = ns[b];
    if (k0 >= n) return;
    if (skip && skip[b]) return;
    // rows below the band are zero and stay zero; `rext` more rows are still written (as zeros) to the
    // packed panel for the two-panel update that reads them
    const int rend = bws ? min(n, k0 + SB + bws[b] + rext) : n;
    double *A = W + moff[b];
    const int nb = min(SB, n - k0);
    const int tid = threadIdx.x;
    if (tid < SB * SB) {
        const int i = tid >> 4, j = tid & 15;
        Ld[i][j] = (i < nb && j <= i) ? A[(size_t)(k0 + j) * n + (k0 + i)] : ((i == j) ? 1.0 : 0.0);
    }
    __syncthreads();
    // Cholesky of the block and its INVERSE by ONE wavefront (the diagonal block is kept as L11^-1,
    // lower part, and its transpose: the solves then apply it as a small matrix product instead of
    // a serial substitution); the other wavefronts wait at a single barrier
    __shared__ int bad;
    __shared__ double Li[SB][SB + 1];
    __shared__ double sg[SB];
    if (tid < 64) {
        const int r = chol16_inverse_wave<SIGNED>(Ld, Li, tid, sg);
        if (tid == 0) bad = r;
    }
    __syncthreads();
    if (SIGNED) {
        if (tid == 0) { if (bad < 0) info[b] = 1; else neg[b] += bad; }
    } else {
        if (tid == 0 && bad) info[b] = 1;
    }
    if (!SIGNED || keep) {
        if (tid < SB * SB) {
            const int i = tid >> 4, j = tid & 15;
            if (i < nb && j <= i) {
                A[(size_t)(k0 + j) * n + (k0 + i)] = Li[i][j];
                A[(size_t)(k0 + i) * n + (k0 + j)] = Li[i][j];
            }
        }
    }
    if (nb < SB) return;
    double *Vp = Vpk + voff[b] * SB, *Zp = Zbuf ? Zbuf + voff[b] * SB : nullptr;
    double *Sp = (SIGNED && Sout) ? Sout + voff[b] * SB : nullptr;
    for (int r = k0 + SB + tid; r < rend; r += NT) {
        double x[SB];
#pragma unroll
        for (int c = 0; c < SB; ++c) x[c] = A[(size_t)(k0 + c) * n + r];
        const size_t pr = (size_t)(r - k0 - SB) * SB;
        // row of L21 = x L11^-T: y_c = sum_j x_j (L11^-1)(c, j) (zero above the diagonal); two columns per
        // trip so that the 256 LDS operands are not all hoisted into registers
#pragma unroll 2
        for (int c = 0; c < SB; ++c) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < SB; ++j) t = fma(x[j], Li[c][j], t);
            if (SIGNED) {
                const double ts = t;                 // = (L21 S)(r, c)
                t *= sg[c];
                Vp[pr + c] = t;
                if (Zp) Zp[pr + c] = 0.5 * ts;
                if (Sp) Sp[pr + c] = ts;
                if (keep) A[(size_t)(k0 + c) * n + r] = t;      // (L only, coalesced; L^T: ss_keep_transpose_kernel, kept matrices only)
            } else {
                A[(size_t)(k0 + c) * n + r] = t;
                A[(size_t)r * n + (k0 + c)] = t;
                Vp[pr + c] = t;
                if (Zp) Zp[pr + c] = 0.5 * t;
            }
        }
    }
}


// Kept inertia factors (eig_subspace_factor): the signed panel kernel writes L only -- the transposed, strided
// writes of L^T cost every matrix of the batch 180 us per panel (0.6 s of config 5's 23 s, where no matrix keeps its
// factor), this pass costs the few that do one sweep.  Column j of L, below its diagonal block, goes to row j of
// the upper triangle; the diagonal blocks (inverses, both triangles) are the panel kernel's.
__global__ __launch_bounds__(256) void ss_keep_transpose_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                                double *__restrict__ W, const int *__restrict__ bws,
                                                                const int *__restrict__ keep) {
    const int b = blockIdx.x;
    if (!keep[b]) return;
    const int n = ns[b], reach = (bws ? bws[b] : n) + 2 * SB;
    double *A = W + moff[b];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = blockIdx.y * 4 + wv; j < n; j += 4 * gridDim.y) {
        const int i0 = (j / SB + 1) * SB, i1 = min(n, j + reach + 1);
        for (int i = i0 + lane; i < i1; i += 64) A[(size_t)i * n + j] = A[(size_t)j * n + i];
    }
}

// full band |i - j| <= bw of every column, packed (column j of matrix b at soff[b] + j (2 bw + 1)):
// saved before / restored after the in-place inertia factorisation of the wide-band matrices
template <bool RESTORE>
__global__ __launch_bounds__(256) void band_copy_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                        double *__restrict__ W, const int *__restrict__ bws,
                                                        const int64_t *__restrict__ soff, double *__restrict__ save,
                                                        const int *__restrict__ skip = nullptr) {
    if (skip && skip[blockIdx.x]) return;
    // (at least the diagonal blocks: a kept factor carries the dense inverses of its SB x SB diagonal blocks)
    const int b = blockIdx.x, n = ns[b], bw = min(max(bws[b], SB - 1), n - 1), w2 = 2 * bw + 1;
    double *A = W + moff[b];
    double *S = save + soff[b];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = blockIdx.y * 4 + wv; j < n; j += 4 * gridDim.y) {
        const int lo = max(0, j - bw), hi = min(n - 1, j + bw);
        for (int i = lo + lane; i <= hi; i += 64) {
            if (RESTORE) A[(size_t)j * n + i] = S[(size_t)j * w2 + (i - j + bw)];
            else S[(size_t)j * w2 + (i - j + bw)] = A[(size_t)j * n + i];
        }
    }
}


// Banded Cholesky with the band resident in LDS: ONE launch factors every matrix whose half
// bandwidth fits the window (bw <= BC_MAXBW), one workgroup per matrix.  The rows k0 .. k0 + SB + bw
// of the band live in a circular BC_WIN x BC_WIN window (entry (i, j) at (i mod BC_WIN, j mod
// BC_WIN)); per block step: Cholesky + inverse of the diagonal block, L21 (kept in LDS for the
// update, written to both triangles of the matrix), the rank-SB update of the window by 4 x 4
// register tiles, and the next SB rows of the band are fetched (from the upper triangle: it still
// holds the original entries, column i rows i - bw .. i are contiguous).  The band is read once and
// written once; nothing else touches HBM.
// The window size is a template parameter: 128 (bands up to 112, one workgroup per CU), 80 (up to 64,
// two per CU) and 68 (up to 52, three per CU) -- the kernel is a chain of short dependent stages, so
// the resident workgroups of other matrices are what fills the CU.
constexpr int BC_NT = 256;
// Pitch of a window row: band + 1 columns, made EVEN wherever that costs no resident workgroup.  The panel stage reads
// entry (r, k0 + c) with thread = row r, i.e. with a stride of pitch - 1 doubles between lanes: 52 doubles (window 68, pitch
// 53) put the 64 lanes on 8 of the 32 bank pairs (round 3's counters: 80 % of the kernel's LDS cycles were bank conflicts),
// 64 doubles (window 80, pitch 65) on ONE; an odd stride spreads them over all.  Window 68 keeps pitch 53 (one more
// double per row would cost the fourth workgroup of a CU by 512 bytes); bands up to 51 -- the 8 x 8 x 4-element agglomerates
// of the Q1 problems have exactly 51 -- take the window of 67 rows, pitch 52.
constexpr int bc_pitch(int win) { return win == 68 ? win - SB + 1 : ((win - SB + 1) + 1) / 2 * 2; }
// INERTIA = true: nothing is written back.  The same window walk factors C - shift I = L S L^T
// (S = diag(+-1), no pivoting) and info[b] receives the number of negative pivots = the number of
// eigenvalues of C below `shift` (Sylvester), or -1 when a pivot was too small to trust the count.
// This is what makes the few-eigenpairs path's count as rigorous as dsygvx's bisection
// (dstebz Sturm counts, amg/src/xpacks.cpp:226-268).
// (The kernel of rounds 2-3 ran the four stages of a block step one after the other; the pipelined kernel that replaced
// it -- chol_band_lds2_kernel -- follows the matrix-core helpers below.)

// Is the start vector x0 = D^1/2 1 / |.| already the one wanted eigenvector?  C x0 over the band of the
// (unshifted, unfactored) matrix, one workgroup per matrix, thread = row: pre[b] = 1 when the certified count
// is 1, the Rayleigh quotient lies in the window and |C x0 - rq x0| <= tol (the acceptance tolerance of the
// iteration).  True for every agglomerate without essential rows of a diffusion-type operator (the element
// matrices have zero row sums): those matrices then need neither the second factorisation nor the solves.
constexpr int NC_NT = 512;
__global__ __launch_bounds__(NC_NT) void ss_nullcheck_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                           const int64_t *__restrict__ voff, const double *__restrict__ W,
                                                           const double *__restrict__ dis, const short *__restrict__ perm,
                                                           const int *__restrict__ bws, const int *__restrict__ inertia,
                                                           double vu, double tol, int *__restrict__ pre,
                                                           double *__restrict__ pre_val,
                                                           const double *__restrict__ x0c = nullptr) {
    extern __shared__ double nc_x[];            // [n] x0 in matrix order
    __shared__ double red[3][NC_NT / 64];
    const int b = blockIdx.x, n = ns[b], tid = threadIdx.x;
    if (inertia[b] != 1) { if (tid == 0) pre[b] = 0; return; }
    const double *A = W + moff[b];
    const double *db = dis + voff[b];
    const short *pm = perm ? perm + voff[b] : nullptr;
    const int bw = bws ? bws[b] : n - 1;
    for (int r = tid; r < n; r += NC_NT) nc_x[pm ? pm[r] : r] = (x0c ? x0c[voff[b] + r] : 1.0) / db[r];
    __syncthreads();
    double xx = 0.0, xy = 0.0, yy = 0.0;
    for (int i = tid; i < n + 63; i += NC_NT) {     // (whole wavefronts: the column loop below is wavefront-uniform)
        // the wavefront owns 64 consecutive rows and walks the columns that touch any of them: every load is
        // one column, 64 consecutive rows (walking each row's own band would put the lanes on a diagonal --
        // 64 cache lines per load); entries outside the row's band are not part of the matrix (never written)
        const int ib = i & ~63, jlo = max(0, ib - bw), jhi = min(n - 1, min(n - 1, ib + 63) + bw);
        const bool live = i < n;
        const double *ap = A + min(i, n - 1);
        double y0 = 0.0, y1 = 0.0;
        int j = jlo;
        for (; j + 8 <= jhi + 1; j += 8) {          // eight columns in flight per lane
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = (live && abs(i - (j + u)) <= bw) ? __builtin_nontemporal_load(ap + (size_t)(j + u) * n) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                y0 = fma(t[u], nc_x[j + u], y0);
                y1 = fma(t[u + 1], nc_x[j + u + 1], y1);
            }
        }
        for (; j <= jhi; ++j) y0 = fma((live && abs(i - j) <= bw) ? ap[(size_t)j * n] : 0.0, nc_x[j], y0);
        if (!live) continue;
        const double y = y0 + y1;
        const double x = nc_x[i];
        xx = fma(x, x, xx);
        xy = fma(x, y, xy);
        yy = fma(y, y, yy);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { xx += __shfl_xor(xx, o, 64); xy += __shfl_xor(xy, o, 64); yy += __shfl_xor(yy, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = xx; red[1][tid >> 6] = xy; red[2][tid >> 6] = yy; }
    __syncthreads();
    if (tid == 0) {
        xx = xy = yy = 0.0;
        for (int q = 0; q < NC_NT / 64; ++q) { xx += red[0][q]; xy += red[1][q]; yy += red[2][q]; }
        const double rq = xy / xx;                                  // Rayleigh quotient of x0
        const double res2 = fmax(0.0, yy / xx - rq * rq);           // |C x - rq x|^2 for the unit vector x = x0 / |x0|
        const bool ok = rq <= vu && sqrt(res2) <= tol;
        pre[b] = ok ? 1 : 0;
        pre_val[2 * b] = rq;
        pre_val[2 * b + 1] = ok ? 1.0 / sqrt(xx) : -sqrt(res2);      // (rejected: the residual, negated, for the debug print)
    }
}

// accepted-before-the-iteration matrices: column 0 of X becomes the unit eigenvector, mu the Ritz value, the
// state "converged, one pair inside the window, one pair returned"
template <int NB = 8>
__global__ __launch_bounds__(256) void ss_preaccept_kernel(const int *__restrict__ ns, const int64_t *__restrict__ voff,
                                                           const int *__restrict__ pre, const double *__restrict__ pre_val,
                                                           const double *__restrict__ sigmas, double *__restrict__ X,
                                                           double *__restrict__ mu, int *__restrict__ state) {
    const int b = blockIdx.x;
    if (!pre[b]) return;
    const int n = ns[b];
    double *Xb = X + voff[b] * SB;
    const double sc = pre_val[2 * b + 1];
    for (int r = threadIdx.x; r < n; r += 256) Xb[(size_t)r * NB] *= sc;
    if (threadIdx.x == 0) {
        mu[(size_t)b * NB] = pre_val[2 * b] - sigmas[b];
        state[b] = 1 | (1 << 4) | (1 << 8);
    }
}

// Start block: column 0 = D^1/2 1 (the exact null vector of C for an agglomerate without essential
// rows -- most of them -- and a smooth first guess otherwise), the rest pseudo-random.  No
// dependence on the batch: the same vectors on any rank / chunking.
template <int NB = 8>
__global__ __launch_bounds__(256) void ss_init_kernel(const int *__restrict__ ns, const int64_t *__restrict__ voff,
                                                      const double *__restrict__ dis, const short *__restrict__ perm,
                                                      double *__restrict__ X, const double *__restrict__ x0c = nullptr) {
    // x0c (coarse levels): the level's representation of the constant vector, R ... R 1, in agglomerate order --
    // column 0 is then D^1/2 of it: the null vector of an agglomerate whose fine agglomerates all carry theirs
    const int b = blockIdx.x, n = ns[b];
    double *Xb = X + voff[b] * SB;
    const double *db = dis + voff[b];
    const short *pm = perm ? perm + voff[b] : nullptr;
    for (int idx = threadIdx.x; idx < n * NB; idx += 256) {
        const int r = idx / NB, pr = pm ? pm[r] : r;       // (dis is in agglomerate order, the matrix in perm order)
        Xb[pr * NB + (idx % NB)] = ((idx % NB) == 0) ? (x0c ? x0c[voff[b] + r] : 1.0) / db[r] : unit_rand_ss((unsigned)(pr * NB + (idx % NB)), (unsigned)n);
    }
}

template <int NB = 8>
__global__ __launch_bounds__(256) void ss_copy_active_kernel(const int *__restrict__ ns, const int64_t *__restrict__ voff,
                                                             const int *__restrict__ active, const double *__restrict__ X,
                                                             double *__restrict__ Z) {
    const int b = active[blockIdx.x];
    const size_t base = (size_t)voff[b] * SB, cnt = (size_t)ns[b] * NB;
    for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < cnt; i += (size_t)256 * gridDim.y) Z[base + i] = X[base + i];
}

// X <- T^-1 X for the lower (UPPER = false: L y = x) or the upper (UPPER = true: L^T z = y) factor,
// SS_B right-hand sides, rows x SS_B row-major.  Right-looking by blocks of SB: the solved block
// is eliminated from the remaining rows with coalesced column reads (the upper triangle holds L^T).
template <bool UPPER, int NT, int NB = 8>
__global__ __launch_bounds__(NT) void ss_trsolve_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                         const int64_t *__restrict__ voff,
                                                         const double *__restrict__ W, double *__restrict__ X,
                                                         const int *__restrict__ state, const int *__restrict__ bws,
        const int *__restrict__ active = nullptr) {
    __shared__ double Td[SB][SB + 1];
    __shared__ double ys[SB][NB];
    __shared__ double xs[2][SB][NB];       // right-hand side rows of the current / next block
    const int b = active ? active[blockIdx.x] : (int)blockIdx.x, n = ns[b];     // (active: the matrices still iterating)
    if (state[b] & 3) return;
    // the factor is zero beyond this distance from the diagonal (at least SB: the rows of the next
    // block are caught in LDS by the threads that update them)
    const int bw = bws ? max(bws[b], SB) : n;
    const double *A = W + moff[b];
    double *Xb = X + voff[b] * SB;
    const int tid = threadIdx.x;
    const int nblk = (n + SB - 1) / SB;
    const int ti = (tid >> 4) & 15, tj = tid & 15;
    auto load_td = [&](int k0) {             // T11^-1 of the block at k0 (stored inverted), 0 outside its triangle
        const int nb = min(SB, n - k0);
        double v = 0.0;
        if (ti < nb && tj < nb && (UPPER ? tj >= ti : tj <= ti)) v = A[(size_t)(k0 + tj) * n + (k0 + ti)];
        return v;
    };
    {
        const int k0 = UPPER ? (nblk - 1) * SB : 0;
        const int nb = min(SB, n - k0);
        if (tid < SB * NB) {
            const int c = tid / NB, jj = tid % NB;
            xs[0][c][jj] = (c < nb) ? Xb[(size_t)(k0 + c) * NB + jj] : 0.0;
        }
    }
    double td_next = load_td(UPPER ? (nblk - 1) * SB : 0);
    for (int bb = 0; bb < nblk; ++bb) {
        const int k0 = UPPER ? (nblk - 1 - bb) * SB : bb * SB;
        const int nb = min(SB, n - k0);
        const int cur = bb & 1;
        if (tid < SB * SB) Td[ti][tj] = td_next;
        __syncthreads();
        if (tid < SB * NB) {      // y = T11^-1 x
            const int c = tid / NB, jj = tid % NB;
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < SB; ++i) t = fma(Td[c][i], xs[cur][i][jj], t);
            ys[c][jj] = (c < nb) ? t : 0.0;
            if (c < nb) Xb[(size_t)(k0 + c) * NB + jj] = t;
        }
        // the next block's inverse is requested now; its rows of the right-hand side are caught in LDS by
        // the threads that update them below
        const int kn = UPPER ? k0 - SB : k0 + SB;
        if (bb + 1 < nblk) td_next = load_td(kn);
        __syncthreads();
        const int r_lo = UPPER ? max(0, k0 - bw) : k0 + nb, r_hi = UPPER ? k0 : min(n, k0 + nb + bw);
        for (int r = r_lo + tid; r < r_hi; r += NT) {
            double acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = Xb[(size_t)r * NB + j];
            const double *ap = A + (size_t)k0 * n + r;        // T(r, k0 + c): column k0 + c, row r
#pragma unroll 1
            for (int c = 0; c < SB; c += 8) {     // eight factor entries in flight (ys is zero past the block)
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ap[(size_t)min(c + u, nb - 1) * n];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[j] = fma(-t[u], ys[c + u][j], acc[j]);
            }
            const bool in_next = (r >= kn && r < kn + SB);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                Xb[(size_t)r * NB + j] = acc[j];
                if (in_next) xs[cur ^ 1][r - kn][j] = acc[j];
            }
        }
        if (bb + 1 < nblk) {        // rows of the next block beyond the matrix (partial last block of the forward solve)
            if (tid < SB * NB) {
                const int c = tid / NB, jj = tid % NB;
                if (kn + c >= n) xs[cur ^ 1][c][jj] = 0.0;
            }
        }
        __syncthreads();
    }
}

// The same solve for wide agglomerates, reorganised around what bound the kernel above on them: every thread read the
// whole 16 x 8 block y from LDS for its row -- 64 full-width LDS reads per wavefront and block step, 2.3 us of LDS time
// per step on config 5's 2 187-row agglomerates (band 668, 137 steps per triangle) on top of four memory latencies in
// a row.  Here the update  X[rows, :] -= T[rows, block] Y  of a step runs on the matrix cores (v_mfma_f64_16x16x4: a
// wavefront takes 16 rows at a time; A operand = factor entries straight from global memory, 16 consecutive rows per
// column; B operand = -Y, four LDS reads per lane and step; eight of the sixteen columns carry right-hand sides), the
// right-hand side rows a step touches -- the block and the bw rows after (before) it -- live in an LDS ring of
// WR >= bw + 32 rows, and everything a step needs from global memory is requested ahead: the factor entries of the
// coming step (at the end of the step before), the 16 rows that enter the window (untouched until then), the inverse
// of the next diagonal block.  The products of a row are accumulated in the same order as above (k ascending, one
// fused multiply-add each).
// Measured on config 5 at 64^3 (rocprofv3 kernel trace, per triangle): a chunk's full launch (577 agglomerates, 6.7 GB
// of factors) takes 2.3 - 4.1 ms -- 1.6 - 3 TB/s, memory-bound beside the next chunk's factorisation; the late launches
// (3 - 57 agglomerates still iterating) take 0.6 - 1.0 ms = 4.5 us per block step for a lone workgroup, which is what a
// single CU needs to pull 85 KB of factor per step one step ahead (in-kernel clocks: half of it waiting for them).
typedef double ss_v4d __attribute__((ext_vector_type(4)));
constexpr int TW_NT = 1024;      // threads
// TW_TPW: tiles of 16 rows per wavefront whose factor entries are requested a step ahead (4: one workgroup per CU;
// 0: nothing ahead, few enough registers for two workgroups per CU -- chunks of more than 256 matrices)
template <bool UPPER, int TW_TPW, int WGS, int NB = 8>
__global__ __launch_bounds__(TW_NT) __attribute__((amdgpu_waves_per_eu(4 * WGS, 4 * WGS))) void ss_trsolve_win_kernel(int WR, const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                                const int64_t *__restrict__ voff,
                                                                const double *__restrict__ W, double *__restrict__ X,
                                                                const int *__restrict__ state, const int *__restrict__ bws,
                                                                const int *__restrict__ active) {
    extern __shared__ __align__(16) double win[];      // [NB][WR]: WR is odd
    __shared__ double Td[SB][SB + 1];
    __shared__ double ys[SB][NB];
    const int b = active ? active[blockIdx.x] : (int)blockIdx.x, n = ns[b];
    if (state[b] & 3) return;
    const int bw = min(bws ? max(bws[b], SB) : n, WR - 2 * SB);
    const double *A = W + moff[b];
    double *Xb = X + voff[b] * SB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = (n + SB - 1) / SB;
    const int ti = (tid >> 4) & 15, tj = tid & 15;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ntile = (bw + 15) >> 4;
    // position of a row relative to the current block: p = row - k0 (lower) or k0 + SB - 1 - row (upper), 0 <= p < bw + SB;
    // slot = (wbase + p) mod WR, wbase advances by SB per step
    auto row_at = [&](int k0, int p) { return UPPER ? k0 + SB - 1 - p : k0 + p; };
    // Loads that are requested ahead are UNCONDITIONAL, from clamped addresses, and masked where they are used: a load
    // under a condition is a branch with its own wait, and twelve of them in a row were twelve latencies in a row.
    auto load_td = [&](int k0) {             // T11^-1 of the block at k0 (stored inverted), raw
        const int kc = min(max(k0, 0), n - 1);
        return A[(size_t)min(kc + tj, n - 1) * n + min(kc + ti, n - 1)];
    };
    auto mask_td = [&](int k0, double v) {   // ... 0 outside its triangle
        const int nb = min(SB, n - k0);
        return (ti < nb && tj < nb && (UPPER ? tj >= ti : tj <= ti)) ? v : 0.0;
    };
    // A operand of tile t of the step at k0: T(row of position SB + 16 t + l15, k0 + 4 kk + l4), 0 outside
    auto load_a = [&](double (&a)[4], int k0, int t) {      // raw
        const int r = min(max(row_at(k0, SB + 16 * t + l15), 0), n - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[kk] = A[(size_t)min(k0 + 4 * kk + l4, n - 1) * n + r];
    };
    auto mask_a = [&](double (&a)[4], int k0, int t) {      // ... 0 outside the band, the matrix and the block
        const int q = 16 * t + l15, r = row_at(k0, SB + q);
        const int nb = min(SB, n - k0);
        const bool ok = q < bw && r >= 0 && r < n;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[kk] = (ok && 4 * kk + l4 < nb) ? a[kk] : 0.0;
    };
    const int kfirst = UPPER ? (nblk - 1) * SB : 0;
    for (int idx = tid; idx < (bw + SB) * NB; idx += TW_NT) {      // the first window: positions 0 .. bw + SB - 1
        const int p = idx / NB, r = row_at(kfirst, p);
        win[(idx % NB) * WR + p] = (r >= 0 && r < n) ? Xb[(size_t)r * NB + (idx % NB)] : 0.0;
    }
    double td_next = load_td(kfirst);
    double an[TW_TPW ? TW_TPW : 1][4];
#pragma unroll
    for (int u = 0; u < TW_TPW; ++u)
        if (wv + 16 * u < ntile) load_a(an[u], kfirst, wv + 16 * u);
    int wbase = 0;
    for (int bb = 0; bb < nblk; ++bb) {
        const int k0 = UPPER ? (nblk - 1 - bb) * SB : bb * SB;
        const int nb = min(SB, n - k0);
        const int kn = UPPER ? k0 - SB : k0 + SB;
        const bool more = bb + 1 < nblk;
        if (tid < SB * SB) Td[ti][tj] = mask_td(k0, td_next);
        __syncthreads();
        if (more) td_next = load_td(kn);
        double enter = 0.0;
        const int pe = bw + SB + (tid / NB);              // position (relative to THIS block) of the rows that enter
        const int re = row_at(k0, pe);
        if (more && tid < SB * NB) enter = Xb[(size_t)min(max(re, 0), n - 1) * NB + (tid % NB)];      // (raw)
        if (tid < SB * NB) {      // y = T11^-1 x
            const int c = tid / NB, jj = tid % NB;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int pi = UPPER ? SB - 1 - i : i;      // position of block row i
                int sl = wbase + pi;
                if (sl >= WR) sl -= WR;
                acc = fma(Td[c][i], win[jj * WR + sl], acc);
            }
            ys[c][jj] = (c < nb) ? acc : 0.0;
            if (c < nb) Xb[(size_t)(k0 + c) * NB + jj] = acc;
        }
        __syncthreads();
        double bq[4];      // B operand: -y(4 kk + l4, l15), zero in the eight unused columns
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bq[kk] = l15 < NB ? -ys[4 * kk + l4][l15 % NB] : 0.0;
        auto do_tile = [&](int t, double (&a)[4]) {      // 16 rows after (before) the block
            mask_a(a, k0, t);
            ss_v4d c;
            int sl[4];
            bool ok[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int q = 16 * t + l4 + 4 * reg, r = row_at(k0, SB + q);
                ok[reg] = l15 < NB && q < bw && r >= 0 && r < n;
                sl[reg] = wbase + SB + q;
                if (sl[reg] >= WR) sl[reg] -= WR;
                c[reg] = ok[reg] ? win[(l15 % NB) * WR + sl[reg]] : 0.0;
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bq[kk], c, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (ok[reg]) win[(l15 % NB) * WR + sl[reg]] = c[reg];
        };
#pragma unroll
        for (int u = 0; u < TW_TPW; ++u)
            if (wv + 16 * u < ntile) do_tile(wv + 16 * u, an[u]);
#pragma unroll 1
        for (int t = wv + 16 * TW_TPW; t < ntile; t += 16) {      // (bands beyond 1 024 rows: entries read where they are used)
            double a[4];
            load_a(a, k0, t);
            do_tile(t, a);
        }
        if (more) {      // the coming step's factor entries: in flight over the barriers and the block solve
#pragma unroll
            for (int u = 0; u < TW_TPW; ++u)
                if (wv + 16 * u < ntile) load_a(an[u], kn, wv + 16 * u);
        }
        if (more && tid < SB * NB) {
            int se = wbase + pe;
            if (se >= WR) se -= WR;
            win[(tid % NB) * WR + se] = (re >= 0 && re < n) ? enter : 0.0;
        }
        wbase += SB;
        if (wbase >= WR) wbase -= WR;
        // (the next step's first barrier orders these LDS writes before its reads)
    }
}

// Both solves in one launch with the right-hand sides resident in LDS (row pitch 9 doubles:
// conflict-free for lanes = rows): the factor is then the only global traffic, read once per
// triangle.  For agglomerates up to ~2 000 rows.
constexpr int XLP = SS_B + 1;
template <int NT>
__global__ __launch_bounds__(NT) void ss_solve_lds_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                           const int64_t *__restrict__ voff,
                                                           const double *__restrict__ W, const double *__restrict__ X,
                                                           double *__restrict__ Zout, const int *__restrict__ state,
                                                           const int *__restrict__ bws,
        const int *__restrict__ active = nullptr) {
    extern __shared__ __align__(16) double xl[];      // [n][XLP]
    __shared__ double Td[SB][SB + 1];
    __shared__ double ys[SB][SS_B];
    const int b = active ? active[blockIdx.x] : (int)blockIdx.x, n = ns[b];     // (active: the matrices still iterating)
    if (state[b] & 3) return;                         // accepted (or given up) in an earlier iteration
    const int bw = bws ? bws[b] : n;                  // the factor is zero beyond this distance from the diagonal
    const double *A = W + moff[b];
    const double *Xb = X + voff[b] * SB;
    double *Zb = Zout + voff[b] * SB;
    const int tid = threadIdx.x;
    const int nblk = (n + SB - 1) / SB;
    const int ti = (tid >> 4) & 15, tj = tid & 15;
    for (int idx = tid; idx < n * SS_B; idx += NT) xl[(idx >> 3) * XLP + (idx & 7)] = Xb[idx];
    for (int pass = 0; pass < 2; ++pass) {
        const bool upper = pass == 1;
        auto load_td = [&](int k0) {
            const int nb = min(SB, n - k0);
            double v = 0.0;
            if (ti < nb && tj < nb && (upper ? tj >= ti : tj <= ti)) v = A[(size_t)(k0 + tj) * n + (k0 + ti)];
            return v;
        };
        double td_next = load_td(upper ? (nblk - 1) * SB : 0);
        for (int bb = 0; bb < nblk; ++bb) {
            const int k0 = upper ? (nblk - 1 - bb) * SB : bb * SB;
            const int nb = min(SB, n - k0);
            if (tid < SB * SB) Td[ti][tj] = td_next;
            __syncthreads();
            if (tid < SB * SS_B) {      // y = T11^-1 x (the stored block is the inverse)
                const int c = tid >> 3, jj = tid & 7;
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < SB; ++i) t = fma(Td[c][i], (i < nb) ? xl[(k0 + i) * XLP + jj] : 0.0, t);
                ys[c][jj] = (c < nb) ? t : 0.0;
            }
            if (bb + 1 < nblk) td_next = load_td(upper ? k0 - SB : k0 + SB);
            __syncthreads();
            if (tid < SB * SS_B) {
                const int c = tid >> 3, jj = tid & 7;
                if (c < nb) xl[(k0 + c) * XLP + jj] = ys[c][jj];
            }
            const int r_lo = upper ? max(0, k0 - bw) : k0 + nb, r_hi = upper ? k0 : min(n, k0 + nb + bw);
            for (int r = r_lo + tid; r < r_hi; r += NT) {
                double acc[SS_B];
#pragma unroll
                for (int j = 0; j < SS_B; ++j) acc[j] = xl[r * XLP + j];
                {
                    // the factor entries of this row from the other triangle (it holds the transpose): SB consecutive
                    // doubles per row.  Past a partial last block (backward pass only) the reads run into the
                    // next column of the matrix -- finite values that meet the zeros of ys.
                    const double *ap = A + (size_t)r * n + k0;
                    double t[SB];
#pragma unroll
                    for (int u = 0; u < SB; ++u) t[u] = ap[u];
#pragma unroll
                    for (int u = 0; u < SB; ++u)
#pragma unroll
                        for (int j = 0; j < SS_B; ++j) acc[j] = fma(-t[u], ys[u][j], acc[j]);
                }
#pragma unroll
                for (int j = 0; j < SS_B; ++j) xl[r * XLP + j] = acc[j];
            }
            __syncthreads();
        }
    }
    for (int idx = tid; idx < n * SS_B; idx += NT) Zb[idx] = xl[(idx >> 3) * XLP + (idx & 7)];
}

// The same for narrow bands (at most NT = 128 rows below a block): the kernel above is bound by the
// bytes it keeps in flight (measured: 0.44 ms per launch with the factor loads removed against 4 ms
// with them), so here the factor entries and the diagonal block of a step are requested more than
// two steps ahead: three rotating register buffers, each refilled as soon as its step has used it
// (the step loop is unrolled by three: no copy waits for a load) and the workgroup is two wavefronts, which lets five of them share a CU.
template <int NT>
__global__ __launch_bounds__(NT) void ss_solve_lds_pf_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                              const int64_t *__restrict__ voff,
                                                              const double *__restrict__ W, const double *__restrict__ X,
                                                              double *__restrict__ Zout, const int *__restrict__ state,
                                                              const int *__restrict__ bws,
        const int *__restrict__ active = nullptr) {
    static_assert(NT == 128, "two diagonal-block entries per thread");
    extern __shared__ __align__(16) double xl[];      // [n][XLP]
    __shared__ double Td[SB][SB + 1];
    __shared__ double ys[SB][SS_B];
    const int b = active ? active[blockIdx.x] : (int)blockIdx.x, n = ns[b];     // (active: the matrices still iterating)
    if (state[b] & 3) return;
    const int bw = bws ? bws[b] : n;
    const double *A = W + moff[b];
    const double *Xb = X + voff[b] * SB;
    double *Zb = Zout + voff[b] * SB;
    const int tid = threadIdx.x;
    const int nblk = (n + SB - 1) / SB, nstep = 2 * nblk;
    for (int idx = tid; idx < n * SS_B; idx += NT) xl[(idx >> 3) * XLP + (idx & 7)] = Xb[idx];
    auto step_k0 = [&](int st) { return st < nblk ? st * SB : (nstep - 1 - st) * SB; };
    // buf[0 .. SB): factor entries of this thread's row, buf[SB], buf[SB + 1]: its two entries of the block inverse
    auto prefetch = [&](int st, double (&buf)[SB + 2]) {
        if (st >= nstep) return;
        const bool upper = st >= nblk;
        const int k0 = step_k0(st), nb = min(SB, n - k0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ti = (tid >> 4) + 8 * h, tj = tid & 15;
            double v = 0.0;
            if (ti < nb && tj < nb && (upper ? tj >= ti : tj <= ti)) v = A[(size_t)(k0 + tj) * n + (k0 + ti)];
            buf[SB + h] = v;
        }
        const int r_lo = upper ? max(0, k0 - bw) : k0 + nb, r_hi = upper ? k0 : min(n, k0 + nb + bw);
        const int r = r_lo + tid;
        if (r < r_hi) {
            const double *ap = A + (size_t)k0 * n + r;
#pragma unroll
            for (int c = 0; c < SB; ++c) buf[c] = ap[(size_t)min(c, nb - 1) * n];
        }
    };
    auto step = [&](int st, double (&cur)[SB + 2]) {
        if (st >= nstep) return;                         // (block-uniform)
        const bool upper = st >= nblk;
        const int k0 = step_k0(st), nb = min(SB, n - k0);
        Td[tid >> 4][tid & 15] = cur[SB];
        Td[(tid >> 4) + 8][tid & 15] = cur[SB + 1];
        __syncthreads();
        {                           // y = T11^-1 x (the stored block is the inverse)
            const int c = tid >> 3, jj = tid & 7;
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < SB; ++i) t = fma(Td[c][i], (i < nb) ? xl[(k0 + i) * XLP + jj] : 0.0, t);
            ys[c][jj] = (c < nb) ? t : 0.0;
        }
        __syncthreads();
        {
            const int c = tid >> 3, jj = tid & 7;
            if (c < nb) xl[(k0 + c) * XLP + jj] = ys[c][jj];
        }
        const int r_lo = upper ? max(0, k0 - bw) : k0 + nb, r_hi = upper ? k0 : min(n, k0 + nb + bw);
        int r = r_lo + tid;
        if (r < r_hi) {
            double acc[SS_B];
#pragma unroll
            for (int j = 0; j < SS_B; ++j) acc[j] = xl[r * XLP + j];
#pragma unroll
            for (int u = 0; u < SB; ++u)
#pragma unroll
                for (int j = 0; j < SS_B; ++j) acc[j] = fma(-cur[u], ys[u][j], acc[j]);
#pragma unroll
            for (int j = 0; j < SS_B; ++j) xl[r * XLP + j] = acc[j];
        }
        for (r += NT; r < r_hi; r += NT) {               // (not reached when the host picked this kernel)
            double acc[SS_B];
#pragma unroll
            for (int j = 0; j < SS_B; ++j) acc[j] = xl[r * XLP + j];
            const double *ap = A + (size_t)k0 * n + r;
#pragma unroll 1
            for (int c = 0; c < SB; c += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ap[(size_t)min(c + u, nb - 1) * n];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < SS_B; ++j) acc[j] = fma(-t[u], ys[c + u][j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < SS_B; ++j) xl[r * XLP + j] = acc[j];
        }
        prefetch(st + 3, cur);                           // this buffer is free again
        __syncthreads();
    };
    double t0[SB + 2], t1[SB + 2], t2[SB + 2];
#pragma unroll
    for (int c = 0; c < SB + 2; ++c) t0[c] = t1[c] = t2[c] = 0.0;
    prefetch(0, t0);
    prefetch(1, t1);
    prefetch(2, t2);
    for (int st = 0; st < nstep; st += 3) {
        step(st, t0);
        step(st + 1, t1);
        step(st + 2, t2);
    }
    for (int idx = tid; idx < n * SS_B; idx += NT) Zb[idx] = xl[(idx >> 3) * XLP + (idx & 7)];
}

// Rayleigh-Ritz on span(Z) from M = Z^T X and G = Z^T Z, inverse residuals of the previous pairs,
// X <- Z C.  state[b]: bit 0 = converged (the wanted pairs and the first unwanted one), count in
// bits 8.., bit 1 = failure (too many wanted pairs / breakdown).  mu[b][SS_B] ascending.
template <int NB = 8>
__global__ __launch_bounds__(256) void ss_rr_kernel(const int *__restrict__ ns, const int64_t *__restrict__ voff,
                                                    double *__restrict__ X, const double *__restrict__ Z,
                                                    double *__restrict__ mu, int *__restrict__ state, int iter,
                                                    const double *__restrict__ sigmas, double vu, const int *__restrict__ inertia,
                                                    double *__restrict__ dbg = nullptr,
        const int *__restrict__ active = nullptr, double *__restrict__ hist = nullptr, int max_iter = 80,
        const int *__restrict__ reshift_ok = nullptr, const int *__restrict__ ndefl = nullptr,
        const int *__restrict__ it0 = nullptr, double SS_TOL = 1e-12) {
    // ndefl[b]: pairs of this matrix locked so far (the block is kept orthogonal to them); it0[b]: the iteration at which
    // its block was last started (0, or the lock): convergence is judged from the second iteration after that
    if (it0) iter -= it0[active ? active[blockIdx.x] : (int)blockIdx.x];
    __shared__ double part[4][2 * NB + 1][NB];  // [wavefront][M rows | G rows | residual][column j]
    __shared__ double Ms[NB][NB], Gs[NB][NB], Cs[NB][NB], res2[NB], mus[NB], mu_old[NB];
    const int b = active ? active[blockIdx.x] : (int)blockIdx.x, n = ns[b];     // (active: the matrices still iterating)
    if (state[b] & 3) return;                       // accepted earlier: X, mu stay as they are
    const double sigma = sigmas[b];
    double *Xb = X + voff[b] * SB;
    const double *Zb = Z + voff[b] * SB;
    const int tid = threadIdx.x;
    constexpr int NG = 256 / NB;                    // row groups
    const int j = tid % NB, grp = tid / NB;         // column j, row group
    if (tid < NB) mu_old[tid] = (iter > 0) ? mu[(size_t)b * NB + tid] : 1.0;
    __syncthreads();
    {
        double am[NB], ag[NB], rs = 0.0;
#pragma unroll
        for (int i = 0; i < NB; ++i) { am[i] = 0.0; ag[i] = 0.0; }
        const double th = 1.0 / mu_old[j];
                // two rows per trip: their loads are independent (four would cost a resident workgroup: 164 VGPRs)
        for (int r0 = grp; r0 < n; r0 += 2 * NG) {
            double xj[2], zjv[2], zr[2][NB];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = min(r0 + NG * u, n - 1);
                xj[u] = Xb[(size_t)r * NB + j];
                zjv[u] = Zb[(size_t)r * NB + j];
#pragma unroll
                for (int i = 0; i < NB; ++i) zr[u][i] = Zb[(size_t)r * NB + i];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (r0 + NG * u < n) {
                    const double zj = zjv[u];
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        am[i] = fma(zr[u][i], xj[u], am[i]);
                        ag[i] = fma(zr[u][i], zj, ag[i]);
                    }
                    const double d = zj - th * xj[u];
                    rs = fma(d, d, rs);
                }
            }
        }
        // the 8 row groups of a wavefront are summed in registers (lane bits 3..5), the 4 wavefronts in LDS:
        // a small footprint lets 8 workgroups share a CU, which is what hides the serial part below
#pragma unroll
        for (int o = NB; o < 64; o <<= 1) {
#pragma unroll
            for (int i = 0; i < NB; ++i) { am[i] += __shfl_xor(am[i], o, 64); ag[i] += __shfl_xor(ag[i], o, 64); }
            rs += __shfl_xor(rs, o, 64);
        }
        if ((tid & 63) < NB) {
            const int w = tid >> 6;
#pragma unroll
            for (int i = 0; i < NB; ++i) { part[w][i][j] = am[i]; part[w][NB + i][j] = ag[i]; }
            part[w][2 * NB][j] = rs;
        }
        __syncthreads();
        for (int idx = tid; idx < (2 * NB + 1) * NB; idx += 256) {
            const int i = idx / NB, jj = idx % NB;
            const double sum = (part[0][i][jj] + part[1][i][jj]) + (part[2][i][jj] + part[3][i][jj]);
            if (i < NB) Ms[i][jj] = sum; else if (i < 2 * NB) Gs[i - NB][jj] = sum; else res2[jj] = sum;
        }
        __syncthreads();
    }
    __shared__ double R[NB][NB + 1], S[NB][NB + 1], V[NB][NB + 1];
    __shared__ double rot_c[NB / 2], rot_s[NB / 2], rdi[NB];     // (rdi: reciprocal diagonal of R)
    __shared__ int sh_st, sh_ok;
    if (tid == 0) {
        int st = 0;
        // convergence of the PREVIOUS pairs (X, mu_old): || C x - lambda x || <= || C - sigma || mu || z - x / mu ||
        if (iter > 0) {
            int k = 0;
            for (int q = 0; q < NB; ++q) if (sigma + mu_old[q] <= vu) ++k;
            if (dbg)        // (SAAMGE_AMD_SS_DEBUG: residual bounds and Ritz values of the previous pairs)
                for (int q = 0; q < NB; ++q) { dbg[(size_t)b * 2 * NB + q] = 2.5 * mu_old[q] * sqrt(res2[q]); dbg[(size_t)b * 2 * NB + NB + q] = sigma + mu_old[q]; }
            const int nd = ndefl ? ndefl[b] : 0;
            const int cert0 = inertia ? inertia[b] : -2;     // certified #{lambda < vu}; -1: not certifiable, -2: none
            const int cert = cert0 > 0 ? cert0 - nd : cert0; // ... of which nd are locked already
            const bool partial = cert > NB - 2;            // more wanted pairs than the block holds: lock the first six
            // hopeless convergence (the wanted pair far above the shift inside a cluster: rate ~ 1): the bound of
            // the slowest wanted pair four iterations ago predicts the iterations still needed; a matrix that
            // cannot make it within the budget gives up now instead of after max_iter iterations
            if (hist && iter >= 1) {
                double worst = 0.0;
                for (int q = 0; q < (partial ? SS_LOCK : max(k, 1)); ++q) worst = fmax(worst, 2.5 * mu_old[q] * sqrt(res2[q]));
                const double old = hist[(size_t)b * 4 + (iter & 3)];
                hist[(size_t)b * 4 + (iter & 3)] = worst;
                if (iter >= 6 && worst > SS_TOL && old > 0.0) {
                    const double r4 = worst / old;                        // decay over four iterations
                    const double need = (r4 < 1.0) ? 4.0 * log(SS_TOL / worst) / log(r4) : 1e30;
                    // state bit 2: the host may move the shift to just below the smallest Ritz value and factor again
                    // (certified count 0: the wanted pair is the smallest one; once per matrix)
                    if (reshift_ok && reshift_ok[b] && cert == 0 && need > 10.0) st |= 4;
                    else if (iter >= 12 && need > (double)(max_iter - iter)) st |= 2;
                }
            }
            if (partial && cert0 <= SS_WANT_MAX) {
                // the six smallest Ritz pairs of the block converged (they lie below the window's end: more than six
                // eigenvalues do): state bit 3 asks the host to lock them
                bool ok = k >= SS_LOCK;
                for (int q = 0; q < SS_LOCK; ++q) ok = ok && (2.5 * mu_old[q] * sqrt(res2[q]) <= SS_TOL);
                if (ok) st |= 8;
            } else if (k > NB - 2 || cert > NB - 2 || cert == -1) st |= 2;
            else if (cert >= 0) {
                // Ritz values approach the eigenvalues from above, so the number inside the window grows to the
                // certified count: accept once it is reached and those pairs have converged (count 0: the
                // smallest pair alone, the reference's "at least one" rule)
                bool ok = k == cert;
                for (int q = 0; q < max(k, 1); ++q) ok = ok && (2.5 * mu_old[q] * sqrt(res2[q]) <= SS_TOL);
                if (ok) st |= 1 | (k << 4) | (max(k, 1) << 8);
            } else {
                // no certificate (SAAMGE_AMD_SS_CERTIFY=0): residual bounds of the wanted pairs; the first unwanted
                // one has to be pinned above the window (an eigenvalue lies within the bound of its Ritz value)
                bool ok = true;
                for (int q = 0; q < max(k, 1); ++q) ok = ok && (2.5 * mu_old[q] * sqrt(res2[q]) <= SS_TOL);
                if (k >= 1) ok = ok && (2.5 * mu_old[k] * sqrt(res2[k]) < 0.5 * (sigma + mu_old[k] - vu));
                if (ok) st |= 1 | (k << 4) | (max(k, 1) << 8);
            }
        }
        sh_st = st;
        sh_ok = 1;
        if (!(st & 3)) {
            // G = R^T R (upper factor, row by row): the only serial piece (~100 flops)
            for (int c = 0; c < NB; ++c)
                for (int r2 = 0; r2 < NB; ++r2) R[r2][c] = 0.0;
            for (int c = 0; c < NB; ++c) {
                double d = Gs[c][c];
                for (int q = 0; q < c; ++q) d -= R[q][c] * R[q][c];
                if (!(d > 0.0)) { sh_ok = 0; d = 1.0; }
                const double ri = fast_rsqrt(d);      // (division and square-root chains: hardware seed + Newton)
                R[c][c] = d * ri;
                rdi[c] = ri;
                for (int c2 = c + 1; c2 < NB; ++c2) {
                    double t = 0.5 * (Gs[c][c2] + Gs[c2][c]);
                    for (int q = 0; q < c; ++q) t -= R[q][c] * R[q][c2];
                    R[c][c2] = t * ri;
                }
            }
        }
    }
    __syncthreads();
    if (NB != 8 && !(sh_st & 3)) {
        // ---- NB = 16: the same with all 256 threads and workgroup barriers (16 x 16 entries: one per thread; eight disjoint
        // rotations per step, fifteen steps per sweep) ----
        const int lr = tid / NB, lc = tid % NB;
        V[lr][lc] = (lr == lc) ? 1.0 : 0.0;
        if (tid < NB) {                     // column c of T1 = R^-T Msym: forward substitution
            const int c = tid;
            for (int r2 = 0; r2 < NB; ++r2) {
                double t = 0.5 * (Ms[r2][c] + Ms[c][r2]);
                for (int q = 0; q < r2; ++q) t -= R[q][r2] * S[q][c];
                S[r2][c] = t * rdi[r2];
            }
        }
        __syncthreads();
        if (tid < NB) {                     // row r of S = T1 R^-1
            const int r2 = tid;
            for (int c = 0; c < NB; ++c) {
                double t = S[r2][c];
                for (int q = 0; q < c; ++q) t -= S[r2][q] * R[q][c];
                S[r2][c] = t * rdi[c];
            }
        }
        __syncthreads();
        {
            const double t = 0.5 * (S[lr][lc] + S[lc][lr]);
            __syncthreads();
            S[lr][lc] = t;
        }
        __syncthreads();
        for (int sweep = 0; sweep < 16; ++sweep) {
            double off = (lr != lc) ? S[lr][lc] * S[lr][lc] : 0.0, dg = (lr == lc) ? S[lr][lc] * S[lr][lc] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
            if ((tid & 63) == 0) { part[0][0][tid >> 6] = off; part[0][1][tid >> 6] = dg; }
            __syncthreads();
            off = (part[0][0][0] + part[0][0][1]) + (part[0][0][2] + part[0][0][3]);
            dg = (part[0][1][0] + part[0][1][1]) + (part[0][1][2] + part[0][1][3]);
            __syncthreads();
            if (off <= 1e-34 * dg) break;      // (uniform)
            for (int step = 0; step < NB - 1; ++step) {
                // round-robin pairing of NB players: the last one fixed, the others rotate
                auto player = [&](int slot) { return slot == NB - 1 ? NB - 1 : (slot + step) % (NB - 1); };
                if (tid < NB / 2) {
                    int p2 = player(tid), q2 = player(NB - 1 - tid);
                    if (p2 > q2) { const int t = p2; p2 = q2; q2 = t; }
                    double cc = 1.0, sn = 0.0;
                    const double apq = S[p2][q2];
                    if (apq != 0.0) {
                        const double d = S[q2][q2] - S[p2][p2];
                        const double den = fabs(d) + fast_sqrt(fma(d, d, 4.0 * apq * apq));
                        const double t = ((d >= 0.0) == (apq >= 0.0) ? 2.0 : -2.0) * fabs(apq) * fast_rcp(den);
                        cc = fast_rsqrt(fma(t, t, 1.0));
                        sn = t * cc;
                    }
                    rot_c[tid] = cc;
                    rot_s[tid] = sn;
                }
                __syncthreads();
                const int m2 = tid % (NB / 2), k2 = (tid / (NB / 2)) % NB;      // pair m2, row / column k2; threads 0..127 S, 128..255 V
                int p2 = player(m2), q2 = player(NB - 1 - m2);
                if (p2 > q2) { const int t = p2; p2 = q2; q2 = t; }
                const double cc = rot_c[m2], sn = rot_s[m2];
                {   // columns p, q of S and of V (the entries (k2, p2), (k2, q2) belong to this thread alone)
                    double (*Mx)[NB + 1] = (tid < NB * NB / 2) ? S : V;
                    const double a = Mx[k2][p2], bq = Mx[k2][q2];
                    Mx[k2][p2] = cc * a - sn * bq;
                    Mx[k2][q2] = sn * a + cc * bq;
                }
                __syncthreads();
                if (tid < NB * NB / 2) {   // rows p, q of S
                    const double a = S[p2][k2], bq = S[q2][k2];
                    S[p2][k2] = cc * a - sn * bq;
                    S[q2][k2] = sn * a + cc * bq;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        if (tid == 0) {
            int order[NB];
            for (int q = 0; q < NB; ++q) order[q] = q;
            for (int a2 = 0; a2 < NB; ++a2)
                for (int b2 = a2 + 1; b2 < NB; ++b2)
                    if (S[order[b2]][order[b2]] < S[order[a2]][order[a2]]) { const int t = order[a2]; order[a2] = order[b2]; order[b2] = t; }
            for (int q = 0; q < NB; ++q) {
                mus[q] = S[order[q]][order[q]];
                if (!(mus[q] > 0.0)) sh_ok = 0;
                res2[q] = (double)order[q];      // (res2 is free now: source column of output q)
            }
        }
        __syncthreads();
        if (tid < NB) {      // column q of C = R^-1 V(:, src)
            const int q = tid, src = (int)res2[q];
            for (int r2 = NB - 1; r2 >= 0; --r2) {
                double t = V[r2][src];
                for (int c2 = r2 + 1; c2 < NB; ++c2) t -= R[r2][c2] * Cs[c2][q];
                Cs[r2][q] = t * rdi[r2];
            }
        }
    }
    if (NB == 8 && !(sh_st & 3) && tid < 64) {
        // ---- S = R^-T Msym R^-1 and its eigen-decomposition by parallel-order Jacobi, one wavefront ----
        const int lr = tid >> 3, lc = tid & 7;
        auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
        V[lr][lc] = (lr == lc) ? 1.0 : 0.0;
        if (tid < NB) {                     // column c of T1 = R^-T Msym: forward substitution
            const int c = tid;
            for (int r2 = 0; r2 < NB; ++r2) {
                double t = 0.5 * (Ms[r2][c] + Ms[c][r2]);
                for (int q = 0; q < r2; ++q) t -= R[q][r2] * S[q][c];
                S[r2][c] = t * rdi[r2];
            }
        }
        wsync();
        if (tid < NB) {                     // row r of S = T1 R^-1
            const int r2 = tid;
            for (int c = 0; c < NB; ++c) {
                double t = S[r2][c];
                for (int q = 0; q < c; ++q) t -= S[r2][q] * R[q][c];
                S[r2][c] = t * rdi[c];
            }
        }
        wsync();
        {
            const double t = 0.5 * (S[lr][lc] + S[lc][lr]);
            wsync();
            S[lr][lc] = t;
        }
        wsync();
        for (int sweep = 0; sweep < 12; ++sweep) {
            double off = (lr != lc) ? S[lr][lc] * S[lr][lc] : 0.0, dg = (lr == lc) ? S[lr][lc] * S[lr][lc] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
            if (off <= 1e-34 * dg) break;
            for (int step = 0; step < 7; ++step) {
                // round-robin pairing of 8 players: player 7 fixed, the others rotate
                auto player = [&](int slot) { return slot == 7 ? 7 : (slot + step) % 7; };
                if (tid < 4) {
                    int p2 = player(tid), q2 = player(7 - tid);
                    if (p2 > q2) { const int t = p2; p2 = q2; q2 = t; }
                    double cc = 1.0, sn = 0.0;
                    const double apq = S[p2][q2];
                    if (apq != 0.0) {
                        // t = sign(tau) / (|tau| + sqrt(1 + tau^2)), tau = d / (2 apq), without the division by apq
                        const double d = S[q2][q2] - S[p2][p2];
                        const double den = fabs(d) + fast_sqrt(fma(d, d, 4.0 * apq * apq));
                        const double t = ((d >= 0.0) == (apq >= 0.0) ? 2.0 : -2.0) * fabs(apq) * fast_rcp(den);
                        cc = fast_rsqrt(fma(t, t, 1.0));
                        sn = t * cc;
                    }
                    rot_c[tid] = cc;
                    rot_s[tid] = sn;
                }
                wsync();
                const int m2 = tid & 3, k2 = (tid >> 2) & 7;      // pair m2, row / column k2; lanes 0..31 S, 32..63 V
                int p2 = player(m2), q2 = player(7 - m2);
                if (p2 > q2) { const int t = p2; p2 = q2; q2 = t; }
                const double cc = rot_c[m2], sn = rot_s[m2];
                {   // columns p, q of S (lanes < 32) and of V (lanes >= 32)
                    double (*Mx)[NB + 1] = (tid < 32) ? S : V;
                    const double a = Mx[k2][p2], bq = Mx[k2][q2];
                    wsync();
                    Mx[k2][p2] = cc * a - sn * bq;
                    Mx[k2][q2] = sn * a + cc * bq;
                }
                wsync();
                if (tid < 32) {   // rows p, q of S
                    const double a = S[p2][k2], bq = S[q2][k2];
                    S[p2][k2] = cc * a - sn * bq;
                    S[q2][k2] = sn * a + cc * bq;
                }
                wsync();
            }
        }
        wsync();
        if (tid == 0) {
            int order[NB];
            for (int q = 0; q < NB; ++q) order[q] = q;
            for (int a2 = 0; a2 < NB; ++a2)
                for (int b2 = a2 + 1; b2 < NB; ++b2)
                    if (S[order[b2]][order[b2]] < S[order[a2]][order[a2]]) { const int t = order[a2]; order[a2] = order[b2]; order[b2] = t; }
            for (int q = 0; q < NB; ++q) {
                mus[q] = S[order[q]][order[q]];
                if (!(mus[q] > 0.0)) sh_ok = 0;
                rot_c[0] = 0.0;
                res2[q] = (double)order[q];      // (res2 is free now: source column of output q)
            }
        }
        wsync();
        if (tid < NB) {      // column q of C = R^-1 V(:, src)
            const int q = tid, src = (int)res2[q];
            for (int r2 = NB - 1; r2 >= 0; --r2) {
                double t = V[r2][src];
                for (int c2 = r2 + 1; c2 < NB; ++c2) t -= R[r2][c2] * Cs[c2][q];
                Cs[r2][q] = t * rdi[r2];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int st = sh_st;
        if (!(st & 3) && !sh_ok) st |= 2;
        state[b] = st;
        Ms[0][0] = (double)st;
    }
    __syncthreads();
    const int st = (int)Ms[0][0];
    if (st & 3) return;            // converged (X, mu stay the accepted pairs) or failed
    if (tid < NB) mu[(size_t)b * NB + tid] = mus[tid];
    {   // X = Z C: one output entry per thread and trip (its column of C in NB registers, coalesced stores)
        const int q = tid % NB;
        double cq[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) cq[i] = Cs[i][q];
        for (int r = tid / NB; r < n; r += NG) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < NB; ++i) t = fma(Zb[(size_t)r * NB + i], cq[i], t);
            Xb[(size_t)r * NB + q] = t;
        }
    }
}

// Z <- Z - V (V^T Z) for the matrices with locked pairs V (orthonormal: converged Ritz vectors of one block): the block
// stays in the complement the remaining wanted pairs live in.  One workgroup per matrix, fixed summation order.
__global__ __launch_bounds__(256) void ss_deflate_kernel(const int *__restrict__ ns, const int64_t *__restrict__ voff,
                                                         const double *__restrict__ Vl, const int *__restrict__ ndefl,
                                                         double *__restrict__ Z, const int *__restrict__ state,
                                                         const int *__restrict__ active) {
    const int b = active ? active[blockIdx.x] : (int)blockIdx.x, n = ns[b], nd = ndefl[b];
    if (nd == 0 || (state[b] & 3)) return;
    __shared__ double part[4][SS_LOCK_PITCH][SS_B], proj[SS_LOCK_PITCH][SS_B];
    double *Zb = Z + voff[b] * SB;
    const double *Vb = Vl + voff[b] * SS_LOCK_PITCH;
    const int tid = threadIdx.x, j = tid & 7, grp = tid >> 3;
    double acc[SS_LOCK_PITCH];
#pragma unroll
    for (int v = 0; v < SS_LOCK_PITCH; ++v) acc[v] = 0.0;
    for (int r = grp; r < n; r += 32) {
        const double z = Zb[(size_t)r * SS_B + j];
#pragma unroll
        for (int v = 0; v < SS_LOCK_PITCH; ++v) acc[v] = fma(Vb[(size_t)r * SS_LOCK_PITCH + v], z, acc[v]);
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int v = 0; v < SS_LOCK_PITCH; ++v) acc[v] += __shfl_xor(acc[v], o, 64);
    if ((tid & 63) < SS_B)
#pragma unroll
        for (int v = 0; v < SS_LOCK_PITCH; ++v) part[tid >> 6][v][j] = acc[v];
    __syncthreads();
    if (tid < SS_LOCK_PITCH * SS_B) {
        const int v = tid >> 3, jj = tid & 7;
        proj[v][jj] = v < nd ? (part[0][v][jj] + part[1][v][jj]) + (part[2][v][jj] + part[3][v][jj]) : 0.0;
    }
    __syncthreads();
    for (int r = grp; r < n; r += 32) {
        double z = Zb[(size_t)r * SS_B + j];
#pragma unroll
        for (int v = 0; v < SS_LOCK_PITCH; ++v) z = fma(-Vb[(size_t)r * SS_LOCK_PITCH + v], proj[v][j], z);
        Zb[(size_t)r * SS_B + j] = z;
    }
}
// Lock the six smallest Ritz pairs of the listed matrices (state bit 3): vectors to Vl, values to lock_mu.  The block's
// last two pairs -- the next wanted ones, usually as good as converged by then -- move to its first two columns, the
// other six start again from pseudo-random vectors (the next Rayleigh-Ritz step sees them after the deflation); the
// convergence history is reset.
__global__ __launch_bounds__(256) void ss_lock_kernel(const int *__restrict__ list, const int *__restrict__ ns,
                                                      const int64_t *__restrict__ voff, double *__restrict__ X,
                                                      double *__restrict__ mu, const double *__restrict__ sigmas,
                                                      double *__restrict__ Vl, double *__restrict__ lock_mu,
                                                      int *__restrict__ ndefl, int *__restrict__ it0, int iter,
                                                      int *__restrict__ state, double *__restrict__ hist) {
    const int b = list[blockIdx.x], n = ns[b];
    double *Xb = X + voff[b] * SB;
    double *Vb = Vl + voff[b] * SS_LOCK_PITCH;
    for (int r = threadIdx.x; r < n; r += 256) {
        double x[SS_B];
#pragma unroll
        for (int q = 0; q < SS_B; ++q) x[q] = Xb[(size_t)r * SS_B + q];
#pragma unroll
        for (int q = 0; q < SS_LOCK_PITCH; ++q) Vb[(size_t)r * SS_LOCK_PITCH + q] = q < SS_LOCK ? x[q] : 0.0;
#pragma unroll
        for (int q = 0; q < SS_B; ++q)
            Xb[(size_t)r * SS_B + q] = q < SS_B - SS_LOCK ? x[SS_LOCK + q] : unit_rand_ss((unsigned)(r * SS_B + q) + 0x9E3779B9u, (unsigned)n + 17u);
    }
    __shared__ double m_old[SS_B];
    if (threadIdx.x < SS_B) m_old[threadIdx.x] = mu[(size_t)b * SS_B + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < SS_LOCK) lock_mu[(size_t)b * SS_LOCK_PITCH + threadIdx.x] = sigmas[b] + m_old[threadIdx.x];
    if (threadIdx.x < SS_B) mu[(size_t)b * SS_B + threadIdx.x] = threadIdx.x < SS_B - SS_LOCK ? m_old[SS_LOCK + threadIdx.x] : 1.0;
    if (threadIdx.x < 4) hist[(size_t)b * 4 + threadIdx.x] = 0.0;
    if (threadIdx.x == 0) { ndefl[b] = SS_LOCK; it0[b] = iter + 1; state[b] = 0; }
}

template <int NB = 8>
__global__ __launch_bounds__(256) void ss_output_kernel(const int *__restrict__ ns, const int64_t *__restrict__ voff,
                                                        const double *__restrict__ X, const double *__restrict__ mu,
                                                        const double *__restrict__ dis, const short *__restrict__ perm,
                                                        const int *__restrict__ ms,
                                                        const int64_t *__restrict__ eoff, const int64_t *__restrict__ xoff,
                                                        double *__restrict__ evals, double *__restrict__ evecs,
                                                        const double *__restrict__ sigmas,
                                                        const int *__restrict__ ndefl = nullptr, const double *__restrict__ Vl = nullptr,
                                                        const double *__restrict__ lock_mu = nullptr) {
    // (locked pairs first: they are the smaller ones)
    const int b = blockIdx.x, n = ns[b], m = ms[b], nd = ndefl ? min(ndefl[b], m) : 0;
    const double sigma = sigmas[b];
    const double *Xb = X + voff[b] * SB;
    const double *Vb = Vl ? Vl + voff[b] * SS_LOCK_PITCH : nullptr;
    const short *pm = perm ? perm + voff[b] : nullptr;
    for (int q = threadIdx.x; q < m; q += 256)
        evals[eoff[b] + q] = q < nd ? lock_mu[(size_t)b * SS_LOCK_PITCH + q] : sigma + mu[(size_t)b * NB + (q - nd)];
    for (int idx = threadIdx.x; idx < n * m; idx += 256) {
        const int r = idx % n, q = idx / n, pr = pm ? pm[r] : r;
        evecs[xoff[b] + (size_t)q * n + r] = dis[voff[b] + r] * (q < nd ? Vb[(size_t)pr * SS_LOCK_PITCH + q] : Xb[(size_t)pr * NB + (q - nd)]);
    }
}

// after a matrix has been factored again at a new shift: its Ritz values are kept as lambda - sigma
template <int NB = 8>
__global__ void ss_reshift_mu_kernel(int nreq, const int *__restrict__ req, const double *__restrict__ delta,
                                     double *__restrict__ mu, double *__restrict__ hist) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nreq * NB) return;
    const int b = req[t / NB], q = t % NB;
    mu[(size_t)b * NB + q] += delta[t / NB];
    if (q < 4) hist[(size_t)b * 4 + q] = 0.0;
}

// ---------------------------------------------------------------------------------------
// The same factorisation with the block steps PIPELINED (round 4)
// ---------------------------------------------------------------------------------------
// The kernel of rounds 2-3 (chol_band_lds_kernel, git history) ran four stages per block step one after the other, each behind a barrier: block fill,
// Cholesky + inverse of the 16 x 16 block by one wavefront (3.3 us), the panel rows by <= 52 threads (136 dependent
// multiply-adds with LDS operands each), the rank-16 update of the window by <= 91 threads -- 17 us per step with four
// workgroups on a CU, 26 steps per 405-row agglomerate, 34 ms per setup of the 256^3 problem for the inertia pass alone.
// Here the panel and the update run on the matrix cores (v_mfma_f64_16x16x4: a tile of 16 rows per wavefront, lane roles as
// in chol_panel_ll_kernel), and the step is split so that only the block factorisation is left on the critical path:
//   phase A  every wavefront: its tiles of L21 = A21 L11^-T (S) -> LDS (+ the factor, to memory)
//   phase B  wavefront 0: the update of the NEXT diagonal block alone, then its Cholesky + inverse;
//            wavefronts 1-3: the other tiles of the update and the 16 rows that enter the window
// with two barriers per step.
constexpr int BC2_LP = 18;      // pitch of a row of L21 in LDS (16-byte aligned rows, conflict-free 16-byte reads)
constexpr size_t bc2_lds_bytes(int win) {      // window, L21 (band rows x 18), the two block buffers, two sets of signs
    return sizeof(double) * ((size_t)win * bc_pitch(win) + (size_t)(win - SB) * BC2_LP + 2 * SB * (SB + 1) + 2 * SB) + 16;
}
template <int BC_WIN, bool INERTIA>
__global__ __launch_bounds__(BC_NT) void chol_band_lds2_kernel(const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                               double *__restrict__ W, const int *__restrict__ bws,
                                                               int *__restrict__ info, double shift = 0.0,
                                                               const int *__restrict__ active = nullptr) {
    constexpr int BC_MAXBW = BC_WIN - SB, BC_P = bc_pitch(BC_WIN), LPR = BC_MAXBW;
    extern __shared__ __align__(16) double bc_lds[];
    double *S = bc_lds;                                  // [BC_WIN][BC_P]
    double *Lp = S + BC_WIN * BC_P;                      // [LPR][BC2_LP]: L21 of the current block, row-major
    double (*Ld)[SB + 1] = (double (*)[SB + 1])(Lp + LPR * BC2_LP);
    double (*Li)[SB + 1] = Ld + SB;
    double *sg = (double *)(Li + SB);                    // [2][SB]: signs of the pivots of the current / the next block
    int *bad = (int *)(sg + 2 * SB);
    const int b = active ? active[blockIdx.x] : (int)blockIdx.x, n = ns[b], bw = bws[b];
    if (bw > BC_MAXBW) return;
    double *A = W + moff[b];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    auto sl = [&](int i, int j) -> double & { return S[(i % BC_WIN) * BC_P + (j - i + BC_MAXBW)]; };     // (0 <= i - j <= BC_MAXBW)
    auto fetch_rows = [&](int i0, int i1) {
        for (int i = i0 + w; i < min(i1, n); i += BC_NT / 64)
            for (int jj = lane; jj <= bw; jj += 64) {
                const int j = i - bw + jj;
                if (j >= 0) sl(i, j) = A[(size_t)i * n + j] - ((INERTIA && j == i) ? shift : 0.0);
            }
    };
    // Cholesky + inverse of the block in Ld (wavefront 0), its signs to sg[which], its inverse to the matrix
    auto factor_block = [&](int k0, int which) {
        const int r = chol16_inverse_wave<INERTIA>(Ld, Li, lane, sg + SB * which);
        if (INERTIA) { if (lane == 0) { if (r < 0) bad[0] = 1; else bad[1] += r; } }
        else if (lane == 0 && r) bad[0] = 1;
        if (!INERTIA) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int nbk = min(SB, n - k0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = l15, j = l4 + 4 * reg;
                if (i < nbk && j <= i) {
                    const double v = Li[i][j];
                    A[(size_t)(k0 + j) * n + (k0 + i)] = v;
                    A[(size_t)(k0 + i) * n + (k0 + j)] = v;
                }
            }
        }
    };
    if (tid == 0) { bad[0] = 0; bad[1] = 0; }
    fetch_rows(0, SB + bw);
    __syncthreads();
    {
        const int nb0 = min(SB, n);
        if (tid < SB * SB) {
            const int i = tid >> 4, j = tid & 15;
            Ld[i][j] = (i < nb0 && j <= i && i - j <= bw) ? sl(i, j) : ((i == j) ? 1.0 : 0.0);
        }
        __syncthreads();
        if (w == 0) factor_block(0, 0);
        __syncthreads();
    }
    int pb = 0;
    for (int k0 = 0; k0 + SB < n; k0 += SB, pb ^= 1) {      // (a last, partial block has nothing below it)
        const int rend = min(n, k0 + SB + bw), m = rend - (k0 + SB);     // rows below the block inside the band (>= 1)
        const int T16 = (m + 15) >> 4;
        const int f0 = k0 + SB + bw, bw1 = bw + 1;      // the 16 rows that enter the window in this step
        const double *sgk = sg + SB * pb;
        // ---- phase A: tiles w, w + 4, ... of the panel ----
        for (int t = w; t < T16; t += BC_NT / 64) {
            const int q = 16 * t + l15;                   // row of the panel
            const int r = k0 + SB + q;
            const bool live = q < m;
            ss_v4d x;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int c = l4 + 4 * reg;
                x[reg] = (live && r - (k0 + c) <= bw) ? sl(r, k0 + c) : 0.0;
            }
            ss_v4d y = ss_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[l15][l4 + 4 * s], x[s], y, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int c = l4 + 4 * reg;
                const double v = INERTIA ? y[reg] * sgk[c] : y[reg];      // L21 = A21 L11^-T (S)
                if (live) {
                    Lp[q * BC2_LP + c] = v;
                    if (!INERTIA) {
                        A[(size_t)(k0 + c) * n + r] = v;
                        A[(size_t)r * n + (k0 + c)] = v;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase B ----
        // window tile (ti, tj) -= (L21 S)(rows of ti) L21(rows of tj)^T: element (row 16 ti + l15, column 16 tj + l4 + 4 reg)
        auto operand = [&](int t, bool signedrow, double (&o)[4]) {
            const int q = 16 * t + l15;
            const double *p = Lp + min(q, LPR - 1) * BC2_LP + 4 * l4;
            const double2 p0 = *(const double2 *)p, p1 = *(const double2 *)(p + 2);
            o[0] = p0.x; o[1] = p0.y; o[2] = p1.x; o[3] = p1.y;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (q >= m) o[s] = 0.0;
                else if (INERTIA && signedrow) o[s] *= -sgk[4 * l4 + s];
                else if (signedrow) o[s] = -o[s];
            }
        };
        auto update_tile = [&](int ti, int tj, ss_v4d &c, bool load) {
            double ob[4], oa[4];
            operand(ti, true, ob);
            operand(tj, false, oa);
            const int i = k0 + SB + 16 * ti + l15;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = k0 + SB + 16 * tj + l4 + 4 * reg;
                c[reg] = (load && i < rend && j <= i) ? sl(i, j) : 0.0;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[s], ob[s], c, 0, 0, 0);
        };
        // (bands narrower than a block: some rows of the next block enter the window only in this step -- its factorisation
        // then follows the step instead of running beside the update)
        const bool ahead = bw >= SB;
        if (w == 0 && !ahead) {
            ss_v4d c;
            update_tile(0, 0, c, true);
            const int i = k0 + SB + l15;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = k0 + SB + l4 + 4 * reg;
                if (i < rend && j <= i) sl(i, j) = c[reg];
            }
        } else if (w == 0) {
            ss_v4d c;
            update_tile(0, 0, c, true);
            const int nb1 = min(SB, n - (k0 + SB));
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = l15, j = l4 + 4 * reg;
                Ld[i][j] = (i < nb1 && j <= i && i - j <= bw) ? c[reg] : ((i == j) ? 1.0 : 0.0);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
                factor_block(k0 + SB, pb ^ 1);
            } else {
            // the next SB rows of the band (wavefronts 1-3 have the time: the block factorisation is what the step waits for);
            // they take the slots of the rows k0 .. k0 + SB - 1, whose block was consumed in the step before
            constexpr int NPF = (SB * (BC_MAXBW + 1) + 191) / 192;
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int idx = (tid - 64) + u * 192, ii = idx / bw1, jj = idx - ii * bw1;
                const int i = f0 + ii, j = i - bw + jj;
                if (ii < SB && i < n && j >= 0) sl(i, j) = A[(size_t)i * n + j] - ((INERTIA && j == i) ? shift : 0.0);
            }
            int cnt = 0;
            for (int ti = 1; ti < T16; ++ti)
                for (int tj = 0; tj <= ti; ++tj, ++cnt) {
                    if (cnt % 3 != w - 1) continue;      // (wave-uniform)
                    ss_v4d c;
                    update_tile(ti, tj, c, true);
                    const int i = k0 + SB + 16 * ti + l15;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int j = k0 + SB + 16 * tj + l4 + 4 * reg;
                        if (i < rend && j <= i) sl(i, j) = c[reg];
                    }
                }
        }
        __syncthreads();
        if (!ahead) {      // (workgroup-uniform)
            const int nb1 = min(SB, n - (k0 + SB));
            if (tid < SB * SB) {
                const int i = tid >> 4, j = tid & 15;
                Ld[i][j] = (i < nb1 && j <= i && i - j <= bw) ? sl(k0 + SB + i, k0 + SB + j) : ((i == j) ? 1.0 : 0.0);
            }
            __syncthreads();
            if (w == 0) factor_block(k0 + SB, pb ^ 1);
            __syncthreads();
        }
    }
    if (INERTIA) { if (tid == 0) info[b] = bad[0] ? -1 : bad[1]; }
    else if (tid == 0 && bad[0]) info[b] = 1;
}

// ---------------------------------------------------------------------------------------
// Wide-band factorisations by OUTER BLOCKS of G 16-column sub-panels (round 4)
// ---------------------------------------------------------------------------------------
// The two-panel walk below passes over the trailing window (bw x bw / 2 entries per matrix, read and written) once per
// 32 columns: 4 flop per byte, 3 TB/s and 12 TFLOP/s on config 5's 2 187-row agglomerates (half bandwidth ~580), and
// three launches per 32 columns whose panel kernels are latency chains.  Here an outer block of G sub-panels is
// factored LEFT-LOOKING -- sub-panel g first receives the updates of sub-panels 0 .. g - 1 of its block (their packed
// copies are a few hundred KB per matrix: L2), then is factored; one launch per outer block, the panel's columns read
// once and written once -- and the trailing window takes ONE rank-16 G update per outer block on the matrix cores
// (v_mfma_f64_16x16x4): 8 (G = 4) or 16 (G = 8) flop per byte of window traffic.
// Packed sub-panels: Pc = column operands (L21), Pr = row operands (L21 again, or L21 S of the signed walk
// C - theta I = L S L^T), sub-panel g at g * pstride, row r of the matrix at packed row r - (k0 + SB (g + 1)),
// rows up to k0 + SB G + bw (zeros past the band: the trailing update reads them with the sub-panel's row shift).
// MFMA lane roles (16 x 16 x 4, f64): lane = (l15, l4); A operand A[row l15][k l4], B operand B[k l4][col l15], C / D
// element [row l4 + 4 reg][col l15].  Everywhere below the MFMA "col" is a matrix ROW (16 consecutive rows of a
// column-major column: 128 contiguous bytes per quarter wavefront) and the MFMA "row" a matrix COLUMN.
template <int NT, int G, bool SIGNED>
__global__ __launch_bounds__(NT) void chol_panel_ll_kernel(int k0, int g0, int g1, const int *__restrict__ ns,
                                                            const int64_t *__restrict__ moff, const int64_t *__restrict__ voff,
                                                            double *__restrict__ W, double *__restrict__ Pc, double *__restrict__ Pr,
                                                            size_t pstride, int *__restrict__ info, const int *__restrict__ bws,
                                                            int *__restrict__ neg, int keep, const int *__restrict__ skip) {
    __shared__ double Ld[SB][SB + 1];
    __shared__ double Li[SB][SB + 1];
    __shared__ double sg[SB];
    __shared__ __align__(16) double Cb[G - 1][SB][18];      // column operands of the earlier sub-panels on the rows of this one
    __shared__ int bad;
    const int b = blockIdx.x, n = ns[b];
    if (skip && skip[b]) return;
    // (the sub-panels g0 .. g1 - 1 of the outer block in ONE launch: a matrix is one workgroup's from the first to the last, the
    // packed copies a sub-panel leaves are read back by the same workgroup behind a barrier -- seven launches less per block)
    for (int g = g0; g < g1; ++g) {
    const int kk = k0 + SB * g;
    if (kk >= n) break;
    const int bw = bws ? bws[b] : n;
    const int rin = min(n, kk + SB + bw);                        // rows from here on are zero in these columns
    const int rend = min(n, kk + SB + bw + SB * (G - 1 - g));    // ... and still written (as zeros) to the packed copies
    double *A = W + moff[b];
    const int nb = min(SB, n - kk);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    constexpr int NW = NT / 64;
    const size_t vo = (size_t)voff[b] * SB;
    for (int idx = tid; idx < g * SB * SB; idx += NT) {
        const int gp = idx >> 8, i = (idx >> 4) & 15, k = idx & 15;
        const int pr = kk + i - (k0 + SB * (gp + 1));
        Cb[gp][i][k] = (kk + i < n) ? Pc[gp * pstride + vo + (size_t)pr * SB + k] : 0.0;
    }
    __syncthreads();
    // the 16 rows from r0 on of the panel's columns with the block's earlier sub-panels applied: element
    // (row r0 + l15, column kk + l4 + 4 reg); rows past the band / the matrix come out as exact zeros
    auto updated = [&](int r0) {
        const int r = r0 + l15;
        const int rc = min(r, rin - 1);
        ss_v4d x;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) x[reg] = A[(size_t)min(kk + l4 + 4 * reg, n - 1) * n + rc];
#pragma unroll
        for (int gp = 0; gp < G - 1; ++gp) {
            if (gp < g) {      // (block-uniform)
                const double *p = Pr + gp * pstride + vo + (size_t)(rc - (k0 + SB * (gp + 1))) * SB + 4 * l4;
                const double2 p0 = *(const double2 *)p, p1 = *(const double2 *)(p + 2);
                // (A operands Cb[gp][column l15][k = 4 l4 + s] read where they are used: 56 VGPRs less at G = 8, two
                // workgroups of 512 per CU instead of one)
                const double2 a0 = *(const double2 *)&Cb[gp][l15][4 * l4], a1 = *(const double2 *)&Cb[gp][l15][4 * l4 + 2];
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, -p0.x, x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, -p0.y, x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, -p1.x, x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, -p1.y, x, 0, 0, 0);
            }
        }
        if (r >= rin) x = ss_v4d{0.0, 0.0, 0.0, 0.0};
        return x;
    };
    // rows below the diagonal block, 16 per wavefront and trip; the first tile's loads and products do not wait for the block
    const int ntile = (rend - kk - SB + 15) >> 4;
    ss_v4d xfirst = ss_v4d{0.0, 0.0, 0.0, 0.0};
    if (nb == SB && wv < ntile) xfirst = updated(kk + SB + 16 * wv);
    if (wv == 0) {
        const ss_v4d x = updated(kk);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = l4 + 4 * reg;
            Ld[l15][c] = (l15 < nb && c <= l15) ? x[reg] : ((l15 == c) ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int r = chol16_inverse_wave<SIGNED>(Ld, Li, tid, sg);
        if (tid == 0) bad = r;
    }
    __syncthreads();
    if (SIGNED) {
        if (tid == 0) { if (bad < 0) info[b] = 1; else neg[b] += bad; }
    } else {
        if (tid == 0 && bad) info[b] = 1;
    }
    if (!SIGNED || keep) {      // the diagonal block is kept INVERTED (both triangles), as the solves expect it
        if (tid < SB * SB) {
            const int i = tid >> 4, j = tid & 15;
            if (i < nb && j <= i) {
                A[(size_t)(kk + j) * n + (kk + i)] = Li[i][j];
                A[(size_t)(kk + i) * n + (kk + j)] = Li[i][j];
            }
        }
    }
    if (nb < SB) break;
    double la[4], sgc[4];     // A operand of the solve: L11^-1[row l15][k = l4 + 4 s] (the k slots of the C layout)
#pragma unroll
    for (int s = 0; s < 4; ++s) { la[s] = Li[l15][l4 + 4 * s]; sgc[s] = SIGNED ? sg[l4 + 4 * s] : 1.0; }
    double *Pcg = Pc + g * pstride + vo, *Prg = Pr + g * pstride + vo;
    for (int t = wv; t < ntile; t += NW) {
        const int r0 = kk + SB + 16 * t;
        const ss_v4d x = (t == wv) ? xfirst : updated(r0);
        ss_v4d y = ss_v4d{0.0, 0.0, 0.0, 0.0};      // row of L21 (S) = x L11^-T: element (row r0 + l15, column l4 + 4 reg)
#pragma unroll
        for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f64_16x16x4f64(la[s], x[s], y, 0, 0, 0);
        const int r = r0 + l15;
        const bool live = r < rend;      // (stores under the mask: no divergent branch around the matrix-core instructions)
        const size_t pr = (size_t)(r - kk - SB) * SB;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = l4 + 4 * reg;
            const double ts = y[reg], tv = SIGNED ? ts * sgc[reg] : ts;
            if (live) Pcg[pr + c] = tv;
            if (SIGNED && live) Prg[pr + c] = ts;
            if (r < rin) {
                if (!SIGNED) {
                    A[(size_t)(kk + c) * n + r] = tv;
                    A[(size_t)r * n + (kk + c)] = tv;
                } else if (keep) {
                    A[(size_t)(kk + c) * n + r] = tv;      // (L only; L^T: ss_keep_transpose_kernel, kept matrices only)
                }
            }
        }
    }
    __threadfence_block();      // (the next sub-panel reads this one's packed copies and reuses the LDS buffers)
    __syncthreads();
    }
}

// A22 -= sum over the G sub-panels of (row operands) (column operands)^T on the lower tiles of the trailing window of the
// outer block at k0.  One workgroup per strip of 32 WV rows (WV wavefronts, 32 rows each: their row operands stay in
// registers, negated), walking the column blocks of 16 up to the strip's diagonal; the column operands of a block are
// staged through LDS once per workgroup (double-buffered, one barrier per column block), the window tiles are requested
// one column block ahead.  Tiles that cross the diagonal are updated whole (the factorisations read the lower triangle).
template <int G, bool SIGNED, int WV>
__global__ __launch_bounds__(64 * WV) void band_trail_mfma_kernel(int k0, const int *__restrict__ ns, const int64_t *__restrict__ moff,
                                                              const int64_t *__restrict__ voff, double *__restrict__ W,
                                                              const double *__restrict__ Pc, const double *__restrict__ Pr,
                                                              size_t pstride, int count, int tiles, const int *__restrict__ bws,
                                                              const int *__restrict__ skip) {
    __shared__ __align__(16) double As[2][G][SB][18];
    int b, blk;
    xcd_decode(tiles, b, blk);
    if (b >= count) return;
    if (skip && skip[b]) return;
    const int n = ns[b], base = k0 + SB * G;
    int np = n - base;
    if (bws) np = min(np, bws[b]);
    if (np < 1) return;
    blk = tiles - 1 - blk;                 // (the long strips first)
    const int i0 = blk * (32 * WV);
    if (i0 >= np) return;
    double *A22 = W + moff[b] + (size_t)base * n + base;
    const size_t vo = (size_t)voff[b] * SB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int iw = i0 + 32 * wv;
    double br[2][G][4];      // row operands, negated: [tile][sub-panel][k = 4 l4 + s]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = min(iw + 16 * t + l15, np - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double *p = Pr + g * pstride + vo + (size_t)(r + SB * (G - 1 - g)) * SB + 4 * l4;
            const double2 p0 = *(const double2 *)p, p1 = *(const double2 *)(p + 2);
            br[t][g][0] = -p0.x; br[t][g][1] = -p0.y; br[t][g][2] = -p1.x; br[t][g][3] = -p1.y;
        }
    }
    const int jend = min(np, i0 + 32 * WV);
    const int nj = (jend + 15) >> 4;
    constexpr int CH = (G + WV - 1) / WV;  // 32-byte pieces of a column block's operands per thread
    double2 sv[CH][2];
    auto stage_load = [&](int jb) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int ch = min(tid + 64 * WV * u, 64 * G - 1), g = ch >> 6, j = (ch >> 2) & 15, q = ch & 3;
            const double *p = Pc + g * pstride + vo + (size_t)(min(16 * jb + j, np - 1) + SB * (G - 1 - g)) * SB + 4 * q;
            sv[u][0] = *(const double2 *)p;
            sv[u][1] = *(const double2 *)(p + 2);
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int ch = min(tid + 64 * WV * u, 64 * G - 1), g = ch >> 6, j = (ch >> 2) & 15, q = ch & 3;      // (duplicates store the same values)
            *(double2 *)&As[buf][g][j][4 * q] = sv[u][0];
            *(double2 *)&As[buf][g][j][4 * q + 2] = sv[u][1];
        }
    };
    // window tile (rows iw + 16 t + l15, columns 16 jb + l4 + 4 reg): always loaded from clamped addresses, stored under the mask
    const int rowc[2] = {min(iw + l15, np - 1), min(iw + 16 + l15, np - 1)};
    auto tile_load = [&](int jb, ss_v4d (&c)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) c[t][reg] = A22[(size_t)min(16 * jb + l4 + 4 * reg, np - 1) * n + rowc[t]];
    };
    stage_load(0);
    ss_v4d cn[2];
    tile_load(0, cn);
    stage_store(0);
    __syncthreads();
    for (int jb = 0; jb < nj; ++jb) {
        const int cur = jb & 1;
        const bool more = jb + 1 < nj;
        ss_v4d c[2] = {cn[0], cn[1]};
        if (more) {
            stage_load(jb + 1);
            tile_load(jb + 1, cn);
        }
        // (wave-uniform) tile t takes part when some column of the block is not beyond its last row and it has rows at all
        const bool on0 = iw < np && 16 * jb <= iw + 15, on1 = iw + 16 < np && 16 * jb <= iw + 31;
        if (on1) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double2 a0 = *(const double2 *)&As[cur][g][l15][4 * l4], a1 = *(const double2 *)&As[cur][g][l15][4 * l4 + 2];
                if (on0) {
                    c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, br[0][g][0], c[0], 0, 0, 0);
                    c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, br[0][g][1], c[0], 0, 0, 0);
                    c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, br[0][g][2], c[0], 0, 0, 0);
                    c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, br[0][g][3], c[0], 0, 0, 0);
                }
                c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, br[1][g][0], c[1], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, br[1][g][1], c[1], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, br[1][g][2], c[1], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, br[1][g][3], c[1], 0, 0, 0);
            }
        } else if (on0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double2 a0 = *(const double2 *)&As[cur][g][l15][4 * l4], a1 = *(const double2 *)&As[cur][g][l15][4 * l4 + 2];
                c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, br[0][g][0], c[0], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, br[0][g][1], c[0], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, br[0][g][2], c[0], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, br[0][g][3], c[0], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!(t ? on1 : on0)) continue;
            const int r = iw + 16 * t + l15;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int col = 16 * jb + l4 + 4 * reg;
                if (r < np && col < np) A22[(size_t)col * n + r] = c[t][reg];
            }
        }
        if (more) stage_store(cur ^ 1);
        __syncthreads();
    }
}


template <int G>
static void ss_factor_blocked(hipStream_t s, EigBatch &b, bool sgn, int *neg, int *info_p, const int *gbw_all, int bwmax, int keep,
                              const int *skip) {
    const int nmax = b.max_n;
    const bool prof = profiler().enabled;
    const size_t rows = (size_t)b.h_voff[b.count];
    const size_t pstride = (rows * SB + 15) / 16 * 16;
    if (b.ss_sub_n < pstride * G * 2) {      // (both operand sets, whichever factorisation comes first)
        b.ss_sub = eig_arena_subpanels(b, pstride * G * 2);
        b.ss_sub_n = pstride * G * 2;
    }
    double *Pc = b.ss_sub;
    double *Pr = sgn ? Pc + pstride * G : Pc;
    // ONE stream: the two halves on two streams that help the two-panel walk below (panels = latency chains beside the
    // other half's updates) cost these kernels a tenth (config 5: 8.44 against 7.98 s per step) -- a whole chunk's panels
    // fill the card two workgroups deep, and two windows' worth of tiles thrash what one leaves in the caches
    struct Part { hipStream_t q; int first, cnt; };
    const Part parts[1] = {{s, 0, b.count}};
    const int nparts = 1;
    auto step = [&](const Part &P, int k0) {
        hipStream_t q = P.q;
        const int f = P.first, cnt = P.cnt;
        const int cnt8 = 8 * div_up(cnt, 8);
        const int *ns = b.n.p + f;
        const int64_t *moff = b.moff.p + f, *voff = b.voff.p + f;
        const int *gbw = gbw_all ? gbw_all + f : nullptr;
        const int *sk = skip ? skip + f : nullptr;
        int *ng = neg ? neg + f : nullptr;
        int *info = info_p + f;
        const bool big = std::min(nmax - k0, bwmax + SB * G) > 448;
        if (prof) profiler().begin(q);
        {
            const int g1 = std::min(G, div_up(nmax - k0, SB));
            if (sgn) {
                if (big) hipLaunchKernelGGL((chol_panel_ll_kernel<512, G, true>), dim3(cnt), dim3(512), 0, q, k0, 0, g1, ns, moff, voff, b.W.p, Pc, Pr,
                                            pstride, info, gbw, ng, keep, sk);
                else hipLaunchKernelGGL((chol_panel_ll_kernel<256, G, true>), dim3(cnt), dim3(256), 0, q, k0, 0, g1, ns, moff, voff, b.W.p, Pc, Pr,
                                        pstride, info, gbw, ng, keep, sk);
            } else {
                if (big) hipLaunchKernelGGL((chol_panel_ll_kernel<512, G, false>), dim3(cnt), dim3(512), 0, q, k0, 0, g1, ns, moff, voff, b.W.p, Pc, Pr,
                                            pstride, info, gbw, (int *)nullptr, 0, sk);
                else hipLaunchKernelGGL((chol_panel_ll_kernel<256, G, false>), dim3(cnt), dim3(256), 0, q, k0, 0, g1, ns, moff, voff, b.W.p, Pc, Pr,
                                        pstride, info, gbw, (int *)nullptr, 0, sk);
            }
        }
        if (prof) profiler().end(q, sgn ? "eig_ss_inertia_panel" : "eig_ss_panel", 0.0, 0.0);
        const int npmax = std::min(nmax - k0 - SB * G, bwmax);
        if (npmax < 1) return false;
        double ub = 0.0;
        if (prof) {      // lower tiles of the trailing windows, read and written once
            for (size_t i = 0; i < b.h_n.size(); ++i) {
                double qn = (double)b.h_n[i] - k0 - SB * G;
                if (!b.h_bw.empty()) qn = std::min(qn, (double)b.h_bw[i]);
                if (qn >= 1.0) ub += 8.0 * qn * qn;
            }
            profiler().begin(q);
        }
        // (two wavefronts per workgroup: a strip's wavefronts stop at their own diagonal and wait for the last one at the
        // barriers, and the last strip of a window is partial -- with four, a third of the wavefront-steps were idle)
        // (four wavefronts per workgroup: with two, fewer wavefront-steps idle at a strip's diagonal but twice the staging
        // traffic -- 8.10 instead of 7.98 s per step on config 5; with eight 8.5)
        constexpr int WV = 4;
        const int tiles = div_up(npmax, 32 * WV);
        if (sgn) hipLaunchKernelGGL((band_trail_mfma_kernel<G, true, WV>), dim3(cnt8 * tiles), dim3(64 * WV), 0, q, k0, ns, moff, voff, b.W.p, Pc, Pr,
                                    pstride, cnt, tiles, gbw, sk);
        else hipLaunchKernelGGL((band_trail_mfma_kernel<G, false, WV>), dim3(cnt8 * tiles), dim3(64 * WV), 0, q, k0, ns, moff, voff, b.W.p, Pc, Pr,
                                pstride, cnt, tiles, gbw, sk);
        if (prof) profiler().end(q, sgn ? "eig_ss_inertia_update" : (npmax > 192 ? "eig_ss_update" : "eig_ss_update1"), ub, 0.0);
        return true;
    };
    for (int k0 = 0; k0 < nmax; k0 += SB * G) {
        bool more = false;
        for (int h = 0; h < nparts; ++h) more = step(parts[h], k0) || more;
        if (!more) break;
    }
    SA_HIP_CHECK(hipGetLastError());
}

bool eig_ss_band_enabled() {
    static const bool v = true;
    return v;
}

// C - sigma I = L L^T for every matrix of the batch (in place, L below / L^T above the diagonal).
// Returns false when a pivot was not positive.
// Blocked right-looking factorisation of every matrix in place, two panels per pass over the trailing
// matrix: panel k, its update of the next SB columns only, panel k + 1, then A22(2 SB:, 2 SB:) -=
// L_k L_k^T + L_{k+1} L_{k+1}^T in one read + write of the lower tiles.  sgn: C - theta I = L S L^T
// (inertia count; the factor is not kept), neg receives the negative pivots.
static void ss_factor_generic(hipStream_t s, EigBatch &b, bool sgn, int *neg, int *info_p, const int *bws, int bwmax,
                              int keep = 0, const int *skip = nullptr) {
    const int nmax = b.max_n;
    const bool prof = profiler().enabled;
    const int *gbw_all = bws ? bws : (sgn ? b.bw.p : nullptr);
    if (options().eig_outer_panels >= 8) return ss_factor_blocked<8>(s, b, sgn, neg, info_p, gbw_all, bwmax, keep, skip);
    if (options().eig_outer_panels >= 4) return ss_factor_blocked<4>(s, b, sgn, neg, info_p, gbw_all, bwmax, keep, skip);
    // Two halves of the batch on two streams (round 4): a panel is a latency chain (one workgroup per matrix: diagonal block by
    // one wavefront, then the panel rows), the trailing update is compute-bound -- one half's panels run beside the other half's
    // updates.  Every per-matrix array of the kernels is indexed by the matrix alone, so a half is the same launch with the
    // arrays shifted.  Not under the profiler (its event pair brackets one stream), not for small batches.
    const bool two = !prof && !env_serial() && b.count >= 64;
    hipStream_t s2 = two ? side_stream(5) : s;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (two) {
        SA_HIP_CHECK(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
        SA_HIP_CHECK(hipEventCreateWithFlags(&ev1, hipEventDisableTiming));
        SA_HIP_CHECK(hipEventRecord(ev0, s));                 // the matrices are assembled
        SA_HIP_CHECK(hipStreamWaitEvent(s2, ev0, 0));
    }
    const int half0 = two ? (b.count / 2 + 7) / 8 * 8 : b.count;      // (multiples of 8: the XCD decode of the tiled kernels)
    struct Part { hipStream_t q; int first, cnt; };
    const Part parts[2] = {{s, 0, half0}, {s2, half0, b.count - half0}};
    const int nparts = (two && b.count - half0 > 0) ? 2 : 1;
    auto step = [&](const Part &P, int k0) {
        hipStream_t q = P.q;
        const int f = P.first, cnt = P.cnt;
        const int cnt8 = 8 * div_up(cnt, 8);
        const int *ns = b.n.p + f;
        const int64_t *moff = b.moff.p + f, *voff = b.voff.p + f, *goff = b.goff.p + f;
        const int *gbw = gbw_all ? gbw_all + f : nullptr;
        const int *sk = skip ? skip + f : nullptr;
        int *ng = neg ? neg + f : nullptr;
        int *info = info_p + f;
        auto panel = [&](int kk, double *Vout, double *Zout, int rext, double *Sout) {
            if (prof) profiler().begin(q);
            const bool big = std::min(nmax, bwmax + 2 * SB) > 768;
            if (sgn) {
                if (big) hipLaunchKernelGGL((chol_panel_kernel<1024, true>), dim3(cnt), dim3(1024), 0, q, kk, ns, moff,
                                            voff, b.W.p, Vout, Zout, info, gbw, rext, Sout, ng, keep, sk);
                else hipLaunchKernelGGL((chol_panel_kernel<256, true>), dim3(cnt), dim3(256), 0, q, kk, ns, moff,
                                        voff, b.W.p, Vout, Zout, info, gbw, rext, Sout, ng, keep, sk);
            } else {
                if (big) hipLaunchKernelGGL((chol_panel_kernel<1024, false>), dim3(cnt), dim3(1024), 0, q, kk, ns, moff,
                                            voff, b.W.p, Vout, Zout, info, gbw, rext, (double *)nullptr, (int *)nullptr, 0, sk);
                else hipLaunchKernelGGL((chol_panel_kernel<256, false>), dim3(cnt), dim3(256), 0, q, kk, ns, moff,
                                        voff, b.W.p, Vout, Zout, info, gbw, rext, (double *)nullptr, (int *)nullptr, 0, sk);
            }
            if (prof) profiler().end(q, sgn ? "eig_ss_inertia_panel" : "eig_ss_panel", 0.0, 0.0);
        };
        const double *vrow = sgn ? b.Xbuf.p : nullptr, *zrow = sgn ? b.Tfac.p : nullptr;
        panel(k0, b.Vpk.p, b.Zbuf.p, SB, b.Xbuf.p);
        const int np1f = nmax - k0 - SB;         // order of the trailing matrix after panel k
        if (np1f < 1) return false;
        const int np1 = std::min(np1f, bwmax);   // ... of its part inside the band
        if (np1 >= 1) {
            if (prof) profiler().begin(q);
            hipLaunchKernelGGL(sbr_panel_update_kernel, dim3(cnt8 * div_up(np1, 256)), dim3(256), 0, q, k0, ns, moff,
                               voff, b.W.p, b.Vpk.p, b.Zbuf.p, cnt, div_up(np1, 256), 1, gbw, sk);
            if (prof) profiler().end(q, sgn ? "eig_ss_inertia_panel" : "eig_ss_panel", 0.0, 0.0);
        }
        panel(k0 + SB, b.Vpk2.p, nullptr, 0, b.Tfac.p);
        const int np = std::min(np1f - SB, bwmax);    // ... after panel k + 1
        if (np >= 1) {
            double ub = 0.0;
            if (prof) {      // lower tiles of the trailing matrices (inside the band), read and written once
                for (size_t i = 0; i < b.h_n.size(); ++i) {
                    double qn = (double)b.h_n[i] - k0 - 2 * SB;
                    if (!b.h_bw.empty()) qn = std::min(qn, (double)b.h_bw[i]);
                    if (qn >= 1.0) ub += 8.0 * qn * qn;
                }
                profiler().begin(q);
            }
            // (one row per lane, 108 VGPRs, four wavefronts per SIMD: 8 % faster on the 2 187-row agglomerates of config 5 than two
            // rows per lane at 212 VGPRs and two wavefronts per SIMD)
            hipLaunchKernelGGL((sbr_fused_kernel<false, 1, 3>), dim3(cnt8 * div_up(np, SF_ROWS)), dim3(S2_NT), 0, q, k0,
                               ns, moff, voff, b.W.p, b.Vpk.p, b.Vpk2.p, b.Vpk.p, b.Xbuf.p, goff,
                               b.Gbuf.p, b.trash.p, cnt, div_up(np, SF_ROWS), SB, gbw, vrow, zrow, sk);
            if (prof) profiler().end(q, sgn ? "eig_ss_inertia_update" : (np > 192 ? "eig_ss_update" : "eig_ss_update1"), ub, 0.0);
        }
        return true;
    };
    for (int k0 = 0; k0 < nmax; k0 += 2 * SB) {
        bool more = false;
        for (int h = 0; h < nparts; ++h) more = step(parts[h], k0) || more;
        if (!more) break;
    }
    SA_HIP_CHECK(hipGetLastError());
    if (two) {
        SA_HIP_CHECK(hipEventRecord(ev1, s2));
        SA_HIP_CHECK(hipStreamWaitEvent(s, ev1, 0));
        SA_HIP_CHECK(hipEventDestroy(ev0));
        SA_HIP_CHECK(hipEventDestroy(ev1));
    }
}

bool eig_subspace_factor(hipStream_t s, EigBatch &b) {
    const int nmax = b.max_n;
    b.ss_nb = 8;      // block of the iteration; 16 where more pairs are wanted than eight vectors reach (decided after the inertia pass)
    b.h_xpoff.clear();
    b.h_goff.assign((size_t)b.count + 1, 0);
    b.h_roff.assign((size_t)b.count + 1, 0);
    eig_batch_two_stage_buffers(b, 1, false, s);
    SA_REQUIRE(b.has_window, "few-eigenpairs path: the eigenvalue window must be set before the factorisation");
    DBuf<int> info((size_t)b.count);
    info.zero(s);
    // banded factorisation (SAAMGE_AMD_SS_BAND=0: treat every matrix as full)
    const bool use_band = eig_ss_band_enabled();
    const int *bws = nullptr;
    int bwmax = nmax;
    b.h_bw.clear();
    if (use_band) {
        if (!b.has_bw) {          // (the fused assembly has already measured them on the sparse rows)
            if (b.bw.n < (size_t)b.count) b.bw.alloc((size_t)b.count);
            profiler().begin(s);
            SA_HIP_CHECK(hipMemsetAsync(b.bw.p, 0, sizeof(int) * (size_t)b.count, s));
            const int ny = std::max(1, std::min(32, std::min(nmax / 64, 4096 / std::max(1, b.count))));
            hipLaunchKernelGGL(ss_band_kernel, dim3(b.count, ny), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, b.bw.p);
            profiler().end(s, "eig_ss_band", 0.0, 0.0);
        }
        b.h_bw.resize((size_t)b.count);
        SA_HIP_CHECK(hipMemcpyAsync(b.h_bw.data(), b.bw.p, sizeof(int) * (size_t)b.count, hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
        bwmax = 0;
        for (int v : b.h_bw) bwmax = std::max(bwmax, v);
        bws = b.bw.p;
    } else {                      // full matrices: bandwidth n - 1 (the inertia pass wants explicit widths)
        std::vector<int> full((size_t)b.count);
        for (int i = 0; i < b.count; ++i) full[i] = b.h_n[i] - 1;
        b.bw.from_host(full, s);
    }
    b.ss_bwmax = bwmax;
    if ((options().debug & 1)) {
        int bwmin = nmax;
        for (int v : b.h_bw) bwmin = std::min(bwmin, v);
        std::fprintf(stderr, "subspace: %d matrices, n max %d, half bandwidth %d .. %d\n", b.count, nmax, b.h_bw.empty() ? nmax : bwmin, bwmax);
    }
    constexpr bool use_lds = true;
    const bool lds_path = bws && use_lds && bwmax <= 128 - SB;      // the band fits the LDS window: one launch
    double cb = 0.0;      // the band read once, the factor written to both triangles
    if (lds_path)
        for (size_t i = 0; i < b.h_n.size(); ++i) cb += 8.0 * (double)b.h_n[i] * (std::min(b.h_bw[i], b.h_n[i] - 1) + 1);
    const bool prof = profiler().enabled;
    auto factor_generic = [&](bool sgn, int *neg) { ss_factor_generic(s, b, sgn, neg, info.p, bws, bwmax); };
    // ---- certified count: inertia of C - vu I, before the matrices are shifted and overwritten ----
    // (dsygvx counts by bisection, amg/src/xpacks.cpp:226-268; a subspace iteration alone cannot prove
    // that no eigenvalue below vu is missing from its block)
    const bool certify = options().eig_certify != 0;
    b.h_inertia.clear();
    b.ss_save = nullptr;
    // Wide-band matrices (coarse levels): a certified count of 0 means that the inertia pass met positive pivots
    // only, i.e. it WAS the Cholesky factorisation of C - vu I -- and vu is the best shift such a matrix can get
    // (its one wanted pair is the smallest).  Those matrices keep that factor: no restore, no second factorisation
    // (SAAMGE_AMD_SS_REUSE=0: factor twice as before).
    const bool reuse = options().eig_keep_inertia_factor != 0;
    bool generic_inertia = false;
    if (certify) {
        DBuf<int> neg((size_t)b.count);
        neg.zero(s);
        if (lds_path) {
            profiler().begin(s);
            auto go2 = [&](auto kern, int win) {
                SA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bc2_lds_bytes(win)));
                hipLaunchKernelGGL(kern, dim3(b.count), dim3(BC_NT), bc2_lds_bytes(win), s, b.n.p, b.moff.p, b.W.p, bws, neg.p,
                                   b.window_vu, (const int *)nullptr);
            };
            if (bwmax <= 67 - SB) go2(chol_band_lds2_kernel<67, true>, 67);
            else if (bwmax <= 68 - SB) go2(chol_band_lds2_kernel<68, true>, 68);
            else if (bwmax <= 80 - SB) go2(chol_band_lds2_kernel<80, true>, 80);
            else go2(chol_band_lds2_kernel<128, true>, 128);
            SA_HIP_CHECK(hipGetLastError());
            profiler().end(s, "eig_ss_inertia_lds", cb, 0.0);
            auto h = neg.to_host(s);
            b.h_inertia.assign(h.begin(), h.end());
        } else {
            // in place on the matrices: the band is saved, C - vu I factored, the band restored
            if (!prof) profiler().begin(s);
            std::vector<int64_t> soff((size_t)b.count + 1, 0);
            for (int i = 0; i < b.count; ++i) {
                const int64_t w = std::min(std::max(b.h_bw.empty() ? b.h_n[i] - 1 : b.h_bw[i], SB - 1), b.h_n[i] - 1);
                soff[(size_t)i + 1] = soff[i] + (int64_t)b.h_n[i] * (2 * w + 1);
            }
            DBuf<int64_t> d_soff;
            d_soff.from_host(soff, s);
            double *save = eig_arena_bandsave(b, (size_t)soff[b.count] + 1);
            b.ss_save = save;
            const int ny = std::max(1, std::min(64, 8192 / std::max(1, b.count)));
            hipLaunchKernelGGL(band_copy_kernel<false>, dim3(b.count, ny), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, b.bw.p, d_soff.p, save);
            hipLaunchKernelGGL(ss_shift_kernel, dim3(b.count), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, b.window_vu);
            // (reuse: the factor is written back; the bands are restored further down, once the counts say which
            // matrices keep it)
            ss_factor_generic(s, b, true, neg.p, info.p, bws, bwmax, reuse ? 1 : 0);
            if (!reuse)
                hipLaunchKernelGGL(band_copy_kernel<true>, dim3(b.count, ny), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, b.bw.p, d_soff.p, save);
            generic_inertia = true;
            SA_HIP_CHECK(hipGetLastError());
            b.ss_soff = std::move(d_soff);
            if (!prof) profiler().end(s, "eig_ss_inertia", 0.0, 0.0);
            auto h = neg.to_host(s);
            auto hi = info.to_host(s);
            b.h_inertia.assign(h.begin(), h.end());
            for (int i = 0; i < b.count; ++i) if (hi[i]) b.h_inertia[i] = -1;
            info.zero(s);
        }
    }

    // a matrix with more wanted pairs than the block holds, or without a certificate, goes to the dense path:
    // known here, before the second factorisation and the iterations are spent on it.  It goes ALONE (h_bad) as
    // long as such matrices are few: one uncertifiable agglomerate must not send the other 14 000 of its chunk
    // through the dense path.
    b.h_bad.assign((size_t)b.count, 0);
    b.nbad = 0;
    auto mark_bad = [&](int i) { if (!b.h_bad[(size_t)i]) { b.h_bad[(size_t)i] = 1; ++b.nbad; } };
    auto too_many_bad = [&]() { return (long)b.nbad * 10 > (long)b.count; };
    // The block of SIXTEEN vectors (round 4): wide-band matrices -- their solves stream the factor through HBM and work on
    // matrix-core tiles that are 16 columns wide whatever the block holds -- take it as soon as one of them wants more than the
    // six pairs a block of eight converges without locking (config 4 with 8 x 8 x 4-AE coarse blocks: eight wanted pairs, 1 035 ->
    // 799 ms per step); small matrices (right-hand sides in LDS beside the band: no room for sixteen) keep the block of eight and
    // its lock up to twelve wanted pairs, and take the block of sixteen -- through the streaming solves -- for 13 and 14.
    // With six pairs or fewer the block of eight stays: measured on the headline's level 1 (two wanted pairs), sixteen vectors
    // need 13 instead of 16 iterations of 1.4 instead of 1.3 ms -- nothing --, and a different basis of a degenerate eigenspace
    // (the six rigid-body modes of config 5) is a different member of the family of valid hierarchies.
    {
        int maxwant = 0;
        for (int v : b.h_inertia) maxwant = std::max(maxwant, v);
        if (maxwant <= 16 - 2 && ((nmax > 1280 && maxwant > SS_B - 2) || maxwant > SS_WANT_MAX)) b.ss_nb = 16;
    }
    for (size_t i = 0; i < b.h_inertia.size(); ++i)
        if (b.h_inertia[i] > (b.ss_nb == 16 ? 16 - 2 : SS_WANT_MAX) || b.h_inertia[i] < 0) {
            SA_REQUIRE(!options().eig_strict, "few-eigenpairs path: a matrix has more wanted pairs than the block holds, or no certificate (strict mode)");
            mark_bad((int)i);
        }
    if (options().eig_force_fallback > 0) {      // tests: every k-th matrix takes the per-matrix fallback
        const int every = options().eig_force_fallback;
        for (int i = 3; every > 0 && i < b.count; i += every) mark_bad(i);
    }
    if (too_many_bad()) return false;
    if (!b.h_inertia.empty()) b.inertia.from_host(b.h_inertia, s);
    // ---- C - sigma I = L L^T ----
    // The iteration converges like (lambda_i - sigma) / (lambda_9 - sigma), so the shift belongs just below the
    // wanted eigenvalues.  C is positive semi-definite (interior agglomerates: exactly singular), which leaves
    // sigma < 0 in general; but a certified count of 0 says that every eigenvalue lies above vu, and the one
    // pair the "at least one" rule then asks for is the smallest: those matrices are shifted to just below vu
    // (agglomerates on an essential boundary, most agglomerates of a coarse level: 17 -> 9 iterations).
    std::vector<int> h_keep((size_t)b.count, 0);
    int nkeep = 0;
    {
        const double vu = b.window_vu;
        const double neg = -std::min(1e-3, std::max(1e-7, std::fabs(vu) / 30.0));
        std::vector<double> sg((size_t)b.count, b.h_inertia.empty() ? SS_SIGMA : neg);
        for (int i = 0; i < b.count && !b.h_inertia.empty(); ++i)
            if (b.h_inertia[i] == 0 && vu > 0.0) {
                sg[i] = vu - std::max(1e-6, 1e-3 * vu);
                if (generic_inertia && reuse && !b.h_bad[i]) { sg[i] = vu; h_keep[(size_t)i] = 1; ++nkeep; }
            }
        b.ss_sigma.from_host(sg, s);
        b.h_sigma = sg;
    }
    DBuf<int> d_skip;
    if (generic_inertia && reuse) {       // the bands of the matrices that do not keep the inertia pass's factor
        d_skip.from_host(h_keep, s);
        const int ny = std::max(1, std::min(64, 8192 / std::max(1, b.count)));
        if (nkeep < b.count)
            hipLaunchKernelGGL(band_copy_kernel<true>, dim3(b.count, ny), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, b.bw.p,
                               b.ss_soff.p, b.ss_save, d_skip.p);
        if (nkeep > 0)
            hipLaunchKernelGGL(ss_keep_transpose_kernel, dim3(b.count, ny), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, bws, d_skip.p);
        SA_HIP_CHECK(hipGetLastError());
    }
    // ---- matrices that are finished before they are factored ----
    // certified count 1 and x0 = D^1/2 1 already an eigenvector to the acceptance tolerance (ss_nullcheck_kernel:
    // one pass over the band): they skip the second factorisation, the solves and the Rayleigh-Ritz steps, which
    // then run on a dense list of the others (85 % of the level-0 agglomerates of the 256^3 problem are of this
    // kind; the iteration accepted them after its first step anyway, at the price of a factorisation and two
    // solves each).  SAAMGE_AMD_SS_NULLCHECK=0 switches the shortcut off.
    b.h_pre.clear();
    DBuf<int> chol_active;
    int nchol = b.count;
    bool use_chol_list = false;
    const bool nullcheck = options().eig_nullcheck != 0;
    const bool generic_reuse = generic_inertia && reuse;
    if ((lds_path || generic_reuse) && nullcheck && !b.h_inertia.empty()) {
        profiler().begin(s);
        b.pre.alloc((size_t)b.count);
        b.pre_val.alloc(2 * (size_t)b.count);
        hipLaunchKernelGGL(ss_nullcheck_kernel, dim3(b.count), dim3(NC_NT), sizeof(double) * (size_t)nmax, s, b.n.p, b.moff.p,
                           b.voff.p, b.W.p, b.dis.p, b.has_perm ? b.perm.p : nullptr, bws, b.inertia.p, b.window_vu, b.ss_tol,
                           b.pre.p, b.pre_val.p, b.has_x0c ? b.x0c.p : (const double *)nullptr);
        SA_HIP_CHECK(hipGetLastError());
        profiler().end(s, "eig_ss_nullcheck", cb, 0.0);
        auto hp = b.pre.to_host(s);
        b.h_pre.assign(hp.begin(), hp.end());
        if ((options().debug & 1) && !lds_path) {
            auto hv = b.pre_val.to_host(s);
            int shown = 0;
            for (int i = 0; i < b.count && shown < 12; ++i)
                if (b.h_inertia[i] == 1) {
                    std::fprintf(stderr, "  nullcheck matrix %d: accepted %d, Rayleigh quotient %.3e, %s %.3e\n", i, hp[i], hv[2 * (size_t)i],
                                 hp[i] ? "1/|x0|" : "residual", std::fabs(hv[2 * (size_t)i + 1]));
                    ++shown;
                }
        }
    }
    if (lds_path && (!b.h_pre.empty() || b.nbad)) {       // the matrices the second factorisation still has to do
        std::vector<int> act;
        for (int i = 0; i < b.count; ++i)
            if (!(b.h_pre.empty() ? 0 : b.h_pre[i]) && !b.h_bad[i]) act.push_back(i);
        nchol = (int)act.size();
        if (nchol) chol_active.from_host(act, s);
        use_chol_list = true;
    }
    int nfactor = b.count;      // generic path: matrices the second factorisation still has to do
    if (generic_reuse) {
        std::vector<double> shifts(b.h_sigma);
        nfactor = 0;
        for (int i = 0; i < b.count; ++i) {
            if (!b.h_pre.empty() && b.h_pre[i]) h_keep[(size_t)i] = 1;
            if (b.h_bad[i]) h_keep[(size_t)i] = 1;
            if (h_keep[(size_t)i]) shifts[(size_t)i] = 0.0; else ++nfactor;
        }
        d_skip.from_host(h_keep, s);
        if (nfactor) {
            DBuf<double> d_shifts;
            d_shifts.from_host(shifts, s);
            hipLaunchKernelGGL(ss_shift_kernel, dim3(b.count), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, 0.0, d_shifts.p);
            SA_HIP_CHECK(hipStreamSynchronize(s));     // (d_shifts leaves scope)
        }
        if ((options().debug & 1))
            std::fprintf(stderr, "subspace: %d of %d wide-band matrices keep the factor of the inertia pass, %d are factored again\n",
                         nkeep, b.count, nfactor);
    } else
    hipLaunchKernelGGL(ss_shift_kernel, dim3(b.count), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, 0.0, b.ss_sigma.p);
    if (lds_path) {
        profiler().begin(s);
        auto go2 = [&](auto kern, int win) {
            if (!nchol) return;
            SA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bc2_lds_bytes(win)));
            hipLaunchKernelGGL(kern, dim3(nchol), dim3(BC_NT), bc2_lds_bytes(win), s, b.n.p, b.moff.p, b.W.p, bws, info.p, 0.0,
                               use_chol_list ? chol_active.p : (const int *)nullptr);
        };
        if (bwmax <= 67 - SB) go2(chol_band_lds2_kernel<67, false>, 67);
        else if (bwmax <= 68 - SB) go2(chol_band_lds2_kernel<68, false>, 68);
        else if (bwmax <= 80 - SB) go2(chol_band_lds2_kernel<80, false>, 80);
        else go2(chol_band_lds2_kernel<128, false>, 128);
        SA_HIP_CHECK(hipGetLastError());
        profiler().end(s, "eig_ss_chol_lds", 3.0 * cb, 0.0);
    } else {
        if (!prof) profiler().begin(s);
        if (!generic_reuse) factor_generic(false, nullptr);
        else if (nfactor) ss_factor_generic(s, b, false, nullptr, info.p, bws, bwmax, 0, d_skip.p);
        if (!prof) profiler().end(s, "eig_ss_cholesky", 0.0, 0.0);
    }
    auto h = info.to_host(s);
    for (int i = 0; i < b.count; ++i)
        if (h[i] && !b.h_bad[i]) {         // a non-positive pivot: that matrix alone
            if ((options().debug & 1))
                std::fprintf(stderr, "subspace: non-positive pivot in matrix %d (n %d, band %d, inertia %d, sigma %g)\n", i, b.h_n[i],
                             b.h_bw.empty() ? -1 : b.h_bw[i], b.h_inertia.empty() ? -2 : b.h_inertia[i], b.h_sigma[i]);
            SA_REQUIRE(!options().eig_strict, "few-eigenpairs path: non-positive pivot (strict mode)");
            mark_bad(i);
        }
    return !too_many_bad();
}

// Subspace iteration; fills b.h_m / b.m.  Returns false when some matrix needs the dense path.
bool eig_subspace_iterate(hipStream_t s, EigBatch &b, double vu) {
    DBuf<int> state((size_t)b.count);
    if (b.h_bad.size() != (size_t)b.count) { b.h_bad.assign((size_t)b.count, 0); b.nbad = 0; }
    {       // matrices already given to the dense path start as "failed": no kernel touches them
        std::vector<int> st0((size_t)b.count, 0);
        for (int i = 0; i < b.count; ++i) if (b.h_bad[i]) st0[i] = 2;
        state.from_host(st0, s);
    }
    auto mark_bad = [&](int i) { if (!b.h_bad[(size_t)i]) { b.h_bad[(size_t)i] = 1; ++b.nbad; } };
    auto too_many_bad = [&]() { return (long)b.nbad * 10 > (long)b.count; };
    double *X = b.Xbuf.p, *Z = b.Vpk2.p, *mu = b.d.p;     // d: rows >= SS_B per matrix is checked by the caller
    const int NB = b.ss_nb;
    // (kernels of the block are templates on its width)
    auto with_nb = [&](auto f) { if (NB == 16) f(std::integral_constant<int, 16>()); else f(std::integral_constant<int, 8>()); };
    DBuf<double> mubuf((size_t)b.count * NB);
    mu = mubuf.p;
    DBuf<double> slow_hist((size_t)b.count * 4);       // bound of the slowest wanted pair, last four iterations
    slow_hist.zero(s);
    // wide-band matrices with a saved band: a matrix (certified count 0) that converges too slowly may ask once
    // for a new shift (state bit 2, ss_rr_kernel); the host then restores the bands, shifts that matrix to just
    // below its smallest Ritz value and factors the batch again (SAAMGE_AMD_SS_RESHIFT=0: never)
    constexpr bool reshift_env = true;
    const bool reshift_on = reshift_env && b.max_n > 1280 && b.ss_save && !b.h_inertia.empty() && b.h_sigma.size() == (size_t)b.count;
    DBuf<int> reshift_ok;
    std::vector<int> h_reshift_ok((size_t)b.count, 1);
    std::vector<double> reshift_prev((size_t)b.count, 0.0), reshift_margin((size_t)b.count, 0.0);
    if (reshift_on) reshift_ok.from_host(h_reshift_ok, s);
    const bool prof = profiler().enabled;
    const int *bws = b.h_bw.empty() ? nullptr : b.bw.p;
    // matrices with seven to twelve wanted pairs: the first six are locked when they have converged (ss_lock_kernel)
    std::vector<int> h_ndefl((size_t)b.count, 0);
    bool any_lock = false;
    for (int i = 0; i < b.count && !b.h_inertia.empty() && NB == 8; ++i) any_lock = any_lock || (!b.h_bad[i] && b.h_inertia[i] > SS_B - 2);
    DBuf<int> it0;
    b.ss_has_lock = false;
    if (any_lock) {
        b.ss_ndefl.alloc((size_t)b.count);
        b.ss_ndefl.zero(s);
        it0.alloc((size_t)b.count);
        it0.zero(s);
        b.ss_lock_mu.alloc((size_t)b.count * SS_LOCK_PITCH);
        b.ss_Vlock.alloc((size_t)b.h_voff[b.count] * SS_LOCK_PITCH);
        b.ss_has_lock = true;
    }
    if ((options().debug & 1) && b.max_n > 1280 && !b.h_inertia.empty()) {
        std::fprintf(stderr, "subspace: certified counts (bad):");
        for (int i = 0; i < b.count; ++i) std::fprintf(stderr, " %d%s", b.h_inertia[i], b.h_bad[i] ? "*" : "");
        std::fprintf(stderr, "\n");
    }
    if (!prof) profiler().begin(s);
    with_nb([&](auto nb) {
        hipLaunchKernelGGL(ss_init_kernel<decltype(nb)::value>, dim3(b.count), dim3(256), 0, s, b.n.p, b.voff.p, b.dis.p,
                           b.has_perm ? b.perm.p : nullptr, X, b.has_x0c ? b.x0c.p : (const double *)nullptr);
    });
    bool done = false, failed = false;
    std::vector<int> hstate;
    // Only the matrices that are still iterating are launched, through a dense index list: accepted ones
    // would exit at once, but the survivors (agglomerates on the domain boundary: every 64th id, runs of 64)
    // would then sit on a few CUs -- block ids map to XCDs and CUs round-robin -- and a launch with 6 % of
    // the matrices active took as long as a full one.
    std::vector<int> h_active;
    h_active.reserve((size_t)b.count);
    for (int i = 0; i < b.count; ++i)
        if ((b.h_pre.empty() || !b.h_pre[i]) && !b.h_bad[i]) h_active.push_back(i);
    if (!b.h_pre.empty())       // finished before the factorisation (eig_subspace_factor): unit vector, Ritz value, state
        with_nb([&](auto nb) {
            hipLaunchKernelGGL(ss_preaccept_kernel<decltype(nb)::value>, dim3(b.count), dim3(256), 0, s, b.n.p, b.voff.p, b.pre.p, b.pre_val.p,
                               b.ss_sigma.p, X, mu, state.p);
        });
    DBuf<int> active((size_t)b.count);
    int nact = (int)h_active.size();
    if (nact) SA_HIP_CHECK(hipMemcpyAsync(active.p, h_active.data(), sizeof(int) * (size_t)nact, hipMemcpyHostToDevice, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    const int max_iter = any_lock ? 2 * SS_MAX_ITER : SS_MAX_ITER;      // (a locked matrix starts a second run)
    for (int iter = 0; iter < max_iter && !done; ++iter) {
        if (nact == 0) {                 // every matrix was finished before the factorisation
            auto t = state.to_host(s);
            hstate.assign(t.begin(), t.end());
            done = true;
            break;
        }
        if (prof) profiler().begin(s);
        const size_t xl_bytes = sizeof(double) * (size_t)b.max_n * XLP + 64;
        if (b.max_n <= 1280 && NB == 8) {
            // narrow bands: the deep-prefetch kernel; wider ones: the plain kernel, 256 or 512 threads by the band
            const int rows = std::min(b.max_n, b.ss_bwmax + SB);     // rows a block step updates
            auto go = [&](auto kern, int nt) {
                SA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                hipLaunchKernelGGL(kern, dim3(nact), dim3(nt), xl_bytes, s, b.n.p, b.moff.p, b.voff.p, b.W.p, X, Z, state.p, bws, active.p);
            };
            if (rows <= 128) go(ss_solve_lds_pf_kernel<128>, 128);
            else if (rows <= 256) go(ss_solve_lds_kernel<256>, 256);      // (>= SB * SB threads: the block loads)
            else go(ss_solve_lds_kernel<512>, 512);
        } else {
        // Z <- X for the matrices still iterating (the whole buffer used to be copied every iteration: 0.6 s of
        // config 5's 23 s, where most matrices of a chunk are done long before the last one)
        with_nb([&](auto nb) {
            hipLaunchKernelGGL(ss_copy_active_kernel<decltype(nb)::value>, dim3(nact, std::max(1, std::min(16, b.max_n / 256))), dim3(256), 0, s, b.n.p,
                               b.voff.p, active.p, X, Z);
        });
        constexpr bool no_win = false;
        const int wr = (std::min(b.max_n, bws ? std::max(b.ss_bwmax, SB) : b.max_n) + 2 * SB) | 1;      // rows of the LDS window (odd)
        const size_t win_bytes = sizeof(double) * (size_t)NB * (size_t)wr;
        if (b.max_n > 768 && !no_win && win_bytes <= 150 * 1024) {
            // more matrices than CUs: the variant without requests ahead, two workgroups per CU (config 5 at 64^3: solves 448 -> 397 ms
            // per step; SAAMGE_AMD_SS_TRSOLVE_TWO=0: one per CU always)
            constexpr bool two_env = true;
            auto go = [&](auto lo, auto up) {
                SA_HIP_CHECK(hipFuncSetAttribute((const void *)lo, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                SA_HIP_CHECK(hipFuncSetAttribute((const void *)up, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                hipLaunchKernelGGL(lo, dim3(nact), dim3(1024), win_bytes, s, wr, b.n.p, b.moff.p, b.voff.p, b.W.p, Z, state.p, bws, active.p);
                hipLaunchKernelGGL(up, dim3(nact), dim3(1024), win_bytes, s, wr, b.n.p, b.moff.p, b.voff.p, b.W.p, Z, state.p, bws, active.p);
            };
            with_nb([&](auto nb) {
                constexpr int B = decltype(nb)::value;
                if (two_env && nact > 256 && win_bytes <= 70 * 1024) go(ss_trsolve_win_kernel<false, 0, 2, B>, ss_trsolve_win_kernel<true, 0, 2, B>);
                else go(ss_trsolve_win_kernel<false, 4, 1, B>, ss_trsolve_win_kernel<true, 4, 1, B>);
            });
        } else if (b.max_n > 768) {
            with_nb([&](auto nb) {
                constexpr int B = decltype(nb)::value;
                hipLaunchKernelGGL((ss_trsolve_kernel<false, 1024, B>), dim3(nact), dim3(1024), 0, s, b.n.p, b.moff.p, b.voff.p, b.W.p, Z, state.p, bws, active.p);
                hipLaunchKernelGGL((ss_trsolve_kernel<true, 1024, B>), dim3(nact), dim3(1024), 0, s, b.n.p, b.moff.p, b.voff.p, b.W.p, Z, state.p, bws, active.p);
            });
        } else {
            with_nb([&](auto nb) {
                constexpr int B = decltype(nb)::value;
                hipLaunchKernelGGL((ss_trsolve_kernel<false, 256, B>), dim3(nact), dim3(256), 0, s, b.n.p, b.moff.p, b.voff.p, b.W.p, Z, state.p, bws, active.p);
                hipLaunchKernelGGL((ss_trsolve_kernel<true, 256, B>), dim3(nact), dim3(256), 0, s, b.n.p, b.moff.p, b.voff.p, b.W.p, Z, state.p, bws, active.p);
            });
        }
        }
        if (prof) {
            double ab = 0.0;      // the factor once per triangle + the right-hand sides, ACTIVE matrices only
            for (int i : h_active) {
                const double n = b.h_n[i], w = b.h_bw.empty() ? n : std::min(n, (double)b.h_bw[i] + SB);
                ab += 8.0 * n * (2.0 * w - w * w / n) + 2.0 * 8.0 * NB * n;
            }
            profiler().end(s, b.max_n <= 1280 ? "eig_ss_solve" : "eig_ss_solve_g", ab, 0.0);
            profiler().begin(s);
        }
        const bool dbg_on = (options().debug & 1) != 0;
        DBuf<double> dbgbuf;
        if (dbg_on) dbgbuf.alloc((size_t)b.count * 2 * NB);
        if (any_lock)
            hipLaunchKernelGGL(ss_deflate_kernel, dim3(nact), dim3(256), 0, s, b.n.p, b.voff.p, b.ss_Vlock.p, b.ss_ndefl.p, Z, state.p, active.p);
        with_nb([&](auto nb) {
            hipLaunchKernelGGL(ss_rr_kernel<decltype(nb)::value>, dim3(nact), dim3(256), 0, s, b.n.p, b.voff.p, X, Z, mu, state.p, iter,
                               b.ss_sigma.p, vu, b.h_inertia.empty() ? (const int *)nullptr : b.inertia.p, dbgbuf.p, active.p,
                               slow_hist.p, max_iter, reshift_on ? reshift_ok.p : (const int *)nullptr,
                               any_lock ? b.ss_ndefl.p : (const int *)nullptr, any_lock ? it0.p : (const int *)nullptr, b.ss_tol);
        });
        if (dbg_on && iter > 0) {
            auto hd = dbgbuf.to_host(s);
            const int show = std::min(b.count, 3);
            const int dbg_one = -1;
            if (dbg_one >= 0 && dbg_one < b.count) {
                std::fprintf(stderr, "  iter %d matrix %d: bounds", iter, dbg_one);
                for (int q = 0; q < 8; ++q) std::fprintf(stderr, " %.2e", hd[(size_t)dbg_one * 16 + q]);
                std::fprintf(stderr, " | lambda");
                for (int q = 0; q < 8; ++q) std::fprintf(stderr, " %.5e", hd[(size_t)dbg_one * 16 + 8 + q]);
                std::fprintf(stderr, " | inertia %d locked %d\n", b.h_inertia.empty() ? -2 : b.h_inertia[dbg_one], h_ndefl[dbg_one]);
            }
            for (int i = 0; i < show && dbg_one < 0; ++i) {
                const int mi = (int)((int64_t)i * (b.count - 1) / std::max(1, show - 1));
                std::fprintf(stderr, "  iter %d matrix %d (n %d): bounds %.2e %.2e %.2e | lambda %.6e %.6e %.6e | inertia %d\n", iter, mi, b.h_n[mi],
                             hd[(size_t)mi * 2 * NB], hd[(size_t)mi * 2 * NB + 1], hd[(size_t)mi * 2 * NB + 2], hd[(size_t)mi * 2 * NB + NB],
                             hd[(size_t)mi * 2 * NB + NB + 1], hd[(size_t)mi * 2 * NB + NB + 2], b.h_inertia.empty() ? -2 : b.h_inertia[mi]);
            }
        }
        if (prof) profiler().end(s, "eig_ss_rr", 0.0, 0.0);
        // The state is read back (and the active list rebuilt) after every iteration of the large matrices, whose
        // iterations cost milliseconds; the small ones (LDS solves, ~0.1 ms for the few hundred survivors of a
        // chunk) are checked after iterations 1, 3, 6, 9, ...: the read-back's host round trip (~0.25 ms) was
        // most of an iteration, an accepted matrix leaves the kernels at once, up to two spare iterations are cheap.
        const bool check = iter >= 1 && (b.max_n > 1280 || iter == 1 || iter % 3 == 0);
        if (check) {
            auto t = state.to_host(s);
            hstate.assign(t.begin(), t.end());
            if (any_lock) {      // lock requests (state bit 3)
                std::vector<int> req;
                for (int i = 0; i < b.count; ++i)
                    if ((hstate[i] & 8) && !(hstate[i] & 3) && h_ndefl[i] == 0 && !b.h_bad[i]) { req.push_back(i); h_ndefl[i] = SS_LOCK; hstate[i] = 0; }
                if (!req.empty()) {
                    DBuf<int> d_req;
                    d_req.from_host(req, s);
                    hipLaunchKernelGGL(ss_lock_kernel, dim3((unsigned)req.size()), dim3(256), 0, s, d_req.p, b.n.p, b.voff.p, X, mu, b.ss_sigma.p,
                                       b.ss_Vlock.p, b.ss_lock_mu.p, b.ss_ndefl.p, it0.p, iter, state.p, slow_hist.p);
                    SA_HIP_CHECK(hipGetLastError());
                    SA_HIP_CHECK(hipStreamSynchronize(s));
                    if ((options().debug & 1))
                        std::fprintf(stderr, "subspace: iteration %d, six pairs of %zu matrices locked (first: matrix %d)\n", iter, req.size(), req[0]);
                }
            }
            done = true;
            int nconv = 0;
            for (int i = 0; i < b.count; ++i) {
                const int v = hstate[i];
                if ((v & 2) && !b.h_bad[i]) {     // gave up during the iteration (too many pairs, breakdown, hopeless rate)
                    SA_REQUIRE(!options().eig_strict, "few-eigenpairs path gave up on a matrix (strict mode)");
                    mark_bad(i);
                }
                if (!(v & 3)) done = false;
                else ++nconv;
            }
            if (too_many_bad()) failed = true;
            if (failed && (options().debug & 1)) {
                std::fprintf(stderr, "subspace: iteration %d: too many matrices gave up (states:", iter);
                for (int i = 0; i < b.count && i < 64; ++i) std::fprintf(stderr, " %x", hstate[i]);
                std::fprintf(stderr, ")\n");
            }
            if (reshift_on && !failed) {
                std::vector<int> req, cand;
                for (int i = 0; i < b.count; ++i)
                    if ((hstate[i] & 4) && !(hstate[i] & 3) && h_reshift_ok[i]) cand.push_back(i);
                std::vector<double> hmu;
                if (!cand.empty()) { auto t2 = mubuf.to_host(s); hmu.assign(t2.begin(), t2.end()); }
                // The smallest Ritz value comes down towards the eigenvalue: the new shift has to stay below where it
                // will end.  A request is granted once the value has moved by less than half a percent in an iteration,
                // and the shift keeps five such steps (at least 2 % of the old distance) below it -- on the 7 900-row
                // level-1 agglomerates of config 4 with 8 x 8 x 4-AE blocks the value of iteration 6 was still more
                // than 2 % too high and the factorisation at the new shift met a negative pivot.
                for (int i : cand) {
                    const double gap = hmu[(size_t)i * NB];
                    const double moved = reshift_prev[(size_t)i] > 0.0 ? reshift_prev[(size_t)i] - gap : 1e300;
                    reshift_prev[(size_t)i] = gap;
                    if (moved <= 0.005 * gap) { reshift_margin[(size_t)i] = std::max(0.02 * gap, 5.0 * std::max(moved, 0.0)); req.push_back(i); }
                }
                if (!req.empty()) {
                    std::vector<double> delta(req.size());
                    for (size_t t = 0; t < req.size(); ++t) {
                        const int i = req[t];
                        const double gap = hmu[(size_t)i * NB];              // smallest Ritz value - sigma (> 0)
                        const double snew = b.h_sigma[i] + gap - reshift_margin[(size_t)i];
                        delta[t] = b.h_sigma[i] - snew;
                        b.h_sigma[i] = snew;
                        h_reshift_ok[i] = 0;
                    }
                    b.ss_sigma.from_host(b.h_sigma, s);
                    reshift_ok.from_host(h_reshift_ok, s);
                    const int ny = std::max(1, std::min(64, 8192 / std::max(1, b.count)));
                    hipLaunchKernelGGL(band_copy_kernel<true>, dim3(b.count, ny), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, b.bw.p,
                                       b.ss_soff.p, b.ss_save);
                    hipLaunchKernelGGL(ss_shift_kernel, dim3(b.count), dim3(256), 0, s, b.n.p, b.moff.p, b.W.p, 0.0, b.ss_sigma.p);
                    DBuf<int> info2((size_t)b.count);
                    info2.zero(s);
                    ss_factor_generic(s, b, false, nullptr, info2.p, bws, b.ss_bwmax);
                    auto hi2 = info2.to_host(s);
                    std::vector<int> st_fix;
                    for (int i = 0; i < b.count; ++i)
                        if (hi2[i] && !(hstate[i] & 1) && !b.h_bad[i]) {     // the new shift was not below the spectrum after all
                            mark_bad(i);
                            hstate[i] |= 2;
                            st_fix.push_back(i);
                        }
                    for (int i : st_fix) {
                        const int two = 2;
                        SA_HIP_CHECK(hipMemcpyAsync(state.p + i, &two, sizeof(int), hipMemcpyHostToDevice, s));
                    }
                    DBuf<int> d_req;
                    DBuf<double> d_delta;
                    d_req.from_host(req, s);
                    d_delta.from_host(delta, s);
                    with_nb([&](auto nb) {
                        hipLaunchKernelGGL(ss_reshift_mu_kernel<decltype(nb)::value>, dim3(div_up((long)req.size() * NB, 256)), dim3(256), 0, s, (int)req.size(),
                                           d_req.p, d_delta.p, mu, slow_hist.p);
                    });
                    SA_HIP_CHECK(hipGetLastError());
                    SA_HIP_CHECK(hipStreamSynchronize(s));
                    if ((options().debug & 1))
                        std::fprintf(stderr, "subspace: iteration %d, %zu matrices factored again at a shift below their smallest Ritz value\n", iter, req.size());
                    if (too_many_bad()) failed = true;
                    if (failed && (options().debug & 1)) {
                        std::fprintf(stderr, "subspace: iteration %d: the factorisation at the new shifts failed (info:", iter);
                        for (int i = 0; i < b.count && i < 64; ++i) std::fprintf(stderr, " %d", hi2[i]);
                        std::fprintf(stderr, ")\n");
                    }
                }
            }
            const bool dbg = (options().debug & 1) != 0;
            if (dbg) std::fprintf(stderr, "subspace: iteration %d, %d of %d matrices accepted (n max %d)\n", iter, nconv, b.count, b.max_n);
            if (failed) break;
            h_active.clear();
            for (int i = 0; i < b.count; ++i)
                if (!(hstate[i] & 3)) h_active.push_back(i);
            nact = (int)h_active.size();
            if (nact) SA_HIP_CHECK(hipMemcpyAsync(active.p, h_active.data(), sizeof(int) * (size_t)nact, hipMemcpyHostToDevice, s));
        }
    }
    SA_HIP_CHECK(hipGetLastError());
    if (!prof) profiler().end(s, "eig_ss_iterate", 0.0, 0.0);
    if (!failed && !done) {      // out of iterations: the unfinished matrices go to the dense path
        SA_REQUIRE(!options().eig_strict, "few-eigenpairs path: no convergence (strict mode)");
        { auto t = state.to_host(s); hstate.assign(t.begin(), t.end()); }
        for (int i = 0; i < b.count; ++i) if (!(hstate[i] & 3)) mark_bad(i);
        if (too_many_bad()) failed = true;
    }
    if (failed) {
        // (SAAMGE_AMD_SS_STRICT: the tests of this path must not pass on the dense fallback)
        SA_REQUIRE(!options().eig_strict, "few-eigenpairs path gave up on a batch (strict mode)");
        return false;
    }
    // certification: the number of Ritz values inside the window must be the number of eigenvalues
    // below vu (inertia of C - vu I); anything else sends the batch to the dense path
    if (!b.h_inertia.empty()) {
        int bad = 0, unsure = 0;
        for (int i = 0; i < b.count; ++i) {
            if (b.h_bad[i]) continue;
            const int k = ((hstate[i] >> 4) & 15) + h_ndefl[i];
            if (b.h_inertia[i] < 0) { ++unsure; mark_bad(i); }
            else if (b.h_inertia[i] != k) { ++bad; mark_bad(i); }
        }
        const bool dbg = (options().debug & 1) != 0;
        if (dbg || bad || unsure)
            std::fprintf(stderr, "saamge_amd: few-eigenpairs batch of %d: %d counts contradicted by the inertia, %d uncertified\n",
                         b.count, bad, unsure);
        if (bad || unsure) {
            SA_REQUIRE(!options().eig_strict, "few-eigenpairs path: count not certified (strict mode)");
            if (too_many_bad()) return false;
        }
    }
    b.h_m.assign((size_t)b.count, 1);
    for (int i = 0; i < b.count; ++i) b.h_m[i] = b.h_bad[i] ? 0 : (hstate[i] >> 8) + h_ndefl[i];
    b.m.from_host(b.h_m, s);
    b.ss_mu = std::move(mubuf);
    return true;
}

void eig_subspace_vectors(hipStream_t s, EigBatch &b, const int64_t *eoff, const int64_t *xoff, double *evals,
                          double *evecs) {
    profiler().begin(s);
    auto launch = [&](auto nb) {
        hipLaunchKernelGGL(ss_output_kernel<decltype(nb)::value>, dim3(b.count), dim3(256), 0, s, b.n.p, b.voff.p, b.Xbuf.p, b.ss_mu.p, b.dis.p,
                           b.has_perm ? b.perm.p : nullptr, b.m.p, eoff, xoff, evals, evecs, b.ss_sigma.p,
                           b.ss_has_lock ? b.ss_ndefl.p : (const int *)nullptr, b.ss_has_lock ? b.ss_Vlock.p : (const double *)nullptr,
                           b.ss_has_lock ? b.ss_lock_mu.p : (const double *)nullptr);
    };
    if (b.ss_nb == 16) launch(std::integral_constant<int, 16>()); else launch(std::integral_constant<int, 8>());
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "eig_ss_output", 0.0, 0.0);
}

}  // namespace saamge_amd
