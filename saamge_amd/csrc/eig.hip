// Batched symmetric eigensolver for gfx950 -- one workgroup per agglomerate matrix.
//
//   phase 1  blocked Householder tridiagonalisation (LAPACK dsytrd/dlatrd 'L' arithmetic):
//            per column one matrix-vector product with the trailing block (streamed from
//            L2/MALL/HBM, coalesced down the columns), thin panel corrections, and one
//            rank-2*NB update per panel with 4x4 register micro-tiles.
//   phase 2  Sturm counts at vl/vu (LAPACK dstebz/dlaebz pivot rule) -> number of wanted pairs.
//   phase 3  eigenvalues by 64-way multisection (one wavefront per eigenvalue, one shift per
//            lane), eigenvectors by inverse iteration with LAPACK dstein's logic
//            (dlagtf/dlagts pivoting LU, cluster re-orthogonalisation), back-transformation
//            by the stored reflectors, row scaling by D^-1/2.
//
// Reference behaviour: amg/src/xpacks.cpp:222-314 (dsygvx, range 'V' on (-1, theta],
// fallback to the single smallest pair), amg/src/spectral.cpp:124-237.
#include "eig.h"

#include <unordered_map>

#include <cfloat>
#include <cstdlib>
#include <string>

namespace saamge_amd {

constexpr int TRI_NT = 1024;  // threads per workgroup in phase 1 (16 wavefronts)
constexpr int VEC_NT = 256;   // threads per workgroup in phase 3
constexpr int VEC_MAXB = 254; // decoupled blocks of one tridiagonal handled separately (more: one last block)

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return __shfl(v, 0, 64);
}

template <int NT>
__device__ inline double block_sum(double v, double *red) {
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

// ---------------------------------------------------------------------------------------
// phase 1
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(TRI_NT) void tridiag_kernel(const int *__restrict__ ns,
                                                         const int64_t *__restrict__ moff,
                                                         const int64_t *__restrict__ voff,
                                                         double *__restrict__ Wm,
                                                         double *__restrict__ panel,
                                                         double *__restrict__ dd,
                                                         double *__restrict__ ee,
                                                         double *__restrict__ tt) {
    constexpr int NT = TRI_NT;
    constexpr int NB = EIG_NB;
    extern __shared__ __align__(16) double lds[];
    const int b = blockIdx.x;
    const int n = ns[b];
    double *A = Wm + moff[b];
    const int64_t vo = voff[b];
    double *d = dd + vo, *e = ee + vo, *tau = tt + vo;
    double *Wp = panel + vo * NB;
    double *v = lds;             // [n]
    double *p = v + n;           // [n]
    double *psum = p + n;        // [NT]
    double *tmpV = psum + NT;    // [NB]
    double *tmpW = tmpV + NB;    // [NB]
    double *red = tmpW + NB;     // [NT/64]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;

    for (int k0 = 0; k0 < n - 1; k0 += NB) {
        const int jb = min(NB, n - 1 - k0);
        for (int j = 0; j < jb; ++j) {
            const int k = k0 + j;
            const int len = n - k - 1;
            double *colk = A + (size_t)k * n;
            // (1) bring column k up to date with the panel's previous reflectors
            if (j > 0) {
                for (int i = k + tid; i < n; i += NT) {
                    double a = colk[i];
                    for (int c = 0; c < j; ++c) {
                        const double *Vc = A + (size_t)(k0 + c) * n;
                        const double *Wc = Wp + (size_t)c * n;
                        a -= Vc[i] * Wc[k] + Wc[i] * Vc[k];
                    }
                    colk[i] = a;
                }
                __syncthreads();
            }
            // (2) Householder vector of colk[k+1:n]  (dlarfg)
            const double alpha = colk[k + 1];
            double ss = 0.0;
            for (int i = k + 2 + tid; i < n; i += NT) {
                const double x = colk[i];
                ss = fma(x, x, ss);
            }
            ss = block_sum<NT>(ss, red);
            double tauk = 0.0, beta = alpha, scale = 0.0;
            if (ss != 0.0) {
                beta = -copysign(sqrt(fma(alpha, alpha, ss)), alpha);
                tauk = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            for (int i = k + 1 + tid; i < n; i += NT) {
                const double x = (i == k + 1) ? 1.0 : colk[i] * scale;
                colk[i] = x;
                v[i - k - 1] = x;
            }
            if (tid == 0) {
                d[k] = colk[k];
                e[k] = beta;
                tau[k] = tauk;
            }
            __syncthreads();
            double *Wj = Wp + (size_t)j * n;
            if (tauk == 0.0) {  // H = I: w = 0
                for (int i = k + 1 + tid; i < n; i += NT) Wj[i] = 0.0;
                __syncthreads();
                continue;
            }
            // (3) p = A22 v, A22 = A[k+1:, k+1:] (not yet updated by this panel)
            int RT = 64;
            while (RT < len && RT < NT) RT <<= 1;
            const int NG = NT / RT;
            {
                const int r = tid & (RT - 1), g = tid / RT;
                const int cb = (int)(((long)len * g) / NG), ce = (int)(((long)len * (g + 1)) / NG);
                const double *A22 = A + (size_t)(k + 1) * n + (k + 1);
                for (int r0 = 0; r0 < len; r0 += RT) {
                    const int row = r0 + r;
                    double sum = 0.0;
                    if (row < len) {
                        const double *Ar = A22 + row;
                        int c = cb;
                        for (; c + 4 <= ce; c += 4) {
                            const double a0 = Ar[(size_t)c * n], a1 = Ar[(size_t)(c + 1) * n];
                            const double a2 = Ar[(size_t)(c + 2) * n], a3 = Ar[(size_t)(c + 3) * n];
                            sum = fma(a0, v[c], sum);
                            sum = fma(a1, v[c + 1], sum);
                            sum = fma(a2, v[c + 2], sum);
                            sum = fma(a3, v[c + 3], sum);
                        }
                        for (; c < ce; ++c) sum = fma(Ar[(size_t)c * n], v[c], sum);
                    }
                    if (NG == 1) {
                        if (row < len) p[row] = sum;
                    } else {
                        psum[g * RT + r] = sum;
                    }
                }
                __syncthreads();
                if (NG > 1) {
                    for (int row = tid; row < len; row += NT) {
                        double s = 0.0;
                        for (int g2 = 0; g2 < NG; ++g2) s += psum[g2 * RT + row];
                        p[row] = s;
                    }
                    __syncthreads();
                }
            }
            // (4) tmpW = W^T v, tmpV = V^T v over rows k+1..n-1 (one wavefront per dot)
            for (int q = wave; q < 2 * j; q += NW) {
                const int c = q >> 1;
                const double *X = (q & 1) ? (A + (size_t)(k0 + c) * n) : (Wp + (size_t)c * n);
                double s = 0.0;
                for (int r = lane; r < len; r += 64) s = fma(X[k + 1 + r], v[r], s);
                s = wave_sum(s);
                if (lane == 0) ((q & 1) ? tmpV : tmpW)[c] = s;
            }
            __syncthreads();
            // (5) p = tau (p - V tmpW - W tmpV);  (6) w = p - tau/2 (p.v) v
            double part = 0.0;
            for (int row = tid; row < len; row += NT) {
                double s = p[row];
                for (int c = 0; c < j; ++c)
                    s -= A[(size_t)(k0 + c) * n + k + 1 + row] * tmpW[c] +
                         Wp[(size_t)c * n + k + 1 + row] * tmpV[c];
                s *= tauk;
                p[row] = s;
                part = fma(s, v[row], part);
            }
            const double pv = block_sum<NT>(part, red);
            const double alpha2 = -0.5 * tauk * pv;
            for (int row = tid; row < len; row += NT) Wj[k + 1 + row] = fma(alpha2, v[row], p[row]);
            __syncthreads();
        }
        // (7) A22 -= V W^T + W V^T on the block behind the panel, 4x4 register micro-tiles
        const int kk = k0 + jb;
        const int nt = n - kk;
        const int tiles = (nt + 3) >> 2;
        const long total = (long)tiles * tiles;
        for (long t = tid; t < total; t += NT) {
            const int ti = (int)(t % tiles), tl = (int)(t / tiles);
            const int i0 = kk + 4 * ti, l0 = kk + 4 * tl;
            double acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) acc[a][c2] = 0.0;
            const bool full = (i0 + 4 <= n) && (l0 + 4 <= n);
            for (int c = 0; c < jb; ++c) {
                const double *Vc = A + (size_t)(k0 + c) * n;
                const double *Wc = Wp + (size_t)c * n;
                double vi[4], wi[4], vl[4], wl[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int ii = (full || i0 + a < n) ? i0 + a : n - 1;
                    const int ll = (full || l0 + a < n) ? l0 + a : n - 1;
                    vi[a] = Vc[ii];
                    wi[a] = Wc[ii];
                    vl[a] = Vc[ll];
                    wl[a] = Wc[ll];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2)
                        acc[a][c2] = fma(vi[a], wl[c2], fma(wi[a], vl[c2], acc[a][c2]));
            }
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (i0 + a < n && l0 + c2 < n) A[(size_t)(l0 + c2) * n + i0 + a] -= acc[a][c2];
        }
        __syncthreads();
    }
    if (tid == 0) {
        d[n - 1] = A[(size_t)(n - 1) * n + (n - 1)];
        e[n - 1] = 0.0;
        tau[n - 1] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------
// Sturm sequence (dlaebz pivot rule): number of eigenvalues <= x
// ---------------------------------------------------------------------------------------
// 1 / q from the hardware seed + two Newton steps (<= 1 ulp): the Sturm recurrence is one long
// chain of dependent divisions, and the IEEE-exact sequence is twice as long.  |q| >= pivmin.
__device__ inline double chain_rcp(double q) {
    double r = __builtin_amdgcn_rcp(q);
    r = fma(fma(-q, r, 1.0), r, r);
    r = fma(fma(-q, r, 1.0), r, r);
    return r;
}

__device__ inline int sturm_count(int n, const double *d, const double *e, double x, double pivmin) {
    double q = d[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    int cnt = (q <= 0.0) ? 1 : 0;
    for (int i = 1; i < n; ++i) {
        const double ei = e[i - 1];
        q = d[i] - (ei * ei) * chain_rcp(q) - x;
        if (fabs(q) < pivmin) q = -pivmin;
        cnt += (q <= 0.0) ? 1 : 0;
    }
    return cnt;
}

// dstebz's splitting criterion: e_i^2 <= ulp^2 |d_i d_{i+1}| + safmin decouples the tridiagonal
__device__ inline bool negligible_offdiag(double di, double dn, double ei) {
    return fabs(di * dn) * (DBL_EPSILON * DBL_EPSILON) + DBL_MIN > ei * ei;
}

__global__ __launch_bounds__(64) void count_kernel(const int *__restrict__ ns,
                                                   const int64_t *__restrict__ voff,
                                                   const double *__restrict__ dd,
                                                   const double *__restrict__ ee, double vl,
                                                   double vu, int *__restrict__ m_out,
                                                   int *__restrict__ j0_out) {
    extern __shared__ __align__(16) double lds[];
    const int b = blockIdx.x;
    const int n = ns[b];
    const double *dg = dd + voff[b], *eg = ee + voff[b];
    double *d = lds, *e = lds + n;
    const int lane = threadIdx.x;
    double emax = 0.0;
    for (int i = lane; i < n; i += 64) {
        d[i] = dg[i];
        const double ei = eg[i];
        e[i] = ei;
        emax = fmax(emax, ei * ei);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) emax = fmax(emax, __shfl_xor(emax, o, 64));
    __syncthreads();
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    for (int i = lane; i + 1 < n; i += 64)      // the splits of dstebz, as in eigvec_kernel
        if (negligible_offdiag(d[i], d[i + 1], e[i])) e[i] = 0.0;
    __syncthreads();
    int cnt = 0;
    if (lane < 2) cnt = sturm_count(n, d, e, lane == 0 ? vl : vu, pivmin);
    const int cl = __shfl(cnt, 0, 64), cu = __shfl(cnt, 1, 64);
    if (lane == 0) {
        int m = cu - cl, j0 = cl;
        if (m <= 0) {  // "atleast_one": the single smallest eigenpair (range 'I', il=iu=1)
            m = 1;
            j0 = -1;   // flag for eigvec_kernel
        }
        m_out[b] = m;
        j0_out[b] = j0;
    }
}

// ---------------------------------------------------------------------------------------
// phase 3
// ---------------------------------------------------------------------------------------
__device__ inline double unit_rand(unsigned a, unsigned b) {  // deterministic uniform(-1,1)
    unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u + (a << 6) + (a >> 2));
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return ((double)h + 0.5) * (2.0 / 4294967296.0) - 1.0;
}

// GWS: the per-matrix work arrays (tridiagonal, LU factors, iterate: ~58 n bytes) live in a global
// workspace instead of LDS -- agglomerates beyond ~2 900 rows, where they no longer fit 160 KiB
// (same code through generic pointers; the workgroup barriers order the accesses).
template <bool GWS>
__global__ __launch_bounds__(VEC_NT) void eigvec_kernel(
    const int *__restrict__ ns, const int64_t *__restrict__ moff, const int64_t *__restrict__ voff,
    const double *__restrict__ Wm, const double *__restrict__ dd, const double *__restrict__ ee,
    const double *__restrict__ tt, const double *__restrict__ dis, const int *__restrict__ ms,
    const int *__restrict__ j0s, const int64_t *__restrict__ eoff, const int64_t *__restrict__ xoff,
    double *__restrict__ evals, double *__restrict__ evecs, int do_backtransform, double vl, double vu,
    double *__restrict__ gws, int64_t gws_stride) {
    constexpr int NT = VEC_NT;
    constexpr int NW = NT / 64;
    extern __shared__ __align__(16) double lds_[];
    double *lds = GWS ? gws + (size_t)blockIdx.x * gws_stride : lds_;
    const int bi = blockIdx.x;
    const int n = ns[bi];
    const int m = ms[bi], j0 = j0s[bi];
    const double *A = Wm + moff[bi];
    const int64_t vo = voff[bi];
    const double *tau = tt + vo;
    double *lam = evals + eoff[bi];
    double *Y = evecs + xoff[bi];
    double *d = lds;        // [n]
    double *e = d + n;      // [n]
    double *la = e + n;     // [n] LU diag
    double *lb = la + n;    // [n] LU super 1
    double *lc = lb + n;    // [n] multipliers
    double *ld2 = lc + n;   // [n] LU super 2
    double *z = ld2 + n;    // [n]
    double *red = z + n;    // [8]
    int *iscr = (int *)(red + 8);                         // [8]
    unsigned short *oblk = (unsigned short *)(iscr + 8);  // [n] block of each output slot
    unsigned short *bstart = oblk + n;                    // [VEC_MAXB + 2] first row of each block
    unsigned short *bslot = bstart + VEC_MAXB + 2;        // [VEC_MAXB + 2] first output slot of each block
    unsigned short *bclo = bslot + VEC_MAXB + 2;          // [VEC_MAXB + 2] local index of its first wanted pair
    unsigned char *pin = (unsigned char *)(bclo + VEC_MAXB + 2);   // [n] LU row interchanges
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    double emax = 0.0, gl = DBL_MAX, gu = -DBL_MAX, onenrm = 0.0;
    for (int i = tid; i < n; i += NT) {
        d[i] = dd[vo + i];
        e[i] = ee[vo + i];
    }
    __syncthreads();
    // Splits (dstebz): a negligible off-diagonal decouples the matrix into blocks; eigenvalues and
    // eigenvectors are computed block by block (dstebz order 'B' + dstein), which is what keeps
    // inverse iteration well defined when decoupled blocks share eigenvalues (e.g. an agglomerate
    // made of identical disconnected pieces: one 30-fold eigenvalue, 30 blocks).
    for (int i = tid; i + 1 < n; i += NT)
        if (negligible_offdiag(d[i], d[i + 1], e[i])) e[i] = 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const double el = (i > 0) ? fabs(e[i - 1]) : 0.0, er = (i < n - 1) ? fabs(e[i]) : 0.0;
        emax = fmax(emax, er * er);
        gl = fmin(gl, d[i] - el - er);
        gu = fmax(gu, d[i] + el + er);
        onenrm = fmax(onenrm, fabs(d[i]) + el + er);
    }
    // block max/min reductions through LDS
    {
        double vals[4] = {emax, -gl, gu, onenrm};
        for (int q = 0; q < 4; ++q) {
            double x = vals[q];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, 64));
            __syncthreads();
            if (lane == 0) red[wave] = x;
            __syncthreads();
            double r = red[0];
            for (int w2 = 1; w2 < NW; ++w2) r = fmax(r, red[w2]);
            vals[q] = r;
        }
        emax = vals[0]; gl = -vals[1]; gu = vals[2]; onenrm = vals[3];
    }
    const double ulp = DBL_EPSILON;  // dlamch('P')
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    const double tnorm = fmax(fabs(gl), fabs(gu));
    gl -= 2.1 * tnorm * ulp * n + 2.1 * pivmin;
    gu += 2.1 * tnorm * ulp * n + 2.1 * pivmin;

    // ---- blocks and the wanted local index range of each ----
    if (tid == 0) iscr[2] = 0;
    __syncthreads();
    for (int i = tid; i + 1 < n; i += NT)
        if (e[i] == 0.0) iscr[2] = 1;        // (benign race: every writer stores the same value)
    __syncthreads();
    const int any_split = iscr[2];
    if (tid == 0) {
        int nb = 0;
        bstart[0] = 0;
        if (any_split)
            for (int i = 0; i + 1 < n; ++i)
                if (e[i] == 0.0 && nb < VEC_MAXB - 1) bstart[++nb] = (unsigned short)(i + 1);   // (beyond VEC_MAXB: one last block)
        ++nb;
        bstart[nb] = (unsigned short)n;
        iscr[0] = nb;
        if (nb == 1) {      // the usual case: the window of the whole matrix is the one of count_kernel
            bclo[0] = (unsigned short)max(j0, 0);
            bslot[0] = (unsigned short)m;
        }
    }
    __syncthreads();
    const int nb = iscr[0];
    const bool smallest_only = j0 < 0;      // nothing in (vl, vu]: the smallest eigenpair of the matrix
    if (wave == 0 && nb > 1) {
        // counts of every block at vl / vu (lane pairs: block = lane >> 1)
        for (int b0 = 0; b0 < nb; b0 += 32) {
            const int bq = b0 + (lane >> 1);
            int c = 0;
            if (bq < nb && !smallest_only) {
                const int bs = bstart[bq], len = bstart[bq + 1] - bs;
                c = sturm_count(len, d + bs, e + bs, (lane & 1) ? vu : vl, pivmin);
            }
            const int cu = __shfl_down(c, 1, 64);
            if (bq < nb && !(lane & 1)) {
                bclo[bq] = (unsigned short)c;
                bslot[bq] = (unsigned short)(smallest_only ? 1 : max(cu - c, 0));   // count for now
            }
        }
    }
    __syncthreads();
    if (tid == 0) {      // exclusive scan of the per-block counts -> first output slot
        int run = 0;
        for (int bq = 0; bq < nb; ++bq) {
            const int c = bslot[bq];
            bslot[bq] = (unsigned short)run;
            run += c;
        }
        bslot[nb] = (unsigned short)run;
        iscr[1] = run;
    }
    __syncthreads();
    // number of (block, local index) candidates: m in window mode, one per block otherwise
    const int ncand = smallest_only ? nb : min(iscr[1], m);
    for (int bq = tid; bq < nb; bq += NT)
        for (int sl = bslot[bq]; sl < bslot[bq + 1] && sl < n; ++sl) oblk[sl] = (unsigned short)bq;
    __syncthreads();

    // ---- A: the wanted eigenvalues of every block by 64-way multisection, one wavefront each ----
    // (smallest-only mode: the smallest eigenvalue of every block goes to z[], the minimum wins)
    for (int sl = wave; sl < ncand; sl += NW) {
        const int bq = oblk[sl];
        const int bs = bstart[bq], len = bstart[bq + 1] - bs;
        const int want = smallest_only ? 0 : bclo[bq] + (sl - bslot[bq]);  // count(lo) <= want < count(hi)
        double lo = gl, hi = gu;
        for (int it = 0; it < 40; ++it) {
            const double width = hi - lo;
            const double tol = fmax(2.0 * ulp * fmax(fabs(lo), fabs(hi)), pivmin);
            if (width <= tol) break;
            const double x = lo + width * ((double)(lane + 1) / 65.0);
            const int c = sturm_count(len, d + bs, e + bs, x, pivmin);
            const unsigned long long ge = __ballot(c >= want + 1);
            double nlo = lo, nhi = hi;
            if (ge == 0ull) {
                nlo = __shfl(x, 63, 64);
            } else {
                const int f = __ffsll((long long)ge) - 1;
                nhi = __shfl(x, f, 64);
                if (f > 0) nlo = __shfl(x, f - 1, 64);
            }
            if (!(nhi - nlo < width)) break;  // no progress (rounding)
            lo = nlo;
            hi = nhi;
        }
        if (lane == 0) {
            if (smallest_only) z[sl] = 0.5 * (lo + hi);
            else lam[sl] = 0.5 * (lo + hi);
        }
    }
    __syncthreads();
    if (smallest_only) {
        if (tid == 0) {
            int best = 0;
            for (int bq = 1; bq < nb; ++bq)
                if (z[bq] < z[best]) best = bq;
            lam[0] = z[best];
            oblk[0] = (unsigned short)best;
            bclo[best] = 0;
            bslot[best] = 0;
        }
        __syncthreads();
    }
    const int mout = smallest_only ? 1 : ncand;    // == m

    // ---- B: inverse iteration (dstein) block by block, vectors stored in Y (unit 2-norm) ----
    const double eps = ulp;
    const double ortol = 1e-3 * onenrm;
    double xjm = 0.0;
    int gpind = 0;
    for (int jj = 0; jj < mout; ++jj) {
        const int bq = oblk[jj];
        const int bs = bstart[bq], nl = bstart[bq + 1] - bs;     // block rows [bs, bs + nl)
        const bool first_in_block = smallest_only || jj == bslot[bq];
        const double *db = d + bs, *eb = e + bs;
        const double dtpcrt = sqrt(0.1 / (double)nl);
        double xj = lam[jj];
        if (!first_in_block) {
            const double pertol = 10.0 * fabs(eps * xj);
            if (xj - xjm < pertol) xj = xjm + pertol;
        }
        double *Yj = Y + (size_t)jj * n;
        for (int i = tid; i < n; i += NT) Yj[i] = 0.0;
        if (nl == 1) {
            if (tid == 0) Yj[bs] = 1.0;
            xjm = xj;
            __syncthreads();
            continue;
        }
        for (int i = tid; i < nl; i += NT) z[i] = unit_rand((unsigned)(bs + i), (unsigned)(jj + 1));
        // dlagtf: LU of T - xj I with partial pivoting (sequential, thread 0)
        if (tid == 0) {
            for (int i = 0; i < nl; ++i) {
                la[i] = db[i] - xj;
                if (i < nl - 1) { lb[i] = eb[i]; lc[i] = eb[i]; }
            }
            double scale1 = fabs(la[0]) + fabs(lb[0]);
            for (int k = 0; k < nl - 1; ++k) {
                double scale2 = fabs(lc[k]) + fabs(la[k + 1]);
                if (k < nl - 2) scale2 += fabs(lb[k + 1]);
                const double piv1 = (la[k] == 0.0) ? 0.0 : fabs(la[k]) * chain_rcp(scale1);
                if (lc[k] == 0.0) {
                    pin[k] = 0;
                    scale1 = scale2;
                    if (k < nl - 2) ld2[k] = 0.0;
                } else {
                    const double piv2 = fabs(lc[k]) * chain_rcp(scale2);
                    if (piv2 <= piv1) {
                        pin[k] = 0;
                        scale1 = scale2;
                        lc[k] = lc[k] * chain_rcp(la[k]);
                        la[k + 1] -= lc[k] * lb[k];
                        if (k < nl - 2) ld2[k] = 0.0;
                    } else {
                        pin[k] = 1;
                        const double mult = la[k] * chain_rcp(lc[k]);
                        la[k] = lc[k];
                        const double temp = la[k + 1];
                        la[k + 1] = lb[k] - mult * temp;
                        if (k < nl - 2) {
                            ld2[k] = lb[k + 1];
                            lb[k + 1] = -mult * ld2[k];
                        }
                        lb[k] = temp;
                        lc[k] = mult;
                    }
                }
            }
        }
        __syncthreads();
        // dlagts tolerance
        double tolp = 0.0;
        for (int i = tid; i < nl; i += NT) {
            double t = fabs(la[i]);
            if (i < nl - 1) t = fmax(t, fabs(lb[i]));
            if (i < nl - 2) t = fmax(t, fabs(ld2[i]));
            tolp = fmax(tolp, t);
        }
        {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tolp = fmax(tolp, __shfl_xor(tolp, o, 64));
            __syncthreads();
            if (lane == 0) red[wave] = tolp;
            __syncthreads();
            tolp = red[0];
            for (int w2 = 1; w2 < NW; ++w2) tolp = fmax(tolp, red[w2]);
            tolp *= eps;
            if (tolp == 0.0) tolp = eps;
        }
        if (first_in_block) gpind = jj;
        else if (fabs(xj - xjm) > ortol) gpind = jj;
        int nrmchk = 0;
        for (int its = 0; its < 5; ++its) {
            // scale the right-hand side
            double zmax = 0.0;
            for (int i = tid; i < nl; i += NT) zmax = fmax(zmax, fabs(z[i]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) zmax = fmax(zmax, __shfl_xor(zmax, o, 64));
            __syncthreads();
            if (lane == 0) red[wave] = zmax;
            __syncthreads();
            zmax = red[0];
            for (int w2 = 1; w2 < NW; ++w2) zmax = fmax(zmax, red[w2]);
            const double scl = (double)nl * onenrm * fmax(eps, fabs(la[nl - 1])) / zmax;
            for (int i = tid; i < nl; i += NT) z[i] *= scl;
            __syncthreads();
            // dlagts job = -1 (sequential, thread 0)
            if (tid == 0) {
                for (int k = 1; k < nl; ++k) {
                    if (pin[k - 1] == 0) {
                        z[k] -= lc[k - 1] * z[k - 1];
                    } else {
                        const double temp = z[k - 1];
                        z[k - 1] = z[k];
                        z[k] = temp - lc[k - 1] * z[k];
                    }
                }
                const double sfmin = DBL_MIN, bignum = 1.0 / DBL_MIN;
                for (int k = nl - 1; k >= 0; --k) {
                    double temp;
                    if (k <= nl - 3) temp = z[k] - lb[k] * z[k + 1] - ld2[k] * z[k + 2];
                    else if (k == nl - 2) temp = z[k] - lb[k] * z[k + 1];
                    else temp = z[k];
                    double ak = la[k];
                    double pert = copysign(tolp, ak);
                    for (;;) {
                        const double absak = fabs(ak);
                        if (absak < 1.0) {
                            if (absak < sfmin) {
                                if (absak == 0.0 || fabs(temp) * sfmin > absak) {
                                    ak += pert;
                                    pert *= 2.0;
                                    continue;
                                } else {
                                    temp *= bignum;
                                    ak *= bignum;
                                }
                            } else if (fabs(temp) > absak * bignum) {
                                ak += pert;
                                pert *= 2.0;
                                continue;
                            }
                        }
                        break;
                    }
                    z[k] = temp * chain_rcp(ak);
                }
            }
            __syncthreads();
            // re-orthogonalise against the cluster (modified Gram-Schmidt; same block)
            for (int g = gpind; g < jj; ++g) {
                const double *Yg = Y + (size_t)g * n + bs;
                double s = 0.0;
                for (int i = tid; i < nl; i += NT) s = fma(z[i], Yg[i], s);
                s = block_sum<NT>(s, red);
                for (int i = tid; i < nl; i += NT) z[i] = fma(-s, Yg[i], z[i]);
                __syncthreads();
            }
            double nrm = 0.0;
            for (int i = tid; i < nl; i += NT) nrm = fmax(nrm, fabs(z[i]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o, 64));
            __syncthreads();
            if (lane == 0) red[wave] = nrm;
            __syncthreads();
            nrm = red[0];
            for (int w2 = 1; w2 < NW; ++w2) nrm = fmax(nrm, red[w2]);
            if (nrm < dtpcrt) continue;
            if (++nrmchk < 3) continue;
            break;
        }
        // normalise: unit 2-norm, largest component positive
        double s2 = 0.0, amax = 0.0;
        int imax = nl;
        for (int i = tid; i < nl; i += NT) {
            s2 = fma(z[i], z[i], s2);
            if (fabs(z[i]) > amax) { amax = fabs(z[i]); imax = i; }
        }
        s2 = block_sum<NT>(s2, red);
        // first index attaining the max (idamax)
        double gm = amax;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o, 64));
        __syncthreads();
        if (lane == 0) red[wave] = gm;
        __syncthreads();
        gm = red[0];
        for (int w2 = 1; w2 < NW; ++w2) gm = fmax(gm, red[w2]);
        int cand = (amax == gm) ? imax : nl;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        __syncthreads();
        if (lane == 0) iscr[wave] = cand;
        __syncthreads();
        cand = iscr[0];
        for (int w2 = 1; w2 < NW; ++w2) cand = min(cand, iscr[w2]);
        double sc = 1.0 / sqrt(s2);
        if (z[cand] < 0.0) sc = -sc;
        __syncthreads();
        for (int i = tid; i < nl; i += NT) Yj[bs + i] = z[i] * sc;
        xjm = xj;
        __syncthreads();
    }
    // ---- ascending eigenvalues across the blocks (dsyevx's final selection sort, pairs move together) ----
    if (nb > 1 && mout > 1) {
        for (int i = 0; i < mout - 1; ++i) {
            int k = i;
            double lk = lam[i];
            for (int j = i + 1; j < mout; ++j)
                if (lam[j] < lk) { k = j; lk = lam[j]; }
            __syncthreads();               // everyone has read lam before it changes
            if (k != i) {
                double *Yi = Y + (size_t)i * n, *Yk = Y + (size_t)k * n;
                for (int r = tid; r < n; r += NT) {
                    const double t = Yi[r];
                    Yi[r] = Yk[r];
                    Yk[r] = t;
                }
                if (tid == 0) { lam[k] = lam[i]; lam[i] = lk; }
            }
            __syncthreads();
        }
    }

    if (!do_backtransform) return;  // two-stage path: eig2.hip applies Q1 Q2 and the row scaling
    // ---- C: back-transformation y = H_0 H_1 ... H_{n-2} z, one wavefront per vector ----
    for (int jj = wave; jj < m; jj += NW) {
        double *Yj = Y + (size_t)jj * n;
        for (int k = n - 2; k >= 0; --k) {
            const double tk = tau[k];
            if (tk == 0.0) continue;
            const int len = n - k - 1;
            const double *vk = A + (size_t)k * n + (k + 1);
            double *y = Yj + (k + 1);
            double s = 0.0;
            for (int r = lane; r < len; r += 64) s = fma(vk[r], y[r], s);
            s = wave_sum(s) * tk;
            for (int r = lane; r < len; r += 64) y[r] = fma(-s, vk[r], y[r]);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // next k reads y through other lanes
        }
        // ---- D: x = D^-1/2 y ----
        for (int r = lane; r < n; r += 64) Yj[r] *= dis[vo + r];
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static bool use_one_stage() { return options().eig_dense_one_stage != 0; }

// Persistent, grow-only device workspace shared by all batches (hipMalloc/hipFree of
// multi-GB buffers costs 0.1-0.4 s each on this platform; the arena is allocated once per
// process and reused by every chunk / level / hierarchy).  Single stream, sequential use.
struct EigArena {
    DBuf<double> W, panel, d, e, tau, dis, Tfac, Xbuf, Zbuf, Vpk, Vpk2, trash, rv, rtau, bandg, Gbuf, Xpart, bandsave, subpanels;
    DBuf<int> n, m, j0;
    DBuf<int64_t> moff, voff, roff, goff, xpoff;
};
static int g_slot = 0;   // workspace of the batch being set up (single host thread)
static EigArena &arena() {
    // heap objects that are never destroyed: a static's destructor would call into HIP (dev_free -> hipEventRecord)
    // after main() has returned, when the runtime -- and a profiler hooked into it -- is already torn down
    static EigArena *a = new EigArena[2];
    return a[g_slot];
}
template <class T>
static void arena_view(DBuf<T> &dst, DBuf<T> &pool, size_t need) {
    if (pool.n < need) pool.alloc(need + need / 8 + 64);
    dst.view(pool.p, need);
}
void eig_arena_release() {
    for (g_slot = 0; g_slot < 2; ++g_slot) {
        EigArena &a = arena();
        a = EigArena();
    }
    g_slot = 0;
}
bool eig_uses_two_stage() { return !use_one_stage(); }
double *eig_arena_bandsave(const EigBatch &b, size_t doubles) {
    g_slot = b.slot;
    EigArena &a = arena();
    if (a.bandsave.n < doubles) a.bandsave.alloc(doubles + doubles / 8 + 64);
    return a.bandsave.p;
}

double *eig_arena_subpanels(const EigBatch &b, size_t doubles) {
    g_slot = b.slot;
    EigArena &a = arena();
    if (a.subpanels.n < doubles) a.subpanels.alloc(doubles + doubles / 8 + 64);
    return a.subpanels.p;
}

void eig_batch_alloc(EigBatch &b, const std::vector<int> &sizes, hipStream_t s, int slot) {
    b.slot = g_slot = slot & 1;
    b.count = (int)sizes.size();
    b.h_n = sizes;
    b.has_perm = false;
    b.has_bw = false;
    b.h_moff.assign(b.count + 1, 0);
    b.h_voff.assign(b.count + 1, 0);
    b.max_n = 0;
    for (int i = 0; i < b.count; ++i) {
        const int n = sizes[i];
        SA_REQUIRE(n >= 1, "empty agglomerate matrix");
        b.h_moff[i + 1] = b.h_moff[i] + (int64_t)n * n;
        b.h_voff[i + 1] = b.h_voff[i] + n;
        if (n > b.max_n) b.max_n = n;
    }
    EigArena &a = arena();
    const size_t rows = (size_t)b.h_voff[b.count];
    arena_view(b.n, a.n, (size_t)b.count);
    arena_view(b.moff, a.moff, (size_t)b.count + 1);
    arena_view(b.voff, a.voff, (size_t)b.count + 1);
    SA_HIP_CHECK(hipMemcpyAsync(b.n.p, b.h_n.data(), 4 * (size_t)b.count, hipMemcpyHostToDevice, s));
    SA_HIP_CHECK(hipMemcpyAsync(b.moff.p, b.h_moff.data(), 8 * ((size_t)b.count + 1), hipMemcpyHostToDevice, s));
    SA_HIP_CHECK(hipMemcpyAsync(b.voff.p, b.h_voff.data(), 8 * ((size_t)b.count + 1), hipMemcpyHostToDevice, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    arena_view(b.W, a.W, (size_t)b.h_moff[b.count]);
    arena_view(b.d, a.d, rows);
    arena_view(b.e, a.e, rows);
    arena_view(b.tau, a.tau, rows);
    arena_view(b.dis, a.dis, rows);
    arena_view(b.m, a.m, (size_t)b.count);
    arena_view(b.j0, a.j0, (size_t)b.count);
    if (use_one_stage()) arena_view(b.panel, a.panel, rows * EIG_NB);
}

void eig_batch_two_stage_buffers(EigBatch &b, size_t nrefl, bool need_bandg, hipStream_t s) {
    g_slot = b.slot;
    EigArena &a = arena();
    const size_t rows = (size_t)b.h_voff[b.count];
    arena_view(b.Tfac, a.Tfac, rows * EIG_SB + EIG_SB * EIG_SB);
    arena_view(b.Xbuf, a.Xbuf, rows * EIG_SB);
    arena_view(b.Zbuf, a.Zbuf, rows * EIG_SB);
    arena_view(b.Vpk, a.Vpk, rows * EIG_SB);
    arena_view(b.Vpk2, a.Vpk2, rows * EIG_SB);
    arena_view(b.trash, a.trash, 256);
    arena_view(b.roff, a.roff, (size_t)b.count + 1);
    arena_view(b.goff, a.goff, (size_t)b.count + 1);
    arena_view(b.Gbuf, a.Gbuf, (size_t)b.h_goff[b.count] + 1);
    SA_HIP_CHECK(hipMemcpyAsync(b.goff.p, b.h_goff.data(), 8 * ((size_t)b.count + 1), hipMemcpyHostToDevice, s));
    SA_HIP_CHECK(hipMemcpyAsync(b.roff.p, b.h_roff.data(), 8 * ((size_t)b.count + 1), hipMemcpyHostToDevice, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    arena_view(b.rv, a.rv, nrefl * EIG_SB + EIG_SB);
    arena_view(b.rtau, a.rtau, nrefl + 1);
    if (need_bandg) arena_view(b.bandg, a.bandg, rows * 2 * EIG_SB);
    if (!b.h_xpoff.empty()) {
        arena_view(b.xpoff, a.xpoff, (size_t)b.count + 1);
        arena_view(b.Xpart, a.Xpart, (size_t)b.h_xpoff[b.count] * 64 * EIG_SB + 64);
        SA_HIP_CHECK(hipMemcpyAsync(b.xpoff.p, b.h_xpoff.data(), 8 * ((size_t)b.count + 1), hipMemcpyHostToDevice, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
}

static size_t tri_lds_bytes(int n) { return sizeof(double) * (2 * (size_t)n + TRI_NT + 2 * EIG_NB + TRI_NT / 64); }
static size_t vec_lds_bytes(int n) {
    return sizeof(double) * (7 * (size_t)n + 8) + sizeof(int) * 8 + sizeof(unsigned short) * ((size_t)n + 3 * (VEC_MAXB + 2)) +
           (size_t)n + 64;
}
constexpr size_t LDS_MAX = 160 * 1024;

// ---------------------------------------------------------------------------------------
// Duplicate agglomerate matrices (round 4)
// ---------------------------------------------------------------------------------------
// On a structured mesh with piecewise constant coefficients most agglomerates are translates of one another: their
// scaled matrices C, scalings D and row orders are IDENTICAL bit for bit (the 256^3 Poisson problem: 65 536 agglomerates
// on the fine level, a few hundred distinct ones; the same on the coarse levels, whose element matrices are built from
// identical eigenvectors by order-preserving sums).  The eigenpairs of a matrix are a function of those bits alone -- no
// kernel of the few-eigenpairs or the dense path looks at a matrix's position in its batch, the start vectors are seeded
// by (row, n) -- so one member of every class is solved and the others receive copies: the hierarchy is the one the
// per-agglomerate computation builds.  Classes are found by a 128-bit hash of everything the eigensolvers read (n, the
// half bandwidth, the band of C in both triangles, D^-1/2, the row order, the coarse start vector) and CONFIRMED by a
// word-by-word comparison with the class representative; a batch with fewer than a quarter of duplicates is left alone
// (variable coefficients, unstructured meshes: the cost is the hash, one pass over the bands).
// saamge_amd_options.eig_dedupe = 0 switches it off.
__device__ inline unsigned long long dd_mix(unsigned long long x) {      // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// The words a matrix consists of, in a fixed order (DdSource, eig.h).  kind 1, the assembled matrix: the band of C in both
// triangles (column by column, 2 bw + 1 slots per column, slots outside the matrix = 0), D^-1/2, the row order, the coarse
// start vector, n and bw.  kind 0, the sparse rows the fused fine-level assembly builds the matrix from: values, columns,
// the row order, n.
__device__ inline long dd_count(const DdSource &v, int b) {
    const long n = v.ns[b];
    if (v.kind == 0) return 2 * n * v.RW + n + 1;
    const long bw = min(v.bws[b], (int)n - 1);
    return n * (2 * bw + 1) + 3 * n + 2;
}
__device__ inline unsigned long long dd_word(const DdSource &v, int b, long idx) {
    const long n = v.ns[b];
    const int64_t vo = v.voff[b];
    if (v.kind == 0) {
        const long nr = n * v.RW;
        if (idx < nr) return (unsigned long long)__double_as_longlong(v.rvals[(size_t)vo * v.RW + idx]);
        if (idx < 2 * nr) return (unsigned long long)(unsigned short)v.rcols[(size_t)vo * v.RW + (idx - nr)];
        if (idx < 2 * nr + n) return v.perm ? (unsigned long long)(unsigned short)v.perm[vo + (idx - 2 * nr)] : 0ull;
        return (unsigned long long)n;
    }
    const long bw = min(v.bws[b], (int)n - 1), w2 = 2 * bw + 1, nb = n * w2;
    if (idx < nb) {
        const long j = idx / w2, i = j - bw + (idx - j * w2);
        return (i >= 0 && i < n) ? (unsigned long long)__double_as_longlong(v.W[v.moff[b] + (size_t)j * n + i]) : 0ull;
    }
    idx -= nb;
    if (idx < n) return (unsigned long long)__double_as_longlong(v.dis[vo + idx]);
    if (idx < 2 * n) return v.perm ? (unsigned long long)(unsigned short)v.perm[vo + (idx - n)] : 0ull;
    if (idx < 3 * n) return v.x0c ? (unsigned long long)__double_as_longlong(v.x0c[vo + (idx - 2 * n)]) : 0ull;
    return idx == 3 * n ? (unsigned long long)n : (unsigned long long)bw;
}
// kind 0, the two long arrays of a matrix (values, then columns): four words per thread and trip, their loads requested
// together (dd_word's chain of branches kept one load in flight per thread).  fn(word, idx) sees every word of [0, 2 nr).
template <class F>
__device__ inline void dd_rows_words(const DdSource &v, int b, long start, long stride, F fn) {
    const long nr = (long)v.ns[b] * v.RW;
    const double *rv = v.rvals + (size_t)v.voff[b] * v.RW;
    const short *rc = v.rcols + (size_t)v.voff[b] * v.RW;
    for (long i0 = start; i0 < nr; i0 += 4 * stride) {
        double x[4];
        short c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = min(i0 + u * stride, nr - 1);
            x[u] = rv[i];
            c[u] = rc[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * stride < nr) {
                fn((unsigned long long)__double_as_longlong(x[u]), i0 + u * stride);
                fn((unsigned long long)(unsigned short)c[u], nr + i0 + u * stride);
            }
    }
}
// grid (matrices, y): partial 128-bit sums of the mixed (word, position) pairs, added into out[2 b], out[2 b + 1] (zeroed)
__global__ __launch_bounds__(256) void dd_hash_kernel(DdSource v, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long cnt = dd_count(v, b);
    unsigned long long h1 = 0, h2 = 0;
    auto add = [&](unsigned long long w, long idx) {
        const unsigned long long k = dd_mix(w + 0x9E3779B97F4A7C15ull * (unsigned long long)(idx + 1));
        h1 += k;      // (sums: independent of the order the words are visited in)
        h2 += (k >> 32) * (k & 0xffffffffull);      // (second sum: the product of the halves of the mixed word; a full second mix was half of the kernel)
    };
    long first = 0;
    if (v.kind == 0) {
        dd_rows_words(v, b, (long)blockIdx.y * 256 + tid, 256l * gridDim.y, add);
        first = 2l * v.ns[b] * v.RW;      // (the short tail -- row order, n -- through dd_word)
    }
    for (long idx = first + (long)blockIdx.y * 256 + tid; idx < cnt; idx += 256l * gridDim.y) add(dd_word(v, b, idx), idx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = h1; red[1][tid >> 6] = h2; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(out + 2 * (size_t)b, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(out + 2 * (size_t)b + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}
// every matrix against the first of its class (rep[b] = that matrix): differ[b] = 1 unless every word is the same
__global__ __launch_bounds__(256) void dd_verify_kernel(DdSource v, const int *__restrict__ rep, int *__restrict__ differ) {
    const int b = blockIdx.x, r0 = rep[b], tid = threadIdx.x;
    if (r0 == b) return;
    const long cnt = dd_count(v, b);
    if (cnt != dd_count(v, r0)) { if (tid == 0) differ[b] = 1; return; }
    int bad = 0;
    long first = 0;
    if (v.kind == 0) {      // (the two long arrays four words at a time, both matrices' loads in flight together)
        const long nr = (long)v.ns[b] * v.RW, stride = 256l * gridDim.y;
        const double *rv = v.rvals + (size_t)v.voff[b] * v.RW, *rv0 = v.rvals + (size_t)v.voff[r0] * v.RW;
        const short *rc = v.rcols + (size_t)v.voff[b] * v.RW, *rc0 = v.rcols + (size_t)v.voff[r0] * v.RW;
        for (long i0 = (long)blockIdx.y * 256 + tid; i0 < nr; i0 += 4 * stride) {
            long long x[4], y[4];
            short c[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = min(i0 + u * stride, nr - 1);
                x[u] = __double_as_longlong(rv[i]);
                y[u] = __double_as_longlong(rv0[i]);
                c[u] = rc[i];
                d[u] = rc0[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) bad |= (x[u] != y[u]) | (c[u] != d[u]);
        }
        first = 2 * nr;
    }
    for (long idx = first + (long)blockIdx.y * 256 + tid; idx < cnt; idx += 256l * gridDim.y) bad |= dd_word(v, b, idx) != dd_word(v, r0, idx);
    if (bad) differ[b] = 1;
}
// list[q] = a matrix of the batch, blobs[q] = where its words go / what they are compared with (differ[q] = 1: not the same)
__global__ __launch_bounds__(256) void dd_pack_kernel(DdSource v, const int *__restrict__ list, unsigned long long *const *__restrict__ blobs) {
    const int b = list[blockIdx.x];
    const long cnt = dd_count(v, b);
    unsigned long long *out = blobs[blockIdx.x];
    for (long idx = (long)blockIdx.y * 256 + threadIdx.x; idx < cnt; idx += 256l * gridDim.y) out[idx] = dd_word(v, b, idx);
}
__global__ __launch_bounds__(256) void dd_compare_kernel(DdSource v, const int *__restrict__ list, const unsigned long long *const *__restrict__ blobs,
                                                         const long *__restrict__ blob_words, int *__restrict__ differ) {
    const int b = list[blockIdx.x];
    const long cnt = dd_count(v, b);
    if (cnt != blob_words[blockIdx.x]) { if (threadIdx.x == 0) differ[blockIdx.x] = 1; return; }
    const unsigned long long *ref = blobs[blockIdx.x];
    int bad = 0;
    for (long idx = (long)blockIdx.y * 256 + threadIdx.x; idx < cnt; idx += 256l * gridDim.y) bad |= dd_word(v, b, idx) != ref[idx];
    if (bad) differ[blockIdx.x] = 1;
}
__global__ void gather_int_kernel(int n, const int *__restrict__ idx, const int *__restrict__ src, int *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// results to every member of the classes: block i copies the eigenvalues and eigenvectors its class was given
__global__ __launch_bounds__(256) void dedupe_expand_kernel(const double *const *__restrict__ src_evals, const double *const *__restrict__ src_evecs,
                                                            const int64_t *__restrict__ eoff, const int64_t *__restrict__ xoff,
                                                            double *__restrict__ evals, double *__restrict__ evecs) {
    const int i = blockIdx.x;
    const int64_t ne = eoff[i + 1] - eoff[i], nx = xoff[i + 1] - xoff[i];
    const double *se = src_evals[i], *sx = src_evecs[i];
    if (blockIdx.y == 0)
        for (int64_t t = threadIdx.x; t < ne; t += 256) evals[eoff[i] + t] = se[t];
    for (int64_t t = blockIdx.y * 256 + threadIdx.x; t < nx; t += 256 * (int64_t)gridDim.y) evecs[xoff[i] + t] = sx[t];
}
void eig_dedupe_expand(hipStream_t s, int count, int max_n, const double *const *src_evals, const double *const *src_evecs,
                       const int64_t *eoff, const int64_t *xoff, double *evals, double *evecs) {
    if (!count) return;
    const int ny = std::max(1, std::min(64, std::min(max_n / 128, 65536 / std::max(1, count))));
    hipLaunchKernelGGL(dedupe_expand_kernel, dim3(count, ny), dim3(256), 0, s, src_evals, src_evecs, eoff, xoff, evals, evecs);
    SA_HIP_CHECK(hipGetLastError());
}

// rep[i] = the first matrix with matrix i's 128-bit hash (hashes: two words per matrix); returns the number of classes
int eig_dedupe_group(const unsigned long long *hh, int count, std::vector<int> &rep) {
    std::unordered_map<DdKey, int, DdKeyHash> first;
    first.reserve((size_t)count / 8 + 16);
    rep.resize((size_t)count);
    int nuniq = 0;
    for (int i = 0; i < count; ++i) {
        auto it = first.emplace(DdKey{hh[2 * (size_t)i], hh[2 * (size_t)i + 1]}, i);
        rep[i] = it.first->second;
        nuniq += it.second ? 1 : 0;
    }
    return nuniq;
}
static int dd_grid_y(const DdSource &src, int count, int max_n) {      // workgroups per matrix: few large matrices need several
    const long words = src.kind == 0 ? 2l * max_n * src.RW : (long)max_n * std::min(2 * max_n, 2048);
    return (int)std::max(1l, std::min(std::min(64l, words / 65536), 4096l / std::max(1, count)));
}

bool eig_dedupe_find(hipStream_t s, const DdSource &src, int count, int max_n, DdClasses &out) {
    out.reps.clear();
    out.rep_of.clear();
    out.rep_hash.clear();
    if (count < 16) return false;
    profiler().begin(s);
    const int ny = dd_grid_y(src, count, max_n);
    DBuf<unsigned long long> hash(2 * (size_t)count);
    hash.zero(s);
    hipLaunchKernelGGL(dd_hash_kernel, dim3(count, ny), dim3(256), 0, s, src, hash.p);
    SA_HIP_CHECK(hipGetLastError());
    auto hh = hash.to_host(s);
    std::vector<int> rep;
    const int nuniq = eig_dedupe_group(hh.data(), count, rep);
    if ((long)nuniq * 4 > (long)count * 3) { profiler().end(s, "eig_dedupe", 0.0, 0.0); return false; }
    DBuf<int> d_rep, differ((size_t)count);
    d_rep.from_host(rep, s);
    differ.zero(s);
    hipLaunchKernelGGL(dd_verify_kernel, dim3(count, ny), dim3(256), 0, s, src, d_rep.p, differ.p);
    SA_HIP_CHECK(hipGetLastError());
    auto hd = differ.to_host(s);
    for (int i = 0; i < count; ++i)
        if (hd[i]) rep[i] = i;      // (a collision of the hash: the matrix stands for itself)
    std::vector<int> pos((size_t)count, -1);
    for (int i = 0; i < count; ++i)
        if (rep[i] == i) {
            pos[i] = (int)out.reps.size();
            out.reps.push_back(i);
            out.rep_hash.push_back(hh[2 * (size_t)i]);
            out.rep_hash.push_back(hh[2 * (size_t)i + 1]);
        }
    out.rep_of.resize((size_t)count);
    for (int i = 0; i < count; ++i) out.rep_of[i] = pos[rep[i]];
    profiler().end(s, "eig_dedupe", 0.0, 0.0);
    if ((options().debug & 1)) std::fprintf(stderr, "duplicate agglomerates (%s): %d distinct of %d\n", src.kind ? "bands" : "sparse rows", (int)out.reps.size(), count);
    return true;
}
// the hashes of SOME matrices of the batch (two words each, in the order of the list)
__global__ __launch_bounds__(256) void dd_hash_list_kernel(DdSource v, const int *__restrict__ list, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[2][4];
    const int b = list[blockIdx.x], tid = threadIdx.x;
    const long cnt = dd_count(v, b);
    unsigned long long h1 = 0, h2 = 0;
    for (long idx = (long)blockIdx.y * 256 + tid; idx < cnt; idx += 256l * gridDim.y) {
        const unsigned long long k = dd_mix(dd_word(v, b, idx) + 0x9E3779B97F4A7C15ull * (unsigned long long)(idx + 1));
        h1 += k;
        h2 += (k >> 32) * (k & 0xffffffffull);      // (second sum: the product of the halves of the mixed word; a full second mix was half of the kernel)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = h1; red[1][tid >> 6] = h2; }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(out + 2 * (size_t)blockIdx.x, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(out + 2 * (size_t)blockIdx.x + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}
std::vector<unsigned long long> eig_dedupe_hash_list(hipStream_t s, const DdSource &src, int max_n, const std::vector<int> &list) {
    std::vector<unsigned long long> out(2 * list.size(), 0);
    if (list.empty()) return out;
    DBuf<int> d_list;
    d_list.from_host(list, s);
    DBuf<unsigned long long> hash(2 * list.size());
    hash.zero(s);
    hipLaunchKernelGGL(dd_hash_list_kernel, dim3((unsigned)list.size(), dd_grid_y(src, (int)list.size(), max_n)), dim3(256), 0, s, src, d_list.p, hash.p);
    SA_HIP_CHECK(hipGetLastError());
    auto hh = hash.to_host(s);
    std::copy(hh.begin(), hh.end(), out.begin());
    return out;
}
// the words of some matrices of the batch, kept for comparisons with matrices of later batches
void eig_dedupe_pack(hipStream_t s, const DdSource &src, int max_n, const std::vector<int> &list, const std::vector<long> &words,
                     std::vector<DBuf<unsigned long long>> &blobs) {
    blobs.clear();
    blobs.resize(list.size());
    if (list.empty()) return;
    std::vector<unsigned long long *> ptrs(list.size());
    for (size_t q = 0; q < list.size(); ++q) { blobs[q].alloc((size_t)words[q]); ptrs[q] = blobs[q].p; }
    DBuf<int> d_list;
    DBuf<unsigned long long *> d_ptrs;
    d_list.from_host(list, s);
    d_ptrs.from_host(ptrs, s);
    hipLaunchKernelGGL(dd_pack_kernel, dim3((unsigned)list.size(), dd_grid_y(src, (int)list.size(), max_n)), dim3(256), 0, s, src, d_list.p, d_ptrs.p);
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));
}
// same[q] = matrix list[q] of the batch consists of exactly the words blobs[q]
void eig_dedupe_compare(hipStream_t s, const DdSource &src, int max_n, const std::vector<int> &list,
                        const std::vector<const unsigned long long *> &blobs, const std::vector<long> &blob_words, std::vector<char> &same) {
    same.assign(list.size(), 0);
    if (list.empty()) return;
    DBuf<int> d_list, differ(list.size());
    DBuf<const unsigned long long *> d_ptrs;
    DBuf<long> d_words;
    d_list.from_host(list, s);
    d_ptrs.from_host(blobs, s);
    d_words.from_host(blob_words, s);
    differ.zero(s);
    hipLaunchKernelGGL(dd_compare_kernel, dim3((unsigned)list.size(), dd_grid_y(src, (int)list.size(), max_n)), dim3(256), 0, s, src, d_list.p, d_ptrs.p,
                       d_words.p, differ.p);
    SA_HIP_CHECK(hipGetLastError());
    auto hd = differ.to_host(s);
    for (size_t q = 0; q < list.size(); ++q) same[q] = hd[q] ? 0 : 1;
}
// word counts of some matrices (host): the half bandwidths come from the device
std::vector<long> eig_dedupe_words(hipStream_t s, const DdSource &src, const std::vector<int> &h_n, const std::vector<int> &list) {
    std::vector<long> w(list.size());
    hvec<int> hb;
    if (src.kind == 1) { DBuf<int> tmp; tmp.view(const_cast<int *>(src.bws), h_n.size()); hb = tmp.to_host(s); }
    for (size_t q = 0; q < list.size(); ++q) {
        const long n = h_n[list[q]];
        if (src.kind == 0) w[q] = 2 * n * src.RW + n + 1;
        else { const long bw = std::min((long)hb[list[q]], n - 1); w[q] = n * (2 * bw + 1) + 3 * n + 2; }
    }
    return w;
}
DdSource eig_dedupe_source(const EigBatch &b) {
    DdSource v{};
    v.kind = 1;
    v.ns = b.n.p; v.moff = b.moff.p; v.voff = b.voff.p; v.W = b.W.p; v.bws = b.bw.p; v.dis = b.dis.p;
    v.perm = b.has_perm ? b.perm.p : nullptr;
    v.x0c = b.has_x0c ? b.x0c.p : nullptr;
    return v;
}

// the batch of the class representatives: the same workspace, per-matrix tables of its own
void eig_batch_compact(hipStream_t s, EigBatch &cb, EigBatch &full, const std::vector<int> &reps) {
    cb = EigBatch();
    cb.slot = g_slot = full.slot;
    cb.count = (int)reps.size();
    cb.h_n.resize(reps.size());
    cb.h_moff.assign(reps.size() + 1, 0);
    cb.h_voff.assign(reps.size() + 1, 0);
    cb.max_n = 0;
    for (size_t i = 0; i < reps.size(); ++i) {
        cb.h_n[i] = full.h_n[reps[i]];
        cb.h_moff[i] = full.h_moff[reps[i]];
        cb.h_voff[i] = full.h_voff[reps[i]];
        cb.max_n = std::max(cb.max_n, cb.h_n[i]);
    }
    cb.h_moff[reps.size()] = full.h_moff[full.count];      // (the totals size the shared buffers)
    cb.h_voff[reps.size()] = full.h_voff[full.count];
    cb.n.from_host(cb.h_n, s);
    cb.moff.from_host(cb.h_moff, s);
    cb.voff.from_host(cb.h_voff, s);
    cb.W.view(full.W.p, full.W.n);
    cb.d.view(full.d.p, full.d.n);
    cb.e.view(full.e.p, full.e.n);
    cb.tau.view(full.tau.p, full.tau.n);
    cb.dis.view(full.dis.p, full.dis.n);
    if (full.panel.n) cb.panel.view(full.panel.p, full.panel.n);
    cb.m.alloc((size_t)cb.count);
    cb.j0.alloc((size_t)cb.count);
    cb.has_perm = full.has_perm;
    if (full.perm.n) cb.perm.view(full.perm.p, full.perm.n);
    if (full.iperm.n) cb.iperm.view(full.iperm.p, full.iperm.n);
    cb.has_bw = full.has_bw;
    if (full.has_bw && cb.count) {
        DBuf<int> idx;
        idx.from_host(reps, s);
        cb.bw.alloc((size_t)cb.count);
        hipLaunchKernelGGL(gather_int_kernel, dim3(div_up(cb.count, 256)), dim3(256), 0, s, cb.count, idx.p, full.bw.p, cb.bw.p);
        SA_HIP_CHECK(hipGetLastError());
        SA_HIP_CHECK(hipStreamSynchronize(s));      // (idx is freed on return)
    }
    cb.has_x0c = full.has_x0c;
    if (full.has_x0c) cb.x0c.view(full.x0c.p, full.x0c.n);
    cb.has_window = full.has_window;
    cb.window_vu = full.window_vu;
    cb.vl = full.vl;
    cb.vu = full.vu;
    cb.ss_tol = full.ss_tol;
    cb.dense_only = full.dense_only;
}

size_t eig_workspace_bytes(int n) {
    const size_t nn = (size_t)n;
    // (the packed sub-panels of the blocked wide-band factorisation: only matrices that can have a band beyond the LDS window)
    const size_t sub = nn > 512 ? 16 * EIG_SB : 0;
    return 8 * (nn * nn + nn * (EIG_NB + 8) + nn * nn / 2 + nn * (3 * EIG_SB + 2 * EIG_SB + sub) + 64);
}

// The few-eigenpairs path (eig2.hip) is the default for batches whose largest agglomerate has at
// least SAAMGE_AMD_SS_MIN_N (64) rows -- measured faster than the dense reduction on the 405-row and
// on the 2 600-row agglomerates of the headline problem; a batch it cannot handle (more than six
// wanted pairs, no convergence) is redone by the dense path.  SAAMGE_AMD_EIG=twostage / onestage /
// dense switches it off.
bool eig_use_subspace() { return true; }      // (the dense path alone: EigBatch::dense_only, saamge_amd_params.eigensolver = 1)

bool eig_batch_takes_subspace(const EigBatch &b) {
    // (saamge_amd_options.eig_min_n: smallest agglomerate size of a batch that takes the few-eigenpairs path)
    return !b.dense_only && b.max_n >= options().eig_min_n;
}

void eig_tridiagonalize(hipStream_t s, EigBatch &b, int phases) {
    if (!b.count) return;
    if (eig_batch_takes_subspace(b)) {
        // few-eigenpairs path: "phase 1" is the Cholesky factorisation, phase 2 has nothing to do
        if (phases & 1) {
            b.subspace = true;
            b.ss_failed = !eig_subspace_factor(s, b);
        }
        return;
    }
    b.subspace = false;
    static bool attr0 = false;
    if (!attr0) {
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)eigvec_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)count_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        attr0 = true;
    }
    if (!use_one_stage()) {
        b.two_stage = true;
        eig_tridiagonalize_two_stage(s, b, phases);
        return;
    }
    b.two_stage = false;
    if (!(phases & 1)) return;   // one-stage: everything happens in "phase 1"
    const size_t lds = tri_lds_bytes(b.max_n);
    SA_REQUIRE(lds <= LDS_MAX, "agglomerate too large for the LDS-resident reflector vectors");
    static bool attr_set = false;
    if (!attr_set) {
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)tridiag_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)eigvec_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)count_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
        attr_set = true;
    }
    double flops = 0.0, bytes = 0.0;
    for (int n : b.h_n) {
        flops += 4.0 / 3.0 * (double)n * n * n;
        bytes += 8.0 * (double)n * n;
    }
    profiler().begin(s);
    hipLaunchKernelGGL(tridiag_kernel, dim3(b.count), dim3(TRI_NT), lds, s, b.n.p, b.moff.p,
                       b.voff.p, b.W.p, b.panel.p, b.d.p, b.e.p, b.tau.p);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "eig_tridiag", bytes, flops);
}

void eig_count(hipStream_t s, EigBatch &b, double vl, double vu) {
    if (!b.count) return;
    if (b.subspace) {
        // (vl = -1 < every eigenvalue of the semidefinite C: the window is lambda <= vu)
        if (!b.ss_failed) b.ss_failed = !eig_subspace_iterate(s, b, vu);
        if (b.ss_failed) b.h_m.assign((size_t)b.count, 1);     // the caller redoes the batch densely
        return;
    }
    const size_t lds = sizeof(double) * 2 * (size_t)b.max_n;
    SA_REQUIRE(lds <= LDS_MAX, "agglomerate too large for the Sturm kernel");
    profiler().begin(s);
    hipLaunchKernelGGL(count_kernel, dim3(b.count), dim3(64), lds, s, b.n.p, b.voff.p, b.d.p,
                       b.e.p, vl, vu, b.m.p, b.j0.p);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "eig_count", 0.0, 0.0);
    b.vl = vl;
    b.vu = vu;
    { auto t_ = b.m.to_host(s); b.h_m.assign(t_.begin(), t_.end()); }
}

void eig_vectors(hipStream_t s, EigBatch &b, const int64_t *eoff, const int64_t *xoff,
                 double *evals, double *evecs) {
    if (!b.count) return;
    if (b.subspace) {
        SA_REQUIRE(!b.ss_failed, "subspace eigensolver failed and was not redone densely");
        eig_subspace_vectors(s, b, eoff, xoff, evals, evecs);
        return;
    }
    const size_t lds = vec_lds_bytes(b.max_n);
    double flops = 0.0;
    for (int i = 0; i < b.count; ++i) flops += 2.0 * (double)b.h_n[i] * b.h_n[i] * b.h_m[i];
    DBuf<double> gws;
    if (lds > LDS_MAX) {      // work arrays in global memory
        SA_REQUIRE(b.max_n < 65536, "agglomerate too large for the inverse iteration's 16-bit block tables");
        gws.alloc((size_t)b.count * (lds / 8 + 2));
    }
    profiler().begin(s);
    if (lds > LDS_MAX)
        hipLaunchKernelGGL(eigvec_kernel<true>, dim3(b.count), dim3(VEC_NT), 64, s, b.n.p, b.moff.p,
                           b.voff.p, b.W.p, b.d.p, b.e.p, b.tau.p, b.dis.p, b.m.p, b.j0.p, eoff, xoff,
                           evals, evecs, b.two_stage ? 0 : 1, b.vl, b.vu, gws.p, (int64_t)(lds / 8 + 2));
    else
        hipLaunchKernelGGL(eigvec_kernel<false>, dim3(b.count), dim3(VEC_NT), lds, s, b.n.p, b.moff.p,
                           b.voff.p, b.W.p, b.d.p, b.e.p, b.tau.p, b.dis.p, b.m.p, b.j0.p, eoff, xoff,
                           evals, evecs, b.two_stage ? 0 : 1, b.vl, b.vu, (double *)nullptr, (int64_t)0);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "eig_vectors", 0.0, flops);
    if (b.two_stage) eig_backtransform_two_stage(s, b, xoff, evecs);
    if (gws.p) SA_HIP_CHECK(hipStreamSynchronize(s));     // (the workspace is released on return)
}

}  // namespace saamge_amd
