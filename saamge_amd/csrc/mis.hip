// MIS stage on gfx950: one wavefront per minimal intersection set gathers the
// MIS-restricted agglomerate eigenvectors, filters essential rows, normalises the columns
// and orthogonalises them with a one-sided (Hestenes) Jacobi SVD -- all cross-lane
// reductions are wavefront shuffles, no LDS traffic besides the tiny sort scratch.
#include "mis.h"
#include "assemble.h"

#include <cfloat>

namespace saamge_amd {

__device__ inline double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(64) void mis_svd_kernel(
    const int *__restrict__ mis2d_I, const int *__restrict__ mis2d_J,
    const int *__restrict__ mis2ae_I, const int *__restrict__ mis2ae_J,
    const int *__restrict__ ae2d_I, const int64_t *__restrict__ pair_loc_off,
    const int *__restrict__ pair_loc, const signed char *__restrict__ flags, MisSvdIO io, int m0,
    const int *__restrict__ list = nullptr) {
    extern __shared__ __align__(16) double lds[];  // sig[ctot], then perm[ctot] (ints)
    const int m = list ? list[blockIdx.x] : m0 + (int)blockIdx.x;      // (list: the first members of the classes of identical MISes)
    const int lane = threadIdx.x;
    const int r = mis2d_I[m + 1] - mis2d_I[m];
    const int *dofs = mis2d_J + mis2d_I[m];
    double *M = io.gather + io.g_off[m];
    double *Uout = io.U + io.u_off[m];
    double *sigout = io.sig + io.s_off[m];
    const int ctot = (int)(io.s_off[m + 1] - io.s_off[m]);
    double *sg = lds;
    int *perm = (int *)(lds + ctot);

    // skip MISes whose dofs are all on the essential boundary (contrib.cpp:578-605)
    if (io.avoid_ess) {
        int interior = 0;
        for (int i = lane; i < r; i += 64) interior |= !(flags[dofs[i]] & FLAG_ON_ESS_BORDER);
        if (__ballot(interior) == 0ull) {
            if (lane == 0) { io.k[m] = 0; io.ncols[m] = 0; }
            return;
        }
    }
    if (r == 1) {  // contrib.cpp:607-612
        if (lane == 0) {
            Uout[0] = 1.0;
            io.k[m] = 1;
            io.ncols[m] = 1;
            if (ctot > 0) sigout[0] = 1.0;
        }
        return;
    }
    // gather + boundary filter + normalisation (contrib.cpp:102-163, xpacks.cpp:537-559).
    // Tall blocks (ctot <= r) are stored column-major r x c and orthogonalised by columns; wide
    // blocks (more candidate vectors than dofs: edge/vertex MISes, many eigenvectors per AE) are
    // stored row-major with leading dimension ctot and orthogonalised by ROWS (Jacobi on M^T,
    // r(r-1)/2 pairs instead of c(c-1)/2), accumulating the rotations: M^T V = W Sigma, so the
    // left singular vectors of M are the columns of V.
    const bool wide = ctot > r;
    const size_t sr = wide ? (size_t)ctot : 1, sc = wide ? 1 : (size_t)r;  // strides of M(i, j)
    int c = 0;
    for (int q = mis2ae_I[m]; q < mis2ae_I[m + 1]; ++q) {
        const int ae = mis2ae_J[q];
        const int na = ae2d_I[ae + 1] - ae2d_I[ae];
        const int ma = io.ae_m[ae];
        const double *X = io.evecs + io.ae_xoff[ae];
        const int *loc = pair_loc + pair_loc_off[q];
        for (int v = 0; v < ma; ++v) {
            double *col = M + (size_t)c * sc;
            double ss = 0.0;
            int nz = 0;
            for (int i = lane; i < r; i += 64) {
                double x = X[(size_t)v * na + loc[i]];
                if (io.avoid_ess && (flags[dofs[i]] & FLAG_ON_ESS_BORDER)) x = 0.0;
                nz |= (x != 0.0);
                ss = fma(x, x, ss);
                col[i * sr] = x;
            }
            if (__ballot(nz) == 0ull) continue;  // entirely zero column: ignored
            const double nrm = sqrt(wsum(ss));
            if (nrm <= 1e-10) continue;          // SA_REAL_ALMOST_LE(norm, 0.)
            const double scl = 1.0 / nrm;
            for (int i = lane; i < r; i += 64) col[i * sr] *= scl;
            ++c;
        }
    }
    for (int q = 0; q < io.nextra; ++q) {   // appended modes, same filter / normalisation
        const double *X = io.extra + (size_t)q * io.ND;
        double *col = M + (size_t)c * sc;
        double ss = 0.0;
        int nz = 0;
        for (int i = lane; i < r; i += 64) {
            double x = X[dofs[i]];
            if (io.avoid_ess && (flags[dofs[i]] & FLAG_ON_ESS_BORDER)) x = 0.0;
            nz |= (x != 0.0);
            ss = fma(x, x, ss);
            col[i * sr] = x;
        }
        if (__ballot(nz) == 0ull) continue;
        const double nrm = sqrt(wsum(ss));
        if (nrm <= 1e-10) continue;
        const double scl = 1.0 / nrm;
        for (int i = lane; i < r; i += 64) col[i * sr] *= scl;
        ++c;
    }
    if (lane == 0) io.ncols[m] = c;
    if (c == 0) {
        if (lane == 0) io.k[m] = 0;
        return;
    }
    // one-sided Jacobi: rotate vector pairs until mutually orthogonal
    const int nv = wide ? r : c;       // vectors being orthogonalised
    const int len = wide ? c : r;      // their length
    const size_t ldv = wide ? (size_t)ctot : (size_t)r;
    double *V = Uout;                  // wide only: r x r accumulated rotations
    if (wide)
        for (int idx = lane; idx < r * r; idx += 64) V[idx] = (idx / r == idx % r) ? 1.0 : 0.0;
    const double tol = DBL_EPSILON * sqrt((double)len);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < nv - 1; ++p) {
            double *ap = M + (size_t)p * ldv;
            for (int q = p + 1; q < nv; ++q) {
                double *aq = M + (size_t)q * ldv;
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = lane; i < len; i += 64) {
                    const double x = ap[i], y = aq[i];
                    al = fma(x, x, al);
                    be = fma(y, y, be);
                    ga = fma(x, y, ga);
                }
                al = wsum(al); be = wsum(be); ga = wsum(ga);
                if (al == 0.0 || be == 0.0) continue;
                if (fabs(ga) <= tol * sqrt(al * be)) continue;
                rotated = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = lane; i < len; i += 64) {
                    const double x = ap[i], y = aq[i];
                    ap[i] = cs * x - sn * y;
                    aq[i] = sn * x + cs * y;
                }
                if (wide) {
                    double *vp = V + (size_t)p * r, *vq = V + (size_t)q * r;
                    for (int i = lane; i < r; i += 64) {
                        const double x = vp[i], y = vq[i];
                        vp[i] = cs * x - sn * y;
                        vq[i] = sn * x + cs * y;
                    }
                }
            }
        }
        if (!rotated) break;
    }
    // singular values = vector norms; order descending (stable)
    for (int j = 0; j < nv; ++j) {
        const double *aj = M + (size_t)j * ldv;
        double ss = 0.0;
        for (int i = lane; i < len; i += 64) ss = fma(aj[i], aj[i], ss);
        ss = wsum(ss);
        if (lane == 0) sg[j] = sqrt(ss);
    }
    __syncthreads();
    for (int j = lane; j < nv; j += 64) {
        const double sj = sg[j];
        int rank = 0;
        for (int i = 0; i < nv; ++i) rank += (sg[i] > sj) || (sg[i] == sj && i < j);
        perm[rank] = j;
    }
    __syncthreads();
    const double s0 = sg[perm[0]];
    const int kmax = min(r, c);  // dgesvd returns min(m, n) singular triplets
    int k = 0;
    while (k < kmax && sg[perm[k]] > 1e-10 * s0) ++k;  // xpack_orth_set
    for (int j = lane; j < ctot; j += 64) sigout[j] = (j < nv) ? sg[perm[j]] : 0.0;
    if (wide) {
        // V sits in the output buffer: park it in the (now free) gather block, then write the
        // kept columns in descending-sigma order
        for (int idx = lane; idx < r * r; idx += 64) M[idx] = V[idx];
        __syncthreads();
        for (int t = 0; t < k; ++t)
            for (int i = lane; i < r; i += 64) Uout[(size_t)t * r + i] = M[(size_t)perm[t] * r + i];
    } else {
        for (int t = 0; t < k; ++t) {
            const double *aj = M + (size_t)perm[t] * r;
            const double inv = 1.0 / sg[perm[t]];
            for (int i = lane; i < r; i += 64) Uout[(size_t)t * r + i] = aj[i] * inv;
        }
    }
    if (lane == 0) io.k[m] = k;
}

// Classes of identical MISes (eig.hip, "Duplicate agglomerate matrices", carried to the MIS stage): what the kernel above
// gathers is a function of the MIS's size, the essential-boundary bits of its dofs, and per agglomerate it belongs to the
// agglomerate's eigenvector block -- identical, bit for bit, for the members of one class of agglomerates (they received
// copies of the class's eigenpairs: ae_ev = the class per agglomerate, -1 = on its own) -- its size and the MIS's rows in the
// agglomerate's numbering; plus the appended modes on its dofs.  mis_walk visits those words for one MIS (a wavefront per
// MIS), summing mixed (word, position) pairs or comparing with the first member of the class in lockstep.
struct MisIn {
    const int *mis2d_I, *mis2d_J, *mis2ae_I, *mis2ae_J, *ae2d_I;
    const int64_t *pair_loc_off;
    const int *pair_loc;
    const signed char *flags;
    MisSvdIO io;
    const int *ae_ev;
};
__device__ inline unsigned long long mi_mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <bool PAIR, class F>
__device__ inline bool mis_walk(const MisIn &v, int m, int m2, int lane, F &&f) {
    const int r = v.mis2d_I[m + 1] - v.mis2d_I[m], qb = v.mis2ae_I[m], nq = v.mis2ae_I[m + 1] - qb;
    const int ctot = (int)(v.io.s_off[m + 1] - v.io.s_off[m]);
    const int *dofs = v.mis2d_J + v.mis2d_I[m];
    const int qb2 = PAIR ? v.mis2ae_I[m2] : 0;
    const int *dofs2 = PAIR ? v.mis2d_J + v.mis2d_I[m2] : nullptr;
    if (PAIR && (r != v.mis2d_I[m2 + 1] - v.mis2d_I[m2] || nq != v.mis2ae_I[m2 + 1] - qb2 ||
                 ctot != (int)(v.io.s_off[m2 + 1] - v.io.s_off[m2]))) return false;
    bool ok = true;
    if (!PAIR && lane == 0) { f(((unsigned long long)(unsigned)r << 32) | (unsigned)nq, 1ull); f((unsigned long long)(unsigned)ctot, 2ull); }
    if (v.io.avoid_ess || v.io.nextra) {
        for (int i = lane; i < r; i += 64) {
            const unsigned long long w = v.io.avoid_ess ? (unsigned long long)(v.flags[dofs[i]] & FLAG_ON_ESS_BORDER) : 0ull;
            if (PAIR) ok = ok && w == (v.io.avoid_ess ? (unsigned long long)(v.flags[dofs2[i]] & FLAG_ON_ESS_BORDER) : 0ull);
            else f(w, (1ull << 40) + (unsigned long long)i);
            for (int q = 0; q < v.io.nextra; ++q) {
                const unsigned long long x = (unsigned long long)__double_as_longlong(v.io.extra[(size_t)q * v.io.ND + dofs[i]]);
                if (PAIR) ok = ok && x == (unsigned long long)__double_as_longlong(v.io.extra[(size_t)q * v.io.ND + dofs2[i]]);
                else f(x, (2ull << 40) + ((unsigned long long)q << 24) + (unsigned long long)i);
            }
        }
    }
    for (int t = 0; t < nq; ++t) {      // (wave-uniform)
        const int ae = v.mis2ae_J[qb + t], na = v.ae2d_I[ae + 1] - v.ae2d_I[ae], ma = v.io.ae_m[ae];
        const int id = v.ae_ev[ae];
        const int *loc = v.pair_loc + v.pair_loc_off[qb + t];
        const unsigned long long tag = (unsigned long long)(t + 3) << 40;
        if (PAIR) {
            const int ae2 = v.mis2ae_J[qb2 + t];
            // (the same eigenvector block: the same class, or the same agglomerate)
            if (!(ae == ae2 || (id >= 0 && id == v.ae_ev[ae2])) || na != v.ae2d_I[ae2 + 1] - v.ae2d_I[ae2] || ma != v.io.ae_m[ae2]) return false;
            const int *loc2 = v.pair_loc + v.pair_loc_off[qb2 + t];
            for (int i = lane; i < r; i += 64) ok = ok && loc[i] == loc2[i];
        } else {
            if (lane == 0) {
                f(id >= 0 ? (unsigned long long)(unsigned)id : 0x8000000000000000ull + (unsigned long long)(unsigned)ae, tag);
                f(((unsigned long long)(unsigned)na << 32) | (unsigned)ma, tag + 1);
            }
            for (int i = lane; i < r; i += 64) f((unsigned long long)(unsigned)loc[i], tag + 2 + (unsigned long long)i);
        }
    }
    return ok;
}
// a wavefront per MIS: its 64-bit hash (two of them: the second is checked before the walk of the verification)
__global__ __launch_bounds__(256) void mis_hash_kernel(MisIn v, int nm, unsigned long long *__restrict__ h) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= nm) return;
    unsigned long long h1 = 0, h2 = 0;
    mis_walk<false>(v, m, 0, lane, [&](unsigned long long w, unsigned long long pos) {
        const unsigned long long k = mi_mix(w + 0x9E3779B97F4A7C15ull * (pos + 1));
        h1 += k;
        h2 += (k >> 32) * (k & 0xffffffffull);      // (second sum: the product of the halves of the mixed word; a full second mix was half of the kernel)
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); }
    if (lane == 0) { h[2 * (size_t)m] = h1; h[2 * (size_t)m + 1] = h2; }
}
// open-addressing table keyed by the first hash: the smallest MIS of every key
constexpr unsigned long long MI_EMPTY = ~0ull;
__global__ __launch_bounds__(256) void mis_group_kernel(int nm, const unsigned long long *__restrict__ h, unsigned long long *__restrict__ keys,
                                                        int *__restrict__ first, unsigned mask, int *__restrict__ slot) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= nm) return;
    unsigned long long key = h[2 * (size_t)m];
    if (key == MI_EMPTY) key = 0;
    unsigned pos = (unsigned)(key ^ (key >> 32)) & mask;
    for (;;) {
        const unsigned long long cur = atomicCAS(keys + pos, MI_EMPTY, key);
        if (cur == MI_EMPTY || cur == key) break;
        pos = (pos + 1) & mask;
    }
    atomicMin(first + pos, m);
    slot[m] = (int)pos;
}
__global__ __launch_bounds__(256) void mis_rep_kernel2(int nm, const int *__restrict__ slot, const int *__restrict__ first, int *__restrict__ rep) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < nm) rep[m] = first[slot[m]];
}
// every MIS against the first of its key, word by word; a MIS that differs stands for itself
__global__ __launch_bounds__(256) void mis_verify_kernel(MisIn v, int nm, const unsigned long long *__restrict__ h, int *__restrict__ rep,
                                                         int *__restrict__ isrep) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= nm) return;
    const int r0 = rep[m];
    bool same = true;
    if (r0 != m) {
        same = h[2 * (size_t)m + 1] == h[2 * (size_t)r0 + 1];
        if (same) same = mis_walk<true>(v, m, r0, lane, [](unsigned long long, unsigned long long) {});
        same = __ballot(!same) == 0ull;
    }
    if (lane == 0) {
        if (!same) rep[m] = m;
        isrep[m] = (r0 == m || !same) ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void mis_list_kernel(int nm, const int *__restrict__ isrep, const int *__restrict__ pos, int *__restrict__ list) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < nm && isrep[m]) list[pos[m]] = m;
}
// results of the first members to the other members of their classes (a wavefront per MIS)
__global__ __launch_bounds__(256) void mis_copy_kernel(MisIn v, int nm, const int *__restrict__ rep) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= nm) return;
    const int r0 = rep[m];
    if (r0 == m) return;
    const int r = v.mis2d_I[m + 1] - v.mis2d_I[m], k = v.io.k[r0], ctot = (int)(v.io.s_off[m + 1] - v.io.s_off[m]);
    if (lane == 0) { v.io.k[m] = k; v.io.ncols[m] = v.io.ncols[r0]; }
    const double *su = v.io.U + v.io.u_off[r0], *ss = v.io.sig + v.io.s_off[r0];
    double *du = v.io.U + v.io.u_off[m], *ds = v.io.sig + v.io.s_off[m];
    for (int i = lane; i < r * k; i += 64) du[i] = su[i];
    for (int i = lane; i < ctot; i += 64) ds[i] = ss[i];
}

void mis_svd(hipStream_t s, const DevRelations &rel, int num_mises, int max_ctot, const MisSvdIO &io, int m0, const int *ae_ev) {
    if (!num_mises) return;
    const size_t lds = (sizeof(double) + sizeof(int)) * (size_t)(max_ctot + 2);
    // classes of identical MISes (single rank, the whole level at once): only their first members go through the SVD
    if (ae_ev && m0 == 0 && num_mises >= 4096) {
        const int nm = num_mises;
        profiler().begin(s);
        MisIn v{rel.mis2d_I.p, rel.mis2d_J.p, rel.mis2ae_I.p, rel.mis2ae_J.p, rel.ae2d_I.p, rel.pair_loc_off.p, rel.pair_loc.p, rel.flags.p, io, ae_ev};
        unsigned tsize = 1024;
        while (tsize < 4u * (unsigned)nm) tsize <<= 1;
        DBuf<unsigned long long> h(2 * (size_t)nm), keys((size_t)tsize);
        DBuf<int> first((size_t)tsize), slot((size_t)nm), rep((size_t)nm), isrep((size_t)nm), pos((size_t)nm + 1), list((size_t)nm);
        SA_HIP_CHECK(hipMemsetAsync(keys.p, 0xff, sizeof(unsigned long long) * (size_t)tsize, s));
        SA_HIP_CHECK(hipMemsetAsync(first.p, 0x7f, sizeof(int) * (size_t)tsize, s));
        hipLaunchKernelGGL(mis_hash_kernel, dim3(div_up(nm, 4)), dim3(256), 0, s, v, nm, h.p);
        hipLaunchKernelGGL(mis_group_kernel, dim3(div_up(nm, 256)), dim3(256), 0, s, nm, h.p, keys.p, first.p, tsize - 1, slot.p);
        hipLaunchKernelGGL(mis_rep_kernel2, dim3(div_up(nm, 256)), dim3(256), 0, s, nm, slot.p, first.p, rep.p);
        hipLaunchKernelGGL(mis_verify_kernel, dim3(div_up(nm, 4)), dim3(256), 0, s, v, nm, h.p, rep.p, isrep.p);
        SA_HIP_CHECK(hipGetLastError());
        exclusive_scan_int(s, nm, isrep.p, pos.p);
        hipLaunchKernelGGL(mis_list_kernel, dim3(div_up(nm, 256)), dim3(256), 0, s, nm, isrep.p, pos.p, list.p);
        SA_HIP_CHECK(hipGetLastError());
        int last[2] = {0, 0};
        SA_HIP_CHECK(hipMemcpyAsync(&last[0], pos.p + (nm - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipMemcpyAsync(&last[1], isrep.p + (nm - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
        const int nrep = last[0] + last[1];
        profiler().end(s, "eig_dedupe", 0.0, 0.0);
        if (options().debug & 1) std::fprintf(stderr, "MISes: %d distinct of %d\n", nrep, nm);
        profiler().begin(s);
        hipLaunchKernelGGL(mis_svd_kernel, dim3(nrep), dim3(64), lds, s, rel.mis2d_I.p,
                           rel.mis2d_J.p, rel.mis2ae_I.p, rel.mis2ae_J.p, rel.ae2d_I.p,
                           rel.pair_loc_off.p, rel.pair_loc.p, rel.flags.p, io, 0, list.p);
        hipLaunchKernelGGL(mis_copy_kernel, dim3(div_up(nm, 4)), dim3(256), 0, s, v, nm, rep.p);
        SA_HIP_CHECK(hipGetLastError());
        SA_HIP_CHECK(hipStreamSynchronize(s));      // (the lists are freed here)
        profiler().end(s, "mis_svd", 0.0, 0.0);
        return;
    }
    profiler().begin(s);
    hipLaunchKernelGGL(mis_svd_kernel, dim3(num_mises), dim3(64), lds, s, rel.mis2d_I.p,
                       rel.mis2d_J.p, rel.mis2ae_I.p, rel.mis2ae_J.p, rel.ae2d_I.p,
                       rel.pair_loc_off.p, rel.pair_loc.p, rel.flags.p, io, m0);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "mis_svd", 0.0, 0.0);
}

// ---------------------------------------------------------------------------------------
// P and R
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void p_count_kernel(int ND, const int *__restrict__ mises,
                                                      const int *__restrict__ k, int *__restrict__ cnt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < ND) cnt[i] = k[mises[i]];
}

__global__ __launch_bounds__(256) void r_len_kernel(int nm, const int *__restrict__ mis2d_I, const int *__restrict__ k,
                                                    const int *__restrict__ coloff, int *__restrict__ len) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= nm) return;
    const int r = mis2d_I[m + 1] - mis2d_I[m], c0 = coloff[m];
    for (int v = 0; v < k[m]; ++v) len[c0 + v] = r;
}

// exclusive scan of ints, three small kernels (tile = 1024); OUT = int, or roff_t for the row offsets of an
// operator (counts are int, their running sum may pass 2^31)
template <class OUT>
__global__ __launch_bounds__(256) void scan_tile_kernel(int n, const int *__restrict__ in,
                                                        OUT *__restrict__ out, OUT *__restrict__ tsum) {
    __shared__ OUT sh[256];
    const long base = (long)blockIdx.x * 1024;
    int v[4];
    OUT run = 0;
    for (int q = 0; q < 4; ++q) {
        const long i = base + threadIdx.x * 4 + q;
        v[q] = (i < n) ? in[i] : 0;
        run += v[q];
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const OUT t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    OUT excl = sh[threadIdx.x] - run;
    for (int q = 0; q < 4; ++q) {
        const long i = base + threadIdx.x * 4 + q;
        if (i < n) out[i] = excl;
        excl += v[q];
    }
    if (threadIdx.x == 255) tsum[blockIdx.x] = sh[255];
}
template <class OUT>
__global__ __launch_bounds__(256) void scan_sums_kernel(int nt, OUT *__restrict__ tsum, OUT *__restrict__ total) {
    __shared__ OUT sh[256];
    __shared__ OUT carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nt; base += 256) {
        const int i = base + threadIdx.x;
        const OUT x = (i < nt) ? tsum[i] : 0;
        sh[threadIdx.x] = x;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const OUT t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nt) tsum[i] = carry + sh[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
template <class OUT>
__global__ __launch_bounds__(256) void scan_add_kernel(int n, OUT *__restrict__ out,
                                                       const OUT *__restrict__ tsum,
                                                       const OUT *__restrict__ total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += tsum[i >> 10];
    if (i == 0) out[n] = *total;
}

// out has n+1 entries
template <class OUT>
static void exclusive_scan_t(hipStream_t s, int n, const int *in, OUT *out) {
    const int nt = div_up(n, 1024);
    DBuf<OUT> tsum((size_t)nt + 1);
    hipLaunchKernelGGL(scan_tile_kernel<OUT>, dim3(nt), dim3(256), 0, s, n, in, out, tsum.p);
    hipLaunchKernelGGL(scan_sums_kernel<OUT>, dim3(1), dim3(256), 0, s, nt, tsum.p, tsum.p + nt);
    hipLaunchKernelGGL(scan_add_kernel<OUT>, dim3(div_up(n, 256)), dim3(256), 0, s, n, out, (const OUT *)tsum.p, (const OUT *)(tsum.p + nt));
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));  // tsum is freed on return
}
void exclusive_scan_int(hipStream_t s, int n, const int *in, int *out) { exclusive_scan_t<int>(s, n, in, out); }
void exclusive_scan_int_async(hipStream_t s, int n, const int *in, int *out, int *tsum) {
    const int nt = div_up(n, 1024);
    hipLaunchKernelGGL(scan_tile_kernel<int>, dim3(nt), dim3(256), 0, s, n, in, out, tsum);
    hipLaunchKernelGGL(scan_sums_kernel<int>, dim3(1), dim3(256), 0, s, nt, tsum, tsum + nt);
    hipLaunchKernelGGL(scan_add_kernel<int>, dim3(div_up(n, 256)), dim3(256), 0, s, n, out, (const int *)tsum, (const int *)(tsum + nt));
    SA_HIP_CHECK(hipGetLastError());
}
void exclusive_scan_off(hipStream_t s, int n, const int *in, roff_t *out) { exclusive_scan_t<roff_t>(s, n, in, out); }

__global__ __launch_bounds__(256) void p_fill_kernel(int ND, const int *__restrict__ mises,
                                                     const int *__restrict__ row_in_mis,
                                                     const int *__restrict__ mis2d_I,
                                                     const int *__restrict__ k,
                                                     const int *__restrict__ coloff,
                                                     const int64_t *__restrict__ u_off,
                                                     const double *__restrict__ U,
                                                     const roff_t *__restrict__ rowptr,
                                                     int *__restrict__ col, double *__restrict__ val) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ND) return;
    const int m = mises[i], km = k[m];
    const int r = mis2d_I[m + 1] - mis2d_I[m];
    const double *Um = U + u_off[m] + row_in_mis[i];
    const roff_t base = rowptr[i];
    const int c0 = coloff[m];
    for (int v = 0; v < km; ++v) {
        col[base + v] = c0 + v;
        val[base + v] = Um[(size_t)v * r];
    }
}

__global__ __launch_bounds__(256) void r_fill_kernel(int num_mises, const int *__restrict__ mis2d_I,
                                                     const int *__restrict__ mis2d_J,
                                                     const int *__restrict__ k,
                                                     const int *__restrict__ coloff,
                                                     const int64_t *__restrict__ u_off,
                                                     const double *__restrict__ U,
                                                     const roff_t *__restrict__ rowptr,
                                                     int *__restrict__ col, double *__restrict__ val) {
    const int m = blockIdx.x;
    const int km = k[m];
    if (km == 0) return;
    const int r = mis2d_I[m + 1] - mis2d_I[m];
    const int *dofs = mis2d_J + mis2d_I[m];
    const double *Um = U + u_off[m];
    for (int idx = threadIdx.x; idx < km * r; idx += 256) {
        const int v = idx / r, i = idx % r;
        const roff_t base = rowptr[coloff[m] + v];
        col[base + i] = dofs[i];
        val[base + i] = Um[(size_t)v * r + i];
    }
}

void build_P_R(hipStream_t s, const DevRelations &rel, const Relations &hrel,
               const std::vector<int> &h_k, const std::vector<int64_t> &h_u_off, const int *d_k,
               const int *d_coloff, const int64_t *d_u_off, const double *U, DCsr &P, DCsr &R) {
    (void)h_u_off;
    const int ND = hrel.ND;
    int nc = 0;
    int64_t nnz = 0;
    for (int m = 0; m < hrel.num_mises; ++m) {
        nc += h_k[m];
        nnz += (int64_t)h_k[m] * hrel.mis_to_dof.row_size(m);
    }
    P.nrows = ND; P.ncols = nc; P.nnz = nnz;
    P.rowptr.alloc((size_t)ND + 1);
    P.col.alloc((size_t)nnz);
    P.val.alloc((size_t)nnz);
    {
        DBuf<int> cnt((size_t)ND);
        hipLaunchKernelGGL(p_count_kernel, dim3(div_up(ND, 256)), dim3(256), 0, s, ND, rel.mises.p, d_k, cnt.p);
        exclusive_scan_off(s, ND, cnt.p, P.rowptr.p);
    }
    hipLaunchKernelGGL(p_fill_kernel, dim3(div_up(ND, 256)), dim3(256), 0, s, ND, rel.mises.p,
                       rel.dof_row_in_mis.p, rel.mis2d_I.p, d_k, d_coloff, d_u_off, U, P.rowptr.p,
                       P.col.p, P.val.p);
    P.lanes_per_row = pick_lanes_per_row(nnz, ND);
    R.nrows = nc; R.ncols = ND; R.nnz = nnz;
    {   // row (m, v) of R has the dofs of MIS m: lengths and their running sum on the device (the host loop appended half a
        // million offsets to a vector and uploaded it between the SVDs and the fill kernels)
        DBuf<int> len((size_t)std::max(nc, 1));
        R.rowptr.alloc((size_t)nc + 1);
        if (hrel.num_mises)
            hipLaunchKernelGGL(r_len_kernel, dim3(div_up(hrel.num_mises, 256)), dim3(256), 0, s, hrel.num_mises, rel.mis2d_I.p, d_k,
                               d_coloff, len.p);
        if (nc > 0) exclusive_scan_off(s, nc, len.p, R.rowptr.p);
        else SA_HIP_CHECK(hipMemsetAsync(R.rowptr.p, 0, sizeof(roff_t), s));
    }
    R.col.alloc((size_t)nnz);
    R.val.alloc((size_t)nnz);
    if (hrel.num_mises)
        hipLaunchKernelGGL(r_fill_kernel, dim3(hrel.num_mises), dim3(256), 0, s, hrel.num_mises,
                           rel.mis2d_I.p, rel.mis2d_J.p, d_k, d_coloff, d_u_off, U, R.rowptr.p,
                           R.col.p, R.val.p);
    SA_HIP_CHECK(hipGetLastError());
    R.lanes_per_row = pick_lanes_per_row(nnz, nc > 0 ? nc : 1);
}

// ---------------------------------------------------------------------------------------
// coarse elem_to_dof + colpos (next level's elements)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ae_mis_k_kernel(long n, const int *__restrict__ ae2mis_J,
                                                       const int *__restrict__ k, int *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = k[ae2mis_J[i]];
}

__global__ __launch_bounds__(256) void ae_ke_max_kernel(int nparts, const int *__restrict__ ae2mis_I,
                                                        const int *__restrict__ colpos_ptr, int *__restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    int ke = 0;
    if (e < nparts) ke = colpos_ptr[ae2mis_I[e + 1]] - colpos_ptr[ae2mis_I[e]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ke = max(ke, __shfl_xor(ke, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, ke);
}

// One workgroup per AE.  The host loop appends coarse dof (m, v) when it meets the first dof of MIS m (in
// AE_to_dof order) whose prolongator entry v is non-zero: the list order is the order of the keys
// (first position, candidate index) -- a position belongs to one MIS, so ties are between entries of the
// same MIS and the candidate index orders them by v.
__global__ __launch_bounds__(256) void coarse_e2d_kernel(
    const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J, const int *__restrict__ ae2mis_I,
    const int *__restrict__ ae2mis_J, const int *__restrict__ mises, const int *__restrict__ mis_k,
    const int *__restrict__ mis_coloff, const int *__restrict__ colpos_ptr, const roff_t *__restrict__ prow,
    const double *__restrict__ pval, int cap, int *__restrict__ colpos, int *__restrict__ e2d_J,
    int *__restrict__ err) {
    extern __shared__ unsigned long long ce_keys[];      // [cap] keys, then [cap] coarse dof ids
    int *cd = (int *)(ce_keys + cap);
    const int e = blockIdx.x, tid = threadIdx.x;
    const int tb = ae2mis_I[e], te = ae2mis_I[e + 1];
    const int c0 = colpos_ptr[tb], KE = colpos_ptr[te] - c0;
    for (int t = tb + tid; t < te; t += 256) {
        const int m = ae2mis_J[t], km = mis_k[m], base = colpos_ptr[t] - c0, co = mis_coloff[m];
        for (int v = 0; v < km; ++v) {
            ce_keys[base + v] = ~0ull;
            cd[base + v] = co + v;
        }
    }
    __syncthreads();
    const int d0 = ae2d_I[e], d1 = ae2d_I[e + 1];
    for (int k = d0 + tid; k < d1; k += 256) {
        const int dof = ae2d_J[k], m = mises[dof], km = mis_k[m];
        if (km == 0) continue;
        int lo = tb, hi = te;                               // AE_to_mis rows are ascending
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ae2mis_J[mid] < m) lo = mid + 1; else hi = mid;
        }
        const int base = colpos_ptr[lo] - c0;
        const roff_t pr = prow[dof];
        for (int v = 0; v < km; ++v)
            if (pval[pr + v] != 0.0)
                atomicMin(&ce_keys[base + v], ((unsigned long long)(unsigned)(k - d0) << 32) | (unsigned)(base + v));
    }
    __syncthreads();
    for (int idx = tid; idx < KE; idx += 256) {
        const unsigned long long key = ce_keys[idx];
        if (key == ~0ull) {                                 // a coarse dof with an all-zero column inside this AE
            colpos[c0 + idx] = -1;
            atomicExch(err, 1);
            continue;
        }
        int rank = 0;
        for (int j = 0; j < KE; ++j) rank += (ce_keys[j] < key) ? 1 : 0;
        colpos[c0 + idx] = rank;
        e2d_J[c0 + rank] = cd[idx];
    }
}

bool coarse_e2d_device(hipStream_t s, const DevRelations &rel, const Relations &hrel, const int *d_mis_k,
                       const int *d_mis_coloff, int ncoarse, const roff_t *p_rowptr, const double *p_val,
                       DBuf<int> &colpos_ptr, DBuf<int> &colpos, Table &e2d) {
    const int nparts = hrel.nparts;
    const long npairs = (long)hrel.AE_to_mis.J.size();
    if (nparts == 0 || npairs == 0) return false;
    DBuf<int> kk((size_t)npairs), info(2);
    info.zero(s);
    hipLaunchKernelGGL(ae_mis_k_kernel, dim3(div_up(npairs, 256)), dim3(256), 0, s, npairs, rel.ae2mis_J.p, d_mis_k, kk.p);
    colpos_ptr.alloc((size_t)npairs + 1);
    exclusive_scan_int(s, (int)npairs, kk.p, colpos_ptr.p);
    hipLaunchKernelGGL(ae_ke_max_kernel, dim3(div_up(nparts, 256)), dim3(256), 0, s, nparts, rel.ae2mis_I.p, colpos_ptr.p, info.p);
    SA_HIP_CHECK(hipGetLastError());
    const int cap = info.to_host(s)[0];
    const size_t lds = (size_t)cap * 12 + 16;
    if (lds > 60 * 1024) return false;
    auto h_ptr = colpos_ptr.to_host(s);
    const int total = h_ptr[(size_t)npairs];
    colpos.alloc((size_t)total + 1);
    DBuf<int> d_J((size_t)total + 1);
    if (total) {
        hipLaunchKernelGGL(coarse_e2d_kernel, dim3(nparts), dim3(256), lds, s, rel.ae2d_I.p, rel.ae2d_J.p, rel.ae2mis_I.p,
                           rel.ae2mis_J.p, rel.mises.p, d_mis_k, d_mis_coloff, colpos_ptr.p, p_rowptr, p_val, cap, colpos.p,
                           d_J.p, info.p + 1);
        SA_HIP_CHECK(hipGetLastError());
    }
    SA_REQUIRE(info.to_host(s)[1] == 0, "coarse dof with an all-zero prolongator column in an AE");
    e2d.ncols = ncoarse;
    e2d.I.resize((size_t)nparts + 1);
    for (int e = 0; e <= nparts; ++e) e2d.I[e] = h_ptr[(size_t)hrel.AE_to_mis.I[e]];
    e2d.J.resize((size_t)total);
    if (total) {
        SA_HIP_CHECK(hipMemcpyAsync(e2d.J.data(), d_J.p, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// Galerkin product through the MIS blocks
// ---------------------------------------------------------------------------------------
constexpr int RAP_NT = 256;
constexpr int RAP_HASH = 2048;
constexpr int RAP_SMALL_ROWS = 64;     // MISes of at most this many dofs take a one-wavefront workgroup in the numeric kernel

// symbolic: neighbour MISes (with k > 0) of every MIS; pass 0 counts, pass 1 writes the
// ascending list.
// NT threads, tables of CAP slots; mlist != nullptr: the workgroups take the MISes mlist[0 .. grid).  The MISes of at most
// RAP_SMALL_ROWS dofs take <64, 256>: one wavefront and 2 KB of tables instead of four and 16 KB (as in rap_numeric_kernel: a
// workgroup costs tens of microseconds whatever its MIS holds, so what counts is how many are resident); one whose neighbours
// do not fit 128 of those 256 slots reports cnt = -1 and is done again by <RAP_NT, RAP_HASH>.
template <int NT, int CAP>
__global__ __launch_bounds__(NT) void rap_symbolic_kernel(
    const int *__restrict__ mlist, int pass, const int *__restrict__ mis2d_I, const int *__restrict__ mis2d_J,
    const roff_t *__restrict__ Arow, const int *__restrict__ Acol, const int *__restrict__ mises,
    const int *__restrict__ k, int *__restrict__ cnt, const int *__restrict__ nbr_ptr,
    int *__restrict__ nbr, int *__restrict__ err, int *__restrict__ stage = nullptr, int stage_cap = 0,
    int maxrow = 0) {
    __shared__ int table[CAP];
    __shared__ int nfound;
    const int m1 = mlist ? mlist[blockIdx.x] : (int)blockIdx.x;
    if (k[m1] == 0) {
        if (pass == 0 && threadIdx.x == 0) cnt[m1] = 0;
        return;
    }
    const int r1 = mis2d_I[m1 + 1] - mis2d_I[m1];
    const int *dofs = mis2d_J + mis2d_I[m1];
    // The table is as large as this MIS can need: a MIS of r1 dofs has at most r1 * maxrow neighbours (most MISes
    // are a vertex, an edge or a face of an agglomerate -- a few dofs: clearing and compacting 2 048 slots for
    // each of them was most of this kernel).  maxrow = 0: the full table.
    // A larger MIS (a face or the interior of an agglomerate) could have many neighbours but on a mesh has a few
    // dozen: it starts with 256 slots and walks its rows again with the full table only if those fill up.
    __shared__ int overflow;
    int HS = CAP;
    if (maxrow > 0) {
        const long bound = 2l * r1 * maxrow;
        HS = 64;
        while (HS < 256 && HS < bound) HS <<= 1;
    }
    const bool certain = maxrow > 0 && 2l * r1 * maxrow <= HS;     // the table cannot fill up
    for (;;) {
        for (int i = threadIdx.x; i < HS; i += NT) table[i] = -1;
        if (threadIdx.x == 0) { nfound = 0; overflow = 0; }
        __syncthreads();
        for (int il = threadIdx.x; il < r1; il += NT) {
            const int g = dofs[il];
            for (roff_t q = Arow[g]; q < Arow[g + 1]; ++q) {
                const int m2 = mises[Acol[q]];
                if (k[m2] == 0) continue;
                unsigned h = hash_home((unsigned)m2, (unsigned)HS);
                for (int probe = 0; probe < HS; ++probe) {
                    const int old = atomicCAS(&table[h], -1, m2);
                    if (old == -1) { atomicAdd(&nfound, 1); break; }
                    if (old == m2) break;
                    h = (h + 1) & (HS - 1);
                    if (probe == HS - 1) { if (HS == RAP_HASH) atomicExch(err, 1); else overflow = 1; }
                }
            }
        }
        __syncthreads();
        // (more than half full counts as full: the probe sequences get long, and the rank pass below is quadratic)
        const bool full = !certain && (overflow || 2 * nfound > HS);
        const bool again = HS < CAP && full;
        __syncthreads();
        if (CAP < RAP_HASH && full && !again) {      // (the small tables are not enough: left to the full-size kernel)
            if (threadIdx.x == 0) cnt[m1] = -1;
            return;
        }
        if (!again) break;
        HS = CAP;
    }
    if (pass == 0) {
        if (threadIdx.x == 0) cnt[m1] = nfound;
        // the list itself goes to a fixed-capacity staging row when it fits (it nearly always does: 27 neighbours
        // on a hexahedral mesh), so that the second pass is a copy instead of a second walk over the rows of A
        if (!stage || nfound > stage_cap) return;
    }
    // compact the table into a short list, then rank inside the list and write in order
    __shared__ int list[CAP];
    __shared__ int nlist;
    if (threadIdx.x == 0) nlist = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < HS; i += NT) {
        const int v = table[i];
        if (v >= 0) list[atomicAdd(&nlist, 1)] = v;
    }
    __syncthreads();
    const int nl = nlist;
    int *out = (pass == 0) ? stage + (size_t)m1 * stage_cap : nbr + nbr_ptr[m1];
    for (int i = threadIdx.x; i < nl; i += NT) {
        const int v = list[i];
        int rank = 0;
        for (int j = 0; j < nl; ++j) rank += (list[j] < v);
        out[rank] = v;
    }
}

__global__ __launch_bounds__(256) void rap_unstage_kernel(int nm, int cap, const int *__restrict__ stage,
                                                          const int *__restrict__ nbr_ptr, int *__restrict__ nbr) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int m = (int)(i / cap), t = (int)(i % cap);
    if (m < nm && t < nbr_ptr[m + 1] - nbr_ptr[m]) nbr[nbr_ptr[m] + t] = stage[i];
}

// NT threads per workgroup; list != nullptr: the workgroups take the MISes list[0 .. grid) instead of m_first + block.
// (A workgroup costs tens of microseconds whatever its MIS holds -- a dozen rounds of dependent gathers and as many
// barriers -- and a CU has room for eight workgroups of four wavefronts: the MISes of at most RAP_SMALL_ROWS dofs, seven
// eighths of the MISes of a hexahedral mesh (vertices, edges and faces of the agglomerates), take one wavefront each instead,
// with 8 KB of LDS: more of them resident, and a face's 49 rows on 64 lanes instead of 256.  Same code, same order of every sum.)
template <int NT>
__global__ __launch_bounds__(NT) void rap_numeric_kernel(
    const int *__restrict__ list, int m_first, const int *__restrict__ mis2d_I, const int *__restrict__ mis2d_J, const roff_t *__restrict__ Arow,
    const int *__restrict__ Acol, const double *__restrict__ Aval, const int *__restrict__ mises,
    const int *__restrict__ row_in_mis, const int *__restrict__ k, const int *__restrict__ coloff,
    const int64_t *__restrict__ u_off, const double *__restrict__ U,
    const int *__restrict__ nbr_ptr, const int *__restrict__ nbr, const roff_t *__restrict__ crowptr,
    int *__restrict__ ccol, double *__restrict__ cval, int lds_doubles) {
    extern __shared__ __align__(16) double lds[];
    const int m1 = list ? list[blockIdx.x] : m_first + (int)blockIdx.x;
    const int k1 = k[m1];
    if (k1 == 0) return;
    const int tid = threadIdx.x;
    const int nn = nbr_ptr[m1 + 1] - nbr_ptr[m1];
    const int *nb = nbr + nbr_ptr[m1];
    const int r1 = mis2d_I[m1 + 1] - mis2d_I[m1];
    const int *dofs = mis2d_J + mis2d_I[m1];
    const double *U1 = U + u_off[m1];
    // LDS: per neighbour MIS its id, k, row count, offset of its basis and first local column
    // (the row loop below is a chain of dependent gathers: everything that can come from LDS does),
    // then acc[k1*ncol], T[RC*ncol]
    const int pos_d = 3 * nn + 2;
    long long *uo = (long long *)lds;         // [nn]
    int *pos = (int *)(uo + nn);              // [nn + 1]
    int *nbl = pos + nn + 1;                  // [nn]
    int *kk = nbl + nn;                       // [nn]
    int *rr = kk + nn;                        // [nn]
    for (int t = tid; t < nn; t += NT) {
        const int m2 = nb[t];
        nbl[t] = m2;
        kk[t] = k[m2];
        rr[t] = mis2d_I[m2 + 1] - mis2d_I[m2];
        uo[t] = u_off[m2];
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < nn; ++t) { pos[t] = run; run += kk[t]; }
        pos[nn] = run;
    }
    __syncthreads();
    const int ncol = pos[nn];
    double *acc = lds + pos_d;
    // LDS rows of ncol doubles: KC accumulator rows + RC rows of T = (A U)[chunk of MIS rows].
    // Normally KC = k1; a very wide block (many eigenvectors per AE) is done in several passes
    // over the output rows, recomputing T for each (host guarantees >= 2 rows).
    const int rows_total = (lds_doubles - pos_d) / ncol;
    const int KC = (k1 + 1 <= rows_total) ? k1 : rows_total / 2;
    const int RC = rows_total - KC;
    double *T = acc + (size_t)KC * ncol;
    // column indices of the k1 output rows
    for (int idx = tid; idx < k1 * ncol; idx += NT) {
        const int v1 = idx / ncol, cc = idx % ncol;
        int lo = 0, hi = nn;  // neighbour owning local column cc
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pos[mid] <= cc) lo = mid; else hi = mid; }
        ccol[crowptr[coloff[m1] + v1] + cc] = coloff[nbl[lo]] + (cc - pos[lo]);
    }
    for (int v0 = 0; v0 < k1; v0 += KC) {
        const int kc = min(KC, k1 - v0);
        __syncthreads();
        for (int i = tid; i < kc * ncol; i += NT) acc[i] = 0.0;
        for (int c0 = 0; c0 < r1; c0 += RC) {
            const int rc = min(RC, r1 - c0);
            __syncthreads();
            for (int i = tid; i < rc * ncol; i += NT) T[i] = 0.0;
            __syncthreads();
            for (int il = tid; il < rc; il += NT) {
                const int g = dofs[c0 + il];
                double *Trow = T + (size_t)il * ncol;
                // four entries of the row at a time: their gather chains (column -> MIS -> slot -> row in the MIS ->
                // basis entry) are independent and are all requested before the first product; the products are
                // then added in the order of the entries, as a plain loop would
                const roff_t qe = Arow[g + 1];
                for (roff_t q0 = Arow[g]; q0 < qe; q0 += 4) {
                    int jj4[4], lo4[4];
                    double a4[4], u4[4];
                    bool on4[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const roff_t q = min(q0 + t, qe - 1);
                        jj4[t] = Acol[q];
                        a4[t] = Aval[q];
                    }
                    int m4[4], rim4[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { m4[t] = mises[jj4[t]]; rim4[t] = row_in_mis[jj4[t]]; }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        int lo = 0, hi = nn;      // (the neighbour list holds exactly the MISes with k > 0)
                        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (nbl[mid] <= m4[t]) lo = mid; else hi = mid; }
                        lo4[t] = lo;
                        on4[t] = q0 + t < qe && nn != 0 && nbl[lo] == m4[t];
                        u4[t] = on4[t] ? U[uo[lo] + rim4[t]] : 0.0;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (!on4[t]) continue;
                        const int lo = lo4[t];
                        const int k2 = kk[lo];
                        const int r2 = rr[lo];
                        const double *U2 = U + uo[lo] + rim4[t];
                        double *dst = Trow + pos[lo];
                        dst[0] = fma(a4[t], u4[t], dst[0]);
                        for (int v = 1; v < k2; ++v) dst[v] = fma(a4[t], U2[(size_t)v * r2], dst[v]);
                    }
                }
            }
            __syncthreads();
            for (int idx = tid; idx < kc * ncol; idx += NT) {
                const int v1 = idx / ncol, cc = idx % ncol;
                double sum = acc[idx];
                const double *u = U1 + (size_t)(v0 + v1) * r1 + c0;
                for (int il = 0; il < rc; ++il) sum = fma(u[il], T[(size_t)il * ncol + cc], sum);
                acc[idx] = sum;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < kc * ncol; idx += NT) {
            const int v1 = idx / ncol, cc = idx % ncol;
            cval[crowptr[coloff[m1] + v0 + v1] + cc] = acc[idx];
        }
    }
}

void rap_mis(hipStream_t s, const DevRelations &rel, const Relations &hrel, const DCsr &A,
             const std::vector<int> &h_k, const std::vector<int> &h_coloff, const int *d_k,
             const int *d_coloff, const int64_t *d_u_off, const double *U, DCsr &Ac, int rank, int world,
             std::vector<long long> *nnz_off) {
    const int nm = hrel.num_mises;
    int nc = 0;
    for (int m = 0; m < nm; ++m) nc += h_k[m];
    Ac.nrows = Ac.ncols = nc;
    Ac.nnz = 0;
    if (nm == 0 || nc == 0) {
        Ac.rowptr.from_host(std::vector<roff_t>(1, 0), s);
        return;
    }
    DBuf<int> cnt((size_t)nm), err(1);
    err.zero(s);
    if (A.max_row < 0) A.max_row = csr_max_row(s, A);
    const int maxrow = A.max_row;
    profiler().begin(s);
    constexpr int STAGE_CAP = 64;
    DBuf<int> stage((size_t)nm * STAGE_CAP);
    {   // (two lists by the number of dofs, as for the numeric kernel below)
        std::vector<int> sym_small, sym_big;
        for (int m = 0; m < nm; ++m) (hrel.mis_to_dof.row_size(m) <= RAP_SMALL_ROWS ? sym_small : sym_big).push_back(m);
        DBuf<int> d_sym_small, d_sym_big;
        d_sym_small.from_host(sym_small, s);
        d_sym_big.from_host(sym_big, s);
        if (!sym_big.empty())
            hipLaunchKernelGGL((rap_symbolic_kernel<RAP_NT, RAP_HASH>), dim3((unsigned)sym_big.size()), dim3(RAP_NT), 0, s, d_sym_big.p, 0,
                               rel.mis2d_I.p, rel.mis2d_J.p, A.rowptr.p, A.col.p, rel.mises.p, d_k, cnt.p, nullptr, nullptr, err.p, stage.p,
                               STAGE_CAP, maxrow);
        if (!sym_small.empty())
            hipLaunchKernelGGL((rap_symbolic_kernel<64, 256>), dim3((unsigned)sym_small.size()), dim3(64), 0, s, d_sym_small.p, 0,
                               rel.mis2d_I.p, rel.mis2d_J.p, A.rowptr.p, A.col.p, rel.mises.p, d_k, cnt.p, nullptr, nullptr, err.p, stage.p,
                               STAGE_CAP, maxrow);
        SA_HIP_CHECK(hipGetLastError());
        SA_HIP_CHECK(hipStreamSynchronize(s));      // (the lists are freed here)
    }
    auto h_cnt = cnt.to_host(s);
    {   // small MISes whose neighbours did not fit the small tables: once more, full size
        std::vector<int> again;
        for (int m = 0; m < nm; ++m)
            if (h_cnt[m] < 0) again.push_back(m);
        if (!again.empty()) {
            DBuf<int> d_again;
            d_again.from_host(again, s);
            hipLaunchKernelGGL((rap_symbolic_kernel<RAP_NT, RAP_HASH>), dim3((unsigned)again.size()), dim3(RAP_NT), 0, s, d_again.p, 0,
                               rel.mis2d_I.p, rel.mis2d_J.p, A.rowptr.p, A.col.p, rel.mises.p, d_k, cnt.p, nullptr, nullptr, err.p, stage.p,
                               STAGE_CAP, maxrow);
            SA_HIP_CHECK(hipGetLastError());
            h_cnt = cnt.to_host(s);
        }
    }
    SA_REQUIRE(err.to_host(s)[0] == 0, "RAP: MIS neighbour table overflow");
    std::vector<int> h_nbr_ptr((size_t)nm + 1, 0);
    int cnt_max = 0;
    for (int m = 0; m < nm; ++m) { h_nbr_ptr[m + 1] = h_nbr_ptr[m] + h_cnt[m]; cnt_max = std::max(cnt_max, (int)h_cnt[m]); }
    DBuf<int> nbr_ptr, nbr((size_t)h_nbr_ptr[nm] + 1);
    nbr_ptr.from_host(h_nbr_ptr, s);
    if (cnt_max <= STAGE_CAP)      // every list was staged by the counting pass
        hipLaunchKernelGGL(rap_unstage_kernel, dim3(div_up((long)nm * STAGE_CAP, 256)), dim3(256), 0, s, nm, STAGE_CAP, stage.p,
                           nbr_ptr.p, nbr.p);
    else
        hipLaunchKernelGGL((rap_symbolic_kernel<RAP_NT, RAP_HASH>), dim3(nm), dim3(RAP_NT), 0, s, (const int *)nullptr, 1, rel.mis2d_I.p,
                           rel.mis2d_J.p, A.rowptr.p, A.col.p, rel.mises.p, d_k, cnt.p, nbr_ptr.p, nbr.p, err.p, (int *)nullptr, 0,
                           maxrow);
    SA_HIP_CHECK(hipGetLastError());
    auto h_nbr = nbr.to_host(s);
    // row pointers of Ac and LDS sizing
    std::vector<roff_t> crow((size_t)nc + 1, 0);
    int64_t nnz = 0;
    size_t need_max = 0, small_max = 0;
    // doubles of LDS a MIS wants in a one-wavefront workgroup: tables + its k output rows + all its rows of T, but no more than
    // 8 KB unless k + 8 rows need it (the kernel walks the MIS's rows in chunks of what fits)
    std::vector<int> small_need((size_t)nm, 0);
    for (int m = 0; m < nm; ++m) {
        if (h_k[m] == 0) continue;
        int ncol = 0;
        for (int t = h_nbr_ptr[m]; t < h_nbr_ptr[m + 1]; ++t) ncol += h_k[h_nbr[t]];
        for (int v = 0; v < h_k[m]; ++v) {
            crow[(size_t)h_coloff[m] + v + 1] = ncol;
            nnz += ncol;
        }
        // all k rows of the block + 1 row of T if that fits 160 KiB, else 2 rows (multi-pass)
        size_t need = (size_t)(3 * h_cnt[m] + 2) + (size_t)(h_k[m] + 1) * ncol;
        const size_t cap = 160 * 1024 / 8;
        if (need > cap) need = std::max((size_t)(3 * h_cnt[m] + 2) + 2 * (size_t)ncol, std::min(need, cap));
        if (need > need_max) need_max = need;
        small_max = std::max(small_max, (size_t)(3 * h_cnt[m] + 2) + (size_t)(h_k[m] + 8) * ncol);
        {
            const size_t tables = (size_t)(3 * h_cnt[m] + 2);
            const size_t all = tables + (size_t)(h_k[m] + hrel.mis_to_dof.row_size(m)) * ncol, floor8 = tables + (size_t)(h_k[m] + 8) * ncol;
            small_need[m] = (int)std::min<size_t>(std::min(all, std::max((size_t)1024, floor8)), 1u << 30);
        }
    }
    for (int i = 0; i < nc; ++i) crow[i + 1] += crow[i];
    // Blocks that fit 64 KiB in one pass get by with the k output rows + >= 8 rows of T per chunk of
    // MIS rows (16 KiB floor): the kernel is bound by the latency of its dependent gathers, and
    // the smaller footprint puts 8 workgroups instead of 2 on a CU.  Wider blocks keep what they need.
    size_t lds_doubles = std::max((size_t)2048, small_max);
    if (need_max > 8192) lds_doubles = need_max;
    SA_REQUIRE(lds_doubles * 8 <= 160 * 1024, "RAP: MIS block too wide for LDS");
    SA_HIP_CHECK(hipFuncSetAttribute((const void *)rap_numeric_kernel<RAP_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SA_HIP_CHECK(hipFuncSetAttribute((const void *)rap_numeric_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Ac.nnz = nnz;
    Ac.rowptr.from_host(crow, s);
    Ac.col.alloc((size_t)nnz);
    Ac.val.alloc((size_t)nnz);
    // Several ranks: the MIS row blocks are split into `world` contiguous ranges balanced by their
    // non-zeros; a rank computes its range only, the caller all-gathers col / val by nnz_off.
    int m_lo = 0, m_hi = nm;
    if (world > 1 && nnz_off) {
        std::vector<int> mb((size_t)world + 1, nm);
        mb[0] = 0;
        int r = 0;
        for (int m = 0; m < nm && r + 1 < world; ++m) {
            const int64_t done = crow[(size_t)h_coloff[m + 1]];   // nnz of the rows of MISes 0..m
            while (r + 1 < world && done >= nnz * (r + 1) / world) mb[++r] = m + 1;
        }
        mb[world] = nm;
        nnz_off->assign((size_t)world + 1, 0);
        for (int q = 0; q <= world; ++q) (*nnz_off)[q] = crow[(size_t)h_coloff[mb[q]]];
        m_lo = mb[rank];
        m_hi = mb[rank + 1];
    }
    // the MISes of this rank's range in two lists: few dofs (one wavefront per MIS, the LDS those need), the others
    std::vector<int> small_list, big_list;
    size_t small_lds = 0;
    for (int m = m_lo; m < m_hi; ++m) {
        if (h_k[m] == 0) continue;
        if (hrel.mis_to_dof.row_size(m) <= RAP_SMALL_ROWS && small_need[m] <= 2048) {
            small_list.push_back(m);
            small_lds = std::max(small_lds, (size_t)small_need[m]);
        } else big_list.push_back(m);
    }
    DBuf<int> d_small, d_big;
    d_small.from_host(small_list, s);
    d_big.from_host(big_list, s);
    if (!big_list.empty())
        hipLaunchKernelGGL(rap_numeric_kernel<RAP_NT>, dim3((unsigned)big_list.size()), dim3(RAP_NT), lds_doubles * 8, s, d_big.p, 0,
                           rel.mis2d_I.p, rel.mis2d_J.p, A.rowptr.p, A.col.p, A.val.p, rel.mises.p,
                           rel.dof_row_in_mis.p, d_k, d_coloff, d_u_off, U, nbr_ptr.p, nbr.p, Ac.rowptr.p,
                           Ac.col.p, Ac.val.p, (int)lds_doubles);
    if (!small_list.empty()) {
        const size_t sl = std::max((size_t)256, small_lds);
        hipLaunchKernelGGL(rap_numeric_kernel<64>, dim3((unsigned)small_list.size()), dim3(64), sl * 8, s, d_small.p, 0,
                           rel.mis2d_I.p, rel.mis2d_J.p, A.rowptr.p, A.col.p, A.val.p, rel.mises.p,
                           rel.dof_row_in_mis.p, d_k, d_coloff, d_u_off, U, nbr_ptr.p, nbr.p, Ac.rowptr.p,
                           Ac.col.p, Ac.val.p, (int)sl);
    }
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));  // nbr buffers are freed on return
    profiler().end(s, "rap", 12.0 * (double)(A.nnz + nnz) + 4.0 * A.nrows, 0.0);
    Ac.lanes_per_row = pick_lanes_per_row(nnz, nc);
}

}  // namespace saamge_amd
