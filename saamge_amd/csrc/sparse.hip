// CSR SpMV family for gfx950: L lanes of a 64-wide wavefront cooperate on one row
// (L chosen from the average row length), val/col streamed with non-temporal loads so
// the XCD L2 keeps the gathered x entries, fused epilogues for the residual, the
// prolongation-add and the polynomial-smoother step.  HBM-bound: 12 B per nonzero.
#include "sparse.h"
#include <climits>

namespace saamge_amd {

enum { MODE_PLAIN = 0, MODE_RESIDUAL = 1, MODE_ADD = 2, MODE_SMOOTH = 3 };

template <int L, int MODE>
__global__ __launch_bounds__(256) void spmv_kernel(int nrows, const roff_t *__restrict__ rowptr,
                                                   const int *__restrict__ col,
                                                   const double *__restrict__ val,
                                                   const double *__restrict__ x,
                                                   double *__restrict__ y,
                                                   const double *__restrict__ b,
                                                   const double *__restrict__ dinv, double scale,
                                                   const double *__restrict__ xrow) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (L - 1);
    const long row = gtid / L;
    if (row >= nrows) return;  // whole L-group leaves together (256 % L == 0)
    const roff_t beg = rowptr[row], end = rowptr[row + 1];
    double sum = 0.0;
    for (roff_t k = beg + lane; k < end; k += L) {
        const double v = __builtin_nontemporal_load(val + k);
        const int c = __builtin_nontemporal_load(col + k);
        sum = fma(v, x[c], sum);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, L);
    if (lane == 0) {
        if (MODE == MODE_PLAIN) {
            y[row] = sum;
        } else if (MODE == MODE_RESIDUAL) {
            y[row] = b[row] - sum;
        } else if (MODE == MODE_ADD) {
            y[row] += sum;
        } else {  // x_out = x_in + scale * dinv_neg * (A x - b)
            y[row] = xrow[row] + scale * (dinv[row] * (sum - b[row]));
        }
    }
}

// SELL-64: one lane per row, the wavefront walks its slice column by column; val/col loads are
// 512 B / 256 B contiguous per step and, for stencil-like matrices, so are the x gathers.
// Slices whose entries use at most 64 distinct column offsets (col - row) -- every slice of a
// stencil matrix -- carry ONE BYTE per entry instead of the 4-byte column: a code into the slice's
// offset table, which sits one entry per lane in a register and is read with a cross-lane
// permute.  9 instead of 12 bytes per stored entry on the level that dominates the solve.
// pair-coded slices: ntab = 256 + pairs (own table at tab[64 slice ..]) or 512 + pairs (the table of the slice's tile of
// four, at the tile's first slice: sell_code_kernel's merge)
__device__ __forceinline__ int pair_count(int nt) { return nt >= 512 ? nt - 512 : nt - 256; }
__device__ __forceinline__ size_t pair_table_at(int gslice, int nt) { return (size_t)(nt >= 512 ? (gslice & ~3) : gslice) * 64; }

// general path: any mix of slice formats, any width.  Returns the row's sum; `slice` / `gslice` are wave-uniform.
__device__ __forceinline__ double sell_row_general(int lane, int grow, int gslice, roff_t beg, roff_t end, int nt,
                                                   const int *__restrict__ col, const double *__restrict__ val,
                                                   const int *__restrict__ tab, const unsigned *__restrict__ codes,
                                                   const double *__restrict__ x, const double *__restrict__ vtab) {
    const int *cp = col + beg + lane;
    const double *vp = val + beg + lane;
    const int w = (int)((end - beg) >> 6);
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    if (nt >= 256) {
        // pair-coded slice: the byte indexes a table of (offset, VALUE) pairs -- at most 64 distinct ones in
        // the slice, which is every slice of a constant-coefficient stencil matrix (the whole fine level of the
        // headline problem).  Neither columns nor values are streamed: 1 byte per stored entry instead of 12.
        const int np = pair_count(nt);
        const int mytab = (lane < np) ? tab[pair_table_at(gslice, nt) + lane] : 0;
        const double myval = (lane < np) ? vtab[pair_table_at(gslice, nt) + lane] : 0.0;
        const unsigned *wp = codes + ((size_t)(beg >> 2) + (size_t)gslice * 64 + lane);
        auto quad = [&](unsigned cw) {
            const int i0 = (int)(cw & 255u), i1 = (int)((cw >> 8) & 255u), i2 = (int)((cw >> 16) & 255u), i3 = (int)(cw >> 24);
            const int c0 = grow + __shfl(mytab, i0), c1 = grow + __shfl(mytab, i1);
            const int c2 = grow + __shfl(mytab, i2), c3 = grow + __shfl(mytab, i3);
            const double v0 = __shfl(myval, i0), v1 = __shfl(myval, i1), v2 = __shfl(myval, i2), v3 = __shfl(myval, i3);
            s0 = fma(v0, x[c0], s0);
            s1 = fma(v1, x[c1], s1);
            s0 = fma(v2, x[c2], s0);
            s1 = fma(v3, x[c3], s1);
        };
        if (w <= 32) {
            // every code word of the row is requested before the first one is used (the loop would wait for one
            // word per trip: load -> permute -> gather is a dependent chain)
            unsigned cws[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cws[q] = (4 * q < w) ? __builtin_nontemporal_load(wp + 64 * q) : 0u;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (4 * q + 4 <= w) quad(cws[q]);
                else if (4 * q < w) {
                    unsigned cw = cws[q];
                    for (int kk = 4 * q; kk < w; ++kk, cw >>= 8) {
                        const int i0 = (int)(cw & 255u);
                        s0 = fma(__shfl(myval, i0), x[grow + __shfl(mytab, i0)], s0);
                    }
                }
            }
            k = w;
        }
        for (; k + 4 <= w; k += 4) quad(__builtin_nontemporal_load(wp + 16 * k));
        if (k < w) {
            unsigned cw = __builtin_nontemporal_load(wp + 16 * k);
            for (; k < w; ++k, cw >>= 8) {
                const int i0 = (int)(cw & 255u);
                s0 = fma(__shfl(myval, i0), x[grow + __shfl(mytab, i0)], s0);
            }
        }
    } else if (nt >= 0) {
        const int mytab = tab[(size_t)gslice * 64 + lane];
        const unsigned *wp = codes + ((size_t)(beg >> 2) + (size_t)gslice * 64 + lane);
        for (; k + 4 <= w; k += 4) {
            const unsigned cw = __builtin_nontemporal_load(wp + 16 * k);
            const double v0 = __builtin_nontemporal_load(vp + 64 * k), v1 = __builtin_nontemporal_load(vp + 64 * (k + 1));
            const double v2 = __builtin_nontemporal_load(vp + 64 * (k + 2)), v3 = __builtin_nontemporal_load(vp + 64 * (k + 3));
            const int c0 = grow + __shfl(mytab, (int)(cw & 255u)), c1 = grow + __shfl(mytab, (int)((cw >> 8) & 255u));
            const int c2 = grow + __shfl(mytab, (int)((cw >> 16) & 255u)), c3 = grow + __shfl(mytab, (int)(cw >> 24));
            s0 = fma(v0, x[c0], s0);
            s1 = fma(v1, x[c1], s1);
            s0 = fma(v2, x[c2], s0);
            s1 = fma(v3, x[c3], s1);
        }
        if (k < w) {
            unsigned cw = __builtin_nontemporal_load(wp + 16 * k);
            for (; k < w; ++k, cw >>= 8)
                s0 = fma(__builtin_nontemporal_load(vp + 64 * k), x[grow + __shfl(mytab, (int)(cw & 255u))], s0);
        }
    }
    for (; k + 4 <= w; k += 4) {
        const int c0 = __builtin_nontemporal_load(cp + 64 * k), c1 = __builtin_nontemporal_load(cp + 64 * (k + 1));
        const int c2 = __builtin_nontemporal_load(cp + 64 * (k + 2)), c3 = __builtin_nontemporal_load(cp + 64 * (k + 3));
        const double v0 = __builtin_nontemporal_load(vp + 64 * k), v1 = __builtin_nontemporal_load(vp + 64 * (k + 1));
        const double v2 = __builtin_nontemporal_load(vp + 64 * (k + 2)), v3 = __builtin_nontemporal_load(vp + 64 * (k + 3));
        s0 = fma(v0, x[c0], s0);
        s1 = fma(v1, x[c1], s1);
        s0 = fma(v2, x[c2], s0);
        s1 = fma(v3, x[c3], s1);
    }
    for (; k < w; ++k) s0 = fma(__builtin_nontemporal_load(vp + 64 * k), x[__builtin_nontemporal_load(cp + 64 * k)], s0);
    return s0 + s1;
}

struct alignas(16) PairEntry {
    int off, pad;
    double val;
};
// (Measured, round 4: the whole 16-byte entry with ONE LDS instruction -- ds_read_b128 instead of the ds_read_b32 + ds_read_b64 a
// member-wise read becomes -- 233 against 230 us per fine-level smoother step: the LDS data path, not its instruction count.)

// ---- the SELL-64 kernels of the whole SpMV family: one wavefront per slice, one lane per row ----------------------
// What bounds them (MI355X, 257^3 rows x 27 entries, counters + tools/spmv_lab, tools/vmem_rate): not HBM.  A
// vector-memory wave-instruction of 8 or 16 bytes per lane occupies the CU's address / L1 path for ~17 cycles
// whatever it returns (4 bytes per lane: 5-9), and a row-per-lane SpMV issues one such instruction PER STORED ENTRY
// for the gather of x: 27 + 17 others per slice = ~620 of the ~890 cycles a CU spends per slice, at 2.9 TB/s of the
// bytes the format needs (a pure stream of the same bytes: 5.1 TB/s).  Before that the cross-lane permutes of the
// table look-ups (96 LDS instructions x 8 cycles per slice) held the same place.  Hence:
//   * sell_staged_kernel: where the offsets of a 256-row tile cluster into few runs, the workgroup loads the x-segments
//     those runs touch into LDS with 16-byte coalesced loads (27 gathers per wavefront -> ~7 loads) and the products
//     read LDS; the slice's (offset, value) table sits in LDS too, one 16-byte entry per lane, read with plain LDS
//     loads (a broadcast for the lanes that share an entry) instead of three permutes per entry.
//   * sell_slice: the same per slice with gathers from global memory (tiles that cannot be staged), up to 16 in
//     flight; offset-coded, plain and wide slices go through sell_row_general.
//   * Consecutive workgroups are dealt to the XCDs round-robin (block -> XCD = block % 8, position = block / 8): XCD k
//     walks the k-th contiguous eighth of the tiles, so that the x-planes a stencil row touches are fetched into ONE L2
//     instead of all eight (PMC: 2.0 -> 1.14 GB per launch, the format's own bytes being 1.12 GB; speed only, any
//     placement gives the same result).
//   * The slice is wave-uniform (readfirstlane): its offsets, width and format come through the scalar cache.
// Same arithmetic and the same order of additions on every path (two accumulators over whole groups of four entries,
// the tail into the first).
template <int MODE>
__device__ __forceinline__ void sell_slice(PairEntry *lt, int nrows, int row0, long row, int slice, int fast_ok,
                                           const roff_t *__restrict__ sptr, const int *__restrict__ col,
                                           const double *__restrict__ val, const int *__restrict__ ntab,
                                           const int *__restrict__ tab, const unsigned *__restrict__ codes,
                                           const double *__restrict__ x, double *__restrict__ y,
                                           const double *__restrict__ b, const double *__restrict__ dinv, double scale,
                                           const double *__restrict__ xrow, const double *__restrict__ vtab) {
    const int lane = threadIdx.x & 63;
    if ((long)slice * 64 >= nrows) return;
    const roff_t beg = sptr[slice], end = sptr[slice + 1];
    const int gslice = (row0 >> 6) + slice;
    const int nt = ntab[gslice];
    const int w = (int)((end - beg) >> 6);
    const bool live = row < nrows;
    const int grow = row0 + (int)row;
    double e_b = 0.0, e_d = 0.0, e_x = 0.0, sum;
    if (fast_ok && nt >= 256 && w <= 32) {
        const int np = pair_count(nt);
        const unsigned *wp = codes + ((size_t)(beg >> 2) + (size_t)gslice * 64 + lane);
        unsigned cws[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cws[q] = (4 * q < w) ? __builtin_nontemporal_load(wp + 64 * q) : 0u;
        const int mytab = (lane < np) ? tab[pair_table_at(gslice, nt) + lane] : 0;
        const double myval = (lane < np) ? vtab[pair_table_at(gslice, nt) + lane] : 0.0;
        if (MODE == MODE_RESIDUAL && live) e_b = b[row];
        if (MODE == MODE_ADD && live) e_x = y[row];
        if (MODE == MODE_SMOOTH && live) { e_b = b[row]; e_d = dinv[row]; e_x = xrow[row]; }
        lt[lane] = PairEntry{mytab, 0, myval};      // (a region private to this wavefront)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const char *xb = (const char *)x;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (16 * h < w) {
                double xs[16], vs[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int idx = (int)((cws[4 * h + (j >> 2)] >> (8 * (j & 3))) & 255u);
                    const PairEntry e = lt[idx];
                    vs[j] = e.val;
                    // (entries past the slice's width carry code 0: no address is formed from them -- table entry 0 belongs
                    // to the tile's FIRST row and can point outside x from a later one)
                    xs[j] = (16 * h + j < w) ? *(const double *)(xb + ((unsigned)(grow + e.off) << 3)) : 0.0;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (16 * h + j < w) {
                        if ((j & 1) && 16 * h + j < (w & ~3)) s1 = fma(vs[j], xs[j], s1);
                        else s0 = fma(vs[j], xs[j], s0);
                    }
                }
            }
        }
        sum = s0 + s1;
    } else {
        sum = sell_row_general(lane, grow, gslice, beg, end, nt, col, val, tab, codes, x, vtab);
        if (MODE == MODE_RESIDUAL && live) e_b = b[row];
        if (MODE == MODE_ADD && live) e_x = y[row];
        if (MODE == MODE_SMOOTH && live) { e_b = b[row]; e_d = dinv[row]; e_x = xrow[row]; }
    }
    if (!live) return;
    if (MODE == MODE_PLAIN) {
        y[row] = sum;
    } else if (MODE == MODE_RESIDUAL) {
        y[row] = e_b - sum;
    } else if (MODE == MODE_ADD) {
        y[row] = e_x + sum;
    } else {
        y[row] = e_x + scale * (e_d * (sum - e_b));
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void sell_spmv_kernel(int nrows, int row0, int nblocks, int per_xcd, int fast_ok,
                                                        const roff_t *__restrict__ sptr,
                                                        const int *__restrict__ col,
                                                        const double *__restrict__ val,
                                                        const int *__restrict__ ntab,
                                                        const int *__restrict__ tab,
                                                        const unsigned *__restrict__ codes,
                                                        const double *__restrict__ x,
                                                        double *__restrict__ y,
                                                        const double *__restrict__ b,
                                                        const double *__restrict__ dinv, double scale,
                                                        const double *__restrict__ xrow,
                                                        const double *__restrict__ vtab) {
    __shared__ PairEntry ltab[4][64];
    const int blk = per_xcd > 0 ? (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (blk >= nblocks) return;
    const long row = (long)blk * 256 + threadIdx.x;
    const int slice = __builtin_amdgcn_readfirstlane((int)(row >> 6));
    sell_slice<MODE>(ltab[threadIdx.x >> 6], nrows, row0, row, slice, fast_ok, sptr, col, val, ntab, tab, codes, x, y, b,
                     dinv, scale, xrow, vtab);
}

// Staged tiles only (the others are left to sell_tiles_kernel).  row0 must be a multiple of 256 (tiles are global);
// dynamic LDS: stage_cap doubles + 4 slice tables.
template <int MODE>
__global__ __launch_bounds__(256) void sell_staged_kernel(int nrows, int row0, int nblocks, int per_xcd, int stage_cap,
                                                          int ncols, int one_table, const roff_t *__restrict__ sptr,
                                                          const int *__restrict__ ntab,
                                                          const int *__restrict__ tab,
                                                          const unsigned *__restrict__ codes,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ y,
                                                          const double *__restrict__ b,
                                                          const double *__restrict__ dinv, double scale,
                                                          const double *__restrict__ xrow,
                                                          const double *__restrict__ vtab,
                                                          const int *__restrict__ tile_nseg,
                                                          const int2 *__restrict__ tile_seg) {
    extern __shared__ __align__(16) double lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // one_table: every staged tile of the operator shares one pair table among its four slices -- 1 KB of LDS for it
    // instead of 4 (one more workgroup per CU); the wavefront's row offset then goes into the x-address, not the table
    PairEntry *lt = (PairEntry *)(lds + stage_cap) + (one_table ? 0 : 64 * wv);
    const int blk = per_xcd > 0 ? (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (blk >= nblocks || (long)blk * 256 + 256 > nrows) return;
    const long row = (long)blk * 256 + threadIdx.x;
    const int slice = __builtin_amdgcn_readfirstlane((int)(row >> 6));
    const int gtile = (row0 >> 8) + blk;
    const int nseg = tile_nseg[gtile];
    if (nseg == 0) return;
    // this slice's streams first: they are in flight while the segments are staged
    const roff_t beg = sptr[slice], end = sptr[slice + 1];
    const int gslice = (row0 >> 6) + slice;
    const int nt = ntab[gslice];
    const int np = pair_count(nt);
    const int w = (int)((end - beg) >> 6);
    const unsigned *wp = codes + ((size_t)(beg >> 2) + (size_t)gslice * 64 + lane);
    unsigned cws[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cws[q] = (4 * q < w) ? __builtin_nontemporal_load(wp + 64 * q) : 0u;
    const int mytab = (lane < np) ? tab[pair_table_at(gslice, nt) + lane] : 0;
    const double myval = (lane < np) ? vtab[pair_table_at(gslice, nt) + lane] : 0.0;
    double e_b = 0.0, e_d = 0.0, e_x = 0.0;
    if (MODE == MODE_RESIDUAL) e_b = b[row];
    if (MODE == MODE_ADD) e_x = y[row];
    if (MODE == MODE_SMOOTH) { e_b = b[row]; e_d = dinv[row]; e_x = xrow[row]; }
    // the tile's segments: x[R0 + lo .. R0 + lo + len) -> lds[pre ..), 16 bytes per lane (starts and lengths are even).
    // Four segments at a time: the loads of their first two passes (512 doubles per pass of the workgroup), then the
    // LDS stores -- one memory latency per batch, not one per segment; the thread assignment rotates by one wavefront
    // per segment so that the partial last passes land on different wavefronts.
    const int2 *sg = tile_seg + (size_t)gtile * SELL_SEG_MAX;
    const int R0 = row0 + blk * 256;
    auto fetch = [&](int g) {
        // (a tile's table is the union over its rows: near the ends of the operator a segment reaches outside
        // [0, ncols), where no row of the tile has an entry)
        double2 v = make_double2(0.0, 0.0);
        if (g >= 0 && g + 1 < ncols) v = *(const double2 *)(x + g);
        else if (g >= 0 && g < ncols) v.x = x[g];
        else if (g == -1 && ncols > 0) v.y = x[0];
        return v;
    };
    int pre = 0, mybase = 0;
    bool longer = false;
#pragma unroll
    for (int batch = 0; batch < SELL_SEG_MAX; batch += 4) {
        if (batch < nseg) {
            double2 st[8];
            int pre_k[4], len_k[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sidx = batch + k;
                pre_k[k] = pre;
                len_k[k] = 0;
                if (sidx < nseg) {
                    const int2 d = sg[sidx];
                    len_k[k] = d.y;
                    const int i = 2 * (int)((threadIdx.x + 64u * sidx) & 255u);
                    if (i < d.y) st[2 * k] = fetch(R0 + d.x + i);
                    if (i + 512 < d.y) st[2 * k + 1] = fetch(R0 + d.x + i + 512);
                    longer = longer || d.y > 1024;
                    if (mytab >= d.x) mybase = pre - d.x;      // (segments ascend: the last one at or below the offset holds it)
                    pre += d.y;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 2 * (int)((threadIdx.x + 64u * (batch + k)) & 255u);
                if (i < len_k[k]) *(double2 *)(lds + pre_k[k] + i) = st[2 * k];
                if (i + 512 < len_k[k]) *(double2 *)(lds + pre_k[k] + i + 512) = st[2 * k + 1];
            }
        }
    }
    if (longer) {      // segments beyond two passes (1024 doubles): the rest of them
        int p2 = 0;
        for (int sidx = 0; sidx < nseg; ++sidx) {
            const int2 d = sg[sidx];
            for (int i = 2 * (int)((threadIdx.x + 64u * sidx) & 255u) + 1024; i < d.y; i += 512) *(double2 *)(lds + p2 + i) = fetch(R0 + d.x + i);
            p2 += d.y;
        }
    }
    // byte offset of the x-entry of the wavefront's (one_table: the tile's) first row in lds
    if (!one_table) lt[lane] = PairEntry{(mybase + mytab + 64 * wv) << 3, 0, myval};
    else if (wv == 0) lt[lane] = PairEntry{(mybase + mytab) << 3, 0, myval};
    __syncthreads();
    const char *lb = (const char *)lds + 8 * (lane + (one_table ? 64 * wv : 0));
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (4 * g + 4 <= w) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const PairEntry e = lt[(cws[g] >> (8 * e4)) & 255u];
                const double xv = *(const double *)(lb + e.off);
                if (e4 & 1) s1 = fma(e.val, xv, s1);
                else s0 = fma(e.val, xv, s0);
            }
        } else if (4 * g < w) {
#pragma unroll
            for (int e4 = 0; e4 < 3; ++e4) {
                if (4 * g + e4 < w) {
                    const PairEntry e = lt[(cws[g] >> (8 * e4)) & 255u];
                    s0 = fma(e.val, *(const double *)(lb + e.off), s0);
                }
            }
        }
    }
    const double sum = s0 + s1;
    if (MODE == MODE_PLAIN) {
        y[row] = sum;
    } else if (MODE == MODE_RESIDUAL) {
        y[row] = e_b - sum;
    } else if (MODE == MODE_ADD) {
        y[row] = e_x + sum;
    } else {
        y[row] = e_x + scale * (e_d * (sum - e_b));
    }
}

// ---- the staged kernel with descriptor-free streams (round 4) -------------------------------------------------------
// sell_staged_kernel starts with a chain: scalar loads of the slice's offset, width and table size (streaming data: they miss
// every cache), THEN the vector loads whose addresses they give -- two memory latencies before the first product, a quarter
// of a tile's ~8 us life at eight workgroups per CU (SQ_WAIT_ANY 66 %, round 3's counters).  Here the code words of the
// staged tiles live a second time in a REGULAR layout -- word q of the row of thread t of tile T at (T wq + q) 256 + t, wq =
// the widest staged row in words -- and the widths of the four slices ride in one descriptor word per tile next to the
// segment count: codes, right-hand side, diagonal, x-row and the tile's one table are requested before any descriptor
// has arrived; only the x-segments wait for their (scalar-loaded) descriptors.  Operators whose staged tiles all share one
// table per tile (sell_one_table); same products in the same order as sell_staged_kernel.
__global__ __launch_bounds__(256) void sell_regular_codes_kernel(int ntiles, int wq, int ncols, const roff_t *__restrict__ sptr,
                                                                 const int *__restrict__ tile_nseg, const int2 *__restrict__ tile_seg,
                                                                 const unsigned *__restrict__ codes,
                                                                 unsigned *__restrict__ codesR, int *__restrict__ tile_desc) {
    const int t = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nseg = tile_nseg[t];
    if (nseg == 0) {
        if (threadIdx.x == 0) tile_desc[t] = 0;
        return;
    }
    const int slice = 4 * t + wv;
    const roff_t beg = sptr[slice];
    const int w = (int)((sptr[slice + 1] - beg) >> 6);
    const unsigned *wp = codes + ((size_t)(beg >> 2) + (size_t)slice * 64 + lane);
    for (int q = 0; q < wq; ++q) codesR[((size_t)t * wq + q) * 256 + threadIdx.x] = (4 * q < w) ? wp[64 * q] : 0u;
    __shared__ int ws[4];
    if (lane == 0) ws[wv] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        // bit 7: every segment lies inside x[0 .. ncols) -- the staging loads of such a tile (all but the first and last few
        // of an operator) need no bounds checks
        bool inside = true;
        for (int q = 0; q < nseg; ++q) {
            const int2 d = tile_seg[(size_t)t * SELL_SEG_MAX + q];
            const long g0 = (long)t * 256 + d.x;
            inside = inside && g0 >= 0 && g0 + d.y <= ncols;
        }
        tile_desc[t] = nseg | (inside ? 128 : 0) | (ws[0] << 8) | (ws[1] << 14) | (ws[2] << 20) | (ws[3] << 26);
    }
}

// The few tiles the staging plan leaves out (an operator whose row count is not a multiple of 256: its last tile), done by the
// trailing workgroups of sell_staged2_kernel -- what sell_tiles_kernel does in a launch of its own, 5 us behind every one of
// the 456 smoother steps of the headline's solve for a single row.  Through sell_slice's general path (fast_ok = 0: the same
// sums in the same order; its sixteen gathers in flight would take the kernel from 63 to 84 registers and from eight to six
// workgroups per CU).
struct SellLeftover {
    int grid_main, n;            // workgroups of the staged tiles; tiles left over (0: none are done here)
    const int *tiles;
    const roff_t *sptr;
    const int *col;
    const double *val;
    const int *ntab;
    const unsigned *codes;
};
template <int MODE>
__global__ __launch_bounds__(256) void sell_staged2_kernel(SellLeftover left, int nrows, int row0, int nblocks, int per_xcd, int stage_cap,
                                                           int ncols, int wq, const unsigned *__restrict__ codesR,
                                                           const int *__restrict__ tab, const double *__restrict__ vtab,
                                                           const int *__restrict__ tile_desc, const int2 *__restrict__ tile_seg,
                                                           const double *__restrict__ x, double *__restrict__ y,
                                                           const double *__restrict__ b, const double *__restrict__ dinv,
                                                           double scale, const double *__restrict__ xrow,
                                                           const unsigned char *__restrict__ dcode = nullptr,
                                                           const double *__restrict__ dtab = nullptr) {
    extern __shared__ __align__(16) double lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= left.grid_main) {      // (a trailing workgroup: one of the tiles that are not staged)
        const long lrow0 = (long)left.tiles[blockIdx.x - left.grid_main] * 256 - row0;      // the tile's first row, local to the range
        if (lrow0 < 0 || lrow0 >= nrows) return;
        const long trow = lrow0 + threadIdx.x;
        const int slice = __builtin_amdgcn_readfirstlane((int)(trow >> 6));
        sell_slice<MODE>((PairEntry *)lds + 64 * wv, nrows, row0, trow, slice, 0, left.sptr, left.col, left.val, left.ntab, tab, left.codes, x, y,
                         b, dinv, scale, xrow, vtab);
        return;
    }
    PairEntry *lt = (PairEntry *)(lds + stage_cap);
    const int blk = per_xcd > 0 ? (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (blk >= nblocks || (long)blk * 256 + 256 > nrows) return;
    const long row = (long)blk * 256 + threadIdx.x;
    const int gtile = (row0 >> 8) + blk;
    // every stream whose address the block index gives: requested before any descriptor is looked at
    const unsigned *wp = codesR + ((size_t)gtile * wq) * 256 + threadIdx.x;
    unsigned cws[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cws[q] = (q < wq) ? __builtin_nontemporal_load(wp + 256 * q) : 0u;
    double e_b = 0.0, e_d = 0.0, e_x = 0.0;
    // (b and D^-1 are read once per application: streamed past the caches like the code words, so that the L2 keeps x)
    if (MODE == MODE_RESIDUAL) e_b = __builtin_nontemporal_load(b + row);
    if (MODE == MODE_ADD) e_x = y[row];
    // (the smoother's x-row is x itself at the tile's own rows: taken from the staged segment that holds them, below)
    // (D^-1 as a byte code into a small table where the operator's rows repeat: 1 instead of 8 bytes per row)
    if (MODE == MODE_SMOOTH) { e_b = __builtin_nontemporal_load(b + row); e_d = dcode ? dtab[__builtin_nontemporal_load(dcode + row)] : __builtin_nontemporal_load(dinv + row); }
    int mytab = 0;
    double myval = 0.0;
    if (wv == 0) {          // the tile's one table sits at its first slice
        mytab = tab[(size_t)gtile * 256 + lane];
        myval = vtab[(size_t)gtile * 256 + lane];
    }
    const int desc = tile_desc[gtile];
    const int nseg = desc & 127;
    if (nseg == 0) return;      // (left to sell_tiles_kernel)
    const int w = __builtin_amdgcn_readfirstlane((desc >> (8 + 6 * wv)) & 63);
    const int2 *sg = tile_seg + (size_t)gtile * SELL_SEG_MAX;
    const int R0 = row0 + blk * 256;
    int mybase = 0, zbase = -1;      // zbase: LDS index of x[R0] when a segment holds the tile's own rows
    // the tile's segments x[R0 + lo .. R0 + lo + len) -> lds[pre ..), 16 bytes per lane, four segments per batch (their loads
    // first, then the LDS stores).  CHECK = false: the descriptor says that every segment lies inside x (bit 7)
    auto stage = [&](auto check_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        auto fetch = [&](int g) {
            if (!CHECK) return *(const double2 *)(x + g);
            double2 v = make_double2(0.0, 0.0);
            if (g >= 0 && g + 1 < ncols) v = *(const double2 *)(x + g);
            else if (g >= 0 && g < ncols) v.x = x[g];
            else if (g == -1 && ncols > 0) v.y = x[0];
            return v;
        };
        int pre = 0;
        bool longer = false;
#pragma unroll
        for (int batch = 0; batch < SELL_SEG_MAX; batch += 4) {
            if (batch < nseg) {
                double2 st[8];
                int pre_k[4], len_k[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int sidx = batch + k;
                    pre_k[k] = pre;
                    len_k[k] = 0;
                    if (sidx < nseg) {
                        const int2 d = sg[sidx];
                        len_k[k] = d.y;
                        const int i = 2 * (int)((threadIdx.x + 64u * sidx) & 255u);
                        if (i < d.y) st[2 * k] = fetch(R0 + d.x + i);
                        if (i + 512 < d.y) st[2 * k + 1] = fetch(R0 + d.x + i + 512);
                        longer = longer || d.y > 1024;
                        if (wv == 0 && mytab >= d.x) mybase = pre - d.x;
                        if (MODE == MODE_SMOOTH && d.x <= 0 && d.x + d.y >= 256) zbase = pre - d.x;
                        pre += d.y;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = 2 * (int)((threadIdx.x + 64u * (batch + k)) & 255u);
                    if (i < len_k[k]) *(double2 *)(lds + pre_k[k] + i) = st[2 * k];
                    if (i + 512 < len_k[k]) *(double2 *)(lds + pre_k[k] + i + 512) = st[2 * k + 1];
                }
            }
        }
        if (longer) {
            int p2 = 0;
            for (int sidx = 0; sidx < nseg; ++sidx) {
                const int2 d = sg[sidx];
                for (int i = 2 * (int)((threadIdx.x + 64u * sidx) & 255u) + 1024; i < d.y; i += 512) *(double2 *)(lds + p2 + i) = fetch(R0 + d.x + i);
                p2 += d.y;
            }
        }
    };
    if (desc & 128) stage(std::false_type());
    else stage(std::true_type());
    if (wv == 0) lt[lane] = PairEntry{(mybase + mytab) << 3, 0, myval};
    if (MODE == MODE_SMOOTH && zbase < 0) e_x = xrow[row];      // (no segment holds the own rows: an operator without a diagonal run)
    __syncthreads();
    if (MODE == MODE_SMOOTH && zbase >= 0) e_x = lds[zbase + threadIdx.x];
    const char *lb = (const char *)lds + 8 * (lane + 64 * wv);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (4 * g + 4 <= w) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const PairEntry e = lt[(cws[g] >> (8 * e4)) & 255u];
                const double xv = *(const double *)(lb + e.off);
                if (e4 & 1) s1 = fma(e.val, xv, s1);
                else s0 = fma(e.val, xv, s0);
            }
        } else if (4 * g < w) {
#pragma unroll
            for (int e4 = 0; e4 < 3; ++e4) {
                if (4 * g + e4 < w) {
                    const PairEntry e = lt[(cws[g] >> (8 * e4)) & 255u];
                    s0 = fma(e.val, *(const double *)(lb + e.off), s0);
                }
            }
        }
    }
    const double sum = s0 + s1;
    if (MODE == MODE_PLAIN) {
        y[row] = sum;
    } else if (MODE == MODE_RESIDUAL) {
        y[row] = e_b - sum;
    } else if (MODE == MODE_ADD) {
        y[row] = e_x + sum;
    } else {
        y[row] = e_x + scale * (e_d * (sum - e_b));
    }
}

// The tiles the staged kernel leaves out, from the list the staging plan made (ids of the whole operator; those
// outside the row range are skipped), one workgroup per tile through sell_slice.
template <int MODE>
__global__ __launch_bounds__(256) void sell_tiles_kernel(int ntl, const int *__restrict__ tiles, int nrows, int row0,
                                                         const roff_t *__restrict__ sptr, const int *__restrict__ col,
                                                         const double *__restrict__ val, const int *__restrict__ ntab,
                                                         const int *__restrict__ tab, const unsigned *__restrict__ codes,
                                                         const double *__restrict__ x, double *__restrict__ y,
                                                         const double *__restrict__ b, const double *__restrict__ dinv,
                                                         double scale, const double *__restrict__ xrow,
                                                         const double *__restrict__ vtab) {
    __shared__ PairEntry ltab[4][64];
    const long lrow0 = (long)tiles[blockIdx.x] * 256 - row0;      // the tile's first row, local to the range
    if (lrow0 < 0 || lrow0 >= nrows) return;
    const long row = lrow0 + threadIdx.x;
    const int slice = __builtin_amdgcn_readfirstlane((int)(row >> 6));
    sell_slice<MODE>(ltab[threadIdx.x >> 6], nrows, row0, row, slice, 1, sptr, col, val, ntab, tab, codes, x, y, b, dinv,
                     scale, xrow, vtab);
}

// x-staging plan of the pair-coded tiles (see DCsr::sell_tile_seg): one wavefront per tile of 4 slices.  The
// distinct column offsets of the four slice tables are sorted (bitonic, 256 keys in LDS).  An offset o needs
// x[R0 + o .. R0 + o + 255] for the tile's 256 rows R0..: offsets no more than 256 + STAGE_GAP apart have ranges that
// touch or overlap and become one segment [lo, hi] covering x[R0 + lo .. R0 + hi + 255] (a 27-point stencil: one
// segment per z-plane, three of 772 doubles), its start moved down to an even index (16-byte aligned loads) and its
// length made even.  Tiles with
// a slice that is not pair-coded / wider than 32, with more than SELL_SEG_MAX segments or more than SELL_STAGE_CAP
// doubles are not staged (nseg = 0) and take the gather path.
constexpr int STAGE_GAP = 32;
__global__ __launch_bounds__(64) void sell_stage_kernel(int ntiles, int nslices, int nrows, const roff_t *__restrict__ sptr,
                                                        const int *__restrict__ ntab, const int *__restrict__ tab,
                                                        int *__restrict__ tile_nseg, int2 *__restrict__ tile_seg,
                                                        int *__restrict__ max_total, int *__restrict__ unstaged) {
    __shared__ int v[256];
    __shared__ int lo[SELL_SEG_MAX], hi[SELL_SEG_MAX];
    const int t = blockIdx.x, lane = threadIdx.x;
    bool ok = (long)(4 * t + 4) * 64 <= nrows;      // four whole slices
    for (int q = 0; q < 4 && ok; ++q) {
        const int sl = 4 * t + q;
        const int nt = ntab[sl];
        const int w = (int)((sptr[sl + 1] - sptr[sl]) >> 6);
        ok = nt >= 256 && w <= 32;
        // (a tile that shares one table: listed once, by its first slice)
        if (ok) v[64 * q + lane] = (lane < pair_count(nt) && (nt < 512 || q == 0)) ? tab[pair_table_at(sl, nt) + lane] : INT_MAX;
    }
    auto give_up = [&]() {      // max_total[2]: number of tiles left to the gather kernel, listed in `unstaged`
        if (lane == 0) {
            tile_nseg[t] = 0;
            unstaged[atomicAdd(max_total + 2, 1)] = t;
        }
    };
    if (!ok) {
        give_up();
        return;
    }
    __syncthreads();
    for (int k = 2; k <= 256; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < 256; i += 64) {
                const int p = i ^ j;
                if (p > i) {
                    const int a = v[i], b = v[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { v[i] = b; v[p] = a; }
                }
            }
            __syncthreads();
        }
    // segment starts among the sorted keys (duplicates and the INT_MAX padding start nothing)
    int nseg = 0, nend = 0;
    for (int c = 0; c < 4; ++c) {
        const int i = 64 * c + lane;
        const int a = v[i];
        const int prev = i ? v[i - 1] : INT_MIN;
        const bool valid = a != INT_MAX;
        const bool start = valid && (i == 0 || (long)a - (long)prev > 256 + STAGE_GAP);
        const int next = i < 255 ? v[i + 1] : INT_MAX;
        const bool end = valid && (next == INT_MAX || (long)next - (long)a > 256 + STAGE_GAP);
        const unsigned long long sm = __ballot(start), em = __ballot(end);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (start) { const int sidx = nseg + __popcll(sm & below); if (sidx < SELL_SEG_MAX) lo[sidx] = a; }
        if (end) { const int eidx = nend + __popcll(em & below); if (eidx < SELL_SEG_MAX) hi[eidx] = a; }
        nseg += __popcll(sm);
        nend += __popcll(em);
    }
    __syncthreads();
    if (nseg > SELL_SEG_MAX) {
        give_up();
        return;
    }
    const long R0 = (long)t * 256;
    int mylo = 0, mylen = 0;
    if (lane < nseg) {
        mylo = lo[lane];
        if ((R0 + mylo) & 1) mylo -= 1;
        mylen = 256 + (hi[lane] - mylo);
        mylen += mylen & 1;
    }
    int total = mylen;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if (total > SELL_STAGE_CAP) {
        give_up();
        return;
    }
    if (lane < SELL_SEG_MAX) tile_seg[(size_t)t * SELL_SEG_MAX + lane] = make_int2(mylo, mylen);
    if (lane == 0) {
        tile_nseg[t] = nseg;
        atomicMax(max_total, total);
        atomicAdd(max_total + 1, 1);
        bool shared_all = true;
        for (int q = 0; q < 4; ++q) shared_all = shared_all && ntab[4 * t + q] >= 512;
        if (!shared_all) atomicAdd(max_total + 3, 1);      // staged tiles whose slices keep tables of their own
    }
}


// ---- operator-level pair dictionary (DCsr::sell_gpair) ------------------------------------------------------------
struct alignas(16) GPair {
    int off, pad;      // off: (col - row) * 8, a byte offset into x (operators below 2^28 columns)
    double val;
};
// x[row + offset]: scalar base + 32-bit byte offset (one add per gather)
__device__ __forceinline__ double gp_x(const double *__restrict__ x, unsigned row8, int off8) {
    return *(const double *)((const char *)x + (row8 + (unsigned)off8));
}
constexpr int GD_CAP = 1 << 18;      // hash slots
constexpr int GP_MAX = 6144;         // pairs accepted: the table has to fit LDS (96 KB; see sell_gpair_kernel)
// One wavefront per slice, one lane per row: every stored entry (padding included) looks its (col - row, value) pair
// up in a hash table shared by the whole operator, inserting it on first sight (the pair's code = order of insertion:
// run-dependent, immaterial -- a code only names its pair).
// A slot is a 16-byte record {value bits; offset, code + 1} that is written once (its second word last, with release
// order, after the slot was claimed through state[]: 0 -> 1 by atomicCAS) and never changes.  The common case -- the
// pair is there -- is ONE ordinary cached 16-byte load: a record whose second word is non-zero is complete (both words
// sit in one cache line and the first was written first) and final, whichever cache it comes from.  Only a record that
// looks empty is looked at again with device-coherent accesses: those bypass the XCD's L2 (the eight L2s are not
// coherent with one another), and with every look-up made of them the build of the 64^3 Q2 operator's dictionary
// (1.6e9 entries) took 205 ms; a workgroup-level LDS cache in front of them did not change that, nor did one look-up
// per distinct pair of a wavefront's 64 entries (570 ms: look-ups one after the other instead of side by side).
// counter[0] = pairs so far; beyond GP_MAX the build is abandoned (counter[1] = 1) and every wavefront leaves.
__device__ __forceinline__ int gdict_lookup(int off, unsigned long long vb, int *__restrict__ state,
                                            unsigned long long *rec, int *__restrict__ counter, GPair *__restrict__ gtab) {
    const unsigned long long hsh = (vb ^ (vb >> 29)) * 0x9E3779B97F4A7C15ull + (unsigned long long)(unsigned)off * 0xC2B2AE3D27D4EB4Full;
    unsigned slot = (unsigned)(hsh >> 40) & (GD_CAP - 1);
    for (int probes = 0; probes < GD_CAP; ++probes) {
        // (an operator without repeated pairs -- a general coarse operator -- can have every resident lane claim a slot
        // before the first of them sees the abandon flag: more claims than GD_CAP slots, and a look-up in a full table
        // would walk all of it.  The flag is looked at every 32 probes, and 1 024 probes without a hit abandon the build.)
        if ((probes & 31) == 31) {
            if (__hip_atomic_load(&counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;
            if (probes >= 1023) {
                __hip_atomic_store(&counter[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0;
            }
        }
        typedef unsigned long long gd_u2 __attribute__((ext_vector_type(2)));
        const gd_u2 r = *(const gd_u2 *)(rec + 2 * (size_t)slot);      // (ordinary, cached)
        unsigned long long w1 = r[1], w0 = r[0];
        if (w1 == 0) {      // empty, or not visible here yet
            int st = __hip_atomic_load(&state[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st == 0) {
                if (__hip_atomic_load(&counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;      // abandoned
                if (__hip_atomic_load(&counter[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= GP_MAX) {   // no room for a new pair: abandon before claiming
                    __hip_atomic_store(&counter[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return 0;
                }
                if (atomicCAS(&state[slot], 0, 1) == 0) {
                    const int id = atomicAdd(&counter[0], 1);
                    int code = id;
                    if (id >= GP_MAX) {
                        __hip_atomic_store(&counter[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        code = 0;
                    } else {
                        gtab[id] = GPair{off * 8, 0, __longlong_as_double((long long)vb)};      // (byte offset)
                    }
                    __hip_atomic_store(&rec[2 * (size_t)slot], vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&rec[2 * (size_t)slot + 1], ((unsigned long long)(unsigned)(code + 1) << 32) | (unsigned)off,
                                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    return code;
                }
            }
            // claimed by somebody else: the record again, with device-coherent loads; not there yet -> the same slot
            // again through the whole loop body (the claimer may be a lane of this wavefront: it publishes inside the
            // body, so waiting anywhere else would wait for a lane that is not running)
            w1 = __hip_atomic_load(&rec[2 * (size_t)slot + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (w1 == 0) {
                if (__hip_atomic_load(&counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;      // abandoned
                --probes;
                continue;
            }
            w0 = __hip_atomic_load(&rec[2 * (size_t)slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((int)(unsigned)(w1 & 0xffffffffull) == off && w0 == vb) return (int)(w1 >> 32) - 1;
        slot = (slot + 1) & (GD_CAP - 1);
    }
    return 0;
}
__global__ __launch_bounds__(256) void sell_gdict_kernel(int nslices, const roff_t *__restrict__ sptr,
                                                         const int *__restrict__ scol, const double *__restrict__ sval,
                                                         int *__restrict__ state, unsigned long long *rec,
                                                         int *__restrict__ counter, GPair *__restrict__ gtab,
                                                         unsigned long long *__restrict__ gcode) {
    const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (slice >= nslices) return;
    const roff_t beg = sptr[slice];
    const int w = (int)((sptr[slice + 1] - beg) >> 6);
    const int row = slice * 64 + lane;
    unsigned long long *wp = gcode + ((size_t)(beg >> 2) + (size_t)slice * 64 + lane);
    unsigned long long word = 0;      // four codes of the row per 8-byte word
    for (int k = 0; k < w; ++k) {
        const int off = scol[beg + 64 * k + lane] - row;
        const unsigned long long vb = (unsigned long long)__double_as_longlong(sval[beg + 64 * k + lane]);
        if ((k & 15) == 0 && __hip_atomic_load(&counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // abandoned (wave-uniform)
        const int code = gdict_lookup(off, vb, state, rec, counter, gtab);
        word |= (unsigned long long)(unsigned)(code & 0xffff) << (16 * (k & 3));
        if ((k & 3) == 3 || k + 1 == w) {
            wp[64 * (size_t)(k >> 2)] = word;      // (the rest of the last word stays 0)
            word = 0;
        }
    }
}

// 3 x 3 node blocks?  One wavefront per 63 rows (21 nodes: sell_gpair3_kernel's waves), one lane per row.  A row is
// REGULAR when its node's rows 3i, 3i+1, 3i+2 hold the same columns, in runs 3j, 3j+1, 3j+2; the others (rows eliminated
// for essential conditions, a last partial node) are listed in irr_rows (count[0] of them, in no particular order).
__global__ __launch_bounds__(256) void sell_bs3_kernel(int nrows, int nwaves, const roff_t *__restrict__ rowptr,
                                                       const int *__restrict__ col, int *__restrict__ irr_rows, int cap,
                                                       int *__restrict__ count) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    const int row = wave * 63 + lane;
    const int a = lane % 3, first = row - a;
    bool ok = lane < 63 && first + 2 < nrows;
    if (ok) {
        const roff_t rb = rowptr[row], fb = rowptr[first];
        const int len = (int)(rowptr[row + 1] - rb);
        ok = len % 3 == 0 && len == (int)(rowptr[first + 1] - fb);
        for (int k = 0; ok && k < len; ++k) {
            const int c = col[rb + k];
            ok = c % 3 == k % 3 && c == col[fb + k] && (k % 3 == 0 || c == col[rb + k - 1] + 1);
        }
    }
    const unsigned long long good = __ballot(ok);      // a node is regular only if its three rows are
    const bool node_ok = lane < 63 && ((good >> (lane - a)) & 7ull) == 7ull;
    if (lane < 63 && row < nrows && !node_ok) {
        const int at = atomicAdd(count, 1);
        if (at < cap) irr_rows[at] = row;
    }
}

// SpMV family on a dictionary-coded operator.  The table lives in LDS (at most GP_MAX pairs = 96 KB: one workgroup of
// 16 wavefronts per CU, persistent -- it copies the table once and then walks the slices of its XCD's contiguous eighth,
// interleaved with the other 31 workgroups of that XCD).  One wavefront per slice, one lane per row; per four entries one
// 8-byte code word, four LDS table reads and four gathers of x, eight entries in flight.  With the table in global
// memory the table look-ups were vector-memory gathers like those of x (scattered 16-byte entries: ~50 cycles each on
// the CU's address path) and the kernel was SLOWER than the plain slices it replaces although it moves a sixth of their
// bytes (3.3 against 3.2 ms per application of the 64^3 Q2 operator); from LDS they cost an LDS read.
// Same order of additions as the plain path (whole groups of four alternate between the two accumulators, the tail
// goes to the first): bit-identical results.
template <int MODE>
__global__ __launch_bounds__(1024) void sell_gpair_kernel(int nrows, int row0, int nslices, int per_xcd, int wg_per_xcd, int ng,
                                                          const roff_t *__restrict__ sptr,
                                                          const unsigned long long *__restrict__ gcode,
                                                          const GPair *__restrict__ gtab,
                                                          const double *__restrict__ x, double *__restrict__ y,
                                                          const double *__restrict__ b,
                                                          const double *__restrict__ dinv, double scale,
                                                          const double *__restrict__ xrow) {
    extern __shared__ __align__(16) GPair ltab[];
    for (int i = threadIdx.x; i < ng; i += 1024) ltab[i] = gtab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xcd = (int)(blockIdx.x & 7u), p = (int)(blockIdx.x >> 3);
    const int c0 = xcd * per_xcd, c1 = min(c0 + per_xcd, nslices);
    const int g0 = row0 >> 6;
    for (int slice = c0 + p * 16 + wv; slice < c1; slice += wg_per_xcd * 16) {
        const roff_t beg = sptr[slice], end = sptr[slice + 1];
        const int gslice = g0 + slice;
        const int w = (int)((end - beg) >> 6);
        const long row = (long)slice * 64 + lane;
        const bool live = row < nrows;
        const int grow = row0 + (int)row;
        double e_b = 0.0, e_d = 0.0, e_x = 0.0;
        if (MODE == MODE_RESIDUAL && live) e_b = b[row];
        if (MODE == MODE_ADD && live) e_x = y[row];
        if (MODE == MODE_SMOOTH && live) { e_b = b[row]; e_d = dinv[row]; e_x = xrow[row]; }
        const unsigned long long *wp = gcode + ((size_t)(beg >> 2) + (size_t)gslice * 64 + lane);
        const unsigned grow8 = (unsigned)grow << 3;
        double s0 = 0.0, s1 = 0.0;
        int k = 0;
        for (; k + 8 <= w; k += 8) {
            const unsigned long long ca = __builtin_nontemporal_load(wp + 16 * k), cb = __builtin_nontemporal_load(wp + 16 * (k + 4));
            GPair e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = ltab[(unsigned)(((j < 4 ? ca : cb) >> (16 * (j & 3))) & 0xffffull)];
            double xs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[j] = gp_x(x, grow8, e[j].off);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j & 1) s1 = fma(e[j].val, xs[j], s1);
                else s0 = fma(e[j].val, xs[j], s0);
            }
        }
        for (; k + 4 <= w; k += 4) {
            const unsigned long long ca = __builtin_nontemporal_load(wp + 16 * k);
            GPair e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = ltab[(unsigned)((ca >> (16 * j)) & 0xffffull)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double xv = gp_x(x, grow8, e[j].off);
                if (j & 1) s1 = fma(e[j].val, xv, s1);
                else s0 = fma(e[j].val, xv, s0);
            }
        }
        if (k < w) {
            unsigned long long ca = __builtin_nontemporal_load(wp + 16 * k);
            for (; k < w; ++k, ca >>= 16) {
                const GPair e = ltab[(unsigned)(ca & 0xffffull)];
                s0 = fma(e.val, gp_x(x, grow8, e.off), s0);
            }
        }
        const double sum = s0 + s1;
        if (live) {
            if (MODE == MODE_PLAIN) {
                y[row] = sum;
            } else if (MODE == MODE_RESIDUAL) {
                y[row] = e_b - sum;
            } else if (MODE == MODE_ADD) {
                y[row] = e_x + sum;
            } else {
                y[row] = e_x + scale * (e_d * (sum - e_b));
            }
        }
    }
}

// The same for operators with 3 x 3 NODE BLOCKS (vector problems numbered node by node, three components each: rows
// 3i..3i+2 store the same columns, in runs 3j..3j+2 -- sell_bs3_kernel).  The three lanes of a node would gather the
// same three entries of x one after the other; here each lane gathers ONE of them (entry 3s + a of its row, a = its
// component: x[3j + a]) and takes the other two from its neighbours with wave shifts (v_mov_b32_dpp wave_shr / wave_shl):
// a third of the gathers, which are what bounds the kernel above (~39 cycles of the CU's address path per instruction,
// whatever the number of active lanes: extra loads for a few lanes of a wave cost as much as for all of them).
// So that no node is cut, a wavefront takes 63 ROWS (21 nodes; lane 63 idles): rows 63 j .. 63 j + 62 of the operator,
// which lie in one or two storage slices -- slice, width and code address are per lane.  Irregular rows (eliminated for
// essential conditions: their lanes compute nothing useful here) are redone by sell_gpair3_fix_kernel afterwards.
// Same products in the same order per row: entries below w & ~3 (w = the row's slice width) alternate between the two
// accumulators, the rest goes to the first; entries past the row's slice (the wave's other slice is wider) add nothing.
__device__ __forceinline__ double wave_from_prev(double v) {      // lane l <- lane l - 1
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ double wave_from_next(double v) {      // lane l <- lane l + 1
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true));
}
// x of the three entries of one node block from the lane's own gather (its component a) and its neighbours'
__device__ __forceinline__ void bs3_share(double own, int a, double &x0, double &x1, double &x2) {
    const double m1 = wave_from_prev(own), m2 = wave_from_prev(m1), p1 = wave_from_next(own), p2 = wave_from_next(p1);
    x0 = a == 0 ? own : (a == 1 ? m1 : m2);
    x1 = a == 0 ? p1 : (a == 1 ? own : m1);
    x2 = a == 0 ? p2 : (a == 1 ? p1 : own);
}
// code of entry 3 t + a of a group of 12 (code words cw[0..2], four codes each; a = the lane's component, t constant)
__device__ __forceinline__ unsigned bs3_own_code(const unsigned long long (&cw)[3], int t, int a) {
    unsigned c[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) c[q] = (unsigned)((cw[(3 * t + q) >> 2] >> (16 * ((3 * t + q) & 3))) & 0xffffull);
    return a == 0 ? c[0] : (a == 1 ? c[1] : c[2]);
}
template <int MODE>
__device__ __forceinline__ void spmv_epilogue(double sum, double *__restrict__ y, const double *__restrict__ b,
                                              const double *__restrict__ dinv, double scale, const double *__restrict__ xrow, long row) {
    if (MODE == MODE_PLAIN) {
        y[row] = sum;
    } else if (MODE == MODE_RESIDUAL) {
        y[row] = b[row] - sum;
    } else if (MODE == MODE_ADD) {
        y[row] = y[row] + sum;
    } else {
        y[row] = xrow[row] + scale * (dinv[row] * (sum - b[row]));
    }
}
// Lean enough for TWO workgroups per CU (32 wavefronts): the table split into values (8 B) and byte offsets (4 B) -- 12 B
// per pair, 66 KB for 5 540 pairs -- and 63 VGPRs.  A version with one 16-byte table (one workgroup per CU) and a software
// pipeline over two register sets (code words of group g + 2 and the look-ups + gathers of group g + 1 issued before the
// products of group g; 95 VGPRs) ran at 1.22 ms per application of the 64^3 Q2 operator against 1.47 without the pipeline;
// its counters showed 71 % of the wave-cycles waiting at four wavefronts per SIMD, and this one, with eight and no
// pipeline, takes 1.05 ms (git history: sell_gpair3_kernel before "two workgroups per CU").
template <int MODE>
__global__ __launch_bounds__(1024, 2) void sell_gpair3_kernel(int nrows_all, int row_lo, int row_hi, int wave0, int nwaves, int per_xcd,
                                                               int wg_per_xcd, int ng, const roff_t *__restrict__ sptr,
                                                               const unsigned long long *__restrict__ gcode,
                                                               const GPair *__restrict__ gtab, const double *__restrict__ x,
                                                               double *__restrict__ y, const double *__restrict__ b,
                                                               const double *__restrict__ dinv, double scale) {
    extern __shared__ __align__(16) double lval[];      // [ng] values, then [ng] byte offsets
    int *loff = (int *)(lval + ng);
    for (int i = threadIdx.x; i < ng; i += 1024) { const GPair e = gtab[i]; lval[i] = e.val; loff[i] = e.off; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xcd = (int)(blockIdx.x & 7u), p = (int)(blockIdx.x >> 3);
    const int c0 = xcd * per_xcd, c1 = min(c0 + per_xcd, nwaves);
    const int a = lane % 3;
    for (int wi = c0 + p * 16 + wv; wi < c1; wi += wg_per_xcd * 16) {
        const int first = (wave0 + wi) * 63;
        const int row = min(first + min(lane, 62), nrows_all - 1);
        const bool live = lane < 63 && first + lane >= row_lo && first + lane < row_hi;
        const int slice = row >> 6;
        const roff_t beg = sptr[slice];
        const int w = (int)((sptr[slice + 1] - beg) >> 6), w4 = w & ~3;
        const unsigned long long *wp = gcode + ((size_t)(beg >> 2) + (size_t)slice * 64 + (row & 63));
        const unsigned short *cp = (const unsigned short *)wp;
        const unsigned row8 = (unsigned)row << 3;
        const int wa = __builtin_amdgcn_readfirstlane(w), wb = __builtin_amdgcn_readlane(w, 62);
        const int wmin = min(wa, wb), wmax = max(wa, wb);
        double s0 = 0.0, s1 = 0.0;
        int k = 0;
        for (; k + 12 <= wmin; k += 12) {
            unsigned long long cw[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) cw[q] = __builtin_nontemporal_load(wp + 16 * (k + 4 * q));
            double own[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) own[t] = gp_x(x, row8, loff[bs3_own_code(cw, t, a)]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double xs[3];
                bs3_share(own[t], a, xs[0], xs[1], xs[2]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int j = 3 * t + c;
                    const double v = lval[(unsigned)((cw[j >> 2] >> (16 * (j & 3))) & 0xffffull)];
                    if (j & 1) s1 = fma(v, xs[c], s1);
                    else s0 = fma(v, xs[c], s0);
                }
            }
        }
        for (; k < wmax; k += 3) {
            int o[3];
            double v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int kk = k + c;
                const unsigned code = kk < w ? cp[64 * (size_t)(kk - (kk & 3)) + (kk & 3)] : 0xffffu;
                o[c] = code != 0xffffu ? loff[code] : 0;
                v[c] = code != 0xffffu ? lval[code] : 0.0;
            }
            asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));
            const double own = gp_x(x, row8, a == 0 ? o[0] : (a == 1 ? o[1] : o[2]));
            double xs[3];
            bs3_share(own, a, xs[0], xs[1], xs[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (k + c < w4 && ((k + c) & 1)) s1 = fma(v[c], xs[c], s1);
                else s0 = fma(v[c], xs[c], s0);
            }
        }
        if (live) spmv_epilogue<MODE>(s0 + s1, y, b, dinv, scale, x, row);
    }
}
// The irregular rows of a node-block operator, one lane per row (a few rows per thousand: strided reads do not matter).
template <int MODE>
__global__ __launch_bounds__(256) void sell_gpair3_fix_kernel(int nirr, const int *__restrict__ irr_rows, int row_lo, int row_hi,
                                                              const roff_t *__restrict__ sptr,
                                                              const unsigned long long *__restrict__ gcode,
                                                              const GPair *__restrict__ gtab,
                                                              const double *__restrict__ x, double *__restrict__ y,
                                                              const double *__restrict__ b,
                                                              const double *__restrict__ dinv, double scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nirr) return;
    const int row = irr_rows[i];
    if (row < row_lo || row >= row_hi) return;
    const int slice = row >> 6;
    const roff_t beg = sptr[slice];
    const int w = (int)((sptr[slice + 1] - beg) >> 6), w4 = w & ~3;
    const unsigned short *cp = (const unsigned short *)(gcode + ((size_t)(beg >> 2) + (size_t)slice * 64 + (row & 63)));
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < w; ++k) {
        const GPair e = gtab[cp[64 * (size_t)(k - (k & 3)) + (k & 3)]];
        if (k < w4 && (k & 1)) s1 = fma(e.val, gp_x(x, (unsigned)row << 3, e.off), s1);
        else s0 = fma(e.val, gp_x(x, (unsigned)row << 3, e.off), s0);
    }
    spmv_epilogue<MODE>(s0 + s1, y, b, dinv, scale, x, row);
}

// Slice census of a SELL copy: cls[0..2] = pair-coded / offset-coded / plain slices, cls[3..5] = their stored entries
// (64 x width), cls[6] = code words of the coded slices, cls[7] = widest slice
__global__ __launch_bounds__(256) void sell_census_kernel(int nslices, const roff_t *__restrict__ sptr,
                                                          const int *__restrict__ ntab,
                                                          unsigned long long *__restrict__ cls) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const bool in = s < nslices;
    const int w = in ? (int)((sptr[s + 1] - sptr[s]) >> 6) : 0;
    const int nt = in ? ntab[s] : -1;
    const int c = nt >= 256 ? 0 : (nt >= 0 ? 1 : 2);
    // (one atomic per wavefront and counter: 265 000 threads adding to the same eight words took 6.5 ms)
    unsigned long long part[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        part[k] = in && c == k ? 1ull : 0ull;
        part[3 + k] = in && c == k ? (unsigned long long)w * 64ull : 0ull;
    }
    part[6] = in && c < 2 ? (unsigned long long)((w + 3) / 4) * 64ull : 0ull;
    part[7] = (unsigned long long)w;
    // (a slice that reads its tile's table has none of its own: counted as a negative share of the table bytes below)
    part[8] = in && nt >= 512 && (s & 3) != 0 ? 1ull : 0ull;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(part[k], o, 64);
            part[k] = k == 7 ? (other > part[k] ? other : part[k]) : part[k] + other;
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (part[k]) atomicAdd(cls + k, part[k]);
        atomicMax(cls + 7, part[7]);
        if (part[8]) atomicAdd(cls + 8, part[8]);
    }
}

__global__ __launch_bounds__(256) void sell_width_kernel(int nrows, const roff_t *__restrict__ rowptr,
                                                         int *__restrict__ width64) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    int len = (row < nrows) ? (int)(rowptr[row + 1] - rowptr[row]) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o, 64));
    const long slice = row >> 6;
    if ((threadIdx.x & 63) == 0 && slice * 64 < nrows) width64[slice] = len * 64;
}

// CSR -> SELL-64, one wavefront per slice.  The slice's CSR entries are one contiguous range: it is read
// with coalesced loads into LDS and every lane then picks its row's entries from there (the column-major
// SELL stores are coalesced by construction).  A lane walking its own CSR row instead touches 64 different
// cache lines per load instruction (measured: 110 GB fetched to convert 5.5 GB).  Slices with more than
// SF_CAP entries (dense coarse-level rows) take that slow walk.
constexpr int SF_CAP = 2048;
// the contiguous CSR range [r0, r0 + len) of a slice into LDS, one wavefront: eight entries per lane and trip are requested
// before the first is stored (one entry per trip waited for a global round trip 27 times per slice of a 27-point operator)
__device__ __forceinline__ void stage_slice(const int *__restrict__ col, const double *__restrict__ val, roff_t r0, int len, int lane,
                                            int *lc, double *lv) {
    constexpr int U = 8;
    for (int t0 = lane; t0 < len; t0 += 64 * U) {
        int c[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = min(t0 + 64 * u, len - 1);
            c[u] = col[r0 + t];
            v[u] = val[r0 + t];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (t0 + 64 * u < len) {
                lc[t0 + 64 * u] = c[u];
                lv[t0 + 64 * u] = v[u];
            }
    }
}
__global__ __launch_bounds__(64) void sell_fill_kernel(int nrows, const roff_t *__restrict__ rowptr,
                                                       const int *__restrict__ col,
                                                       const double *__restrict__ val,
                                                       const roff_t *__restrict__ sptr,
                                                       int *__restrict__ scol, double *__restrict__ sval) {
    __shared__ double lv[SF_CAP];
    __shared__ int lc[SF_CAP];
    const int slice = blockIdx.x, lane = threadIdx.x;
    const long row = (long)slice * 64 + lane;
    const roff_t beg = sptr[slice];
    const int w = (int)((sptr[slice + 1] - beg) >> 6);
    roff_t rb = 0, re = 0;
    if (row < nrows) { rb = rowptr[row]; re = rowptr[row + 1]; }
    const int pad = (row < nrows) ? (int)row : nrows - 1;
    const roff_t r0 = rowptr[(long)slice * 64], r1 = rowptr[min((long)slice * 64 + 64, (long)nrows)];
    if (r1 - r0 <= SF_CAP) {
        stage_slice(col, val, r0, (int)(r1 - r0), lane, lc, lv);
        __syncthreads();
        for (int k = 0; k < w; ++k) {
            const bool in = rb + k < re;
            scol[beg + 64 * k + lane] = in ? lc[rb - r0 + k] : pad;
            sval[beg + 64 * k + lane] = in ? lv[rb - r0 + k] : 0.0;
        }
        return;
    }
    for (int k = 0; k < w; ++k) {
        const bool in = rb + k < re;
        scol[beg + 64 * k + lane] = in ? col[rb + k] : pad;
        sval[beg + 64 * k + lane] = in ? val[rb + k] : 0.0;
    }
}

// Tables and byte codes of the coded slices (see sell_spmv_kernel): one wavefront per slice.  First
// the distinct (offset, value) PAIRS are collected in first-appearance order, one table entry per lane
// (ntab = 256 + count: neither columns nor values are streamed for such a slice); a slice with more than
// 64 pairs falls back to the distinct offsets col - row alone (ntab = count, values streamed), and one with
// more than 64 offsets keeps its 4-byte columns (ntab = -1).  The codes of four consecutive entries of a
// row share one 32-bit word at codes[sptr/4 + 64 slice + 64 (k / 4) + lane].
__global__ __launch_bounds__(256) void sell_code_kernel(int nslices, const roff_t *__restrict__ sptr,
                                                        const int *__restrict__ scol, const double *__restrict__ sval,
                                                        int *__restrict__ ntab, int *__restrict__ tab,
                                                        double *__restrict__ vtab, unsigned *__restrict__ codes,
                                                        int with_values, int share_tables) {
    // (tile = the four slices of this block; merge phase below)
    __shared__ int m_off[4][64];
    __shared__ long long m_val[4][64];
    __shared__ int m_np[4];                 // pairs of the slice's own table, -1: the slice is not (short) pair-coded
    __shared__ unsigned char m_map[4][64];  // own code -> code in the tile's table
    __shared__ int m_ok, m_nt;
    const int wv = threadIdx.x >> 6;
    const int slice = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    const bool have = slice < nslices;
    const roff_t beg = have ? sptr[slice] : 0;
    const int w = have ? (int)((sptr[slice + 1] - beg) >> 6) : 0;
    const int row = slice * 64 + lane;
    unsigned *wp = codes + ((size_t)(beg >> 2) + (size_t)slice * 64 + lane);
    int my_np = -1;
    for (int pass = with_values ? 0 : 1; pass < 2 && have; ++pass) {
        const bool pairs = pass == 0;
        int mytab = 0, nt = 0;
        long long myval = 0;
        unsigned cw = 0;
        bool ok = true;
        for (int k = 0; k < w && ok; ++k) {
            const int delta = scol[beg + 64 * k + lane] - row;
            const long long vb = pairs ? __double_as_longlong(sval[beg + 64 * k + lane]) : 0;
            // One trip per DISTINCT pair among the 64 entries of this column of the slice (one to three on a stencil operator):
            // the first lane without a code broadcasts its pair, the table -- entry t in lane t -- answers with one ballot, and a
            // pair that is not in it is appended.  (Until round 4 every lane first walked the whole table through two shuffles
            // per entry -- 27 x 27 x 3 cross-lane reads per slice of a 27-point operator, the kernel's bound -- before this loop
            // added what was missing.  Same tables, same codes: the pairs are appended in the order of the lanes that bring them.)
            int code = -1;
            unsigned long long pending = ~0ull;
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                // (both shuffles outside any condition: a cross-lane read under a divergent branch returns nothing from
                // the lanes that did not take it)
                const int d = __shfl(delta, leader);
                const long long dv = __shfl(vb, leader);
                const unsigned long long hit = __ballot(lane < nt && mytab == d && myval == dv);
                int c;
                if (hit) c = __ffsll((long long)hit) - 1;
                else {
                    if (nt == 64) { ok = false; break; }
                    if (lane == nt) { mytab = d; myval = dv; }
                    c = nt++;
                }
                if (code < 0 && delta == d && vb == dv) code = c;
                pending = __ballot(code < 0);
            }
            cw |= (unsigned)(code & 255) << (8 * (k & 3));
            if ((k & 3) == 3 || k == w - 1) {
                wp[16 * (k & ~3)] = cw;
                cw = 0;
            }
        }
        if (ok || !pairs) {
            tab[(size_t)slice * 64 + lane] = mytab;
            if (pairs) vtab[(size_t)slice * 64 + lane] = __longlong_as_double(myval);
            if (lane == 0) ntab[slice] = ok ? (pairs ? 256 + nt : nt) : -1;
            if (ok && pairs && w <= 32) {
                my_np = nt;
                m_off[wv][lane] = mytab;
                m_val[wv][lane] = myval;
            }
            break;
        }
    }
    // ---- one table for the tile (share_tables): the four slice tables of a stencil operator hold the same pairs in
    // different orders (12 bytes per row of table traffic, a third of the matrix stream); merged into one of at most
    // 64 pairs the tile's four wavefronts read the same 768 bytes.  ntab = 512 + pairs marks "table at the tile's
    // first slice".  Lossless: every code still names the same (offset, value) pair.
    if (!share_tables) return;
    if (lane == 0) m_np[wv] = my_np;
    if (threadIdx.x == 0) m_ok = 1;
    __syncthreads();
    if (m_np[0] < 0 || m_np[1] < 0 || m_np[2] < 0 || m_np[3] < 0) return;      // (uniform: a partial or mixed tile keeps its own tables)
    if (wv == 0) {
        int t_off = m_off[0][lane];
        long long t_val = m_val[0][lane];
        int nT = m_np[0];
        m_map[0][lane] = (unsigned char)lane;
        bool ok = true;
        for (int q = 1; q < 4 && ok; ++q)
            for (int e = 0; e < m_np[q]; ++e) {
                const int d = m_off[q][e];
                const long long dv = m_val[q][e];
                const unsigned long long hit = __ballot(lane < nT && t_off == d && t_val == dv);
                int code;
                if (hit) code = __ffsll((long long)hit) - 1;
                else {
                    if (nT == 64) { ok = false; break; }
                    if (lane == nT) { t_off = d; t_val = dv; }
                    code = nT++;
                }
                if (lane == 0) m_map[q][e] = (unsigned char)code;
            }
        if (lane == 0) { m_ok = ok ? 1 : 0; m_nt = nT; }
        if (ok) {
            m_off[0][lane] = t_off;
            m_val[0][lane] = t_val;
        }
    }
    __syncthreads();
    if (!m_ok) return;
    // every wavefront rewrites ITS OWN code words (written above by the same lanes) through the map
    for (int q4 = 0; 4 * q4 < w; ++q4) {
        const unsigned cw = wp[64 * q4];
        unsigned out = 0;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4)
            if (4 * q4 + e4 < w) out |= (unsigned)m_map[wv][(cw >> (8 * e4)) & 63u] << (8 * e4);
        wp[64 * q4] = out;
    }
    const int first = blockIdx.x * 4;
    if (wv == 0) {
        tab[(size_t)first * 64 + lane] = lane < m_nt ? m_off[0][lane] : 0;
        vtab[(size_t)first * 64 + lane] = lane < m_nt ? __longlong_as_double(m_val[0][lane]) : 0.0;
    }
    if (lane == 0) ntab[slice] = 512 + m_nt;
}

__global__ __launch_bounds__(256) void widen_offsets_kernel(long n, const int *__restrict__ in, roff_t *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ __launch_bounds__(256) void narrow_offsets_kernel(long n, const roff_t *__restrict__ in, int *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int)in[i];
}

void import_rowptr(DBuf<roff_t> &dst, const void *src, int bits, size_t n, hipStream_t s) {
    SA_REQUIRE(bits == 32 || bits == 64, "row offsets must be 32 or 64 bits wide");
    if (bits == 64) {
        import_array(dst, (const roff_t *)src, n, s);
        return;
    }
    DBuf<int> narrow;
    import_array(narrow, (const int *)src, n, s);
    dst.alloc(n);
    if (n) hipLaunchKernelGGL(widen_offsets_kernel, dim3(div_up((long)n, 256)), dim3(256), 0, s, (long)n, narrow.p, dst.p);
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));      // `narrow` may be an upload that is freed on return
}

void export_rowptr32(int *dst_host, const DBuf<roff_t> &src, size_t n, hipStream_t s) {
    if (!n) return;
    roff_t last = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&last, src.p + (n - 1), sizeof(roff_t), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_REQUIRE(last < ((roff_t)1 << 31), "operator has more than 2^31 entries: use the 64-bit getter");
    DBuf<int> narrow(n);
    hipLaunchKernelGGL(narrow_offsets_kernel, dim3(div_up((long)n, 256)), dim3(256), 0, s, (long)n, src.p, narrow.p);
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipMemcpyAsync(dst_host, narrow.p, sizeof(int) * n, hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
}

void build_sell(hipStream_t s, DCsr &A) {
    A.has_sell = false;
    A.sell_dcode.release();      // (codes of the smoother's diagonal belong to the operator they were made for: build_dinv_codes)
    A.sell_dsrc = nullptr;
    if (A.nrows == 0) return;
    A.nslices = div_up(A.nrows, 64);
    DBuf<int> w64((size_t)A.nslices);
    const int grid = div_up((long)A.nslices * 64, 256);
    hipLaunchKernelGGL(sell_width_kernel, dim3(grid), dim3(256), 0, s, A.nrows, A.rowptr.p, w64.p);
    A.sell_ptr.alloc((size_t)A.nslices + 1);
    exclusive_scan_off(s, A.nslices, w64.p, A.sell_ptr.p);
    roff_t total = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&total, A.sell_ptr.p + A.nslices, sizeof(roff_t), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    A.sell_size = total;
    A.sell_col.alloc((size_t)total + 64);
    A.sell_val.alloc((size_t)total + 64);
    hipLaunchKernelGGL(sell_fill_kernel, dim3(A.nslices), dim3(64), 0, s, A.nrows, A.rowptr.p, A.col.p,
                       A.val.p, A.sell_ptr.p, A.sell_col.p, A.sell_val.p);
    SA_HIP_CHECK(hipGetLastError());
    // byte codes for the slices with few distinct column offsets (saamge_amd_options.sell bit 0 cleared: none)
    const bool no_codes = !(options().sell & 1);
    A.sell_ntab.alloc((size_t)A.nslices);
    A.sell_tab.alloc((size_t)A.nslices * 64);
    A.sell_code.alloc((size_t)total / 4 + (size_t)A.nslices * 64 + 64);
    A.sell_vtab.alloc((size_t)A.nslices * 64);
    // (bit 1 cleared: offset codes only, values always streamed)
    const bool no_vals = !(options().sell & 2);
    constexpr bool no_share = false;
    if (no_codes)
        SA_HIP_CHECK(hipMemsetAsync(A.sell_ntab.p, 0xff, sizeof(int) * (size_t)A.nslices, s));
    else
        hipLaunchKernelGGL(sell_code_kernel, dim3(div_up(A.nslices, 4)), dim3(256), 0, s, A.nslices, A.sell_ptr.p,
                           A.sell_col.p, A.sell_val.p, A.sell_ntab.p, A.sell_tab.p, A.sell_vtab.p, A.sell_code.p,
                           no_vals ? 0 : 1, no_share ? 0 : 1);
    SA_HIP_CHECK(hipGetLastError());
    // what the copy holds, per slice format: the bytes one application has to move (the roofline of the SpMV family
    // prices THESE, bench.py) and whether the all-pair-coded fast path applies
    DBuf<unsigned long long> cls(9);
    SA_HIP_CHECK(hipMemsetAsync(cls.p, 0, 9 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(sell_census_kernel, dim3(div_up(A.nslices, 256)), dim3(256), 0, s, A.nslices, A.sell_ptr.p,
                       A.sell_ntab.p, cls.p);
    SA_HIP_CHECK(hipGetLastError());
    unsigned long long h[9];
    SA_HIP_CHECK(hipMemcpyAsync(h, cls.p, sizeof(h), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    for (int c = 0; c < 3; ++c) { A.sell_class_slices[c] = (int64_t)h[c]; A.sell_class_entries[c] = (int64_t)h[3 + c]; }
    // codes 4 B per word, tables 4 (+8) B per lane of a coded slice, values / columns of the formats that stream them,
    // 8 B slice offset + 4 B table size per slice
    A.sell_stream_bytes = 4.0 * (double)h[6] + 12.0 * 64.0 * (double)(h[0] - h[8]) + 4.0 * 64.0 * (double)h[1] +
                          8.0 * (double)h[4] + 12.0 * (double)h[5] + 12.0 * (double)A.nslices;
    // operator-level pair dictionary for operators that are all plain slices (bit 3 cleared: never)
    const bool no_gpair = !(options().sell & 8);
    A.sell_gpair = false;
    if (!no_gpair && !no_codes && h[0] == 0 && h[1] == 0 && A.nnz >= (1 << 22) && A.ncols < (1 << 28)) {
        DBuf<int> st((size_t)GD_CAP), ctr(2);
        DBuf<unsigned long long> rec(2 * (size_t)GD_CAP);
        SA_HIP_CHECK(hipMemsetAsync(st.p, 0, sizeof(int) * (size_t)GD_CAP, s));
        SA_HIP_CHECK(hipMemsetAsync(rec.p, 0, 16 * (size_t)GD_CAP, s));
        SA_HIP_CHECK(hipMemsetAsync(ctr.p, 0, 2 * sizeof(int), s));
        A.sell_gcode.alloc((size_t)total / 4 + (size_t)A.nslices * 64 + 64);
        A.sell_gtab.alloc(GP_MAX);
        hipLaunchKernelGGL(sell_gdict_kernel, dim3(div_up(A.nslices, 4)), dim3(256), 0, s, A.nslices, A.sell_ptr.p, A.sell_col.p,
                           A.sell_val.p, st.p, rec.p, ctr.p, (GPair *)A.sell_gtab.p, A.sell_gcode.p);
        SA_HIP_CHECK(hipGetLastError());
        int hc[2];
        SA_HIP_CHECK(hipMemcpyAsync(hc, ctr.p, sizeof(hc), hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
        if (!hc[1] && hc[0] <= GP_MAX) {      // (the table has to fit LDS: see sell_gpair_kernel)
            A.sell_gpair = true;
            A.sell_ng = hc[0];
            // codes 2 B per stored entry (rounded up to four per row), the table, 8 B per slice
            const double words = (double)total / 4.0 + 64.0 * (double)A.nslices;           // upper bound of the 8-byte code words
            A.sell_stream_bytes = 8.0 * words + 16.0 * (double)hc[0] + 8.0 * (double)A.nslices;
            A.sell_col.release();      // the dictionary replaces the streamed columns and values (12 B per stored entry)
            A.sell_val.release();
            const bool no_bs3 = !(options().sell & 16);
            A.sell_bs3 = false;
            if (!no_bs3 && A.ncols == A.nrows && A.nrows >= 63) {
                const int nwaves = div_up(A.nrows, 63), cap = A.nrows / 16 + 1;      // at most a sixteenth of the rows on their own
                A.sell_irr.alloc((size_t)cap);
                SA_HIP_CHECK(hipMemsetAsync(ctr.p, 0, sizeof(int), s));
                hipLaunchKernelGGL(sell_bs3_kernel, dim3(div_up(nwaves, 4)), dim3(256), 0, s, A.nrows, nwaves, A.rowptr.p, A.col.p,
                                   A.sell_irr.p, cap, ctr.p);
                SA_HIP_CHECK(hipGetLastError());
                SA_HIP_CHECK(hipMemcpyAsync(&A.sell_nirr, ctr.p, sizeof(int), hipMemcpyDeviceToHost, s));
                SA_HIP_CHECK(hipStreamSynchronize(s));
                A.sell_bs3 = A.sell_nirr <= cap;
                if (!A.sell_bs3) A.sell_irr.release();
            }
        } else {
            A.sell_gcode.release();
            A.sell_gtab.release();
        }
        if ((options().debug & 2))
            std::fprintf(stderr, "build_sell: operator-level pair dictionary: %d pairs%s%s\n", hc[0], A.sell_gpair ? "" : " (abandoned)",
                         A.sell_gpair && A.sell_bs3 ? ", 3 x 3 node blocks" : "");
        if ((options().debug & 2) && A.sell_gpair)
            std::fprintf(stderr, "build_sell: %d of %d rows outside regular node blocks\n", A.sell_nirr, A.nrows);
    }
    const bool no_fast = !(options().sell & 4);
    A.sell_fast_ok = !no_fast && A.ncols < (1 << 29);      // (32-bit byte offsets into x on the short-chain path)
    // x-staging plan of the pair-coded tiles
    constexpr bool no_stage = false;
    A.sell_stage_cap = 0;
    int staged_tiles = 0;
    if (A.sell_fast_ok && !no_stage && h[0] * 2 >= (unsigned long long)A.nslices) {      // (worth a plan: most slices pair-coded)
        const int ntiles = div_up(A.nslices, 4);
        A.sell_tile_nseg.alloc((size_t)ntiles);
        A.sell_tile_seg.alloc((size_t)ntiles * SELL_SEG_MAX);
        A.sell_unstaged.alloc((size_t)ntiles);
        DBuf<int> mx(4);
        SA_HIP_CHECK(hipMemsetAsync(mx.p, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(sell_stage_kernel, dim3(ntiles), dim3(64), 0, s, ntiles, A.nslices, A.nrows, A.sell_ptr.p, A.sell_ntab.p,
                           A.sell_tab.p, A.sell_tile_nseg.p, A.sell_tile_seg.p, mx.p, A.sell_unstaged.p);
        SA_HIP_CHECK(hipGetLastError());
        int hm[4];
        SA_HIP_CHECK(hipMemcpyAsync(hm, mx.p, sizeof(hm), hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
        A.sell_stage_cap = hm[0];
        staged_tiles = hm[1];
        A.sell_nunstaged = hm[2];
        if (A.sell_nunstaged * 4 > ntiles) A.sell_stage_cap = 0;      // too few tiles staged to be worth two launches
        A.sell_one_table = hm[3] == 0;
        A.sell_stream_bytes += (4.0 + 8.0 * SELL_SEG_MAX) * ntiles;
        // the regular second copy of the staged tiles' code words + one descriptor word per tile (sell_staged2_kernel)
        A.sell_wq = 0;
        if (A.sell_stage_cap > 0 && A.sell_one_table) {
            A.sell_wq = (int)std::min<unsigned long long>(8, (h[7] + 3) / 4);      // (h[7]: the widest slice; staged ones are <= 32)
            A.sell_codeR.alloc((size_t)ntiles * A.sell_wq * 256);
            A.sell_tile_desc.alloc((size_t)ntiles);
            hipLaunchKernelGGL(sell_regular_codes_kernel, dim3(ntiles), dim3(256), 0, s, ntiles, A.sell_wq, A.ncols, A.sell_ptr.p,
                               A.sell_tile_nseg.p, A.sell_tile_seg.p, A.sell_code.p, A.sell_codeR.p, A.sell_tile_desc.p);
            SA_HIP_CHECK(hipGetLastError());
        }
    }
    if ((options().debug & 2))
        std::fprintf(stderr, "build_sell: staging plan: %d of %d tiles, largest %d doubles\n", staged_tiles, div_up(A.nslices, 4), A.sell_stage_cap);
    if ((options().debug & 2))
        std::fprintf(stderr, "build_sell: %d rows, slices pair/offset/plain %lld/%lld/%lld, widest %llu, stream bytes %.0f, fast path %d\n",
                     A.nrows, (long long)h[0], (long long)h[1], (long long)h[2], h[7], A.sell_stream_bytes, (int)A.sell_fast_ok);
    A.has_sell = true;
}

// Rows [rr.row0, rr.row0 + rr.nrows) of A (row0 a multiple of 64 so that SELL slices line up);
// x is indexed by GLOBAL column, the row-indexed arrays y, b, dinv are global-length too.
template <int MODE>
static void launch_spmv(hipStream_t s, const DCsr &A, RowRange rr, const double *x, double *y,
                        const double *b, const double *dinv, double scale) {
    const int row0 = rr.nrows < 0 ? 0 : rr.row0;
    const int nrows = rr.nrows < 0 ? A.nrows : rr.nrows;
    if (nrows == 0) return;
    SA_REQUIRE(row0 % 64 == 0 && row0 + nrows <= A.nrows, "bad row range");
    y += row0;
    if (b) b += row0;
    const unsigned char *dcode = (dinv && A.sell_dcode.n == (size_t)A.nrows && dinv == A.sell_dsrc) ? A.sell_dcode.p + row0 : nullptr;
    if (dinv) dinv += row0;
    const double *xrow = x + row0;
    // (y += A x with irregular rows: the wave kernel's result for them would be added before the fix kernel adds the right
    // one -- the plain dictionary kernel takes that case)
    if (A.has_sell && A.sell_gpair && A.sell_bs3 && !(MODE == MODE_ADD && A.sell_nirr > 0)) {
        // waves of 63 rows, numbered over the whole operator; the kernels take whole-operator arrays
        const int wave0 = row0 / 63, nw = (row0 + nrows - 1) / 63 - wave0 + 1;
        const int per_xcd = div_up(div_up(nw, 8), 16) * 16;
        const int wgl = std::min(64, div_up(per_xcd, 16));      // (two workgroups per CU)
        auto kern = sell_gpair3_kernel<MODE>;
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 12 * GP_MAX + 16));
        hipLaunchKernelGGL(kern, dim3(wgl * 8), dim3(1024), 12 * (size_t)A.sell_ng + 16, s, A.nrows, row0, row0 + nrows, wave0, nw, per_xcd, wgl, A.sell_ng,
                           A.sell_ptr.p, A.sell_gcode.p, (const GPair *)A.sell_gtab.p, x, y - row0, b ? b - row0 : nullptr,
                           dinv ? dinv - row0 : nullptr, scale);
        SA_HIP_CHECK(hipGetLastError());
        if (A.sell_nirr) {
            hipLaunchKernelGGL(sell_gpair3_fix_kernel<MODE>, dim3(div_up(A.sell_nirr, 256)), dim3(256), 0, s, A.sell_nirr, A.sell_irr.p,
                               row0, row0 + nrows, A.sell_ptr.p, A.sell_gcode.p, (const GPair *)A.sell_gtab.p, x, y - row0,
                               b ? b - row0 : nullptr, dinv ? dinv - row0 : nullptr, scale);
            SA_HIP_CHECK(hipGetLastError());
        }
        return;
    }
    if (A.has_sell && A.sell_gpair) {
        const int nsl = div_up(nrows, 64);
        const int per_xcd = div_up(div_up(nsl, 8), 16) * 16;
        const int wg_per_xcd = std::min(32, div_up(per_xcd, 16));
        const size_t lds = sizeof(GPair) * (size_t)A.sell_ng;
        auto kern = sell_gpair_kernel<MODE>;
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(GPair) * GP_MAX)));
        hipLaunchKernelGGL(kern, dim3(wg_per_xcd * 8), dim3(1024), lds, s, nrows, row0, nsl, per_xcd, wg_per_xcd, A.sell_ng,
                           A.sell_ptr.p + row0 / 64, A.sell_gcode.p, (const GPair *)A.sell_gtab.p, x, y, b, dinv, scale, xrow);
        SA_HIP_CHECK(hipGetLastError());
        return;
    }
    // (tiles are global: a row range takes the staged kernel when it starts on a tile and ends on one or with the operator)
    if (A.has_sell && A.sell_stage_cap > 0 && row0 % 256 == 0 && (nrows % 256 == 0 || row0 + nrows == A.nrows)) {
        const int nblocks = div_up((long)div_up(nrows, 64) * 64, 256);
        constexpr bool no_xcd = false;
        const int per_xcd = no_xcd ? 0 : div_up(nblocks, 8);
        const size_t lds_bytes = 8 * (size_t)A.sell_stage_cap + (A.sell_one_table ? 1 : 4) * 64 * sizeof(PairEntry);
        // (up to eight tiles that are not staged ride at the end of the staged kernel's grid instead of a launch of their own)
        const int grid_main = no_xcd ? nblocks : per_xcd * 8;
        const bool fold = A.sell_wq > 0 && A.sell_nunstaged > 0 && A.sell_nunstaged <= 8 && 8 * (size_t)A.sell_stage_cap >= 4 * 64 * sizeof(PairEntry);
        if (A.sell_wq > 0) {
            const SellLeftover left{grid_main, fold ? A.sell_nunstaged : 0, A.sell_unstaged.p, A.sell_ptr.p + row0 / 64, A.sell_col.p, A.sell_val.p,
                                    A.sell_ntab.p, A.sell_code.p};
            hipLaunchKernelGGL((sell_staged2_kernel<MODE>), dim3(grid_main + left.n), dim3(256), lds_bytes, s, left, nrows, row0,
                               nblocks, per_xcd, A.sell_stage_cap, A.ncols, A.sell_wq, A.sell_codeR.p, A.sell_tab.p, A.sell_vtab.p,
                               A.sell_tile_desc.p, A.sell_tile_seg.p, x, y, b, dinv, scale, xrow, dcode, (const double *)A.sell_dtab.p);
        } else
        hipLaunchKernelGGL((sell_staged_kernel<MODE>), dim3(no_xcd ? nblocks : per_xcd * 8), dim3(256), lds_bytes, s, nrows, row0,
                           nblocks, per_xcd, A.sell_stage_cap, A.ncols, (int)A.sell_one_table, A.sell_ptr.p + row0 / 64, A.sell_ntab.p, A.sell_tab.p,
                           A.sell_code.p, x, y, b, dinv, scale, xrow, A.sell_vtab.p, A.sell_tile_nseg.p, A.sell_tile_seg.p);
        if (A.sell_nunstaged > 0 && !fold)
            hipLaunchKernelGGL((sell_tiles_kernel<MODE>), dim3(A.sell_nunstaged), dim3(256), 0, s, A.sell_nunstaged, A.sell_unstaged.p,
                               nrows, row0, A.sell_ptr.p + row0 / 64, A.sell_col.p, A.sell_val.p, A.sell_ntab.p, A.sell_tab.p,
                               A.sell_code.p, x, y, b, dinv, scale, xrow, A.sell_vtab.p);
        SA_HIP_CHECK(hipGetLastError());
        return;
    }
    if (A.has_sell) {
        const int nblocks = div_up((long)div_up(nrows, 64) * 64, 256);
        constexpr bool no_xcd = false;
        const int per_xcd = no_xcd ? 0 : div_up(nblocks, 8);     // (0: blocks in launch order)
        hipLaunchKernelGGL((sell_spmv_kernel<MODE>), dim3(no_xcd ? nblocks : per_xcd * 8), dim3(256), 0, s, nrows, row0, nblocks, per_xcd,
                           (int)A.sell_fast_ok, A.sell_ptr.p + row0 / 64, A.sell_col.p, A.sell_val.p, A.sell_ntab.p,
                           A.sell_tab.p, A.sell_code.p, x, y, b, dinv, scale, xrow, A.sell_vtab.p);
        SA_HIP_CHECK(hipGetLastError());
        return;
    }
    const int L = A.lanes_per_row;
    const long threads = (long)nrows * L;
    const int grid = div_up(threads, 256);
#define SA_CASE(LL)                                                                              \
    case LL:                                                                                     \
        hipLaunchKernelGGL((spmv_kernel<LL, MODE>), dim3(grid), dim3(256), 0, s, nrows,          \
                           A.rowptr.p + row0, A.col.p, A.val.p, x, y, b, dinv, scale, xrow);     \
        break;
    switch (L) {
        SA_CASE(1) SA_CASE(2) SA_CASE(4) SA_CASE(8) SA_CASE(16) SA_CASE(32) SA_CASE(64)
        default: SA_REQUIRE(false, "bad lanes_per_row");
    }
#undef SA_CASE
    SA_HIP_CHECK(hipGetLastError());
}

static inline double spmv_bytes(const DCsr &A, RowRange rr) {
    const double frac = (rr.nrows < 0 || A.nrows == 0) ? 1.0 : (double)rr.nrows / A.nrows;
    return (12.0 * A.nnz + 20.0 * A.nrows) * frac;
}
static inline double spmv_rows(const DCsr &A, RowRange rr) { return rr.nrows < 0 ? A.nrows : rr.nrows; }
// bytes of matrix data in the format the kernels run (build_sell's census) + x read and y written once
static inline double spmv_fmt_bytes(const DCsr &A, RowRange rr) {
    if (!A.has_sell) return 0.0;
    const double frac = (rr.nrows < 0 || A.nrows == 0) ? 1.0 : (double)rr.nrows / A.nrows;
    return (A.sell_stream_bytes + 16.0 * A.nrows) * frac;
}
// the SpMV family is listed per operator size: "<name>@<rows>" (the levels of a hierarchy have very different formats)
static inline std::string spmv_label(const char *name, const DCsr &A) { return std::string(name) + "@" + std::to_string(A.nrows); }
static inline double spmv_flops(const DCsr &A, RowRange rr) {
    return 2.0 * A.nnz * ((rr.nrows < 0 || A.nrows == 0) ? 1.0 : (double)rr.nrows / A.nrows);
}

void spmv(hipStream_t s, const DCsr &A, const double *x, double *y, RowRange rr) {
    profiler().begin(s);
    launch_spmv<MODE_PLAIN>(s, A, rr, x, y, nullptr, nullptr, 0.0);
    profiler().end(s, spmv_label("spmv", A).c_str(), spmv_bytes(A, rr), spmv_flops(A, rr), spmv_fmt_bytes(A, rr));
}
void spmv_residual(hipStream_t s, const DCsr &A, const double *x, const double *b, double *r, RowRange rr) {
    profiler().begin(s);
    launch_spmv<MODE_RESIDUAL>(s, A, rr, x, r, b, nullptr, 0.0);
    profiler().end(s, spmv_label("spmv_residual", A).c_str(), spmv_bytes(A, rr) + 8.0 * spmv_rows(A, rr), spmv_flops(A, rr),
                   spmv_fmt_bytes(A, rr) + (A.has_sell ? 8.0 * spmv_rows(A, rr) : 0.0));
}
void spmv_add(hipStream_t s, const DCsr &P, const double *xc, double *x, RowRange rr) {
    profiler().begin(s);
    launch_spmv<MODE_ADD>(s, P, rr, xc, x, nullptr, nullptr, 0.0);
    profiler().end(s, spmv_label("spmv_add", P).c_str(), spmv_bytes(P, rr) + 8.0 * spmv_rows(P, rr), spmv_flops(P, rr),
                   spmv_fmt_bytes(P, rr) + (P.has_sell ? 8.0 * spmv_rows(P, rr) : 0.0));
}
void smooth_step(hipStream_t s, const DCsr &A, const double *dinv_neg, const double *b,
                 const double *xin, double *xout, double scale, RowRange rr) {
    profiler().begin(s);
    launch_spmv<MODE_SMOOTH>(s, A, rr, xin, xout, b, dinv_neg, scale);
    // (b, and D^-1 as it is read: 8 bytes per row, or its byte code where the operator carries one -- build_dinv_codes)
    const bool coded_d = A.sell_dcode.n == (size_t)A.nrows && dinv_neg == A.sell_dsrc && A.sell_wq > 0 && A.sell_stage_cap > 0;
    profiler().end(s, spmv_label("smooth_step", A).c_str(), spmv_bytes(A, rr) + 24.0 * spmv_rows(A, rr), spmv_flops(A, rr),
                   spmv_fmt_bytes(A, rr) + (A.has_sell ? (coded_d ? 9.0 : 16.0) * spmv_rows(A, rr) : 0.0));
}

__global__ __launch_bounds__(256) void smooth_first_kernel(int n, const double *__restrict__ dinv,
                                                           const double *__restrict__ b,
                                                           double *__restrict__ xout, double scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) xout[i] = scale * (dinv[i] * (0.0 - b[i]));
}
void smooth_first(hipStream_t s, int n, const double *dinv_neg, const double *b, double *xout,
                  double scale) {
    if (!n) return;
    profiler().begin(s);
    hipLaunchKernelGGL(smooth_first_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, dinv_neg, b,
                       xout, scale);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "smooth_first", 24.0 * n, 2.0 * n);
}

// ---- weighted-l1 smoother diagonal ----------------------------------------------------
// (8 lanes per row: a lane walking its own row alone touches one cache line per lane and load)
__global__ __launch_bounds__(256) void sqrt_abs_diag_kernel(int nrows, const roff_t *__restrict__ rowptr,
                                                            const int *__restrict__ col,
                                                            const double *__restrict__ val,
                                                            double *__restrict__ sd) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 7;
    const long row = gtid >> 3;
    if (row >= nrows) return;
    double d = 0.0;
    for (roff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 8)
        if (col[k] == row) d += val[k];
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) d += __shfl_down(d, o, 8);
    if (lane == 0) sd[row] = sqrt(fabs(d));
}

template <int L>
__global__ __launch_bounds__(256) void dinv_neg_kernel(int nrows, const roff_t *__restrict__ rowptr,
                                                       const int *__restrict__ col,
                                                       const double *__restrict__ val,
                                                       const double *__restrict__ sd,
                                                       double *__restrict__ out) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & (L - 1);
    const long row = gtid / L;
    if (row >= nrows) return;
    double sum = 0.0;
    for (roff_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += L) sum += fabs(val[k]) / sd[col[k]];
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, L);
    if (lane == 0) out[row] = -1.0 / (sd[row] * sum);
}

void build_dinv_neg(hipStream_t s, const DCsr &A, double *sd, double *out) {
    if (!A.nrows) return;
    hipLaunchKernelGGL(sqrt_abs_diag_kernel, dim3(div_up((long)A.nrows * 8, 256)), dim3(256), 0, s, A.nrows,
                       A.rowptr.p, A.col.p, A.val.p, sd);
    const int grid = div_up((long)A.nrows * 8, 256);
    hipLaunchKernelGGL((dinv_neg_kernel<8>), dim3(grid), dim3(256), 0, s, A.nrows, A.rowptr.p,
                       A.col.p, A.val.p, sd, out);
    SA_HIP_CHECK(hipGetLastError());
}

// The smoother's diagonal factor as byte codes (DCsr::sell_dcode): every value goes into an open-addressing table of 256
// slots (key = its bits); more than 256 distinct values (variable coefficients) -- no codes.  The code of a row is the
// slot of its value, the table is read as doubles.
constexpr unsigned long long DINV_EMPTY = 0x7FF8DEADBEEF0001ull;      // (a NaN payload no diagonal produces)
__device__ inline unsigned dinv_slot0(unsigned long long bits) {
    bits ^= bits >> 33; bits *= 0xFF51AFD7ED558CCDull; bits ^= bits >> 29;
    return (unsigned)bits & 255u;
}
__global__ __launch_bounds__(256) void dinv_tab_kernel(int n, const double *__restrict__ dinv, unsigned long long *__restrict__ tab,
                                                       int *__restrict__ overflow) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || *overflow) return;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(dinv[i]);
    unsigned h = dinv_slot0(bits);
    for (int probe = 0; probe < 256; ++probe, h = (h + 1) & 255u) {
        unsigned long long cur = tab[h];
        if (cur == bits) return;
        if (cur == DINV_EMPTY) {
            cur = atomicCAS(tab + h, DINV_EMPTY, bits);
            if (cur == DINV_EMPTY || cur == bits) return;
        }
    }
    *overflow = 1;
}
__global__ __launch_bounds__(256) void dinv_code_kernel(int n, const double *__restrict__ dinv, const unsigned long long *__restrict__ tab,
                                                        unsigned char *__restrict__ code) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(dinv[i]);
    unsigned h = dinv_slot0(bits);
    for (int probe = 0; probe < 256 && tab[h] != bits; ++probe) h = (h + 1) & 255u;
    code[i] = (unsigned char)h;
}
void build_dinv_codes(hipStream_t s, DCsr &A, const double *dinv) {
    A.sell_dcode.release();
    A.sell_dsrc = nullptr;
    // (only the staged kernel of the coded formats reads them.  options().sell bit 5, off by default: measured on the 256^3
    // problem, 960 instead of 1 079 MB per application (PMC: 967 / 1 086) and 205.1 instead of 210.5 us -- the kernel is not
    // bound by its bytes alone, 2.3 % for 11 % of them)
    if (!A.nrows || !A.has_sell || A.sell_wq <= 0 || !(options().sell & 32)) return;
    A.sell_dtab.alloc(256);
    std::vector<unsigned long long> empty(256, DINV_EMPTY);
    SA_HIP_CHECK(hipMemcpyAsync(A.sell_dtab.p, empty.data(), 256 * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    DBuf<int> overflow(1);
    overflow.zero(s);
    SA_HIP_CHECK(hipStreamSynchronize(s));      // (empty is a local)
    hipLaunchKernelGGL(dinv_tab_kernel, dim3(div_up(A.nrows, 256)), dim3(256), 0, s, A.nrows, dinv, A.sell_dtab.p, overflow.p);
    SA_HIP_CHECK(hipGetLastError());
    if (overflow.to_host(s)[0]) return;
    A.sell_dcode.alloc((size_t)A.nrows);
    hipLaunchKernelGGL(dinv_code_kernel, dim3(div_up(A.nrows, 256)), dim3(256), 0, s, A.nrows, dinv, A.sell_dtab.p, A.sell_dcode.p);
    SA_HIP_CHECK(hipGetLastError());
    A.sell_dsrc = dinv;
}

// ---- deterministic dot product -----------------------------------------------------------
__device__ inline double block_sum_256(double v, double *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;  // valid on thread 0
}

__global__ __launch_bounds__(256) void dot_partial_kernel(int n, const double *__restrict__ a,
                                                          const double *__restrict__ b,
                                                          double *__restrict__ partials) {
    __shared__ double sh[4];
    double v = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        v = fma(a[i], b[i], v);
    const double r = block_sum_256(v, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void dot_final_kernel(int np, const double *__restrict__ partials,
                                                        double *__restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) v += partials[i];
    const double r = block_sum_256(v, sh);
    if (threadIdx.x == 0) out[0] = r;
}

void dot(hipStream_t s, int n, const double *a, const double *b, double *partials, double *out) {
    int grid = div_up(n, 256 * 8);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    profiler().begin(s);
    hipLaunchKernelGGL(dot_partial_kernel, dim3(grid), dim3(256), 0, s, n, a, b, partials);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, grid, partials, out);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "dot", 16.0 * n, 2.0 * n);
}

__global__ __launch_bounds__(256) void pcg_update_xr_kernel(int n, const double *__restrict__ sc,
                                                            double *__restrict__ x,
                                                            double *__restrict__ r,
                                                            const double *__restrict__ d,
                                                            const double *__restrict__ z) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double alpha = sc[0] / sc[1];
    x[i] = fma(alpha, d[i], x[i]);
    r[i] = fma(-alpha, z[i], r[i]);
}
void pcg_update_xr(hipStream_t s, int n, const double *sc, double *x, double *r, const double *d,
                   const double *z) {
    if (!n) return;
    profiler().begin(s);
    hipLaunchKernelGGL(pcg_update_xr_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, sc, x, r, d, z);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "pcg_update_xr", 48.0 * n, 4.0 * n);
}

__global__ __launch_bounds__(256) void pcg_update_d_kernel(int n, const double *__restrict__ sc,
                                                           double *__restrict__ d,
                                                           const double *__restrict__ z) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double beta = sc[2] / sc[0];
    d[i] = fma(beta, d[i], z[i]);
}
void pcg_update_d(hipStream_t s, int n, const double *sc, double *d, const double *z) {
    if (!n) return;
    profiler().begin(s);
    hipLaunchKernelGGL(pcg_update_d_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, sc, d, z);
    SA_HIP_CHECK(hipGetLastError());
    profiler().end(s, "pcg_update_d", 24.0 * n, 2.0 * n);
}

void vec_copy(hipStream_t s, int n, const double *src, double *dst) {
    if (n) SA_HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
}
void vec_zero(hipStream_t s, int n, double *dst) {
    if (n) SA_HIP_CHECK(hipMemsetAsync(dst, 0, sizeof(double) * n, s));
}
__global__ __launch_bounds__(256) void axpy_kernel(int n, double a, const double *__restrict__ x,
                                                   double *__restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = fma(a, x[i], y[i]);
}
void vec_axpy(hipStream_t s, int n, double a, const double *x, double *y) {
    if (!n) return;
    hipLaunchKernelGGL(axpy_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, a, x, y);
    SA_HIP_CHECK(hipGetLastError());
}

}  // namespace saamge_amd
