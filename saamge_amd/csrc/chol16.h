// The 16 x 16 diagonal-block kernel of the banded factorisations: Cholesky (or signed L S L^T) factor and its
// inverse by ONE wavefront.  Included inside namespace saamge_amd by eig2.hip (SB = 16 defined there).
#pragma once

// a pivot of C - theta I below this (in modulus) means theta sits on an eigenvalue of a leading block to
// within round-off growth: the inertia count is then not trusted (the caller takes the dense path, whose Sturm
// count has no such restriction)
constexpr double SS_PIV_TINY = 1e-7;

__device__ inline double readlane_f64(double v, int src) {       // src: wave-uniform
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// (Measured alternative, round 2: the same elimination with DPP row broadcasts instead of readlane + the LDS
// forward substitution gives bit-identical factors and inverses but 4.1 us per call against 3.3 us, 4 wavefronts
// per CU -- the VALU -> DPP hazards cost more than the scalar round trips.)
// Cholesky L of the SB x SB block in Ld (lower triangle; rows past a partial block = identity) and
// Li = L^-1, by one wavefront.  Lane i < SB keeps row i in registers; the pivot row entries reach
// the others through readlane (no LDS round trips, no barriers), 1 / sqrt(pivot) by rsq + Newton.
// Then L goes to Ld once and lane j builds column j of the inverse by forward substitution with
// the reciprocal pivots.  Returns non-zero when a pivot was not positive.
// SIGNED: the block may be indefinite -- A = L S L^T with S = diag(+-1) (LDL^T without pivoting,
// |pivot|^1/2 folded into L): sg[j] receives the sign of pivot j, the return value is the number
// of negative pivots, or -1 when a pivot is too small for the count to be trusted (the inertia of
// C - theta I certifies the number of eigenvalues below theta; a pivot below SS_PIV_TINY means
// theta sits on an eigenvalue of a leading block to within round-off growth: the caller falls
// back to the dense path, whose Sturm count has no such restriction).
template <bool SIGNED>
__device__ __forceinline__ int chol16_inverse_wave(double (*Ld)[SB + 1], double (*Li)[SB + 1], int lane, double *sg = nullptr) {
    const int li = lane & (SB - 1);
    double a[SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) a[c] = Ld[li][c];
    int isbad = 0, nneg = 0;
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const double dj = readlane_f64(a[j], j);
        const double ad = SIGNED ? fabs(dj) : dj;
        const bool ok = SIGNED ? (ad > SS_PIV_TINY) : (dj > 0.0);
        isbad |= !ok;
        double r = 1.0;
        if (ok) {                                   // (wave-uniform)
            r = __builtin_amdgcn_rsq(ad);
            r = r * fma(-0.5 * ad * r, r, 1.5);
            r = r * fma(-0.5 * ad * r, r, 1.5);
        }
        if (lane == 0) Li[0][j] = r;                // reciprocal pivots: parked in row 0 of Li until the inverse is written
        double sj = 1.0;
        if (SIGNED) {
            sj = (dj < 0.0) ? -1.0 : 1.0;
            nneg += (dj < 0.0) ? 1 : 0;
            if (lane == 0) sg[j] = sj;
            r *= sj;                                // L(i, j) = s_j A(i, j) / |d_j|^1/2, L(j, j) = |d_j|^1/2
        }
        a[j] *= r;                                  // column j of L on the lanes i >= j (lane j: sqrt(dj))
        const double ajs = SIGNED ? sj * a[j] : a[j];
#pragma unroll
        for (int c = j + 1; c < SB; ++c) {
            const double lcj = readlane_f64(a[j], c);
            a[c] = fma(-ajs, lcj, a[c]);            // meaningful on the lanes i >= c
        }
    }
    if (lane < SB) {
#pragma unroll
        for (int c = 0; c < SB; ++c) Ld[lane][c] = a[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < SB) {
        const int j = lane;
        double x[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            double t = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < i; ++q) t = fma(-Ld[i][q], x[q], t);     // (x[q] = 0 above the diagonal)
            x[i] = (i >= j) ? t * Li[0][i] : 0.0;
            __builtin_amdgcn_sched_barrier(0);      // (keeps the 120 LDS operands from being hoisted into registers at once)
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) Li[i][j] = x[i];
    }
    if (SIGNED) return isbad ? -1 : nneg;
    return isbad;
}

