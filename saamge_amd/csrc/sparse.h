// CSR SpMV family, fused polynomial-smoother step, vector kernels (launch wrappers).
#pragma once
#include "common.h"

namespace saamge_amd {

// Optional row range of an operator application (row-partitioned solve: a rank applies only its
// own rows; x stays indexed by global column).  nrows < 0 = all rows.
struct RowRange {
    int row0 = 0, nrows = -1;
};

// build the SELL-64 copy of A (used by every routine below when present)
void build_sell(hipStream_t s, DCsr &A);
// y = A x
void spmv(hipStream_t s, const DCsr &A, const double *x, double *y, RowRange rr = RowRange());
// r = b - A x                                   (reference: amg/src/tg.cpp:115-116)
void spmv_residual(hipStream_t s, const DCsr &A, const double *x, const double *b, double *r,
                   RowRange rr = RowRange());
// x += P xc                                     (reference: amg/src/tg.cpp:129)
void spmv_add(hipStream_t s, const DCsr &P, const double *xc, double *x, RowRange rr = RowRange());
// xout = xin + scale * dinv_neg .* (A xin - b)  (reference: amg/inc/smpr.hpp:330-338)
void smooth_step(hipStream_t s, const DCsr &A, const double *dinv_neg, const double *b,
                 const double *xin, double *xout, double scale, RowRange rr = RowRange());
// xout = scale * dinv_neg .* (-b)   (the same step for xin == 0, no matrix traffic)
void smooth_first(hipStream_t s, int n, const double *dinv_neg, const double *b, double *xout,
                  double scale);
// dinv_neg_i = -1 / ( sqrt|a_ii| * sum_j |a_ij| / sqrt|a_jj| )
//                                               (reference: amg/src/mbox.cpp:1839-1861)
void build_dinv_neg(hipStream_t s, const DCsr &A, double *sqrt_diag_tmp, double *dinv_neg);
// byte codes of the smoother's diagonal factor (operators with <= 256 distinct values of it and the staged coded format)
void build_dinv_codes(hipStream_t s, DCsr &A, const double *dinv_neg);

// deterministic dot product: out[0] = sum a_i b_i ; `partials` holds >= 1024 doubles
void dot(hipStream_t s, int n, const double *a, const double *b, double *partials, double *out);
// PCG fused vector updates, scalars on the device:
// sc[0]=nom sc[1]=den sc[2]=betanom
void pcg_update_xr(hipStream_t s, int n, const double *sc, double *x, double *r,
                   const double *d, const double *z);   // alpha = nom/den
void pcg_update_d(hipStream_t s, int n, const double *sc, double *d, const double *z);  // beta = betanom/nom
void vec_copy(hipStream_t s, int n, const double *src, double *dst);
void vec_zero(hipStream_t s, int n, double *dst);
void vec_axpy(hipStream_t s, int n, double a, const double *x, double *y);  // y += a x

}  // namespace saamge_amd
