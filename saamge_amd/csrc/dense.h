// Dense Cholesky coarsest solver (see dense.hip).
#pragma once
#include "common.h"

namespace saamge_amd {

// L (n x n, column-major, lower) = chol(A).  Returns false when a pivot is not positive.
bool dense_cholesky_factor(hipStream_t s, const DCsr &A, DBuf<double> &L);
// x = L^-T L^-1 b ; `work` holds 2 n doubles; b and x may alias
void dense_cholesky_solve(hipStream_t s, int n, const double *L, const double *b, double *x, double *work);

}  // namespace saamge_amd
