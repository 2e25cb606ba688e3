// Dense direct coarsest solver: explicit inverse by block Gauss-Jordan (see dense.hip).
#pragma once
#include "common.h"

namespace saamge_amd {

// X (n x n, column-major, symmetric) = A^-1 for an SPD operator.  Returns false when a pivot is not
// positive (semi-definite operator): the caller falls back to the inner PCG.
bool dense_inverse_spd(hipStream_t s, const DCsr &A, DBuf<double> &X);
// y = X b, or y += X b ; b and y must not alias
void dense_symv(hipStream_t s, int n, const double *X, const double *b, double *y, bool add);

}  // namespace saamge_amd
