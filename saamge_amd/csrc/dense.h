// Dense direct coarsest solver: explicit inverse by block Gauss-Jordan (see dense.hip).
#pragma once
#include "common.h"

namespace saamge_amd {

// X (n x n, column-major, symmetric) = A^-1 for an SPD operator.  Returns false when a pivot is not
// positive (semi-definite operator): the caller falls back to the inner PCG.
bool dense_inverse_spd(hipStream_t s, const DCsr &A, DBuf<double> &X);
// The same elimination on a dense symmetric positive definite matrix already in device memory (n x n, in place), without
// any host synchronisation: w.info (one int, zeroed when the buffers are first reserved) becomes non-zero when a pivot
// was not positive -- in this or any earlier call with the same work buffers.
struct GjWork {
    DBuf<double> Pinv, Rp, Col;
    DBuf<int> info;
    void reserve(int nmax, hipStream_t s);
};
void dense_inverse_inplace(hipStream_t s, int n, double *X, GjWork &w);
void dense_zero(hipStream_t s, size_t nn, double *X);
// X[col * n + row] = val for the entries of a CSR block with LOCAL column indices (rowptr: absolute offsets into col / val)
void dense_scatter(hipStream_t s, int n, const roff_t *rowptr, const int *col, const double *val, double *X);
// y = X b, or y += X b ; b and y must not alias
void dense_symv(hipStream_t s, int n, const double *X, const double *b, double *y, bool add);

}  // namespace saamge_amd
