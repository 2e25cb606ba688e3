// General sparse matrix products and transpose (see spgemm.hip): the smoothed-prolongator path.
#pragma once
#include "common.h"

namespace saamge_amd {

// C = beta E + alpha diag(d) A B.  E (optional) has A's rows and B's columns; d (optional) has one
// entry per row of A.  Rows of C come out sorted by column; summation order is fixed.
void spgemm(hipStream_t s, const DCsr &A, const DCsr &B, const DCsr *E, const double *d, double alpha,
            double beta, DCsr &C);
// R = P^T, rows sorted by column
void csr_transpose(hipStream_t s, const DCsr &P, DCsr &R);
// C = entries of A with |v| > tol, order kept (AltThreshold, amg/src/interp.cpp:89-170)
void csr_threshold(hipStream_t s, const DCsr &A, double tol, DCsr &C);

}  // namespace saamge_amd
