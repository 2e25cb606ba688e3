// Direct coarsest solver for operators beyond the reach of one explicit dense inverse (see blocktri.hip).
#pragma once
#include "common.h"

namespace saamge_amd {

struct BlockTri {
    int n = 0, nblk = 0;
    std::vector<int> off;          // block k = rows [off[k], off[k+1]) of the permuted operator
    std::vector<size_t> soff;      // Sinv_k (n_k x n_k, symmetric) at Sinv.p + soff[k]
    DBuf<int> perm;                // perm[new] = old
    DBuf<double> Sinv;
    // couplings in the permuted numbering: Lo = entries of a row in the PREVIOUS block, Up = in the NEXT block
    DBuf<roff_t> lo_ptr, up_ptr;
    DBuf<int> lo_col, up_col;
    DBuf<double> lo_val, up_val;
    mutable DBuf<double> bp, z, t, xp, r, dx;      // work vectors
    int max_block = 0;
};

// Level structure + block factorisation.  Returns false (B released) when the operator cannot be handled: a level set
// beyond the block limit, or a non-positive pivot (semi-definite operator) -- the caller falls back to the inner PCG.
bool blocktri_factor(hipStream_t s, const DCsr &A, BlockTri &B);
// x = A^-1 b: block forward / backward substitution with the explicit inverses, then one step of iterative refinement
void blocktri_solve(hipStream_t s, const DCsr &A, const BlockTri &B, const double *b, double *x);

}  // namespace saamge_amd
