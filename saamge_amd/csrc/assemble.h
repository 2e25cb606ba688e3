// Batched agglomerate (AE) matrix assembly, weighted-l1 scaling, coarse element matrices.
#pragma once
#include "common.h"
#include "eig.h"
#include "topology.h"

namespace saamge_amd {

// Packed element matrices of one level: element e is a dense nd_e x nd_e block
// (nd_e = elem_to_dof row size) stored row-major at val[off[e]].
struct DevElmats {
    DBuf<int64_t> off;  // [NE+1]
    DBuf<double> val;
    int nde = 0;        // > 0: every element has exactly nde dofs and e2d_J / val are dense arrays
    // per-rank inputs: val holds the matrices of the elements [first, first + count) only; kernels that index the array by
    // the element id itself (no `off`) take dense() -- they touch the rank's own elements only
    int64_t first = 0;
    const double *dense() const { return val.p - first * (int64_t)nde * nde; }
    int algebraic = 0;       // element-free mode: the AE matrices come from A (no element matrices);
                             // 1 = ExtractSubMatrices, 2 = WindowSubMatrices
};

// Dense AE matrices for the AEs [ae0, ae0+count) into batch.W (column-major, ld = n_i).
//   fine level  (A != null): agg_build_AE_stiffm_with_global, amg/src/aggregates.cpp:855-945
//   coarse level (A == null): agg_build_AE_stiffm (plain sum), amg/src/aggregates.cpp:959-1086
// `banded` (few large agglomerates in permuted order, coarse levels): the half bandwidths come from the topology
// (batch.bw, has_bw) and only the band the eigensolver reads is cleared
void ae_assemble(hipStream_t s, const DevRelations &rel, const DCsr *A, const DevElmats &el,
                 int ae0, EigBatch &batch, bool banded = false, const int *ae_ids = nullptr);

// D_ii = sum_j |a_ij| sqrt(a_ii/a_jj)  (amg/src/mbox.cpp:913-949);  batch.dis = D^-1/2 and
// W <- D^-1/2 W D^-1/2 in place.  Dout (optional, packed like batch.d) receives D.
// split: -1 = chosen by the batch's shape (few large agglomerates: row sums and scaling spread over several kernels), 0 / 1 forced
void ae_scale(hipStream_t s, EigBatch &batch, double *Dout, int split = -1);

// longest row of a CSR operator (one small kernel + a read-back; callers cache it in A.max_row)
int csr_max_row(hipStream_t s, const DCsr &A);

// ae_assemble (+ ae_scale when `scale`): on the fine level, when the sparse rows of one AE fit in
// LDS, as ONE fused kernel that writes the dense image once (see assemble.hip).
// `rows` (optional): where the chunk's first row sits among the rows of all agglomerates of the level
// and how many there are -- the sparse rows are then kept for the rest of the hierarchy build
// (ae_rows_new_build() starts a new one) and not recomputed by later passes over the same AEs.
struct RowsSpan {
    int64_t first = 0, total = 0;
};
void ae_rows_new_build();
// classes (optional, in/out): where the assembly can tell identical agglomerates apart before it builds their matrices
// (the fused fine-level path), it builds the first member of every class only and reports the classes here (early = false:
// every matrix of the batch was built).
struct AeClasses {
    bool early = false;     // the classes below were found on the sparse rows and only cls.reps were built
    bool searched = false;  // the sparse rows were searched (early = false then: too few identical agglomerates to bother)
    DdSource src;           // ... where those rows are (valid until the next ae_build)
    DdClasses cls;
};
void ae_build(hipStream_t s, const DevRelations &rel, const DCsr *A, const DevElmats &el, int ae0,
              EigBatch &batch, bool scale, double *Dout, const RowsSpan *rows = nullptr, AeClasses *classes = nullptr);

// Fine level, 8-dof elements: the sparse rows of the AE matrices of a chunk (RW slots per row at
// rv / rc[(batch.voff[b] + row) * RW + slot], column -1 = empty).  false: not applicable.
bool ae_sparse_rows(hipStream_t s, const DevRelations &rel, const DCsr &A, const DevElmats &el, int ae0,
                    const EigBatch &batch, int &RW, const double *&rv, const short *&rc,
                    const RowsSpan *rows = nullptr);
// E_e from those rows (no dense AE matrix)
void coarse_elmats_sparse(hipStream_t s, const DevRelations &rel, int ae0, const EigBatch &batch, int RW,
                          const double *rv, const short *rc, const int *mis_k, const int64_t *mis_u_off,
                          const double *mis_u, const int *colpos_ptr, const int *colpos, const int64_t *out_off,
                          double *out, double *scratch, const int64_t *scratch_off, int kmax, const int *ae_class = nullptr);

// Coarse element matrices E_e = P_loc^T A_e P_loc for AEs [ae0, ae0+count)
// (ElementMatrixParallelCoarse::GetMatrix, amg/src/elmat.cpp:105-195).  batch.W must hold
// the *unscaled* AE matrices.  Output row-major at out[out_off[e]], k_e x k_e.
void coarse_elmats(hipStream_t s, const DevRelations &rel, int ae0, const EigBatch &batch,
                   const int *mis_k, const int64_t *mis_u_off, const double *mis_u,
                   const int *colpos_ptr, const int *colpos, const int64_t *out_off, double *out,
                   double *scratch, const int64_t *scratch_off);

}  // namespace saamge_amd
