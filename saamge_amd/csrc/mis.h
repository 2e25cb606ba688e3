// Per-MIS gather + SVD (tentative prolongator blocks), P / R assembly, Galerkin product.
#pragma once
#include "common.h"
#include "topology.h"

namespace saamge_amd {

struct MisSvdIO {
    // inputs
    const int *ae_m = nullptr;          // [nparts] eigenvectors per AE
    const int64_t *ae_xoff = nullptr;   // [nparts] offset of AE i's vectors (n_i x m_i, col-major)
    const double *evecs = nullptr;
    const int64_t *g_off = nullptr;     // [num_mises+1] offsets of the gathered matrix (r x ctot)
    double *gather = nullptr;           // scratch, sum r*ctot
    const int64_t *u_off = nullptr;     // [num_mises+1] offsets of U storage (r x min(r,ctot))
    const int64_t *s_off = nullptr;     // [num_mises+1] offsets of singular values (ctot each)
    // outputs
    double *U = nullptr;
    double *sig = nullptr;
    int *k = nullptr;                   // [num_mises] kept vectors  (mis_numcoarsedof)
    int *ncols = nullptr;               // [num_mises] columns that entered the SVD
    int avoid_ess = 1;
    // extra per-dof modes appended after the spectral columns (ExtendWithPolynomials / RBMs,
    // amg/src/contrib.cpp:302-436): ND x nextra, column-major; level 0 only
    const double *extra = nullptr;
    int nextra = 0, ND = 0;
};

// ContribTent::SVDInsert on every MIS (amg/src/contrib.cpp:551-687): essential-boundary
// filter (:102-163), column normalisation + SVD (amg/src/xpacks.cpp:494-589, one-sided
// Jacobi instead of dgesvd), cut at sigma > 1e-10 sigma_max (:591-620).
// `m0`: first MIS of the range [m0, m0 + num_mises) this call works on (ranks of a multi-GPU setup take
// contiguous ranges and all-gather the bases).
// ae_ev (optional, single rank, the whole level in one call): per agglomerate the class whose eigenpairs it received a copy of
// (-1: computed on its own) -- MISes with identical inputs are then decomposed once (mis.hip, "Classes of identical MISes").
void mis_svd(hipStream_t s, const DevRelations &rel, int num_mises, int max_ctot, const MisSvdIO &io, int m0 = 0,
             const int *ae_ev = nullptr);

// P (ND x nc) and R = P^T from the MIS blocks (contrib_tent_insert_simple,
// amg/src/contrib.cpp:170-194; explicit zeros are kept in the block pattern).
void build_P_R(hipStream_t s, const DevRelations &rel, const Relations &hrel,
               const std::vector<int> &h_k, const std::vector<int64_t> &h_u_off, const int *d_k,
               const int *d_coloff, const int64_t *d_u_off, const double *U, DCsr &P, DCsr &R);

// Coarse elements of the next level on the device: coarse elem_to_dof = AE_to_dof x pattern(P_tent) in
// first-encounter order (agg_create_rels_except_elem_coarse, amg/src/aggregates.cpp:1510-1514; the pattern
// is the numerically non-zero entries, contrib_tent_insert_simple, amg/src/contrib.cpp:186-187), and for
// every (AE, MIS) incidence t (aligned with AE_to_mis.J) the positions colpos[colpos_ptr[t] + v] of the
// MIS's coarse dofs in the coarse element's dof list.  e2d is fetched to the host (the next level's
// topology is built from it).  Returns false when an AE has too many coarse dofs for the LDS kernel
// (the caller then takes the host path).
bool coarse_e2d_device(hipStream_t s, const DevRelations &rel, const Relations &hrel, const int *d_mis_k,
                       const int *d_mis_coloff, int ncoarse, const roff_t *p_rowptr, const double *p_val,
                       DBuf<int> &colpos_ptr, DBuf<int> &colpos, Table &e2d);

// Ac = P^T A P exploiting the MIS block structure of P (tg_coarse_matr, amg/inc/tg.hpp:696-709).
void rap_mis(hipStream_t s, const DevRelations &rel, const Relations &hrel, const DCsr &A,
             const std::vector<int> &h_k, const std::vector<int> &h_coloff, const int *d_k,
             const int *d_coloff, const int64_t *d_u_off, const double *U, DCsr &Ac, int rank = 0,
             int world = 1, std::vector<long long> *nnz_off = nullptr);

}  // namespace saamge_amd
