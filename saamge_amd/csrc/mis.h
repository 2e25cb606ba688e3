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
void mis_svd(hipStream_t s, const DevRelations &rel, int num_mises, int max_ctot, const MisSvdIO &io);

// P (ND x nc) and R = P^T from the MIS blocks (contrib_tent_insert_simple,
// amg/src/contrib.cpp:170-194; explicit zeros are kept in the block pattern).
void build_P_R(hipStream_t s, const DevRelations &rel, const Relations &hrel,
               const std::vector<int> &h_k, const std::vector<int64_t> &h_u_off, const int *d_k,
               const int *d_coloff, const int64_t *d_u_off, const double *U, DCsr &P, DCsr &R);

// Ac = P^T A P exploiting the MIS block structure of P (tg_coarse_matr, amg/inc/tg.hpp:696-709).
void rap_mis(hipStream_t s, const DevRelations &rel, const Relations &hrel, const DCsr &A,
             const std::vector<int> &h_k, const std::vector<int> &h_coloff, const int *d_k,
             const int *d_coloff, const int64_t *d_u_off, const double *U, DCsr &Ac, int rank = 0,
             int world = 1, std::vector<long long> *nnz_off = nullptr);

}  // namespace saamge_amd
