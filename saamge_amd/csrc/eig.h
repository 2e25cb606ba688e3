// Batched dense symmetric eigensolver: all eigenpairs of C_i with eigenvalue in
// (vl, vu], at least one (the smallest) -- the device counterpart of
// xpacks_calc_lower_eigens_dense / LAPACK dsygvx (reference: amg/src/xpacks.cpp:222-314)
// after the generalized problem A x = lambda D x has been reduced with the diagonal D:
// C = D^-1/2 A D^-1/2, x = D^-1/2 y.
#pragma once
#include "common.h"

namespace saamge_amd {

constexpr int EIG_NB = 32;  // panel width of the one-stage blocked tridiagonalisation
constexpr int EIG_SB = 16;  // band width of the two-stage reduction

struct EigBatch {
    int count = 0;          // matrices in the batch
    int slot = 0;           // which of the two persistent workspaces backs this batch (chunk pipelining)
    int max_n = 0;
    DBuf<int> n;            // [count] sizes
    DBuf<int64_t> moff;     // [count] offset of matrix i in W (doubles), column-major ld = n_i
    DBuf<int64_t> voff;     // [count] offset of matrix i's row block in the per-row arrays
    DBuf<double> W;         // matrices, overwritten by the reflectors
    DBuf<double> panel;     // [sum n_i * EIG_NB] W-panel of the blocked reduction
    DBuf<double> d, e, tau; // [sum n_i] tridiagonal + reflector scalars
    DBuf<double> dis;       // [sum n_i] D^-1/2 (row scaling applied to the vectors)
    DBuf<int> m, j0;        // [count] number of wanted pairs, index of the first (-1: nothing in (vl, vu], smallest pair)
    double vl = 0.0, vu = 0.0;   // the window eig_count was called with (eig_vectors selects per decoupled block)
    // two-stage reduction (eig2.hip)
    bool two_stage = false;
    DBuf<double> Tfac;      // [sum n_i * SB] compact-WY T factors, one SB x SB block per panel
    DBuf<double> Xbuf, Zbuf;// [sum n_i * SB] panel products (Z row-major: Z(r, c) at r * SB + c)
    DBuf<double> Vpk, Vpk2; // [sum n_i * SB] panel V (current / next), row-major, unit diagonal / zeros explicit
    DBuf<double> trash;     // 128 slots: target of the masked-off rows of the update kernels
    DBuf<double> rv, rtau;  // bulge-chasing reflectors (SB entries each) and their scalars
    DBuf<double> bandg;     // band + bulge storage when it does not fit in LDS
    DBuf<int64_t> roff;     // [count+1] reflector offsets
    DBuf<double> Gbuf;      // per (matrix, 64-row block) partial V^T X (SB x SB each)
    DBuf<int64_t> goff;
    // symmetric fused update (SAAMGE_AMD_EIG_FUSED=3): per matrix the 64 x SB partial products of
    // the tiles below the diagonal, tile (I, J) at xpoff[b] + I (I - 1) / 2 + J
    DBuf<double> Xpart;
    DBuf<int64_t> xpoff;
    std::vector<int64_t> h_xpoff;
    std::vector<int64_t> h_roff, h_goff;
    // few-eigenpairs path (SAAMGE_AMD_EIG=subspace): Ritz values of the accepted block; `dense_only`
    // forces the dense path for this batch (fallback after a failed subspace attempt)
    DBuf<double> ss_mu;
    // locked (converged and deflated) pairs of matrices with more wanted pairs than one block holds: see ss_lock_kernel
    DBuf<double> ss_Vlock, ss_lock_mu;
    DBuf<int> ss_ndefl;
    bool ss_has_lock = false;
    DBuf<double> ss_sigma;          // shift of every matrix (few-eigenpairs path), see eig_subspace_factor
    // [sum n_i] position of agglomerate-local row r in the matrix as assembled (rows ordered by global
    // dof number: a far narrower band than the first-encounter order of the tables); has_perm = false:
    // identity.  Only the fused assembly sets it, and only for batches that take this path.
    DBuf<short> perm, iperm;      // (iperm: row of the agglomerate at a position of the matrix)
    bool has_perm = false;
    bool has_bw = false;    // bw was filled by the assembly (from the sparse rows): no scan of the dense matrices
    DBuf<int> bw;           // [count] half bandwidths (banded Cholesky), host copy; empty = full matrices
    std::vector<int> h_bw;
    int ss_bwmax = 0;
    // certified count: the upper end of the window must be known when the matrices are factored
    // (set_window before eig_tridiagonalize); h_inertia[i] = #{eigenvalues of C_i < vu} from the
    // inertia of C_i - vu I, -1 = not trustworthy (tiny pivot)
    bool has_window = false;
    double window_vu = 0.0;
    std::vector<int> h_inertia;
    DBuf<int> inertia;      // device copy (the iteration accepts a matrix when its Ritz count reaches it)
    // matrices whose only wanted pair is known before any factorisation (certified count 1 and the start vector
    // D^1/2 1 already an eigenvector to the acceptance tolerance: every agglomerate without essential rows of a
    // diffusion-type operator): pre[i] = 1, pre_val[2 i] = Rayleigh quotient, pre_val[2 i + 1] = 1 / |x0|
    std::vector<int> h_pre;
    DBuf<int> pre;
    DBuf<double> pre_val;
    // coarse levels: the level's representation of the constant vector (R ... R 1) on the rows of the batch, in
    // agglomerate order -- D^1/2 of it is the first start vector of the few-eigenpairs iteration (instead of
    // D^1/2 1) and the candidate of the known-null-vector shortcut
    DBuf<double> x0c;
    bool has_x0c = false;
    void set_window(double vu_) { has_window = true; window_vu = vu_; }
    bool subspace = false, dense_only = false, ss_failed = false;
    int ss_nb = 8;                // vectors of the iteration's block: 8 (bands resident in LDS), 16 (wide bands: the solves' matrix-core tiles
                                  // are 16 columns wide either way, the block converges at (lambda_i - sigma) / (lambda_17 - sigma))
    double ss_tol = 1e-12;        // acceptance bound of a Ritz pair's residual (saamge_amd_params.eig_tol)
    // few-eigenpairs path, per matrix: h_bad[i] = 1 -- this matrix has to be redone by the dense path (more wanted
    // pairs than the block holds, no certificate, a non-positive pivot, no or hopeless convergence, a count that
    // contradicts the inertia).  The others are finished; h_m[i] = 0 for the bad ones.  With more than a tenth of
    // the batch bad the whole batch fails instead (ss_failed).
    std::vector<char> h_bad;
    int nbad = 0;
    // wide-band matrices: the band of every matrix as saved before the in-place inertia factorisation (arena
    // buffer + offsets): a matrix whose convergence rate is hopeless at its first shift is restored from it and
    // factored again at a shift just below its smallest Ritz value (eig_subspace_iterate)
    double *ss_save = nullptr;
    // packed sub-panels of the blocked wide-band factorisations (arena buffer): taken from the arena by the FIRST factorisation of
    // the batch -- on the thread that owns the workspace -- and kept here, so that a later factorisation from the iteration's own
    // thread (a matrix factored again at a better shift) does not touch the arena
    double *ss_sub = nullptr;
    size_t ss_sub_n = 0;
    DBuf<int64_t> ss_soff;
    std::vector<double> h_sigma;
    std::vector<int> h_n, h_m;
    std::vector<int64_t> h_moff, h_voff;
};

// sizes known on the host; allocates everything but leaves W/dis to be filled by the caller
void eig_batch_alloc(EigBatch &b, const std::vector<int> &sizes, hipStream_t s, int slot = 0);

// Phase 1: tridiagonalise every matrix in place.  The default is the two-stage reduction
// (eig2.hip); SAAMGE_AMD_EIG=onestage selects the one-stage blocked Householder kernel.
// `phases`: 1 = dense -> band only, 2 = band -> tridiagonal only, 3 = both.  The split lets the
// caller run the (latency-bound) bulge chasing of one chunk beside the (bandwidth-hungry)
// band reduction of the next one on another stream.
void eig_tridiagonalize(hipStream_t s, EigBatch &b, int phases = 3);
void eig_tridiagonalize_two_stage(hipStream_t s, EigBatch &b, int phases);
bool eig_uses_two_stage();
void eig_backtransform_two_stage(hipStream_t s, EigBatch &b, const int64_t *xoff, double *evecs);
int64_t chase_reflector_count(int n);
void eig_batch_two_stage_buffers(EigBatch &b, size_t nrefl, bool need_bandg, hipStream_t s);
// few-eigenpairs path (eig2.hip), see there
bool eig_use_subspace();
bool eig_ss_band_enabled();                          // banded factorisation on (SAAMGE_AMD_SS_BAND != 0)
bool eig_batch_takes_subspace(const EigBatch &b);   // what eig_tridiagonalize will decide for this batch
bool eig_subspace_factor(hipStream_t s, EigBatch &b);
bool eig_subspace_iterate(hipStream_t s, EigBatch &b, double vu);
void eig_subspace_vectors(hipStream_t s, EigBatch &b, const int64_t *eoff, const int64_t *xoff, double *evals,
                          double *evecs);
void eig_arena_release();   // frees the persistent workspace
double *eig_arena_bandsave(const EigBatch &b, size_t doubles);   // persistent scratch of the inertia pass
double *eig_arena_subpanels(const EigBatch &b, size_t doubles); // packed sub-panels of the outer blocks of the wide-band factorisations
// Duplicate matrices (eig.hip, "Duplicate agglomerate matrices").  DdSource: where the words of the matrices of a batch are
// (kind 1: the assembled matrices; kind 0: the sparse rows the fused fine-level assembly builds them from).
struct DdSource {
    int kind = 1;
    const int *ns = nullptr;
    const int64_t *moff = nullptr, *voff = nullptr;
    const double *W = nullptr;
    const int *bws = nullptr;
    const double *dis = nullptr;
    const short *perm = nullptr;      // or null
    const double *x0c = nullptr;      // or null
    const double *rvals = nullptr;    // kind 0
    const short *rcols = nullptr;
    int RW = 0;
};
struct DdKey { unsigned long long a, b; bool operator==(const DdKey &o) const { return a == o.a && b == o.b; } };
struct DdKeyHash { size_t operator()(const DdKey &k) const { return (size_t)(k.a ^ (k.b * 0x9E3779B97F4A7C15ull)); } };
// classes of a batch: reps = the first matrix of every class of bitwise identical matrices (confirmed word by word),
// rep_of[i] = the position in reps of matrix i's class, rep_hash = the 128-bit hash of every class (two words each)
struct DdClasses {
    std::vector<int> reps, rep_of;
    std::vector<unsigned long long> rep_hash;
};
DdSource eig_dedupe_source(const EigBatch &b);
bool eig_dedupe_find(hipStream_t s, const DdSource &src, int count, int max_n, DdClasses &out);      // false: fewer than a quarter duplicates
int eig_dedupe_group(const unsigned long long *hh, int count, std::vector<int> &rep);      // rep[i] = first matrix with i's hash; returns the classes
std::vector<unsigned long long> eig_dedupe_hash_list(hipStream_t s, const DdSource &src, int max_n, const std::vector<int> &list);
std::vector<long> eig_dedupe_words(hipStream_t s, const DdSource &src, const std::vector<int> &h_n, const std::vector<int> &list);
void eig_dedupe_pack(hipStream_t s, const DdSource &src, int max_n, const std::vector<int> &list, const std::vector<long> &words,
                     std::vector<DBuf<unsigned long long>> &blobs);
void eig_dedupe_compare(hipStream_t s, const DdSource &src, int max_n, const std::vector<int> &list,
                        const std::vector<const unsigned long long *> &blobs, const std::vector<long> &blob_words, std::vector<char> &same);
// the batch of some matrices of `full` (class representatives) in the same workspace; results to every member of the classes
void eig_batch_compact(hipStream_t s, EigBatch &cb, EigBatch &full, const std::vector<int> &reps);
void eig_dedupe_expand(hipStream_t s, int count, int max_n, const double *const *src_evals, const double *const *src_evecs,
                       const int64_t *eoff, const int64_t *xoff, double *evals, double *evecs);
// bytes of device workspace one matrix of size n needs (for chunk sizing)
size_t eig_workspace_bytes(int n);
// Phase 2: count eigenvalues in (vl, vu] (Sturm); fills b.m / b.j0 and the host copy b.h_m
// (m_i = max(count, 1): the reference takes the single smallest pair when none qualifies).
void eig_count(hipStream_t s, EigBatch &b, double vl, double vu);
// Phase 3: eigenvalues by multisection, vectors by inverse iteration, back-transform,
// scale rows by dis.  evals/evecs are packed per matrix: eoff[i] (eigenvalues),
// xoff[i] (vectors, column-major n_i x m_i).
void eig_vectors(hipStream_t s, EigBatch &b, const int64_t *eoff, const int64_t *xoff,
                 double *evals, double *evecs);

}  // namespace saamge_amd
