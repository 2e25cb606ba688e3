/* saamge_amd -- C ABI of the MI355X-native SAAMGE hot path (setup + solve).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Every
 * entry point names the reference interface (LLNL/saamge, paths relative to amg/) it
 * replaces.  Array arguments may be HOST or DEVICE pointers unless stated otherwise:
 * device pointers are used in place (zero copy), host pointers are uploaded once.
 * All indices are 32-bit like the reference's hypre/MFEM `int`; all reals are fp64.
 *
 * Return value: 0 on success, non-zero on failure (the reference aborts through
 * SA_ASSERT -> MPI_Abort, inc/common.hpp:635-647; here the message is kept in
 * saamge_amd_last_error()).  Not thread-safe (neither is the reference: global
 * singletons, src/process.cpp:46-48); one hierarchy may be used from one thread at a time.
 */
#ifndef SAAMGE_AMD_H
#define SAAMGE_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define SAAMGE_AMD_MAX_LEVELS 8

/* agg_dof_status_t bit flags, inc/aggregates.hpp:102-105 */
#define SAAMGE_AMD_BETWEEN_AES 0x01
#define SAAMGE_AMD_ON_ESS_DOMAIN_BORDER 0x02
#define SAAMGE_AMD_ON_PROC_IFACE 0x04
#define SAAMGE_AMD_OWNED 0x08

typedef struct saamge_amd_hierarchy saamge_amd_hierarchy; /* == ml_data_t, inc/ml.hpp:118-120 */

/* Options of the library's own machinery (no counterpart in the reference): what is left of the environment switches of
 * rounds 1-3.  The variants that were measured without gain are gone; these remain because tests need them to reach a code
 * path or because a caller may want them.  PROCESS-WIDE: saamge_amd_set_options() sets them, and every
 * saamge_amd_ml_produce_data* call sets them from params->options before it builds (the last hierarchy built wins; do not
 * build hierarchies with different options concurrently).  Environment variables that remain: SAAMGE_AMD_TIMING (phase
 * times on stderr), SAAMGE_AMD_SERIAL (no worker threads in the setup: counter passes), SAAMGE_AMD_POOL_MAX_GB (device
 * block cache, default 64), SAAMGE_AMD_THREADS (host threads of the host topology builds). */
typedef struct saamge_amd_options {
    int eig_strict;               /* 0.  1: a fallback of the few-eigenpairs path to the dense path is an ERROR */
    int eig_certify;              /* 1.  0: no inertia certificate (residual bounds only): kept to show what the certificate is for */
    int eig_min_n;                /* 64: smallest agglomerate of a batch that takes the few-eigenpairs path */
    int eig_force_fallback;       /* 0.  k > 0: every k-th matrix of a batch takes the per-matrix dense fallback (tests) */
    int eig_dense_only;           /* 0.  1: saamge_amd_lower_eigens_batched uses the dense path (a hierarchy: params.eigensolver) */
    int eig_dense_one_stage;      /* 0.  1: dense path by the one-stage blocked Householder reduction instead of the two-stage one */
    int eig_nullcheck;            /* 1: agglomerates whose one wanted pair is the known null vector skip the iteration */
    int eig_keep_inertia_factor;  /* 1: wide-band matrices with certified count 0 keep the factor of the inertia pass */
    int band_assembly;            /* 1: coarse-level agglomerate matrices are assembled, summed and scaled inside their band */
    int eig_dedupe;               /* 1: bitwise identical agglomerate matrices of a batch (structured meshes, piecewise constant
                                   * coefficients) are solved once, their eigenpairs copied to the other members of the class */
    int eig_outer_panels;         /* 8: 16-column panels per outer block of the wide-band factorisations (left-looking panels,
                                   * one rank-128 update of the trailing window on the matrix cores); 4: rank-64; 2: the
                                   * right-looking two-panel walk of rounds 2-3 */
    int overlap;                  /* 15: bit 0 subspace iteration of a chunk beside the next chunk's assembly, bit 1 halo exchange
                                   * beside the interior rows, bit 2 Galerkin product beside the next level's eigenproblems,
                                   * bit 3 the fine operator's SELL copy and smoother diagonal beside the AE tables */
    int sell;                     /* 31: SELL slice formats: bit 0 coded slices at all, bit 1 pair coding (values in the table),
                                   * bit 2 the short-chain kernel path, bit 3 operator-level dictionary, bit 4 3 x 3 node blocks,
                                   * bit 5 (off) the smoother's diagonal as byte codes where the rows of the operator repeat */
    int spmv_sell;                /* 0.  1: saamge_amd_spmv / spmv64 build and use the SELL copy (tests of the SELL kernels) */
    int debug;                    /* 0: bit 0 iteration traces of the few-eigenpairs path, bit 1 operator format census on stderr,
                                   * bit 2 level tags in the kernel profile */
    int host_heap_pad_mb;         /* 256: the first hierarchy of the process asks glibc (mallopt) never to trim its heap, to serve
                                   * blocks up to 32 MB from it and to grow it in steps of this many MiB.  The setup's host tables
                                   * (some tens of MB per hierarchy, some of them copied to and from the device as pageable
                                   * memory) otherwise go back to the kernel with every hierarchy; unmapping pages the GPU driver
                                   * has registered stalled the process's queues for ~20 ms at the start of the next setup in
                                   * half of the processes (measured: DESIGN.md section 7.0).  0: the allocator is left alone. */
} saamge_amd_options;
void saamge_amd_options_default(saamge_amd_options *o);
void saamge_amd_set_options(const saamge_amd_options *o);
void saamge_amd_get_options(saamge_amd_options *o);

/* == MultilevelParameters, inc/ml.hpp:59-114 (+ the hidden defaults of src/ml.cpp:64-67) */
typedef struct saamge_amd_params {
    int num_coarsenings;                      /* levels - 1 */
    double theta[SAAMGE_AMD_MAX_LEVELS];      /* spectral tolerance per coarsening */
    int nu_relax[SAAMGE_AMD_MAX_LEVELS];      /* smoother: SAS polynomial of degree 3 nu + 1 */
    int nu_pro[SAAMGE_AMD_MAX_LEVELS];        /* prolongator smoothing degree (interp_smooth, src/interp.cpp:172-229); 0 = tentative */
    int avoid_ess_bdr_dofs;                   /* src/ml.cpp:64, always true in the reference */
    int testmesh;                             /* mltest fixture: ones-vector on AE 0, src/interp.cpp:510-524 */
    int coarse_solver;                        /* 0 auto (explicit dense inverse up to 8192 rows, else inner PCG); 1 direct = the reference's
                                               * coarse_direct (src/tg.cpp:990-996): dense inverse up to 16384 rows, beyond that block-
                                               * tridiagonal elimination over a level structure of the operator's graph; 2 inner PCG;
                                               * 3 block-tridiagonal elimination whatever the size */
    double coarse_rtol;                       /* inner PCG tolerance on (B r, r), un-squared */
    int coarse_max_iter;
    long long workspace_bytes;                /* dense AE matrices are processed in chunks of this size (default 32 GiB) */
    int keep_debug;                           /* keep eigenpairs / singular values for inspection */
    /* Multi-GPU, one process per GPU (reference: MPI ranks, src/process.cpp).  Every rank passes
     * the SAME global problem; the AEs of each level are split into `world` contiguous ranges,
     * a rank solves the local spectral problems (src/interp.cpp:387-556) of its range only and
     * the eigenvectors are all-gathered in place through `allgather` (buf is a device pointer;
     * rank r owns bytes [byte_off[r], byte_off[r+1]); return 0 on success).  The bench/tests
     * implement it with torch.distributed (nccl = RCCL on ROCm, gloo for rehearsals). */
    int rank, world;
    int (*allgather)(void *ctx, void *buf_dev, const long long *byte_off);
    void *allgather_ctx;
    /* Row-partitioned solve (reference: HypreParMatrix::Mult's halo exchange in every operator
     * application of src/tg.cpp:91-132, MPI_Allreduce of the inner products of
     * src/mfem_addons.cpp:106-248).  When both callbacks are set, levels with at least
     * dist_min_local_rows rows per rank are applied by contiguous row blocks: before each SpMV
     * the interface entries of the input vector travel through `alltoallv` (device buffers; the
     * part for / from rank r is [byte_off[r], byte_off[r+1]) ), inner products and the restricted
     * residual are summed with `allreduce_sum`, coarse corrections and the final solution come
     * back through `allgather`.  Smaller levels and the coarsest solve stay replicated.  Both
     * callbacks receive allgather_ctx.  comm_stream_ordered != 0 promises that the callbacks
     * enqueue their work on the hierarchy's stream (RCCL through torch.distributed on the same
     * stream), so the library does not synchronise the stream around them. */
    int (*allreduce_sum)(void *ctx, double *buf_dev, long long count);
    int (*alltoallv)(void *ctx, const void *send_dev, const long long *send_byte_off, void *recv_dev,
                     const long long *recv_byte_off);
    long long dist_min_local_rows;            /* default 262144 */
    int comm_stream_ordered;
    /* MultilevelParameters::use_correct_nullspace (inc/ml.hpp, default true in the reference's
     * drivers): one more two-grid level under the coarsest spectral operator, interp = scaling_P
     * (src/contrib.cpp:655-668, src/interp.cpp:842-909), SAS smoother nu = 3 (CorrectNullspace,
     * src/solve.cpp:52-164, src/ml.cpp:225-236).  Its own coarse solve (one BoomerAMG V-cycle in
     * the reference) is this library's coarsest solver.  Default 0. */
    int correct_nullspace;
    /* ContribTent::ExtendWithPolynomials / ExtendWithRBMs (src/contrib.cpp:302-436): extra modes
     * given per fine dof -- constants, coordinates, rigid-body modes: the caller evaluates them --
     * are restricted to every MIS and appended after the spectral columns before the SVD (finest
     * level only, like the reference).  n x num_extra_modes, column-major, host or device. */
    const double *extra_modes;
    int num_extra_modes;
    /* Element-free mode (tg_produce_data_algebraic / ExtractSubMatrices, src/tg.cpp:579-672,
     * :862-886): elements are the dofs, partitions[0] maps DOFS to (non-overlapping) AEs, the AE
     * matrices are principal submatrices of A made rowsum-free; pass NE = n, nde = 1,
     * elem_to_dof = elmat = bdr_dofs = NULL.  1 = ExtractSubMatrices; 2 = WindowSubMatrices
     * (src/tg.cpp:741-858, `use_window`): A_TT + A_TX E, outside neighbours replaced by the
     * row-weighted average of the inside ones. */
    int algebraic;
    /* MultilevelParameters::smooth_drop_tol (inc/ml.hpp:93-113; interp_smooth -> AltThreshold,
     * src/interp.cpp:89-229): entries of the SMOOTHED prolongator (nu_pro > 0) with
     * |value| <= tol are dropped before R = P^T and Ac = RAP.  0 = keep everything. */
    double smooth_drop_tol;
    /* MultilevelParameters::do_aggregates (inc/ml.hpp; src/ml.cpp:149): on the LAST coarsening the
     * minimal intersection sets are replaced by one aggregate per AE; dofs shared by several AEs
     * are distributed greedily by strength of connection (agg_construct_aggregate_mises,
     * src/aggregates.cpp:324-487; Arbitrator::suggest, src/arbitrator.cpp:93-204).  Lower
     * operator complexity on the coarsest level. */
    int do_aggregates;
    /* Local eigensolver (Eigensolver::SolveDirect -> dsygvx, src/spectral.cpp:124-237,
     * src/xpacks.cpp:222-314).  0 (default): few-eigenpairs path -- banded Cholesky + shift-invert
     * subspace iteration, the count #{lambda <= theta} certified by the inertia of C - theta I,
     * any batch that fails certification redone by the dense path; 1: dense path only (dsygvx's
     * own algorithm: reduction to tridiagonal form, Sturm counts, inverse iteration). */
    int eigensolver;
    /* Few-eigenpairs path: a Ritz pair is accepted when the bound of its residual || C x - lambda x || (C scaled to
     * lambda_max <= 1, like the reference's generalised problem, src/spectral.cpp:134) is at most eig_tol.  Default 1e-12:
     * an invariant subspace is then determined to eig_tol / gap (the reference's dsygvx delivers eps / gap), i.e. to
     * ~1e-8 for the 7e-5 gaps of BASELINE config 4 (theta = 1e-4).  Measured there (tests/test_gpu_baseline_sizes.py, the
     * well-posed config-4 golden): 1e-12, 1e-13 and 1e-14 give the same level dimensions, the same counts and the same PCG
     * history to 3 digits of its 6e-8 deviation from the oracle's -- that deviation comes from the singular vectors of the
     * smallest kept singular values, not from the eigenvectors.  Allowed: 1e-15 ... 1e-8. */
    double eig_tol;
    saamge_amd_options options;               /* applied process-wide by saamge_amd_ml_produce_data* (see saamge_amd_options) */
} saamge_amd_params;

void saamge_amd_params_default(saamge_amd_params *p);

/* ---- native collectives: RCCL over xGMI, one process per GPU (csrc/comm.hip) --------------------------------
 * Rank 0 draws a unique id and ships it to the other ranks by any means (MPI_Bcast, a file, torch.distributed's
 * store); every rank then creates its communicator on the stream its hierarchy will run on and installs it in
 * the parameters: rank, world and the three collectives are filled in, comm_stream_ordered = 1.  The stream MUST be
 * the one later given to saamge_amd_ml_produce_data (the collectives are ordered with the hierarchy's kernels only
 * there): a mismatch is an error of saamge_amd_ml_produce_data.  The allgather / allreduce_sum / alltoallv callbacks
 * remain the plug for MPI host codes. */
typedef struct saamge_amd_comm saamge_amd_comm;
int saamge_amd_comm_unique_id(char id[128]);
int saamge_amd_comm_create(int rank, int world, const char id[128], void *stream, saamge_amd_comm **out);
void saamge_amd_comm_destroy(saamge_amd_comm *c);
int saamge_amd_params_set_comm(saamge_amd_params *p, saamge_amd_comm *c);
/* one all-reduce, all-gather and all-to-all of known data, checked: 0 = the communicator works */
int saamge_amd_comm_selftest(saamge_amd_comm *c);
const char *saamge_amd_comm_last_error(void);
/* hipMemcpy(hipMemcpyDefault) helper for all-gather callbacks written outside C (host or device
 * pointers on either side) */
int saamge_amd_memcpy(void *dst, const void *src, long long bytes);
const char *saamge_amd_last_error(void);

/* ml_produce_data (inc/ml.hpp:192-194, src/ml.cpp:379-472) on raw arrays:
 *   A            n x n CSR, essential rows/cols eliminated with the diagonal kept
 *                (== HypreParMatrix Ag / SparseMatrix Al of test/mltest/mltest.cpp:619-621)
 *   elem_to_dof  NE x nde, elmat NE x nde x nde row-major raw element matrices
 *                (== ElementMatrixProvider::GetMatrix, inc/elmat.hpp:53-78)
 *   bdr_dofs     n flags (== fem_find_bdr_dofs, src/fem.cpp:87-140); may be NULL
 *   partitions   partitions[k][e] = AE of level-k element e (level-k elements are the
 *                level-(k-1) AEs); replaces METIS, exactly like the non-NULL
 *                `partitioning` argument of agg_create_partitioning_fine
 *                (inc/aggregates.hpp:385-390)
 *   nparts       nparts[k] = number of AEs of coarsening k
 *   stream       hipStream_t to run on (NULL = default stream)                          */
int saamge_amd_ml_produce_data(int n, const int *rowptr, const int *col, const double *val,
                               int NE, int nde, const int *elem_to_dof, const double *elmat,
                               const signed char *bdr_dofs, const int *const *partitions,
                               const int *nparts, const saamge_amd_params *params, void *stream,
                               saamge_amd_hierarchy **out);
/* The same with 64-bit row offsets, for an operator with more than 2^31 stored entries on one GPU (the
 * reference's HYPRE_Int is 32-bit and reaches such sizes only split over MPI ranks): Q2 elasticity on 96^3
 * elements has 4.2e9.  Column indices and dimensions stay 32-bit.  Inside the library every operator carries
 * 64-bit row offsets; the 32-bit entry widens its input on the device. */
int saamge_amd_ml_produce_data64(int n, const long long *rowptr, const int *col, const double *val,
                                 int NE, int nde, const int *elem_to_dof, const double *elmat,
                                 const signed char *bdr_dofs, const int *const *partitions,
                                 const int *nparts, const saamge_amd_params *params, void *stream,
                                 saamge_amd_hierarchy **out);
/* ml_produce_data from PER-RANK inputs (one process per GPU; params->rank / world and the collectives set): what the
 * reference's multi-rank drivers pass (pmltest, amg/CMakeLists.txt:198-203; test/mltest/mltest.cpp:619-745):
 *   A            this rank's row block as a HypreParMatrix holds it -- hypre's ParCSR split (hypre_ParCSRMatrix: `diag` =
 *                the columns of the rank's own range with LOCAL indices, `offd` = the others, compressed, col_map_offd[c] =
 *                global column of offd column c; inc/SharedEntityCommunication.hpp:74-245 works on the same split).  Row
 *                blocks are contiguous and ordered by rank (row_starts, world + 1 entries, may be NULL); entries of a row
 *                may come in any order.
 *   elem_to_dof  NE_local x nde, GLOBAL (true) dof ids -- the reference's local dofs mapped through Dof_TrueDof;
 *   elmat        the matrices of the rank's OWN elements: they stay on this rank and are never exchanged;
 *   bdr_dofs     flags of the rank's own rows (A->nrows of them; may be NULL);
 *   partitions   partitions[0][e] = local agglomerate (0 .. nparts_local[0] - 1) of local element e; partitions[k], k > 0,
 *                maps the rank's level-(k-1) agglomerates to its level-k agglomerates: no agglomerate straddles ranks
 *                (the reference's invariant, src/aggregates.cpp:1340-1443).
 * Global numbering = rank order (rank 0's elements / agglomerates first).  The hierarchy is identical to the one
 * saamge_amd_ml_produce_data builds from the assembled global problem.  A rank assembles, factors and coarsens the
 * agglomerates made of its own elements; round 4: the operator's rows and the integer topology are all-gathered inside
 * (DESIGN.md section 6 lists what is replicated), the element matrices are not.  Host or device pointers. */
typedef struct saamge_amd_parcsr {
    long long global_rows;              /* 0 = the sum of the ranks' rows */
    const long long *row_starts;        /* world + 1 entries, or NULL */
    int nrows;                          /* rows of this rank */
    const int *diag_i, *diag_j;         /* nrows x nrows */
    const double *diag_a;
    const int *offd_i, *offd_j;         /* nrows x num_cols_offd; offd_i may be NULL when num_cols_offd == 0 */
    const double *offd_a;
    int num_cols_offd;
    const long long *col_map_offd;
} saamge_amd_parcsr;
int saamge_amd_ml_produce_data_parcsr(const saamge_amd_parcsr *A, int NE_local, int nde, const int *elem_to_dof,
                                      const double *elmat, const signed char *bdr_dofs, const int *const *partitions,
                                      const int *nparts_local, const saamge_amd_params *params, void *stream,
                                      saamge_amd_hierarchy **out);
/* adapt_update_operators(A, ml_data, mlp, resmooth_interp = true), src/adapt.cpp:188-219: the
 * matrix values changed (same sparsity and topology): every interpolation is kept -- no local
 * eigenproblem is solved again --, smoother diagonals, smoothed prolongators (nu_pro > 0), all
 * Galerkin operators and the coarsest solver are rebuilt.  new_val: nnz(A) values in the order
 * given at setup (host or device); NULL if the caller changed its device array in place. */
int saamge_amd_update_operators(saamge_amd_hierarchy *h, const double *new_val);
/* tg_update_coarse_operator(A, tg_data, perform_solve_init, coarse_direct), inc/tg.hpp:610-612: the same update with
 * the coarsest solver chosen again -- coarse_solver with the meaning of the params field of that name: 1 = the
 * reference's coarse_direct = true, 2 = its CG on the coarsest operator; -1 keeps the one the hierarchy was built with. */
int saamge_amd_update_operators2(saamge_amd_hierarchy *h, const double *new_val, int coarse_solver);
/* ml_free_data, inc/ml.hpp:196 */
void saamge_amd_ml_free_data(saamge_amd_hierarchy *h);

/* VCycleSolver::Mult with iterative_mode = false (inc/solve.hpp:129-143,
 * src/solve.cpp:309-323 -> tg_cycle_atb src/tg.cpp:91-132): x = B b */
int saamge_amd_vcycle_mult(saamge_amd_hierarchy *h, const double *b, double *x);
/* VCycleSolver::Mult for either iterative_mode (src/solve.cpp:309-323): iterative_mode != 0 keeps the
 * caller's x as the start vector, x <- x + B (b - A x) (the cycle is a stationary linear iteration, so this
 * equals tg_cycle_atb started from x). */
int saamge_amd_vcycle(saamge_amd_hierarchy *h, const double *b, double *x, int iterative_mode);
/* tg_data_t::coarse_solver (inc/tg_data.hpp:71; assigned by callers, e.g. test/algebraic/algebraic.cpp:282-283;
 * invoked as coarse_solver.Mult(RESC, XC) with XC pre-zeroed, src/tg.cpp:110-126): replaces the library's
 * coarsest solve by a host callback, xc = solve(rc) on `n` = coarsest dimension host doubles; return 0 on
 * success.  NULL restores the built-in solver. */
typedef int (*saamge_amd_coarse_solve_fn)(void *ctx, int n, const double *rc_host, double *xc_host);
int saamge_amd_set_coarse_solver(saamge_amd_hierarchy *h, saamge_amd_coarse_solve_fn fn, void *ctx);
/* The smoother plug of the cycle: typedef void (*smpr_ft)(HypreParMatrix& A, const Vector& b, Vector& x, void *data)
 * (inc/smpr.hpp:59-60), selected per tg_data_t through the TG options pre_smoother / post_smoother (inc/tg.hpp:99-119,
 * copied at src/tg.cpp:411-414) and called by tg_cycle_atb as pre_smoother(A, b, x, data) ... post_smoother(A, b, x, data)
 * (src/tg.cpp:113,131) with the semantics x += M^-1 (b - A x).  Here: host callbacks on `n` = rows of the level's
 * operator; the library copies b and x to the host, calls, and copies x back (pre-smoothing of a cycle from a zero
 * start vector hands over x = 0).  NULL for either restores the built-in polynomial smoother (smpr_sym_poly) in that
 * place.  A level that is row-partitioned over several ranks refuses a plug (error at the cycle).  Return 0 on success. */
typedef int (*saamge_amd_smoother_fn)(void *ctx, int level, int n, const double *b_host, double *x_host);
int saamge_amd_set_smoother(saamge_amd_hierarchy *h, int level, saamge_amd_smoother_fn pre, saamge_amd_smoother_fn post,
                            void *ctx);
/* smpr_sym_poly on one level (inc/smpr.hpp:59-60, src/smpr.cpp:213-234): x += M^-1 (b - A x) */
int saamge_amd_smoother(saamge_amd_hierarchy *h, int level, const double *b, double *x);
/* Outer Krylov loop: MFEM CGSolver as driven by test/mltest/mltest.cpp:773-781
 * (squared_tol = 1: stop when (B r,r) < max(rel_tol^2 (B r0,r0), abs_tol^2)) or
 * kalchev_pcg, inc/mfem_addons.hpp:276 (squared_tol = 0).  hist (host, max_iter+1
 * doubles, may be NULL) receives (B r_k, r_k).  zero_guess != 0 starts from x = 0. */
int saamge_amd_pcg(saamge_amd_hierarchy *h, const double *b, double *x, double rel_tol,
                   double abs_tol, int max_iter, int squared_tol, int zero_guess, int *iters,
                   int *converged, double *hist);

/* ---- inspection (ml_print_dims / tg_data_t field access, src/ml.cpp:296-355) ---- */
int saamge_amd_num_levels(const saamge_amd_hierarchy *h); /* number of operators = coarsenings + 1 */
/* info[0]=rows(A_l) [1]=nnz(A_l) [2]=nparts [3]=num_mises [4]=coarse dim [5]=nnz(P) [6]=nnz(Ac)
 * [7]=total eigenvectors [8]=inner PCG iterations of the last coarsest solve
 * [12]=1 if the level is row-partitioned [13]=first own row [14]=own rows [15]=halo entries received */
int saamge_amd_level_info(const saamge_amd_hierarchy *h, int level, long long info[16]);
/* Storage formats of the level operator's SELL-64 copy (no reference counterpart: hypre keeps CSR): info[0..2] = slices
 * that are pair-coded / offset-coded / plain, [3..5] = their stored entries, [6] = 256-row tiles whose x-segments are
 * staged through LDS, [7] = bytes of matrix data one application streams in these formats, [8] = pairs of the operator-level
 * (offset, value) dictionary that replaces the plain slices' columns and values by 16-bit codes (0: none), [9] = 1 if the
 * lanes of a 3 x 3 node block share their gathers of x, [10] = rows outside regular node blocks, [11] = local eigenproblems this rank SOLVED on the level (its other
 * agglomerates are bitwise identical to one of those and received a copy: saamge_amd_options.eig_dedupe). */
int saamge_amd_level_format(const saamge_amd_hierarchy *h, int level, long long info[12]);
/* which: 0 A_l, 1 interp, 2 restr, 3 Ac (host output buffers sized from level_info) */
int saamge_amd_get_csr(const saamge_amd_hierarchy *h, int level, int which, int *rowptr, int *col,
                       double *val);       /* fails on an operator with more than 2^31 - 1 entries */
int saamge_amd_get_csr64(const saamge_amd_hierarchy *h, int level, int which, long long *rowptr, int *col,
                         double *val);
/* which: 0 AE_to_dof, 1 dof_to_AE, 2 mis_to_dof, 3 mis_to_AE, 4 AE_to_mis, 5 elem_to_dof.
 * Pass I = J = NULL to query sizes: *nrows, *nconn. */
int saamge_amd_get_table(const saamge_amd_hierarchy *h, int level, int which, int *nrows,
                         long long *nconn, int *I, int *J);
/* per-dof MIS id (n), per-MIS kept vectors / SVD columns (num_mises), agg_flags (n) */
int saamge_amd_get_mis(const saamge_amd_hierarchy *h, int level, int *mises, int *mis_k,
                       int *mis_ncols, signed char *agg_flags);
/* keep_debug only: eigenvector counts per AE (nparts), packed eigenvalues / eigenvectors
 * (AE i: n_i x m_i column-major, AEs concatenated); sizes via level_info[7] and tables. */
int saamge_amd_get_ae_eigens(const saamge_amd_hierarchy *h, int level, int *ae_m, double *evals,
                             double *evecs, double *ae_D);
/* keep_debug only: packed per-MIS singular values (sum of SVD input columns before dropping) and
 * tentative blocks (MIS m: r_m x k_m column-major, concatenated). */
int saamge_amd_get_mis_svd(const saamge_amd_hierarchy *h, int level, long long *sig_off,
                           double *sig, double *U);

/* ---- operator-level entry points (the same kernels, usable on their own) ---- */
/* y = A x, hypre ParCSRMatrixMatvec as used by src/tg.cpp:115 */
int saamge_amd_spmv(int nrows, int ncols, const int *rowptr, const int *col, const double *val,
                    const double *x, double *y);
int saamge_amd_spmv64(int nrows, int ncols, const long long *rowptr, const int *col, const double *val,
                      const double *x, double *y);
/* xpacks_calc_lower_eigens_dense batched (src/xpacks.cpp:222-314) for A x = lambda D x with
 * diagonal D: matrices packed column-major one after the other, D packed likewise.
 * Outputs (host): m[i]; evals at offset sum_{j<i} n_j; evecs at offset sum_{j<i} n_j^2
 * (first m[i] columns), D-orthonormal. */
int saamge_amd_lower_eigens_batched(int count, const int *n, const double *A, const double *D,
                                    double vl, double vu, int *m, double *evals, double *evecs);

/* The certificate of the few-eigenpairs path on its own: neg[i] = number of eigenvalues of
 * A_i x = lambda D_i x below vu, from the inertia of D^-1/2 A D^-1/2 - vu I (banded L S L^T without
 * pivoting; Sylvester) -- what dsygvx obtains from dstebz's Sturm counts (src/xpacks.cpp:226-268).
 * -1: a pivot was too small for the count to be trusted (the setup then takes the dense path).
 * Same packing as saamge_amd_lower_eigens_batched. */
int saamge_amd_inertia_batched(int count, const int *n, const double *A, const double *D, double vu,
                               int *neg);

/* ---- device memory kept by the library between calls ----
 * Freed device blocks are cached and reused by later calls (hipMalloc / hipFree stall the host and, for
 * hipFree, the whole device); the eigensolver workspace is persistent.  saamge_amd_release_cached_memory()
 * returns all of it to the driver (call it with no hierarchy call in flight);
 * saamge_amd_cached_memory_bytes() reports the idle cached bytes (workspace not included).
 * Environment: SAAMGE_AMD_POOL_MAX_GB (default 64; 0 disables the cache). */
void saamge_amd_release_cached_memory(void);
long long saamge_amd_cached_memory_bytes(void);
/* device bytes the library holds right now (hierarchies, workspace in use; the caller's own arrays and the idle cache
 * are not counted) and their high-water mark since the last call with reset_peak != 0 */
void saamge_amd_memory_stats(long long *live_bytes, long long *peak_bytes, int reset_peak);
/* Requests of the library's cache of device blocks that went to the driver since the last reset: counts[0] = hipMalloc calls,
 * [1] = their bytes, [2] = hipFree of cached blocks, [3] = bytes idle in the cache now.  A steady-state setup makes none. */
void saamge_amd_pool_counts(long long counts[4], int reset);

/* ---- per-kernel timing for bench.py's roofline leg (HIP events around every launch) ---- */
void saamge_amd_profile_enable(int on);
void saamge_amd_profile_reset(void);
int saamge_amd_profile_count(void);
int saamge_amd_profile_get(int i, char *name, int name_len, double *ms, long long *launches,
                           double *bytes, double *flops);
/* The same, plus the bytes the launches had to move IN THE FORMAT THEY RAN (coded SELL slices + tables + vectors for the
 * SpMV family; equal to `bytes` elsewhere): the figure a roofline fraction is computed from.  The SpMV family is listed
 * per operator as "<name>@<rows>". */
int saamge_amd_profile_get2(int i, char *name, int name_len, double *ms, long long *launches, double *bytes, double *flops,
                            double *fmt_bytes);

#ifdef __cplusplus
}
#endif
#endif
