// The multilevel hierarchy: device-side counterpart of ml_data_t / levels_level_t /
// tg_data_t / interp_data_t (reference: amg/inc/ml.hpp:118-120, amg/inc/levels.hpp:47-64,
// amg/inc/tg_data.hpp:47-83, amg/inc/interp.hpp:54-100).
#pragma once
#include <exception>
#include <memory>
#include <thread>

#include "assemble.h"
#include "blocktri.h"
#include "common.h"
#include "eig.h"
#include "mis.h"
#include "sparse.h"
#include "topology.h"

namespace saamge_amd {

constexpr int MAX_LEVELS = 8;

struct Params {                 // MultilevelParameters (amg/inc/ml.hpp:59-114)
    int num_coarsenings = 1;
    double theta[MAX_LEVELS];
    int nu_relax[MAX_LEVELS];
    int nu_pro[MAX_LEVELS];     // prolongator smoothing degree (0 = tentative)
    int avoid_ess_bdr_dofs = 1; // amg/src/ml.cpp:64
    int testmesh = 0;           // mltest fixture: extra ones-vector on AE 0 (amg/src/interp.cpp:510-524)
    int coarse_solver = 0;      // 0 auto, 1 direct (dense inverse up to 16 384 rows, block-tridiagonal beyond), 2 inner PCG, 3 block-tridiagonal
    double coarse_rtol = 1e-14; // inner PCG: relative (B r, r) reduction, un-squared
    int coarse_max_iter = 2000;
    size_t workspace_bytes = (size_t)32 << 30;  // dense AE matrices are processed in chunks of this size
    int keep_debug = 0;         // keep per-AE eigenpairs / per-MIS data for parity tests
    // multi-GPU (one process per GPU): the AEs of every level are split into `world` contiguous
    // ranges; a rank solves the local eigenproblems of its range only and the results are
    // all-gathered in place through `allgather` (buf is a device pointer, rank r's part is
    // [byte_off[r], byte_off[r+1]) ).  Everything else is replicated.
    int rank = 0, world = 1;
    int (*allgather)(void *ctx, void *buf_dev, const long long *byte_off) = nullptr;
    void *allgather_ctx = nullptr;
    // Row-partitioned solve: levels with at least dist_min_local_rows
    // rows per rank are applied by row blocks -- every SpMV is preceded by a halo exchange of
    // the interface entries (`alltoallv`), dot products and the restricted residual are summed
    // with `allreduce_sum`, corrections come back through `allgather`.  Smaller levels and the
    // coarsest solve stay replicated.  Both callbacks get the same ctx as `allgather`.
    int (*allreduce_sum)(void *ctx, double *buf_dev, long long count) = nullptr;
    int (*alltoallv)(void *ctx, const void *send_dev, const long long *send_byte_off, void *recv_dev,
                     const long long *recv_byte_off) = nullptr;
    long long dist_min_local_rows = 262144;
    int comm_stream_ordered = 0;  // callbacks enqueue on the hierarchy's stream (no host sync needed)
    int correct_nullspace = 0;    // extra scaling_P level under the coarsest spectral operator
    const double *extra_modes = nullptr;  // level 0: n x num_extra_modes (column-major) appended to every MIS block
    int num_extra_modes = 0;
    double smooth_drop_tol = 0.0; // |entries| <= tol of the smoothed P are dropped (AltThreshold)
    int do_aggregates = 0;        // aggregates with arbitration instead of MISes on the last coarsening
    int algebraic = 0;            // element-free mode (tg_produce_data_algebraic): elements = dofs
    int eigensolver = 0;          // 0 few-eigenpairs path (certified count, dense fallback), 1 dense path only
    double eig_tol = 1e-12;       // few-eigenpairs path: acceptance bound of a Ritz pair's residual
};

struct NextPrep {               // host half of the next level's inputs (prepare_next_host)
    bool ready = false;
    Table e2d;                  // coarse elem_to_dof
    std::vector<int> colpos_ptr, colpos;        // host build ...
    DBuf<int> d_colpos_ptr, d_colpos;           // ... or the device build (coarse_e2d_device), already in place
    bool on_device = false;
};

struct Level {                  // tg_data_t + interp_data_t + agg_partitioning_relations_t
    DCsr A;                     // level operator (level 0: the user's matrix, viewed or copied)
    DCsr P, R, Ac;              // interp, restr, coarse operator
    DCsr Ptent;                 // tentative prolongator, kept apart only when nu_pro > 0 (P is then the smoothed one)
    DBuf<double> dinv_neg;      // smpr_poly_data_t::Dinv_neg
    std::vector<double> roots;  // smpr_poly_data_t::roots (SAS)
    double theta = 0.0;
    int nu_relax = 3;
    Relations rel;              // host topology
    bool rel_prebuilt = false;  // rel was built ahead, beside the coarse element matrices of the level above (prepare_next_level)
    NextPrep next_prep;         // filled beside the Galerkin product, consumed by prepare_next_level
    DevRelations drel;          // device mirror
    DevElmats elmat;            // element matrices of this level
    // interp_data_t
    std::vector<int> ae_begin;          // [world+1] AE ownership ranges of the ranks (eigenproblems, coarse element matrices)
    std::vector<int> ae_m;              // eigenvectors per AE
    std::vector<int> ae_class;          // per agglomerate: its class of identical SPARSE ROWS (the eigenproblem stage's, fused fine-level assembly), or -1
    std::vector<int> ae_evclass;        // per agglomerate: the class whose eigenpairs it holds a copy of (any kind of class), or -1
    long long ae_solved = 0;            // local eigenproblems actually solved on this level by this rank (the others: copies of a class)
    std::vector<int64_t> ae_xoff, ae_eoff;
    DBuf<double> evals, evecs;          // cut_evects_arr (packed)
    DBuf<double> ae_D;                  // rhs_matrices_arr (diagonals, packed per AE rows), debug only
    std::vector<int> mis_k;             // mis_numcoarsedof
    std::vector<int> mis_coloff;        // mis_coarsedofoffsets
    std::vector<int> mis_ncols;         // columns that entered each SVD (debug)
    std::vector<int64_t> mis_u_off, mis_s_off;
    DBuf<double> mis_U, mis_sig;        // mis_tent_interps (packed r x k, column-major), singular values
    DBuf<int> d_mis_k, d_mis_coloff;
    DBuf<int64_t> d_mis_u_off;
    // solve-phase work vectors
    DBuf<double> x, b, r, t0, t1;
    DBuf<double> extra;                 // level 0: device copy / view of Params::extra_modes
    // R (R ... 1): the NEXT level's representation of the constant vector -- first start vector of that level's
    // few-eigenpairs iteration and the candidate of its known-null-vector shortcut (eig.h: EigBatch::x0c)
    DBuf<double> cvec_next;
    // row-partitioned solve: own rows [row_off[rank], row_off[rank+1]) (multiples of 64), halo
    // exchange lists of A's input vector (global indices, grouped by peer rank)
    struct Dist {
        bool on = false;
        int row0 = 0, nloc = 0;
        std::vector<int> row_off;
        std::vector<long long> send_off, recv_off;  // byte offsets into send_buf / recv_buf, world+1
        std::vector<long long> own_off;             // byte offsets of the ownership ranges (all-gather)
        DBuf<int> send_idx, recv_idx;
        DBuf<double> send_buf, recv_buf;
        DBuf<double> rs_buf;                        // reduce-scatter of the restricted residual: world slabs of nloc
        std::vector<long long> rs_off;
        int nsend = 0, nrecv = 0;
        // longest run of own SELL slices whose rows read no halo entry: applied on a side stream while the
        // exchange is in flight (dist.hip: halo_then); int_nrows = 0: no overlap on this level
        int int_row0 = 0, int_nrows = 0;
    } dist;
};

struct KernelTiming { double setup_ms = 0, solve_ms = 0; };

// Per-rank inputs (dist_input.hip): a rank's row block in hypre's ParCSR split, as a HypreParMatrix holds it
struct ParCsrIn {
    long long global_rows = 0;              // 0: not given (the sum of the ranks' rows)
    const long long *row_starts = nullptr;  // world + 1 entries or null: contiguous row blocks in rank order
    int nrows = 0;
    const int *diag_i = nullptr, *diag_j = nullptr;      // nrows x nrows, local column indices
    const double *diag_a = nullptr;
    const int *offd_i = nullptr, *offd_j = nullptr;      // nrows x num_cols_offd (null: no off-diagonal block)
    const double *offd_a = nullptr;
    int num_cols_offd = 0;
    const long long *col_map_offd = nullptr;             // global column of every offd column
};
// ... gathered into the replicated form the setup works on; owned by the hierarchy
struct DistIn {
    DBuf<roff_t> rowptr;
    DBuf<int> col, e2d;
    DBuf<double> val;
    DBuf<signed char> bdr;
    std::vector<DBuf<int>> parts;
    std::vector<int> nparts;
    int elem0 = 0, NE_loc = 0;                            // this rank's elements: [elem0, elem0 + NE_loc); only THEIR matrices exist here
    std::vector<std::vector<long long>> ae_begin;         // per level: agglomerate ownership ranges, dictated by the inputs
};

struct Hierarchy {              // ml_data_t
    Params params;
    hipStream_t stream = 0;
    int device = 0;             // the GPU this hierarchy lives on (current device of the creating thread)
    std::vector<std::unique_ptr<Level>> levels;
    // coarsest solver
    int coarse_kind = 2;        // 1 explicit dense inverse, 2 inner PCG, 3 block-tridiagonal direct solve (blocktri.hip)
    BlockTri c_bt;
    DBuf<double> c_dinv, c_r, c_z, c_d, c_q, c_t0, c_t1, c_b, c_x;
    DBuf<double> c_L, c_work;   // explicit inverse of the coarsest operator (coarse_kind 1)
    std::vector<double> c_roots;
    // PCG scratch
    DBuf<double> pcg_r, pcg_z, pcg_d, pcg_q, scal, partials;
    int last_coarse_iters = 0;
    // tg_data_t::coarse_solver plug: host callback replacing the built-in coarsest solve
    int (*user_coarse_solve)(void *ctx, int n, const double *rc_host, double *xc_host) = nullptr;
    void *user_coarse_ctx = nullptr;
    // smoother plug per level (saamge_amd_set_smoother): host callbacks, x += M^-1 (b - A x)
    struct UserSmoother {
        int (*pre)(void *ctx, int level, int n, const double *b_host, double *x_host) = nullptr;
        int (*post)(void *ctx, int level, int n, const double *b_host, double *x_host) = nullptr;
        void *ctx = nullptr;
    };
    std::vector<UserSmoother> user_smoothers;
    DBuf<int> own_e2d;          // element-free mode: the generated identity elem_to_dof
    // host copies of the coarse partitions (levels >= 1: a few thousand ints), fetched once at the start of the setup: the host
    // build of a coarse level's AE tables starts while the GPU still computes that level's element matrices
    std::vector<hvec<int>> coarse_parts;
    std::vector<int> nparts_in;
    std::unique_ptr<DistIn> dist_in;      // per-rank inputs: the gathered operator / topology the level-0 arrays view
    // setup only: the Galerkin product of level `galerkin_lev` runs on its own thread and stream beside the
    // next level's element matrices and eigenproblems (which need the level's size, not its operator)
    std::thread galerkin_thread;
    std::exception_ptr galerkin_err;
    int galerkin_lev = -1;
    // row-partitioned solve: fork / join events of the interior-rows stream (halo_then)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    ~Hierarchy() {
        if (galerkin_thread.joinable()) galerkin_thread.join();
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
};

// ml_produce_data (amg/src/ml.cpp:379-472).  All array arguments may be host or device
// pointers.  partitions[k] maps level-k elements to level-k AEs.
Hierarchy *hierarchy_create(int n, const void *Arow, int rowptr_bits, const int *Acol, const double *Aval, int NE,
                            int nde, const int *elem_to_dof, const double *elmat,
                            const signed char *bdr, const int *const *partitions,
                            const int *nparts, const Params &p, hipStream_t stream, std::unique_ptr<DistIn> din = nullptr);
// the same from per-rank inputs (dist_input.hip): elem_to_dof holds GLOBAL dof ids, elmat the matrices of the rank's own
// elements, bdr_own the flags of the rank's own rows, partitions[l] maps the rank's level-l elements (level 0: its
// elements; above: its agglomerates of the level below) to its own agglomerates 0 .. nparts_loc[l] - 1
Hierarchy *hierarchy_create_dist(const ParCsrIn &A, int NE_loc, int nde, const int *elem_to_dof, const double *elmat,
                                 const signed char *bdr_own, const int *const *partitions, const int *nparts_loc,
                                 const Params &p, hipStream_t stream);

// adapt_update_operators (amg/src/adapt.cpp:171-219): new matrix values (same pattern; host or
// device pointer, nullptr = the level-0 values were changed in place), all interpolations kept.
void hierarchy_update_operators(Hierarchy &h, const double *new_val);

// VCycleSolver::Mult with iterative_mode = false (amg/src/solve.cpp:309-323): x = B b.
void vcycle_apply(Hierarchy &h, int level, const double *b, double *x);
// smpr_sym_poly (amg/src/smpr.cpp:213-234): x += M^-1 (b - A x)
void smoother_apply(Hierarchy &h, int level, const double *b, double *x);
// MFEM CGSolver / kalchev_pcg (amg/src/mfem_addons.cpp:106-248).  b, x device pointers.
// Returns iterations; hist (host, may be null) receives (B r_k, r_k), k = 0..iters.
int pcg_solve(Hierarchy &h, const double *b, double *x, double rel_tol, double abs_tol,
              int max_iter, int squared_tol, int zero_guess, int *converged, double *hist);

}  // namespace saamge_amd
