// extern "C" boundary (include/saamge_amd.h).  Exceptions stop here.
#include <cstdlib>
#include "../../include/saamge_amd.h"

#include <string>

#include "hierarchy.h"

using namespace saamge_amd;

struct saamge_amd_hierarchy {
    Hierarchy *H;
};

static std::string g_last_error;

#define SA_API_BEGIN try {
#define SA_API_END                                   \
    }                                                \
    catch (const Error &e) {                         \
        g_last_error = e.what();                     \
        return e.code ? e.code : 1;                  \
    }                                                \
    catch (const std::exception &e) {                \
        g_last_error = e.what();                     \
        return 1;                                    \
    }                                                \
    return 0;

// every call on a hierarchy must come from a thread whose current HIP device is the one the
// hierarchy was built on (HIP's current device is per host thread)
static void require_device(const Hierarchy &H) {
    SA_REQUIRE(current_device() == H.device, "the calling thread's current HIP device is not the hierarchy's device");
    set_thread_stream(H.stream);     // device blocks freed by this call are ordered after the hierarchy's stream
}

extern "C" int saamge_amd_comm_native_stream(const saamge_amd_params *p, void **stream);      // comm.hip

extern "C" {

const char *saamge_amd_last_error(void) { return g_last_error.c_str(); }

static void options_to_c(const Options &o, saamge_amd_options *c) {
    c->eig_strict = o.eig_strict; c->eig_certify = o.eig_certify; c->eig_min_n = o.eig_min_n;
    c->eig_force_fallback = o.eig_force_fallback; c->eig_dense_only = o.eig_dense_only; c->eig_dense_one_stage = o.eig_dense_one_stage;
    c->eig_nullcheck = o.eig_nullcheck; c->eig_keep_inertia_factor = o.eig_keep_inertia_factor; c->band_assembly = o.band_assembly;
    c->eig_dedupe = o.eig_dedupe; c->eig_outer_panels = o.eig_outer_panels; c->overlap = o.overlap; c->sell = o.sell; c->spmv_sell = o.spmv_sell; c->debug = o.debug;
    c->host_heap_pad_mb = o.host_heap_pad_mb;
}
void saamge_amd_options_default(saamge_amd_options *o) { options_to_c(Options(), o); }
void saamge_amd_get_options(saamge_amd_options *o) { options_to_c(options(), o); }
void saamge_amd_set_options(const saamge_amd_options *c) {
    Options &o = options();
    o.eig_strict = c->eig_strict; o.eig_certify = c->eig_certify; o.eig_min_n = c->eig_min_n;
    o.eig_force_fallback = c->eig_force_fallback; o.eig_dense_only = c->eig_dense_only; o.eig_dense_one_stage = c->eig_dense_one_stage;
    o.eig_nullcheck = c->eig_nullcheck; o.eig_keep_inertia_factor = c->eig_keep_inertia_factor; o.band_assembly = c->band_assembly;
    o.eig_dedupe = c->eig_dedupe; o.eig_outer_panels = c->eig_outer_panels; o.overlap = c->overlap; o.sell = c->sell; o.spmv_sell = c->spmv_sell; o.debug = c->debug;
    o.host_heap_pad_mb = c->host_heap_pad_mb;
}

void saamge_amd_params_default(saamge_amd_params *p) {
    // defaults of test/mltest/mltest.cpp:332-419
    p->num_coarsenings = 1;
    for (int i = 0; i < SAAMGE_AMD_MAX_LEVELS; ++i) {
        p->theta[i] = 0.003;
        p->nu_relax[i] = 3;
        p->nu_pro[i] = 0;
    }
    p->avoid_ess_bdr_dofs = 1;
    p->testmesh = 0;
    p->coarse_solver = 0;
    p->coarse_rtol = 1e-14;
    p->coarse_max_iter = 2000;
    p->workspace_bytes = (long long)32 << 30;
    p->keep_debug = 0;
    p->rank = 0;
    p->world = 1;
    p->allgather = nullptr;
    p->allgather_ctx = nullptr;
    p->allreduce_sum = nullptr;
    p->alltoallv = nullptr;
    p->dist_min_local_rows = 262144;
    p->comm_stream_ordered = 0;
    p->correct_nullspace = 0;
    p->extra_modes = nullptr;
    p->num_extra_modes = 0;
    p->algebraic = 0;
    p->smooth_drop_tol = 0.0;
    p->do_aggregates = 0;
    p->eigensolver = 0;
    p->eig_tol = 1e-12;
    saamge_amd_options_default(&p->options);
}

int saamge_amd_memcpy(void *dst, const void *src, long long bytes) {
    if (bytes <= 0) return 0;
    if (hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDefault) != hipSuccess) {
        g_last_error = "saamge_amd_memcpy failed";
        (void)hipGetLastError();
        return 2;
    }
    return 0;
}

static Params convert_params(const saamge_amd_params *params, void *stream) {
    saamge_amd_set_options(&params->options);
    host_heap_policy();
    Params p;
    p.num_coarsenings = params->num_coarsenings;
    SA_REQUIRE(p.num_coarsenings >= 1 && p.num_coarsenings < MAX_LEVELS, "bad num_coarsenings");
    for (int i = 0; i < MAX_LEVELS; ++i) {
        p.theta[i] = params->theta[i];
        p.nu_relax[i] = params->nu_relax[i];
        p.nu_pro[i] = params->nu_pro[i];
    }
    p.avoid_ess_bdr_dofs = params->avoid_ess_bdr_dofs;
    p.testmesh = params->testmesh;
    p.coarse_solver = params->coarse_solver;
    p.coarse_rtol = params->coarse_rtol;
    p.coarse_max_iter = params->coarse_max_iter;
    p.workspace_bytes = (size_t)params->workspace_bytes;
    p.keep_debug = params->keep_debug;
    p.rank = params->rank;
    p.world = params->world > 1 ? params->world : 1;
    p.allgather = params->allgather;
    p.allgather_ctx = params->allgather_ctx;
    p.allreduce_sum = params->allreduce_sum;
    p.alltoallv = params->alltoallv;
    p.dist_min_local_rows = params->dist_min_local_rows;
    p.comm_stream_ordered = params->comm_stream_ordered;
    {   // a native communicator enqueues its collectives on the stream it was created with: only on the hierarchy's own
        // stream are they ordered with the hierarchy's kernels (round-2 advisor finding: nothing checked this on the C side)
        void *cs = nullptr;
        SA_REQUIRE(!saamge_amd_comm_native_stream(params, &cs) || cs == stream,
                   "the communicator was created on another stream than the one given to saamge_amd_ml_produce_data: "
                   "create it with saamge_amd_comm_create(..., stream, ...) on the hierarchy's stream");
    }
    p.correct_nullspace = params->correct_nullspace;
    p.extra_modes = params->extra_modes;
    p.num_extra_modes = params->num_extra_modes;
    p.algebraic = params->algebraic;
    p.smooth_drop_tol = params->smooth_drop_tol;
    p.do_aggregates = params->do_aggregates;
    p.eigensolver = params->eigensolver;
    SA_REQUIRE(p.eigensolver == 0 || p.eigensolver == 1, "bad eigensolver selector");
    p.eig_tol = params->eig_tol;
    SA_REQUIRE(p.eig_tol >= 1e-15 && p.eig_tol <= 1e-8, "eig_tol must lie in [1e-15, 1e-8] (saamge_amd_params_default sets 1e-12)");
    SA_REQUIRE(p.world == 1 || (p.rank >= 0 && p.rank < p.world), "bad rank");
    return p;
}

static int produce_data(int n, const void *rowptr, int rowptr_bits, const int *col, const double *val,
                        int NE, int nde, const int *elem_to_dof, const double *elmat,
                        const signed char *bdr_dofs, const int *const *partitions,
                        const int *nparts, const saamge_amd_params *params, void *stream,
                        saamge_amd_hierarchy **out) {
    SA_API_BEGIN
    SA_REQUIRE(out && params && rowptr && col && val && (params->algebraic || (elem_to_dof && elmat)) && partitions && nparts,
               "null argument");
    const Params p = convert_params(params, stream);
    set_thread_stream((hipStream_t)stream);
    Hierarchy *H = hierarchy_create(n, rowptr, rowptr_bits, col, val, NE, nde, elem_to_dof, elmat, bdr_dofs,
                                    partitions, nparts, p, (hipStream_t)stream);
    *out = new saamge_amd_hierarchy{H};
    SA_API_END
}

int saamge_amd_ml_produce_data_parcsr(const saamge_amd_parcsr *A, int NE_local, int nde, const int *elem_to_dof,
                                      const double *elmat, const signed char *bdr_dofs, const int *const *partitions,
                                      const int *nparts_local, const saamge_amd_params *params, void *stream,
                                      saamge_amd_hierarchy **out) {
    SA_API_BEGIN
    SA_REQUIRE(out && params && A && A->diag_i && (A->nrows == 0 || (A->diag_j && A->diag_a)) && partitions && nparts_local &&
                   (NE_local == 0 || (elem_to_dof && elmat)), "null argument");
    SA_REQUIRE(!A->offd_i || A->num_cols_offd == 0 || (A->offd_j && A->offd_a && A->col_map_offd), "offd block without its arrays");
    const Params p = convert_params(params, stream);
    ParCsrIn in;
    in.global_rows = A->global_rows;
    in.row_starts = A->row_starts;
    in.nrows = A->nrows;
    in.diag_i = A->diag_i; in.diag_j = A->diag_j; in.diag_a = A->diag_a;
    in.offd_i = (A->offd_i && A->num_cols_offd > 0) ? A->offd_i : nullptr;
    in.offd_j = A->offd_j; in.offd_a = A->offd_a;
    in.num_cols_offd = A->num_cols_offd;
    in.col_map_offd = A->col_map_offd;
    set_thread_stream((hipStream_t)stream);
    Hierarchy *H = hierarchy_create_dist(in, NE_local, nde, elem_to_dof, elmat, bdr_dofs, partitions, nparts_local, p, (hipStream_t)stream);
    *out = new saamge_amd_hierarchy{H};
    SA_API_END
}

int saamge_amd_ml_produce_data(int n, const int *rowptr, const int *col, const double *val,
                               int NE, int nde, const int *elem_to_dof, const double *elmat,
                               const signed char *bdr_dofs, const int *const *partitions,
                               const int *nparts, const saamge_amd_params *params, void *stream,
                               saamge_amd_hierarchy **out) {
    return produce_data(n, rowptr, 32, col, val, NE, nde, elem_to_dof, elmat, bdr_dofs, partitions, nparts, params, stream, out);
}

int saamge_amd_ml_produce_data64(int n, const long long *rowptr, const int *col, const double *val,
                                 int NE, int nde, const int *elem_to_dof, const double *elmat,
                                 const signed char *bdr_dofs, const int *const *partitions,
                                 const int *nparts, const saamge_amd_params *params, void *stream,
                                 saamge_amd_hierarchy **out) {
    return produce_data(n, rowptr, 64, col, val, NE, nde, elem_to_dof, elmat, bdr_dofs, partitions, nparts, params, stream, out);
}

int saamge_amd_update_operators(saamge_amd_hierarchy *h, const double *new_val) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    require_device(*h->H);
    hierarchy_update_operators(*h->H, new_val);
    SA_API_END
}

int saamge_amd_update_operators2(saamge_amd_hierarchy *h, const double *new_val, int coarse_solver) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    SA_REQUIRE(coarse_solver >= -1 && coarse_solver <= 3, "coarse_solver: -1 (keep), 0 (auto), 1 (direct), 2 (inner PCG) or 3 (block-tridiagonal direct)");
    require_device(*h->H);
    if (coarse_solver >= 0) h->H->params.coarse_solver = coarse_solver;
    hierarchy_update_operators(*h->H, new_val);
    SA_API_END
}

void saamge_amd_ml_free_data(saamge_amd_hierarchy *h) {
    if (!h) return;
    hipStream_t hs = nullptr;
    const bool had = h->H != nullptr;
    if (had) {
        hs = h->H->stream;
        set_thread_stream(hs);
    }
    delete h->H;
    delete h;
    // the caller may destroy its stream right after this call: no batch of frees stays open on it
    if (had) {
        dev_pool_close_stream(hs);
        unset_thread_stream();      // ... and this thread no longer frees into it
    }
}

// stage a host vector on the device when needed
struct VecIn {
    DBuf<double> buf;
    const double *p;
    VecIn(const double *src, size_t n, hipStream_t s) {
        if (is_device_ptr(src)) p = src;
        else { buf.assign(src, n, s); p = buf.p; }
    }
};
struct VecOut {
    DBuf<double> buf;
    double *p, *host = nullptr;
    size_t n;
    hipStream_t s;
    VecOut(double *dst, size_t n_, hipStream_t s_, bool load) : n(n_), s(s_) {
        if (is_device_ptr(dst)) p = dst;
        else {
            host = dst;
            if (load) buf.assign(dst, n, s); else buf.alloc(n);
            p = buf.p;
        }
    }
    void finish() {
        if (host && n) SA_HIP_CHECK(hipMemcpyAsync(host, p, 8 * n, hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
};

int saamge_amd_vcycle_mult(saamge_amd_hierarchy *h, const double *b, double *x) {
    SA_API_BEGIN
    SA_REQUIRE(h && b && x, "null argument");
    Hierarchy &H = *h->H;
    require_device(H);
    const size_t n = (size_t)H.levels[0]->A.nrows;
    VecIn vb(b, n, H.stream);
    VecOut vx(x, n, H.stream, false);
    vcycle_apply(H, 0, vb.p, vx.p);
    vx.finish();
    SA_API_END
}

int saamge_amd_vcycle(saamge_amd_hierarchy *h, const double *b, double *x, int iterative_mode) {
    if (!iterative_mode) return saamge_amd_vcycle_mult(h, b, x);
    SA_API_BEGIN
    SA_REQUIRE(h && b && x, "null argument");
    Hierarchy &H = *h->H;
    require_device(H);
    Level &L0 = *H.levels[0];
    SA_REQUIRE(!L0.dist.on, "iterative_mode is not available on a row-partitioned hierarchy");
    const size_t n = (size_t)L0.A.nrows;
    VecIn vb(b, n, H.stream);
    VecOut vx(x, n, H.stream, true);
    // x <- x + B (b - A x)
    spmv_residual(H.stream, L0.A, vx.p, vb.p, H.pcg_r.p);
    vcycle_apply(H, 0, H.pcg_r.p, H.pcg_z.p);
    vec_axpy(H.stream, (int)n, 1.0, H.pcg_z.p, vx.p);
    vx.finish();
    SA_API_END
}

int saamge_amd_set_coarse_solver(saamge_amd_hierarchy *h, saamge_amd_coarse_solve_fn fn, void *ctx) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    h->H->user_coarse_solve = fn;
    h->H->user_coarse_ctx = ctx;
    SA_API_END
}

int saamge_amd_set_smoother(saamge_amd_hierarchy *h, int level, saamge_amd_smoother_fn pre, saamge_amd_smoother_fn post,
                            void *ctx) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    Hierarchy &H = *h->H;
    SA_REQUIRE(level >= 0 && level < (int)H.levels.size(), "bad level");
    if (H.user_smoothers.size() < H.levels.size()) H.user_smoothers.resize(H.levels.size());
    H.user_smoothers[(size_t)level].pre = pre;
    H.user_smoothers[(size_t)level].post = post;
    H.user_smoothers[(size_t)level].ctx = ctx;
    SA_API_END
}

int saamge_amd_smoother(saamge_amd_hierarchy *h, int level, const double *b, double *x) {
    SA_API_BEGIN
    SA_REQUIRE(h && b && x, "null argument");
    Hierarchy &H = *h->H;
    require_device(H);
    SA_REQUIRE(level >= 0 && level < (int)H.levels.size(), "bad level");
    const size_t n = (size_t)H.levels[level]->A.nrows;
    VecIn vb(b, n, H.stream);
    VecOut vx(x, n, H.stream, true);
    smoother_apply(H, level, vb.p, vx.p);
    vx.finish();
    SA_API_END
}

int saamge_amd_pcg(saamge_amd_hierarchy *h, const double *b, double *x, double rel_tol,
                   double abs_tol, int max_iter, int squared_tol, int zero_guess, int *iters,
                   int *converged, double *hist) {
    SA_API_BEGIN
    SA_REQUIRE(h && b && x && iters, "null argument");
    Hierarchy &H = *h->H;
    require_device(H);
    const size_t n = (size_t)H.levels[0]->A.nrows;
    VecIn vb(b, n, H.stream);
    VecOut vx(x, n, H.stream, !zero_guess);
    int conv = 0;
    *iters = pcg_solve(H, vb.p, vx.p, rel_tol, abs_tol, max_iter, squared_tol, zero_guess, &conv, hist);
    if (converged) *converged = conv;
    vx.finish();
    SA_API_END
}

int saamge_amd_num_levels(const saamge_amd_hierarchy *h) { return h ? (int)h->H->levels.size() + 1 : 0; }

static const DCsr &level_op(const Hierarchy &H, int level, int which) {
    const int nl = (int)H.levels.size();
    SA_REQUIRE(level >= 0 && level < nl, "bad level");
    const Level &L = *H.levels[level];
    switch (which) {
        case 0: return L.A;
        case 1: return L.P;
        case 2: return L.R;
        case 3: return (level + 1 < nl) ? H.levels[level + 1]->A : L.Ac;
    }
    throw Error(1, "bad operator selector");
}

int saamge_amd_level_info(const saamge_amd_hierarchy *h, int level, long long info[16]) {
    SA_API_BEGIN
    SA_REQUIRE(h && info, "null argument");
    const Hierarchy &H = *h->H;
    for (int i = 0; i < 16; ++i) info[i] = 0;
    const Level &L = *H.levels.at(level);
    info[0] = L.A.nrows;
    info[1] = L.A.nnz;
    info[2] = L.rel.nparts;
    info[3] = L.rel.num_mises;
    info[4] = L.P.ncols;
    info[5] = L.P.nnz;
    info[6] = level_op(H, level, 3).nnz;
    long long tot = 0;
    for (int m : L.ae_m) tot += m;
    info[7] = tot;
    info[8] = H.last_coarse_iters;
    info[9] = L.ae_xoff.empty() ? 0 : L.ae_xoff.back();
    info[10] = L.mis_s_off.empty() ? 0 : L.mis_s_off.back();
    long long usz = 0;
    for (int m = 0; m < L.rel.num_mises; ++m) usz += (long long)L.mis_k[m] * L.rel.mis_to_dof.row_size(m);
    info[11] = usz;
    info[12] = L.dist.on ? 1 : 0;
    info[13] = L.dist.row0;
    info[14] = L.dist.nloc;
    info[15] = L.dist.nrecv;
    SA_API_END
}

int saamge_amd_level_format(const saamge_amd_hierarchy *h, int level, long long info[12]) {
    SA_API_BEGIN
    SA_REQUIRE(h && info, "null argument");
    const Hierarchy &H = *h->H;
    const DCsr &A = H.levels.at(level)->A;
    for (int i = 0; i < 12; ++i) info[i] = 0;
    for (int c = 0; c < 3; ++c) { info[c] = A.sell_class_slices[c]; info[3 + c] = A.sell_class_entries[c]; }
    if (A.sell_gpair) { info[8] = A.sell_ng; info[9] = A.sell_bs3 ? 1 : 0; info[10] = A.sell_bs3 ? A.sell_nirr : 0; }
    info[6] = A.sell_stage_cap > 0 ? (long long)div_up(A.nslices, 4) - A.sell_nunstaged : 0;
    info[7] = (long long)A.sell_stream_bytes;
    info[11] = H.levels.at(level)->ae_solved;
    SA_API_END
}

static int get_csr(const saamge_amd_hierarchy *h, int level, int which, void *rowptr, int rowptr_bits, int *col, double *val) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    const Hierarchy &H = *h->H;
    require_device(H);
    const DCsr &M = level_op(H, level, which);
    hipStream_t s = H.stream;
    if (rowptr && rowptr_bits == 64)
        SA_HIP_CHECK(hipMemcpyAsync(rowptr, M.rowptr.p, sizeof(roff_t) * ((size_t)M.nrows + 1), hipMemcpyDeviceToHost, s));
    else if (rowptr)
        export_rowptr32((int *)rowptr, M.rowptr, (size_t)M.nrows + 1, s);
    if (col && M.nnz) SA_HIP_CHECK(hipMemcpyAsync(col, M.col.p, 4 * (size_t)M.nnz, hipMemcpyDeviceToHost, s));
    if (val && M.nnz) SA_HIP_CHECK(hipMemcpyAsync(val, M.val.p, 8 * (size_t)M.nnz, hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_API_END
}
int saamge_amd_get_csr(const saamge_amd_hierarchy *h, int level, int which, int *rowptr, int *col, double *val) {
    return get_csr(h, level, which, rowptr, 32, col, val);
}
int saamge_amd_get_csr64(const saamge_amd_hierarchy *h, int level, int which, long long *rowptr, int *col, double *val) {
    return get_csr(h, level, which, rowptr, 64, col, val);
}

int saamge_amd_get_table(const saamge_amd_hierarchy *h, int level, int which, int *nrows,
                         long long *nconn, int *I, int *J) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    require_device(*h->H);
    {   // tables kept on the device by the device topology build: host copies on demand
        Level &Lw = *h->H->levels.at(level);
        fetch_relations_ae_host(Lw.rel, Lw.drel, h->H->stream);
    }
    const Relations &r = h->H->levels.at(level)->rel;
    const Table *T = nullptr;
    switch (which) {
        case 0: T = &r.AE_to_dof; break;
        case 1: T = &r.dof_to_AE; break;
        case 2: T = &r.mis_to_dof; break;
        case 3: T = &r.mis_to_AE; break;
        case 4: T = &r.AE_to_mis; break;
        case 5: {
            if (r.elem_to_dof.I.empty()) {   // built on the device: fetch the host copy on demand
                Level &L = *h->H->levels.at(level);
                Relations &rw = L.rel;
                rw.elem_to_dof.I = L.drel.e2d_I.to_host(h->H->stream);
                rw.elem_to_dof.J = L.drel.e2d_J.to_host(h->H->stream);
                rw.elem_to_dof.ncols = rw.ND;
            }
            T = &r.elem_to_dof;
            break;
        }
        default: throw Error(1, "bad table selector");
    }
    if (nrows) *nrows = T->nrows();
    if (nconn) *nconn = (long long)T->J.size();
    if (I) std::copy(T->I.begin(), T->I.end(), I);
    if (J) std::copy(T->J.begin(), T->J.end(), J);
    SA_API_END
}

int saamge_amd_get_mis(const saamge_amd_hierarchy *h, int level, int *mises, int *mis_k,
                       int *mis_ncols, signed char *agg_flags) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    require_device(*h->H);
    {
        Level &Lw = *h->H->levels.at(level);
        fetch_relations_ae_host(Lw.rel, Lw.drel, h->H->stream);
    }
    const Level &L = *h->H->levels.at(level);
    if (mises) std::copy(L.rel.mises.begin(), L.rel.mises.end(), mises);
    if (mis_k) std::copy(L.mis_k.begin(), L.mis_k.end(), mis_k);
    if (mis_ncols) std::copy(L.mis_ncols.begin(), L.mis_ncols.end(), mis_ncols);
    if (agg_flags) std::copy(L.rel.agg_flags.begin(), L.rel.agg_flags.end(), agg_flags);
    SA_API_END
}

int saamge_amd_get_ae_eigens(const saamge_amd_hierarchy *h, int level, int *ae_m, double *evals,
                             double *evecs, double *ae_D) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    const Hierarchy &H = *h->H;
    const Level &L = *H.levels.at(level);
    hipStream_t s = H.stream;
    if (ae_m) std::copy(L.ae_m.begin(), L.ae_m.end(), ae_m);
    if (evals || evecs || ae_D) SA_REQUIRE(H.params.keep_debug, "hierarchy was built without keep_debug");
    if (evals && L.evals.n) SA_HIP_CHECK(hipMemcpyAsync(evals, L.evals.p, 8 * L.evals.n, hipMemcpyDeviceToHost, s));
    if (evecs && L.evecs.n) SA_HIP_CHECK(hipMemcpyAsync(evecs, L.evecs.p, 8 * L.evecs.n, hipMemcpyDeviceToHost, s));
    if (ae_D && L.ae_D.n) SA_HIP_CHECK(hipMemcpyAsync(ae_D, L.ae_D.p, 8 * L.ae_D.n, hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_API_END
}

int saamge_amd_get_mis_svd(const saamge_amd_hierarchy *h, int level, long long *sig_off,
                           double *sig, double *U) {
    SA_API_BEGIN
    SA_REQUIRE(h, "null argument");
    const Hierarchy &H = *h->H;
    const Level &L = *H.levels.at(level);
    hipStream_t s = H.stream;
    const int nm = L.rel.num_mises;
    if (sig_off) for (int m = 0; m <= nm; ++m) sig_off[m] = L.mis_s_off[m];
    if (sig) {
        SA_REQUIRE(H.params.keep_debug, "hierarchy was built without keep_debug");
        if (L.mis_s_off[nm]) SA_HIP_CHECK(hipMemcpyAsync(sig, L.mis_sig.p, 8 * (size_t)L.mis_s_off[nm], hipMemcpyDeviceToHost, s));
    }
    if (U) {
        auto all = L.mis_U.to_host(s);
        size_t o = 0;
        for (int m = 0; m < nm; ++m) {
            const size_t cnt = (size_t)L.mis_k[m] * L.rel.mis_to_dof.row_size(m);
            std::copy(all.begin() + L.mis_u_off[m], all.begin() + L.mis_u_off[m] + cnt, U + o);
            o += cnt;
        }
    }
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_API_END
}

static int spmv_entry(int nrows, int ncols, const void *rowptr, int rowptr_bits, const int *col, const double *val,
                      const double *x, double *y) {
    SA_API_BEGIN
    SA_REQUIRE(rowptr && col && val && x && y && nrows >= 0, "bad argument");
    hipStream_t s = 0;
    set_thread_stream(s);
    DCsr A;
    A.nrows = nrows;
    A.ncols = ncols;
    import_rowptr(A.rowptr, rowptr, rowptr_bits, (size_t)nrows + 1, s);
    roff_t nnz = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&nnz, A.rowptr.p + nrows, sizeof(roff_t), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    A.nnz = nnz;
    import_array(A.col, col, (size_t)A.nnz, s);
    import_array(A.val, val, (size_t)A.nnz, s);
    A.lanes_per_row = pick_lanes_per_row(A.nnz, nrows > 0 ? nrows : 1);
    // SAAMGE_AMD_SPMV_SELL=1 (tests): through the SELL-64 copy and its coded slices, the format of the level operators
    if (nrows == ncols && options().spmv_sell) build_sell(s, A);
    VecIn vx(x, (size_t)ncols, s);
    VecOut vy(y, (size_t)nrows, s, false);
    spmv(s, A, vx.p, vy.p);
    vy.finish();
    SA_API_END
}
int saamge_amd_spmv(int nrows, int ncols, const int *rowptr, const int *col, const double *val,
                    const double *x, double *y) {
    return spmv_entry(nrows, ncols, rowptr, 32, col, val, x, y);
}
int saamge_amd_spmv64(int nrows, int ncols, const long long *rowptr, const int *col, const double *val,
                      const double *x, double *y) {
    return spmv_entry(nrows, ncols, rowptr, 64, col, val, x, y);
}

__global__ void apply_dscale_kernel(int count, const int *ns, const int64_t *moff, const int64_t *voff,
                                    double *W, const double *D, double *dis) {
    const int b = blockIdx.x;
    const int n = ns[b];
    double *Wm = W + moff[b];
    const double *Dm = D + voff[b];
    for (int i = threadIdx.x; i < n; i += blockDim.x) dis[voff[b] + i] = 1.0 / sqrt(Dm[i]);
    __syncthreads();
    for (size_t idx = threadIdx.x; idx < (size_t)n * n; idx += blockDim.x) {
        const int r = (int)(idx % n), c = (int)(idx / n);
        Wm[idx] = dis[voff[b] + r] * Wm[idx] * dis[voff[b] + c];
    }
}

int saamge_amd_lower_eigens_batched(int count, const int *n, const double *A, const double *D,
                                    double vl, double vu, int *m, double *evals, double *evecs) {
    SA_API_BEGIN
    SA_REQUIRE(count >= 0 && n && A && D && m && evals && evecs, "bad argument");
    hipStream_t s = 0;
    set_thread_stream(s);
    std::vector<int> sizes(n, n + count);
    EigBatch b;
    eig_batch_alloc(b, sizes, s);
    b.set_window(vu);
    b.dense_only = options().eig_dense_only != 0;
    SA_HIP_CHECK(hipMemcpyAsync(b.W.p, A, 8 * (size_t)b.h_moff[count], hipMemcpyDefault, s));
    DBuf<double> dD;
    dD.assign(D, (size_t)b.h_voff[count], s);
    hipLaunchKernelGGL(apply_dscale_kernel, dim3(count), dim3(256), 0, s, count, b.n.p, b.moff.p,
                       b.voff.p, b.W.p, dD.p, b.dis.p);
    eig_tridiagonalize(s, b);
    eig_count(s, b, vl, vu);
    if (b.ss_failed || b.nbad) {     // few-eigenpairs path gave up (on the batch or on some matrices): the dense path on a fresh copy
        b.dense_only = true;
        b.subspace = b.ss_failed = false;
        SA_HIP_CHECK(hipMemcpyAsync(b.W.p, A, 8 * (size_t)b.h_moff[count], hipMemcpyDefault, s));
        hipLaunchKernelGGL(apply_dscale_kernel, dim3(count), dim3(256), 0, s, count, b.n.p, b.moff.p,
                           b.voff.p, b.W.p, dD.p, b.dis.p);
        eig_tridiagonalize(s, b);
        eig_count(s, b, vl, vu);
    }
    std::vector<int64_t> eoff((size_t)count + 1, 0), xoff((size_t)count + 1, 0);
    for (int i = 0; i < count; ++i) {
        eoff[i + 1] = eoff[i] + b.h_m[i];
        xoff[i + 1] = xoff[i] + (int64_t)b.h_m[i] * sizes[i];
        m[i] = b.h_m[i];
    }
    DBuf<int64_t> de, dx;
    de.from_host(eoff, s);
    dx.from_host(xoff, s);
    DBuf<double> ev((size_t)eoff[count] + 1), xv((size_t)xoff[count] + 1);
    eig_vectors(s, b, de.p, dx.p, ev.p, xv.p);
    auto hev = ev.to_host(s);
    auto hxv = xv.to_host(s);
    for (int i = 0; i < count; ++i) {
        std::copy(hev.begin() + eoff[i], hev.begin() + eoff[i + 1], evals + b.h_voff[i]);
        std::copy(hxv.begin() + xoff[i], hxv.begin() + xoff[i + 1], evecs + b.h_moff[i]);
    }
    SA_API_END
}

int saamge_amd_inertia_batched(int count, const int *n, const double *A, const double *D, double vu, int *neg) {
    SA_API_BEGIN
    SA_REQUIRE(count >= 0 && n && A && D && neg, "bad argument");
    hipStream_t s = 0;
    set_thread_stream(s);
    std::vector<int> sizes(n, n + count);
    EigBatch b;
    eig_batch_alloc(b, sizes, s);
    b.set_window(vu);
    SA_HIP_CHECK(hipMemcpyAsync(b.W.p, A, 8 * (size_t)b.h_moff[count], hipMemcpyDefault, s));
    DBuf<double> dD;
    dD.assign(D, (size_t)b.h_voff[count], s);
    hipLaunchKernelGGL(apply_dscale_kernel, dim3(count), dim3(256), 0, s, count, b.n.p, b.moff.p,
                       b.voff.p, b.W.p, dD.p, b.dis.p);
    (void)eig_subspace_factor(s, b);      // (the inertia pass runs ahead of the Cholesky factorisation)
    SA_REQUIRE((int)b.h_inertia.size() == count, "the inertia pass is switched off");
    std::copy(b.h_inertia.begin(), b.h_inertia.end(), neg);
    SA_API_END
}

void saamge_amd_release_cached_memory(void) {
    (void)hipDeviceSynchronize();
    eig_arena_release();
    dev_pool_release();
}
long long saamge_amd_cached_memory_bytes(void) { return (long long)dev_pool_idle_bytes(); }
void saamge_amd_memory_stats(long long *live_bytes, long long *peak_bytes, int reset_peak) {
    size_t l = 0, pk = 0;
    dev_memory_stats(&l, &pk, reset_peak != 0);
    if (live_bytes) *live_bytes = (long long)l;
    if (peak_bytes) *peak_bytes = (long long)pk;
}
void saamge_amd_pool_counts(long long counts[4], int reset) {
    long nm = 0, nf = 0;
    size_t mb = 0;
    dev_pool_counts(&nm, &nf, &mb, reset != 0);
    if (counts) { counts[0] = nm; counts[1] = (long long)mb; counts[2] = nf; counts[3] = (long long)dev_pool_idle_bytes(); }
}

void saamge_amd_profile_enable(int on) { profiler().enabled = on != 0; }
void saamge_amd_profile_reset(void) { profiler().stats.clear(); }
int saamge_amd_profile_count(void) { return (int)profiler().stats.size(); }
int saamge_amd_profile_get(int i, char *name, int name_len, double *ms, long long *launches,
                           double *bytes, double *flops) {
    if (i < 0 || i >= (int)profiler().stats.size()) return 1;
    const KernelStat &k = profiler().stats[i];
    if (name && name_len > 0) {
        std::snprintf(name, (size_t)name_len, "%s", k.name.c_str());
    }
    if (ms) *ms = k.ms;
    if (launches) *launches = k.launches;
    if (bytes) *bytes = k.bytes;
    if (flops) *flops = k.flops;
    return 0;
}

// the same with the bytes of the format in use (KernelStat::fmt_bytes) and, for the SpMV family, the operator's slice census
int saamge_amd_profile_get2(int i, char *name, int name_len, double *ms, long long *launches, double *bytes, double *flops,
                            double *fmt_bytes) {
    if (saamge_amd_profile_get(i, name, name_len, ms, launches, bytes, flops)) return 1;
    if (fmt_bytes) *fmt_bytes = profiler().stats[i].fmt_bytes;
    return 0;
}

}  // extern "C"
