// Setup (ml_produce_data) and solve (V-cycle, PCG) orchestration.  Host C++ driving the HIP
// kernels; all numerics run on the device, the integer topology on the host.
#include "hierarchy.h"
#include "dist.h"
#include "dense.h"
#include "spgemm.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <thread>

namespace saamge_amd {

Profiler &profiler() {
    static Profiler p;
    return p;
}

// smpr_sas_poly_roots (amg/src/smpr.cpp:282-306)
static std::vector<double> sas_poly_roots(int nu) {
    SA_REQUIRE(nu > 0, "nu_relax must be positive");
    std::vector<double> r;
    const double denom = (double)(2 * nu + 1);
    for (int i = 0; i <= 2 * nu; ++i) {
        const double v = std::cos(((double)i * M_PI) / denom);
        r.push_back(v * v);
    }
    for (int i = 1; i <= nu; ++i) {
        const double v = std::sin(((double)i * M_PI) / denom);
        r.push_back(v * v);
    }
    return r;
}

// SAAMGE_AMD_TIMING=1 prints host-side phase times (the reference's "TIMING:" lines)
struct PhaseTimer {
    bool on, host;
    hipStream_t s;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(hipStream_t st) : on(env_timing() || env_timing_host()), host(env_timing_host()), s(st) {
        t0 = std::chrono::steady_clock::now();
    }
    void lap(const char *what, int lev) {
        if (!on) return;
        if (!host) (void)hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "TIMING: level %d %-28s %9.3f ms\n", lev, what,
                     std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = std::chrono::steady_clock::now();
    }
};

// grow-only scratch buffers that survive across chunks / levels / hierarchies
static DBuf<double> &scratch_pool(int slot) {
    static DBuf<double> *pools = new DBuf<double>[4];     // never destroyed: no HIP calls from static destructors
    return pools[slot];
}
static double *scratch_get(int slot, size_t need) {
    DBuf<double> &p = scratch_pool(slot);
    if (p.n < need) p.alloc(need + need / 8 + 64);
    return p.p;
}

static void finish_csr(DCsr &A) { A.lanes_per_row = pick_lanes_per_row(A.nnz, A.nrows > 0 ? A.nrows : 1); }

__global__ void iota_int_kernel(long n, int *p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int)i;
}

__global__ void iota64_kernel(long n, int64_t scale, int64_t *p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = i * scale;
}

__global__ void gather_kernel(long n, const int *__restrict__ idx, const double *__restrict__ src, double *__restrict__ dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void fill_kernel(long n, double *p, double v) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------
// one coarsening: tg_init_data + tg_build_hierarchy + tg_update_coarse_operator
// ---------------------------------------------------------------------------------------
// P (smoothed when nu_pro > 0), R and Ac = R A P of level `lev` from its tentative prolongator
// (tg_smooth_interp + tg_coarse_matr, amg/inc/tg.hpp:679-709).  `first`: P / R still hold the
// tentative pair just built; otherwise (operator update) the stored tentative one is re-smoothed.
static void level_galerkin(Hierarchy &H, int lev, bool first, hipStream_t s) {
    Level &L = *H.levels[lev];
    const Params &P = H.params;
    const Relations &rel = L.rel;
    const int world = P.world > 1 ? P.world : 1;
    if (P.nu_pro[lev] == 0) {
        std::vector<long long> nnz_off;
        rap_mis(s, L.drel, rel, L.A, L.mis_k, L.mis_coloff, L.d_mis_k.p, L.d_mis_coloff.p,
                L.d_mis_u_off.p, L.mis_U.p, L.Ac, P.rank, world, world > 1 ? &nnz_off : nullptr);
        if (world > 1 && L.Ac.nnz > 0) {   // every rank computed the row blocks of its MIS range
            std::vector<long long> off((size_t)world + 1);
            for (int r = 0; r <= world; ++r) off[r] = 4ll * nnz_off[r];
            SA_REQUIRE(P.allgather(P.allgather_ctx, L.Ac.col.p, off.data()) == 0, "all-gather (Ac columns) failed");
            for (int r = 0; r <= world; ++r) off[r] = 8ll * nnz_off[r];
            SA_REQUIRE(P.allgather(P.allgather_ctx, L.Ac.val.p, off.data()) == 0, "all-gather (Ac values) failed");
        }
    } else {
        // interp_smooth (amg/src/interp.cpp:172-229): P = prod_k (I + (1/tau_k) Dinv_neg A) P_tent with
        // tau_k = sin^2(k pi / (2 nu + 1)) (smpr_sa_poly_roots, amg/src/smpr.cpp:266-280); then
        // R = P^T and Ac = R A P (mfem::RAP, amg/inc/tg.hpp:696-709) as general sparse products.
        const int nu = P.nu_pro[lev];
        if (first) L.Ptent = std::move(L.P);
        DCsr tmp[2];
        const DCsr *cur = &L.Ptent;
        for (int k = 1; k <= nu; ++k) {
            const double sn = std::sin((double)k * M_PI / (double)(2 * nu + 1));
            DCsr &dst = tmp[k & 1];
            dst = DCsr();
            spgemm(s, L.A, *cur, cur, L.dinv_neg.p, 1.0 / (sn * sn), 1.0, dst);
            cur = &dst;
        }
        if (P.smooth_drop_tol != 0.0)   // AltThreshold (amg/src/interp.cpp:219-227)
            csr_threshold(s, tmp[nu & 1], P.smooth_drop_tol, L.P);
        else
            L.P = std::move(tmp[nu & 1]);
        csr_transpose(s, L.P, L.R);
        DCsr AP;
        spgemm(s, L.A, L.P, nullptr, nullptr, 1.0, 0.0, AP);
        spgemm(s, L.R, AP, nullptr, nullptr, 1.0, 0.0, L.Ac);
    }
}

static void level_galerkin(Hierarchy &H, int lev, bool first) { level_galerkin(H, lev, first, H.stream); }

// The deferred Galerkin product (see Hierarchy::galerkin_thread): wait for it and hand its result to the
// next level.
static void join_galerkin(Hierarchy &H) {
    if (H.galerkin_lev < 0) return;
    if (H.galerkin_thread.joinable()) H.galerkin_thread.join();
    const int lev = H.galerkin_lev;
    H.galerkin_lev = -1;
    if (H.galerkin_err) {
        std::exception_ptr e = H.galerkin_err;
        H.galerkin_err = nullptr;
        std::rethrow_exception(e);
    }
    H.levels[lev + 1]->A = std::move(H.levels[lev]->Ac);     // A_{l+1} = Ac_l  (amg/src/ml.cpp:134)
}

struct DeviceInputs {   // level-0 inputs that already live on the device (see hierarchy_create)
    const int *e2d = nullptr, *part = nullptr;
    const signed char *bdr = nullptr;
    int NE = 0, nde = 0;
};

static void prepare_next_host(const Level &L, const roff_t *p_rowptr_dev, const double *p_val_dev, int p_nrows,
                              int64_t p_nnz, hipStream_t s, NextPrep &out);

static void build_level(Hierarchy &H, int lev, Table &&e2d, const hvec<int> &part, int nparts,
                        const signed char *bdr_host, const DeviceInputs *din = nullptr) {
    Level &L = *H.levels[lev];
    hipStream_t s = H.stream;
    const Params &P = H.params;
    L.theta = P.theta[lev];
    L.nu_relax = P.nu_relax[lev];
    SA_REQUIRE(P.nu_pro[lev] >= 0 && P.nu_pro[lev] <= 8, "bad prolongator smoothing degree");
    PhaseTimer tm(s);
    // Fine level, device-resident inputs: the SELL copy of the operator and the smoother diagonal (bandwidth work on A alone) are
    // built on a thread and stream of their own BESIDE the device build of the AE tables (integer work on elem_to_dof: sorts,
    // hash sets, atomics), and joined where they used to run, before the eigenproblems.
    std::thread op_thread;
    std::exception_ptr op_err;
    struct OpJoiner { std::thread &t; ~OpJoiner() { if (t.joinable()) t.join(); } } op_joiner{op_thread};
    const bool op_early = lev == 0 && din && !env_serial() && !profiler().enabled && H.galerkin_lev < 0 && (options().overlap & 8);
    if (op_early) {
        hipStream_t os = side_stream(6);
        const int dev0 = current_device();
        SA_HIP_CHECK(hipStreamSynchronize(s));          // (the operator's arrays are complete)
        op_thread = std::thread([&L, &op_err, os, dev0]() {
            try {
                adopt_device(dev0);
                set_thread_stream(os);
                build_sell(os, L.A);
                L.dinv_neg.alloc((size_t)L.A.nrows);
                DBuf<double> tmp((size_t)L.A.nrows);
                build_dinv_neg(os, L.A, tmp.p, L.dinv_neg.p);
                build_dinv_codes(os, L.A, L.dinv_neg.p);
                SA_HIP_CHECK(hipStreamSynchronize(os));
            } catch (...) { op_err = std::current_exception(); }
        });
    }
    bool on_device = false;
    if (din) {
        on_device = build_relations_ae_device(L.rel, L.drel, din->e2d, din->NE, din->nde, din->part, nparts,
                                              L.A.nrows, din->bdr, s);
        tm.lap("device topology (AE tables)", lev);
    }
    if (!on_device) {
        hvec<int> part_h;
        hvec<signed char> bdr_h;
        if (din) {   // agglomerates too large for the device kernels: fetch and take the host path
            e2d.J = fetch_host(din->e2d, (size_t)din->NE * din->nde, s);
            e2d.I.resize((size_t)din->NE + 1);
            for (int e = 0; e <= din->NE; ++e) e2d.I[e] = e * din->nde;
            e2d.ncols = L.A.nrows;
            part_h = fetch_host(din->part, (size_t)din->NE, s);
            if (din->bdr) bdr_h = fetch_host(din->bdr, (size_t)L.A.nrows, s);
        }
        if (!L.rel_prebuilt)
            build_relations_ae(L.rel, std::move(e2d), din ? part_h : part, nparts, L.A.nrows,
                               din ? (din->bdr ? bdr_h.data() : nullptr) : bdr_host);
        tm.lap(L.rel_prebuilt ? "host topology (built beside the element matrices)" : "host topology (AE tables)", lev);
        upload_relations_ae(L.drel, L.rel, s);
        tm.lap("upload topology", lev);
    }
    // the MIS tables are built on a host thread while the GPU solves the local eigenproblems
    // (do_aggregates, amg/src/ml.cpp:149: aggregates with arbitration on the LAST coarsening; the
    // greedy arbitration reads the level matrix on the host)
    HostCsr aggA;
    const bool aggregates = P.do_aggregates && lev == P.num_coarsenings - 1;
    if (aggregates) {
        join_galerkin(H);     // the arbitration reads this level's operator
        aggA.nrows = L.A.nrows;
        aggA.rowptr.resize((size_t)L.A.nrows + 1);
        aggA.col.resize((size_t)L.A.nnz);
        aggA.val.resize((size_t)L.A.nnz);
        SA_HIP_CHECK(hipMemcpyAsync(aggA.rowptr.data(), L.A.rowptr.p, sizeof(roff_t) * ((size_t)L.A.nrows + 1), hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipMemcpyAsync(aggA.col.data(), L.A.col.p, sizeof(int) * (size_t)L.A.nnz, hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipMemcpyAsync(aggA.val.data(), L.A.val.p, sizeof(double) * (size_t)L.A.nnz, hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
    // (the tables go to the device from the same thread, on a side stream, while the eigensolver runs)
    hipStream_t mis_stream = side_stream(0);
    const int dev = current_device();
    std::exception_ptr mis_err;
    // SAAMGE_AMD_SERIAL=1: no worker threads anywhere in the setup (counter passes attribute launches per thread; the
    // work runs in line, same streams).  The "stream_stack.cpp: Check failed" aborts once seen under rocprofv3 --pmc came
    // from static destructors calling HIP at exit (fixed in round 4: those objects are never destroyed), not from threads
    const bool serial = env_serial();
    auto mis_work = [&]() {
        try {
            adopt_device(dev);   // the worker allocates and copies: same GPU as the caller
            set_thread_stream(mis_stream);
            // MIS tables on the device (SAAMGE_AMD_HOST_MIS=1: host build); aggregates with arbitration are
            // sequential by definition and stay on the host
            constexpr bool host_mis = false;      // (the host build stays the fallback after a hash collision and for aggregates with arbitration)
            PhaseTimer tmis(mis_stream);
            bool on_dev = false;
            if (!aggregates && !host_mis) on_dev = build_relations_mis_device(L.rel, L.drel, mis_stream);
            if (on_dev) tmis.lap("    device MIS tables", lev);
            if (!on_dev) {
                fetch_relations_ae_host(L.rel, L.drel, mis_stream);
                build_relations_mis(L.rel, aggregates ? &aggA : nullptr);
                upload_relations_mis(L.drel, L.rel, mis_stream);
            }
            SA_HIP_CHECK(hipStreamSynchronize(mis_stream));
        } catch (...) { mis_err = std::current_exception(); }
    };
    std::thread mis_thread;
    if (serial) {
        mis_work();
        set_thread_stream(s);
    } else {
        mis_thread = std::thread(mis_work);
    }
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{mis_thread};
    const Relations &rel = L.rel;
    // SELL-64 copy of the level operator for the SpMV family + smoother data (smpr_init_poly_data,
    // amg/src/smpr.cpp:359-423); on a coarse level the operator may still be in the making (deferred
    // Galerkin product of the finer level): then after the eigenproblems, which do not read it
    auto operator_data = [&]() {
        join_galerkin(H);
        build_sell(s, L.A);
        L.dinv_neg.alloc((size_t)L.A.nrows);
        DBuf<double> tmp((size_t)L.A.nrows);
        build_dinv_neg(s, L.A, tmp.p, L.dinv_neg.p);
        build_dinv_codes(s, L.A, L.dinv_neg.p);
        SA_HIP_CHECK(hipStreamSynchronize(s));
    };
    const bool operator_pending = H.galerkin_lev >= 0;
    // (Measured in round 4 and dropped: the fine level's SELL copy and D^-1 -- 13 ms of bandwidth work the setup itself does not
    // need -- on a thread and stream of their own beside the next level: setup 247 -> 243 ms, but the smoother then ran at 221
    // instead of 216 us per step and the step took 399 instead of 390 ms.)
    if (op_early) {
        op_thread.join();
        if (op_err) std::rethrow_exception(op_err);
    } else if (!operator_pending) operator_data();
    L.roots = sas_poly_roots(L.nu_relax);

    // ---- local spectral problems, chunked over AEs (interp_compute_vectors) ----
    // (the sparse rows of the fine AE matrices are kept for the coarse element matrices of the next
    // level when there is one: ~10 bytes per stored entry of the overlapping AE rows)
    const bool keep_rows = lev == 0 && lev + 1 < P.num_coarsenings &&
                           (double)L.rel.AE_to_dof.I[nparts] * 30.0 * 10.0 < 32e9;
    std::vector<int> sizes((size_t)nparts);
    for (int p = 0; p < nparts; ++p) sizes[p] = rel.AE_to_dof.row_size(p);
    L.ae_m.assign((size_t)nparts, 0);
    L.ae_solved = 0;
    L.ae_class.assign((size_t)nparts, -1);
    L.ae_evclass.assign((size_t)nparts, -1);
    struct Chunk { int ae0, count; DBuf<double> evals, evecs; std::vector<int64_t> eoff, xoff; DBuf<int64_t> d_eoff, d_xoff; };
    std::vector<Chunk> chunks;
    if (P.keep_debug) L.ae_D.alloc((size_t)rel.AE_to_dof.I[nparts]);
    // AE ownership: contiguous ranges balanced by the n^3 cost of the eigenproblems
    const int world = P.world > 1 ? P.world : 1;
    std::vector<int> ae_begin((size_t)world + 1, nparts);
    {
        double total = 0.0;
        for (int p = 0; p < nparts; ++p) total += (double)sizes[p] * sizes[p] * sizes[p];
        double run = 0.0;
        int r = 0;
        ae_begin[0] = 0;
        for (int p = 0; p < nparts && r + 1 < world; ++p) {
            run += (double)sizes[p] * sizes[p] * sizes[p];
            while (r + 1 < world && run >= total * (r + 1) / world) ae_begin[++r] = p + 1;
        }
        ae_begin[world] = nparts;
    }
    if (H.dist_in && lev < (int)H.dist_in->ae_begin.size()) {
        // per-rank inputs: a rank owns the agglomerates made of its own elements (only their element matrices are here)
        const std::vector<long long> &ab = H.dist_in->ae_begin[lev];
        SA_REQUIRE((int)ab.size() == world + 1 && ab[world] == nparts, "per-rank inputs: agglomerate ranges do not match the level");
        for (int r = 0; r <= world; ++r) ae_begin[r] = (int)ab[r];
    }
    if (world > 1) {
        SA_REQUIRE(P.allgather != nullptr, "world > 1 needs an all-gather callback");
        SA_REQUIRE(!(P.testmesh && lev == 0), "the mltest fixture is single-rank only");
    }
    const int ae_lo = ae_begin[world > 1 ? P.rank : 0], ae_hi = ae_begin[world > 1 ? P.rank + 1 : 1];
    L.ae_begin = ae_begin;
    int64_t row0 = 0;
    for (int p = 0; p < ae_lo; ++p) row0 += sizes[p];
    // (a two-stream pipeline over the chunks -- band reduction of chunk i beside the chase of chunk
    // i-1 -- was measured without gain on MI355X, 256^3: 3.48 s vs 3.29 s, and removed)
    hipStream_t qa = s, qb = s;
    SA_HIP_CHECK(hipStreamSynchronize(s));
    const size_t chunk_bytes = P.workspace_bytes;
    // (batches: what the eigensolvers run on -- the assembled batch itself, or the batch of the representatives of its
    // classes of bitwise identical matrices, eig.hip "Duplicate agglomerate matrices"; rep_of: empty = no classes)
    EigBatch batches[2], assembled[2];
    // Classes of bitwise identical agglomerates of this level (a rank's part of it): the first member met is solved, in
    // the chunk it sits in; every later member -- of that chunk or a later one -- receives a copy.  A class keeps the words
    // it consists of (the first member's: sparse rows or band, eig.h DdSource), against which the candidates of later
    // chunks are compared word by word, and where its eigenpairs are.
    struct SolvedClass {
        int n = 0, m = 0, kind = 0;
        bool bad = false;           // the few-eigenpairs path gave up on it: every member is redone by the dense path
        long words = 0;
        DBuf<unsigned long long> blob;
        const double *evals = nullptr, *evecs = nullptr;
    };
    std::vector<SolvedClass> lvl_classes;
    std::unordered_map<DdKey, std::vector<int>, DdKeyHash> lvl_by_hash;
    std::vector<DBuf<double>> kept;          // packed eigenpairs of the chunks' representatives
    std::vector<int> cls_of[2];              // per slot: class of every agglomerate of the chunk (empty: no classes)
    std::vector<int> solve_cls[2];           // per slot: classes of the matrices of the batch that is solved, in its order
    const bool dedupe = options().eig_dedupe != 0 && !(P.testmesh && lev == 0);
    bool dedupe_level = true;
    int pend_ae0[2] = {0, 0}, pend_cnt[2] = {0, 0};
    int64_t pend_row0[2] = {0, 0};
    // The subspace iteration of a chunk (a few hundred to a few thousand small matrices still active: launches that
    // fill a fraction of the chip, with a host round trip every few iterations) runs on its own thread and stream
    // BESIDE the assembly and the factorisations of the next chunk (SAAMGE_AMD_EIG_OVERLAP=0: one after the other).
    // Only eig_subspace_iterate runs there: it touches its own batch (the other workspace slot) and nothing else;
    // everything that assembles or allocates workspace stays on this thread.
    const bool overlap_env = (options().overlap & 1) && !env_serial();
    const bool overlap_iter = overlap_env && !profiler().enabled;
    hipStream_t iter_stream = overlap_iter ? side_stream(3) : s;
    std::thread iter_thread;
    std::exception_ptr iter_err;
    struct IterJoiner { std::thread &t; ~IterJoiner() { if (t.joinable()) t.join(); } } iter_joiner{iter_thread};
    bool counted[2] = {false, false};
    auto start_iterate = [&](int slot) {
        EigBatch &batch = batches[slot];
        counted[slot] = false;
        if (!overlap_iter || !batch.subspace || batch.ss_failed || !batch.count) return;
        counted[slot] = true;
        iter_thread = std::thread([&, slot]() {
            try {
                adopt_device(dev);
                set_thread_stream(iter_stream);
                eig_count(iter_stream, batches[slot], -1.0, L.theta);
                SA_HIP_CHECK(hipStreamSynchronize(iter_stream));
            } catch (...) { iter_err = std::current_exception(); }
        });
    };
    auto join_iterate = [&]() {
        if (iter_thread.joinable()) iter_thread.join();
        if (iter_err) { std::exception_ptr e = iter_err; iter_err = nullptr; std::rethrow_exception(e); }
    };
    auto post = [&](int slot) {   // band -> tridiagonal, counts, eigenvectors of the chunk in `slot`
        EigBatch &batch = batches[slot];
        const int ae0 = pend_ae0[slot], cnt = pend_cnt[slot];
        const std::vector<int> &co = cls_of[slot];
        bool have = batch.count > 0;      // (with classes: the batch of the NEW classes' representatives, possibly empty)
        if (have && !counted[slot]) {
            eig_tridiagonalize(qb, batch, 2);
            eig_count(qb, batch, -1.0, L.theta);
        }
        if (have && batch.ss_failed && !co.empty()) {      // the new classes go to the dense path, member by member (below)
            for (int id : solve_cls[slot]) lvl_classes[id].bad = true;
            have = false;
        }
        if (have && batch.ss_failed) {   // few-eigenpairs path gave up on this chunk: dense path on re-assembled matrices
            batch.dense_only = true;
            batch.subspace = batch.ss_failed = false;
            const RowsSpan span{(int64_t)L.rel.AE_to_dof.I[ae0] - (int64_t)L.rel.AE_to_dof.I[ae_lo], (int64_t)L.rel.AE_to_dof.I[ae_hi] - (int64_t)L.rel.AE_to_dof.I[ae_lo]};      // (positions among the rows of this rank's agglomerates)
            ae_build(qb, L.drel, lev == 0 ? &L.A : nullptr, L.elmat, ae0, batch, true,
                     P.keep_debug ? L.ae_D.p + pend_row0[slot] : nullptr, keep_rows ? &span : nullptr);
            eig_tridiagonalize(qb, batch, 3);
            eig_count(qb, batch, -1.0, L.theta);
        }
        chunks.emplace_back();
        Chunk &c = chunks.back();
        c.ae0 = ae0;
        c.count = cnt;
        c.eoff.assign((size_t)cnt + 1, 0);
        c.xoff.assign((size_t)cnt + 1, 0);
        // per agglomerate of the chunk: eigenvector count and "redo by the dense path" -- its own, or its class's
        std::vector<int> hm((size_t)cnt);
        std::vector<char> hbad((size_t)cnt, 0);
        bool some_bad = have && batch.subspace && batch.nbad > 0;
        if (!co.empty() && have) {      // the eigenpairs of the classes solved in this chunk, packed; kept for the later members
            const int nr = batch.count;
            std::vector<int64_t> re((size_t)nr + 1, 0), rx((size_t)nr + 1, 0);
            for (int r = 0; r < nr; ++r) {
                re[r + 1] = re[r] + batch.h_m[r];
                rx[r + 1] = rx[r] + (int64_t)batch.h_m[r] * batch.h_n[r];
            }
            kept.emplace_back((size_t)re[nr] + 1);
            double *revals = kept.back().p;
            kept.emplace_back((size_t)rx[nr] + 1);
            double *revecs = kept.back().p;
            DBuf<int64_t> d_re, d_rx;
            d_re.from_host(re, qb);
            d_rx.from_host(rx, qb);
            eig_vectors(qb, batch, d_re.p, d_rx.p, revals, revecs);
            SA_HIP_CHECK(hipStreamSynchronize(qb));
            for (int r = 0; r < nr; ++r) {
                SolvedClass &sc = lvl_classes[solve_cls[slot][r]];
                sc.m = batch.h_m[r];
                sc.bad = some_bad && batch.h_bad[r];
                sc.evals = revals + re[r];
                sc.evecs = revecs + rx[r];
            }
        }
        some_bad = co.empty() ? some_bad : false;
        for (int i = 0; i < cnt; ++i) {
            if (co.empty()) {
                hm[i] = batch.h_m[i];
                if (some_bad) hbad[i] = batch.h_bad[i];
            } else {
                const SolvedClass &sc = lvl_classes[co[i]];
                if (sc.kind == 0) L.ae_class[ae0 + i] = co[i];
                if (!sc.bad) L.ae_evclass[ae0 + i] = co[i];
                hm[i] = sc.bad ? 0 : sc.m;
                hbad[i] = sc.bad ? 1 : 0;
                some_bad = some_bad || sc.bad;
            }
        }
        for (int i = 0; i < cnt; ++i) {
            L.ae_m[ae0 + i] = hm[i];
            c.eoff[i + 1] = c.eoff[i] + hm[i];
            c.xoff[i + 1] = c.xoff[i] + (int64_t)hm[i] * sizes[ae0 + i];
        }
        c.evals.alloc((size_t)c.eoff[cnt]);
        c.evecs.alloc((size_t)c.xoff[cnt]);
        c.d_eoff.from_host(c.eoff, qb);
        c.d_xoff.from_host(c.xoff, qb);
        if (co.empty()) {
            eig_vectors(qb, batch, c.d_eoff.p, c.d_xoff.p, c.evals.p, c.evecs.p);
        } else {      // a copy of its class's eigenpairs to every agglomerate
            std::vector<const double *> pe((size_t)cnt), px((size_t)cnt);
            for (int i = 0; i < cnt; ++i) { pe[i] = lvl_classes[co[i]].evals; px[i] = lvl_classes[co[i]].evecs; }
            DBuf<const double *> d_pe, d_px;
            d_pe.from_host(pe, qb);
            d_px.from_host(px, qb);
            eig_dedupe_expand(qb, cnt, sizes.empty() ? 1 : *std::max_element(sizes.begin() + ae0, sizes.begin() + ae0 + cnt), d_pe.p, d_px.p,
                              c.d_eoff.p, c.d_xoff.p, c.evals.p, c.evecs.p);
        }
        SA_HIP_CHECK(hipStreamSynchronize(qb));
        if (!some_bad) return;
        // ---- the few-eigenpairs path finished all but a few matrices of the chunk (h_bad): those are redone by
        // the dense path, as contiguous runs of agglomerates (runs closer than three apart are joined), in the
        // workspace the chunk has just left; then the chunk's packed results are put together ----
        std::vector<char> bad(hbad.begin(), hbad.end());
        for (int i = 0; i < cnt; ++i)
            if (bad[i])
                for (int j = i + 1; j < std::min(cnt, i + 4); ++j)
                    if (bad[j]) { for (int q = i + 1; q < j; ++q) bad[q] = 1; break; }
        std::vector<int> m_all(hm.begin(), hm.end());
        struct Redo { int a, b; DBuf<double> evals, evecs; std::vector<int64_t> eoff, xoff; };
        std::vector<Redo> redo;
        for (int a = 0; a < cnt;) {
            if (!bad[a]) { ++a; continue; }
            int e = a;
            while (e < cnt && bad[e]) ++e;
            redo.emplace_back();
            Redo &r = redo.back();
            r.a = a;
            r.b = e;
            EigBatch sub;
            eig_batch_alloc(sub, std::vector<int>(sizes.begin() + ae0 + a, sizes.begin() + ae0 + e), qb, slot);
            sub.dense_only = true;
            sub.set_window(L.theta);
            int64_t rows_before = 0;
            for (int i = 0; i < a; ++i) rows_before += sizes[ae0 + i];
            const RowsSpan span{(int64_t)L.rel.AE_to_dof.I[ae0 + a] - (int64_t)L.rel.AE_to_dof.I[ae_lo], (int64_t)L.rel.AE_to_dof.I[ae_hi] - (int64_t)L.rel.AE_to_dof.I[ae_lo]};
            ae_build(qb, L.drel, lev == 0 ? &L.A : nullptr, L.elmat, ae0 + a, sub, true,
                     P.keep_debug ? L.ae_D.p + pend_row0[slot] + rows_before : nullptr, keep_rows ? &span : nullptr);
            eig_tridiagonalize(qb, sub, 3);
            eig_count(qb, sub, -1.0, L.theta);
            r.eoff.assign((size_t)(e - a) + 1, 0);
            r.xoff.assign((size_t)(e - a) + 1, 0);
            for (int i = a; i < e; ++i) {
                m_all[i] = sub.h_m[i - a];
                r.eoff[i - a + 1] = r.eoff[i - a] + sub.h_m[i - a];
                r.xoff[i - a + 1] = r.xoff[i - a] + (int64_t)sub.h_m[i - a] * sizes[ae0 + i];
            }
            r.evals.alloc((size_t)r.eoff.back() + 1);
            r.evecs.alloc((size_t)r.xoff.back() + 1);
            DBuf<int64_t> de, dx;
            de.from_host(r.eoff, qb);
            dx.from_host(r.xoff, qb);
            eig_vectors(qb, sub, de.p, dx.p, r.evals.p, r.evecs.p);
            SA_HIP_CHECK(hipStreamSynchronize(qb));
            a = e;
        }
        // packed results of the whole chunk: stretches of finished matrices from the first pass, the runs from `redo`
        std::vector<int64_t> eoff((size_t)cnt + 1, 0), xoff((size_t)cnt + 1, 0);
        for (int i = 0; i < cnt; ++i) {
            L.ae_m[ae0 + i] = m_all[i];
            eoff[i + 1] = eoff[i] + m_all[i];
            xoff[i + 1] = xoff[i] + (int64_t)m_all[i] * sizes[ae0 + i];
        }
        DBuf<double> evals((size_t)eoff[cnt]), evecs((size_t)xoff[cnt]);
        auto copy = [&](double *dst, const double *src, int64_t n_) {
            if (n_ > 0) SA_HIP_CHECK(hipMemcpyAsync(dst, src, 8 * (size_t)n_, hipMemcpyDeviceToDevice, qb));
        };
        size_t ri = 0;
        for (int i = 0; i < cnt;) {
            if (ri < redo.size() && redo[ri].a == i) {
                const Redo &r = redo[ri++];
                copy(evals.p + eoff[i], r.evals.p, r.eoff.back());
                copy(evecs.p + xoff[i], r.evecs.p, r.xoff.back());
                i = r.b;
            } else {
                const int e = ri < redo.size() ? redo[ri].a : cnt;       // finished matrices i .. e - 1: contiguous in both
                copy(evals.p + eoff[i], c.evals.p + c.eoff[i], c.eoff[e] - c.eoff[i]);
                copy(evecs.p + xoff[i], c.evecs.p + c.xoff[i], c.xoff[e] - c.xoff[i]);
                i = e;
            }
        }
        SA_HIP_CHECK(hipStreamSynchronize(qb));
        c.evals = std::move(evals);
        c.evecs = std::move(evecs);
        c.eoff = eoff;
        c.xoff = xoff;
    };
    int prev = -1, idx = 0;
    for (int ae0 = ae_lo; ae0 < ae_hi; ++idx) {
        size_t bytes = 0;
        int cnt = 0;
        while (ae0 + cnt < ae_hi) {
            const size_t add = eig_workspace_bytes(sizes[ae0 + cnt]);
            if (cnt > 0 && bytes + add > chunk_bytes) break;
            bytes += add;
            ++cnt;
        }
        // Large agglomerates (the wide-band path: one workgroup per matrix in the panel and solve kernels, one or two
        // workgroups per CU): a chunk of 577 of them is two full rounds over the 256 CUs and a third at a quarter of
        // the card -- whole multiples of 512 instead (config 5: 27 chunks of 512 instead of 24 of 577).
        constexpr bool round_chunks = true;
        if (round_chunks && cnt > 512 && ae0 + cnt < ae_hi && sizes[ae0] > 1280) cnt = (cnt / 512) * 512;
        const int slot = idx & 1;
        EigBatch &batch = batches[slot];
        batch = EigBatch();
        eig_batch_alloc(batch, std::vector<int>(sizes.begin() + ae0, sizes.begin() + ae0 + cnt), qa, slot);
        batch.dense_only = P.eigensolver == 1;
        batch.ss_tol = P.eig_tol;
        batch.set_window(L.theta);
        if (lev > 0 && H.levels[lev - 1]->cvec_next.n == (size_t)L.A.nrows) {
            const size_t rows = (size_t)batch.h_voff[cnt];
            batch.x0c.alloc(rows);
            hipLaunchKernelGGL(gather_kernel, dim3(div_up((long)rows, 256)), dim3(256), 0, qa, (long)rows,
                               L.drel.ae2d_J.p + L.rel.AE_to_dof.I[ae0], H.levels[lev - 1]->cvec_next.p, batch.x0c.p);
            batch.has_x0c = true;
        }
        const RowsSpan span{(int64_t)L.rel.AE_to_dof.I[ae0] - (int64_t)L.rel.AE_to_dof.I[ae_lo], (int64_t)L.rel.AE_to_dof.I[ae_hi] - (int64_t)L.rel.AE_to_dof.I[ae_lo]};      // (positions among the rows of this rank's agglomerates)
        AeClasses classes;
        ae_build(qa, L.drel, lev == 0 ? &L.A : nullptr, L.elmat, ae0, batch, true,
                 P.keep_debug ? L.ae_D.p + row0 : nullptr, keep_rows ? &span : nullptr,
                 dedupe && dedupe_level && !batch.dense_only ? &classes : nullptr);
        const int64_t rows_chunk = batch.h_voff[cnt];
        cls_of[slot].clear();
        solve_cls[slot].clear();
        if (dedupe && dedupe_level && !batch.dense_only) {
            // classes within the chunk: known before its matrices were built (the fused fine-level assembly) or found on them
            DdSource src = classes.src;
            bool found = classes.early;
            if (!found && !classes.searched && batch.has_bw) {      // (distinct sparse rows: distinct matrices)
                src = eig_dedupe_source(batch);
                found = eig_dedupe_find(qa, src, cnt, batch.max_n, classes.cls);
            }
            // a level whose first chunk has (almost) no identical agglomerates is not searched further: variable coefficients
            if (!found && lvl_classes.empty()) dedupe_level = false;
            if (found) {
                const DdClasses &cl = classes.cls;
                const int nl = (int)cl.reps.size();
                std::vector<int> l2g((size_t)nl, -1);
                // ... against the classes of the earlier chunks: same hash, then word by word against the class's first member
                std::vector<int> list, qidx, cand;
                std::vector<const unsigned long long *> blobs;
                std::vector<long> bwords;
                for (int q = 0; q < nl; ++q) {
                    auto it = lvl_by_hash.find(DdKey{cl.rep_hash[2 * (size_t)q], cl.rep_hash[2 * (size_t)q + 1]});
                    if (it == lvl_by_hash.end()) continue;
                    for (int id : it->second)
                        if (lvl_classes[id].kind == src.kind) {
                            list.push_back(cl.reps[q]); qidx.push_back(q); cand.push_back(id);
                            blobs.push_back(lvl_classes[id].blob.p); bwords.push_back(lvl_classes[id].words);
                        }
                }
                std::vector<char> same;
                eig_dedupe_compare(qa, src, batch.max_n, list, blobs, bwords, same);
                for (size_t t = 0; t < list.size(); ++t)
                    if (same[t] && l2g[qidx[t]] < 0) l2g[qidx[t]] = cand[t];
                // the new classes: their words are kept, their first members are what this chunk solves
                std::vector<int> newq, newlist;
                for (int q = 0; q < nl; ++q)
                    if (l2g[q] < 0) { newq.push_back(q); newlist.push_back(cl.reps[q]); }
                const std::vector<long> nwords = eig_dedupe_words(qa, src, batch.h_n, newlist);
                std::vector<DBuf<unsigned long long>> nblobs;
                eig_dedupe_pack(qa, src, batch.max_n, newlist, nwords, nblobs);
                // (local classes found on the INPUTS of the assembly can share their assembled matrix -- agglomerates that differ
                // in the flags of their surface dofs only: candidates with the hash of an earlier candidate are compared with it)
                std::vector<int> alias(newq.size(), -1);
                {
                    std::unordered_map<DdKey, int, DdKeyHash> firstc;
                    std::vector<int> clist, cwho;
                    std::vector<const unsigned long long *> cblobs;
                    std::vector<long> cwords;
                    for (size_t t = 0; t < newq.size(); ++t) {
                        auto it = firstc.emplace(DdKey{cl.rep_hash[2 * (size_t)newq[t]], cl.rep_hash[2 * (size_t)newq[t] + 1]}, (int)t);
                        if (it.second) continue;
                        clist.push_back(newlist[t]); cwho.push_back((int)t);
                        cblobs.push_back(nblobs[it.first->second].p); cwords.push_back(nwords[it.first->second]);
                    }
                    std::vector<char> csame;
                    eig_dedupe_compare(qa, src, batch.max_n, clist, cblobs, cwords, csame);
                    for (size_t u = 0; u < clist.size(); ++u)
                        if (csame[u]) alias[cwho[u]] = firstc[DdKey{cl.rep_hash[2 * (size_t)newq[cwho[u]]], cl.rep_hash[2 * (size_t)newq[cwho[u]] + 1]}];
                }
                std::vector<int> solve_list;
                std::vector<int> id_of(newq.size(), -1);
                for (size_t t = 0; t < newq.size(); ++t) {
                    if (alias[t] >= 0) { id_of[t] = id_of[alias[t]]; l2g[newq[t]] = id_of[t]; continue; }
                    solve_list.push_back(newlist[t]);
                    const int id = (int)lvl_classes.size();
                    id_of[t] = id;
                    lvl_classes.emplace_back();
                    SolvedClass &sc = lvl_classes.back();
                    sc.n = batch.h_n[newlist[t]];
                    sc.kind = src.kind;
                    sc.words = nwords[t];
                    sc.blob = std::move(nblobs[t]);
                    lvl_by_hash[DdKey{cl.rep_hash[2 * (size_t)newq[t]], cl.rep_hash[2 * (size_t)newq[t] + 1]}].push_back(id);
                    l2g[newq[t]] = id;
                    solve_cls[slot].push_back(id);
                }
                cls_of[slot].resize((size_t)cnt);
                for (int i = 0; i < cnt; ++i) cls_of[slot][i] = l2g[cl.rep_of[i]];
                assembled[slot] = std::move(batch);
                eig_batch_compact(qa, batch, assembled[slot], solve_list);
            }
        }
        if (batch.count) eig_tridiagonalize(qa, batch, 1);
        L.ae_solved += batch.count;      // (eigenproblems that are solved, not copied)
        pend_ae0[slot] = ae0;
        pend_cnt[slot] = cnt;
        pend_row0[slot] = row0;
        if (prev >= 0) { join_iterate(); post(prev); }
        start_iterate(slot);
        prev = slot;
        row0 += rows_chunk;
        ae0 += cnt;
    }
    if (prev >= 0) { join_iterate(); post(prev); }
    tm.lap("local eigenproblems", lev);
    if (operator_pending) {
        operator_data();
        tm.lap("operator (deferred Galerkin product) + SELL + D", lev);
    }
    if (world > 1) {   // exchange the number of eigenvectors per AE
        DBuf<int> d_m;
        d_m.from_host(L.ae_m, s);
        std::vector<long long> off((size_t)world + 1);
        for (int r = 0; r <= world; ++r) off[r] = 4ll * ae_begin[r];
        SA_REQUIRE(P.allgather(P.allgather_ctx, d_m.p, off.data()) == 0, "all-gather (counts) failed");
        auto t_ = d_m.to_host(s);
        L.ae_m.assign(t_.begin(), t_.end());
    }
    // concatenate (+ the mltest fixture's extra all-ones vector on AE 0 of the finest level)
    const bool extra = (P.testmesh && lev == 0);
    L.ae_xoff.assign((size_t)nparts + 1, 0);
    L.ae_eoff.assign((size_t)nparts + 1, 0);
    std::vector<int> m_tot = L.ae_m;
    if (extra) m_tot[0] += 1;
    for (int p = 0; p < nparts; ++p) {
        L.ae_xoff[p + 1] = L.ae_xoff[p] + (int64_t)m_tot[p] * sizes[p];
        L.ae_eoff[p + 1] = L.ae_eoff[p] + L.ae_m[p];
    }
    L.evals.alloc((size_t)L.ae_eoff[nparts]);
    L.evecs.alloc((size_t)L.ae_xoff[nparts]);
    for (Chunk &c : chunks) {
        if (c.evals.n)
            SA_HIP_CHECK(hipMemcpyAsync(L.evals.p + L.ae_eoff[c.ae0], c.evals.p, 8 * c.evals.n,
                                        hipMemcpyDeviceToDevice, s));
        if (extra && c.ae0 == 0) {
            const int64_t first = (int64_t)L.ae_m[0] * sizes[0];
            SA_HIP_CHECK(hipMemcpyAsync(L.evecs.p, c.evecs.p, 8 * first, hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(fill_kernel, dim3(div_up(sizes[0], 256)), dim3(256), 0, s, (long)sizes[0],
                               L.evecs.p + first, 1.0);
            if (c.evecs.n > (size_t)first)
                SA_HIP_CHECK(hipMemcpyAsync(L.evecs.p + L.ae_xoff[1], c.evecs.p + first,
                                            8 * (c.evecs.n - first), hipMemcpyDeviceToDevice, s));
        } else if (c.evecs.n) {
            SA_HIP_CHECK(hipMemcpyAsync(L.evecs.p + L.ae_xoff[c.ae0], c.evecs.p, 8 * c.evecs.n,
                                        hipMemcpyDeviceToDevice, s));
        }
    }
    SA_HIP_CHECK(hipStreamSynchronize(s));
    chunks.clear();
    L.ae_m = m_tot;
    if (world > 1) {   // all-gather the eigenvectors (and, for inspection, eigenvalues and D) in place
        std::vector<long long> off((size_t)world + 1);
        for (int r = 0; r <= world; ++r) off[r] = 8ll * L.ae_xoff[ae_begin[r]];
        SA_REQUIRE(P.allgather(P.allgather_ctx, L.evecs.p, off.data()) == 0, "all-gather (eigenvectors) failed");
        if (P.keep_debug) {
            for (int r = 0; r <= world; ++r) off[r] = 8ll * L.ae_eoff[ae_begin[r]];
            SA_REQUIRE(P.allgather(P.allgather_ctx, L.evals.p, off.data()) == 0, "all-gather (eigenvalues) failed");
            int64_t rows = 0;
            std::vector<int64_t> rowoff((size_t)nparts + 1, 0);
            for (int p = 0; p < nparts; ++p) rowoff[p + 1] = rowoff[p] + sizes[p];
            (void)rows;
            for (int r = 0; r <= world; ++r) off[r] = 8ll * rowoff[ae_begin[r]];
            SA_REQUIRE(P.allgather(P.allgather_ctx, L.ae_D.p, off.data()) == 0, "all-gather (D) failed");
        }
        tm.lap("all-gather eigenvectors", lev);
    }

    // ---- MIS stage (ContribTent::contrib_mises) ----
    if (mis_thread.joinable()) mis_thread.join();
    if (mis_err) std::rethrow_exception(mis_err);
    tm.lap("MIS tables (join)", lev);
    const int nm = rel.num_mises;
    std::vector<int64_t> g_off((size_t)nm + 1, 0);
    L.mis_u_off.assign((size_t)nm + 1, 0);
    L.mis_s_off.assign((size_t)nm + 1, 0);
    int max_ctot = 0;
    const int nextra = (lev == 0 && P.extra_modes) ? P.num_extra_modes : 0;
    if (nextra) import_array(L.extra, P.extra_modes, (size_t)nextra * L.A.nrows, s);
    for (int m = 0; m < nm; ++m) {
        int ctot = nextra;
        for (int q = rel.mis_to_AE.I[m]; q < rel.mis_to_AE.I[m + 1]; ++q) ctot += L.ae_m[rel.mis_to_AE.J[q]];
        const int r = rel.mis_to_dof.row_size(m);
        g_off[m + 1] = g_off[m] + (int64_t)r * ctot;
        L.mis_u_off[m + 1] = L.mis_u_off[m] + (int64_t)r * std::max(1, std::min(r, ctot));
        L.mis_s_off[m + 1] = L.mis_s_off[m] + ctot;
        max_ctot = std::max(max_ctot, ctot);
    }
    L.mis_U.alloc((size_t)L.mis_u_off[nm]);
    L.mis_sig.alloc((size_t)L.mis_s_off[nm] + 1);
    L.mis_sig.zero(s);
    L.d_mis_k.alloc((size_t)nm);
    {
        double *gather = scratch_get(1, (size_t)g_off[nm] + 1);
        DBuf<int64_t> d_goff, d_soff, d_xoff;
        DBuf<int> d_aem, d_ncols((size_t)nm);
        d_goff.from_host(g_off, s);
        d_soff.from_host(L.mis_s_off, s);
        L.d_mis_u_off.from_host(L.mis_u_off, s);
        d_xoff.from_host(L.ae_xoff, s);
        d_aem.from_host(L.ae_m, s);
        MisSvdIO io;
        io.ae_m = d_aem.p;
        io.ae_xoff = d_xoff.p;
        io.evecs = L.evecs.p;
        io.g_off = d_goff.p;
        io.gather = gather;
        io.u_off = L.d_mis_u_off.p;
        io.s_off = d_soff.p;
        io.U = L.mis_U.p;
        io.sig = L.mis_sig.p;
        io.k = L.d_mis_k.p;
        io.ncols = d_ncols.p;
        io.avoid_ess = P.avoid_ess_bdr_dofs;
        io.extra = nextra ? L.extra.p : nullptr;
        io.nextra = nextra;
        io.ND = L.A.nrows;
        if (world > 1) {
            // every rank takes a contiguous range of MISes (balanced by the r c^2 cost of the SVDs) and the bases,
            // singular values and counts are all-gathered in place: the counterpart of the reference's
            // owner-computes SVD + broadcast (SharedEntityCommunication, amg/src/contrib.cpp:519-548)
            std::vector<int> mb((size_t)world + 1, nm);
            double total = 0.0;
            auto cost = [&](int m) {
                const double r = rel.mis_to_dof.row_size(m), c = (double)(L.mis_s_off[m + 1] - L.mis_s_off[m]);
                return r * std::min(r, c) * std::max(r, c) + 64.0;
            };
            for (int m = 0; m < nm; ++m) total += cost(m);
            double run = 0.0;
            int rk = 0;
            mb[0] = 0;
            for (int m = 0; m < nm && rk + 1 < world; ++m) {
                run += cost(m);
                while (rk + 1 < world && run >= total * (rk + 1) / world) mb[++rk] = m + 1;
            }
            mb[world] = nm;
            L.d_mis_k.zero(s);
            d_ncols.zero(s);
            mis_svd(s, L.drel, mb[P.rank + 1] - mb[P.rank], max_ctot, io, mb[P.rank]);
            std::vector<long long> off((size_t)world + 1);
            for (int r = 0; r <= world; ++r) off[r] = 8ll * L.mis_u_off[mb[r]];
            SA_REQUIRE(P.allgather(P.allgather_ctx, L.mis_U.p, off.data()) == 0, "all-gather (MIS bases) failed");
            for (int r = 0; r <= world; ++r) off[r] = 8ll * L.mis_s_off[mb[r]];
            SA_REQUIRE(P.allgather(P.allgather_ctx, L.mis_sig.p, off.data()) == 0, "all-gather (singular values) failed");
            for (int r = 0; r <= world; ++r) off[r] = 4ll * mb[r];
            SA_REQUIRE(P.allgather(P.allgather_ctx, L.d_mis_k.p, off.data()) == 0, "all-gather (MIS counts) failed");
            SA_REQUIRE(P.allgather(P.allgather_ctx, d_ncols.p, off.data()) == 0, "all-gather (MIS columns) failed");
        } else {
            // (agglomerates that hold copies of one class's eigenpairs: MISes with identical inputs are decomposed once)
            DBuf<int> d_ev;
            bool any = false;
            for (int v_ : L.ae_evclass) any = any || v_ >= 0;
            if (any && options().eig_dedupe != 0 && !(P.testmesh && lev == 0)) d_ev.from_host(L.ae_evclass, s);
            mis_svd(s, L.drel, nm, max_ctot, io, 0, d_ev.n ? d_ev.p : nullptr);
        }
        { auto t_ = L.d_mis_k.to_host(s); L.mis_k.assign(t_.begin(), t_.end()); }
        { auto t_ = d_ncols.to_host(s); L.mis_ncols.assign(t_.begin(), t_.end()); }
    }
    tm.lap("MIS gather + SVD", lev);
    L.mis_coloff.assign((size_t)nm + 1, 0);
    for (int m = 0; m < nm; ++m) L.mis_coloff[m + 1] = L.mis_coloff[m] + L.mis_k[m];
    L.d_mis_coloff.from_host(L.mis_coloff, s);
    build_P_R(s, L.drel, rel, L.mis_k, L.mis_u_off, L.d_mis_k.p, L.d_mis_coloff.p, L.d_mis_u_off.p,
              L.mis_U.p, L.P, L.R);
    if (lev + 1 < P.num_coarsenings) {     // the next level's representation of the constant vector (P^T P = I)
        L.cvec_next.alloc((size_t)L.R.nrows);
        const double *cv = nullptr;
        DBuf<double> ones;
        if (lev > 0 && H.levels[lev - 1]->cvec_next.n == (size_t)L.A.nrows) cv = H.levels[lev - 1]->cvec_next.p;
        else {
            ones.alloc((size_t)L.A.nrows);
            hipLaunchKernelGGL(fill_kernel, dim3(div_up((long)L.A.nrows, 256)), dim3(256), 0, s, (long)L.A.nrows, ones.p, 1.0);
            cv = ones.p;
        }
        spmv(s, L.R, cv, L.cvec_next.p);
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
    // the host half of the next level's inputs runs beside the Galerkin product (own thread, own
    // stream); with a smoothed prolongator level_galerkin moves P, so it is done afterwards
    std::exception_ptr prep_err;
    std::thread prep_thread;
    struct PrepJoiner { std::thread &t; ~PrepJoiner() { if (t.joinable()) t.join(); } } prep_joiner{prep_thread};
    // (device build first: ~1 ms on the main stream; the host build is the fallback for AEs with more
    // coarse dofs than the kernel's LDS holds)
    constexpr bool host_e2d = false;
    if (lev + 1 < P.num_coarsenings && P.nu_pro[lev] == 0 && !host_e2d) {
        NextPrep &np = L.next_prep;
        np.on_device = coarse_e2d_device(s, L.drel, rel, L.d_mis_k.p, L.d_mis_coloff.p, L.mis_coloff.back(),
                                         L.P.rowptr.p, L.P.val.p, np.d_colpos_ptr, np.d_colpos, np.e2d);
        np.ready = np.on_device;
    }
    if (lev + 1 < P.num_coarsenings && P.nu_pro[lev] == 0 && !L.next_prep.ready) {
        hipStream_t side = side_stream(1);
        SA_HIP_CHECK(hipStreamSynchronize(s));       // P is complete
        const roff_t *prp = L.P.rowptr.p;
        const double *pvl = L.P.val.p;
        const int pnr = L.P.nrows;
        const int64_t pnz = L.P.nnz;
        hipStream_t sd = side;
        prep_thread = std::thread([&L, &prep_err, prp, pvl, pnr, pnz, sd, dev]() {
            try {
                adopt_device(dev);
                set_thread_stream(sd);
                prepare_next_host(L, prp, pvl, pnr, pnz, sd, L.next_prep);
                SA_HIP_CHECK(hipStreamSynchronize(sd));      // nothing of this thread is in flight after the join
            } catch (...) { prep_err = std::current_exception(); }
        });
    }
    // With another spectral level to come the product runs beside that level's element matrices and
    // eigenproblems (they need its size only); SAAMGE_AMD_NO_OVERLAP=1 and the profiled step keep it in line.
    const bool no_overlap = !(options().overlap & 4) || env_serial();
    const bool defer = lev + 1 < P.num_coarsenings && P.nu_pro[lev] == 0 && world == 1 && !no_overlap &&
                       !profiler().enabled;
    if (defer) {
        SA_HIP_CHECK(hipStreamSynchronize(s));       // the operands are complete
        hipStream_t gs = side_stream(2);
        H.galerkin_lev = lev;
        H.galerkin_thread = std::thread([&H, lev, gs, dev]() {
            try {
                adopt_device(dev);
                set_thread_stream(gs);
                level_galerkin(H, lev, true, gs);
                SA_HIP_CHECK(hipStreamSynchronize(gs));
            } catch (...) { H.galerkin_err = std::current_exception(); }
        });
    } else {
        level_galerkin(H, lev, true);
    }
    if (prep_thread.joinable()) prep_thread.join();
    if (prep_err) std::rethrow_exception(prep_err);
    tm.lap(defer ? "P, R (RAP deferred)" : "P, R, RAP", lev);
    if (!P.keep_debug) {
        L.evals.release();
        L.evecs.release();
        L.mis_sig.release();
    }
    // work vectors
    const size_t n = (size_t)L.A.nrows;
    L.x.alloc(n); L.b.alloc(n); L.r.alloc(n); L.t0.alloc(n);
}

// ---------------------------------------------------------------------------------------
// next level's inputs: coarse elements = AEs, coarse element matrices = P_loc^T A_e P_loc
// (agg_create_partitioning_coarse / elmat_parallel)
// ---------------------------------------------------------------------------------------
// Host half of the next level's inputs (everything that needs only the topology, the MIS sizes
// and the numerically non-zero pattern of the tentative prolongator).  Runs on its own thread
// and stream beside the Galerkin product when the prolongator is not smoothed.
static void prepare_next_host(const Level &L, const roff_t *p_rowptr_dev, const double *p_val_dev, int p_nrows,
                              int64_t p_nnz, hipStream_t s, NextPrep &out) {
    fetch_relations_ae_host(const_cast<Relations &>(L.rel), L.drel, s);
    const Relations &rel = L.rel;
    const int nparts = rel.nparts;
    // coarse elem_to_dof = AE_to_dof x pattern(P_tent), first-encounter order
    // (agg_create_rels_except_elem_coarse, amg/src/aggregates.cpp:1510-1514).  The pattern is
    // the *numerically non-zero* entries of P_tent (contrib_tent_insert_simple drops exact
    // zeros, amg/src/contrib.cpp:186-187), so P's values are inspected on the host.
    Table &e2d = out.e2d;
    e2d.ncols = L.mis_coloff.back();
    e2d.I.assign((size_t)nparts + 1, 0);
    hvec<roff_t> p_rowptr((size_t)p_nrows + 1);
    hvec<double> p_val((size_t)p_nnz + 1);
    SA_HIP_CHECK(hipMemcpyAsync(p_rowptr.data(), p_rowptr_dev, sizeof(roff_t) * ((size_t)p_nrows + 1), hipMemcpyDeviceToHost, s));
    if (p_nnz) SA_HIP_CHECK(hipMemcpyAsync(p_val.data(), p_val_dev, sizeof(double) * (size_t)p_nnz, hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    // colpos: for every (AE, MIS) incidence (aligned with AE_to_mis.J) the position of each of
    // the MIS's coarse dofs in the coarse element's dof list
    std::vector<int> &colpos_ptr = out.colpos_ptr;
    colpos_ptr.assign(rel.AE_to_mis.J.size() + 1, 0);
    for (int e = 0; e < nparts; ++e)
        for (int t = rel.AE_to_mis.I[e]; t < rel.AE_to_mis.I[e + 1]; ++t)
            colpos_ptr[(size_t)t + 1] = colpos_ptr[t] + L.mis_k[rel.AE_to_mis.J[t]];
    std::vector<int> &colpos = out.colpos;
    colpos.assign((size_t)colpos_ptr.back(), -1);
    // two passes over the AEs (count, then fill), both split over host threads
    {
        int T = (int)std::thread::hardware_concurrency();
        if (T < 1) T = 1;
        if (T > 16) T = 16;
        if (nparts < 64) T = 1;
        auto walk = [&](int e, std::vector<int> &stamp, int *dst, bool fill) -> int {
            int run = 0;
            const int *misrow = rel.AE_to_mis.row(e);
            const int nmis = rel.AE_to_mis.row_size(e);
            for (int k = rel.AE_to_dof.I[e]; k < rel.AE_to_dof.I[e + 1]; ++k) {
                const int dof = rel.AE_to_dof.J[k];
                const int m = rel.mises[dof];
                const int km = L.mis_k[m];
                if (km == 0) continue;
                const int t = rel.AE_to_mis.I[e] + (int)(std::lower_bound(misrow, misrow + nmis, m) - misrow);
                for (int v = 0; v < km; ++v) {
                    const int cd = L.mis_coloff[m] + v;
                    if (stamp[cd] == e) continue;
                    if (p_val[(size_t)p_rowptr[dof] + v] == 0.0) continue;
                    stamp[cd] = e;
                    if (fill) {
                        colpos[(size_t)colpos_ptr[t] + v] = run;
                        dst[run] = cd;
                    }
                    ++run;
                }
            }
            return run;
        };
        const int dev = current_device();
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t, pass]() {
                    (void)hipSetDevice(dev);
                    std::vector<int> stamp((size_t)e2d.ncols, -1);
                    const int eb = (int)((int64_t)nparts * t / T), ee = (int)((int64_t)nparts * (t + 1) / T);
                    for (int e = eb; e < ee; ++e) {
                        if (pass == 0) e2d.I[e + 1] = walk(e, stamp, nullptr, false);
                        else walk(e, stamp, e2d.J.data() + e2d.I[e], true);
                    }
                });
            for (auto &x : th) x.join();
            if (pass == 0) {
                for (int e = 0; e < nparts; ++e) e2d.I[e + 1] += e2d.I[e];
                e2d.J.resize((size_t)e2d.I[nparts]);
            }
        }
    }
    out.ready = true;
}

static Table prepare_next_level(Hierarchy &H, int lev) {
    Level &L = *H.levels[lev];
    hipStream_t s = H.stream;
    const Relations &rel = L.rel;
    const int nparts = rel.nparts;
    if (!L.next_prep.ready) {
        // (always the TENTATIVE prolongator: the coarse elements are built from mis_tent_interps)
        const DCsr &PT = L.Ptent.nrows ? L.Ptent : L.P;
        NextPrep &np = L.next_prep;
        constexpr bool host_e2d = false;
        if (!host_e2d)
            np.on_device = coarse_e2d_device(s, L.drel, rel, L.d_mis_k.p, L.d_mis_coloff.p, L.mis_coloff.back(),
                                             PT.rowptr.p, PT.val.p, np.d_colpos_ptr, np.d_colpos, np.e2d);
        if (!np.on_device) prepare_next_host(L, PT.rowptr.p, PT.val.p, PT.nrows, PT.nnz, s, L.next_prep);
    }
    Table e2d = std::move(L.next_prep.e2d);
    DBuf<int> d_colpos_ptr, d_colpos;
    if (L.next_prep.on_device) {
        d_colpos_ptr = std::move(L.next_prep.d_colpos_ptr);
        d_colpos = std::move(L.next_prep.d_colpos);
    } else {
        for (int v : L.next_prep.colpos) SA_REQUIRE(v >= 0, "coarse dof with an all-zero prolongator column in an AE");
        d_colpos_ptr.from_host(L.next_prep.colpos_ptr, s);
        d_colpos.from_host(L.next_prep.colpos, s);
    }
    L.next_prep = NextPrep();
    // coarse element matrices
    Level &N = *H.levels[lev + 1];
    std::vector<int64_t> out_off((size_t)nparts + 1, 0);
    for (int e = 0; e < nparts; ++e) {
        const int64_t ke = e2d.row_size(e);
        out_off[e + 1] = out_off[e] + ke * ke;
    }
    N.elmat.off.from_host(out_off, s);
    N.elmat.val.alloc((size_t)out_off[nparts] + 1);
    // The next level's AE tables are a HOST build (7.8 ms on the headline's level 1).  Its inputs are complete here -- the coarse
    // element lists just made, the caller's partition -- and nothing below needs its result: it runs on a thread of its own
    // beside the coarse element matrices (20 ms of device work during which this thread only waits).
    std::exception_ptr topo_err;
    std::thread topo_thread;
    struct TopoJoiner { std::thread &t; ~TopoJoiner() { if (t.joinable()) t.join(); } } topo_joiner{topo_thread};
    if (lev + 1 < H.params.num_coarsenings && lev + 1 < (int)H.coarse_parts.size() && !H.coarse_parts[(size_t)lev + 1].empty() &&
        !env_serial()) {
        const int dev = current_device();
        const int nd_next = L.mis_coloff.back(), np_next = H.nparts_in[(size_t)lev + 1];
        Table *e2d_copy = new Table(e2d);
        const hvec<int> *part_next = &H.coarse_parts[(size_t)lev + 1];
        topo_thread = std::thread([&N, &topo_err, e2d_copy, part_next, nd_next, np_next, dev]() {
            std::unique_ptr<Table> own(e2d_copy);
            try {
                adopt_device(dev);
                build_relations_ae(N.rel, std::move(*own), *part_next, np_next, nd_next, nullptr);
                N.rel_prebuilt = true;
            } catch (...) { topo_err = std::current_exception(); }
        });
    }
    std::vector<int> sizes((size_t)nparts);
    for (int p = 0; p < nparts; ++p) sizes[p] = rel.AE_to_dof.row_size(p);
    // several ranks: each computes the coarse element matrices of its AE range (the ranges of
    // the eigenproblems), then one in-place all-gather
    const int world = H.params.world > 1 ? H.params.world : 1;
    const int ae_lo = world > 1 ? L.ae_begin[H.params.rank] : 0;
    const int ae_hi = world > 1 ? L.ae_begin[H.params.rank + 1] : nparts;
    DBuf<int> d_ae_class;
    for (int ae0 = ae_lo; ae0 < ae_hi;) {
        size_t bytes = 0;
        int cnt = 0;
        while (ae0 + cnt < ae_hi) {
            const size_t n = (size_t)sizes[ae0 + cnt];
            const size_t add = 8 * (n * n + n * (EIG_NB + 8) + n * (size_t)e2d.row_size(ae0 + cnt));
            if (cnt > 0 && bytes + add > H.params.workspace_bytes) break;
            bytes += add;
            ++cnt;
        }
        EigBatch batch;
        eig_batch_alloc(batch, std::vector<int>(sizes.begin() + ae0, sizes.begin() + ae0 + cnt), s);
        std::vector<int64_t> soff((size_t)cnt + 1, 0);
        for (int i = 0; i < cnt; ++i) soff[i + 1] = soff[i] + (int64_t)sizes[ae0 + i] * e2d.row_size(ae0 + i);
        DBuf<int64_t> d_soff;
        d_soff.from_host(soff, s);
        double *scratch = scratch_get(0, (size_t)soff[cnt] + 1);
        int RW = 0;
        const double *rv = nullptr;
        const short *rc = nullptr;
        constexpr bool dense_only = false;
        const RowsSpan span{(int64_t)rel.AE_to_dof.I[ae0] - (int64_t)rel.AE_to_dof.I[ae_lo], (int64_t)rel.AE_to_dof.I[ae_hi] - (int64_t)rel.AE_to_dof.I[ae_lo]};
        if (lev == 0 && !dense_only && ae_sparse_rows(s, L.drel, L.A, L.elmat, ae0, batch, RW, rv, rc, &span)) {
            // fine level: straight from the sparse rows of the AE matrices
            int kmax = 0;
            for (int km : L.mis_k) kmax = std::max(kmax, km);
            // (classes of the agglomerates' sparse rows, where the eigenproblem stage found them for every agglomerate of the range)
            bool have_classes = options().eig_dedupe != 0 && (int)L.ae_class.size() == nparts;
            for (int i = ae0; have_classes && i < ae0 + cnt; ++i) have_classes = L.ae_class[i] >= 0;
            if (have_classes && !d_ae_class.n) d_ae_class.from_host(L.ae_class, s);
            coarse_elmats_sparse(s, L.drel, ae0, batch, RW, rv, rc, L.d_mis_k.p, L.d_mis_u_off.p, L.mis_U.p,
                                 d_colpos_ptr.p, d_colpos.p, N.elmat.off.p, N.elmat.val.p, scratch, d_soff.p, kmax,
                                 have_classes ? d_ae_class.p : nullptr);
        } else {
            ae_build(s, L.drel, lev == 0 ? &L.A : nullptr, L.elmat, ae0, batch, false, nullptr);
            coarse_elmats(s, L.drel, ae0, batch, L.d_mis_k.p, L.d_mis_u_off.p, L.mis_U.p, d_colpos_ptr.p,
                          d_colpos.p, N.elmat.off.p, N.elmat.val.p, scratch, d_soff.p);
        }
        // (single rank, last chunk: no wait -- everything that reads the element matrices follows on this stream,
        // the buffers released here are stream-ordered, and the host goes on to the next level's topology while
        // the kernel runs)
        if (world > 1 || ae0 + cnt < ae_hi) SA_HIP_CHECK(hipStreamSynchronize(s));
        ae0 += cnt;
    }
    if (world > 1) {
        std::vector<long long> off((size_t)world + 1);
        for (int r = 0; r <= world; ++r) off[r] = 8ll * out_off[L.ae_begin[r]];
        SA_REQUIRE(H.params.allgather(H.params.allgather_ctx, N.elmat.val.p, off.data()) == 0,
                   "all-gather (coarse element matrices) failed");
    }
    if (topo_thread.joinable()) topo_thread.join();
    if (topo_err) std::rethrow_exception(topo_err);
    return e2d;
}

// ---------------------------------------------------------------------------------------
// coarsest solver: PCG on the coarsest operator, preconditioned by its own polynomial smoother
// (the reference's parallel "coarse direct" is likewise an inner PCG to 1e-16,
// amg/src/tg.cpp:998-1003)
// ---------------------------------------------------------------------------------------
static const DCsr &coarsest_op(const Hierarchy &H) { return H.levels.back()->Ac; }

static void setup_coarse_solver(Hierarchy &H) {
    DCsr &Ac = H.levels.back()->Ac;
    hipStream_t s = H.stream;
    const size_t n = (size_t)Ac.nrows;
    build_sell(s, Ac);
    H.coarse_kind = 2;
    H.c_dinv.alloc(n);
    H.c_r.alloc(n); H.c_z.alloc(n); H.c_d.alloc(n); H.c_q.alloc(n); H.c_t0.alloc(n); H.c_t1.alloc(n);
    if (n) {
        DBuf<double> tmp(n);
        build_dinv_neg(s, Ac, tmp.p, H.c_dinv.p);
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
    H.c_roots = sas_poly_roots(H.levels.back()->nu_relax);
    // Dense direct solve (the reference's serial `--coarse-direct`, amg/src/tg.cpp:979-1014) as an explicit
    // inverse when the coarsest operator is small enough (coarse_solver 0 = auto: n <= 8192, i.e. up to 512 MiB;
    // 1 = always): at n = 3317 the inverse costs a few ms once and 20 us per V-cycle, the inner PCG ~2.4 ms per
    // V-cycle.  A non-positive pivot (semi-definite operator) falls back to the inner PCG.
    const int want = H.params.coarse_solver;
    constexpr long dense_max = 8192;
    H.c_bt = BlockTri();
    if (n && (want == 1 || (want == 0 && (long)n <= dense_max)) && n <= 16384) {
        if (dense_inverse_spd(s, Ac, H.c_L)) H.coarse_kind = 1;
        else H.c_L.release();
    } else if (n && (want == 1 || want == 3)) {
        // a DIRECT solve was asked for (the reference's coarse_direct, src/tg.cpp:990-996) on an operator beyond one dense
        // inverse: block-tridiagonal elimination over a level structure of its graph (blocktri.hip)
        if (blocktri_factor(s, Ac, H.c_bt)) H.coarse_kind = 3;
    }
}

static inline RowRange rows_of(const Level::Dist *D) {
    RowRange rr;
    if (D) { rr.row0 = D->row0; rr.nrows = D->nloc; }
    return rr;
}

// x = poly(b) starting from x = 0; result lands in x, `tmp` is the ping-pong partner.
// Row-partitioned (D != null): own rows only; the halo of the result is NOT refreshed (the next
// operator application does that, halo_then).
static void smooth_from_zero(Hierarchy &H, const DCsr &A, const double *dinv,
                             const std::vector<double> &roots, const double *b, double *x, double *tmp,
                             Level::Dist *D = nullptr) {
    hipStream_t s = H.stream;
    const int deg = (int)roots.size();
    const int off = D ? D->row0 : 0, nl = D ? D->nloc : A.nrows;
    double *cur = ((deg - 1) % 2 == 0) ? x : tmp;
    double *oth = (cur == x) ? tmp : x;
    smooth_first(s, nl, dinv + off, b + off, cur + off, 1.0 / roots[0]);
    for (int i = 1; i < deg; ++i) {
        const double scale = 1.0 / roots[i];
        halo_then(H, D, cur, [&](hipStream_t q, RowRange rr) { smooth_step(q, A, dinv, b, cur, oth, scale, rr); });
        std::swap(cur, oth);
    }
}

// x += M^-1 (b - A x).  Row-partitioned: x is valid on the own rows on entry and on return (every step
// refreshes the halo it reads).
static void smooth_inplace(Hierarchy &H, const DCsr &A, const double *dinv,
                           const std::vector<double> &roots, const double *b, double *x, double *tmp,
                           Level::Dist *D = nullptr) {
    hipStream_t s = H.stream;
    const int deg = (int)roots.size();
    const int off = D ? D->row0 : 0, nl = D ? D->nloc : A.nrows;
    double *cur = x, *oth = tmp;
    for (int i = 0; i < deg; ++i) {
        const double scale = 1.0 / roots[i];
        halo_then(H, D, cur, [&](hipStream_t q, RowRange rr) { smooth_step(q, A, dinv, b, cur, oth, scale, rr); });
        std::swap(cur, oth);
    }
    if (cur != x) vec_copy(s, nl, cur + off, x + off);
}

static double read_scalar(hipStream_t s, const double *dptr) {
    double v;
    SA_HIP_CHECK(hipMemcpyAsync(&v, dptr, sizeof(double), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    return v;
}

// The PCG loop shared by the outer solve and the coarsest solver (MFEM CGSolver::Mult order
// of operations == kalchev_pcg, amg/src/mfem_addons.cpp:106-248).  Row-partitioned (D != null):
// vector updates and inner products run on the own rows, the products are summed over ranks,
// the search direction's halo is refreshed before each A d.
static int pcg_loop(Hierarchy &H, const DCsr &A, const std::function<void(const double *, double *)> &prec,
                    const double *b, double *x, double *r, double *z, double *d, double *q,
                    double rel_tol, double abs_tol, int max_iter, bool squared, bool zero_guess,
                    int *converged, double *hist, double *sc, Level::Dist *D = nullptr) {
    hipStream_t s = H.stream;
    const int off = D ? D->row0 : 0, n = D ? D->nloc : A.nrows;
    auto pdot = [&](const double *u, const double *v, double *out) {
        dot(s, n, u + off, v + off, H.partials.p, out);
        if (D) dist_allreduce(H, out, 1);
    };
    if (zero_guess) {
        vec_zero(s, n, x + off);
        vec_copy(s, n, b + off, r + off);
    } else {
        halo_then(H, D, x, [&](hipStream_t q, RowRange rq) { spmv_residual(q, A, x, b, r, rq); });
    }
    prec(r, z);
    vec_copy(s, n, z + off, d + off);
    pdot(d, r, sc + 0);
    const double nom0 = read_scalar(s, sc + 0);
    if (hist) hist[0] = nom0;
    const double r0 = squared ? std::max(nom0 * rel_tol * rel_tol, abs_tol * abs_tol)
                              : std::max(nom0 * rel_tol, abs_tol);
    if (converged) *converged = 0;
    if (nom0 <= r0) {
        if (converged) *converged = 1;
        return 0;
    }
    halo_then(H, D, d, [&](hipStream_t st, RowRange rq) { spmv(st, A, d, q, rq); });
    pdot(q, d, sc + 1);
    if (read_scalar(s, sc + 1) == 0.0) return 0;
    int i = 1;
    int final_iter = max_iter;
    for (;;) {
        pcg_update_xr(s, n, sc, x + off, r + off, d + off, q + off);
        prec(r, z);
        pdot(r, z, sc + 2);
        const double betanom = read_scalar(s, sc + 2);
        if (hist) hist[i] = betanom;
        if (betanom < r0) {
            if (converged) *converged = 1;
            final_iter = i;
            break;
        }
        if (++i > max_iter) break;
        pcg_update_d(s, n, sc, d + off, z + off);
        halo_then(H, D, d, [&](hipStream_t st, RowRange rq) { spmv(st, A, d, q, rq); });
        pdot(d, q, sc + 1);
        SA_HIP_CHECK(hipMemcpyAsync(sc + 0, sc + 2, sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    return final_iter;
}

static void coarse_solve(Hierarchy &H, const double *rc, double *xc) {
    const DCsr &Ac = coarsest_op(H);
    if (Ac.nrows == 0) return;
    if (H.user_coarse_solve) {      // tg_data_t::coarse_solver assigned by the caller (host solver)
        const size_t n = (size_t)Ac.nrows;
        hvec<double> hr(n), hx(n, 0.0);
        SA_HIP_CHECK(hipMemcpyAsync(hr.data(), rc, 8 * n, hipMemcpyDeviceToHost, H.stream));
        SA_HIP_CHECK(hipStreamSynchronize(H.stream));
        SA_REQUIRE(H.user_coarse_solve(H.user_coarse_ctx, (int)n, hr.data(), hx.data()) == 0, "user coarse solver failed");
        SA_HIP_CHECK(hipMemcpyAsync(xc, hx.data(), 8 * n, hipMemcpyHostToDevice, H.stream));
        SA_HIP_CHECK(hipStreamSynchronize(H.stream));
        H.last_coarse_iters = 0;
        return;
    }
    if (H.coarse_kind == 1) {
        // x = X b and one step of iterative refinement, x += X (b - Ac x): the elimination's round-off
        // (~cond(Ac) eps) is pushed to the level of the residual evaluation
        const int n = Ac.nrows;
        dense_symv(H.stream, n, H.c_L.p, rc, xc, false);
        spmv_residual(H.stream, Ac, xc, rc, H.c_r.p);
        dense_symv(H.stream, n, H.c_L.p, H.c_r.p, xc, true);
        H.last_coarse_iters = 0;
        return;
    }
    if (H.coarse_kind == 3) {
        blocktri_solve(H.stream, Ac, H.c_bt, rc, xc);
        H.last_coarse_iters = 0;
        return;
    }
    auto prec = [&](const double *r, double *z) {
        smooth_from_zero(H, Ac, H.c_dinv.p, H.c_roots, r, z, H.c_t0.p);
    };
    // scalar slots 4.. so the outer PCG's scalars (slots 0..2) survive
    int conv = 0;
    H.last_coarse_iters = pcg_loop(H, Ac, prec, rc, xc, H.c_r.p, H.c_z.p, H.c_d.p, H.c_q.p,
                                   H.params.coarse_rtol, 0.0, H.params.coarse_max_iter, false, true,
                                   &conv, nullptr, H.scal.p + 4);
}

static Level::Dist *dist_of(Level &L) { return L.dist.on ? &L.dist : nullptr; }

// a caller's smoother (saamge_amd_set_smoother; the reference's smpr_ft plug, src/tg.cpp:113,131): x += M^-1 (b - A x)
// on host copies.  zero_start: the cycle's x is not initialised yet, the callback receives x = 0.
static void user_smooth(Hierarchy &H, int level, int (*fn)(void *, int, int, const double *, double *), void *ctx,
                        const double *b, double *x, bool zero_start) {
    Level &L = *H.levels[level];
    SA_REQUIRE(!L.dist.on, "a caller's smoother cannot run on a level that is row-partitioned over several ranks");
    const size_t n = (size_t)L.A.nrows;
    hvec<double> hb(n), hx(n, 0.0);
    SA_HIP_CHECK(hipMemcpyAsync(hb.data(), b, 8 * n, hipMemcpyDeviceToHost, H.stream));
    if (!zero_start) SA_HIP_CHECK(hipMemcpyAsync(hx.data(), x, 8 * n, hipMemcpyDeviceToHost, H.stream));
    SA_HIP_CHECK(hipStreamSynchronize(H.stream));
    SA_REQUIRE(fn(ctx, level, (int)n, hb.data(), hx.data()) == 0, "the caller's smoother failed");
    SA_HIP_CHECK(hipMemcpyAsync(x, hx.data(), 8 * n, hipMemcpyHostToDevice, H.stream));
    SA_HIP_CHECK(hipStreamSynchronize(H.stream));
}

// One V(1,1)-cycle from x = 0 (tg_cycle_atb, amg/src/tg.cpp:91-132).  On a row-partitioned
// level b is read and x is written on the own rows only.
static void vcycle_rec(Hierarchy &H, int level, const double *b, double *x) {
    Level &L = *H.levels[level];
    hipStream_t s = H.stream;
    Level::Dist *D = dist_of(L);
    const bool last = (level + 1 == (int)H.levels.size());
    const Hierarchy::UserSmoother us = level < (int)H.user_smoothers.size() ? H.user_smoothers[level] : Hierarchy::UserSmoother();
    if (us.pre) user_smooth(H, level, us.pre, us.ctx, b, x, true);
    else smooth_from_zero(H, L.A, L.dinv_neg.p, L.roots, b, x, L.t0.p, D);   // pre_smoother, x0 = 0
    halo_then(H, D, x, [&](hipStream_t q, RowRange rr) { spmv_residual(q, L.A, x, b, L.r.p, rr); });   // res = b - A x
    double *rc = last ? H.c_b.p : H.levels[level + 1]->b.p;
    double *xc = last ? H.c_x.p : H.levels[level + 1]->x.p;
    spmv(s, L.R, L.r.p, rc);                                            // resc = R res
    if (D) {      // res is zero outside the own rows: partial sums
        constexpr bool no_rs = false;
        // a row-partitioned coarser level reads the restricted residual on its own rows only: reduce-scatter
        if (!last && !no_rs && H.levels[level + 1]->dist.on) dist_reduce_scatter_rows(H, H.levels[level + 1]->dist, rc);
        else dist_allreduce(H, rc, L.R.nrows);
    }
    if (last) {
        coarse_solve(H, rc, xc);
    } else {
        vcycle_rec(H, level + 1, rc, xc);
        Level &N = *H.levels[level + 1];
        if (N.dist.on) dist_allgather_rows(H, N.dist, xc);
    }
    spmv_add(s, L.P, xc, x, rows_of(D));                                // x += P xc
    if (us.post) user_smooth(H, level, us.post, us.ctx, b, x, false);
    else smooth_inplace(H, L.A, L.dinv_neg.p, L.roots, b, x, L.t0.p, D);     // post_smoother
}

void smoother_apply(Hierarchy &H, int level, const double *b, double *x) {
    Level &L = *H.levels[level];
    Level::Dist *D = dist_of(L);
    smooth_inplace(H, L.A, L.dinv_neg.p, L.roots, b, x, L.t0.p, D);
    if (D) dist_allgather_rows(H, *D, x);
}

void vcycle_apply(Hierarchy &H, int level, const double *b, double *x) {
    vcycle_rec(H, level, b, x);
    Level &L = *H.levels[level];
    if (L.dist.on) dist_allgather_rows(H, L.dist, x);
}

int pcg_solve(Hierarchy &H, const double *b, double *x, double rel_tol, double abs_tol,
              int max_iter, int squared_tol, int zero_guess, int *converged, double *hist) {
    Level &L0 = *H.levels[0];
    Level::Dist *D = dist_of(L0);
    auto prec = [&](const double *r, double *z) { vcycle_rec(H, 0, r, z); };
    PhaseTimer tm(H.stream);
    struct Lap { PhaseTimer &t; ~Lap() { t.lap("TOTAL pcg_solve", 0); } } lap{tm};
    const int it = pcg_loop(H, L0.A, prec, b, x, H.pcg_r.p, H.pcg_z.p, H.pcg_d.p, H.pcg_q.p, rel_tol,
                            abs_tol, max_iter, squared_tol != 0, zero_guess != 0, converged, hist,
                            H.scal.p, D);
    if (D) dist_allgather_rows(H, *D, x);  // every rank returns the full solution
    return it;
}

// ---------------------------------------------------------------------------------------
// corrected null-space level (CorrectNullspace, amg/src/solve.cpp:52-164, configured by
// amg/src/ml.cpp:225-236): one more two-grid level under the coarsest spectral operator.
// interp = scaling_P (amg/src/contrib.cpp:655-668, amg/src/interp.cpp:842-909): one column per
// MIS that has coarse dofs, holding the normalised coefficients of the constant vector in the
// MIS's orthonormal basis (x = U^T 1 / |U^T 1|).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void scaling_p_kernel(const int *__restrict__ mis2d_I, const int *__restrict__ k,
                                                       const int *__restrict__ coloff,
                                                       const int64_t *__restrict__ u_off,
                                                       const double *__restrict__ U,
                                                       const int *__restrict__ active, roff_t *__restrict__ prow,
                                                       int *__restrict__ pcol, double *__restrict__ pval) {
    extern __shared__ double xs[];
    const int m = blockIdx.x, lane = threadIdx.x;
    const int km = k[m];
    if (km == 0) return;
    const int r = mis2d_I[m + 1] - mis2d_I[m];
    const double *Um = U + u_off[m];
    double nn = 0.0;
    for (int v = 0; v < km; ++v) {
        double sum = 0.0;
        for (int i = lane; i < r; i += 64) sum += Um[(size_t)v * r + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) xs[v] = sum;
        nn = fma(sum, sum, nn);
    }
    __syncthreads();
    const double inv = 1.0 / sqrt(nn);
    for (int v = lane; v < km; v += 64) {
        const int row = coloff[m] + v;
        prow[row] = row;
        pcol[row] = active[m];
        pval[row] = xs[v] * inv;
    }
}

static void add_nullspace_level(Hierarchy &H) {
    hipStream_t s = H.stream;
    Level &L = *H.levels.back();
    const int nm = L.rel.num_mises;
    const int nc = L.Ac.nrows;
    SA_REQUIRE(!H.params.testmesh, "the corrected null-space level is not available with the mltest fixture vectors");
    std::vector<int> active((size_t)nm, -1);
    int nact = 0, kmax = 1;
    for (int m = 0; m < nm; ++m)
        if (L.mis_k[m] > 0) { active[m] = nact++; kmax = std::max(kmax, L.mis_k[m]); }
    H.levels.emplace_back(new Level);
    Level &N = *H.levels.back();
    N.A = std::move(L.Ac);
    N.theta = 0.0;
    N.nu_relax = 3;
    N.roots = sas_poly_roots(3);
    build_sell(s, N.A);
    N.dinv_neg.alloc((size_t)nc);
    {
        DBuf<double> tmp((size_t)nc);
        build_dinv_neg(s, N.A, tmp.p, N.dinv_neg.p);
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
    N.P.nrows = nc;
    N.P.ncols = nact;
    N.P.nnz = nc;
    N.P.rowptr.alloc((size_t)nc + 1);
    N.P.col.alloc((size_t)nc + 1);
    N.P.val.alloc((size_t)nc + 1);
    DBuf<int> d_active;
    d_active.from_host(active, s);
    hipLaunchKernelGGL(scaling_p_kernel, dim3(nm), dim3(64), sizeof(double) * (size_t)kmax, s, L.drel.mis2d_I.p,
                       L.d_mis_k.p, L.d_mis_coloff.p, L.d_mis_u_off.p, L.mis_U.p, d_active.p, N.P.rowptr.p,
                       N.P.col.p, N.P.val.p);
    SA_HIP_CHECK(hipGetLastError());
    const roff_t nc_off = nc;
    SA_HIP_CHECK(hipMemcpyAsync(N.P.rowptr.p + nc, &nc_off, sizeof(roff_t), hipMemcpyHostToDevice, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    N.P.lanes_per_row = 1;
    csr_transpose(s, N.P, N.R);
    DCsr AP;
    spgemm(s, N.A, N.P, nullptr, nullptr, 1.0, 0.0, AP);
    spgemm(s, N.R, AP, nullptr, nullptr, 1.0, 0.0, N.Ac);
    N.rel.ND = nc;
    N.rel.nparts = nact;
    N.x.alloc((size_t)nc); N.b.alloc((size_t)nc); N.r.alloc((size_t)nc); N.t0.alloc((size_t)nc);
}

// ---------------------------------------------------------------------------------------
// ml_produce_data
// ---------------------------------------------------------------------------------------
Hierarchy *hierarchy_create(int n, const void *Arow, int rowptr_bits, const int *Acol, const double *Aval, int NE,
                            int nde, const int *elem_to_dof, const double *elmat,
                            const signed char *bdr, const int *const *partitions,
                            const int *nparts, const Params &p, hipStream_t stream, std::unique_ptr<DistIn> dist_inputs) {
    SA_REQUIRE(p.num_coarsenings >= 1 && p.num_coarsenings < MAX_LEVELS, "bad number of coarsenings");
    ae_rows_new_build();     // nothing cached from an earlier hierarchy is reused
    // element-free mode: elements = dofs (identity elem_to_dof generated here), no element matrices
    DBuf<int> iota_e2d;   // (moved into the hierarchy below: the device topology views it)
    if (p.algebraic) {
        NE = n;
        nde = 1;
        iota_e2d.alloc((size_t)n);
        hipLaunchKernelGGL(iota_int_kernel, dim3(div_up(n, 256)), dim3(256), 0, stream, (long)n, iota_e2d.p);
        SA_HIP_CHECK(hipGetLastError());
        elem_to_dof = iota_e2d.p;
        elmat = nullptr;
        bdr = nullptr;
    }
    SA_REQUIRE(n > 0 && NE > 0 && nde > 0, "empty problem");
    PhaseTimer tm_all(stream), tm0(stream);
    std::unique_ptr<Hierarchy> Hp(new Hierarchy);
    Hierarchy &H = *Hp;
    H.params = p;
    H.stream = stream;
    H.device = current_device();
    H.own_e2d = std::move(iota_e2d);
    H.dist_in = std::move(dist_inputs);
    hipStream_t s = stream;
    H.scal.alloc(8);
    H.partials.alloc(1024);
    for (int l = 0; l < p.num_coarsenings; ++l) H.levels.emplace_back(new Level);
    // level 0 inputs
    Level &L0 = *H.levels[0];
    {
        import_rowptr(L0.A.rowptr, Arow, rowptr_bits, (size_t)n + 1, s);
        roff_t nnz = 0;
        SA_HIP_CHECK(hipMemcpyAsync(&nnz, L0.A.rowptr.p + n, sizeof(roff_t), hipMemcpyDeviceToHost, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
        L0.A.nrows = L0.A.ncols = n;
        L0.A.nnz = nnz;
        import_array(L0.A.col, Acol, (size_t)nnz, s);
        import_array(L0.A.val, Aval, (size_t)nnz, s);
        finish_csr(L0.A);
    }
    {
        L0.elmat.off.alloc((size_t)NE + 1);
        if (H.dist_in) {
            // per-rank inputs: only the matrices of the rank's own elements exist (they are never exchanged); the offsets of
            // the other elements stay 0 -- no kernel of this rank touches an agglomerate made of them
            L0.elmat.off.zero(s);
            hipLaunchKernelGGL(iota64_kernel, dim3(div_up((long)H.dist_in->NE_loc + 1, 256)), dim3(256), 0, s, (long)H.dist_in->NE_loc + 1,
                               (int64_t)nde * nde, L0.elmat.off.p + H.dist_in->elem0);
            SA_HIP_CHECK(hipGetLastError());
            import_array(L0.elmat.val, elmat, (size_t)H.dist_in->NE_loc * nde * nde, s);
            L0.elmat.first = H.dist_in->elem0;
        } else {
        hipLaunchKernelGGL(iota64_kernel, dim3(div_up((long)NE + 1, 256)), dim3(256), 0, s, (long)NE + 1,
                           (int64_t)nde * nde, L0.elmat.off.p);
        SA_HIP_CHECK(hipGetLastError());
        if (!p.algebraic) import_array(L0.elmat.val, elmat, (size_t)NE * nde * nde, s);
        }
        L0.elmat.nde = nde;
        L0.elmat.algebraic = p.algebraic;
    }
    // Level-0 topology inputs that are device-resident stay there (device build of the AE tables)
    constexpr bool host_topo = false;      // (host inputs take the host build; device inputs the device build)
    DeviceInputs din;
    const bool dev_inputs = !host_topo && is_device_ptr(elem_to_dof) && is_device_ptr(partitions[0]) &&
                            (!bdr || is_device_ptr(bdr));
    Table e2d;
    hvec<signed char> bdr_h;
    if (dev_inputs) {
        din.e2d = elem_to_dof;
        din.part = partitions[0];
        din.bdr = bdr;
        din.NE = NE;
        din.nde = nde;
    } else {
        auto J = fetch_host(elem_to_dof, (size_t)NE * nde, s);
        e2d.J = std::move(J);
        e2d.I.resize((size_t)NE + 1);
        for (int e = 0; e <= NE; ++e) e2d.I[e] = e * nde;
        e2d.ncols = n;
        if (bdr) bdr_h = fetch_host(bdr, (size_t)n, s);
    }
    H.coarse_parts.resize((size_t)p.num_coarsenings);
    H.nparts_in.assign(nparts, nparts + p.num_coarsenings);
    {
        int ne = nparts[0];
        for (int lev = 1; lev < p.num_coarsenings; ++lev) {       // (level-lev elements = level-(lev - 1) agglomerates)
            H.coarse_parts[(size_t)lev] = fetch_host(partitions[lev], (size_t)ne, s);
            ne = nparts[lev];
        }
    }
    tm0.lap("inputs fetched/imported", 0);
    int n_elem = NE;
    for (int lev = 0; lev < p.num_coarsenings; ++lev) {
        hvec<int> part;
        if (lev > 0) part = H.coarse_parts[(size_t)lev];
        else if (!dev_inputs) part = fetch_host(partitions[lev], (size_t)n_elem, s);
        const bool tag_levels = (options().debug & 4) != 0;
        profiler().level_tag = tag_levels ? lev : 0;
        build_level(H, lev, std::move(e2d), part, nparts[lev], (lev == 0 && bdr && !dev_inputs) ? bdr_h.data() : nullptr,
                    (lev == 0 && dev_inputs) ? &din : nullptr);
        Level &L = *H.levels[lev];
        if (lev + 1 < p.num_coarsenings) {
            PhaseTimer tm(s);
            e2d = prepare_next_level(H, lev);
            tm.lap("next-level elements", lev);
            Level &N = *H.levels[lev + 1];
            if (H.galerkin_lev == lev) {       // product still running: the size is known, the arrays follow at the join
                N.A = DCsr();
                N.A.nrows = N.A.ncols = L.mis_coloff.back();
            } else {
                N.A = std::move(L.Ac);  // A_{l+1} = Ac_l  (amg/src/ml.cpp:134)
            }
            n_elem = L.rel.nparts;
        }
    }
    profiler().level_tag = 0;
    join_galerkin(H);
    tm0 = PhaseTimer(s);
    if (p.correct_nullspace) {
        add_nullspace_level(H);
        tm0.lap("corrected null-space level", 0);
    }
    setup_coarse_solver(H);
    const size_t nc = (size_t)coarsest_op(H).nrows;
    H.c_b.alloc(nc);
    H.c_x.alloc(nc);
    H.pcg_r.alloc((size_t)n); H.pcg_z.alloc((size_t)n); H.pcg_d.alloc((size_t)n); H.pcg_q.alloc((size_t)n);
    SA_HIP_CHECK(hipStreamSynchronize(s));
    tm0.lap("coarse solver + vectors", 0);
    if (p.world > 1 && p.alltoallv) {
        for (int lev = 0; lev < p.num_coarsenings; ++lev)
            if (!dist_setup_level(H, lev)) break;
        tm0.lap("row-partitioned solve: halo lists", 0);
    }
    tm_all.lap("TOTAL ml_produce_data", 0);
    if (env_timing()) {
        long nm_ = 0, nf_ = 0;
        size_t mb_ = 0;
        dev_pool_counts(&nm_, &nf_, &mb_, true);
        std::fprintf(stderr, "TIMING: device pool: %ld hipMalloc (%.1f MB), %ld hipFree of cached blocks, %.1f GB idle\n", nm_, mb_ / 1048576.0, nf_,
                     dev_pool_idle_bytes() / 1073741824.0);
    }
    return Hp.release();
}

// adapt_update_operators (amg/src/adapt.cpp:171-219): the matrix changed (same sparsity, same
// topology): keep every interpolation (the eigenproblems are NOT solved again), refresh the
// smoother diagonals, re-smooth P where nu_pro > 0, rebuild all Galerkin operators and the coarsest
// solver.
void hierarchy_update_operators(Hierarchy &H, const double *new_val) {
    hipStream_t s = H.stream;
    const int nspec = H.params.num_coarsenings;
    Level &L0 = *H.levels[0];
    if (new_val && new_val != L0.A.val.p) {
        if (is_device_ptr(new_val)) {
            L0.A.val.view(const_cast<double *>(new_val), (size_t)L0.A.nnz);   // device arrays are used in place
        } else {
            if (!L0.A.val.owned) L0.A.val.alloc((size_t)L0.A.nnz);           // never write into the caller's array
            SA_HIP_CHECK(hipMemcpyAsync(L0.A.val.p, new_val, sizeof(double) * (size_t)L0.A.nnz,
                                        hipMemcpyHostToDevice, s));
            SA_HIP_CHECK(hipStreamSynchronize(s));
        }
    }
    for (int lev = 0; lev < nspec; ++lev) {
        Level &L = *H.levels[lev];
        build_sell(s, L.A);
        {
            DBuf<double> tmp((size_t)L.A.nrows);
            build_dinv_neg(s, L.A, tmp.p, L.dinv_neg.p);
            build_dinv_codes(s, L.A, L.dinv_neg.p);
            SA_HIP_CHECK(hipStreamSynchronize(s));
        }
        L.Ac = DCsr();
        level_galerkin(H, lev, false);
        if (lev + 1 < (int)H.levels.size()) H.levels[lev + 1]->A = std::move(L.Ac);
    }
    if (H.params.correct_nullspace) {   // the scaling_P level: same interpolation, new operators
        Level &N = *H.levels.back();
        build_sell(s, N.A);
        DBuf<double> tmp((size_t)N.A.nrows);
        build_dinv_neg(s, N.A, tmp.p, N.dinv_neg.p);
        SA_HIP_CHECK(hipStreamSynchronize(s));
        DCsr AP;
        spgemm(s, N.A, N.P, nullptr, nullptr, 1.0, 0.0, AP);
        N.Ac = DCsr();
        spgemm(s, N.R, AP, nullptr, nullptr, 1.0, 0.0, N.Ac);
    }
    H.c_L.release();
    setup_coarse_solver(H);
    SA_HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace saamge_amd
