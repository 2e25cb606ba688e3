// CPU RESTATEMENT -- TEST INFRASTRUCTURE + bench.py's cpu_baseline ("port").  NOT part of the product:
// nothing under saamge_amd/ links, loads or calls this file.
//
// A threaded C++ restatement of the reference's setup + solve hot path, the
// way the reference runs it on a CPU node: one agglomerate per core (the reference is MPI-parallel
// over AEs, one rank per core), LAPACK dsygvx / dgesvd for the local problems
// (src/xpacks.cpp:222-314, :494-589), sparse RAP, polynomial-smoothed V-cycle, PCG.  Same
// algorithms, orders and thresholds as oracle/saamge_oracle.py (which is pinned by the reference's
// ctest iteration counts, tests/test_oracle_kat.py); tests/test_cpu_ref.py checks the two against
// each other.  The reference itself needs MFEM + hypre and cannot be built here.
//
// Scope: the default configuration of the headline benchmark -- element-based spectral AMGe,
// nu_pro = 0, no corrected null-space level, exact coarsest solve.
// Citations are into /root/reference/amg.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

#ifndef LAPACK_NAME
#define LAPACK_NAME(x) x##_
#endif
extern "C" {
void LAPACK_NAME(dsygvx)(const int *itype, const char *jobz, const char *range, const char *uplo, const int *n,
                         double *a, const int *lda, double *b, const int *ldb, const double *vl, const double *vu,
                         const int *il, const int *iu, const double *abstol, int *m, double *w, double *z,
                         const int *ldz, double *work, const int *lwork, int *iwork, int *ifail, int *info);
void LAPACK_NAME(dgesvd)(const char *jobu, const char *jobvt, const int *m, const int *n, double *a, const int *lda,
                         double *s, double *u, const int *ldu, double *vt, const int *ldvt, double *work,
                         const int *lwork, int *info);
void LAPACK_NAME(dpotrf)(const char *uplo, const int *n, double *a, const int *lda, int *info);
void LAPACK_NAME(dpbtrf)(const char *uplo, const int *n, const int *kd, double *ab, const int *ldab, int *info);
void LAPACK_NAME(dpbtrs)(const char *uplo, const int *n, const int *kd, const int *nrhs, const double *ab, const int *ldab,
                         double *b, const int *ldb, int *info);
void LAPACK_NAME(dpotrs)(const char *uplo, const int *n, const int *nrhs, const double *a, const int *lda, double *b,
                         const int *ldb, int *info);
double LAPACK_NAME(dlamch)(const char *cmach);
#ifdef BLAS_SET_THREADS
void BLAS_SET_THREADS(int);
#endif
}

namespace {

constexpr int BETWEEN = 0x01;     // AGG_BETWEEN_AES_FLAG, inc/aggregates.hpp:102
constexpr int ESS = 0x02;         // AGG_ON_ESS_DOMAIN_BORDER_FLAG, inc/aggregates.hpp:103
constexpr double SVD_EPS = 1e-10; // ContribTent::svd_eps, src/contrib.cpp:61
constexpr double DIFF_EPS = 1e-10;// GLOBAL.diff_eps, inc/config.hpp:68

int g_threads = 1;
bool g_verbose = false;
void note(const char *what, long long v = 0) { if (g_verbose) { std::fprintf(stderr, "cpu_ref: %s %lld\n", what, v); std::fflush(stderr); } }

void parallel_for(int64_t n, const std::function<void(int64_t, int)> &fn) {   // dynamic, one item at a time
    const int T = (int)std::min<int64_t>(g_threads, std::max<int64_t>(n, 1));
    if (T <= 1) {
        for (int64_t i = 0; i < n; ++i) fn(i, 0);
        return;
    }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n) break;
                fn(i, t);
            }
        });
    for (auto &x : th) x.join();
}
void parallel_ranges(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn) {   // static contiguous ranges
    const int T = (int)std::min<int64_t>(g_threads, std::max<int64_t>(n / 4096, 1));
    if (T <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { fn(n * t / T, n * (t + 1) / T, t); });
    for (auto &x : th) x.join();
}

// ---- mfem::Table algebra (third-party semantics restated) ------------------------------------
struct Table {
    std::vector<int> I, J;
    int ncols = 0;
    int nrows() const { return (int)I.size() - 1; }
    const int *row(int i) const { return J.data() + I[i]; }
    int size(int i) const { return I[i + 1] - I[i]; }
};
Table transpose(const Table &T) {      // mfem::Transpose: row j lists the i with j in row i, ascending i
    Table R;
    R.ncols = T.nrows();
    R.I.assign((size_t)T.ncols + 1, 0);
    for (int v : T.J) R.I[(size_t)v + 1]++;
    for (int i = 0; i < T.ncols; ++i) R.I[i + 1] += R.I[i];
    R.J.resize(T.J.size());
    std::vector<int> pos(R.I.begin(), R.I.end() - 1);
    for (int i = 0; i < T.nrows(); ++i)
        for (int k = T.I[i]; k < T.I[i + 1]; ++k) R.J[pos[T.J[k]]++] = i;
    return R;
}
Table mult(const Table &A, const Table &B) {   // mfem::Mult: union of B's rows in first-encounter order
    Table C;
    C.ncols = B.ncols;
    C.I.assign((size_t)A.nrows() + 1, 0);
    std::vector<int> stamp((size_t)B.ncols, -1);
    for (int i = 0; i < A.nrows(); ++i) {
        for (int k = A.I[i]; k < A.I[i + 1]; ++k)
            for (int q = B.I[A.J[k]]; q < B.I[A.J[k] + 1]; ++q)
                if (stamp[B.J[q]] != i) { stamp[B.J[q]] = i; C.J.push_back(B.J[q]); }
        C.I[i + 1] = (int)C.J.size();
    }
    return C;
}

struct Csr {
    int nrows = 0, ncols = 0;
    std::vector<int64_t> I;
    std::vector<int> J;
    std::vector<double> V;
};
void spmv(const Csr &A, const double *x, double *y) {
    parallel_ranges(A.nrows, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            double s = 0.0;
            for (int64_t k = A.I[i]; k < A.I[i + 1]; ++k) s += A.V[k] * x[A.J[k]];
            y[i] = s;
        }
    });
}
Csr csr_transpose(const Csr &A) {
    Csr T;
    T.nrows = A.ncols;
    T.ncols = A.nrows;
    T.I.assign((size_t)A.ncols + 1, 0);
    for (int c : A.J) T.I[(size_t)c + 1]++;
    for (int i = 0; i < A.ncols; ++i) T.I[i + 1] += T.I[i];
    T.J.resize(A.J.size());
    T.V.resize(A.V.size());
    std::vector<int64_t> pos(T.I.begin(), T.I.end() - 1);
    for (int i = 0; i < A.nrows; ++i)
        for (int64_t k = A.I[i]; k < A.I[i + 1]; ++k) {
            const int64_t p = pos[A.J[k]]++;
            T.J[p] = i;
            T.V[p] = A.V[k];
        }
    return T;
}
Csr spgemm(const Csr &A, const Csr &B) {   // rows sorted by column; row-parallel, two passes
    Csr C;
    C.nrows = A.nrows;
    C.ncols = B.ncols;
    C.I.assign((size_t)A.nrows + 1, 0);
    std::vector<std::vector<int>> stamp((size_t)g_threads);
    std::vector<std::vector<double>> acc((size_t)g_threads);
    auto pass = [&](bool fill) {
        parallel_ranges(A.nrows, [&](int64_t b, int64_t e, int t) {
            if (stamp[t].empty()) { stamp[t].assign((size_t)B.ncols, -1); acc[t].assign((size_t)B.ncols, 0.0); }
            std::vector<int> cols;
            for (int64_t i = b; i < e; ++i) {
                cols.clear();
                for (int64_t k = A.I[i]; k < A.I[i + 1]; ++k) {
                    const int j = A.J[k];
                    const double a = A.V[k];
                    for (int64_t q = B.I[j]; q < B.I[j + 1]; ++q) {
                        const int c = B.J[q];
                        if (stamp[t][c] != (int)i) { stamp[t][c] = (int)i; acc[t][c] = 0.0; cols.push_back(c); }
                        if (fill) acc[t][c] += a * B.V[q];
                    }
                }
                if (!fill) { C.I[i + 1] = (int64_t)cols.size(); continue; }
                std::sort(cols.begin(), cols.end());
                int64_t p = C.I[i];
                for (int c : cols) { C.J[p] = c; C.V[p++] = acc[t][c]; }
            }
            for (int64_t i = b; i < e && fill; ++i) stamp[t][0] = stamp[t][0];
        });
    };
    pass(false);
    for (int i = 0; i < A.nrows; ++i) C.I[i + 1] += C.I[i];
    C.J.resize((size_t)C.I[A.nrows]);
    C.V.resize((size_t)C.I[A.nrows]);
    for (auto &s : stamp) std::fill(s.begin(), s.end(), -1);
    pass(true);
    return C;
}

// ---- a1/a2: partitioning relations -------------------------------------------------------------
struct Relations {      // agg_partitioning_relations_t (inc/aggregates.hpp:120-179), single rank
    int ND = 0, nparts = 0, num_mises = 0;
    Table elem_to_dof, AE_to_elem, AE_to_dof, dof_to_AE, mis_to_dof, mis_to_AE, AE_to_mis;
    std::vector<int> mises;
    std::vector<int> flags;
    std::vector<int> mis_coloff;     // mis_coarsedofoffsets of the FINER level (set on coarse relations)
};

// agg_construct_mises_local (src/aggregates.cpp:501-653): MIS = dofs with the same AE set; ids in order of
// first appearance scanning dofs upward, dofs inside a MIS ascending.  (Signature map instead of the
// reference's O(#MIS x ND) loop, same numbering.)
void construct_mises(Relations &r) {
    const int ND = r.ND;
    r.mises.assign((size_t)ND, -1);
    // hash of the sorted AE row -> candidate representatives
    std::vector<std::vector<int>> buckets(1 << 16);
    std::vector<std::vector<int>> rows;
    std::vector<int> key;
    auto same = [&](int a, int b) {
        if (r.dof_to_AE.size(a) != r.dof_to_AE.size(b)) return false;
        std::vector<int> x(r.dof_to_AE.row(a), r.dof_to_AE.row(a) + r.dof_to_AE.size(a));
        std::vector<int> y(r.dof_to_AE.row(b), r.dof_to_AE.row(b) + r.dof_to_AE.size(b));
        std::sort(x.begin(), x.end());
        std::sort(y.begin(), y.end());
        return x == y;
    };
    std::vector<int> single((size_t)r.nparts, -1);       // the MIS of the dofs that belong to one AE only
    for (int i = 0; i < ND; ++i) {
        const int rs = r.dof_to_AE.size(i);
        int m = -1;
        if (rs == 1) {
            int &sm = single[r.dof_to_AE.row(i)[0]];
            if (sm < 0) { sm = (int)rows.size(); rows.emplace_back(); }
            m = sm;
        } else {
            key.assign(r.dof_to_AE.row(i), r.dof_to_AE.row(i) + rs);
            std::sort(key.begin(), key.end());
            uint64_t h = 1469598103934665603ull;
            for (int v : key) h = (h ^ (uint64_t)v) * 1099511628211ull;
            auto &bk = buckets[h & 0xffff];
            for (int rep : bk)
                if (same(rep, i)) { m = r.mises[rep]; break; }
            if (m < 0) { m = (int)rows.size(); rows.emplace_back(); bk.push_back(i); }
        }
        r.mises[i] = m;
        rows[m].push_back(i);
    }
    r.num_mises = (int)rows.size();
    r.mis_to_dof.ncols = ND;
    r.mis_to_dof.I.assign(1, 0);
    for (auto &row : rows) {
        r.mis_to_dof.J.insert(r.mis_to_dof.J.end(), row.begin(), row.end());
        r.mis_to_dof.I.push_back((int)r.mis_to_dof.J.size());
    }
}

// agg_create_partitioning_tables (src/aggregates.cpp:1357-1443) + agg_construct_agg_flags (:198-216)
void build_relations(Relations &r, Table e2d, const int *part, int nparts, int ND, const signed char *bdr) {
    r.ND = ND;
    r.nparts = nparts;
    r.elem_to_dof = std::move(e2d);
    const int NE = r.elem_to_dof.nrows();
    Table e2AE;
    e2AE.ncols = nparts;
    e2AE.I.resize((size_t)NE + 1);
    for (int e = 0; e <= NE; ++e) e2AE.I[e] = e;
    e2AE.J.assign(part, part + NE);
    r.AE_to_elem = transpose(e2AE);                          // :1381
    r.AE_to_dof = mult(r.AE_to_elem, r.elem_to_dof);         // :1383
    r.dof_to_AE = transpose(r.AE_to_dof);                    // :1385
    construct_mises(r);
    r.mis_to_AE = mult(r.mis_to_dof, r.dof_to_AE);           // :776
    r.AE_to_mis = transpose(r.mis_to_AE);                    // :777
    r.flags.assign((size_t)ND, 0);
    for (int i = 0; i < ND; ++i) {
        int f = bdr ? (int)bdr[i] : 0;
        if (r.dof_to_AE.size(i) > 1) f |= BETWEEN;
        r.flags[i] = f;
    }
}

struct Dense {        // column-major
    int r = 0, c = 0;
    std::vector<double> v;
    Dense() {}
    Dense(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return v[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
};

struct Level {        // tg_data_t + interp_data_t (inc/tg_data.hpp:47-83, inc/interp.hpp:54-100)
    Csr A, P, R, Ac;
    Relations rel;
    std::vector<Dense> AEs_stiffm;
    std::vector<std::vector<double>> evals;
    std::vector<Dense> evects;
    std::vector<Dense> mis_tent;          // mis_tent_interps
    std::vector<int> mis_k;               // mis_numcoarsedof
    std::vector<double> sv_kept, sv_dropped;   // per MIS: smallest kept / largest dropped sigma / sigma_0 (the cut of xpack_orth_set)
    // lean mode (fine level of problems whose dense AE matrices do not fit the host: 256^3 = 86 GB): the AE matrix is
    // rebuilt from the element matrices whenever it is needed instead of being kept (same arithmetic, same order)
    bool lean = false;
    const double *elmat = nullptr;
    int nde = 0;
    std::vector<int> mis_coloff;          // mis_coarsedofoffsets
    std::vector<double> dinv_neg, roots;
    std::vector<double> t0, r, xc, rc;    // solve-phase work vectors
};

struct Hier {
    std::vector<std::unique_ptr<Level>> levels;
    std::vector<double> coarse_chol;      // Cholesky factor of the coarsest operator (exact solve)
    int nc = 0;
    int coarse_kd = -1;                   // >= 0: coarse_chol holds the BAND Cholesky factor (dpbtrf, lower, kd sub-diagonals)
    double setup_s = 0.0;
};

// a3: agg_build_AE_stiffm_with_global (src/aggregates.cpp:855-945) + agg_assemble_value (:68-184)
Dense fine_AE_matrix(const Csr &A, int p, const Relations &rel, const double *elmat, int nde, std::vector<int> &pos) {
    const int n = rel.AE_to_dof.size(p);
    const int *dofs = rel.AE_to_dof.row(p);
    for (int i = 0; i < n; ++i) pos[dofs[i]] = i;
    Dense out(n, n), M(n, n);
    for (int q = rel.AE_to_elem.I[p]; q < rel.AE_to_elem.I[p + 1]; ++q) {      // ascending element id
        const int e = rel.AE_to_elem.J[q];
        const int *ed = rel.elem_to_dof.row(e);
        const double *Ke = elmat + (size_t)e * nde * nde;
        for (int a = 0; a < nde; ++a)
            for (int b = 0; b < nde; ++b) M(pos[ed[a]], pos[ed[b]]) += Ke[a * nde + b];
    }
    for (int i = 0; i < n; ++i) {
        const int g = dofs[i];
        for (int64_t k = A.I[g]; k < A.I[g + 1]; ++k) {
            const int c = A.J[k];
            const int j = pos[c];
            if (j < 0) continue;
            const bool both_between = (rel.flags[g] & BETWEEN) && (rel.flags[c] & BETWEEN);
            const bool ess_pair = (rel.flags[g] & ESS) || (rel.flags[c] & ESS);
            const double v = (both_between && !(ess_pair && c != g)) ? M(i, j) : A.V[k];
            if (v != 0.0) out(i, j) = v;
        }
    }
    for (int i = 0; i < n; ++i) pos[dofs[i]] = -1;
    return out;
}

// AEs_stiffm[p] of a level, or (lean fine level) that matrix rebuilt into `tmp`; pos: ND entries of -1
const Dense &AE_matrix(const Level &L, int p, std::vector<int> &pos, Dense &tmp);

// a5: mbox_snd_D_sparse_from_sparse (src/mbox.cpp:913-949)
std::vector<double> snd_D(const Dense &A) {
    const int n = A.r;
    std::vector<double> D((size_t)n, 0.0);
    for (int j = 0; j < n; ++j) {
        const double ajj = A(j, j);
        for (int i = 0; i < n; ++i) {
            const double a = A(i, j);
            if (a != 0.0) D[i] += std::fabs(a) * std::sqrt(A(i, i) / ajj);
        }
    }
    return D;
}

const Dense &AE_matrix(const Level &L, int p, std::vector<int> &pos, Dense &tmp) {
    if (!L.lean) return L.AEs_stiffm[(size_t)p];
    tmp = fine_AE_matrix(L.A, p, L.rel, L.elmat, L.nde, pos);
    return tmp;
}

// a6: xpacks_calc_lower_eigens_dense (src/xpacks.cpp:222-314): dsygvx itype 1, range 'V' on (-1, theta], abstol =
// 2 dlamch('S'), uplo 'U'; nothing found -> the single smallest pair (range 'I', il = iu = 1)
void lower_eigens(const Dense &A, const std::vector<double> &D, double theta, std::vector<double> &w_out, Dense &Z_out) {
    const int n = A.r;
    const double abstol = 2.0 * LAPACK_NAME(dlamch)("S");
    std::vector<double> a, b((size_t)n * n), w((size_t)n), z, work;
    std::vector<int> iwork((size_t)5 * n), ifail((size_t)n);
    const int itype = 1;
    int m = 0, info = 0;
    auto run = [&](const char *range, int il, int iu) {
        a = A.v;
        std::fill(b.begin(), b.end(), 0.0);
        for (int i = 0; i < n; ++i) b[(size_t)i * n + i] = D[i];
        const double vl = -1.0, vu = theta;
        z.assign((size_t)n * n, 0.0);
        int lwork = -1;
        double wq = 0.0;
        LAPACK_NAME(dsygvx)(&itype, "V", range, "U", &n, a.data(), &n, b.data(), &n, &vl, &vu, &il, &iu, &abstol, &m,
                            w.data(), z.data(), &n, &wq, &lwork, iwork.data(), ifail.data(), &info);
        lwork = std::max((int)wq, 8 * n);
        work.resize((size_t)lwork);
        LAPACK_NAME(dsygvx)(&itype, "V", range, "U", &n, a.data(), &n, b.data(), &n, &vl, &vu, &il, &iu, &abstol, &m,
                            w.data(), z.data(), &n, work.data(), &lwork, iwork.data(), ifail.data(), &info);
        if (info != 0) throw std::runtime_error("dsygvx failed");
    };
    run("V", 1, 1);
    if (m <= 0) run("I", 1, 1);
    w_out.assign(w.begin(), w.begin() + m);
    Z_out = Dense(n, m);
    std::copy(z.begin(), z.begin() + (size_t)n * m, Z_out.v.begin());
}

// a7/a8: ContribTent::contrib_mises (src/contrib.cpp:492-687), xpack_svd_dense_arr (src/xpacks.cpp:494-589),
// xpack_orth_set (:591-620)
void mis_block(const Level &L, int mis, std::vector<int> &pos, Dense &U, int &k, double &sv_kept, double &sv_dropped) {
    sv_kept = INFINITY;
    sv_dropped = 0.0;
    const Relations &rel = L.rel;
    const int dim = rel.mis_to_dof.size(mis);
    const int *mdofs = rel.mis_to_dof.row(mis);
    bool all_ess = true;
    for (int j = 0; j < dim; ++j) all_ess = all_ess && (rel.flags[mdofs[j]] & ESS);
    k = 0;
    U = Dense(dim, 0);
    if (all_ess) return;                                  // :578-605
    if (dim == 1) {                                       // :607-612
        U = Dense(1, 1);
        U(0, 0) = 1.0;
        k = 1;
        return;
    }
    std::vector<std::vector<double>> cols;
    for (int q = rel.mis_to_AE.I[mis]; q < rel.mis_to_AE.I[mis + 1]; ++q) {     // :525-542
        const int AE = rel.mis_to_AE.J[q];
        const int *ad = rel.AE_to_dof.row(AE);
        const int na = rel.AE_to_dof.size(AE);
        for (int i = 0; i < na; ++i) pos[ad[i]] = i;
        const Dense &Z = L.evects[AE];
        for (int c = 0; c < Z.c; ++c) {
            std::vector<double> col((size_t)dim);
            bool any = false;
            for (int j = 0; j < dim; ++j) {
                double v = Z(pos[mdofs[j]], c);
                if (rel.flags[mdofs[j]] & ESS) v = 0.0;  // contrib_filter_boundary (:102-163)
                col[j] = v;
                any = any || v != 0.0;
            }
            if (any) cols.push_back(std::move(col));
        }
        for (int i = 0; i < na; ++i) pos[ad[i]] = -1;
    }
    // normalise, drop (near-)zero columns (src/xpacks.cpp:537-559)
    std::vector<double> a;
    int nc = 0;
    for (auto &col : cols) {
        double nn = 0.0;
        for (double v : col) nn += v * v;
        const double nrm = std::sqrt(nn);
        if (nrm <= 0.0 + DIFF_EPS) continue;
        for (double v : col) a.push_back(v / nrm);
        ++nc;
    }
    if (nc == 0) return;
    const int mn = std::min(dim, nc);
    std::vector<double> s((size_t)mn), u((size_t)dim * mn), work(1);
    int lwork = -1, info = 0, one = 1;
    double vt = 0.0;
    LAPACK_NAME(dgesvd)("S", "N", &dim, &nc, a.data(), &dim, s.data(), u.data(), &dim, &vt, &one, work.data(), &lwork, &info);
    lwork = std::max((int)work[0], 5 * std::max(dim, nc));
    work.resize((size_t)lwork);
    LAPACK_NAME(dgesvd)("S", "N", &dim, &nc, a.data(), &dim, s.data(), u.data(), &dim, &vt, &one, work.data(), &lwork, &info);
    if (info != 0) throw std::runtime_error("dgesvd failed");
    const double eps = SVD_EPS * s[0];
    while (k < mn && s[k] > eps) ++k;
    if (k > 0) sv_kept = s[k - 1] / s[0];
    if (k < mn) sv_dropped = s[k] / s[0];
    U = Dense(dim, k);
    std::copy(u.begin(), u.begin() + (size_t)dim * k, U.v.begin());
}

// mbox_build_Dinv_neg_parallel_matrix (src/mbox.cpp:1839-1861)
std::vector<double> build_dinv_neg(const Csr &A) {
    std::vector<double> diag((size_t)A.nrows, 0.0), out((size_t)A.nrows);
    for (int i = 0; i < A.nrows; ++i)
        for (int64_t k = A.I[i]; k < A.I[i + 1]; ++k)
            if (A.J[k] == i) diag[i] = std::fabs(A.V[k]);
    for (int i = 0; i < A.nrows; ++i) {
        double y = 0.0;
        for (int64_t k = A.I[i]; k < A.I[i + 1]; ++k) y += std::fabs(A.V[k]) / std::sqrt(diag[A.J[k]]);
        out[i] = -1.0 / (std::sqrt(diag[i]) * y);
    }
    return out;
}
// smpr_sas_poly_roots (src/smpr.cpp:282-306)
std::vector<double> sas_roots(int nu) {
    std::vector<double> r;
    const double den = 2.0 * nu + 1.0;
    for (int i = 0; i <= 2 * nu; ++i) { const double v = std::cos(i * M_PI / den); r.push_back(v * v); }
    for (int i = 1; i <= nu; ++i) { const double v = std::sin(i * M_PI / den); r.push_back(v * v); }
    return r;
}

// tg_init_data + tg_build_hierarchy + tg_update_coarse_operator (src/tg.cpp:402-430, :502-540, :979-1014)
void build_level(Level &L, double theta, int nu_relax) {
    const Relations &rel = L.rel;
    const int np = rel.nparts;
    L.dinv_neg = build_dinv_neg(L.A);
    L.roots = sas_roots(nu_relax);
    L.evals.resize((size_t)np);
    L.evects.resize((size_t)np);
    {
        std::vector<std::vector<int>> apos((size_t)g_threads);
        std::vector<Dense> atmp((size_t)g_threads);
        parallel_for(np, [&](int64_t p, int t) {
            if (L.lean && apos[t].empty()) apos[t].assign((size_t)rel.ND, -1);
            const Dense &Ae = AE_matrix(L, (int)p, apos[t], atmp[t]);
            lower_eigens(Ae, snd_D(Ae), theta, L.evals[(size_t)p], L.evects[(size_t)p]);
        });
    }
    note("eigenproblems done");
    const int nm = rel.num_mises;
    L.mis_tent.resize((size_t)nm);
    L.mis_k.assign((size_t)nm, 0);
    L.sv_kept.assign((size_t)nm, INFINITY);
    L.sv_dropped.assign((size_t)nm, 0.0);
    std::vector<std::vector<int>> pos((size_t)g_threads);
    parallel_for(nm, [&](int64_t m, int t) {
        if (pos[t].empty()) pos[t].assign((size_t)rel.ND, -1);
        mis_block(L, (int)m, pos[t], L.mis_tent[(size_t)m], L.mis_k[(size_t)m], L.sv_kept[(size_t)m], L.sv_dropped[(size_t)m]);
    });
    L.mis_coloff.assign((size_t)nm + 1, 0);
    for (int m = 0; m < nm; ++m) L.mis_coloff[m + 1] = L.mis_coloff[m] + L.mis_k[m];
    // contrib_tent_insert_simple (src/contrib.cpp:170-194): exact zeros are not inserted
    Csr &P = L.P;
    P.nrows = rel.ND;
    P.ncols = L.mis_coloff[nm];
    P.I.assign((size_t)rel.ND + 1, 0);
    std::vector<int> row_in_mis((size_t)rel.ND, 0);
    for (int m = 0; m < nm; ++m)
        for (int j = 0; j < rel.mis_to_dof.size(m); ++j) row_in_mis[rel.mis_to_dof.row(m)[j]] = j;
    for (int d = 0; d < rel.ND; ++d) {
        const int m = rel.mises[d];
        int cnt = 0;
        for (int c = 0; c < L.mis_k[m]; ++c) cnt += std::fabs(L.mis_tent[m](row_in_mis[d], c)) > 0.0;
        P.I[d + 1] = P.I[d] + cnt;
    }
    P.J.resize((size_t)P.I[rel.ND]);
    P.V.resize((size_t)P.I[rel.ND]);
    for (int d = 0; d < rel.ND; ++d) {
        const int m = rel.mises[d];
        int64_t p = P.I[d];
        for (int c = 0; c < L.mis_k[m]; ++c) {
            const double v = L.mis_tent[m](row_in_mis[d], c);
            if (std::fabs(v) > 0.0) { P.J[p] = L.mis_coloff[m] + c; P.V[p++] = v; }
        }
    }
    note("MIS SVDs done, P assembled");
    L.R = csr_transpose(P);
    L.Ac = spgemm(L.R, spgemm(L.A, P));      // tg_coarse_matr == RAP (inc/tg.hpp:696-709)
    const size_t n = (size_t)L.A.nrows;
    L.t0.assign(n, 0.0);
    L.r.assign(n, 0.0);
    L.rc.assign((size_t)P.ncols, 0.0);
    L.xc.assign((size_t)P.ncols, 0.0);
}

// ElementMatrixParallelCoarse::GetMatrix (src/elmat.cpp:105-195): P_loc^T AEs_stiffm[e] P_loc
Dense coarse_element_matrix(int e, const Level &F, const Relations &rc, std::vector<int> &pos, std::vector<int> &cpos) {
    const Relations &rf = F.rel;
    Dense aetmp;
    const Dense &Ae = AE_matrix(F, e, pos, aetmp);
    const int nf = Ae.r;
    const int *fd = rf.AE_to_dof.row(e);
    for (int i = 0; i < nf; ++i) pos[fd[i]] = i;
    const int *ed = rc.elem_to_dof.row(e);
    const int nce = rc.elem_to_dof.size(e);
    for (int j = 0; j < nce; ++j) cpos[ed[j]] = j;
    Dense Pl(nf, nce);
    for (int q = rf.AE_to_mis.I[e]; q < rf.AE_to_mis.I[e + 1]; ++q) {     // (transpose rows are ascending == sorted)
        const int m = rf.AE_to_mis.J[q];
        const Dense &U = F.mis_tent[(size_t)m];
        for (int c = 0; c < F.mis_k[m]; ++c) {
            const int col = cpos[F.mis_coloff[m] + c];
            if (col < 0) throw std::runtime_error("coarse dof with an all-zero prolongator column in an AE");
            for (int j = 0; j < rf.mis_to_dof.size(m); ++j) Pl(pos[rf.mis_to_dof.row(m)[j]], col) += U(j, c);
        }
    }
    Dense T(nf, nce), out(nce, nce);
    for (int c = 0; c < nce; ++c)
        for (int k = 0; k < nf; ++k) {
            const double p = Pl(k, c);
            if (p == 0.0) continue;
            for (int i = 0; i < nf; ++i) T(i, c) += Ae(i, k) * p;
        }
    for (int c = 0; c < nce; ++c)
        for (int a = 0; a < nce; ++a) {
            double s = 0.0;
            for (int i = 0; i < nf; ++i) s += Pl(i, a) * T(i, c);
            out(a, c) = s;
        }
    for (int i = 0; i < nf; ++i) pos[fd[i]] = -1;
    for (int j = 0; j < nce; ++j) cpos[ed[j]] = -1;
    return out;
}

// smpr_compute_poly (inc/smpr.hpp:320-339): x += (1/tau) Dinv_neg (A x - b), roots in array order
void smooth(const Level &L, const double *b, double *x, double *tmp) {
    const int n = L.A.nrows;
    for (double tau : L.roots) {
        spmv(L.A, x, tmp);
        const double it = 1.0 / tau;
        parallel_ranges(n, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) x[i] += it * (L.dinv_neg[i] * (tmp[i] - b[i]));
        });
    }
}

// tg_cycle_atb (src/tg.cpp:91-132) through ml_impose_cycle (src/ml.cpp:361-377), x0 = 0
void vcycle(Hier &H, int lev, const double *b, double *x) {
    Level &L = *H.levels[(size_t)lev];
    const int n = L.A.nrows;
    std::fill(x, x + n, 0.0);
    smooth(L, b, x, L.t0.data());
    spmv(L.A, x, L.r.data());
    for (int i = 0; i < n; ++i) L.r[i] = b[i] - L.r[i];
    spmv(L.R, L.r.data(), L.rc.data());
    if (lev + 1 < (int)H.levels.size()) {
        vcycle(H, lev + 1, L.rc.data(), L.xc.data());
    } else {
        L.xc = L.rc;
        const int one = 1;
        int info = 0;
        if (H.nc && H.coarse_kd >= 0) {
            const int ldab = H.coarse_kd + 1;
            LAPACK_NAME(dpbtrs)("L", &H.nc, &H.coarse_kd, &one, H.coarse_chol.data(), &ldab, L.xc.data(), &H.nc, &info);
        } else if (H.nc)
            LAPACK_NAME(dpotrs)("L", &H.nc, &one, H.coarse_chol.data(), &H.nc, L.xc.data(), &H.nc, &info);
    }
    spmv(L.P, L.xc.data(), L.t0.data());
    for (int i = 0; i < n; ++i) x[i] += L.t0[i];
    smooth(L, b, x, L.t0.data());
}

double dot(const double *a, const double *b, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

}  // namespace

extern "C" {

struct cpu_ref_hier { Hier H; std::string err; };

// ml_produce_data (src/ml.cpp:379-472) on raw arrays; same inputs as saamge_amd_ml_produce_data
// thetas: one spectral tolerance per coarsening (first_theta / theta of MultilevelParameters, inc/ml.hpp:66-70);
// lean != 0: the fine level's dense AE matrices are rebuilt on demand instead of kept
cpu_ref_hier *cpu_ref_setup2(int n, const int *rowptr, const int *col, const double *val, int NE, int nde,
                             const int *elem_to_dof, const double *elmat, const signed char *bdr, int ncoarsen,
                             const int *const *partitions, const int *nparts, const double *thetas, int nu_relax,
                             int threads, int lean) {
    cpu_ref_hier *h = new cpu_ref_hier;
    try {
        g_threads = std::max(1, threads);
        g_verbose = std::getenv("CPU_REF_VERBOSE") != nullptr;
#ifdef BLAS_SET_THREADS
        BLAS_SET_THREADS(1);
#endif
        const auto t0 = std::chrono::steady_clock::now();
        Hier &H = h->H;
        H.levels.emplace_back(new Level);
        Level &L0 = *H.levels[0];
        L0.A.nrows = L0.A.ncols = n;
        L0.A.I.assign(rowptr, rowptr + n + 1);
        L0.A.J.assign(col, col + rowptr[n]);
        L0.A.V.assign(val, val + rowptr[n]);
        Table e2d;
        e2d.ncols = n;
        e2d.I.resize((size_t)NE + 1);
        for (int e = 0; e <= NE; ++e) e2d.I[e] = e * nde;
        e2d.J.assign(elem_to_dof, elem_to_dof + (size_t)NE * nde);
        build_relations(L0.rel, std::move(e2d), partitions[0], nparts[0], n, bdr);
        L0.lean = lean != 0;
        L0.elmat = elmat;
        L0.nde = nde;
        if (!L0.lean) L0.AEs_stiffm.resize((size_t)nparts[0]);
        if (!L0.lean) {
            std::vector<std::vector<int>> pos((size_t)g_threads);
            parallel_for(nparts[0], [&](int64_t p, int t) {
                if (pos[t].empty()) pos[t].assign((size_t)n, -1);
                L0.AEs_stiffm[(size_t)p] = fine_AE_matrix(L0.A, (int)p, L0.rel, elmat, nde, pos[t]);
            });
        }
        note("level 0: relations built, AEs", nparts[0]);
        build_level(L0, thetas[0], nu_relax);
        note("level 0 done, coarse dim", L0.P.ncols);
        for (int k = 1; k < ncoarsen; ++k) {
            Level &F = *H.levels.back();
            H.levels.emplace_back(new Level);
            Level &L = *H.levels.back();
            L.A = F.Ac;                                              // src/ml.cpp:134
            // agg_create_partitioning_coarse (src/aggregates.cpp:1610-1832): coarse elements = fine AEs,
            // elem_to_dof = AE_to_dof x pattern(P_tent), no essential flags
            Table f2c;
            f2c.ncols = F.P.ncols;
            f2c.I.assign(F.P.I.begin(), F.P.I.end());
            f2c.J = F.P.J;
            Table ce2d = mult(F.rel.AE_to_dof, f2c);
            build_relations(L.rel, std::move(ce2d), partitions[k], nparts[k], F.P.ncols, nullptr);
            const int nel = F.rel.nparts;
            std::vector<Dense> cel((size_t)nel);
            {
                std::vector<std::vector<int>> pos((size_t)g_threads), cpos((size_t)g_threads);
                parallel_for(nel, [&](int64_t e, int t) {
                    if (pos[t].empty()) { pos[t].assign((size_t)F.rel.ND, -1); cpos[t].assign((size_t)F.P.ncols, -1); }
                    cel[(size_t)e] = coarse_element_matrix((int)e, F, L.rel, pos[t], cpos[t]);
                });
            }
            // agg_build_AE_stiffm (src/aggregates.cpp:959-1086): plain sum of the coarse element matrices
            L.AEs_stiffm.resize((size_t)nparts[k]);
            {
                std::vector<std::vector<int>> pos((size_t)g_threads);
                parallel_for(nparts[k], [&](int64_t p, int t) {
                    if (pos[t].empty()) pos[t].assign((size_t)L.rel.ND, -1);
                    const int na = L.rel.AE_to_dof.size((int)p);
                    const int *ad = L.rel.AE_to_dof.row((int)p);
                    for (int i = 0; i < na; ++i) pos[t][ad[i]] = i;
                    Dense out(na, na);
                    for (int q = L.rel.AE_to_elem.I[p]; q < L.rel.AE_to_elem.I[p + 1]; ++q) {
                        const int e = L.rel.AE_to_elem.J[q];
                        const int *ed = L.rel.elem_to_dof.row(e);
                        const int ne = L.rel.elem_to_dof.size(e);
                        const Dense &Ke = cel[(size_t)e];
                        for (int b = 0; b < ne; ++b)
                            for (int a = 0; a < ne; ++a) out(pos[t][ed[a]], pos[t][ed[b]]) += Ke(a, b);
                    }
                    for (int i = 0; i < na; ++i) pos[t][ad[i]] = -1;
                    L.AEs_stiffm[(size_t)p] = std::move(out);
                });
            }
            note("coarse level: AE matrices assembled, AEs", nparts[k]);
            build_level(L, thetas[k], nu_relax);
            note("coarse level done, coarse dim", L.P.ncols);
        }
        // exact coarsest solve (the reference's --coarse-direct, src/tg.cpp:989-997)
        const Csr &Ac = H.levels.back()->Ac;
        H.nc = Ac.nrows;
        int info = 0;
        const char *bm = std::getenv("CPU_REF_BAND_MIN");      // (tests lower it to reach the band path on small problems)
        if (H.nc > (bm ? std::atoi(bm) : 8192)) {      // (2-level hierarchies of the large problems: 67 975 rows at 128^3) band Cholesky, still exact
            int kd = 0;
            for (int i = 0; i < H.nc; ++i)
                for (int64_t k = Ac.I[i]; k < Ac.I[i + 1]; ++k) kd = std::max(kd, std::abs(i - Ac.J[k]));
            H.coarse_kd = kd;
            note("coarsest: band Cholesky, half bandwidth", kd);
            const int ldab = kd + 1;
            H.coarse_chol.assign((size_t)H.nc * ldab, 0.0);
            for (int i = 0; i < H.nc; ++i)
                for (int64_t k = Ac.I[i]; k < Ac.I[i + 1]; ++k)
                    if (Ac.J[k] <= i) H.coarse_chol[(size_t)Ac.J[k] * ldab + (i - Ac.J[k])] = Ac.V[k];
            LAPACK_NAME(dpbtrf)("L", &H.nc, &kd, H.coarse_chol.data(), &ldab, &info);     // (single-threaded BLAS: n kd^2 = 2.4e11 flop at 128^3)
        } else {
            H.coarse_chol.assign((size_t)H.nc * H.nc, 0.0);
            for (int i = 0; i < H.nc; ++i)
                for (int64_t k = Ac.I[i]; k < Ac.I[i + 1]; ++k) H.coarse_chol[(size_t)Ac.J[k] * H.nc + i] = Ac.V[k];
            if (H.nc) LAPACK_NAME(dpotrf)("L", &H.nc, H.coarse_chol.data(), &H.nc, &info);
        }
        if (info != 0) throw std::runtime_error("coarsest operator is not positive definite");
        H.setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (const std::exception &e) {
        h->err = e.what();
    }
    return h;
}

cpu_ref_hier *cpu_ref_setup(int n, const int *rowptr, const int *col, const double *val, int NE, int nde,
                            const int *elem_to_dof, const double *elmat, const signed char *bdr, int ncoarsen,
                            const int *const *partitions, const int *nparts, double theta, int nu_relax, int threads) {
    std::vector<double> th((size_t)std::max(ncoarsen, 1), theta);
    return cpu_ref_setup2(n, rowptr, col, val, NE, nde, elem_to_dof, elmat, bdr, ncoarsen, partitions, nparts, th.data(),
                          nu_relax, threads, 0);
}

const char *cpu_ref_error(const cpu_ref_hier *h) { return h->err.empty() ? nullptr : h->err.c_str(); }
void cpu_ref_free(cpu_ref_hier *h) { delete h; }
double cpu_ref_setup_seconds(const cpu_ref_hier *h) { return h->H.setup_s; }
int cpu_ref_num_levels(const cpu_ref_hier *h) { return (int)h->H.levels.size(); }
// info: [0] rows [1] nnz [2] nparts [3] num_mises [4] coarse dim [5] nnz(P) [6] nnz(Ac)
void cpu_ref_level_info(const cpu_ref_hier *h, int l, long long info[8]) {
    const Level &L = *h->H.levels[(size_t)l];
    info[0] = L.A.nrows; info[1] = L.A.I[L.A.nrows]; info[2] = L.rel.nparts; info[3] = L.rel.num_mises;
    info[4] = L.P.ncols; info[5] = L.P.I[L.P.nrows]; info[6] = L.Ac.I[L.Ac.nrows]; info[7] = 0;
}
// which: 0 eigenvectors per AE (nparts), 1 coarse dofs per MIS (num_mises), 2 MIS of each dof (rows)
void cpu_ref_get_ints(const cpu_ref_hier *h, int l, int which, int *out) {
    const Level &L = *h->H.levels[(size_t)l];
    if (which == 0) for (int p = 0; p < L.rel.nparts; ++p) out[p] = L.evects[(size_t)p].c;
    if (which == 1) std::copy(L.mis_k.begin(), L.mis_k.end(), out);
    if (which == 2) std::copy(L.rel.mises.begin(), L.rel.mises.end(), out);
}
// largest kept eigenvalue per AE (nparts)
void cpu_ref_get_evals_max(const cpu_ref_hier *h, int l, double *out) {
    const Level &L = *h->H.levels[(size_t)l];
    for (int p = 0; p < L.rel.nparts; ++p) out[p] = L.evals[(size_t)p].back();
}
// all kept eigenvalues, AE after AE (sum of cpu_ref_get_ints(.., 0, ..) values)
void cpu_ref_get_evals(const cpu_ref_hier *h, int l, double *out) {
    const Level &L = *h->H.levels[(size_t)l];
    for (int p = 0; p < L.rel.nparts; ++p)
        for (double w : L.evals[(size_t)p]) *out++ = w;
}
// mis_to_AE as CSR: I (num_mises + 1), J (I[num_mises]); pass J = NULL to get I alone
void cpu_ref_get_mis_to_AE(const cpu_ref_hier *h, int l, int *I, int *J) {
    const Table &T = h->H.levels[(size_t)l]->rel.mis_to_AE;
    std::copy(T.I.begin(), T.I.end(), I);
    if (J) std::copy(T.J.begin(), T.J.end(), J);
}
// per MIS (num_mises): which 0 = smallest kept sigma / sigma_0 (inf: nothing kept or no SVD), 1 = largest dropped (0: none)
void cpu_ref_get_sv_ratios(const cpu_ref_hier *h, int l, int which, double *out) {
    const Level &L = *h->H.levels[(size_t)l];
    const std::vector<double> &v = which == 0 ? L.sv_kept : L.sv_dropped;
    std::copy(v.begin(), v.end(), out);
}
// trace of the level's Galerkin operator
double cpu_ref_Ac_trace(const cpu_ref_hier *h, int l) {
    const Csr &Ac = h->H.levels[(size_t)l]->Ac;
    double s = 0.0;
    for (int i = 0; i < Ac.nrows; ++i)
        for (int64_t k = Ac.I[i]; k < Ac.I[i + 1]; ++k)
            if (Ac.J[k] == i) s += Ac.V[k];
    return s;
}
// squared Frobenius norm of the level's Galerkin operator (like the trace: invariant under an orthogonal change of basis of
// the coarse space, i.e. under the freedom a degenerate local eigenspace leaves)
double cpu_ref_Ac_fro2(const cpu_ref_hier *h, int l) {
    const Csr &Ac = h->H.levels[(size_t)l]->Ac;
    double s = 0.0;
    for (int64_t k = 0; k < Ac.I[Ac.nrows]; ++k) s += Ac.V[k] * Ac.V[k];
    return s;
}
void cpu_ref_vcycle(cpu_ref_hier *h, const double *b, double *x) { vcycle(h->H, 0, b, x); }

// MFEM CGSolver::Mult as driven by test/mltest/mltest.cpp:773-781 (squared_tol) == kalchev_pcg
// (src/mfem_addons.cpp:106-248); hist receives (B r_k, r_k), k = 0..iters
int cpu_ref_pcg(cpu_ref_hier *h, const double *b, double *x, double rel_tol, int max_iter, int squared_tol,
                int *converged, double *hist, double *solve_s) {
    const auto t0 = std::chrono::steady_clock::now();
    Hier &H = h->H;
    const Csr &A = H.levels[0]->A;
    const int64_t n = A.nrows;
    std::vector<double> r(b, b + n), z((size_t)n), d((size_t)n), q((size_t)n);
    std::fill(x, x + n, 0.0);
    vcycle(H, 0, r.data(), z.data());
    d = z;
    double nom = dot(d.data(), r.data(), n);
    hist[0] = nom;
    const double r0 = squared_tol ? nom * rel_tol * rel_tol : nom * rel_tol;
    *converged = 0;
    int final_iter = max_iter;
    if (nom <= r0) { *converged = 1; return 0; }
    spmv(A, d.data(), q.data());
    double den = dot(q.data(), d.data(), n);
    if (den == 0.0) return 0;
    int i = 1;
    for (;;) {
        const double alpha = nom / den;
        for (int64_t k = 0; k < n; ++k) { x[k] += alpha * d[k]; r[k] -= alpha * q[k]; }
        vcycle(H, 0, r.data(), z.data());
        const double betanom = dot(r.data(), z.data(), n);
        hist[i] = betanom;
        if (betanom < r0) { *converged = 1; final_iter = i; break; }
        if (++i > max_iter) break;
        const double beta = betanom / nom;
        for (int64_t k = 0; k < n; ++k) d[k] = z[k] + beta * d[k];
        spmv(A, d.data(), q.data());
        den = dot(d.data(), q.data(), n);
        nom = betanom;
    }
    if (solve_s) *solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return final_iter;
}

}  // extern "C"
