// saamge_amd_mfem.hpp -- the reference's own C++ entry points for the setup + solve hot path, in
// namespace saamge, as thin adaptors over the C ABI (saamge_amd.h).  Compiled only where MFEM + hypre
// headers exist (SAAMGE_AMD_WITH_MFEM; not in the build container -- tests/test_cxx_boundary.py compiles it
// against tests/mfem_stub/, a declaration-only stand-in for the handful of MFEM types used here).
//
// A driver written against <saamge.hpp> (amg/test/mltest/mltest.cpp:667-793 is the model) keeps its
// call sequence:
//     agg_part_rels = agg_create_partitioning_fine(A, NE, elem_to_dof, elem_to_elem, partitioning,
//                                                  bdr_dofs, nparts, dof_truedof, do_aggregates);
//     emp = new ElementMatrixStandardGeometric(*agg_part_rels, Al, a);
//     MultilevelParameters mlp(coarsenings, nparts, first_nu_pro, nu_pro, nu_relax, first_theta, theta,
//                              polynomial_coarse, correct_nulspace, use_arpack, do_aggregates);
//     ml_data = ml_produce_data(*Ag, agg_part_rels, emp, mlp);
//     level = levels_list_get_level(ml_data->levels_list, 0);
//     Bprec = new VCycleSolver(level->tg_data, false);  Bprec->SetOperator(*Ag);
//     CGSolver ... SetPreconditioner(*Bprec) ... Mult(b, x);
//     ml_free_data(ml_data);  agg_free_partitioning(agg_part_rels);
// What differs from the reference, by design: the topology tables, the local spectral problems, P, RAP
// and the V-cycle live on the GPU; agg_partitioning_relations_t therefore carries the INPUTS of
// agg_create_partitioning_fine (partitioning, elem_to_dof, flags) and the tables are fetched from the
// hierarchy on request (agg_fetch_tables).  Coarse partitions (METIS in the reference,
// src/part.cpp:170-183 -- third party) come from a hook, ml_set_coarse_partitioner(); the default cuts
// the coarse elements into contiguous index ranges.  One MPI rank per GPU.  A HypreParMatrix that is DISTRIBUTED
// over several ranks is taken as it is (round 4): ml_produce_data hands the rank's diag / offd / col_map_offd blocks, the
// rank's own elements (local dofs mapped to true dofs through agg_part_rels' Dof_TrueDof) and its own partition to
// saamge_amd_ml_produce_data_parcsr, with MPI-backed collectives (detail::MpiCollectives: staged through host memory,
// the plug for MPI host codes; a driver that owns an RCCL communicator installs saamge_amd_params_set_comm through
// MultilevelParameters::p instead).  Vectors at this boundary are the rank's true-dof blocks, like HypreParVectors; the
// host copies of P / R / Ac (tg_data_t::interp, restr, Ac) are not made for a distributed hierarchy.
#ifndef SAAMGE_AMD_MFEM_HPP
#define SAAMGE_AMD_MFEM_HPP

#include <mfem.hpp>

#include <cmath>
#include <functional>
#include <vector>

#include "saamge_amd.hpp"

namespace saamge {

typedef char agg_dof_status_t;                       // inc/aggregates.hpp:115
constexpr agg_dof_status_t AGG_BETWEEN_AES_FLAG = 0x01, AGG_ON_ESS_DOMAIN_BORDER_FLAG = 0x02,
                           AGG_ON_PROC_IFACE_FLAG = 0x04, AGG_OWNED_FLAG = 0x08;     // :102-105

// == agg_partitioning_relations_t (inc/aggregates.hpp:120-179), the fields drivers read.
struct agg_partitioning_relations_t {
    int ND = 0;                       // dofs
    int nparts = 0;                   // AEs
    int *partitioning = nullptr;      // element -> AE (owned, like the reference: src/aggregates.cpp:1834-1870)
    mfem::Table *elem_to_dof = nullptr, *elem_to_elem = nullptr;      // owned
    agg_dof_status_t *agg_flags = nullptr;                            // owned copy of bdr_dofs (BETWEEN added by agg_fetch_tables)
    // filled by agg_fetch_tables() from a built hierarchy (NULL before):
    mfem::Table *AE_to_dof = nullptr, *dof_to_AE = nullptr, *mis_to_dof = nullptr, *mis_to_AE = nullptr,
                *AE_to_mis = nullptr;
    int num_mises = 0;
    int *mises = nullptr;
    bool testmesh = false;
    bool do_aggregates = false;
    int NE = 0;
    mfem::HypreParMatrix *Dof_TrueDof = nullptr;      // not owned (the caller's fes.Dof_TrueDof_Matrix(), src/aggregates.cpp:1347)
};

// agg_create_partitioning_fine (inc/aggregates.hpp:385-390, src/aggregates.cpp:1316-1355): takes ownership
// of elem_to_dof, elem_to_elem and partitioning.  `partitioning` must be given (the reference calls METIS
// when it is NULL -- third party, not reproduced); *nparts is the number of AEs.
inline agg_partitioning_relations_t *agg_create_partitioning_fine(
    mfem::HypreParMatrix &A, int NE, mfem::Table *elem_to_dof, mfem::Table *elem_to_elem, int *partitioning,
    const agg_dof_status_t *bdr_dofs, int *nparts, mfem::HypreParMatrix *dof_truedof, bool do_aggregates,
    bool testmesh = false) {
    if (!partitioning) mfem::mfem_error("agg_create_partitioning_fine: a partitioning array is required (METIS is not part of this library)");
    agg_partitioning_relations_t *r = new agg_partitioning_relations_t;
    r->Dof_TrueDof = dof_truedof;
    // ND = LOCAL dofs (shared copies included): the rows of Dof_TrueDof; on one rank the matrix's own rows
    r->ND = dof_truedof ? dof_truedof->Height() : A.Height();
    r->NE = NE;
    r->nparts = *nparts;
    r->partitioning = partitioning;
    r->elem_to_dof = elem_to_dof;
    r->elem_to_elem = elem_to_elem;
    r->agg_flags = new agg_dof_status_t[r->ND];
    for (int i = 0; i < r->ND; ++i) r->agg_flags[i] = bdr_dofs ? bdr_dofs[i] : AGG_OWNED_FLAG;
    r->testmesh = testmesh;
    r->do_aggregates = do_aggregates;
    return r;
}
inline void agg_free_partitioning(agg_partitioning_relations_t *r) {     // src/aggregates.cpp:1834-1870
    if (!r) return;
    delete[] r->partitioning;
    delete r->elem_to_dof;
    delete r->elem_to_elem;
    delete[] r->agg_flags;
    delete r->AE_to_dof; delete r->dof_to_AE; delete r->mis_to_dof; delete r->mis_to_AE; delete r->AE_to_mis;
    delete[] r->mises;
    delete r;
}

// == ElementMatrixProvider (inc/elmat.hpp:53-78)
class ElementMatrixProvider {
public:
    ElementMatrixProvider(const agg_partitioning_relations_t &agg_part_rels_) : agg_part_rels(agg_part_rels_), is_geometric(false) {}
    virtual ~ElementMatrixProvider() {}
    virtual mfem::Matrix *GetMatrix(int elno, bool &free_matr) const = 0;
    // (the AE matrices are assembled on the GPU from the element matrices: a provider's BuildAEStiff is never called)
    virtual mfem::SparseMatrix *BuildAEStiff(int elno) const { (void)elno; return nullptr; }
    bool IsGeometric() { return is_geometric; }
protected:
    const agg_partitioning_relations_t &agg_part_rels;
    bool is_geometric;
};
// == ElementMatrixStandardGeometric (inc/elmat.hpp:80-100, src/elmat.cpp:46-88): element matrices from the
// bilinear form (no boundary conditions imposed: bdr_cond_imposed = true means the caller's global matrix has them)
class ElementMatrixStandardGeometric : public ElementMatrixProvider {
public:
    ElementMatrixStandardGeometric(const agg_partitioning_relations_t &agg_part_rels_,
                                   mfem::SparseMatrix *assembled_processor_matrix, mfem::ParBilinearForm *form_)
        : ElementMatrixProvider(agg_part_rels_), assembled(assembled_processor_matrix), form(form_) { is_geometric = true; }
    virtual mfem::Matrix *GetMatrix(int elno, bool &free_matr) const {
        mfem::DenseMatrix *m = new mfem::DenseMatrix;
        form->ComputeElementMatrix(elno, *m);
        free_matr = true;
        return m;
    }
private:
    mfem::SparseMatrix *assembled;
    mfem::ParBilinearForm *form;
};

// == MultilevelParameters (inc/ml.hpp:59-114): the same class as saamge_amd::api::MultilevelParameters
// (11-argument constructor of inc/ml.hpp:66-70, getters / setters of :80-101)
typedef saamge_amd::api::MultilevelParameters MultilevelParameters;

// == smoother plug (inc/smpr.hpp:59-60) and polynomial smoother data (inc/smpr.hpp:89-108)
typedef void (*smpr_ft)(mfem::HypreParMatrix &A, const mfem::Vector &b, mfem::Vector &x, void *data);
struct smpr_poly_data_t {
    saamge_amd_hierarchy *h;     // the hierarchy that owns D^-1 and the roots
    int level;
};
// smpr_sym_poly (inc/smpr.hpp, src/smpr.cpp:213-234): x += M^-1 (b - A x) with the level's SAS polynomial
inline void smpr_sym_poly(mfem::HypreParMatrix &A, const mfem::Vector &b, mfem::Vector &x, void *data) {
    (void)A;
    smpr_poly_data_t *d = (smpr_poly_data_t *)data;
    if (saamge_amd_smoother(d->h, d->level, b.GetData(), x.GetData())) mfem::mfem_error(saamge_amd_last_error());
}

namespace detail { struct DistLayout; }
struct interp_data_t;       // (opaque here: the per-AE eigenpairs live on the GPU; saamge_amd_get_ae_eigens exports them)

// == tg_data_t (inc/tg_data.hpp:47-83).  Ac / interp / restr are host copies for inspection; the GPU owns the
// operators the cycle uses.  coarse_solver may be assigned by the caller (test/algebraic/algebraic.cpp:282-283):
// VCycleSolver::Mult then routes the coarsest solve through it.
struct tg_data_t {
    interp_data_t *interp_data = nullptr;
    mfem::HypreParMatrix *Ac = nullptr, *interp = nullptr, *restr = nullptr;
    mfem::SparseMatrix *ltent_interp = nullptr;
    mfem::HypreParMatrix *tent_interp = nullptr, *scaling_P = nullptr;
    bool smooth_interp = false;
    double theta = 0.0;
    smpr_ft pre_smoother = smpr_sym_poly, post_smoother = smpr_sym_poly;
    mfem::Solver *coarse_solver = nullptr;
    smpr_poly_data_t *poly_data = nullptr;
    bool use_w_cycle = false;
    int polynomial_coarse_space = -1;
    bool doing_spectral = true;
    int tag = 0;
    ElementMatrixProvider *elem_data = nullptr;
    // ---- this library ----
    saamge_amd_hierarchy *h = nullptr;     // shared by the tg_data of every level of one ml_data_t
    int level = 0;
    bool owns_h = false;
    tg_data_t *coarser_tg = nullptr;       // the next level's tg_data (NULL on the last spectral level)
    // what tg_init_data recorded for tg_build_hierarchy (inc/tg.hpp:428-432,478-481)
    int init_nu_pro = 0, init_nu_relax = 3;
    double init_smooth_drop_tol = 0.0;
    bool init_use_arpack = false;
    mfem::SparseMatrix *Ac_diag = nullptr, *interp_diag = nullptr, *restr_diag = nullptr;
    HYPRE_Int row_starts[2][2];
    // a hierarchy built from a DISTRIBUTED HypreParMatrix: the ranks' row blocks (vectors at this boundary are the rank's
    // true-dof block) and the MPI collectives the library calls back into; owned by the finest level's tg_data
    detail::DistLayout *dist = nullptr;
};

// == levels (inc/levels.hpp:47-64)
struct levels_level_t {
    levels_level_t *finer = nullptr;
    agg_partitioning_relations_t *agg_part_rels = nullptr;
    tg_data_t *tg_data = nullptr;
    levels_level_t *coarser = nullptr;
};
struct levels_list_t {
    int num_levels = 0;
    levels_level_t *finest = nullptr, *coarsest = nullptr;
};
inline levels_level_t *levels_list_get_level(const levels_list_t &l, int num) {     // inc/levels.hpp
    levels_level_t *p = l.finest;
    for (int i = 0; i < num && p; ++i) p = p->coarser;
    return p;
}
struct ml_data_t {         // inc/ml.hpp:118-120
    levels_list_t levels_list;
};

// Coarse partitions: partition(level k >= 1, number of level-k elements = AEs of level k-1, number of parts,
// elem_to_elem of level k, out[n_elem]).  Default: contiguous index ranges.  Plug METIS here to reproduce the
// reference's partitions (src/aggregates.cpp:1797-1804).
typedef std::function<void(int level, int n_elem, int nparts, const mfem::Table &elem_to_elem, int *partition)> ml_partitioner_t;
inline ml_partitioner_t &ml_coarse_partitioner() {
    static ml_partitioner_t p = [](int, int n_elem, int nparts, const mfem::Table &, int *out) {
        for (int e = 0; e < n_elem; ++e) out[e] = (int)((long long)e * nparts / n_elem);
    };
    return p;
}
inline void ml_set_coarse_partitioner(const ml_partitioner_t &p) { ml_coarse_partitioner() = p; }
// Fine partition for the entry points that derive it themselves in the reference (SpectralAMGSolver ->
// fem_create_partitioning -> METIS, src/solve.cpp:185-187, src/part.cpp:170-183): partition(level 0, NE, nparts,
// elem_to_elem, out[NE]).  Default: contiguous element ranges.
inline ml_partitioner_t &ml_fine_partitioner() {
    static ml_partitioner_t p = [](int, int n_elem, int nparts, const mfem::Table &, int *out) {
        for (int e = 0; e < n_elem; ++e) out[e] = (int)((long long)e * nparts / n_elem);
    };
    return p;
}
inline void ml_set_fine_partitioner(const ml_partitioner_t &p) { ml_fine_partitioner() = p; }

namespace detail {

// ---- MPI-backed collectives for the C ABI's callback plug (staged through host memory) --------------------------
struct MpiCollectives {
    MPI_Comm comm;
    int rank, world;
    std::vector<char> sbuf, rbuf;
    explicit MpiCollectives(MPI_Comm c) : comm(c), rank(0), world(1) {
        MPI_Comm_rank(c, &rank);
        MPI_Comm_size(c, &world);
    }
    static int allgather(void *ctx, void *buf, const long long *off) {           // in place, rank r owns bytes [off[r], off[r+1])
        MpiCollectives &m = *(MpiCollectives *)ctx;
        const long long total = off[m.world], lo = off[m.rank], hi = off[m.rank + 1];
        m.sbuf.resize((size_t)(hi - lo));
        m.rbuf.resize((size_t)total);
        if (saamge_amd_memcpy(m.sbuf.data(), (const char *)buf + lo, hi - lo)) return 1;
        std::vector<int> cnt((size_t)m.world), dsp((size_t)m.world);
        for (int r = 0; r < m.world; ++r) { cnt[(size_t)r] = (int)(off[r + 1] - off[r]); dsp[(size_t)r] = (int)off[r]; }
        if (total >= (1ll << 31)) return 2;      // (one MPI call: split larger gathers on the caller's side)
        if (MPI_Allgatherv(m.sbuf.data(), (int)(hi - lo), MPI_BYTE, m.rbuf.data(), cnt.data(), dsp.data(), MPI_BYTE, m.comm)) return 3;
        if (lo > 0 && saamge_amd_memcpy(buf, m.rbuf.data(), lo)) return 1;
        if (total > hi && saamge_amd_memcpy((char *)buf + hi, m.rbuf.data() + hi, total - hi)) return 1;
        return 0;
    }
    static int allreduce(void *ctx, double *buf, long long count) {
        MpiCollectives &m = *(MpiCollectives *)ctx;
        std::vector<double> h((size_t)count), out((size_t)count);
        if (saamge_amd_memcpy(h.data(), buf, 8 * count)) return 1;
        if (MPI_Allreduce(h.data(), out.data(), (int)count, MPI_DOUBLE, MPI_SUM, m.comm)) return 3;
        return saamge_amd_memcpy(buf, out.data(), 8 * count);
    }
    static int alltoallv(void *ctx, const void *send, const long long *soff, void *recv, const long long *roff) {
        MpiCollectives &m = *(MpiCollectives *)ctx;
        m.sbuf.resize((size_t)soff[m.world]);
        m.rbuf.resize((size_t)roff[m.world]);
        if (soff[m.world] && saamge_amd_memcpy(m.sbuf.data(), send, soff[m.world])) return 1;
        std::vector<int> sc((size_t)m.world), sd((size_t)m.world), rc((size_t)m.world), rd((size_t)m.world);
        for (int r = 0; r < m.world; ++r) {
            sc[(size_t)r] = (int)(soff[r + 1] - soff[r]); sd[(size_t)r] = (int)soff[r];
            rc[(size_t)r] = (int)(roff[r + 1] - roff[r]); rd[(size_t)r] = (int)roff[r];
        }
        if (MPI_Alltoallv(m.sbuf.data(), sc.data(), sd.data(), MPI_BYTE, m.rbuf.data(), rc.data(), rd.data(), MPI_BYTE, m.comm)) return 3;
        if (roff[m.world] && saamge_amd_memcpy(recv, m.rbuf.data(), roff[m.world])) return 1;
        return 0;
    }
};
// what a distributed hierarchy needs at the vector boundary: the ranks' row blocks of the global numbering
struct DistLayout {
    MpiCollectives *coll = nullptr;           // owned
    std::vector<int> row_count, row_first;    // per rank
    int nloc = 0, nglob = 0;
    ~DistLayout() { delete coll; }
    // the rank's block b -> the global vector g (every rank gets all of it)
    void gather(const double *b, std::vector<double> &g) const {
        g.resize((size_t)nglob);
        MPI_Allgatherv(const_cast<double *>(b), nloc, MPI_DOUBLE, g.data(), const_cast<int *>(row_count.data()),
                       const_cast<int *>(row_first.data()), MPI_DOUBLE, coll->comm);
    }
};
inline bool is_distributed(const mfem::HypreParMatrix &A) {
    return A.GetGlobalNumRows() != A.Height() || A.GetGlobalNumCols() != A.Width();
}
// local dof -> global true dof through Dof_TrueDof (one entry per row: in diag for a dof this rank owns, in offd otherwise)
inline void local_to_true(const mfem::HypreParMatrix &D, std::vector<int> &gdof, std::vector<char> &owned) {
    mfem::SparseMatrix diag, offd;
    HYPRE_Int *cmap = nullptr;
    D.GetDiag(diag);
    D.GetOffd(offd, cmap);
    const int nl = diag.Height();
    const HYPRE_Int first = D.ColPart()[0];
    gdof.assign((size_t)nl, -1);
    owned.assign((size_t)nl, 0);
    for (int i = 0; i < nl; ++i) {
        if (diag.GetI()[i + 1] > diag.GetI()[i]) { gdof[(size_t)i] = (int)(first + diag.GetJ()[diag.GetI()[i]]); owned[(size_t)i] = 1; }
        else if (offd.GetI()[i + 1] > offd.GetI()[i]) gdof[(size_t)i] = (int)cmap[offd.GetJ()[offd.GetI()[i]]];
        else mfem::mfem_error("saamge_amd: a local dof without a true dof in Dof_TrueDof");
    }
}

struct HostCsr {
    std::vector<int> I, J;
    std::vector<double> V;
};
// the rank's matrix as sorted CSR (hypre keeps the diagonal entry first in each row: src/mbox.cpp:1664-1667)
inline HostCsr csr_of(mfem::HypreParMatrix &A) {
    // one MPI rank per hierarchy: a matrix distributed over several ranks has an off-diagonal block that GetDiag()
    // does not return -- refused loudly instead of solving a truncated operator (multi-GPU runs go through the C ABI's
    // rank / world parameters)
    if (A.GetGlobalNumRows() != A.Height() || A.GetGlobalNumCols() != A.Width())
        mfem::mfem_error("saamge_amd: the HypreParMatrix is distributed over several MPI ranks (global size != local size); "
                         "this adaptor takes the rank's whole matrix");
    mfem::SparseMatrix diag;
    A.GetDiag(diag);
    const int n = diag.Height();
    const int *I = diag.GetI(), *J = diag.GetJ();
    const double *V = diag.GetData();
    HostCsr c;
    c.I.assign(I, I + n + 1);
    c.J.resize((size_t)I[n]);
    c.V.resize((size_t)I[n]);
    std::vector<std::pair<int, double> > row;
    for (int i = 0; i < n; ++i) {
        row.clear();
        for (int k = I[i]; k < I[i + 1]; ++k) row.push_back(std::make_pair(J[k], V[k]));
        std::sort(row.begin(), row.end());
        for (int k = I[i]; k < I[i + 1]; ++k) { c.J[(size_t)k] = row[(size_t)(k - I[i])].first; c.V[(size_t)k] = row[(size_t)(k - I[i])].second; }
    }
    return c;
}
inline mfem::Table *table_from(const saamge_amd_hierarchy *h, int level, int which) {
    int nrows = 0;
    long long nconn = 0;
    if (saamge_amd_get_table(h, level, which, &nrows, &nconn, nullptr, nullptr)) mfem::mfem_error(saamge_amd_last_error());
    std::vector<int> I((size_t)nrows + 1), J((size_t)nconn);
    if (saamge_amd_get_table(h, level, which, nullptr, nullptr, I.data(), J.data())) mfem::mfem_error(saamge_amd_last_error());
    mfem::Table *t = new mfem::Table;
    t->SetDims(nrows, (int)nconn);
    std::copy(I.begin(), I.end(), t->GetI());
    std::copy(J.begin(), J.end(), t->GetJ());
    return t;
}
// host copies of the level's operators as HypreParMatrix (one rank)
inline void fetch_operators(tg_data_t &tg, MPI_Comm comm) {
    long long info[16];
    if (saamge_amd_level_info(tg.h, tg.level, info)) mfem::mfem_error(saamge_amd_last_error());
    const int n = (int)info[0], nc = (int)info[4];
    struct Sel { int which, rows, cols; long long nnz; mfem::SparseMatrix **diag; mfem::HypreParMatrix **par; };
    Sel sel[3] = {{1, n, nc, info[5], &tg.interp_diag, &tg.interp},
                  {2, nc, n, info[5], &tg.restr_diag, &tg.restr},
                  {3, nc, nc, info[6], &tg.Ac_diag, &tg.Ac}};
    tg.row_starts[0][0] = 0; tg.row_starts[0][1] = n;
    tg.row_starts[1][0] = 0; tg.row_starts[1][1] = nc;
    for (int q = 0; q < 3; ++q) {
        int *I = new int[sel[q].rows + 1];
        int *J = new int[sel[q].nnz];
        double *V = new double[sel[q].nnz];
        if (saamge_amd_get_csr(tg.h, tg.level, sel[q].which, I, J, V)) mfem::mfem_error(saamge_amd_last_error());
        *sel[q].diag = new mfem::SparseMatrix(I, J, V, sel[q].rows, sel[q].cols);      // takes ownership of the arrays
        HYPRE_Int *rs = sel[q].rows == n ? tg.row_starts[0] : tg.row_starts[1];
        HYPRE_Int *cs = sel[q].cols == n ? tg.row_starts[0] : tg.row_starts[1];
        *sel[q].par = new mfem::HypreParMatrix(comm, sel[q].rows, sel[q].cols, rs, cs, *sel[q].diag);
    }
}
inline int coarse_solver_trampoline(void *ctx, int n, const double *rc, double *xc) {
    mfem::Solver *s = *(mfem::Solver **)ctx;
    mfem::Vector R(const_cast<double *>(rc), n), X(xc, n);
    X = 0.0;                        // "XC pre-zeroed", src/tg.cpp:110-126
    s->Mult(R, X);
    return 0;
}

// the smpr_ft plug: the C ABI calls back with host arrays, the caller's smoother gets the level's operator and its
// polynomial data exactly as tg_cycle_atb passes them (src/tg.cpp:113,131)
struct SmootherPlug {
    tg_data_t *tg;
    mfem::HypreParMatrix *A;
};
inline int smoother_trampoline(SmootherPlug *p, smpr_ft fn, int n, const double *b, double *x) {
    if (!p->A) mfem::mfem_error("VCycleSolver: a caller's smoother needs the level's operator (SetOperator / tg_data->Ac)");
    mfem::Vector B(const_cast<double *>(b), n), X(x, n);
    fn(*p->A, B, X, p->tg->poly_data);
    return 0;
}
inline int pre_smoother_trampoline(void *ctx, int level, int n, const double *b, double *x) {
    (void)level;
    SmootherPlug *p = (SmootherPlug *)ctx;
    return smoother_trampoline(p, p->tg->pre_smoother, n, b, x);
}
inline int post_smoother_trampoline(void *ctx, int level, int n, const double *b, double *x) {
    (void)level;
    SmootherPlug *p = (SmootherPlug *)ctx;
    return smoother_trampoline(p, p->tg->post_smoother, n, b, x);
}

}  // namespace detail

// ml_produce_data (inc/ml.hpp:192-194, src/ml.cpp:379-472).  Takes ownership of the provider (freed by
// ml_free_data, like tg_free_data does in the reference, src/tg.cpp:946); agg_part_rels stays the caller's.
// GetMatrix is called once per element.
inline ml_data_t *ml_produce_data(mfem::HypreParMatrix &Ag, agg_partitioning_relations_t *agg_part_rels,
                                  ElementMatrixProvider *elem_data_finest, const MultilevelParameters &mlp) {
    const int nco = mlp.get_num_coarsenings();
    agg_partitioning_relations_t &r = *agg_part_rels;
    const bool distributed = detail::is_distributed(Ag);
    if (distributed && !r.Dof_TrueDof)
        mfem::mfem_error("ml_produce_data: a distributed HypreParMatrix needs agg_create_partitioning_fine's dof_truedof argument");
    detail::HostCsr A;
    if (!distributed) A = detail::csr_of(Ag);
    const int n = Ag.Height(), NE = r.NE;
    // element matrices, raw (row-major nde x nde per element); uniform element size required by the batched assembly
    const int nde = r.elem_to_dof->RowSize(0);
    std::vector<double> elmat((size_t)NE * nde * nde);
    for (int e = 0; e < NE; ++e) {
        if (r.elem_to_dof->RowSize(e) != nde) mfem::mfem_error("ml_produce_data: elements with different numbers of dofs are not supported");
        bool free_matr = false;
        mfem::Matrix *m = elem_data_finest->GetMatrix(e, free_matr);
        for (int a = 0; a < nde; ++a)
            for (int b = 0; b < nde; ++b) elmat[((size_t)e * nde + a) * nde + b] = m->Elem(a, b);
        if (free_matr) delete m;
    }
    // partitions of every coarsening: level 0 from agg_part_rels, coarser ones from the partitioner hook on the
    // AE adjacency graph elem_to_elem_{k+1} = AE_to_elem x elem_to_elem x elem_to_AE (src/aggregates.cpp:1768-1771)
    std::vector<std::vector<int> > parts((size_t)nco);
    std::vector<const int *> part_ptrs((size_t)nco);
    parts[0].assign(r.partitioning, r.partitioning + NE);
    mfem::Table *e2e = r.elem_to_elem;
    std::vector<mfem::Table *> owned;
    for (int k = 1; k < nco; ++k) {
        const int n_el_prev = (int)parts[(size_t)k - 1].size(), n_el = mlp.get_nparts(k - 1);
        mfem::Table e2AE;                                  // element -> AE of level k-1
        e2AE.MakeI(n_el_prev);
        for (int e = 0; e < n_el_prev; ++e) e2AE.AddAColumnInRow(e);
        e2AE.MakeJ();
        for (int e = 0; e < n_el_prev; ++e) e2AE.AddConnection(e, parts[(size_t)k - 1][(size_t)e]);
        e2AE.ShiftUpI();
        mfem::Table *AE2e = mfem::Transpose(e2AE);
        mfem::Table *next = nullptr;
        if (e2e) {
            mfem::Table *t = mfem::Mult(*AE2e, *e2e);
            next = mfem::Mult(*t, e2AE);
            delete t;
        } else {
            next = new mfem::Table;
        }
        delete AE2e;
        owned.push_back(next);
        parts[(size_t)k].resize((size_t)n_el);
        ml_coarse_partitioner()(k, n_el, mlp.get_nparts(k), *next, parts[(size_t)k].data());
        e2e = next;
    }
    for (size_t i = 0; i < owned.size(); ++i) delete owned[i];
    for (int k = 0; k < nco; ++k) part_ptrs[(size_t)k] = parts[(size_t)k].data();
    saamge_amd_params p = mlp.p;
    p.testmesh = r.testmesh ? 1 : 0;
    std::vector<int> nparts((size_t)nco);
    for (int k = 0; k < nco; ++k) nparts[(size_t)k] = mlp.get_nparts(k);
    saamge_amd_hierarchy *h = nullptr;
    detail::DistLayout *layout = nullptr;
    if (!distributed) {
        std::vector<signed char> bdr((size_t)n);
        for (int i = 0; i < n; ++i) bdr[(size_t)i] = (signed char)r.agg_flags[i];
        if (saamge_amd_ml_produce_data(n, A.I.data(), A.J.data(), A.V.data(), NE, nde, r.elem_to_dof->GetJ(), elmat.data(),
                                       bdr.data(), part_ptrs.data(), nparts.data(), &p, nullptr, &h))
            mfem::mfem_error(saamge_amd_last_error());
    } else {
        // per-rank inputs (saamge_amd_ml_produce_data_parcsr): the rank's row block in hypre's own split, its elements with
        // local dofs mapped to true dofs, the flags of the rows it owns, its own partitions (local agglomerate ids)
        layout = new detail::DistLayout;
        layout->coll = new detail::MpiCollectives(Ag.GetComm());
        const int world = layout->coll->world, rank = layout->coll->rank;
        mfem::SparseMatrix diag, offd;
        HYPRE_Int *cmap = nullptr;
        Ag.GetDiag(diag);
        Ag.GetOffd(offd, cmap);
        std::vector<long long> cmap64((size_t)std::max(offd.Width(), 1), 0);
        for (int c = 0; c < offd.Width(); ++c) cmap64[(size_t)c] = (long long)cmap[c];
        std::vector<int> gdof;
        std::vector<char> owned;
        detail::local_to_true(*r.Dof_TrueDof, gdof, owned);
        if ((int)gdof.size() != r.ND) mfem::mfem_error("ml_produce_data: Dof_TrueDof does not match the local dofs of agg_part_rels");
        std::vector<int> e2d_glob((size_t)NE * nde);
        const int *J = r.elem_to_dof->GetJ();
        for (size_t q = 0; q < e2d_glob.size(); ++q) e2d_glob[q] = gdof[(size_t)J[q]];
        const HYPRE_Int first_row = Ag.RowPart()[0];
        std::vector<signed char> bdr_own((size_t)n, (signed char)AGG_OWNED_FLAG);
        for (int i = 0; i < r.ND; ++i)
            if (owned[(size_t)i]) bdr_own[(size_t)(gdof[(size_t)i] - first_row)] = (signed char)r.agg_flags[i];
        layout->nloc = n;
        layout->row_count.assign((size_t)world, 0);
        layout->row_first.assign((size_t)world, 0);
        MPI_Allgather(const_cast<int *>(&n), 1, MPI_INT, layout->row_count.data(), 1, MPI_INT, layout->coll->comm);
        for (int q = 1; q < world; ++q) layout->row_first[(size_t)q] = layout->row_first[(size_t)q - 1] + layout->row_count[(size_t)q - 1];
        layout->nglob = layout->row_first[(size_t)world - 1] + layout->row_count[(size_t)world - 1];
        if (!p.allgather) {          // (a driver that installed a communicator of its own keeps it)
            p.rank = rank;
            p.world = world;
            p.allgather = detail::MpiCollectives::allgather;
            p.allreduce_sum = detail::MpiCollectives::allreduce;
            p.alltoallv = detail::MpiCollectives::alltoallv;
            p.allgather_ctx = layout->coll;
            p.comm_stream_ordered = 0;
        }
        saamge_amd_parcsr P;
        P.global_rows = (long long)Ag.GetGlobalNumRows();
        P.row_starts = nullptr;
        P.nrows = n;
        P.diag_i = diag.GetI(); P.diag_j = diag.GetJ(); P.diag_a = diag.GetData();
        P.offd_i = offd.GetI(); P.offd_j = offd.GetJ(); P.offd_a = offd.GetData();
        P.num_cols_offd = offd.Width();
        P.col_map_offd = cmap64.data();
        if (saamge_amd_ml_produce_data_parcsr(&P, NE, nde, e2d_glob.data(), elmat.data(), bdr_own.data(), part_ptrs.data(),
                                              nparts.data(), &p, nullptr, &h))
            mfem::mfem_error(saamge_amd_last_error());
    }
    // the list of levels (ml_produce_hierarchy_from_level, src/ml.cpp:111-236)
    ml_data_t *ml = new ml_data_t;
    levels_level_t *prev = nullptr;
    for (int k = 0; k < nco; ++k) {
        levels_level_t *lv = new levels_level_t;
        tg_data_t *tg = new tg_data_t;
        tg->h = h;
        tg->level = k;
        tg->owns_h = (k == 0);
        tg->theta = mlp.get_theta(k);
        tg->smooth_interp = mlp.get_smooth_interp(k);
        tg->polynomial_coarse_space = mlp.get_polynomial_coarse_space(k);
        tg->poly_data = new smpr_poly_data_t;
        tg->poly_data->h = h;
        tg->poly_data->level = k;
        tg->elem_data = (k == 0) ? elem_data_finest : nullptr;
        if (k == 0) tg->dist = layout;
        if (!distributed) detail::fetch_operators(*tg, Ag.GetComm());     // (host copies of P / R / Ac: single-rank hierarchies only)
        lv->tg_data = tg;
        lv->agg_part_rels = (k == 0) ? agg_part_rels : nullptr;
        lv->finer = prev;
        if (prev) { prev->coarser = lv; prev->tg_data->coarser_tg = tg; } else ml->levels_list.finest = lv;
        prev = lv;
    }
    ml->levels_list.coarsest = prev;
    ml->levels_list.num_levels = nco;
    return ml;
}

inline void tg_free_data(tg_data_t *tg) {           // inc/tg.hpp:576, src/tg.cpp:930-950
    if (!tg) return;
    delete tg->Ac; delete tg->interp; delete tg->restr;
    delete tg->Ac_diag; delete tg->interp_diag; delete tg->restr_diag;
    delete tg->poly_data;
    delete tg->elem_data;
    if (tg->owns_h) saamge_amd_ml_free_data(tg->h);
    delete tg->dist;          // (after the hierarchy: its collectives are the hierarchy's callbacks)
    delete tg;
}
inline void ml_free_data(ml_data_t *ml) {           // inc/ml.hpp:196
    if (!ml) return;
    levels_level_t *lv = ml->levels_list.coarsest;  // (the finest level owns the hierarchy: freed last)
    while (lv) {
        levels_level_t *f = lv->finer;
        tg_free_data(lv->tg_data);
        delete lv;
        lv = f;
    }
    delete ml;
}

// tg_produce_data (inc/tg.hpp:565-570): the two-level method = one coarsening
inline tg_data_t *tg_produce_data(mfem::HypreParMatrix &Ag, const agg_partitioning_relations_t &agg_part_rels, int nu_pro,
                                  int nu_relax, ElementMatrixProvider *elem_data_finest, double theta, bool smooth_interp,
                                  int polynomial_coarse_arg, bool use_arpack, bool avoid_ess_bdr_dofs) {
    (void)avoid_ess_bdr_dofs;       // always true in the reference (src/ml.cpp:64)
    int nparts = agg_part_rels.nparts;
    MultilevelParameters mlp(1, &nparts, smooth_interp ? nu_pro : 0, nu_pro, nu_relax, theta, theta, polynomial_coarse_arg,
                             false, use_arpack, agg_part_rels.do_aggregates);
    ml_data_t *ml = ml_produce_data(Ag, const_cast<agg_partitioning_relations_t *>(&agg_part_rels), elem_data_finest, mlp);
    tg_data_t *tg = ml->levels_list.finest->tg_data;
    delete ml->levels_list.finest;
    delete ml;
    return tg;
}
// tg_init_data + tg_build_hierarchy (inc/tg.hpp:428-432, 478-481; src/tg.cpp:402-430, 502-540): the split form of
// tg_produce_data.  tg_init_data records the parameters (and the smoother plugs: pre_smoother / post_smoother default to
// smpr_sym_poly like the TG options, a caller may assign others before or after building); tg_build_hierarchy runs the
// setup into the SAME struct.  use_arpack is accepted and ignored: the local eigenproblems are always solved directly.
inline tg_data_t *tg_init_data(mfem::HypreParMatrix &A, const agg_partitioning_relations_t &agg_part_rels, int nu_pro,
                               int nu_relax, double theta, bool smooth_interp, double smooth_drop_tol, bool use_arpack) {
    (void)A; (void)agg_part_rels;
    tg_data_t *tg = new tg_data_t;
    tg->theta = theta;
    tg->smooth_interp = smooth_interp;
    tg->init_nu_pro = nu_pro;
    tg->init_nu_relax = nu_relax;
    tg->init_smooth_drop_tol = smooth_drop_tol;
    tg->init_use_arpack = use_arpack;
    return tg;
}
inline void tg_build_hierarchy(mfem::HypreParMatrix &Ag, tg_data_t &tg_data, const agg_partitioning_relations_t &agg_part_rels,
                               ElementMatrixProvider *elem_data, bool avoid_ess_bdr_dofs) {
    (void)avoid_ess_bdr_dofs;
    if (tg_data.h) mfem::mfem_error("tg_build_hierarchy: this tg_data already holds a hierarchy");
    if (!elem_data) mfem::mfem_error("tg_build_hierarchy: coarse-level construction (elem_data == NULL) happens inside ml_produce_data here");
    int nparts = agg_part_rels.nparts;
    MultilevelParameters mlp(1, &nparts, tg_data.smooth_interp ? tg_data.init_nu_pro : 0, tg_data.init_nu_pro, tg_data.init_nu_relax,
                             tg_data.theta, tg_data.theta, tg_data.polynomial_coarse_space, false, tg_data.init_use_arpack,
                             agg_part_rels.do_aggregates);
    mlp.set_smooth_drop_tol(tg_data.init_smooth_drop_tol);
    ml_data_t *ml = ml_produce_data(Ag, const_cast<agg_partitioning_relations_t *>(&agg_part_rels), elem_data, mlp);
    tg_data_t *built = ml->levels_list.finest->tg_data;
    // move what the setup made into the caller's struct; its own choices (plugs, tag, coarse_solver) stay
    const smpr_ft pre = tg_data.pre_smoother, post = tg_data.post_smoother;
    mfem::Solver *cs = tg_data.coarse_solver;
    const int tag = tg_data.tag;
    const int inp = tg_data.init_nu_pro, inr = tg_data.init_nu_relax;
    const double idt = tg_data.init_smooth_drop_tol;
    const bool iua = tg_data.init_use_arpack;
    tg_data = *built;
    tg_data.pre_smoother = pre; tg_data.post_smoother = post; tg_data.coarse_solver = cs; tg_data.tag = tag;
    tg_data.init_nu_pro = inp; tg_data.init_nu_relax = inr; tg_data.init_smooth_drop_tol = idt; tg_data.init_use_arpack = iua;
    delete built;                      // (shallow: every owned pointer moved to tg_data)
    delete ml->levels_list.finest;
    delete ml;
}

// tg_produce_data_algebraic (inc/tg.hpp:518-523, src/tg.cpp:862-886): the element-free two-level method -- only the
// matrix and the dof -> AE map of agg_part_rels (fem_create_partitioning_from_matrix makes every dof its own
// "element": partitioning[dof] = AE, NE = ND).  The agglomerate matrices are principal submatrices of A with
// diagonal compensation (ExtractSubMatrices, src/tg.cpp:579-672) or, with use_window, the window matrices
// (WindowSubMatrices, src/tg.cpp:741-858).  Alocal is the diagonal block of Ag on one rank and is not read.
inline tg_data_t *tg_produce_data_algebraic(const mfem::SparseMatrix &Alocal, mfem::HypreParMatrix &Ag,
                                            const agg_partitioning_relations_t &agg_part_rels, int nu_pro, int nu_relax,
                                            double spectral_tol, bool smooth_interp, int polynomial_coarse_arg, bool use_window,
                                            bool use_arpack, bool avoid_ess_bdr_dofs) {
    (void)Alocal; (void)avoid_ess_bdr_dofs;
    const agg_partitioning_relations_t &r = agg_part_rels;
    const int n = Ag.Height();
    if (r.NE != n || !r.partitioning) mfem::mfem_error("tg_produce_data_algebraic: agg_part_rels must map every dof to its AE (NE = ND)");
    int nparts = r.nparts;
    MultilevelParameters mlp(1, &nparts, smooth_interp ? nu_pro : 0, nu_pro, nu_relax, spectral_tol, spectral_tol,
                             polynomial_coarse_arg, false, use_arpack, r.do_aggregates);
    saamge_amd_params p = mlp.p;
    p.algebraic = use_window ? 2 : 1;
    detail::HostCsr A = detail::csr_of(Ag);
    const int *part_ptr = r.partitioning;
    saamge_amd_hierarchy *h = nullptr;
    if (saamge_amd_ml_produce_data(n, A.I.data(), A.J.data(), A.V.data(), n, 1, nullptr, nullptr, nullptr, &part_ptr, &nparts, &p,
                                   nullptr, &h))
        mfem::mfem_error(saamge_amd_last_error());
    tg_data_t *tg = new tg_data_t;
    tg->h = h;
    tg->level = 0;
    tg->owns_h = true;
    tg->theta = spectral_tol;
    tg->smooth_interp = smooth_interp;
    tg->polynomial_coarse_space = polynomial_coarse_arg;
    tg->poly_data = new smpr_poly_data_t;
    tg->poly_data->h = h;
    tg->poly_data->level = 0;
    tg->elem_data = nullptr;
    detail::fetch_operators(*tg, Ag.GetComm());
    return tg;
}
// tg_fillin_coarse_operator (inc/tg.hpp:641-657, 711-732): "if Ac is empty, it is computed".  The setup here always
// builds Ac = P^T A P on the device; the call only (re)fetches the host copy a caller may have freed with
// tg_free_coarse_operator.  The coarse solver stays whatever tg_data->coarse_solver is (test/algebraic/
// algebraic.cpp:282-283 assigns its own right after this call).
inline void tg_fillin_coarse_operator(mfem::HypreParMatrix &A, tg_data_t *tg_data, bool perform_solve_init) {
    (void)perform_solve_init;
    if (!tg_data || !tg_data->interp || !tg_data->restr) mfem::mfem_error("tg_fillin_coarse_operator: no interpolation");
    if (tg_data->Ac) return;
    delete tg_data->interp; delete tg_data->restr; delete tg_data->interp_diag; delete tg_data->restr_diag;
    tg_data->interp = tg_data->restr = nullptr;
    tg_data->interp_diag = tg_data->restr_diag = nullptr;
    detail::fetch_operators(*tg_data, A.GetComm());
}
// tg_free_coarse_operator (inc/tg.hpp:659-667, 735-745): the caller's copy of Ac goes; the device hierarchy keeps its own
inline void tg_free_coarse_operator(tg_data_t &tg_data) {
    delete tg_data.Ac;
    delete tg_data.Ac_diag;
    tg_data.Ac = nullptr;
    tg_data.Ac_diag = nullptr;
}
// tg_update_coarse_operator (inc/tg.hpp:610-612): the matrix values changed, interpolation kept; the coarsest solver
// is set up again as coarse_direct asks (true: the explicit inverse, false: CG on the coarsest operator), as the
// reference does (src/tg.cpp, tg_update_coarse_operator -> solve_init with the new Ac).  perform_solve_init = false in
// the reference leaves the old solver object in place for the caller to replace: here the caller's coarse_solver plug
// (saamge_amd_set_coarse_solver), if any, stays installed either way.
inline void tg_update_coarse_operator(mfem::HypreParMatrix &A, tg_data_t *tg_data, bool perform_solve_init, bool coarse_direct) {
    detail::HostCsr c = detail::csr_of(A);
    if (saamge_amd_update_operators2(tg_data->h, c.V.data(), perform_solve_init ? (coarse_direct ? 1 : 2) : -1))
        mfem::mfem_error(saamge_amd_last_error());
}

// Tables built on the GPU, on request (agg_partitioning_relations_t fields of inc/aggregates.hpp:120-179)
inline void agg_fetch_tables(agg_partitioning_relations_t &r, const ml_data_t &ml) {
    const saamge_amd_hierarchy *h = ml.levels_list.finest->tg_data->h;
    r.AE_to_dof = detail::table_from(h, 0, 0);
    r.dof_to_AE = detail::table_from(h, 0, 1);
    r.mis_to_dof = detail::table_from(h, 0, 2);
    r.mis_to_AE = detail::table_from(h, 0, 3);
    r.AE_to_mis = detail::table_from(h, 0, 4);
    r.num_mises = r.mis_to_dof->Size();
    r.mises = new int[r.ND];
    std::vector<signed char> flags((size_t)r.ND);
    if (saamge_amd_get_mis(h, 0, r.mises, nullptr, nullptr, flags.data())) mfem::mfem_error(saamge_amd_last_error());
    for (int i = 0; i < r.ND; ++i) r.agg_flags[i] = (agg_dof_status_t)flags[(size_t)i];
}
inline void ml_get_dims(const ml_data_t &ml, mfem::Array<int> &dims) {      // inc/ml.hpp (ml_get_dims)
    const saamge_amd_hierarchy *h = ml.levels_list.finest->tg_data->h;
    const int nl = ml.levels_list.num_levels;
    dims.SetSize(nl + 1);
    for (int l = 0; l < nl; ++l) {
        long long info[16];
        if (saamge_amd_level_info(h, l, info)) mfem::mfem_error(saamge_amd_last_error());
        dims[l] = (int)info[0];
        dims[l + 1] = (int)info[4];
    }
}

// == VCycleSolver (inc/solve.hpp:129-143, src/solve.cpp:290-323)
class VCycleSolver : public mfem::Solver {
public:
    VCycleSolver(tg_data_t *tg_data_in, bool iterative_mode_)
        : mfem::Solver(tg_data_in->dist ? tg_data_in->dist->nloc : tg_data_in->restr->Width(), iterative_mode_), tg_data(tg_data_in),
          A(NULL), plugged(NULL) {
        if (tg_data->level != 0) mfem::mfem_error("VCycleSolver: only the finest level's tg_data can be cycled from outside");
        for (tg_data_t *t = tg_data; t; t = t->coarser_tg) {
            detail::SmootherPlug pl = {t, NULL};
            plugs.push_back(pl);
            installed.push_back(std::make_pair((smpr_ft)smpr_sym_poly, (smpr_ft)smpr_sym_poly));
        }
    }
    virtual ~VCycleSolver() {}
    virtual void SetOperator(const mfem::Operator &op) {
        A = const_cast<mfem::HypreParMatrix *>(dynamic_cast<const mfem::HypreParMatrix *>(&op));
        if (A == NULL) mfem::mfem_error("VCycleSolver::SetOperator : not HypreParMatrix!");     // src/solve.cpp:301-307
    }
    virtual void Mult(const mfem::Vector &b, mfem::Vector &x) const {
        // a coarse_solver assigned by the caller takes over the coarsest solve (tg_data_t::coarse_solver plug)
        if (tg_data->coarse_solver != plugged) {
            plugged = tg_data->coarse_solver;
            if (saamge_amd_set_coarse_solver(tg_data->h, plugged ? detail::coarse_solver_trampoline : nullptr,
                                             (void *)&tg_data->coarse_solver))
                mfem::mfem_error(saamge_amd_last_error());
        }
        // pre_smoother / post_smoother assigned by the caller (the reference copies the TG options into every tg_data_t,
        // src/tg.cpp:411-414, and tg_cycle_atb calls them, :113,131): anything but smpr_sym_poly is routed through the
        // C ABI's smoother plug, with the level's operator (the finer level's Ac below the finest)
        size_t k = 0;
        for (tg_data_t *t = tg_data; t; t = t->coarser_tg, ++k) {
            plugs[k].A = (k == 0) ? A : plugs[k - 1].tg->Ac;
            if (t->pre_smoother == installed[k].first && t->post_smoother == installed[k].second) continue;
            installed[k] = std::make_pair(t->pre_smoother, t->post_smoother);
            if (saamge_amd_set_smoother(tg_data->h, t->level,
                                        t->pre_smoother != smpr_sym_poly ? detail::pre_smoother_trampoline : nullptr,
                                        t->post_smoother != smpr_sym_poly ? detail::post_smoother_trampoline : nullptr,
                                        (void *)&plugs[k]))
                mfem::mfem_error(saamge_amd_last_error());
        }
        if (tg_data->dist) {
            // distributed: b and x are the rank's true-dof blocks (HypreParVector data); the library's vectors are global
            const detail::DistLayout &d = *tg_data->dist;
            std::vector<double> bg, xg((size_t)d.nglob, 0.0);
            d.gather(b.GetData(), bg);
            if (iterative_mode) d.gather(x.GetData(), xg);
            if (saamge_amd_vcycle(tg_data->h, bg.data(), xg.data(), iterative_mode ? 1 : 0)) mfem::mfem_error(saamge_amd_last_error());
            std::copy(xg.begin() + d.row_first[(size_t)d.coll->rank], xg.begin() + d.row_first[(size_t)d.coll->rank] + d.nloc, x.GetData());
            return;
        }
        if (saamge_amd_vcycle(tg_data->h, b.GetData(), x.GetData(), iterative_mode ? 1 : 0)) mfem::mfem_error(saamge_amd_last_error());
    }
    // the PCG loop of kalchev_pcg on the GPU with this cycle as the preconditioner
    int pcg(const double *b, double *x, int print_iter, int max_num_iter, double RTOLERANCE, double ATOLERANCE,
            bool zero_rhs) const {
        try {
            if (tg_data->dist) {
                const detail::DistLayout &d = *tg_data->dist;
                std::vector<double> bg, xg;
                d.gather(b, bg);
                d.gather(x, xg);
                const int it = saamge_amd::api::kalchev_pcg(tg_data->h, bg.data(), xg.data(), print_iter, max_num_iter, RTOLERANCE,
                                                            ATOLERANCE, zero_rhs);
                std::copy(xg.begin() + d.row_first[(size_t)d.coll->rank], xg.begin() + d.row_first[(size_t)d.coll->rank] + d.nloc, x);
                return it;
            }
            return saamge_amd::api::kalchev_pcg(tg_data->h, b, x, print_iter, max_num_iter, RTOLERANCE, ATOLERANCE, zero_rhs);
        } catch (const std::exception &e) {
            mfem::mfem_error(e.what());
        }
        return 0;
    }
private:
    tg_data_t *tg_data;
    mfem::HypreParMatrix *A;
    mutable mfem::Solver *plugged;
    mutable std::vector<detail::SmootherPlug> plugs;                   // one per level (their addresses are the plug's ctx)
    mutable std::vector<std::pair<smpr_ft, smpr_ft> > installed;       // what the C ABI currently holds per level
};

// == SpectralAMGSolver (inc/solve.hpp:149-177, src/solve.cpp:167-230).  The reference derives the partition
// with METIS from the mesh (fem_create_partitioning -- third party, out of the hot path); here the element
// partition is an argument.  polynomial_coarse = -1: corrected null-space level, like the reference.
class SpectralAMGSolver : public mfem::Solver {
public:
    // The reference's constructor (inc/solve.hpp:160-166, src/solve.cpp:167-212; test/encapsulate/encapsulate.cpp:282-284):
    // essential dofs from fes.GetEssentialVDofs(ess_bdr, .) (fem_find_bdr_dofs), nparts[0] = NE / elems_per_agg, the
    // element partition from the fine-partitioner hook (METIS in the reference), then the 15-argument form below.
    SpectralAMGSolver(mfem::HypreParMatrix &Ag, mfem::ParBilinearForm &aform, mfem::SparseMatrix &Alocal,
                      mfem::Array<int> &ess_bdr, int elems_per_agg, int num_levels, int nu_pro, int nu_relax,
                      double theta, int polynomial_coarse, bool coarse_direct)
        : mfem::Solver(Ag.Height(), false), Ag_(Ag) {
        mfem::ParFiniteElementSpace *fes = aform.ParFESpace();
        const int NE = fes->GetParMesh()->GetNE();
        mfem::Array<int> ess_dofs;
        fes->GetEssentialVDofs(ess_bdr, ess_dofs);
        std::vector<agg_dof_status_t> bdr((size_t)Ag.Height());
        for (int i = 0; i < Ag.Height(); ++i)
            bdr[(size_t)i] = (agg_dof_status_t)(AGG_OWNED_FLAG | (ess_dofs[i] ? AGG_ON_ESS_DOMAIN_BORDER_FLAG : 0));
        mfem::Table *e2d = new mfem::Table(fes->GetElementToDofTable());                    // (owned by agg_part_rels_)
        mfem::Table *e2e = new mfem::Table(fes->GetParMesh()->ElementToElementTable());
        int nparts0 = NE / elems_per_agg;
        if (nparts0 < 1) nparts0 = 1;
        int *partitioning = new int[NE];
        ml_fine_partitioner()(0, NE, nparts0, *e2e, partitioning);
        build(aform, Alocal, bdr.data(), e2d, e2e, partitioning, nparts0, elems_per_agg, num_levels, nu_pro, nu_relax, theta,
              polynomial_coarse, coarse_direct);
    }
    // the same with the topology as arguments (no mesh needed)
    SpectralAMGSolver(mfem::HypreParMatrix &Ag, mfem::ParBilinearForm &aform, mfem::SparseMatrix &Alocal,
                      const agg_dof_status_t *bdr_dofs, mfem::Table *elem_to_dof, mfem::Table *elem_to_elem,
                      int *partitioning, int nparts0, int elems_per_agg, int num_levels, int nu_pro, int nu_relax,
                      double theta, int polynomial_coarse, bool coarse_direct)
        : mfem::Solver(Ag.Height(), false), Ag_(Ag) {
        build(aform, Alocal, bdr_dofs, elem_to_dof, elem_to_elem, partitioning, nparts0, elems_per_agg, num_levels, nu_pro,
              nu_relax, theta, polynomial_coarse, coarse_direct);
    }
    ~SpectralAMGSolver() {
        delete[] nparts_arr_;
        delete v_cycle_;
        ml_free_data(ml_data_);
        agg_free_partitioning(agg_part_rels_);
    }
    void SetOperator(const mfem::Operator &op) { (void)op; }       // "implemented in constructor", src/solve.cpp:219-223
    void Mult(const mfem::Vector &x, mfem::Vector &y) const { v_cycle_->Mult(x, y); }
private:
    void build(mfem::ParBilinearForm &aform, mfem::SparseMatrix &Alocal, const agg_dof_status_t *bdr_dofs,
               mfem::Table *elem_to_dof, mfem::Table *elem_to_elem, int *partitioning, int nparts0, int elems_per_agg,
               int num_levels, int nu_pro, int nu_relax, double theta, int polynomial_coarse, bool coarse_direct) {
        mfem::HypreParMatrix &Ag = Ag_;
        nparts_arr_ = new int[num_levels - 1];
        nparts_arr_[0] = nparts0;
        for (int i = 1; i < num_levels - 1; ++i) {
            nparts_arr_[i] = (int)std::floor((double)nparts_arr_[i - 1] / (double)elems_per_agg + 0.5);
            if (nparts_arr_[i] < 1) nparts_arr_[i] = 1;
        }
        agg_part_rels_ = agg_create_partitioning_fine(Ag, elem_to_dof->Size(), elem_to_dof, elem_to_elem, partitioning,
                                                      bdr_dofs, nparts_arr_, nullptr, false);
        ElementMatrixProvider *emp = new ElementMatrixStandardGeometric(*agg_part_rels_, &Alocal, &aform);
        const bool correct_nulspace = (polynomial_coarse == -1);
        MultilevelParameters mlp(num_levels - 1, nparts_arr_, nu_pro, nu_pro, nu_relax, theta, theta, polynomial_coarse,
                                 correct_nulspace, true, false);
        if (coarse_direct) mlp.set_coarse_direct(true);
        ml_data_ = ml_produce_data(Ag_, agg_part_rels_, emp, mlp);
        v_cycle_ = new VCycleSolver(levels_list_get_level(ml_data_->levels_list, 0)->tg_data, false);
        v_cycle_->SetOperator(Ag_);
    }
    mfem::HypreParMatrix &Ag_;
    int *nparts_arr_;
    agg_partitioning_relations_t *agg_part_rels_;
    ml_data_t *ml_data_;
    VCycleSolver *v_cycle_;
};

// kalchev_pcg (inc/mfem_addons.hpp:276, src/mfem_addons.cpp:106-248) for B = a VCycleSolver of this library:
// the loop runs on the GPU (device vectors, one scalar read-back per iteration); any other preconditioner
// belongs in MFEM's own CGSolver.
inline int kalchev_pcg(const mfem::HypreParMatrix &A, const mfem::Operator &B, const mfem::HypreParVector &b,
                       mfem::HypreParVector &x, int print_iter, int max_num_iter, double RTOLERANCE, double ATOLERANCE,
                       bool zero_rhs) {
    (void)A;
    const VCycleSolver *vc = dynamic_cast<const VCycleSolver *>(&B);
    if (!vc) mfem::mfem_error("kalchev_pcg: B must be a saamge::VCycleSolver of this library");
    return vc->pcg(b.GetData(), x.GetData(), print_iter, max_num_iter, RTOLERANCE, ATOLERANCE, zero_rhs);
}

}  // namespace saamge
#endif  // SAAMGE_AMD_MFEM_HPP
