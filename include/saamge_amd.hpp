// saamge_amd.hpp -- C++ host-side mirror of the SAAMGE solver/operator API for the hot path,
// implemented as thin adaptors over the C ABI (saamge_amd.h).
//
// Two layers:
//   1. namespace saamge_amd::api  -- always available, raw arrays, same names / argument
//      meaning / error behaviour (negative iteration count on failure, inc/tg.hpp:291-293)
//      as the reference's free functions: ml_produce_data, ml_free_data, tg_cycle_atb's
//      entry VCycleSolver::Mult, smpr_sym_poly, kalchev_pcg.
//   2. namespace saamge (only when SAAMGE_AMD_WITH_MFEM is defined, i.e. where <mfem.hpp> and
//      hypre exist -- not in the build container): VCycleSolver / SpectralAMGSolver-shaped
//      mfem::Solver subclasses so that amg/test drivers link unchanged.  It converts
//      HypreParMatrix / mfem::Vector to raw pointers (zero copy on the host, one upload).
#ifndef SAAMGE_AMD_HPP
#define SAAMGE_AMD_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "saamge_amd.h"

namespace saamge_amd {
namespace api {

// == MultilevelParameters (inc/ml.hpp:59-114): constructor argument order follows
// src/ml.cpp:54-91 where it applies.
struct MultilevelParameters {
    saamge_amd_params p;
    std::vector<int> nparts;
    MultilevelParameters(int coarsenings, const int *nparts_arr, int first_nu_pro, int nu_pro,
                         int nu_relax, double first_theta, double theta,
                         bool use_correct_nullspace = false, bool use_arpack = false) {
        saamge_amd_params_default(&p);
        p.correct_nullspace = use_correct_nullspace ? 1 : 0;   // CorrectNullspace on scaling_P (src/ml.cpp:225-236)
        (void)use_arpack;  // the direct (dense) eigensolver path is always taken
        p.num_coarsenings = coarsenings;
        nparts.assign(nparts_arr, nparts_arr + coarsenings);
        for (int i = 0; i < coarsenings && i < SAAMGE_AMD_MAX_LEVELS; ++i) {
            p.theta[i] = i ? theta : first_theta;
            p.nu_pro[i] = i ? nu_pro : first_nu_pro;
            p.nu_relax[i] = nu_relax;
        }
    }
    int get_num_coarsenings() const { return p.num_coarsenings; }
    bool get_use_correct_nullspace() const { return p.correct_nullspace != 0; }
    // polynomial / rigid-body coarse-space extension (ContribTent::ExtendWithPolynomials / RBMs): the
    // modes are evaluated by the caller, n x count column-major
    void set_extra_coarse_modes(const double *modes, int count) { p.extra_modes = modes; p.num_extra_modes = count; }
    // element-free mode (tg_produce_data_algebraic): pass NE = n, nde = 1 and NULL element arrays
    void set_algebraic(bool on, bool use_window = false) { p.algebraic = on ? (use_window ? 2 : 1) : 0; }
    bool get_do_aggregates() const { return p.do_aggregates != 0; }
    void set_do_aggregates(bool on) { p.do_aggregates = on ? 1 : 0; }
    double get_smooth_drop_tol() const { return p.smooth_drop_tol; }
    void set_smooth_drop_tol(double tol) { p.smooth_drop_tol = tol; }
};

// Raw-array view of the reference's setup inputs (HypreParMatrix Ag, elem_to_dof Table,
// ElementMatrixProvider, bdr flags, partitioning arrays).
struct ProblemArrays {
    int n = 0;
    const int *rowptr = nullptr, *col = nullptr;
    const double *val = nullptr;
    int NE = 0, nde = 0;
    const int *elem_to_dof = nullptr;
    const double *elmat = nullptr;
    const signed char *bdr_dofs = nullptr;
    std::vector<const int *> partitions;  // one per coarsening
};

typedef saamge_amd_hierarchy ml_data_t;  // inc/ml.hpp:118-120

// ml_produce_data (inc/ml.hpp:192-194)
inline ml_data_t *ml_produce_data(const ProblemArrays &a, const MultilevelParameters &mlp,
                                  void *stream = nullptr) {
    ml_data_t *h = nullptr;
    if (saamge_amd_ml_produce_data(a.n, a.rowptr, a.col, a.val, a.NE, a.nde, a.elem_to_dof, a.elmat,
                                   a.bdr_dofs, a.partitions.data(), mlp.nparts.data(), &mlp.p, stream, &h))
        throw std::runtime_error(saamge_amd_last_error());
    return h;
}
// ml_free_data (inc/ml.hpp:196)
inline void ml_free_data(ml_data_t *h) { saamge_amd_ml_free_data(h); }
// adapt_update_operators (inc/adapt.hpp, src/adapt.cpp:188-219): new matrix values, same pattern
inline void adapt_update_operators(ml_data_t *h, const double *new_values) {
    if (saamge_amd_update_operators(h, new_values)) throw std::runtime_error(saamge_amd_last_error());
}

// VCycleSolver (inc/solve.hpp:129-143): Mult zeroes x (iterative_mode = false).
class VCycleSolver {
    ml_data_t *h_;
public:
    explicit VCycleSolver(ml_data_t *h, bool iterative_mode = false) : h_(h) {
        if (iterative_mode) throw std::invalid_argument("VCycleSolver: iterative_mode is not supported");
    }
    void Mult(const double *b, double *x) const {
        if (saamge_amd_vcycle_mult(h_, b, x)) throw std::runtime_error(saamge_amd_last_error());
    }
};

// smpr_ft-shaped call (inc/smpr.hpp:59-60): x += M^-1 (b - A x) on `level`
inline void smpr_sym_poly(ml_data_t *h, int level, const double *b, double *x) {
    if (saamge_amd_smoother(h, level, b, x)) throw std::runtime_error(saamge_amd_last_error());
}

// kalchev_pcg (inc/mfem_addons.hpp:276): returns the iteration count, negative when the
// loop did not converge (src/mfem_addons.cpp:226-231).
inline int kalchev_pcg(ml_data_t *h, const double *b, double *x, int max_num_iter, double rtol,
                       double atol, bool zero_guess = true) {
    int iters = 0, conv = 0;
    if (saamge_amd_pcg(h, b, x, rtol, atol, max_num_iter, /*squared_tol=*/0, zero_guess ? 1 : 0,
                       &iters, &conv, nullptr))
        throw std::runtime_error(saamge_amd_last_error());
    return conv ? iters : -iters;
}

}  // namespace api
}  // namespace saamge_amd

#ifdef SAAMGE_AMD_WITH_MFEM
// ---------------------------------------------------------------------------------------
// MFEM-facing adaptors (compiled only where MFEM + hypre headers exist).
// ---------------------------------------------------------------------------------------
#include <mfem.hpp>

namespace saamge {

// Drop-in for saamge::VCycleSolver (inc/solve.hpp:129-143) on top of a hierarchy produced by
// saamge_amd::api::ml_produce_data.  Serial (one rank) HypreParMatrix only in this round.
class VCycleSolver : public mfem::Solver {
    saamge_amd_hierarchy *h_;
public:
    VCycleSolver(saamge_amd_hierarchy *h, bool iterative_mode_)
        : mfem::Solver(0, iterative_mode_), h_(h) {
        if (iterative_mode_) mfem::mfem_error("VCycleSolver: iterative_mode is not supported");
    }
    virtual void SetOperator(const mfem::Operator &op) {
        if (!dynamic_cast<const mfem::HypreParMatrix *>(&op))
            mfem::mfem_error("VCycleSolver::SetOperator : not HypreParMatrix!");  // src/solve.cpp:301-307
        height = width = op.Height();
    }
    virtual void Mult(const mfem::Vector &b, mfem::Vector &x) const {
        if (saamge_amd_vcycle_mult(h_, b.GetData(), x.GetData())) mfem::mfem_error(saamge_amd_last_error());
    }
};

// Raw views of the reference's setup objects; see INTEGRATION.md for the full recipe.
inline saamge_amd::api::ProblemArrays view_problem(mfem::SparseMatrix &Al, const mfem::Table &elem_to_dof,
                                                   const double *elmats, int nde,
                                                   const signed char *bdr_dofs) {
    saamge_amd::api::ProblemArrays a;
    a.n = Al.Height();
    a.rowptr = Al.GetI();
    a.col = Al.GetJ();
    a.val = Al.GetData();
    a.NE = elem_to_dof.Size();
    a.nde = nde;
    a.elem_to_dof = elem_to_dof.GetJ();
    a.elmat = elmats;
    a.bdr_dofs = bdr_dofs;
    return a;
}

}  // namespace saamge
#endif  // SAAMGE_AMD_WITH_MFEM
#endif  // SAAMGE_AMD_HPP
