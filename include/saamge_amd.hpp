// saamge_amd.hpp -- C++ host-side mirror of the SAAMGE solver/operator API for the hot path,
// implemented as thin adaptors over the C ABI (saamge_amd.h).
//
// Two layers:
//   1. namespace saamge_amd::api (this file) -- always available, raw arrays, same names / argument
//      order / error behaviour (negative iteration count on failure, inc/tg.hpp:291-293) as the
//      reference's free functions: MultilevelParameters, ml_produce_data, ml_free_data,
//      VCycleSolver::Mult, smpr_sym_poly, kalchev_pcg, adapt_update_operators.
//   2. namespace saamge (saamge_amd_mfem.hpp, included from here when SAAMGE_AMD_WITH_MFEM is
//      defined, i.e. where <mfem.hpp> and hypre exist -- not in the build container): the
//      reference's own types and entry points -- agg_partitioning_relations_t,
//      ElementMatrixProvider, MultilevelParameters, tg_data_t, ml_data_t, ml_produce_data,
//      tg_produce_data, VCycleSolver, SpectralAMGSolver, kalchev_pcg -- so that an amg/test driver
//      compiles and links against this library for the setup + solve path.
#ifndef SAAMGE_AMD_HPP
#define SAAMGE_AMD_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "saamge_amd.h"

namespace saamge_amd {
namespace api {

// == MultilevelParameters (inc/ml.hpp:59-114): the reference's 11-argument constructor
// (inc/ml.hpp:66-70, src/ml.cpp:54-91), same argument order and getters.
class MultilevelParameters {
public:
    MultilevelParameters(int coarsenings, int *nparts_arr, int first_nu_pro, int nu_pro, int nu_relax,
                         double first_theta, double theta, int polynomial_coarse_space,
                         bool use_correct_nullspace, bool use_arpack, bool do_aggregates)
        : nparts_(nparts_arr, nparts_arr + coarsenings), polynomial_coarse_space_(coarsenings, polynomial_coarse_space),
          use_arpack_(use_arpack), use_double_cycle_(false), coarse_direct_(false) {
        if (coarsenings < 1 || coarsenings >= SAAMGE_AMD_MAX_LEVELS)
            throw std::invalid_argument("MultilevelParameters: 1 <= coarsenings < SAAMGE_AMD_MAX_LEVELS");
        saamge_amd_params_default(&p);
        p.num_coarsenings = coarsenings;
        for (int i = 0; i < coarsenings; ++i) {
            p.theta[i] = i ? theta : first_theta;
            p.nu_pro[i] = i ? nu_pro : first_nu_pro;
            p.nu_relax[i] = nu_relax;
        }
        p.correct_nullspace = use_correct_nullspace ? 1 : 0;   // CorrectNullspace on scaling_P (src/ml.cpp:225-236)
        p.do_aggregates = do_aggregates ? 1 : 0;               // src/ml.cpp:149
        p.avoid_ess_bdr_dofs = 1;                              // src/ml.cpp:64
        // use_arpack: the reference switches to ARPACK above ARPACK_SIZE_THRESHOLD (inc/interp.hpp:104);
        // this library always solves the local problems with its own batched eigensolver (same pairs).
        // polynomial_coarse_space >= 0 asks for ExtendWithPolynomials / ExtendWithRBMs: the modes need the
        // dof coordinates, so they are passed already evaluated through set_extra_coarse_modes(); a
        // non-negative order without modes is refused at ml_produce_data.
    }
    int get_num_coarsenings() const { return p.num_coarsenings; }
    int get_nu_pro(int j) const { return p.nu_pro[j]; }
    int get_nu_relax(int j) const { return p.nu_relax[j]; }
    double get_theta(int j) const { return p.theta[j]; }
    bool get_smooth_interp(int j) const { return p.nu_pro[j] > 0; }
    int get_polynomial_coarse_space(int j) const { return polynomial_coarse_space_[j]; }
    bool get_use_correct_nullspace() const { return p.correct_nullspace != 0; }
    bool get_use_arpack() const { return use_arpack_; }
    bool get_do_aggregates() const { return p.do_aggregates != 0; }
    int get_nparts(int j) const { return nparts_[j]; }
    bool get_avoid_ess_bdr_dofs() const { return p.avoid_ess_bdr_dofs != 0; }
    bool get_use_double_cycle() const { return use_double_cycle_; }
    double get_smooth_drop_tol() const { return p.smooth_drop_tol; }
    void set_polynomial_coarse_space(int j, int val) { polynomial_coarse_space_[j] = val; }
    void set_use_double_cycle(bool use) {
        if (use) throw std::invalid_argument("MultilevelParameters: the double cycle is outside the hot path (SURVEY section 2)");
        use_double_cycle_ = use;
    }
    bool get_coarse_direct() const { return coarse_direct_; }
    void set_coarse_direct(bool cd) { coarse_direct_ = cd; p.coarse_solver = cd ? 1 : 0; }
    void set_smooth_drop_tol(double tol) { p.smooth_drop_tol = tol; }
    // ---- additions of this library ----
    // polynomial / rigid-body coarse-space extension: modes evaluated by the caller, n x count column-major
    void set_extra_coarse_modes(const double *modes, int count) { p.extra_modes = modes; p.num_extra_modes = count; }
    // element-free mode (tg_produce_data_algebraic): pass NE = n, nde = 1 and NULL element arrays
    void set_algebraic(bool on, bool use_window = false) { p.algebraic = on ? (use_window ? 2 : 1) : 0; }
    void set_do_aggregates(bool on) { p.do_aggregates = on ? 1 : 0; }
    void set_eigensolver(int which) { p.eigensolver = which; }      // 0 few-eigenpairs (certified), 1 dense

    saamge_amd_params p;
    const int *nparts_data() const { return nparts_.data(); }

private:
    std::vector<int> nparts_, polynomial_coarse_space_;
    bool use_arpack_, use_double_cycle_, coarse_direct_;
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
inline ml_data_t *ml_produce_data(const ProblemArrays &a, const MultilevelParameters &mlp, void *stream = nullptr) {
    for (int j = 0; j < mlp.get_num_coarsenings(); ++j)
        if (mlp.get_polynomial_coarse_space(j) >= 0 && mlp.p.num_extra_modes == 0)
            throw std::invalid_argument("polynomial_coarse_space >= 0: pass the evaluated modes with set_extra_coarse_modes()");
    if ((int)a.partitions.size() < mlp.get_num_coarsenings())
        throw std::invalid_argument("ml_produce_data: one partition array per coarsening is required");
    ml_data_t *h = nullptr;
    if (saamge_amd_ml_produce_data(a.n, a.rowptr, a.col, a.val, a.NE, a.nde, a.elem_to_dof, a.elmat, a.bdr_dofs,
                                   a.partitions.data(), mlp.nparts_data(), &mlp.p, stream, &h))
        throw std::runtime_error(saamge_amd_last_error());
    return h;
}
// ml_free_data (inc/ml.hpp:196)
inline void ml_free_data(ml_data_t *h) { saamge_amd_ml_free_data(h); }
// adapt_update_operators (inc/adapt.hpp, src/adapt.cpp:188-219): new matrix values, same pattern
inline void adapt_update_operators(ml_data_t *h, const double *new_values) {
    if (saamge_amd_update_operators(h, new_values)) throw std::runtime_error(saamge_amd_last_error());
}

// VCycleSolver (inc/solve.hpp:129-143, src/solve.cpp:309-323): Mult zeroes x unless iterative_mode.
class VCycleSolver {
    ml_data_t *h_;
    bool iterative_mode_;
public:
    explicit VCycleSolver(ml_data_t *h, bool iterative_mode = false) : h_(h), iterative_mode_(iterative_mode) {}
    void Mult(const double *b, double *x) const {
        if (saamge_amd_vcycle(h_, b, x, iterative_mode_ ? 1 : 0)) throw std::runtime_error(saamge_amd_last_error());
    }
};

// smpr_ft-shaped call (inc/smpr.hpp:59-60): x += M^-1 (b - A x) on `level`
inline void smpr_sym_poly(ml_data_t *h, int level, const double *b, double *x) {
    if (saamge_amd_smoother(h, level, b, x)) throw std::runtime_error(saamge_amd_last_error());
}

// kalchev_pcg (inc/mfem_addons.hpp:276, src/mfem_addons.cpp:106-248) with B = the hierarchy's V-cycle:
// iterates from the caller's x; stops when (B r, r) < max(RTOLERANCE (B r0, r0), ATOLERANCE);
// returns the iteration count, its negative when the loop did not converge, -1 when the start vector
// already satisfies the criterion (:150-162).  zero_rhs (the reference's A-norm stopping rule for b = 0,
// :142-148, :205) is not implemented: refused.
inline int kalchev_pcg(ml_data_t *h, const double *b, double *x, int print_iter = 0, int max_num_iter = 1000,
                       double RTOLERANCE = 10e-12, double ATOLERANCE = 10e-24, bool zero_rhs = false) {
    if (zero_rhs) throw std::invalid_argument("kalchev_pcg: zero_rhs (A-norm stopping rule) is not supported");
    (void)print_iter;
    int iters = 0, conv = 0;
    if (saamge_amd_pcg(h, b, x, RTOLERANCE, ATOLERANCE, max_num_iter, /*squared_tol=*/0, /*zero_guess=*/0, &iters, &conv,
                       nullptr))
        throw std::runtime_error(saamge_amd_last_error());
    if (conv && iters == 0) return -1;
    return conv ? iters : -iters;
}

}  // namespace api
}  // namespace saamge_amd

#ifdef SAAMGE_AMD_WITH_MFEM
#include "saamge_amd_mfem.hpp"
#endif
#endif  // SAAMGE_AMD_HPP
