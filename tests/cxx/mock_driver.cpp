// TEST: the call sequence of an amg/test driver (amg/test/mltest/mltest.cpp:667-793, test/algebraic/algebraic.cpp:282-283)
// against include/saamge_amd.hpp with SAAMGE_AMD_WITH_MFEM, compiled (not linked) against tests/mfem_stub.
#include "saamge_amd.hpp"

using namespace mfem;
using namespace saamge;

// a driver's own smoother with the smpr_ft signature (inc/smpr.hpp:59-60)
static void my_jacobi(HypreParMatrix &A, const Vector &b, Vector &x, void *data) {
    (void)A; (void)b; (void)x; (void)data;
}

int mock_driver(HypreParMatrix *Ag, SparseMatrix *Al, ParBilinearForm *a, Table *elem_to_dof, Table *elem_to_elem,
                int *partitioning, const agg_dof_status_t *bdr_dofs, HypreParVector *bg, HypreParVector *pxg,
                Solver *my_coarse_solver, int num_levels, int elems_per_agg) {
    int *nparts_arr = new int[num_levels - 1];
    nparts_arr[0] = elem_to_dof->Size() / elems_per_agg;
    for (int i = 1; i < num_levels - 1; ++i) nparts_arr[i] = nparts_arr[i - 1] / elems_per_agg;
    agg_partitioning_relations_t *agg_part_rels = agg_create_partitioning_fine(
        *Ag, elem_to_dof->Size(), elem_to_dof, elem_to_elem, partitioning, bdr_dofs, nparts_arr, NULL, false);
    ElementMatrixProvider *emp = new ElementMatrixStandardGeometric(*agg_part_rels, Al, a);
    const int first_nu_pro = 0, nu_pro = 0, nu_relax = 3, polynomial_coarse = -1;
    const double first_theta = 0.003, theta = 0.003;
    const bool correct_nulspace = true, direct_eigensolver = true, do_aggregates = false;
    MultilevelParameters mlp(num_levels - 1, nparts_arr, first_nu_pro, nu_pro, nu_relax, first_theta, theta, polynomial_coarse,
                             correct_nulspace, !direct_eigensolver, do_aggregates);      // the reference's 11 arguments
    mlp.set_coarse_direct(true);
    mlp.set_smooth_drop_tol(0.0);
    ml_data_t *ml_data = ml_produce_data(*Ag, agg_part_rels, emp, mlp);
    levels_level_t *level = levels_list_get_level(ml_data->levels_list, 0);
    tg_data_t *tg = level->tg_data;
    const int nc = tg->Ac->Height() + tg->interp->Width() + tg->restr->Height();       // fields read directly by callers
    tg->coarse_solver = my_coarse_solver;                                              // test/algebraic/algebraic.cpp:282-283
    tg->tag = 1;
    Solver *Bprec = new VCycleSolver(tg, false);
    Bprec->SetOperator(*Ag);
    Bprec->Mult(*bg, *pxg);
    VCycleSolver it(tg, true);                                                         // iterative_mode
    it.SetOperator(*Ag);
    it.Mult(*bg, *pxg);
    const int iters = kalchev_pcg(*Ag, *Bprec, *bg, *pxg, 0, 1000, 1e-12, 1e-24, false);
    tg->pre_smoother(*Ag, *bg, *pxg, tg->poly_data);                                   // smpr_ft plug
    tg->pre_smoother = my_jacobi;                                                      // a caller's smoother (inc/tg.hpp:99-119): honoured by the next Mult
    tg->post_smoother = my_jacobi;
    Bprec->Mult(*bg, *pxg);
    Array<int> dims;
    ml_get_dims(*ml_data, dims);
    agg_fetch_tables(*agg_part_rels, *ml_data);
    delete Bprec;
    ml_free_data(ml_data);
    agg_free_partitioning(agg_part_rels);
    delete[] nparts_arr;
    return iters + nc + dims.Size();
}

// the element-free driver: amg/test/algebraic/algebraic.cpp:257-283 (the dof -> AE map is the caller's here: the
// reference's fem_create_partitioning_from_matrix calls METIS)
int mock_algebraic_driver(HypreParMatrix *Ag, SparseMatrix *Al, int *dof_partitioning, int nparts, Solver *my_coarse_solver,
                          HypreParVector *bg, HypreParVector *pxg, bool window_amg) {
    const int n = Ag->Height();
    Table *dof_to_dof = new Table;                 // every dof its own "element"
    dof_to_dof->MakeI(n);
    for (int i = 0; i < n; ++i) dof_to_dof->AddAColumnInRow(i);
    dof_to_dof->MakeJ();
    for (int i = 0; i < n; ++i) dof_to_dof->AddConnection(i, i);
    dof_to_dof->ShiftUpI();
    agg_partitioning_relations_t *agg_part_rels =
        agg_create_partitioning_fine(*Ag, n, dof_to_dof, NULL, dof_partitioning, NULL, &nparts, NULL, false);
    const int first_nu_pro = 0, nu_pro = 0, nu_relax = 3, polynomial_coarse = -1;
    const double first_theta = 0.003;
    const bool use_arpack = true, avoid_ess_bdr_dofs = true;
    tg_data_t *tg_data = tg_produce_data_algebraic(*Al, *Ag, *agg_part_rels, first_nu_pro, nu_relax, first_theta, (nu_pro > 0),
                                                   polynomial_coarse, window_amg, use_arpack, avoid_ess_bdr_dofs);
    tg_fillin_coarse_operator(*Ag, tg_data, false);
    tg_data->coarse_solver = my_coarse_solver;
    VCycleSolver prec(tg_data, false);
    prec.SetOperator(*Ag);
    const int iters = kalchev_pcg(*Ag, prec, *bg, *pxg, 0, 1000, 1e-12, 1e-24, false);
    tg_free_coarse_operator(*tg_data);
    tg_fillin_coarse_operator(*Ag, tg_data, false);
    const int nc = tg_data->Ac->Height();
    tg_free_data(tg_data);
    agg_free_partitioning(agg_part_rels);
    return iters + nc;
}

// test/encapsulate/encapsulate.cpp:282-284: the solver object built from the bilinear form alone, the reference's 11 arguments
int mock_encapsulate_driver(HypreParMatrix *Ag, ParBilinearForm *aform, Array<int> &ess_bdr, HypreParVector *bg, HypreParVector *pxg,
                            int elems_per_agg, int num_levels, int nu_pro, int nu_relax, double theta) {
    int polynomial_coarse = -1;
    const bool coarse_direct = false;
    SpectralAMGSolver spectral_pc(*Ag, *aform, aform->SpMat(), ess_bdr, elems_per_agg, num_levels, nu_pro, nu_relax, theta,
                                  polynomial_coarse, coarse_direct);
    spectral_pc.Mult(*bg, *pxg);
    return spectral_pc.Height();
}

// the split two-level setup, inc/tg.hpp:428-432 + 478-481 (tg_produce_data's two halves, src/tg.cpp:542-578)
int mock_tg_split_driver(HypreParMatrix *Ag, SparseMatrix *Al, ParBilinearForm *a, Table *elem_to_dof, Table *elem_to_elem,
                         int *partitioning, const agg_dof_status_t *bdr_dofs, int nparts, HypreParVector *bg, HypreParVector *pxg) {
    agg_partitioning_relations_t *agg_part_rels =
        agg_create_partitioning_fine(*Ag, elem_to_dof->Size(), elem_to_dof, elem_to_elem, partitioning, bdr_dofs, &nparts, NULL, false);
    ElementMatrixProvider *emp = new ElementMatrixStandardGeometric(*agg_part_rels, Al, a);
    tg_data_t *tg_data = tg_init_data(*Ag, *agg_part_rels, 0, 3, 0.003, false, 0.0, false);
    tg_data->tag = 7;
    tg_build_hierarchy(*Ag, *tg_data, *agg_part_rels, emp, true);
    VCycleSolver prec(tg_data, false);
    prec.SetOperator(*Ag);
    prec.Mult(*bg, *pxg);
    const int nc = tg_data->Ac->Height() + tg_data->tag;
    tg_free_data(tg_data);
    agg_free_partitioning(agg_part_rels);
    return nc;
}
