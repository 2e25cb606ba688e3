// TEST: links libsaamge_amd.so through the C++ mirror saamge_amd::api (no GPU needed for the argument checks).
#include <cstdio>
#include <cstring>

#include "saamge_amd.hpp"

using namespace saamge_amd::api;

int main() {
    int nparts[2] = {4, 1};
    MultilevelParameters mlp(2, nparts, 0, 0, 3, 0.003, 0.003, -1, false, true, false);
    if (mlp.get_num_coarsenings() != 2 || mlp.get_nparts(0) != 4 || mlp.get_nu_relax(1) != 3 || mlp.get_theta(1) != 0.003 ||
        mlp.get_use_correct_nullspace() || !mlp.get_use_arpack() || mlp.get_do_aggregates() || !mlp.get_avoid_ess_bdr_dofs())
        return 1;
    // the library answers: NULL arguments are refused with a message, nothing touches the GPU
    saamge_amd_hierarchy *h = nullptr;
    saamge_amd_params p;
    saamge_amd_params_default(&p);
    if (saamge_amd_ml_produce_data(0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &p, nullptr, &h) == 0)
        return 2;
    if (!std::strstr(saamge_amd_last_error(), "null argument")) return 3;
    // refused loudly, like the header says
    bool threw = false;
    try { (void)kalchev_pcg(nullptr, nullptr, nullptr, 0, 10, 1e-12, 1e-24, /*zero_rhs=*/true); } catch (const std::invalid_argument &) { threw = true; }
    if (!threw) return 4;
    threw = false;
    try {
        ProblemArrays a;
        int zero = 0; double one = 1.0; int e2d = 0; int part = 0;
        a.n = 1; a.rowptr = &zero; a.col = &zero; a.val = &one; a.NE = 1; a.nde = 1; a.elem_to_dof = &e2d; a.elmat = &one;
        a.partitions.push_back(&part);      // one partition array for two coarsenings
        (void)ml_produce_data(a, mlp);
    } catch (const std::invalid_argument &) { threw = true; }
    if (!threw) return 5;
    std::printf("api link test ok\n");
    return 0;
}
