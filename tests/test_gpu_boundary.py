"""The two plugs of the drop-in boundary that round 1 lacked, through the C ABI on the GPU:
VCycleSolver's iterative_mode (src/solve.cpp:309-323) and an assignable tg_data_t::coarse_solver
(inc/tg_data.hpp:71, used by test/algebraic/algebraic.cpp:282-283)."""
import numpy as np
import pytest

from saamge_amd import problems as pr

pytestmark = pytest.mark.gpu


def _setup(levels):
    from saamge_amd import capi
    from oracle import saamge_oracle as o
    prob = pr.poisson3d_problem((12, 8, 8), blk=(4, 4, 4), coarse_blk=[(2, 2, 2)] * (levels - 2), coef="skew")
    params = capi.default_params(num_coarsenings=levels - 1, theta=0.003, nu_relax=3)
    h = capi.Hierarchy.from_problem(prob, params)
    H = o.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions[:levels - 1], theta=0.003, nu_relax=3)
    return capi, o, prob, h, H


@pytest.mark.parametrize("levels", [2, 3])
def test_vcycle_iterative_mode_matches_reference_cycle_from_x(levels):
    capi, o, prob, h, H = _setup(levels)
    rng = np.random.default_rng(1)
    x0 = rng.standard_normal(prob.A.shape[0])
    x0[prob.ess] = 0.0
    ref = o.vcycle_iterative(H, prob.b, x0)
    x = h.vcycle_iterative(prob.b, x0.copy())
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    # and it is the stationary iteration x + B (b - A x)
    alt = x0 + h.vcycle(prob.b - prob.A @ x0)
    assert np.linalg.norm(x - alt) <= 1e-12 * np.linalg.norm(alt)
    h.close()


def test_assigned_coarse_solver_is_used():
    capi, o, prob, h, H = _setup(2)
    Ac = h.get_csr(0, "Ac").toarray()
    calls = []

    def mine(rc):
        calls.append(rc.size)
        return np.linalg.solve(Ac, rc)

    x_builtin = h.vcycle(prob.b)
    h.set_coarse_solver(mine)
    x_plug = h.vcycle(prob.b)
    assert calls == [Ac.shape[0]]
    assert np.linalg.norm(x_plug - x_builtin) <= 1e-10 * np.linalg.norm(x_builtin)
    # a different coarse solver changes the cycle exactly as in the reference's tg_cycle_atb
    h.set_coarse_solver(lambda rc: 0.5 * np.linalg.solve(Ac, rc))
    x_half = h.vcycle(prob.b)
    # (the oracle's coarse basis may differ from the GPU's by signs: each side halves ITS exact coarse solve)
    ref = o.vcycle_iterative(H, prob.b, np.zeros_like(prob.b), coarse=lambda rc: 0.5 * np.linalg.solve(H.coarse_dense, rc))
    assert np.linalg.norm(x_half - ref) <= 1e-9 * np.linalg.norm(ref)
    h.set_coarse_solver(None)
    assert np.linalg.norm(h.vcycle(prob.b) - x_builtin) <= 1e-14 * np.linalg.norm(x_builtin)
    h.close()


@pytest.mark.parametrize("levels", [2, 3])
def test_assigned_smoothers_change_the_cycle_as_in_the_reference(levels):
    """tg_data_t::pre_smoother / post_smoother (inc/smpr.hpp:59-60, src/tg.cpp:113,131,411-414) through
    saamge_amd_set_smoother: a caller's damped Jacobi in the place of the polynomial smoother -- before the coarse
    correction only, after it only, on both sides of level 0, and (three levels) on level 1 -- gives the cycle the
    oracle gives with the same smoother in the same place."""
    capi, o, prob, h, H = _setup(levels)
    calls = []

    def jacobi_for(level):
        A = h.get_csr(level, "A").tocsr()
        d = A.diagonal()
        def gpu_side(lev, b, x):
            calls.append((lev, "zero start" if not np.any(x) else "from x"))
            return x + 0.6 * (b - A @ x) / d
        return gpu_side

    def jacobi_oracle(A, b, x):
        x += 0.6 * (b - A @ x) / A.diagonal()

    x_builtin = h.vcycle(prob.b)
    cases = [({0: ("pre",)}, "pre only"), ({0: ("post",)}, "post only"), ({0: ("pre", "post")}, "both")]
    if levels == 3:
        cases.append(({1: ("pre", "post")}, "level 1"))
        cases.append(({0: ("pre", "post"), 1: ("pre", "post")}, "levels 0 and 1"))
    for where, tag in cases:
        del calls[:]
        plugs = {}
        for lev in range(levels - 1):
            pre = jacobi_for(lev) if "pre" in where.get(lev, ()) else None
            post = jacobi_for(lev) if "post" in where.get(lev, ()) else None
            h.set_smoother(lev, pre, post)
            plugs[lev] = (jacobi_oracle if pre else None, jacobi_oracle if post else None)
        x = h.vcycle(prob.b)
        ref = o.vcycle(H, prob.b, 0, plugs)
        assert len(calls) == sum(len(v) for v in where.values()), (tag, calls)
        assert all(c[1] == "zero start" for c in calls[:1] if "pre" in where.get(0, ())), (tag, calls)
        assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref), tag
        assert np.linalg.norm(x - x_builtin) > 1e-7 * np.linalg.norm(x_builtin), tag      # (the plug is not ignored)
    for lev in range(levels - 1):
        h.set_smoother(lev, None, None)
    assert np.linalg.norm(h.vcycle(prob.b) - x_builtin) <= 1e-14 * np.linalg.norm(x_builtin)
    h.close()
