"""oracle/cpu_ref.cpp (the threaded C++ restatement behind bench.py's cpu_baseline) against the
Python oracle: identical topology, coarse-space dimensions and iteration counts, V-cycle and PCG
history to round-off.  Both are test infrastructure; this keeps the timed baseline honest."""
import numpy as np
import pytest

from oracle import cpu_ref, saamge_oracle as o
from saamge_amd import problems as pr

CASES = {
    "p8_2level": (lambda: pr.poisson3d_problem((8, 8, 8), blk=(4, 4, 2)), 1),
    "p16_3level": (lambda: pr.poisson3d_problem((16, 16, 16), blk=(4, 4, 4), coarse_blk=[(2, 2, 2)]), 2),
    "skew_3level": (lambda: pr.poisson3d_problem((16, 12, 8), blk=(4, 4, 2), coarse_blk=[(2, 3, 2)], coef="skew"), 2),
    "aniso_2level": (lambda: pr.poisson3d_problem((12, 8, 4), blk=(4, 4, 2), K=(1, 1, 1000.0)), 1),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_cpu_ref_matches_python_oracle(name):
    make, nco = CASES[name]
    prob = make()
    h = cpu_ref.Hierarchy(prob, num_coarsenings=nco, theta=0.003, nu_relax=3, threads=4)
    H = o.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions[:nco], theta=0.003, nu_relax=3)
    assert h.level_dims() == [lv.A.shape[0] for lv in H.levels] + [H.levels[-1].Ac.shape[0]]
    for l in range(nco):
        lv = H.levels[l]
        assert np.array_equal(h.mises(l), lv.rel.mises)                                   # bit-exact topology
        assert np.array_equal(h.ae_m(l), [z.shape[1] for z in lv.evects])                 # eigenvectors per AE
        assert np.array_equal(h.mis_k(l), lv.mis_numcoarsedof)                            # coarse dofs per MIS
        assert np.allclose(h.evals_max(l), [w[-1] for w in lv.evals], rtol=0, atol=1e-11)
    x, xr = h.vcycle(prob.b), o.vcycle(H, prob.b)
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    xs, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8)
    xo, ito, convo, histo = o.solve(H, prob.b, rel_tol=1e-8)
    assert conv and convo and it == ito
    assert np.max(np.abs(hist - np.array(histo)) / np.array(histo)) <= 1e-8
    h.close()


def test_cpu_ref_thread_count_does_not_change_the_answer():
    prob = pr.poisson3d_problem((16, 16, 8), blk=(4, 4, 4), coarse_blk=[(2, 2, 2)])
    res = []
    for t in (1, 5):
        h = cpu_ref.Hierarchy(prob, num_coarsenings=2, threads=t)
        xs, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8)
        res.append((h.level_dims(), it, hist))
        h.close()
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    assert np.array_equal(res[0][2], res[1][2])      # same order of every reduction
