"""BASELINE config 5's shape (Q2 elasticity, three levels): what the oracle's level-1 local eigenvalues -- and with them
the coarsest space and the iteration count -- depend on that the reference does NOT determine, shown with the ORACLE ALONE.

Every elasticity agglomerate has a six-fold zero eigenvalue (the rigid-body modes); dsygvx returns one D-orthonormal
basis of that eigenspace among all of them.  The columns are normalised one by one before the MIS SVD
(src/xpacks.cpp:537-559), so the singular vectors depend on that basis although their span does not, and the
weighted-l1 diagonal D_ii = sum_j |a_ij| sqrt(a_ii / a_jj) of the next level's agglomerate matrices
(src/mbox.cpp:913-949) is not invariant under the resulting change of basis of the coarse space.

The test rotates the oracle's own eigenvectors inside each group of equal eigenvalues by a random orthogonal matrix
(oracle.EVECTS_HOOK) and requires: the same level dimensions, the level-1 OPERATOR's spectrum unchanged to 1e-10 (the
coarse space is the same space) -- and level-1 local eigenvalues that move by several per cent.  On the full-size
variant of this problem (8 x 8 x 4 elements, 2 187-dof agglomerates; a minute per run, not part of the suite) the
same rotations move the smallest level-1 eigenvalue 5.26e-4 -> 5.05e-4 / 5.35e-4 / 4.73e-4 and the PCG iteration count
16 -> 17: the library's 5.22e-4 / 16 iterations against the oracle's 5.27e-4 / 17 (round 2) is one member of that
family, not a defect (tests/test_gpu_parity.py::test_q2_elasticity3d_three_level_matches_oracle pins what IS invariant)."""
import numpy as np

from oracle import saamge_oracle as o
from saamge_amd import problems as pr


def _run(prob, hook):
    o.EVECTS_HOOK = hook
    try:
        H = o.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions[:2], theta=0.003, nu_relax=3)
    finally:
        o.EVECTS_HOOK = None
    x, it, conv, hist = o.solve(H, prob.b, rel_tol=1e-8)
    dims = [lv.A.shape[0] for lv in H.levels] + [H.levels[-1].Ac.shape[0]]
    return dims, it, [w.copy() for w in H.levels[1].evals], np.linalg.eigvalsh(H.levels[1].A.toarray()), \
        [w.copy() for w in H.levels[0].evals]


def test_rotating_degenerate_eigenvectors_moves_the_next_levels_eigenvalues():
    prob = pr.elasticity3d_q2_problem((4, 4, 2), blk=(2, 2, 2))
    assert int(prob.partitions[0].max()) + 1 == 4
    prob.partitions = [prob.partitions[0], np.array([0, 0, 1, 1], dtype=np.int32)]
    dims0, it0, ev0, spec0, fine0 = _run(prob, None)
    assert any(np.sum(np.abs(w) < 1e-12) == 6 for w in fine0)          # six rigid-body modes at zero in the free agglomerates
    moved = []
    for seed in range(3):
        rng = np.random.default_rng(seed)

        def hook(i, w, Z):
            Z = Z.copy()
            a = 0
            while a < len(w):
                b = a + 1
                while b < len(w) and abs(w[b] - w[b - 1]) < 1e-9:
                    b += 1
                if b - a > 1:
                    Q, _ = np.linalg.qr(rng.standard_normal((b - a, b - a)))
                    Z[:, a:b] = Z[:, a:b] @ Q
                a = b
            return Z

        dims1, it1, ev1, spec1, _ = _run(prob, hook)
        assert dims1 == dims0
        assert np.allclose(spec1, spec0, rtol=0, atol=1e-10 * spec0.max())      # the same coarse SPACE
        moved.append(max(abs(a[0] - b[0]) / b[0] for a, b in zip(ev1, ev0)))
        assert abs(it1 - it0) <= 1
    print("smallest level-1 eigenvalue moved by %s (relative) under rotations inside the degenerate eigenspaces" % moved)
    assert max(moved) > 0.03
