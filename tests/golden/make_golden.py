#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (numpy + LAPACK dsygvx/dgesvd).

The reference itself cannot be built or run here (MFEM/hypre/METIS absent), so these
vectors are outputs of the restatement, pinned by the reference's ctest iteration counts
(see tests/test_oracle_kat.py).  Only sign/rotation-invariant quantities are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import saamge_oracle as o  # noqa: E402
from saamge_amd import problems as pr  # noqa: E402


def invariants(H, b):
    out = {}
    for l, lv in enumerate(H.levels):
        rel = lv.rel
        out["l%d_mises" % l] = rel.mises.astype(np.int32)
        out["l%d_mis_to_dof_I" % l] = rel.mis_to_dof.I.astype(np.int32)
        out["l%d_mis_to_dof_J" % l] = rel.mis_to_dof.J.astype(np.int32)
        out["l%d_mis_to_AE_J" % l] = rel.mis_to_AE.J.astype(np.int32)
        out["l%d_AE_sizes" % l] = np.diff(rel.AE_to_dof.I).astype(np.int32)
        out["l%d_ae_m" % l] = np.array([z.shape[1] for z in lv.evects], dtype=np.int32)
        out["l%d_evals" % l] = np.concatenate(lv.evals)
        out["l%d_mis_k" % l] = lv.mis_numcoarsedof.astype(np.int32)
        out["l%d_svals" % l] = np.concatenate([s for s in lv.mis_svals if s is not None] or [np.zeros(0)])
        Ac = lv.Ac
        out["l%d_Ac_dim" % l] = np.array([Ac.shape[0]], dtype=np.int32)
        out["l%d_Ac_trace" % l] = np.array([Ac.diagonal().sum()])
        out["l%d_Ac_fro" % l] = np.array([np.sqrt((Ac.multiply(Ac)).sum())])
    x = o.vcycle(H, b)
    out["vcycle_x_norm"] = np.array([np.linalg.norm(x)])
    out["vcycle_Ax_dot_b"] = np.array([float((H.levels[0].A @ x) @ b)])
    if len(H.levels) == 1:
        out["vcycle_x"] = x      # two-level cycle with an exact coarse solve is basis independent
    xs, it, conv, hist = o.solve(H, b, rel_tol=1e-6)
    out["pcg_iters"] = np.array([it], dtype=np.int32)
    out["pcg_hist"] = np.array(hist)
    out["pcg_x"] = xs
    return out


def main():
    cases = {
        "mltest_q1_2level": (pr.mltest_problem(order=1, levels=2), 1, True),
        "mltest_q1_3level": (pr.mltest_problem(order=1, levels=3), 2, True),
        "mltest_q2_2level": (pr.mltest_problem(order=2, levels=2), 1, True),
        "poisson3d_8_2level": (pr.poisson3d_problem((8, 8, 8), blk=(4, 4, 2)), 1, False),
        "poisson3d_aniso_2level": (pr.poisson3d_problem((12, 8, 4), blk=(4, 4, 2), K=(1, 1, 1000.0)), 1, False),
    }
    for name, (prob, nco, testmesh) in cases.items():
        theta = 0.02 if "aniso" in name else 0.003
        H = o.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions[:nco],
                              theta=theta, nu_relax=3, testmesh=testmesh)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **invariants(H, prob.b))
        print(name, "levels", [lv.A.shape[0] for lv in H.levels], "->", H.levels[-1].Ac.shape[0])


if __name__ == "__main__":
    main()
