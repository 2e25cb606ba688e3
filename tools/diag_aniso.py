"""Diagnostic: accuracy of the batched AE eigensolver on the anisotropic 16^3 case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from saamge_amd import problems as pr, capi
from oracle import saamge_oracle as o

prob = pr.poisson3d_problem((16, 16, 16), blk=(8, 8, 4), K=(1, 1.3, 1000.0001))
params = capi.default_params(num_coarsenings=1, theta=0.003, nu_relax=3, keep_debug=True)
h = capi.Hierarchy.from_problem(prob, params)
H = o.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions[:1], theta=0.003, nu_relax=3)
lv = H.levels[0]
m, ev, X, Ds = h.get_ae_eigens(0)
for i in [0, 4, 5]:
    A = lv.AE_mats[i] if hasattr(lv, "AE_mats") else None
    Xo, Do = lv.evects[i], lv.Ds[i]
    print("AE", i, "m", m[i], Xo.shape[1], "max|ev - ev_or|", np.abs(ev[i] - lv.evals[i][:len(ev[i])]).max())
    G = X[i].T @ (Ds[i][:, None] * X[i])
    print("   D-orth err", np.abs(G - np.eye(m[i])).max())
    # subspace distance
    Q1 = np.linalg.qr(np.sqrt(Ds[i])[:, None] * X[i])[0]
    Q2 = np.linalg.qr(np.sqrt(Do)[:, None] * Xo)[0]
    print("   subspace dist", np.linalg.norm(Q1 - Q2 @ (Q2.T @ Q1), 2))
off, sig, U = h.get_mis_svd(0)
mises, k, ncols, flags = h.get_mis(0)
for mis in [22, 40]:
    print("mis", mis, "k", k[mis], "ncols", ncols[mis], "sig", sig[off[mis]:off[mis] + 4], "oracle", lv.mis_svals[mis][:4])
