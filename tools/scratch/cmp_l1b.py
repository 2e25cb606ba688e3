import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from saamge_amd import capi, problems as pr
g = np.load("tests/golden/base_elasticity_q2_32.npz")
n = int(g["dims"][0]); cblk = [tuple(int(v) for v in row) for row in g["coarse_blk"]]; nco = len(g["thetas"])
prob = pr.elasticity3d_q2_device(n, blk=(4, 4, 4), coarse_blk=cblk, device="cuda:0")
capi.set_options(debug=1)
for G in (2, 8):
    capi.set_options(eig_outer_panels=G)
    params = capi.default_params(num_coarsenings=nco, theta=float(g["thetas"][0]), nu_relax=3, keep_debug=True)
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions, prob.nparts, params, prob.NE_, 81)
    m, ev, X, Ds = h.get_ae_eigens(1)
    I, _ = h.get_table(1, "AE_to_dof")
    print("G", G, "theta", float(g["thetas"][0]), "sizes", np.diff(I)[[19, 20, 21, 24, 36, 40]])
    for i in (19, 20, 24, 36, 40):
        print("  AE", i, "n", I[i + 1] - I[i], "m", m[i], "evals", ev[i])
    h.close()
