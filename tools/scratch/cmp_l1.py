import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from saamge_amd import capi, problems as pr
g = np.load("tests/golden/base_elasticity_q2_32.npz")
n = int(g["dims"][0]); cblk = [tuple(int(v) for v in row) for row in g["coarse_blk"]]; nco = len(g["thetas"])
prob = pr.elasticity3d_q2_device(n, blk=(4, 4, 4), coarse_blk=cblk, device="cuda:0")
out = {}
for G in (2, 8):
    capi.set_options(eig_outer_panels=G)
    params = capi.default_params(num_coarsenings=nco, theta=float(g["thetas"][0]), nu_relax=3)
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions, prob.nparts, params, prob.NE_, 81)
    infos = [h.level_info(l) for l in range(nco)]
    res = []
    for l in range(nco):
        m = np.zeros(infos[l]["nparts"], dtype=np.int32)
        capi._check(capi.load().saamge_amd_get_ae_eigens(h.h, C.c_int(l), capi._ptr(m), None, None, None))
        _, k, _, _ = h.get_mis(l)
        res.append((m.copy(), np.asarray(k).copy()))
    out[G] = (infos, res)
    print("G", G, "dims", [i["n"] for i in infos] + [infos[-1]["ncoarse"]])
    h.close()
for l in range(nco):
    m2, k2 = out[2][1][l]; m8, k8 = out[8][1][l]
    print("level", l, "m equal", np.array_equal(m2, m8), "sum m", m2.sum(), m8.sum(), "k equal", np.array_equal(k2, k8), "sum k", k2.sum(), k8.sum())
    if not np.array_equal(m2, m8):
        d = np.nonzero(m2 != m8)[0]; print("  AE diffs", d[:20], m2[d][:20], m8[d][:20])
    if not np.array_equal(k2, k8):
        d = np.nonzero(k2 != k8)[0]; print("  MIS diffs", len(d), d[:20], k2[d][:20], k8[d][:20])
