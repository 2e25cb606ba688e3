#!/usr/bin/env python3
"""Config 4 (K = diag(1,1,1000), 128^3, bench shape): where the library's coarse dofs per MIS differ from the oracle's
golden (tests/golden/base_aniso128.npz), with the singular values on both sides.
    python tools/aniso_cut_report.py [subspace|dense] [golden name]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from saamge_amd import capi, problems as pr  # noqa: E402

eig = sys.argv[1] if len(sys.argv) > 1 else "subspace"
name = sys.argv[2] if len(sys.argv) > 2 else "base_aniso128"
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
n = int(g["dims"][0])
thetas = [float(v) for v in g["thetas"]]
cblk = [tuple(int(v) for v in row) for row in g["coarse_blk"]]
params = capi.default_params(num_coarsenings=len(thetas), theta=thetas[0], nu_relax=3, eigensolver=eig, keep_debug=True)
for l in range(1, capi.MAX_LEVELS):
    params.theta[l] = thetas[min(l, len(thetas) - 1)]
prob = pr.poisson3d_device(n, blk=(8, 8, 4), coarse_blk=cblk, K=tuple(float(v) for v in g["K"]), device="cuda:0")
h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr,
                   prob.partitions, prob.nparts, params, prob.NE_, 8)
infos = [h.level_info(l) for l in range(len(thetas))]
print("eigensolver %s: dims %s, oracle %s" % (eig, [i["n"] for i in infos] + [infos[-1]["ncoarse"]], g["level_dims"].tolist()))
l = 0
_, k, nc, _ = h.get_mis(l)
gk = g["l0_mis_k"].astype(np.int32)
kept, dropped = g["l0_sv_min_kept"], g["l0_sv_max_dropped"]
off, sig, U = h.get_mis_svd(l)
diff = np.nonzero(k != gk)[0]
print("level 0: %d MISes, k differs on %d; sum k here %d, oracle %d" % (k.size, diff.size, k.sum(), gk.sum()))
# classes of differing MISes by (k here, k oracle)
import collections
cls = collections.Counter((int(k[m]), int(gk[m])) for m in diff)
print("  (k here, k oracle) -> count:", dict(cls))
for m in diff[:12]:
    s = sig[off[m]:off[m + 1]]
    r = s / s[0] if s.size else s
    print("  MIS %6d: columns %d, k here %d oracle %d | ratios here %s | oracle: smallest kept %.3e largest dropped %.3e"
          % (m, nc[m], k[m], gk[m], np.array2string(r, precision=3, max_line_width=200), kept[m], dropped[m]))
# the other way round: the oracle's borderline family, and what the library finds there
for lo, hi, what in ((0.0, 1e-9, "oracle kept below 1e-9"),):
    fam = np.nonzero(kept < hi)[0]
    agree = int((k[fam] == gk[fam]).sum())
    print("  %s: %d MISes, library agrees on %d" % (what, fam.size, agree))
fam = np.nonzero(dropped > 1e-11)[0]
print("  oracle dropped above 1e-11: %d MISes, library agrees on %d" % (fam.size, int((k[fam] == gk[fam]).sum())))
for m in fam[:4]:
    s = sig[off[m]:off[m + 1]]
    print("     MIS %d: ratios here %s, oracle dropped %.3e" % (m, np.array2string(s / s[0], precision=3), dropped[m]))
h.close()
