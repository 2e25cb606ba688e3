"""Robustness run: 3-D elasticity (3 dofs per vertex), 2 and 3 levels, with and without RBM extra modes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from saamge_amd import capi, problems as pr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t0 = time.time()
prob = pr.elasticity3d_problem((n, n, n), blk=(4, 4, 4), coarse_blk=[(2, 2, 2)])
print("problem %.1f s, dofs %d" % (time.time() - t0, prob.ND))
for nco in (1, 2):
    params = capi.default_params(num_coarsenings=nco, theta=0.003)
    t0 = time.time()
    h = capi.Hierarchy.from_problem(prob, params)
    t1 = time.time()
    x, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8)
    t2 = time.time()
    infos = [h.level_info(l) for l in range(nco)]
    print("levels %d: setup %.3f s solve %.3f s its %d conv %s dims %s nvec/AE %s relres %.2e" % (
        nco + 1, t1 - t0, t2 - t1, it, conv, [i["n"] for i in infos] + [infos[-1]["ncoarse"]],
        [round(i["nvec"] / i["nparts"], 2) for i in infos],
        np.linalg.norm(prob.A @ x - prob.b) / np.linalg.norm(prob.b)))
    h.close()
