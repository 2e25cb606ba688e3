"""Q2 elasticity (BASELINE config 5) through the host generator at a moderate size: does the whole path hold?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from saamge_amd import capi, problems as pr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time()
prob = pr.elasticity3d_q2_problem(n, blk=(4, 4, 4))
if levels > 2:
    p1, nb = pr.block_partition((n // 4,) * 3, (2, 2, 2))
    prob.partitions.append(p1)
print("problem: %d dofs, %d nnz, %.1f s" % (prob.A.shape[0], prob.A.nnz, time.time() - t0), flush=True)
params = capi.default_params(num_coarsenings=levels - 1, theta=0.003, nu_relax=3)
for rep in range(2):
    t0 = time.time()
    h = capi.Hierarchy.from_problem(prob, params)
    t1 = time.time()
    x, it, conv, hist = h.pcg(prob.b, rel_tol=1e-8, max_iter=300)
    t2 = time.time()
    infos = [h.level_info(l) for l in range(levels - 1)]
    print("setup %.2f s solve %.2f s its %d conv %s dims %s vec/AE %s relres %.2e" % (
        t1 - t0, t2 - t1, it, conv, [i["n"] for i in infos] + [infos[-1]["ncoarse"]],
        [round(i["nvec"] / i["nparts"], 2) for i in infos],
        np.linalg.norm(prob.A @ x - prob.b) / np.linalg.norm(prob.b)), flush=True)
    h.close()
