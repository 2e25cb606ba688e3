"""Angle between the eigenspaces returned by the dense and the few-eigenpairs path, per AE (anisotropic workload)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from saamge_amd import capi, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
K = (1.0, 1.0, float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0)
prob = problems.poisson3d_device(n, blk=(8, 8, 4), coarse_blk=None, K=K, device="cuda:0")
out = {}
for es in ("dense", "subspace"):
    params = capi.default_params(num_coarsenings=1, theta=theta, eigensolver=es, keep_debug=1)
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions,
                       prob.nparts, params, prob.NE_, 8)
    out[es] = h.get_ae_eigens(0)
    h.close()
md, ed, Xd, Dd = out["dense"]
ms, es_, Xs, Ds = out["subspace"]
worst = []
for a in range(len(md)):
    if md[a] != ms[a]:
        print("count differs at", a, md[a], ms[a]); continue
    D = Dd[a]
    G = Xd[a].T @ (D[:, None] * Xs[a])
    R = Xs[a] - Xd[a] @ G
    ang = np.sqrt(np.max(np.sum(R * (D[:, None] * R), axis=0)))
    orth_d = np.max(np.abs(Xd[a].T @ (D[:, None] * Xd[a]) - np.eye(md[a])))
    orth_s = np.max(np.abs(Xs[a].T @ (D[:, None] * Xs[a]) - np.eye(ms[a])))
    worst.append((ang, a, md[a], orth_d, orth_s))
worst.sort(reverse=True)
for w in worst[:8]:
    a = w[1]
    print("AE %d m %d: subspace angle %.2e, D-orthonormality dense %.2e subspace %.2e, evals %s" % (a, w[2], w[0], w[3], w[4], np.array2string(ed[a], precision=9)))
print("median angle %.2e" % np.median([w[0] for w in worst]))
