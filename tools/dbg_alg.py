import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from saamge_amd import capi, problems as pr
A = pr.poisson3d_problem((8, 8, 8), blk=(4, 4, 2)).A.tocsr()
part = (np.arange(A.shape[0]) // 60).astype(np.int32)
n = A.shape[0]
b = np.ones(n)
params = capi.default_params(num_coarsenings=1, theta=0.01, keep_debug=True, coarse_rtol=1e-28, algebraic=True)
h = capi.Hierarchy.from_matrix(A, part, params)
print(h.level_info(0))
x = h.vcycle(b)
print("vcycle norm", np.linalg.norm(x), "smoother:", np.linalg.norm(h.smoother(0, b, np.zeros(n))))
xs, it, conv, hist = h.pcg(b, rel_tol=1e-6)
print("pcg", it, conv, hist, np.linalg.norm(A @ xs - b))
