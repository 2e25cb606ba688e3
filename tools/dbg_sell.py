import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SAAMGE_AMD_SPMV_SELL"] = "1"
import numpy as np
from saamge_amd import capi, problems as pr
A = pr.poisson3d_problem((8, 8, 8), blk=(4, 4, 2)).A.tocsr()
n = A.shape[0]
x = np.random.default_rng(0).standard_normal(n)
y = capi.spmv(A, x)
ref = A @ x
bad = np.nonzero(np.abs(y - ref) > 1e-12)[0]
print("bad rows", bad[:20], len(bad), "of", n)
for r in bad[:5]:
    print(r, y[r], ref[r], A.indptr[r+1]-A.indptr[r])
