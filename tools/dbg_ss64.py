import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saamge_amd import capi, problems
n = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 64, 64)
prob = problems.poisson3d_device(n, blk=(8,8,4), coarse_blk=None, device="cuda:0")
params = capi.default_params(num_coarsenings=1)
for rep in range(2):
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions, prob.nparts, params, prob.NE_, 8)
    print(h.level_info(0))
    h.close()
