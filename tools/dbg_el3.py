import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from saamge_amd import capi, problems as pr
from oracle import saamge_oracle as o
prob = pr.elasticity3d_q2_problem((8, 8, 4), blk=(4, 4, 4))
prob.partitions = [prob.partitions[0], np.array([0, 0, 1, 1], dtype=np.int32)]
params = capi.default_params(num_coarsenings=2, theta=0.003, nu_relax=3, keep_debug=True, coarse_rtol=1e-28)
h = capi.Hierarchy.from_problem(prob, params)
H = o.ml_produce_data(prob.A, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions[:2], theta=0.003, nu_relax=3)
A1g = h.get_csr(1, "A").toarray(); A1o = H.levels[1].A.toarray()
print("A1 spectrum diff", np.max(np.abs(np.linalg.eigvalsh(A1g) - np.linalg.eigvalsh(A1o))))
m, ev, X, Ds = h.get_ae_eigens(1)
for i in range(2):
    print("L1 AE", i, "n", len(Ds[i]), "m", m[i], "gpu evals", ev[i], "oracle evals", H.levels[1].evals[i][:m[i] + 2])
    print("   D diff", np.max(np.abs(np.sort(Ds[i]) - np.sort(H.levels[1].Ds[i]))), "D min/max", Ds[i].min(), Ds[i].max())
A2g = h.get_csr(1, "Ac").toarray(); A2o = H.levels[1].Ac.toarray()
print("A2 spectrum gpu", np.linalg.eigvalsh(A2g)); print("A2 spectrum ora", np.linalg.eigvalsh(A2o))
# level-1 element matrices: compare the AE matrices through their spectra
I, J = h.get_table(1, "AE_to_dof")
print("L1 AE_to_dof sizes", np.diff(I), "oracle", [len(H.levels[1].rel.AE_to_dof.row(i)) for i in range(2)])
