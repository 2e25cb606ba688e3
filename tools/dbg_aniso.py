"""Compare the per-AE eigenvector counts and eigenvalues of the two local eigensolvers on the anisotropic workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from saamge_amd import capi, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
prob = problems.poisson3d_device(n, blk=(8, 8, 4), coarse_blk=None, K=(1.0, 1.0, 1000.0), device="cuda:0")
res = {}
for es in ("dense", "subspace"):
    params = capi.default_params(num_coarsenings=1, theta=theta, eigensolver=es, keep_debug=1)
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions,
                       prob.nparts, params, prob.NE_, 8)
    m, ev, X, Ds = h.get_ae_eigens(0)
    _, k, nc, _ = h.get_mis(0)
    res[es] = (np.array(m), ev, h.level_info(0), np.array(k))
    print(es, h.level_info(0)["ncoarse"], int(np.sum(m)))
    h.close()
md, ed, _, kd = res["dense"]
ms, es_, _, ks = res["subspace"]
bad = np.nonzero(md != ms)[0]
print("AEs with different counts:", len(bad), bad[:20])
for a in bad[:10]:
    print(a, "dense", ed[a], "subspace", es_[a])
print("MIS k differences:", int(np.sum(kd != ks)), "sum dense", int(kd.sum()), "sum subspace", int(ks.sum()))
same = md == ms
if same.all():
    print("max eigenvalue difference", max(float(np.max(np.abs(a - b))) if len(a) else 0.0 for a, b in zip(ed, es_)))
# singular values of the MISes whose kept count differs
sv = {}
for es in ("dense", "subspace"):
    params = capi.default_params(num_coarsenings=1, theta=theta, eigensolver=es, keep_debug=1)
    h = capi.Hierarchy(prob.rowptr, prob.col, prob.val, prob.n, prob.elem_to_dof, prob.elmat, prob.bdr, prob.partitions,
                       prob.nparts, params, prob.NE_, 8)
    off, sig, U = h.get_mis_svd(0)
    sv[es] = (off.copy(), sig.copy())
    h.close()
diff = np.nonzero(kd != ks)[0]
for mth in diff[:12]:
    o0, o1 = sv["dense"][0][mth], sv["dense"][0][mth + 1]
    a = sv["dense"][1][o0:o1]; b = sv["subspace"][1][o0:o1]
    print("MIS", mth, "k dense", kd[mth], "k subspace", ks[mth])
    print("   dense    sig/sig0", np.array2string(a / a[0], precision=3))
    print("   subspace sig/sig0", np.array2string(b / b[0], precision=3))
allr = []
for es in ("dense", "subspace"):
    off, sig = sv[es]
    r = []
    for mth in range(len(off) - 1):
        s_ = sig[off[mth]:off[mth + 1]]
        if len(s_) > 1 and s_[0] > 0:
            r.extend((s_[1:] / s_[0]).tolist())
    r = np.array(r)
    print(es, "singular value ratios in [1e-13, 1e-7]:", np.sort(r[(r > 1e-13) & (r < 1e-7)])[:40], "count", int(((r > 1e-13) & (r < 1e-7)).sum()))
