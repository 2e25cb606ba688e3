"""The SELL-64 SpMV of the level operators against scipy, in each of its three slice formats: pair-coded
(<= 64 distinct (offset, value) pairs per slice: constant-coefficient stencils), offset-coded (<= 64
distinct column offsets, values streamed: variable-coefficient stencils) and plain.  Bit-level claim:
the coded formats are lossless, so all three give the same sums in the same order."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r"""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, %r)
from saamge_amd import capi, problems as pr
rng = np.random.default_rng(3)
mats = {
  "stencil_const_short_rows": pr.poisson3d_problem((8, 8, 8), blk=(4, 4, 2)).A.tocsr(),     # boundary rows shorter than the slice: padded entries
  "stencil_const": pr.poisson3d_problem((20, 12, 9), blk=(4, 4, 3)).A.tocsr(),
  "stencil_skew": pr.poisson3d_problem((12, 12, 12), blk=(4, 4, 4), coef="skew").A.tocsr(),
  "random": sp.random(1000, 1000, density=0.02, random_state=rng, format="csr"),
  # 30 diagonals spread over +-600 with one value each: pair-coded, ONE x-segment of 256 + 1200 doubles per tile (longer than
  # two passes of the staging loop), segments reaching outside [0, n) at both ends of the operator
  "toeplitz_wide_band": sp.diags([np.full(6000 - abs(o), 1.0 + 0.01 * j) for j, o in enumerate(range(-600, 600, 40))],
                                 list(range(-600, 600, 40)), format="csr"),
  # 9-point stencil on a 300 x 40 grid: three x-segments per tile, tiles that straddle grid lines
  "stencil_2d": (sp.kron(sp.diags([1.0, 2.0, 1.0], [-1, 0, 1], shape=(40, 40)), sp.diags([1.0, -4.0, 1.0], [-1, 0, 1], shape=(300, 300)))).tocsr(),
  "tiny": sp.csr_matrix(np.array([[2.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 2.0]])),
}
out = {}
for name, A in mats.items():
    A = A.tocsr(); A.sort_indices()
    x = rng.standard_normal(A.shape[1])
    y = capi.spmv(A, x)
    ref = A @ x
    err = np.abs(y - ref).max() / (np.abs(A) @ np.abs(x)).max()
    print("RESULT", name, repr(float(err)), y.tobytes().hex()[:64])
"""


def _run(codes, pair_fast=None):
    # saamge_amd_options.sell: bit 0 coded slices at all, 1 pair coding, 2 short-chain path, 3 dictionary, 4 node blocks
    sell = 31
    if pair_fast is not None and int(pair_fast) == 0:
        sell &= ~4
    if codes is not None:
        sell &= {0: ~3, 1: ~2}.get(int(codes), ~0)       # codes = 0: plain slices only; 1: offset codes only
    env = dict(os.environ, SAAMGE_AMD_TEST_OPTIONS="spmv_sell=1,sell=%d" % (sell & 31))
    o = subprocess.run([sys.executable, "-c", _CODE % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert o.returncode == 0, o.stdout + o.stderr
    res = {}
    for line in o.stdout.splitlines():
        if line.startswith("RESULT"):
            _, name, err, digest = line.split()
            res[name] = (float(err), digest)
    return res


def test_sell_formats_match_scipy_and_each_other():
    full = _run(None)          # pair codes where possible (all-pair operators: the fast kernel, sell_pair_kernel)
    slow = _run(None, 0)       # the same through the general kernel
    assert {k: v[1] for k, v in slow.items()} == {k: v[1] for k, v in full.items()}      # identical bits
    offs = _run(1)             # offset codes only
    plain = _run(0)            # no codes
    assert set(full) == set(offs) == set(plain) and len(full) == 7
    for name in full:
        for res in (full, offs, plain):
            assert res[name][0] <= 1e-15, (name, res[name])
        assert full[name][1] == offs[name][1] == plain[name][1], name      # identical bits


_CODE_GPAIR = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from saamge_amd import capi, problems as pr
A = pr.elasticity3d_q2_problem((10, 10, 10), blk=(2, 2, 2)).A.tocsr()      # 243 entries per row, 4.8e6 stored entries, ~4 900 pairs
A.sort_indices()
x = np.random.default_rng(5).standard_normal(A.shape[1])
y = capi.spmv(A, x)
err = np.abs(y - A @ x).max() / (np.abs(A) @ np.abs(x)).max()
ys = [y]
for other in (x[::-1].copy(), np.abs(x)):          # (two more vectors)
    ys.append(capi.spmv(A, other))
# rows outside regular node blocks (every 97th dof keeps its diagonal entry only, as an eliminated row would): they are
# redone by sell_gpair3_fix_kernel
C = A.tocoo()
keep = (C.row %% 97 != 5) | (C.row == C.col)
import scipy.sparse as sp
A2 = sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=A.shape)
A2.sort_indices()
y2 = capi.spmv(A2, x)
err = max(err, np.abs(y2 - A2 @ x).max() / (np.abs(A2) @ np.abs(x)).max())
ys.append(y2)
# the other members of the family (residual, fused smoother step) through a two-level hierarchy on the same operator:
# the (B r, r) history of a few PCG iterations and the solution, bit for bit
prob = pr.elasticity3d_q2_problem((10, 10, 10), blk=(2, 2, 2))
h = capi.Hierarchy.from_problem(prob, capi.default_params(num_coarsenings=1, theta=0.003, nu_relax=3))
xs, it, conv, hist = h.pcg(prob.b, rel_tol=1e-6, max_iter=6)
ys += [np.asarray(hist, dtype=np.float64), np.asarray(xs, dtype=np.float64)]
h.close()
import hashlib
print("RESULT", repr(float(err)), hashlib.sha256(b"".join(v.tobytes() for v in ys)).hexdigest())
"""


def test_operator_level_pair_dictionary_is_lossless():
    """An operator whose slices cannot be coded one by one (Q2 elasticity: 243 entries per row) but whose (offset, value)
    pairs repeat across the whole operator: 16-bit codes into one table (csrc/sparse.hip, sell_gdict_kernel /
    sell_gpair_kernel) give the same bits as the plain SELL slices they replace -- with the lanes of a 3 x 3 node block
    sharing their gathers of x (the clamped face's eliminated rows and the nodes cut by slice boundaries are the
    irregular lanes), without the sharing, and without the dictionary."""
    outs = []
    for gpair, bs3 in (("1", "1"), ("1", "0"), ("0", "0")):
        sell = 31 & ~(0 if gpair == "1" else 8) & ~(0 if bs3 == "1" else 16)
        env = dict(os.environ, SAAMGE_AMD_TEST_OPTIONS="spmv_sell=1,debug=2,sell=%d" % sell)
        o = subprocess.run([sys.executable, "-c", _CODE_GPAIR % ROOT], env=env, capture_output=True, text=True, timeout=900)
        assert o.returncode == 0, o.stdout + o.stderr
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")][0].split()
        import re
        irregular = [int(m) for m in re.findall(r"build_sell: (\d+) of \d+ rows outside regular node blocks", o.stderr)]
        outs.append((float(line[1]), line[2], "pair dictionary" in o.stderr and "abandoned" not in o.stderr,
                     ", 3 x 3 node blocks" in o.stderr, max(irregular) if irregular else -1))
        print([l for l in o.stderr.splitlines() if "dictionary" in l or "rows outside" in l])
    assert outs[0][2] and outs[1][2] and not outs[2][2]          # the dictionary was built in the first two runs only
    assert outs[0][3] and not outs[1][3]                         # node blocks were found, and used in the first run only
    assert outs[0][4] > 100                                      # ... with rows outside them on the second operator (the fix kernel ran)
    assert all(o[0] <= 1e-15 for o in outs)
    assert outs[0][1] == outs[1][1] == outs[2][1]                # identical bits
