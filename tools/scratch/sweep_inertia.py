import os, sys
import numpy as np, scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "tests")
from saamge_amd import capi
from test_gpu_inertia import _banded_spsd, _snd_D
rng = np.random.default_rng(1)
bad = 0
for trial in range(12):
    mats, diags, shapes = [], [], []
    for _ in range(6):
        n = int(rng.integers(150, 1100))
        bw = int(rng.integers(113, n - 1)) if trial % 2 else int(rng.integers(max(113, n - 140), n - 1))
        A = _banded_spsd(n, bw, rng, shift=1e-3)
        A += np.diag(np.where(np.diag(A) <= 0, 1.0, 0.0))
        mats.append(A); diags.append(_snd_D(A)); shapes.append((n, bw))
    for G in (2, 8):
        capi.set_options(eig_outer_panels=G)
        for vu in (0.003, 0.05):
            neg = capi.inertia_batched(mats, diags, vu)
            for A, D, k, sh in zip(mats, diags, neg, shapes):
                w = sla.eigh(A, np.diag(D), eigvals_only=True)
                ref = int(np.sum(w < vu)); gap = np.min(np.abs(w - vu))
                if k != ref and not (k == -1 and gap < 1e-6):
                    bad += 1
                    print("MISMATCH G", G, "shape", sh, "vu", vu, "got", k, "ref", ref, "gap", gap, flush=True)
    print("trial", trial, shapes, "bad so far", bad, flush=True)
