#!/usr/bin/env python3
"""Times the solve phase alone (PCG on an existing hierarchy): python tools/time_solve.py [n] [levels]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from saamge_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prob = bench.build_problem(n, levels, "cuda:0")
params = capi.default_params()
h, x, it, conv, hist = bench.one_step(capi, prob, params)
for rep in range(3):
    x = torch.zeros_like(prob.b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, it, conv, hist = h.pcg(prob.b, x, rel_tol=1e-8, max_iter=200)
    torch.cuda.synchronize()
    print("solve %.1f ms, %d iterations, coarse graph %s" % ((time.perf_counter() - t0) * 1e3, it,
          os.environ.get("SAAMGE_AMD_COARSE_GRAPH", "1")), flush=True)
