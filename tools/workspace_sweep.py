#!/usr/bin/env python3
"""The headline workload with several values of saamge_amd_params.workspace_bytes (the chunking of the agglomerates):
setup / solve wall time of warm steps and the device-memory high-water mark.   python tools/workspace_sweep.py 32 64 128"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from saamge_amd import capi

dev = torch.device("cuda:0")
prob = bench.build_problem((256, 256, 256), 3, dev)
for gib in [int(v) for v in sys.argv[1:]] or [32, 64]:
    params = capi.default_params(num_coarsenings=2, theta=0.003, nu_relax=3, workspace_bytes=gib << 30)
    out = []
    for step in range(5):
        if step == 2:
            capi.memory_stats(reset_peak=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h, x, it, conv, hist = bench.one_step(capi, prob, params)
        torch.cuda.synchronize()
        out.append((1e3 * (time.perf_counter() - t0),) + bench.one_step.last_split[:2])
        h.close()
    live, peak = capi.memory_stats()
    print("workspace %4d GiB: steps %s  its %d  peak %.1f GB" % (gib, " ".join("%.0f=%.0f+%.0f" % o for o in out[2:]), it, peak / 1e9), flush=True)
