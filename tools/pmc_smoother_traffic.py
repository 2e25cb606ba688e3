#!/usr/bin/env python3
"""Summarises the two counter passes of tools/pmc_smoother_traffic.sh into <dir>/pmc_traffic.json:
{"smooth_step@<rows>": {"fetch_bytes_raw", "write_bytes", "launches", "format": {...}}} -- bench.py's `roofline.traffic`
(FETCH_SIZE is raw: bench.py doubles it, MI355X_MICROARCH.md's gfx950 correction; both counters are in KiB)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
fmt = {}
rows0 = None
for line in open(os.path.join(d, "FETCH_SIZE.log")):
    if line.startswith("FORMAT 0 "):
        _, _, rows0, js = line.split(" ", 3)
        fmt = json.loads(js)
        rows0 = int(rows0)


def is_smoother(name):      # the fused smoother step of the staged kernels (MODE 3)
    name = name.replace("(int)", "")
    return "sell_staged_kernel<3>" in name or "sell_staged2_kernel<3>" in name


def total(counter):
    path = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, n = 0.0, 0
    grids = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not is_smoother(r["Kernel_Name"]):
            continue
        g = int(r["Grid_Size"])
        grids[g] = grids.get(g, 0) + 1
    big = max(grids) if grids else 0          # the fine level's launches have the largest grid
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(int)", "")
        g = int(r["Grid_Size"])
        if is_smoother(name) and g == big:
            tot += float(r["Counter_Value"]) * 1024.0
            n += 1
    return tot, n


f, nf = total("FETCH_SIZE")
w, nw = total("WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/smoother_bench.py (the library's own "
                  "smoother applications on the bench's hierarchy); sell_staged2_kernel<3> launches of the fine level; FETCH_SIZE raw "
                  "(bench.py doubles it: MI355X_MICROARCH.md, gfx950); tools/pmc_smoother_traffic.sh",
       "smooth_step@%d" % rows0: {"fetch_bytes_raw": f, "write_bytes": w, "launches": nf, "format": fmt}}
json.dump(out, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
print("smooth_step@%d: %d launches, %.4f GB per launch (2 x FETCH + WRITE), format %s" % (rows0, nf, (2 * f + w) / nf / 1e9, fmt))
