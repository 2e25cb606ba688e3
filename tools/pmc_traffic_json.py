#!/usr/bin/env python3
"""profiles/r03_pmc_traffic.json from the counter passes of tools/pmc_lab.sh over tools/spmv_lab (the headline's fine-level
operator shape: 257^3 rows, 27-point stencil).  Keys are bench.py's profiler labels "<name>@<rows>".
    python tools/pmc_traffic_json.py gpurun_out/<tag>/pmc_lab profiles/r03_pmc_traffic.json"""
import collections
import csv
import glob
import json
import os
import sys

root, out = sys.argv[1], sys.argv[2]
LABEL = {"sell_staged_kernel<3>": "smooth_step", "sell_staged_kernel<0>": "spmv", "sell_staged_kernel<1>": "spmv_residual"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("saamge_amd::", "").replace("void ", "")
        if k in LABEL and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = 257 ** 3
            acc["%s@%d" % (LABEL[k], rows)][r["Counter_Name"]].append(float(r["Counter_Value"]) * 1024.0)
res = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/spmv_lab 257: the library's kernels on an "
                  "operator of the headline's fine-level shape and size; FETCH_SIZE raw (bench.py doubles it: MI355X_MICROARCH.md, gfx950)"}
for key, c in acc.items():
    n = min(len(c["FETCH_SIZE"]), len(c["WRITE_SIZE"]))
    res[key] = {"fetch_bytes_raw": sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * n, "write_bytes": sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * n,
                "launches": n}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
