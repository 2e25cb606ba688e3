#!/usr/bin/env python3
"""Per-(kernel, grid size) summary of a directory of rocprofv3 passes (tools/pmc_smoother.sh):
average launch duration from the kernel traces and per-launch averages of every collected counter.
    python tools/pmc_summary.py <dir> [kernel substring]
FETCH_SIZE is printed raw (KB) and as bytes with the gfx950 correction of MI355X_MICROARCH.md (x2 for wide
coalesced reads); WRITE_SIZE in KB -> bytes."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""


def short(k):
    return k.split("(")[0].replace("saamge_amd::", "").replace("void ", "")


dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        dur[(f, key)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
best = {}
for (f, key), v in dur.items():          # the pass without counters (stats/) has the undisturbed durations
    if "stats" in f or key not in best:
        best[key] = (sum(v) / len(v), len(v))
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0))
        ctr[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(best, key=lambda k: -best[k][0] * best[k][1]):
    if sub not in key[0]:
        continue
    us, n = best[key]
    line = "%-44s grid %10d  %6d launches  %9.1f us" % (key[0][:44], key[1], n, us)
    c = {k: sum(v) / len(v) for k, v in ctr.get(key, {}).items()}
    extra = []
    if "FETCH_SIZE" in c:
        extra.append("fetch raw %.1f MB (x2 = %.1f MB)" % (c["FETCH_SIZE"] / 1024, 2 * c["FETCH_SIZE"] / 1024))
    if "WRITE_SIZE" in c:
        extra.append("write %.1f MB" % (c["WRITE_SIZE"] / 1024))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        extra.append("HBM-side %.2f TB/s" % ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / (us * 1e-6) / 1e12))
    if "TCC_HIT_sum" in c:
        extra.append("L2 hit %.3f" % (c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)))
    for k in sorted(c):
        if k not in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
            extra.append("%s %.4g" % (k, c[k]))
    print(line + ("  " + "; ".join(extra) if extra else ""))
