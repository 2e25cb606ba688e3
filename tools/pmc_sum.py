#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: python tools/pmc_sum.py counter_collection.csv [kernel substring]"""
import collections, csv, sys
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("saamge_amd::", "").replace("void ", "")
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k in per:
    if sub in k:
        print(k, {c: "%.4g" % v for c, v in per[k].items()})
