#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> out.json [n levels]

(n, levels: the bench configuration the passes were taken on, default 256 3 = bench.py's default; bench.py only uses
the file for that configuration.)

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
reads exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane); other widths
are uncalibrated, so the factor for OUR 8-byte-per-lane streams is calibrated on kernels with
a known byte count (smooth_first: reads 16 B/row, writes 8 B/row; dot_partial: reads 16 B/row)
and applied to the reads of every kernel.  WRITE_SIZE is taken as is.
"""
import collections
import csv
import json
import sys


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        short = name.split("(")[0].replace("saamge_amd::", "").replace("void ", "")
        if "<" in name.split("(")[0]:
            short = name.split("(")[0].replace("saamge_amd::", "").replace("void ", "")
        per[short][0] += 1
        per[short][1] += float(r["Counter_Value"]) * 1024.0
        per[short][2][int(r["Grid_Size"])] += 1
    return per


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if "at::" in k or "rocprim" in k or "elementwise" in k or "__amd" in k:
            continue
        f = fetch.get(k, [0, 0.0])
        w = write.get(k, [0, 0.0])
        out[k] = {"launches": f[0] or w[0], "fetch_bytes_raw": f[1], "write_bytes": w[1]}
    out["_config"] = {"n": int(sys.argv[4]) if len(sys.argv) > 4 else 256, "levels": int(sys.argv[5]) if len(sys.argv) > 5 else 3}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) of `python3 bench.py "
                    "--no-cpu-baseline --no-roofline --warmup 0 --steps 1`; bytes summed over the launches of one step; "
                    "FETCH_SIZE is raw (gfx950 counts half of a streaming read: bench.py doubles it, MI355X_MICROARCH.md "
                    "HBM section); tools/pmc_traffic.py, tools/collect_profiles.sh")
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    del out["_config"], out["_note"]
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_bytes_raw"] + kv[1]["write_bytes"]))[:16]:
        print("%-44s %6d launches  fetch(raw) %10.1f MB  write %10.1f MB" %
              (k[:44], v["launches"], v["fetch_bytes_raw"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
