#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals of the LAST `--last` fraction of
the run (the timed step of bench.py after its warm-up), the idle gaps between kernels, and optionally the
per-launch durations of one kernel.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--kernel NAME] [--from-frac 0.5]
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"saamge_amd::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--from-frac", type=float, default=0.0, help="ignore dispatches that start before this fraction of the trace")
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--timeline", action="store_true", help="print the dispatches in time order, runs of one kernel merged")
    ap.add_argument("--min-ms", type=float, default=0.3, help="timeline: fold runs shorter than this into the next line")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + a.from_frac * (t1 - t0)
    rows = [r for r in rows if r[1] >= cut]
    tot = {}
    busy = 0
    gaps = 0
    last_end = rows[0][1]
    for name, s, e in rows:
        k = short(name)
        d = tot.setdefault(k, [0, 0])
        d[0] += e - s
        d[1] += 1
        if s > last_end:
            gaps += s - last_end
        busy += max(0, e - max(s, last_end))
        last_end = max(last_end, e)
    span = rows[-1][2] - rows[0][1]
    print("span %.3f ms, busy %.3f ms, gaps %.3f ms, %d dispatches" % (span / 1e6, busy / 1e6, gaps / 1e6, len(rows)))
    for k, (ns, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print("%-72s %10.3f ms %6d launches %9.1f us avg" % (k, ns / 1e6, n, ns / 1e3 / n))
    if a.timeline:
        base = rows[0][1]
        run = None      # [name, start, end, busy, count]
        pend = [0.0, 0]
        for name, s_, e_ in rows + [("", rows[-1][2], rows[-1][2])]:
            k = short(name)
            if run and k == run[0]:
                run[2] = e_; run[3] += e_ - s_; run[4] += 1
                continue
            if run:
                if (run[2] - run[1]) / 1e6 < a.min_ms:
                    pend[0] += (run[2] - run[1]) / 1e6; pend[1] += run[4]
                else:
                    if pend[1]:
                        print("   ... %d short dispatches, %.3f ms" % (pend[1], pend[0]))
                        pend = [0.0, 0]
                    print("%9.3f ms  %-60s x%-4d span %8.3f ms busy %8.3f ms" % ((run[1] - base) / 1e6, run[0], run[4], (run[2] - run[1]) / 1e6, run[3] / 1e6))
            run = [k, s_, e_, e_ - s_, 1]
    if a.kernel:
        print("launches of", a.kernel)
        for name, s, e in rows:
            if a.kernel in name:
                print("  start %.3f ms  dur %.1f us" % ((s - rows[0][1]) / 1e6, (e - s) / 1e3))


if __name__ == "__main__":
    main()
