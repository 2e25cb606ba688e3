"""Debug helper: in a rocprofv3 --kernel-trace --hip-trace (csv) output directory, find HIP API calls longer than
5 ms inside the timed steps and print what ran on the GPU / which API calls were in flight during them."""
import csv, sys, glob, os
d = sys.argv[1]
api = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
ker = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
A = list(csv.DictReader(open(api)))
K = list(csv.DictReader(open(ker)))
for r in A:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
for r in K:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
A.sort(key=lambda r: r["s"])
t_end = max(r["e"] for r in K)
long = [r for r in A if r["e"] - r["s"] > 5e6 and r["s"] > t_end - 1.7e9]
def busy_of(r):
    return sum(min(k["e"], r["e"]) - max(k["s"], r["s"]) for k in K if k["e"] > r["s"] and k["s"] < r["e"])
idle = [r for r in long if busy_of(r) < 0.5 * (r["e"] - r["s"])]
print("long API calls in the last 1.7 s:", len(long), " of which mostly GPU-idle:", len(idle))
sel = idle[:10]
for r in sel:
    i = A.index(r)
    print("-- preceding API calls on any thread:")
    for a in A[max(0, i - 12):i]:
        print("      %s tid %s %.3f ms (starts %.3f ms before)" % (a["Function"], a["Thread_Id"], (a["e"] - a["s"]) / 1e6, (r["s"] - a["s"]) / 1e6))
for r in sel:
    print("== %s tid %s  %.2f ms  (at %.1f ms before the end)" % (r["Function"], r["Thread_Id"], (r["e"] - r["s"]) / 1e6, (t_end - r["s"]) / 1e6))
    ks = [k for k in K if k["e"] > r["s"] and k["s"] < r["e"]]
    busy = sum(min(k["e"], r["e"]) - max(k["s"], r["s"]) for k in ks)
    print("   kernels overlapping: %d, busy %.2f ms" % (len(ks), busy / 1e6))
    names = {}
    for k in ks:
        n = k["Kernel_Name"][:70]
        names[n] = names.get(n, 0) + (min(k["e"], r["e"]) - max(k["s"], r["s"])) / 1e6
    for n, t in sorted(names.items(), key=lambda x: -x[1])[:6]:
        print("      %8.2f ms %s" % (t, n))
    oth = [a for a in A if a is not r and a["e"] > r["s"] and a["s"] < r["e"] and a["e"] - a["s"] > 1e6]
    for a in oth[:8]:
        print("   concurrent API: %s tid %s %.2f ms" % (a["Function"], a["Thread_Id"], (a["e"] - a["s"]) / 1e6))
