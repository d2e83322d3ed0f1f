#!/usr/bin/env python3
"""tools/ab_traffic_summary.py <workload> — table of tools/ab_traffic.sh's passes: Msamples/s (by kernel time: mean of the frame means / best frame; by the wall time of
runRenderer, delivery included: mean) and, for the LAST full frame,
the bytes that leave the L2 (2 x FETCH_SIZE + WRITE_SIZE, KB -> MB): of the two render dispatches alone (what profiles/traffic.json has always held) and of
EVERY kernel between the previous frame's last render dispatch and this frame's (ordering pass, fills, copy kernels if the runtime uses any)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
W = sys.argv[1] if len(sys.argv) > 1 else "C2"
w = bench.WORKLOADS[W]
alg = w["nx"] * w["ny"] * 12 / 1e6
d = os.path.join(ROOT, "gpurun_out", f"ab_traffic_{W}")


def frame(path):
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ren = [int(r["Dispatch_Id"]) for r in rows if "k_render" in r["Kernel_Name"]]
    if len(ren) < 3:
        return None, None, {}
    lo, hi = ren[-3], ren[-1]
    kern = sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) in ren[-2:])
    allk = sum(float(r["Counter_Value"]) for r in rows if lo < int(r["Dispatch_Id"]) <= hi)
    by = {}
    for r in rows:
        if lo < int(r["Dispatch_Id"]) <= hi:
            k = r["Kernel_Name"].split("(")[0][-60:]
            by[k] = by.get(k, 0.0) + float(r["Counter_Value"])
    return kern, allk, by


print(f"{W}: {w['name']}; framebuffer {alg:.1f} MB.  MB that leave the L2 in the last full frame = 2 x FETCH_SIZE + WRITE_SIZE")
print("%-14s %9s %9s %9s | render dispatches: %7s %7s %7s %6s | every kernel of the frame: %7s %7s %7s %6s" %
      ("form", "mean", "best", "wall", "read", "written", "sum", "x fb", "read", "written", "sum", "x fb"))
detail = []
for t in sorted(glob.glob(os.path.join(d, "time_*.txt")), key=os.path.getmtime):
    name = os.path.basename(t)[5:-4]
    ts = [tuple(map(float, ln.split())) for ln in open(t) if ln.strip()]
    mean = sum(x[0] for x in ts) / len(ts); best = max(x[1] for x in ts); wall = sum(x[2] for x in ts) / len(ts)
    rk, ra, rby = frame(os.path.join(d, name, "rd"))
    wk, wa, wby = frame(os.path.join(d, name, "wr"))
    if rk is None or wk is None:
        print("%-14s %9.1f %9.1f %9.1f | (no counters)" % (name, mean, best, wall)); continue
    rk, ra = 2 * rk / 1024, 2 * ra / 1024; wk, wa = wk / 1024, wa / 1024
    f = 1.048576        # KB (1024) -> 1e6
    print("%-14s %9.1f %9.1f %9.1f | %25.1f %7.1f %7.1f %6.1f | %32.1f %7.1f %7.1f %6.1f" %
          (name, mean, best, wall, rk * f, wk * f, (rk + wk) * f, (rk + wk) * f / alg, ra * f, wa * f, (ra + wa) * f, (ra + wa) * f / alg))
    detail.append((name, rby, wby))
print("\nper kernel (MB read x2 / written), every dispatch of the frame:")
for name, rby, wby in detail:
    print(" ", name)
    for k in sorted(set(rby) | set(wby)):
        print("      %-62s %8.1f %8.1f" % (k, 2 * rby.get(k, 0) / 1024 * 1.048576, wby.get(k, 0) / 1024 * 1.048576))
