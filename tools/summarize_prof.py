"""Condenses the rocprofv3 CSVs of tools/profile.sh into one text summary (kernel stats + per-dispatch
counter averages of the render kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, "**", pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f, r in rows("*kernel_stats.csv"):
    print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})

print("== per-dispatch durations from the kernel trace (render kernels only) ==")
dur = defaultdict(list)
meta = {}
for f, r in rows("*kernel_trace.csv"):
    name = r.get("Kernel_Name", "")
    if "k_render" in name:
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                             "Workgroup_Size", "Grid_Size")}
for name, d in dur.items():
    print(name[:80], "calls", len(d), "avg ms %.3f min %.3f max %.3f" % (sum(d) / len(d), min(d), max(d)), meta[name])

print("== PMC counters: mean per dispatch of the render kernel ==")
acc = defaultdict(list)
for f, r in rows("*counter_collection.csv"):
    if "k_render" in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
vals = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(vals):
    print("%-28s %.6g   (n=%d)" % (k, vals[k], len(acc[k])))
g = vals.get
if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
    print("VALU insts per wave            %.4g" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    print("VALU lane utilisation          %.3f  (SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)... see DESIGN.md)" %
          (g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))))
if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
    print("VALU busy (ACTIVE_INST_VALU*4/BUSY_CYCLES/SIMDs-per-SE..) raw ratio %.4g" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES")))
if g("FETCH_SIZE") is not None:
    print("FETCH_SIZE KB %.6g  -> x2 gfx950 correction = %.6g KB" % (g("FETCH_SIZE"), 2 * g("FETCH_SIZE")))
if g("WRITE_SIZE") is not None:
    print("WRITE_SIZE KB %.6g" % g("WRITE_SIZE"))
