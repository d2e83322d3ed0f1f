"""Condenses the rocprofv3 CSVs of tools/profile.sh into one text summary (kernel stats + counter totals of the
render kernel PER FRAME: the reference-stream mode renders a frame in two dispatches of the same kernel, so the
counters of a frame's dispatches are added)."""
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

for name, d in dur.items():
    per_frame = 2 if any("k_order_by_cost" in r.get("Kernel_Name", "") for f, r in rows("*kernel_trace.csv")) else 1
    frames = max(len(d) // per_frame, 1)
    print("render kernel time per frame: %.3f ms (%d dispatches per frame, %d frames)" % (sum(d) / frames, per_frame, frames))

print("== PMC counters: total per FRAME of the render kernel (dispatches of one frame added) ==")
acc = defaultdict(list)
two_phase = False
for f, r in rows("*counter_collection.csv"):
    two_phase = two_phase or "k_order_by_cost" in r.get("Kernel_Name", "")
    if "k_render" in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
per_frame = 2 if two_phase else 1
vals = {k: sum(v) / max(len(v) // per_frame, 1) for k, v in acc.items()}
for k in sorted(vals):
    print("%-28s %.6g   (%d dispatches, %d per frame)" % (k, vals[k], len(acc[k]), per_frame))
g = vals.get
if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
    print("VALU insts per wave            %.4g" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    print("VALU lane utilisation          %.3f  (SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)... see DESIGN.md)" %
          (g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))))
if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
    cyc = g("GRBM_GUI_ACTIVE") / 8.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    print("kernel cycles per frame (GRBM_GUI_ACTIVE / 8 XCDs) %.4g ; VALUBusy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles) = %.3f" %
          (cyc, g("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * cyc)))
if g("FETCH_SIZE") is not None:
    print("FETCH_SIZE KB %.6g  -> x2 gfx950 correction = %.6g KB" % (g("FETCH_SIZE"), 2 * g("FETCH_SIZE")))
if g("WRITE_SIZE") is not None:
    print("WRITE_SIZE KB %.6g" % g("WRITE_SIZE"))
