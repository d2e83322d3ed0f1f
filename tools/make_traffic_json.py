#!/usr/bin/env python3
"""tools/make_traffic_json.py <tag> — condenses gpurun_out/prof_<tag>_{C2,C4}/ (tools/measure_traffic.sh) into
gpurun_out/<tag>_{C2,C4}_summary.txt and gpurun_out/traffic.json; copy both to profiles/ to have them judged.
Per-frame figures = the dispatches of the LAST full frame of each pass (the render kernel runs twice per frame: the
first samples, then the cost-ordered rest)."""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {"kernel_source_hash": bench.kernel_source_hash(), "tag": tag}


def rows(d, pattern):
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


for W, per_frame in (("C2", 2), ("C4", 2), ("C3", 2), ("C5", 2)):      # (C4: two dispatches per frame since round 4)
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{W}")
    if not os.path.isdir(d):
        continue
    lines = []
    lines.append(f"== {W}: {bench.WORKLOADS[W]['name']} ==")
    lines.append("== kernel stats (rocprofv3 --kernel-trace --stats), all dispatches of the run (1 spp warm-up frame included) ==")
    for r in rows(d, "*kernel_stats.csv"):
        lines.append(str({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}))
    tr = sorted((r for r in rows(d, "*kernel_trace.csv") if "k_render" in r.get("Kernel_Name", "")), key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
    frames = [sum(dur[i:i + per_frame]) for i in range(per_frame, len(dur) - per_frame + 1, per_frame)]     # skip the warm-up frame
    if frames:
        lines.append("render-kernel time per full frame (ms, %d dispatch(es) per frame): %s  avg %.3f" % (per_frame, [round(x, 3) for x in frames], sum(frames) / len(frames)))
        lines.append("code object: %s" % {k: tr[-1].get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")} +
                     "  (rocprofv3's view; tools/kernel_resources.sh prints the code object's own metadata)")
    acc = defaultdict(list)
    for r in sorted(rows(d, "*counter_collection.csv"), key=lambda r: int(r.get("Dispatch_Id", 0))):
        if "k_render" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    vals = {k: sum(v[-per_frame:]) for k, v in acc.items()}                      # the last full frame
    lines.append("== PMC counters of the LAST full frame (its %d dispatch(es) added) ==" % per_frame)
    for k in sorted(vals):
        lines.append("%-28s %.6g" % (k, vals[k]))
    g = vals.get
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        lines.append("VALU lane utilisation          %.3f  (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU))" % (g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))))
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if g(k):
                lines.append("%-20s / SQ_WAVE_CYCLES = %.3f" % (k, g(k) / g("SQ_WAVE_CYCLES")))
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        lines.append("kernel cycles per frame (GRBM_GUI_ACTIVE / 8 XCDs) %.4g ; VALU wave-instructions per SIMD-cycle = %.3f (1024 SIMDs; 0.5 = the 2-cycle issue limit)" %
                     (cyc, g("SQ_INSTS_VALU") / (1024.0 * cyc)))
    pmc = {}
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE") and g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        issue = g("SQ_INSTS_VALU") / (1024.0 * cyc)
        lane = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
        pmc = {"valu_insts_per_simd_cycle": issue, "valu_issue_frac": issue / 0.5, "valu_lane_utilisation": lane,
               "useful_lane_issue_frac": issue / 0.5 * lane,
               "definition": "SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) against the 0.5 per cycle issue limit (one wave64 VALU instruction per 2 "
                             "cycles per SIMD); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); their product = the share of the "
                             "chip's lane-issue slots that did useful work"}
        if g("SQ_WAVE_CYCLES"):
            for k, nm in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any")):
                if g(k):
                    pmc[nm] = g(k) / g("SQ_WAVE_CYCLES")
        if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
            pmc["lds_bank_conflict_ratio"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
            lines.append("SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f" % pmc["lds_bank_conflict_ratio"])
        lines.append("useful lane-issue fraction = (VALU insts per SIMD-cycle / 0.5) x lane utilisation = %.3f x %.3f = %.3f" % (issue / 0.5, lane, issue / 0.5 * lane))
    plain = os.path.join(d, "plain.log")
    if os.path.exists(plain):
        for ln in open(plain):
            if ln.startswith("{"):
                pj = json.loads(ln)
                lines.append("== un-profiled run of the same program: kernel ms per frame %s ==" % [round(x, 3) for x in pj["kernel_ms"]])
                dc = pj.get("device_counters")
                if dc:
                    lines.append("device counters of one frame (the inputs of bench.py's roofline.frac): %s" % json.dumps(dc))
                    if W != "C4":
                        ex = 18.0 * dc["exec_tests"] + 20.0 * dc["box_tests"] + 80.0 * dc["rays"]
                        ms = min(pj["kernel_ms"])
                        lines.append("executed flops = 18 x exec_tests + 20 x box_tests + 80 x rays = %.4g ; / %.3f ms = %.2f TFLOP/s = %.3f of 157.3 (roofline.frac)" %
                                     (ex, ms, ex / ms / 1e9, ex / ms / 1e9 / 157.3))
                        eff = dc["rays"] * (18.0 * 488 + 80.0)
                        lines.append("brute-force-equivalent flops = rays x (18 x 488 + 80) = %.4g ; -> %.2f TFLOP/s = %.3f of 157.3 (roofline.effective_frac)" %
                                     (eff, eff / ms / 1e9, eff / ms / 1e9 / 157.3))
                    else:
                        by = 48.0 * dc["node_visits"] + 64.0 * dc["prim_tests"] + 64.0 * dc["rays"]
                        ms = min(pj["kernel_ms"])
                        lines.append("algorithmic gather bytes = 48 x node_visits + 64 x prim_tests + 64 x rays = %.4g ; / %.3f ms = %.2f TB/s = %.3f of 18.8 (roofline.frac)" %
                                     (by, ms, by / ms / 1e9, by / ms / 1e9 / 18.8))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        by = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
        w = bench.WORKLOADS[W]
        alg = w["nx"] * w["ny"] * 12
        lines.append("bytes that leave L2 per frame = (2 x FETCH_SIZE + WRITE_SIZE) KB = %.1f MB read + %.1f MB written = %.1f MB; algorithmic framebuffer bytes %.1f MB -> x%.1f" %
                     (2 * g("FETCH_SIZE") / 1024, g("WRITE_SIZE") / 1024, by / 1e6, alg / 1e6, by / alg))
        if W == "C4":
            lines.append("(C4: the reads are node records and triangle slots - 6.7 MB touched, one XCD's L2 holds 4 MB, the Infinity Cache the rest; the kernel's "
                         "algorithmic GATHER bytes, which its roofline counts, are ~4 TB per frame: bench.py other_configs.C4.roofline)")
        out[W] = {"pmc": pmc, "bytes_per_frame": by, "fetch_kb_x2": 2 * g("FETCH_SIZE"), "write_kb": g("WRITE_SIZE"), "frame_ms": (sum(frames) / len(frames)) if frames else None,
                  "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes around tools/one_frame.py, FETCH x2 (gfx950), last full frame"}
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_{W}_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w"), indent=1)
print(json.dumps(out))
