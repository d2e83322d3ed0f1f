#!/usr/bin/env python3
"""tools/one_frame.py <C2|C3|C4|C5> [frames] — renders bench.py's workload of that name: one tiny warm-up frame (1 spp), then `frames`
full frames, and nothing else.  The program rocprofv3 is wrapped around (tools/measure_traffic.sh): per-frame counters are the
LAST frames' dispatches."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = bench.WORKLOADS[name]
b = bench.HipBackend()
b.open(w, 0, 1, None)
b.step(1)
ms = [b.step() for _ in range(frames)]
out = {"workload": name, "frames": frames, "kernel_ms": ms, "Msamples_per_s": w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3}
if os.environ.get("RT_ONE_FRAME_COUNTERS"):          # the inputs of bench.py's roofline.frac: device counters of the same frame (an extra, untimed frame)
    # (the counting instantiations are slow - the mesh one does a device atomic per statistics event: count a few samples per pixel and scale, as bench.py does)
    cspp = min(w["spp"], 100 if w["kind"] == "spheres" else 4)
    c = b.counted(cspp)
    out["device_counters"] = {k: (v * w["spp"] / cspp if k != "spp" else w["spp"]) for k, v in c.items()}
    out["device_counters_counted_spp"] = cspp
b.close()
print(json.dumps(out))
