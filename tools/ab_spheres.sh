# tools/ab_spheres.sh reps lib... — A/B/... of builds of librt_mi355x.so, alternating, in ONE gpurun call.  Workload: bench.py's C2 frame (10 steps), or
# AB_W=C3|C5 with AB_SPP samples per pixel (C5 at 256 spp: a throughput-bound frame of 0.2 s - the C2 frame is bound by its longest pixel chain and hides
# what a change does to the dense iteration)
N=$1; shift
for i in $(seq $N); do for L in "$@"; do
  RT_LIB=$L python3 - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import cuda_raytracing_optimized_amd as rt
rt.RENDERER_LIB = os.path.abspath(os.environ["RT_LIB"])
import bench
b = bench.HipBackend(); w = dict(bench.WORKLOADS[os.environ.get("AB_W", "C2")])
if os.environ.get("AB_SPP"):
    w["spp"] = int(os.environ["AB_SPP"])
b.open(w, 0, 1, None); b.step(4)
ms = [b.step() for _ in range(int(os.environ.get("AB_STEPS", "10")))]
b.close()
print(os.environ["RT_LIB"], round(w["nx"] * w["ny"] * w["spp"] / (sum(ms) / len(ms)) / 1e3, 1), "min-frame", round(w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3, 1))
PY
done; done
