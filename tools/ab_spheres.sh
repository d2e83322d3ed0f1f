# tools/ab_spheres.sh reps lib... — A/B/... of builds of librt_mi355x.so on the C2 frame (bench.py's headline job, 10 steps), alternating, in ONE gpurun call
N=$1; shift
for i in $(seq $N); do for L in "$@"; do
  RT_LIB=$L python3 - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import cuda_raytracing_optimized_amd as rt
rt.RENDERER_LIB = os.path.abspath(os.environ["RT_LIB"])
import bench
b = bench.HipBackend(); w = bench.WORKLOADS["C2"]
b.open(w, 0, 1, None); b.step(4)
ms = [b.step() for _ in range(10)]
b.close()
print(os.environ["RT_LIB"], round(w["nx"] * w["ny"] * w["spp"] / (sum(ms) / len(ms)) / 1e3, 1), "min-frame", round(w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3, 1))
PY
done; done
