# tools/sweep_env.sh VAR v1 v2 ... — frame time of SWEEP_W (default C2) with environment variable VAR set to each value, alternating REPS times, in ONE GPU call
VAR=$1; shift
for rep in $(seq ${REPS:-2}); do for v in "$@"; do
  env $VAR=$v python3 - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import bench
b = bench.HipBackend(); w = bench.WORKLOADS[os.environ.get("SWEEP_W", "C2")]
b.open(w, 0, 1, None); b.step(4)
n = int(os.environ.get("SWEEP_FRAMES", "12"))
ms = [b.step() for _ in range(n)]
b.close()
print("$VAR", "$v", "Msamples/s mean", round(w["nx"] * w["ny"] * w["spp"] / (sum(ms) / len(ms)) / 1e3, 1), "best", round(w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3, 1), "ms", round(sum(ms) / len(ms), 3))
PY
done; done
