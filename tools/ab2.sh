#!/bin/bash
# tools/ab2.sh reps spec... — like tools/ab_spheres.sh, but every contestant is "lib[;VAR=value...]": a build of librt_mi355x.so plus the environment it
# runs under, alternating, in ONE gpurun call.  AB_W=C2|C3|C5, AB_SPP, AB_STEPS as there; AB_NX / AB_NY override the image size.
N=$1; shift
for i in $(seq $N); do for S in "$@"; do
  L=${S%%;*}; E=""
  if [ "$L" != "$S" ]; then E=$(echo "${S#*;}" | tr ";" " "); fi
  env $E RT_LIB=$L RT_SPEC=$S python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import cuda_raytracing_optimized_amd as rt
rt.RENDERER_LIB = os.path.abspath(os.environ["RT_LIB"])
import bench
b = bench.HipBackend(); w = dict(bench.WORKLOADS[os.environ.get("AB_W", "C2")])
if os.environ.get("AB_SPP"):
    w["spp"] = int(os.environ["AB_SPP"])
if os.environ.get("AB_NX"):
    w["nx"], w["ny"] = int(os.environ["AB_NX"]), int(os.environ["AB_NY"])
b.open(w, 0, 1, None); b.step(4)
ms = [b.step() for _ in range(int(os.environ.get("AB_STEPS", "10")))]
b.close()
print(os.environ["RT_SPEC"], round(w["nx"] * w["ny"] * w["spp"] / (sum(ms) / len(ms)) / 1e3, 1), "min-frame", round(w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3, 1), flush=True)
PY
done; done
