# tools/sweep_tail.sh — C2 frame time under scheduling constants of the chain tier (RT_TUNE = chain_every,chain_waves,heavy_thr,n_chain,boost,chain_pixels,chain_pixels_long;
# RT_CHAIN_SINGLE=1: chain waves grab one pixel per atomic), all in ONE GPU call, each point 12 frames after a warm-up.
for spec in ${SWEEP:-"0:1,1,8,3,2,4,2" "1:1,1,8,3,2,4,2" "0:1,1,8,3,2,4,1" "1:1,1,8,3,2,4,1" "1:1,1,8,3,2,2,1" "1:1,2,8,3,2,4,1" "1:1,2,8,3,2,2,1" "1:1,1,8,3,2,3,1"}; do
  S=${spec%%:*}; T=${spec#*:}
  RT_CHAIN_SINGLE=$S RT_TUNE=$T python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
b = bench.HipBackend(); w = bench.WORKLOADS[os.environ.get("SWEEP_W", "C2")]
b.open(w, 0, 1, None); b.step(4)
ms = [b.step() for _ in range(12)]
b.close()
print("single", os.environ["RT_CHAIN_SINGLE"], "tune", os.environ["RT_TUNE"], "Msamples/s mean", round(w["nx"] * w["ny"] * w["spp"] / (sum(ms) / len(ms)) / 1e3, 1), "best", round(w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3, 1), "ms", round(sum(ms) / len(ms), 3))
PY
done
