# tools/sweep_tail.sh — C2 frame time under the scheduling constants of the chain tier, all in ONE GPU call, each point 12 frames after a warm-up.
# spec = single_ray:grab_single:top_thr:RT_TUNE   (RT_SINGLE_RAY: scan_single for waves with one live ray; RT_CHAIN_SINGLE=1: chain waves grab one pixel per
# atomic; RT_TOP_THR: 16 x rays per sample from which a pixel goes to chain list 0; RT_TUNE = chain_every,chain_waves,heavy_thr,n_chain,boost,chain_pixels,
# chain_pixels of list 0,1,2)
for spec in ${SWEEP:-"0:0:416:1,1,8,4,2,4,4,4,4" "1:0:416:1,1,8,4,2,4,4,4,4" "1:1:416:1,1,8,4,2,4,1,4,4" "1:1:384:1,1,8,4,2,4,1,4,4" "1:1:448:1,1,8,4,2,4,1,4,4" "1:1:416:1,1,8,4,2,4,1,2,4" "1:1:416:1,1,8,4,2,4,1,1,4" "1:1:416:2,3,8,4,2,4,1,1,4" "1:1:416:2,3,8,4,2,4,1,2,4" "1:1:416:2,3,8,4,2,4,1,1,2" "1:0:416:2,3,8,4,2,2,2,2,2"}; do
  SR=${spec%%:*}; R=${spec#*:}; S=${R%%:*}; R=${R#*:}; TH=${R%%:*}; T=${R#*:}
  RT_SINGLE_RAY=$SR RT_CHAIN_SINGLE=$S RT_TOP_THR=$TH RT_TUNE=$T python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
b = bench.HipBackend(); w = bench.WORKLOADS[os.environ.get("SWEEP_W", "C2")]
b.open(w, 0, 1, None); b.step(4)
ms = [b.step() for _ in range(12)]
b.close()
print("single_ray", os.environ["RT_SINGLE_RAY"], "grab1", os.environ["RT_CHAIN_SINGLE"], "top_thr", os.environ["RT_TOP_THR"], "tune", os.environ["RT_TUNE"], "Msamples/s mean", round(w["nx"] * w["ny"] * w["spp"] / (sum(ms) / len(ms)) / 1e3, 1), "best", round(w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3, 1), "ms", round(sum(ms) / len(ms), 3))
PY
done
