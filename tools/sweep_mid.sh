# tools/sweep_mid.sh — C2 frame time when the chain waves hold MORE pixels in a wider sparse form: spec = variant:RT_TUNE (variant bits 30..31: sparse form at <= 4 / 8 / 12 / 16
# live rays; RT_TUNE = chain_every,chain_waves,heavy_thr,n_chain,boost,chain_pixels,caps of list 0,1,2), one GPU call
for spec in ${SWEEP}; do
  V=${spec%%:*}; T=${spec#*:}
  RT_BENCH_VARIANT=$V RT_TUNE=$T python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
b = bench.HipBackend(variant=int(os.environ["RT_BENCH_VARIANT"])); w = bench.WORKLOADS[os.environ.get("SWEEP_W", "C2")]
b.open(w, 0, 1, None); b.step(4)
ms = [b.step() for _ in range(10)]
b.close()
print("variant", os.environ["RT_BENCH_VARIANT"], "tune", os.environ["RT_TUNE"], "Msamples/s mean", round(w["nx"] * w["ny"] * w["spp"] / (sum(ms) / len(ms)) / 1e3, 1), "best", round(w["nx"] * w["ny"] * w["spp"] / min(ms) / 1e3, 1), "ms", round(sum(ms) / len(ms), 3))
PY
done
