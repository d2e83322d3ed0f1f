#!/bin/bash
# tools/ab_traffic.sh [workload] — the traffic forms of the sphere kernel's two-dispatch frame (RT_ORD_PACKED / RT_XCD_QUEUES / RT_P1_TILE / RT_FB_DIRECT), each
# combination measured in ONE GPU call: wall time per frame (bench.HipBackend: what bench.py's `value` is made of) and the bytes that leave the L2 per frame
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes around tools/one_frame.py).  Output: gpurun_out/ab_traffic_<workload>.txt
W=${1:-C2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $REPO
O=$REPO/gpurun_out/ab_traffic_$W
rm -rf $O; mkdir -p $O
# name : RT_ORD_PACKED RT_XCD_QUEUES RT_P1_TILE RT_FB_DIRECT
COMBOS=${COMBOS:-"base:0:0:0:1 packed:1:0:0:1 p1tile:0:0:1:1 p1seg:0:0:2:1 xcd:0:1:0:1 devfb:0:0:0:0 xcd_devfb:0:1:0:0 all:1:1:1:0 all_seg:1:1:2:0 all_seg_direct:1:1:2:1"}
for rep in 1 2; do for c in $COMBOS; do
  IFS=: read name a b t d <<< "$c"
  export RT_ORD_PACKED=$a RT_XCD_QUEUES=$b RT_P1_TILE=$t RT_FB_DIRECT=$d
  SWEEP_W=$W python3 - >> $O/time_$name.txt <<PY
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
b = bench.HipBackend(); w = dict(bench.WORKLOADS[os.environ.get("SWEEP_W", "C2")])
if os.environ.get("SWEEP_SPP"): w["spp"] = int(os.environ["SWEEP_SPP"])        # (timing only: a shorter frame of the same geometry)
b.open(w, 0, 1, None); b.step(4)
n = int(os.environ.get("SWEEP_FRAMES", "12"))
ms = []; wall = []
for _ in range(n):
    t0 = time.perf_counter(); ms.append(b.step()); wall.append((time.perf_counter() - t0) * 1e3)      # kernel time (HIP events) / wall time of runRenderer (delivery included)
b.close()
r = lambda x: round(w["nx"] * w["ny"] * w["spp"] / x / 1e3, 1)
print(r(sum(ms) / len(ms)), r(min(ms)), r(sum(wall) / len(wall)))
PY
  echo "$name rep $rep: $(tail -1 $O/time_$name.txt)"
done; done
for c in $COMBOS; do
  [ -n "$NO_PMC" ] && break
  IFS=: read name a b t d <<< "$c"
  export RT_ORD_PACKED=$a RT_XCD_QUEUES=$b RT_P1_TILE=$t RT_FB_DIRECT=$d
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$name/rd -o rd -- python3 tools/one_frame.py $W 2 > $O/$name.rd.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$name/wr -o wr -- python3 tools/one_frame.py $W 2 > $O/$name.wr.log 2>&1
  echo "$name pmc done"
done
python3 tools/ab_traffic_summary.py $W | tee $REPO/gpurun_out/ab_traffic_$W.txt
