#!/bin/bash
# tools/pmc_mesh_ta.sh — texture-addresser / L1 counters of the mesh kernel (C4 frame), separate --pmc passes
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $REPO
OUT=$REPO/gpurun_out/pmc_mesh_ta
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python3 tools/one_frame.py C4 1 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_mesh_ta"
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); last = {}
    for r in csv.DictReader(open(f)):
        if "mesh_queue" not in r["Kernel_Name"]: continue
        last[(r["Counter_Name"])] = (r["Dispatch_Id"], float(r["Counter_Value"]))
        acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    disp = max(int(d) for d, _ in acc) if acc else None
    for (d, n), v in sorted(acc.items()):
        if int(d) == disp: print(f"{n:45s} {v:.4g}")
PY
