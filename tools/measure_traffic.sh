#!/bin/bash
# tools/measure_traffic.sh <tag> — rocprofv3 around tools/one_frame.py for C2 (sphere kernel) and C4 (mesh kernel), and the traffic passes for C3 / C5:
#   pass 1  --kernel-trace --stats            (per-kernel durations)
#   pass 2  --pmc SQ_* (issue / lane utilisation / waits)
#   pass 3  --pmc GRBM_GUI_ACTIVE FETCH_SIZE  pass 4  --pmc WRITE_SIZE      (HBM-side traffic; separate passes, never with other traces)
# then tools/make_traffic_json.py writes profiles/<tag>_{C2,C4,C3,C5}_summary.txt and profiles/traffic.json (keyed by the kernel source hash).
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $REPO
for W in C2 C4 C3 C5; do
  OUT=$REPO/gpurun_out/prof_${TAG}_$W
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/one_frame.py $W 3 > $OUT/trace.log 2>&1
  RT_ONE_FRAME_COUNTERS=1 python3 tools/one_frame.py $W 3 > $OUT/plain.log 2>&1        # un-profiled frame times + the device counters of the frame
  if [ $W = C3 ] || [ $W = C5 ]; then FR=1; else FR=2; fi      # (C3 / C5: the traffic passes only - their frames are the C2 kernel at more samples)
  if [ $FR = 2 ]; then
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o pmc1 -- python3 tools/one_frame.py $W 2 > $OUT/pmc1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc2 -o pmc2 -- python3 tools/one_frame.py $W 2 > $OUT/pmc2.log 2>&1
  fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc3 -- python3 tools/one_frame.py $W $FR > $OUT/pmc3.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o pmc4 -- python3 tools/one_frame.py $W $FR > $OUT/pmc4.log 2>&1
  echo "$W done"; tail -1 $OUT/trace.log
done
python3 tools/make_traffic_json.py $TAG
