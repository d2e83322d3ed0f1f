#!/bin/bash
# tools/profile.sh <tag> [bench args...] — rocprofv3 passes for bench.py on the GPU box.
# Pass 1: kernel trace + stats.  Passes 2..4: PMC counters, each in its own run (never with other traces).
# Results land under gpurun_out/prof_<tag>/ ; copy the summaries to profiles/ to have them judged.
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
PROG=bench.py
if [ -n "${PROF_PROG:-}" ]; then PROG=$PROF_PROG; ARGS="$*"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $PROG $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o pmc1 -- python3 $PROG $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc2 -o pmc2 -- python3 $PROG $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc3 -- python3 $PROG $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o pmc4 -- python3 $PROG $ARGS > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -20
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
