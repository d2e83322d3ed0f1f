#!/bin/bash
# tools/collect_artefacts.sh <tag> — everything profiles/ holds for one kernel generation, in one GPU call:
# GPU test suite, contract bench line, rocprofv3 passes, wave / pixel time lines, the other modes and configs.
TAG=${1:-run}
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -q 2>&1 | tail -2 || exit 1
python3 bench.py > $O/bench_${TAG}.json || exit 1
bash tools/profile.sh $TAG > $O/profile_${TAG}.log 2>&1
RT_WAVE_DEBUG=$O/wave_dbg.bin python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pixel_debug.py $O/wave_dbg.bin.px > $O/pixel_timeline_${TAG}.txt
python3 tools/wave_debug.py $O/wave_dbg.bin > $O/wave_timeline_${TAG}.txt
rm -f $O/wave_dbg.bin*
python3 bench.py --no-cpu-baseline --fp fast > $O/bench_${TAG}_fpfast.json
python3 bench.py --no-cpu-baseline --rng counter > $O/bench_${TAG}_rngcounter.json
python3 bench.py --no-cpu-baseline --fp fast --rng counter > $O/bench_${TAG}_fpfastrngcounter.json
cat $O/bench_${TAG}.json $O/bench_${TAG}_fpfast.json $O/bench_${TAG}_rngcounter.json $O/bench_${TAG}_fpfastrngcounter.json | python3 tools/brief.py
timeout -k 10 600 python3 tools/bench_configs.py --full > $O/configs_${TAG}.txt
cat $O/configs_${TAG}.txt
