#!/bin/bash
# tools/final_pass.sh <tag> — what profiles/ holds for the committed kernels, in ONE GPU call: the rocprofv3 passes (traffic.json keyed by the kernel source hash),
# the GPU test suite, the contract bench line (which then finds the matching traffic.json), the other modes, and a two-rank rehearsal of the N > 1 path on the one GPU.
TAG=${1:-r04}
O=gpurun_out
bash tools/measure_traffic.sh $TAG > $O/${TAG}_measure.log 2>&1 || exit 1
cp $O/traffic.json $O/${TAG}_C2_summary.txt $O/${TAG}_C4_summary.txt $O/${TAG}_C3_summary.txt $O/${TAG}_C5_summary.txt profiles/
{ bash tools/kernel_resources.sh spheres parity; bash tools/kernel_resources.sh mesh parity; } > profiles/${TAG}_kernel_resources.txt 2>&1      # the code objects of THESE sources
cp profiles/${TAG}_kernel_resources.txt $O/
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/${TAG}_gpu_tests_final.txt 2>&1; tail -2 $O/${TAG}_gpu_tests_final.txt
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_parity.json 2> $O/${TAG}_bench_parity.err || exit 1
python3 bench.py --no-cpu-baseline --fp fast > $O/${TAG}_bench_fpfast.json 2>/dev/null
python3 bench.py --no-cpu-baseline --rng counter > $O/${TAG}_bench_rngcounter.json 2>/dev/null
python3 bench.py --no-cpu-baseline --fp fast --rng counter > $O/${TAG}_bench_fpfastrngcounter.json 2>/dev/null
env -u HSA_ENABLE_IPC_MODE_LEGACY timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > $O/${TAG}_bench_n2_rehearsal.json 2> $O/${TAG}_bench_n2_rehearsal.err
# the traffic forms of the two-dispatch frame, each against the legacy form, on these sources: time + bytes leaving the L2 (C2), time (C3, C5 at 256 spp)
COMBOS="legacy:0:0:0:1 packed:1:0:0:1 xcd:0:1:0:1 p1seg:0:0:2:1 devfb:0:0:0:0 runs:1:1:2:0 runs_direct:1:1:2:1" timeout -k 10 600 bash tools/ab_traffic.sh C2 > $O/${TAG}_ab_traffic_c2.log 2>&1; cp $O/ab_traffic_C2.txt $O/${TAG}_ab_traffic_c2.txt
COMBOS="legacy:0:0:0:1 runs:1:1:2:0 runs_direct:1:1:2:1" NO_PMC=1 SWEEP_FRAMES=4 timeout -k 10 300 bash tools/ab_traffic.sh C3 > $O/${TAG}_ab_traffic_c3.log 2>&1; cp $O/ab_traffic_C3.txt $O/${TAG}_ab_traffic_c3.txt
COMBOS="legacy:0:0:0:1 xcd:0:1:0:1 runs:1:1:2:0 runs_direct:1:1:2:1" NO_PMC=1 SWEEP_SPP=256 SWEEP_FRAMES=8 timeout -k 10 300 bash tools/ab_traffic.sh C5 > $O/${TAG}_ab_traffic_c5_256spp.log 2>&1; cp $O/ab_traffic_C5.txt $O/${TAG}_ab_traffic_c5_256spp.txt
cat $O/${TAG}_bench_parity.json $O/${TAG}_bench_fpfast.json $O/${TAG}_bench_rngcounter.json $O/${TAG}_bench_fpfastrngcounter.json | python3 tools/brief.py
tail -c 600 $O/${TAG}_bench_n2_rehearsal.json
