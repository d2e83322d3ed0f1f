# tools/ab_mesh.sh reps lib... — A/B/... of builds of librt_mi355x.so on the C4 frame, alternating, in ONE gpurun call (boxes differ by a few %)
N=$1; shift
for i in $(seq $N); do for L in "$@"; do
  python3 tools/bench_mesh.py --steps 4 --lib $L 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['Msamples_per_s'],1))"
done; done
