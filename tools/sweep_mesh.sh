# tools/sweep_mesh.sh — thresholds of the mesh kernel's phase machine: variant = trav << 24 | min_traversing << 16 | leaf_thr << 26
for trav in ${TRAV:-0 2}; do for mt in ${MT:-16 24 32}; do for lt in ${LT:-12 16 24}; do
  v=$(( (trav<<24) | (mt<<16) | (lt<<26) ))
  python3 tools/bench_mesh.py --steps 1 --spp 128 --variant $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trav $trav min_traversing $mt leaf_thr $lt', round(d['Msamples_per_s'],1))"
done; done; done
