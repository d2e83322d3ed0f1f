# tools/sweep_mesh_occupancy.sh — the mesh kernel at 1..4 workgroups (= waves per SIMD) per CU: how the C4 frame scales with the waves in flight
for rep in 1 2; do for w in 4 3 2 1; do
  v=$(( w<<8 ))
  python3 tools/bench_mesh.py --steps 2 --spp 128 --variant $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves_per_simd $w', round(d['Msamples_per_s'],1))"
done; done
