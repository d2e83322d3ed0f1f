# tools/sweep_mesh_top.sh — LDS-resident top of the BVH (RT_MESH_TOP child-pair records per workgroup) on scenes whose whole tree fits
run() { # lib variant top detail
  RT_MESH_TOP=$3 python3 tools/bench_mesh.py --steps 1 --spp 128 --detail $4 --variant $2 ${1:+--lib $1} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib ${1:-default} variant $2 top $3 detail $4', round(d['Msamples_per_s'],1), 'nodes/sample', round(d['nodes'],1))"
}
for t in 0 512; do run "" 0 $t 1; done
for t in 0 2048; do run build/exp/librt_w16.so $((1<<8)) $t 2; done
for t in 0 2048; do run build/exp/librt_w16.so $((1<<8)) $t 4; done
