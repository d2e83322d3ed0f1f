# tools/sweep_chain.sh — A/B of the scheduling constants of the cost-ordered second phase (RT_TUNE = chain_every,chain_waves,heavy_thr,n_chain,boost)
for t in ${SWEEP:-"1,1,10,3,4" "1,1,8,3,2"}; do
  RT_TUNE=$t python3 bench.py --no-cpu-baseline --no-other-configs --steps 8 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value']), round(d['frame_ms_kernel'],3))"
done
