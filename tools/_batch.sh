timeout -k 10 500 python3 -m pytest tests/test_gpu_parity_spheres.py -m gpu -q -x > gpurun_out/r04_t10_hyb_tests.txt 2>&1; tail -3 gpurun_out/r04_t10_hyb_tests.txt
export SWEEP_N=1500,2000,2600,4000
python3 tools/sweep_scene_sizes.py plane > gpurun_out/r04_sweep_sizes_plane_d.txt 2>&1; cat gpurun_out/r04_sweep_sizes_plane_d.txt
python3 tools/sweep_scene_sizes.py volume > gpurun_out/r04_sweep_sizes_volume_d.txt 2>&1; cat gpurun_out/r04_sweep_sizes_volume_d.txt
echo "== RT_HYBRID_TWO=1"
RT_HYBRID_TWO=1 python3 tools/sweep_scene_sizes.py plane 2>&1 | tee -a gpurun_out/r04_sweep_sizes_plane_d.txt
RT_HYBRID_TWO=1 python3 tools/sweep_scene_sizes.py volume 2>&1 | tee -a gpurun_out/r04_sweep_sizes_volume_d.txt
