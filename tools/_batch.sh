timeout -k 10 400 python3 -m pytest tests/test_gpu_parity_spheres.py tests/test_gpu_parity_configs.py -m gpu -q -x > gpurun_out/r04_t12_fast_tests.txt 2>&1; tail -3 gpurun_out/r04_t12_fast_tests.txt
AB_STEPS=10 bash tools/ab2.sh 3 build/ab/cur.so build/ab/fast.so > gpurun_out/r04_ab_fastpath_c2.txt 2>&1; cat gpurun_out/r04_ab_fastpath_c2.txt
AB_W=C3 AB_STEPS=2 bash tools/ab2.sh 2 build/ab/cur.so build/ab/fast.so > gpurun_out/r04_ab_fastpath_c3.txt 2>&1; cat gpurun_out/r04_ab_fastpath_c3.txt
AB_W=C5 AB_SPP=256 AB_STEPS=5 bash tools/ab2.sh 2 build/ab/cur.so build/ab/fast.so > gpurun_out/r04_ab_fastpath_c5.txt 2>&1; cat gpurun_out/r04_ab_fastpath_c5.txt
