timeout -k 10 500 python3 -m pytest tests/test_gpu_parity_spheres.py tests/test_gpu_parity_configs.py -m gpu -q -x > gpurun_out/r04_t9_op4_tests.txt 2>&1; tail -5 gpurun_out/r04_t9_op4_tests.txt
