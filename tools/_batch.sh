timeout -k 10 300 python3 -m pytest tests/test_gpu_parity_mesh.py tests/test_gpu_parity_spheres.py -m gpu -q -x > gpurun_out/r04_t6_mesh_tests.txt 2>&1; tail -3 gpurun_out/r04_t6_mesh_tests.txt
RT_MESH_LEAN=0 bash tools/ab_mesh.sh 1 build/ab/mesh4.so > gpurun_out/r04_ab_mesh_lean.txt 2>&1
bash tools/ab_mesh.sh 2 build/ab/r03.so build/ab/mesh4.so build/ab/mesh5.so build/ab/mesh6.so >> gpurun_out/r04_ab_mesh_lean.txt 2>&1; cat gpurun_out/r04_ab_mesh_lean.txt
