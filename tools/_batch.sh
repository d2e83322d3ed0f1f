timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > gpurun_out/r04_t13_div64_tests.txt 2>&1; tail -4 gpurun_out/r04_t13_div64_tests.txt
AB_STEPS=10 bash tools/ab2.sh 3 build/ab/nodiv64.so build/ab/div64.so > gpurun_out/r04_ab_div64_c2.txt 2>&1; cat gpurun_out/r04_ab_div64_c2.txt
AB_W=C5 AB_SPP=256 AB_STEPS=5 bash tools/ab2.sh 2 build/ab/nodiv64.so build/ab/div64.so "build/ab/nodiv64.so;RT_LEAN6_PIXELS=0" "build/ab/div64.so;RT_LEAN6_PIXELS=0" > gpurun_out/r04_ab_div64_c5.txt 2>&1; cat gpurun_out/r04_ab_div64_c5.txt
bash tools/ab_mesh.sh 2 build/ab/nodiv64.so build/ab/div64.so > gpurun_out/r04_ab_div64_c4.txt 2>&1; cat gpurun_out/r04_ab_div64_c4.txt
