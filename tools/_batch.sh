timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > gpurun_out/r04_t5_tests.txt 2>&1; tail -3 gpurun_out/r04_t5_tests.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err; python3 tools/brief.py < gpurun_out/r04_bench_a.json
