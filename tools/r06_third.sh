set -x
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_default3.json 2> gpurun_out/r6_bench_default3.err ) 2> gpurun_out/r6_bench_default3.time
cat gpurun_out/r6_bench_default3.time; cp gpurun_out/bench_extras.json gpurun_out/r6_bench_default3_extras.json
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/r6_gpu_tests_a.log 2>&1; tail -40 gpurun_out/r6_gpu_tests_a.log
