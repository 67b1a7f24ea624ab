set -x
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err ) 2> gpurun_out/r6_bench_default.time
wc -c gpurun_out/r6_bench_default.json; cat gpurun_out/r6_bench_default.time
cp gpurun_out/bench_extras.json gpurun_out/r6_bench_default_extras.json
for q in 4 8; do GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline > gpurun_out/r6_q$q.json 2>/dev/null; cp gpurun_out/bench_extras.json gpurun_out/r6_q${q}_extras.json; done
for q in 4 8; do GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline > gpurun_out/r6_q${q}b.json 2>/dev/null; done
timeout -k 10 900 python -m pytest tests/test_bench_gpu.py -x -q > gpurun_out/r6_test_bench.log 2>&1; tail -5 gpurun_out/r6_test_bench.log
