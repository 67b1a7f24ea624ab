set -x
timeout -k 10 1000 python -m pytest tests/test_fullsize_gpu.py -x -q --durations=12 > gpurun_out/r6_test_fullsize.log 2>&1; tail -25 gpurun_out/r6_test_fullsize.log
for i in 1 2; do
  MDQE_BENCH_SIDE_CONFIGS=0 MDQE_BENCH_ROOT_LOAD_LEG=0 python bench_r05_tmp.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05 bench.py', d['value'], d['value_median'])" | tee -a gpurun_out/r6_bench_regress.txt
  python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r06 bench.py', d['value'], d['value_median'])" | tee -a gpurun_out/r6_bench_regress.txt
done
bash tools/msda_aux_ab.sh run > gpurun_out/r6_msda_aux_ab.txt 2>&1; tail -50 gpurun_out/r6_msda_aux_ab.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_default4.json 2> gpurun_out/r6_bench_default4.err ) 2> gpurun_out/r6_bench_default4.time
cat gpurun_out/r6_bench_default4.time; cp gpurun_out/bench_extras.json gpurun_out/r6_bench_default4_extras.json; wc -c gpurun_out/r6_bench_default4.json
