set -x
free -g | head -2
python tools/msda_dec_scaling.py > gpurun_out/r6_msda_scaling.txt 2>&1; tail -12 gpurun_out/r6_msda_scaling.txt
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "shipped_schedule" > gpurun_out/r6_test_shipped.log 2>&1; tail -5 gpurun_out/r6_test_shipped.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_default2.json 2> gpurun_out/r6_bench_default2.err ) 2> gpurun_out/r6_bench_default2.time
cat gpurun_out/r6_bench_default2.time
