set -x
python tools/msda_patch_ab.py > gpurun_out/r6_msda_patch_ab.txt 2>&1; tail -8 gpurun_out/r6_msda_patch_ab.txt
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py tests/test_pipeline_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu > gpurun_out/r6_tests_patch.log 2>&1; tail -5 gpurun_out/r6_tests_patch.log
for i in 1 2; do for pz in 1 0; do MDQE_MSDA_PATCH=$pz python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('patch=$pz', d['value'], d['value_median'], d['roofline_msda']['avg_launch_us'], d['roofline_msda']['avg_launch_us_isolated'])" | tee -a gpurun_out/r6_patch_bench_ab.txt; done; done
