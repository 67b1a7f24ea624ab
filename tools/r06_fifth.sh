set -x
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r6_gpu_tests_b.log 2>&1; tail -30 gpurun_out/r6_gpu_tests_b.log
