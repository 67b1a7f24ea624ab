#!/bin/bash
# Round 6, last call on the final code: the GPU suite (log + parity margins) and the driver's bench command.     bash tools/r06_final.sh
cd "$(dirname "$0")/.."
o=gpurun_out/r06
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=12 > $o/r06_gpu_tests.log 2>&1; tail -20 $o/r06_gpu_tests.log
cp gpurun_out/parity_margins.txt $o/r06_parity_margins.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06_bench_line_360p.json 2> $o/r06_bench_line_360p.err ) 2> $o/r06_bench_line_360p.time; echo "bench rc=$?"; cat $o/r06_bench_line_360p.time; wc -c $o/r06_bench_line_360p.json
cp gpurun_out/bench_extras.json $o/r06_bench_extras_360p.json
python -c 'import __graft_entry__ as g; g.smoke()' > $o/r06_smoke.txt 2>&1; tail -3 $o/r06_smoke.txt
