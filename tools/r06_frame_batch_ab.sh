#!/bin/bash
# Round 6: the pass size of the per-frame stages (MDQE_FRAME_BATCH; default 40 frames at 360p = 20/40/40/20 with the tapered first and last pass) re-swept
# on the round's kernels, two interleaved rounds in one call + a 400-step sustained run.     bash tools/r06_frame_batch_ab.sh
cd "$(dirname "$0")/.."
out=gpurun_out/r6_frame_batch_ab.txt; : > $out
for rep in 1 2; do for fb in 40 30 48 60 24; do
  MDQE_FRAME_BATCH=$fb python bench.py --steps 10 --warmup 3 --no-fast-mode --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frame_batch $fb:', d['value'], d['value_median'], 'roofline', d['roofline']['frac'])" | tee -a $out
done; done
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-fast-mode > gpurun_out/r6_bench_sustained.json 2>/dev/null; python -c "import json; d=json.loads(open('gpurun_out/r6_bench_sustained.json').read()); print('sustained 400 steps:', d['value'], d['value_median'], d['ms_per_step'])" | tee -a $out
