#!/bin/bash
# Round 6 evidence in one gpurun call (final code): kernel statistics of the bench's main leg (rocprofv3 --kernel-trace --stats), the default bench line +
# its extras file, the 640p / Swin-L lines, the N > 1 path through a one-rank RCCL communicator and a four-rank gloo rehearsal (measured plan), the long
# fuzz runs, the GPU suite with durations + parity margins.      bash tools/r06_evidence.sh
cd "$(dirname "$0")/.."
root=$(pwd)
o=gpurun_out/r06
mkdir -p $o
# (the profiler's preload initialises the GPU: the program behind `--` must be the leg itself, not the orchestrator that starts children)
( cd /tmp && export TMPDIR=/tmp && MDQE_BENCH_LEG=main MDQE_BENCH_SIDE_CONFIGS=0 GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats -d "$root/$o/prof" -o bench_f32 -- python3 "$root/bench.py" --steps 5 --warmup 1 --no-fast-mode --no-cpu-baseline > "$root/$o/prof_bench.json" 2> "$root/$o/prof_bench.err" )
db=$(find $o/prof -name "*.db" | head -1)
echo "db: $db"
if [ -n "$db" ]; then
  python3 tools/rocprof_db_stats.py "$db" $o/r06_bench_f32_kernel_stats.csv > $o/r06_bench_f32_kernel_summary.txt 2>&1
  head -14 $o/r06_bench_f32_kernel_summary.txt
  rm -rf $o/prof
fi
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06_bench_line_360p.json 2> $o/r06_bench_line_360p.err ) 2> $o/r06_bench_line_360p.time; echo "bench rc=$?"; cat $o/r06_bench_line_360p.time; wc -c $o/r06_bench_line_360p.json
cp gpurun_out/bench_extras.json $o/r06_bench_extras_360p.json
python bench.py --config R50_ovis_720 --frames 60 --steps 5 --warmup 2 --no-fast-mode --no-cpu-baseline > $o/r06_bench_line_640p.json 2>/dev/null
python bench.py --config swinl_ovis --frames 40 --steps 5 --warmup 2 --no-fast-mode --no-cpu-baseline > $o/r06_bench_line_swinl.json 2>/dev/null
MDQE_BENCH_FORCE_SHARDED=1 MDQE_BENCH_LINE=full python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode > $o/r06_bench_sharded_one_rank_rccl.json 2>/dev/null
MDQE_BENCH_BACKEND=gloo MDQE_BENCH_ONE_DEVICE=1 MDQE_BENCH_HALO_AB=1 python bench.py --gpus 4 --steps 2 --warmup 1 --frames 48 --no-cpu-baseline --no-fast-mode > $o/r06_bench_n4_gloo_rehearsal.json 2>/dev/null; cp gpurun_out/bench_extras.json $o/r06_bench_n4_gloo_rehearsal_extras.json
export PYTHONPATH=$(pwd):$(pwd)/oracle:$PYTHONPATH
export OMP_NUM_THREADS=16                                   # the oracle inside the fuzzers: a GPU box shows 256 logical CPUs, its share is ~16
python tools/fuzz_msda.py 400 2>&1 | tail -3 > $o/r06_fuzz_msda.txt
python tools/fuzz_msda_fused.py 200 2>&1 | tail -3 > $o/r06_fuzz_msda_fused.txt
python tools/fuzz_tracker.py 400 --gpu 2>&1 | tail -3 > $o/r06_fuzz_tracker.txt
python tools/fuzz_pipeline.py 60 2>&1 | tail -3 > $o/r06_fuzz_pipeline.txt
python tools/fuzz_inference_clip.py 100 2>&1 | tail -12 > $o/r06_fuzz_inference_clip.txt
tail -2 $o/r06_fuzz_*.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=12 > $o/r06_gpu_tests.log 2>&1; tail -18 $o/r06_gpu_tests.log
cp gpurun_out/parity_margins.txt $o/r06_parity_margins.txt
