#!/bin/bash
# Round 5 evidence in one gpurun call: kernel statistics of the bench (rocprofv3 --kernel-trace --stats), the copyBuffer histogram, the PMC passes over the
# round's GEMM kernels, the root-load rehearsal at N = 1 / 2 / 4 / 8, the default bench line.      bash tools/r05_evidence.sh
cd "$(dirname "$0")/.."
root=$(pwd)
mkdir -p gpurun_out/r05
export MDQE_BENCH_SIDE_CONFIGS=0 MDQE_BENCH_ROOT_LOAD_LEG=0 GPU_MAX_HW_QUEUES=8
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/r05/prof" -o bench_f32 -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-fast-mode --no-cpu-baseline > "$root/gpurun_out/r05/prof_bench.json" 2> "$root/gpurun_out/r05/prof_bench.err" )
db=$(find gpurun_out/r05/prof -name "*.db" | head -1)
echo "db: $db"
if [ -n "$db" ]; then
  python3 tools/rocprof_db_stats.py "$db" gpurun_out/r05/r05_bench_f32_kernel_stats.csv > gpurun_out/r05/r05_bench_f32_kernel_summary.txt 2>&1
  python3 tools/rocprof_db_hist.py "$db" copyBuffer > gpurun_out/r05/r05_copybuffer_hist.txt 2>&1
  head -12 gpurun_out/r05/r05_bench_f32_kernel_summary.txt
  rm -rf gpurun_out/r05/prof
fi
unset MDQE_BENCH_SIDE_CONFIGS MDQE_BENCH_ROOT_LOAD_LEG
bash tools/pmc_gemm_r05.sh gpurun_out/r05/pmc > gpurun_out/r05/r05_pmc_gemm_summary.txt 2>&1
find gpurun_out/r05/pmc -type d -name "p[0-9]" -exec rm -rf {} + 2>/dev/null
tail -40 gpurun_out/r05/r05_pmc_gemm_summary.txt
bash tools/root_load.sh 1 2 4 8
python bench.py > gpurun_out/r05/r05_bench_line_360p.json 2> gpurun_out/r05/r05_bench_line_360p.err; echo "bench rc=$?"
