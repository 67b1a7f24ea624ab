cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  MDQE_DEC_TWO_STREAMS=$v MDQE_BENCH_FORCE_SHARDED=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sh$v -o sh -- python3 $R/bench.py --steps 3 --warmup 1 --no-fast-mode --no-cpu-baseline > $R/gpurun_out/sh$v.json 2> $R/gpurun_out/sh$v.err
  python3 $R/tools/rocprof_db_stats.py $(find $R/gpurun_out/prof_sh$v -name "*.db" | head -1) > $R/gpurun_out/r03_shard_prof_two$v.txt 2>&1
  rm -rf $R/gpurun_out/prof_sh$v
  tail -1 $R/gpurun_out/sh$v.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('TWO_STREAMS=$v', d['value'], d['ms_per_step'])"
  head -24 $R/gpurun_out/r03_shard_prof_two$v.txt
done
