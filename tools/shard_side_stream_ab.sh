#!/bin/bash
# The decoder's side stream inside the sharded schedule (MDQE_SHARD_SIDE_STREAMS=1) under shifted stream -> hardware-queue deals, as rank 1 of an
# 8-rank job (halo exchange): does ANY deal give the sharded schedule what the single-GPU path gets from the second decoder stream (+2.3 %)?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {   # name, env...
  name=$1; shift
  env "$@" MDQE_BENCH_ROOT_LOAD=8 MDQE_BENCH_AS_RANK=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast-mode --halo-exchange 2>/dev/null > gpurun_out/ssab.json
  python - "$name" <<'P'
import json, sys
d = json.load(open("gpurun_out/ssab.json"))
print("%-40s %.1f ms/step  compute %.1f" % (sys.argv[1], d["ms_per_step"], d["scaling_breakdown"]["per_rank_ms"]["compute"][0]), flush=True)
P
}
run "side off (default)" MDQE_SHARD_SIDE_STREAMS=0
run "side on" MDQE_SHARD_SIDE_STREAMS=1
for p in 1 2 3 5; do run "side on, pad $p" MDQE_SHARD_SIDE_STREAMS=1 MDQE_STREAM_PAD=$p MDQE_STREAM_TOUCH=1; done
run "side on, order cftwia" MDQE_SHARD_SIDE_STREAMS=1 MDQE_STREAM_ORDER=cftwia
run "side on, order wicfta" MDQE_SHARD_SIDE_STREAMS=1 MDQE_STREAM_ORDER=wicfta
run "side on, order cfitwa" MDQE_SHARD_SIDE_STREAMS=1 MDQE_STREAM_ORDER=cfitwa
run "side off (default) again" MDQE_SHARD_SIDE_STREAMS=0
