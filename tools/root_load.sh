#!/bin/bash
# The N = 2 / 4 / 8 ROOT-LOAD rehearsal on one GPU (bench.py MDQE_BENCH_ROOT_LOAD, sharding.expand_root_load): rank 0 of an N-rank job computes its
# own 120 frames per step while its replay thread is fed the clips of all N ranks.  One bench line per N -> gpurun_out/root_load_N.json (the
# summaries judged are copied to profiles/r05_root_load_N.json).      bash tools/root_load.sh [worlds...]
# HALO=1: the same with the halo exchange (a chunk's own tail stands in for its neighbour's message) -> gpurun_out/root_load_N_halo.json
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
extra=""; sfx=""
if [ "${HALO:-0}" = "1" ]; then extra="--halo-exchange"; sfx="_halo"; fi
for w in "${@:-1 2 4 8}"; do
  for ww in $w; do
    until=0
    if [ "$ww" -ge 2 ]; then      # rank 0 rests in the last round: the OTHER ranks first (rank 1's chunks, compute + gather, no replay) ...
      MDQE_BENCH_ROOT_LOAD=$ww MDQE_BENCH_AS_RANK=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fast-mode $extra > gpurun_out/root_load_${ww}${sfx}_rank1.json 2> gpurun_out/root_load_${ww}${sfx}_rank1.err
      until=$(python - "$ww$sfx" <<'P'
import json, sys
d = json.load(open("gpurun_out/root_load_%s_rank1.json" % sys.argv[1]))
p = d["scaling_breakdown"]["per_rank_ms"]
sys.stderr.write("root load %s, as rank 1: %.1f ms/step; compute %.1f\n" % (sys.argv[1], d["ms_per_step"], p["compute"][0]))
print("%.2f" % (p["compute"][0] + p["pack"][0]))
P
)
    fi
    # ... then rank 0, held at the last gather until rank 1 would have delivered
    MDQE_BENCH_REST_UNTIL_MS=$until MDQE_BENCH_ROOT_LOAD=$ww python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fast-mode $extra > gpurun_out/root_load_$ww$sfx.json 2> gpurun_out/root_load_$ww$sfx.err
    python - "$ww$sfx" <<'P'
import json, sys
w = sys.argv[1]
d = json.load(open("gpurun_out/root_load_%s.json" % w))
sb = d["scaling_breakdown"]["per_rank_ms"]
print("root load %s: %.1f frames/s per rank, %.1f ms/step; compute %.1f replay_busy %.1f replay_exposed %.1f gather %.1f tracks %s" % (
    w, d["value"], d["ms_per_step"], sb["compute"][0], sb["replay_busy"][0], sb["replay_exposed"][0], sb["gather_wait"][0] + sb["gather_payload"][0],
    d["config"]["tracked_instances"]), flush=True)
P
  done
done
