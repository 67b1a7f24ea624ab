#!/bin/bash
# A/B runs of bench.py under different environments / arguments, one line of summary each.
#   bash tools/bench_ab.sh OUT.txt "label|ENV=1 ENV2=x|--bench --args" ...
# Each case runs `python bench.py --no-cpu-baseline --no-fast-mode <args>` with the environment given; the JSON lines go to gpurun_out/ab_<label>.json.
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p gpurun_out
: > "$out"
for spec in "$@"; do
  IFS='|' read -r label envs args <<< "$spec"
  env $envs MDQE_BENCH_SIDE_CONFIGS=0 MDQE_BENCH_ROOT_LOAD_LEG=0 python bench.py --no-cpu-baseline --no-fast-mode $args > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { echo "$label: FAILED" | tee -a "$out"; tail -5 gpurun_out/ab_$label.err | tee -a "$out"; continue; }
  grep "tracker native" gpurun_out/ab_$label.err | tee -a "$out"
  python - "$label" "$envs" "$args" <<'P' | tee -a "$out"
import json, sys
label, envs, args = sys.argv[1:4]
d = json.load(open("gpurun_out/ab_%s.json" % label))
s = "%-28s %7.1f frames/s (median %7.1f) %6.1f ms/step" % (label, d["value"], d.get("value_median") or 0, d["ms_per_step"])
sb = d.get("scaling_breakdown")
if sb:
    p = sb["per_rank_ms"]
    s += "  compute %.1f replay_busy %.1f exposed %.1f gather %.1f halo_frac %.3f" % (p["compute"][0], p.get("replay_busy", [0])[0], p["replay_exposed"][0],
                                                                                      p["gather_wait"][0] + p["gather_payload"][0], sb["halo_frac"])
    tn = sb.get("tracker_native_ms_per_step")
    if tn:
        s += "  trk[launch %.1f wait %.1f decide %.1f acc %.1f | %d upd]" % (tn["counts_launch"], tn["counts_wait"], tn["decision"], tn["accumulate_launch"], tn["updates_per_step"])
    s += "  rounds %s" % d["config"]["parallelism"].split("-frame chunks")[0].split(";")[-1].strip()
cs = d.get("clip_stage")
if cs:
    s += "  clip_stage %.1f ms frac %.3f" % (cs["ms_per_step"], cs["frac"])
print(s + "   [%s %s]" % (envs, args), flush=True)
P
done
