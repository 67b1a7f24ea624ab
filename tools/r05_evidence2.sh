#!/bin/bash
# Round 5 evidence, second part (final code): the default bench line (with config_R50_ovis_720 / config_swinl_ovis / root_load / autocast_f16), the
# root-load rehearsal at N = 1 / 2 / 4 / 8, the replay profile, the D2H / pinned probes, the long fuzz runs.      bash tools/r05_evidence2.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
python bench.py > gpurun_out/r05/r05_bench_line_360p.json 2> gpurun_out/r05/r05_bench_line_360p.err; echo "bench rc=$?"
bash tools/root_load.sh 1 2 4 8
for w in 1 2 4 8; do cp gpurun_out/root_load_$w.json gpurun_out/r05/r05_root_load_$w.json; done
( python tools/replay_profile.py 8 1; python tools/replay_profile.py 8 0; python tools/replay_profile.py 1 1 ) 2>&1 | grep "^rep" > gpurun_out/r05/r05_replay_profile.txt
( python tools/d2h_probe.py; python tools/pinned_probe.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/r05_d2h_pinned_probe.txt
export PYTHONPATH=$(pwd):$(pwd)/oracle:$PYTHONPATH
python tools/fuzz_msda.py 400 2>&1 | tail -3 > gpurun_out/r05/r05_fuzz_msda.txt
python tools/fuzz_msda_fused.py 200 2>&1 | tail -3 > gpurun_out/r05/r05_fuzz_msda_fused.txt
python tools/fuzz_tracker.py 400 --gpu 2>&1 | tail -3 > gpurun_out/r05/r05_fuzz_tracker.txt
python tools/fuzz_pipeline.py 60 2>&1 | tail -3 > gpurun_out/r05/r05_fuzz_pipeline.txt
python tools/fuzz_inference_clip.py 100 2>&1 | tail -12 > gpurun_out/r05/r05_fuzz_inference_clip.txt
tail -2 gpurun_out/r05/r05_fuzz_*.txt
