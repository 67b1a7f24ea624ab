"""Workload for rocprofv3: the per-clip stage ALONE (decoder + inference_clip of one batch of 40 clips over a cached 43-frame pass), x reps.
    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 tools/clip_stage_profile.py [reps] [frames]
then tools/kernel_summary.py-style reading of the *_kernel_stats.csv.  Prints the wall time per batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
eng = model.engine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 43
T = cfg.n_frames_test
video = synth_video(0, NF, seed=0, h=360, w=640).cuda()
with torch.no_grad():
    geo = eng.geometry(360, 640)
    c = model._frame_cache(video, geo)
    starts = list(range(NF - T + 1))
    for rep in range(reps + 2):
        if rep == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = eng.decode_clips(c, starts, T, geo)
        eng.inference_clips(outs, c["mf"], starts, T)
    torch.cuda.synchronize()
    print("per batch of %d clips: %.3f ms" % (len(starts), 1e3 * (time.perf_counter() - t0) / reps), flush=True)
