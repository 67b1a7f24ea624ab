"""Workload for rocprofv3: the per-frame stages only (one 40-frame pass at 360p), x N reps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
model = MDQE(cfg, state_dict=random_state(cfg, seed=0)).eval()
eng = model.engine
video = synth_video(0, 40, seed=0).cuda()
with torch.no_grad():
    geo = eng.geometry(360, 640)
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = model._frame_cache(video, geo)
        torch.cuda.synchronize()
        print("frame pass (40 frames) %.2f ms" % (1e3 * (time.perf_counter() - t0)))
        del c
