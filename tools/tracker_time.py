"""Tracker replay cost per clip on the bench workload (the serial part of the multi-GPU schedule)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE, ClipMerger
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
print("bias shift", calibrate_synthetic_scores(model, sd, cfg, 360, 640))
L = 120
video = synth_video(0, L, seed=0).cuda()
with torch.no_grad():
    clips = model.clip_schedule(L, cfg.n_frames_test, cfg.clip_stride)
    for rep in range(2):
        res = list(model.iter_clip_results(video, clips, 0))
        torch.cuda.synchronize()
        geo = model.engine.geometry(360, 640)
        ms = cfg.match_stride
        m = ClipMerger(model, (360, 640), (360, 640), (geo.Hp // ms, geo.Wp // ms))
        t0 = time.perf_counter()
        for item in res:
            m.feed(*item)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = m.finish()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n_inst = [len(r[3]["scores"]) for r in res]
        print("rep %d: %d clips, instances/clip mean %.1f max %d; tracker replay %.2f ms total = %.3f ms/clip; finish (video merge) %.2f ms; tracked instances %d"
              % (rep, len(res), sum(n_inst) / len(n_inst), max(n_inst), 1e3 * (t1 - t0), 1e3 * (t1 - t0) / len(res), 1e3 * (t2 - t1), m.tracker.num_inst))
