"""Workload for rocprofv3: the per-clip stages only (decoder + inference_clip of one 30-frame chunk), x N reps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
from mdqe_cvpr2023_amd import ops
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
if len(sys.argv) > 1:
    ops.set_gemm_precision(sys.argv[1])
eng = model.engine
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 30           # frames in the cache: NF - 3 clips per decoder batch (40 -> the bench's 37)
video = synth_video(0, NF, seed=0).cuda()
with torch.no_grad():
    geo = eng.geometry(360, 640)
    c = model._frame_cache(video, geo)
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = eng.decode_clips(c, list(range(NF - 3)), 4, geo)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res = eng.inference_clips(outs, [c["mf"][i:i + 4] for i in range(NF - 3)])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("decode %.2f ms  inference_clips %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
