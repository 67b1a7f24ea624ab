"""Host time of MDQE.upload_frames inside consecutive model(inputs) calls, and the per-video wall time with host frames vs resident frames."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
video = synth_video(0, 120, seed=0).pin_memory()
frames = list(video)
res = video.cuda()
orig = model.upload_frames
ts = []
def timed(imgs):
    t0 = time.perf_counter(); r = orig(imgs); ts.append(1e3 * (time.perf_counter() - t0)); return r
model.upload_frames = timed
import gc
with torch.no_grad():
    for name, img in (("host", frames), ("resident", res), ("host gc.freeze", frames), ("host gc.disable", frames), ("host", frames)):
        if "freeze" in name:
            gc.collect(); gc.freeze()
        if "disable" in name:
            gc.disable()
        elif name == "host":
            gc.enable()
        inp = [{"image": img, "height": 360, "width": 640}]
        for _ in range(3):
            model(inp)
        torch.cuda.synchronize(); ts.clear(); per = []
        for _ in range(10):
            t0 = time.perf_counter(); model(inp); per.append(1e3 * (time.perf_counter() - t0))
        print("%-16s per video %s ms (mean %.2f) | upload_frames host time %s ms" % (name, " ".join("%.1f" % v for v in per), sum(per) / len(per), " ".join("%.2f" % v for v in ts)), flush=True)
