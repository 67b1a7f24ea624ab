import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
video = synth_video(0, 60, seed=0).cuda()
with torch.no_grad():
    d = model([{"image": video, "height": 360, "width": 640}])
    model.rle_output = True
    torch.cuda.synchronize(); t0 = time.time()
    o = model([{"image": video, "height": 360, "width": 640}])
    torch.cuda.synchronize(); print("rle forward s", time.time() - t0)
lens = [len(r["counts"]) for inst in o["pred_rles"] for r in inst]
print("rle string length: mean %.0f max %d; dense mask fill %.3f" % (sum(lens) / len(lens), max(lens), float(torch.stack(d["pred_masks"]).float().mean())))
