"""forward_stream against separate calls on the bench video, alternated in one process.  python tools/stream_mode_ab.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
bench.calibrate_synthetic_scores(model, sd, cfg, 360, 640)
video = bench.synth_video(0, 120, seed=0).pin_memory()
inp = [{"image": list(video), "height": 360, "width": 640}]
K = 8
with torch.no_grad():
    for r in range(3):
        model(inp); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            model(inp)
        torch.cuda.synchronize(); a = 120 * K / (time.perf_counter() - t0)
        for _ in model.forward_stream(inp for _ in range(2)):
            pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in model.forward_stream(inp for _ in range(K)):
            pass
        torch.cuda.synchronize(); b = 120 * K / (time.perf_counter() - t0)
        print("separate calls %.1f frames/s   forward_stream %.1f frames/s" % (a, b), flush=True)
