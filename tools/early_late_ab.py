"""Same-process alternation: final masks per flushed window (early) against one pass at the end (late), per config; and what the
bench's per-launch GEMM event pairs cost inside the timed region.   python tools/early_late_ab.py [config] [frames] [rounds]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
config = sys.argv[1] if len(sys.argv) > 1 else "R50_ovis_360"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = PRESETS[config]
fh, fw = {"R50_ovis_360": (360, 640), "R50_ovis_720": (640, 1138), "swinl_ovis": (480, 853)}[config]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
bench.calibrate_synthetic_scores(model, sd, cfg, fh, fw)
meter = bench.GemmMeter(); meter.install()
video = bench.synth_video(0, frames, seed=0, h=fh, w=fw).pin_memory()
inp = [{"image": list(video), "height": fh, "width": fw}]


def run(k=6):
    with torch.no_grad():
        model(inp); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            model(inp)
        torch.cuda.synchronize()
    return frames * k / (time.perf_counter() - t0)


for r in range(rounds):
    for early in (True, False):
        for met in (False, True):
            model.early_masks = early
            meter.enabled = met; meter.rec = []
            print("%s  early_masks=%d  gemm event pairs=%d  %.1f frames/s" % (config, early, met, run()), flush=True)
meter.enabled = False
