"""Same-process alternation (round 4): the decoder of a group starts once ITS inputs of the group's last frame pass exist (query selection +
value projections; the mask-feature head, which only inference_clip reads, runs beside the decoder's first layers) against waiting for
the whole pass -- `model.early_decode`.  Same kernels, same inputs: the outputs are compared bit for bit.
python tools/early_decode_ab.py [config] [frames] [rounds]"""
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
video = bench.synth_video(0, frames, seed=0, h=fh, w=fw).pin_memory()
inp = [{"image": list(video), "height": fh, "width": fw}]


def run(k=8):
    with torch.no_grad():
        o = model(inp); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            model(inp)
        torch.cuda.synchronize()
    return frames * k / (time.perf_counter() - t0), o


outs = {}
for r in range(rounds):
    for early in (True, False):
        model.early_decode = early
        fps, o = run()
        outs[early] = o
        print("%s  early_decode=%d  %.1f frames/s" % (config, early, fps), flush=True)
ok, why = bench.same_output(outs[True], outs[False])
print("equal bits:", ok, why)
