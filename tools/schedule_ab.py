"""Same-process alternation of the schedule knobs of MDQE.forward on the bench video: frames per pass, the last pass's size, look-ahead.
python tools/schedule_ab.py [rounds]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
bench.calibrate_synthetic_scores(model, sd, cfg, 360, 640)
video = bench.synth_video(0, 120, seed=0).pin_memory()
inp = [{"image": list(video), "height": 360, "width": 640}]


def run(k=6):
    with torch.no_grad():
        model(inp); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            model(inp)
        torch.cuda.synchronize()
    return 120 * k / (time.perf_counter() - t0)


variants = [dict(), dict(taper_tail=10), dict(taper_tail=14), dict(frame_batch=30), dict(frame_batch=48), dict(lookahead=3), dict(lookahead=1),
            dict(frame_batch=32, taper_tail=12)]
base = dict(taper_tail=model.taper_tail, frame_batch=model.frame_batch, lookahead=model.lookahead)
for r in range(rounds):
    for v in variants:
        for k, val in dict(base, **v).items():
            setattr(model, k, val)
        b = model.pass_bounds(120, model.frame_batch if model.frame_batch > 0 else 40, model.taper_passes, model.taper_tail)
        print("%-40s passes %-22s %.1f frames/s" % (v or "default", b, run()), flush=True)
