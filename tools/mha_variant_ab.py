"""The 196-token decoder attention's variants (waves per (sequence, head); V from LDS or from global memory) INSIDE the pipeline, where its
60-KB blocks compete with the frame stream's GEMM blocks for LDS: same-process alternation on the bench video.  python tools/mha_variant_ab.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
from mdqe_cvpr2023_amd._lib import lib
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


names = {1: "7 waves, K/V in LDS (default)", 2: "3 waves", 5: "4 waves", 4: "13 waves", 6: "7 waves, V from global", 7: "4 waves, V from global"}
for r in range(3):
    for v in (1, 2, 5, 4, 6, 7):
        lib.mdqe_debug_mha_variant(v)
        print("mha variant %d (%-30s) %.1f frames/s" % (v, names[v], run()), flush=True)
lib.mdqe_debug_mha_variant(1)
