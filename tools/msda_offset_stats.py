"""How far do the encoder's deformable samples land from their query?  Captures the sampling offsets of every encoder layer of the bench
model on bench frames and prints, per level, the share of samples whose row distance to the query's own row (on that level) is within
+-k rows -- what a row band of a fine level staged in LDS per run of queries would catch.   python tools/msda_offset_stats.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
from mdqe_cvpr2023_amd import ops
cfg = PRESETS["R50_ovis_360"]
for init in ("workload", "reference"):
    sd = random_state(cfg, seed=0, remove_zero_init_trap=(init == "workload"))
    model = MDQE(cfg, state_dict=sd).eval()
    eng = model.engine
    frames = bench.synth_video(0, 4, seed=0).cuda()
    geo = eng.geometry(360, 640)
    caps = []
    orig = ops.msda_fused

    def cap(value, offs, logits, ref, levels, B, Q, M, D, L, P, mode=0, **kw):
        if mode == 0:
            caps.append((offs.clone(), ref.clone(), levels))
        return orig(value, offs, logits, ref, levels, B, Q, M, D, L, P, mode=mode, **kw)
    ops.msda_fused = cap
    with torch.no_grad():
        eng.encode(eng.backbone(frames, geo), geo)
    ops.msda_fused = orig
    for li, (offs, ref, levels) in enumerate(caps):
        Hs, Ws, Ss = levels
        N = ref.shape[0]
        o = offs.view(-1, N, 8, 4, 4, 2)                       # [frames, query, head, level, point, xy]
        line = "init %-9s layer %d:" % (init, li)
        for l in range(4):
            dy = (o[..., l, :, 1] / 8.0 * Hs[l]).abs()         # rows on level l between the sample and the query's reference row
            line += "  L%d(%2d rows) |dy|<=2: %4.1f%% <=4: %4.1f%% <=6: %4.1f%% <=8: %4.1f%%" % (
                l, Hs[l], 100 * float((dy <= 2).float().mean()), 100 * float((dy <= 4).float().mean()), 100 * float((dy <= 6).float().mean()),
                100 * float((dy <= 8).float().mean()))
        print(line, flush=True)
