"""Per-kernel micro-benchmarks on one MI355X (HIP events on the launch stream).

    python tools/kbench.py msda            # encoder-shaped MSDA at R50_ovis_360 sizes
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_ms(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_msda(args):
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    res = []
    for name, B, shapes, Q in (("enc360_B4", 4, [(48, 80), (24, 40), (12, 20), (6, 10)], None),
                               ("enc360_B30", 30, [(48, 80), (24, 40), (12, 20), (6, 10)], None),
                               ("dec360_B4_Q196", 4, [(48, 80), (24, 40), (12, 20), (6, 10)], 196),
                               ("enc640_B4", 4, [(80, 144), (40, 72), (20, 36), (10, 18)], None)):
        S = sum(h * w for h, w in shapes)
        Q = Q or S
        M, D, L, P = 8, 32, 4, 4
        g = torch.Generator().manual_seed(0)
        v = torch.randn(B, S, M, D, generator=g).cuda()
        ref = torch.rand(B, Q, 1, 1, 1, 2, generator=g)
        loc = (ref + 0.05 * torch.randn(B, Q, M, L, P, 2, generator=g)).cuda()
        at = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P).cuda()
        sh = torch.tensor(shapes).cuda()
        st = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1])).cuda()
        ms = time_ms(lambda: MSDA.ms_deform_attn_forward(v, sh, st, loc, at, 64))
        comp = (v.numel() + loc.numel() + at.numel() + B * Q * M * D) * 4
        gath = B * Q * M * L * P * 4 * D * 4
        res.append(dict(case=name, ms=ms, compulsory_MB=comp / 1e6, GBps_compulsory=comp / ms / 1e6,
                        gathered_MB=gath / 1e6, GBps_gathered=gath / ms / 1e6))
        print(json.dumps(res[-1]))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["msda", "gemm", "all"])
    a = ap.parse_args()
    if a.what in ("msda", "all"):
        bench_msda(a)
