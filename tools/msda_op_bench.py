"""Timing of the native op ms_deform_attn_forward (reference ABI) on the R50_ovis_360 encoder shape, 40 batch elements: the
LDS-staged form (msda_fwd_v3_kernel) against the gather form (msda_fwd_v2_kernel).  python tools/msda_op_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
from mdqe_cvpr2023_amd._lib import lib
g = torch.Generator().manual_seed(0)
for name, B, shapes in (("360p", 40, [(48, 80), (24, 40), (12, 20), (6, 10)]), ("640p", 20, [(80, 144), (40, 72), (20, 36), (10, 18)])):
    M, D, L, P = 8, 32, 4, 4
    S = sum(h * w for h, w in shapes)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    sh, st = torch.tensor(shapes, dtype=torch.int64).cuda(), torch.tensor(starts, dtype=torch.int64).cuda()
    v = torch.randn(B, S, M, D, generator=g).cuda()
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(a) + 0.5) / a, (torch.arange(c) + 0.5) / c, indexing="ij"), -1).reshape(-1, 2).flip(-1) for a, c in shapes])
    loc = (ref[None, :, None, None, None, :] + torch.randn(B, S, M, L, P, 2, generator=g) / 8).cuda()          # as tools/pmc_msda.py: N(0,1) image-eighths
    at = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P).cuda()
    comp = (2 * v.numel() + loc.numel() + at.numel()) * 4.0                    # value + output + locations + weights
    outs = []
    for staged in (0, 1):
        lib.mdqe_debug_msda_op_staged(staged)
        for _ in range(3):
            o = MSDA.ms_deform_attn_forward(v, sh, st, loc, at, 64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o = MSDA.ms_deform_attn_forward(v, sh, st, loc, at, 64)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        outs.append(o)
        print("%s B=%d S=Q=%d: %s %.1f us = %.2f TB/s of the %.0f MB compulsory bytes" % (name, B, S, "coarse levels in LDS (v3)" if staged else "gather form (v2)      ", us, comp / us / 1e6, comp / 1e6), flush=True)
    print("   equal bits:", bool(torch.equal(outs[0], outs[1])))
lib.mdqe_debug_msda_op_staged(1)
