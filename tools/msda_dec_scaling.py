"""How the decoder's two deformable launches scale with the batch (360p geometry, the pipeline's strides: value rows of 12 x 256 floats,
offsets | logits side by side): microseconds per launch for 5 .. 80 clips, staged (v3 / tp) and gather-only (v2) forms, + the encoder's
launch for 5 .. 40 frames.  Round 6: is a launch bound by block rounds (blocks / resident slots) or by throughput?
    python tools/msda_dec_scaling.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms

shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
Q, M, L, P, D, T, C = 196, 8, 4, 4, 32, 4, 256
N = sum(h * w for h, w in shapes)
starts = [0]
for h, w in shapes[:-1]:
    starts.append(starts[-1] + h * w)
g = torch.Generator().manual_seed(0)
for Bc in (5, 10, 20, 40, 80):
    F = Bc + T - 1
    BT = Bc * T
    vals = torch.randn(F * N, 12 * C, generator=g).cuda()
    boxes = (torch.rand(BT, Q, 4, generator=g) * torch.tensor([1, 1, 0.3, 0.3])).cuda()
    grid = torch.randn(M * L * P * 2, generator=g).cuda()
    line = "%2d clips:" % Bc
    # box level: one element per (clip, frame)
    pr = torch.randn(BT * Q, 3 * M * L * P, generator=g).cuda()
    vidx = torch.tensor([[c + t for t in range(T)] for c in range(Bc)], dtype=torch.int32).reshape(-1).cuda()
    nq = 2 * M * L * P
    lv = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
    out = torch.empty(BT * Q, C, device="cuda")
    for staged in (1, 0):
        lib.mdqe_debug_msda_dec_staged(staged)
        ms = time_ms(lambda: ops.msda_fused(vals[:, :C], pr[:, :nq], pr[:, nq:], boxes, lv, BT, Q, M, D, L, P, mode=1, grid=grid, v_brows=N, vidx=vidx, out=out), iters=30, warm=5)
        line += "  box %s %6.1f us" % ("v3" if staged else "v2", 1e3 * ms)
    lib.mdqe_debug_msda_dec_staged(1)
    # temporal: one element per clip, T frames x 4 levels
    pr2 = torch.randn(Bc * Q, 3 * M * T * P, generator=g).cuda()
    ibox = (torch.rand(Bc, Q, 4, generator=g) * torch.tensor([1, 1, 0.3, 0.3])).cuda()
    vidx2 = torch.arange(Bc, dtype=torch.int32).cuda()
    lv_tp = ([s[0] for s in shapes for _ in range(T)], [s[1] for s in shapes for _ in range(T)], [f * N + starts[gq] for gq in range(4) for f in range(T)])
    out2 = torch.empty(Bc * Q, C, device="cuda")
    for staged in (1, 0):
        lib.mdqe_debug_msda_tp_staged(staged)
        ms = time_ms(lambda: ops.msda_fused(vals[:, C:2 * C], pr2[:, :nq], pr2[:, nq:], ibox, lv_tp, Bc, Q, M, D, T, P, mode=1, grid=grid, groups=4, scale=0.25,
                                            v_brows=N, vidx=vidx2, out=out2), iters=30, warm=5)
        line += "  temporal %s %6.1f us" % ("tp" if staged else "v2", 1e3 * ms)
    lib.mdqe_debug_msda_tp_staged(1)
    print(line, flush=True)
    del vals, pr, pr2
for NI in (5, 10, 20, 40):
    proj = torch.randn(NI * N, 640, generator=g).cuda()
    proj[:, 256:512] *= 0.3
    ref = torch.rand(N, 2, generator=g).cuda()
    lv = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
    out = torch.empty(NI * N, C, device="cuda")
    ms = time_ms(lambda: ops.msda_fused(proj[:, :C], proj[:, C:C + 256], proj[:, C + 256:], ref, lv, NI, N, M, D, L, P, mode=0, v_brows=N, out=out), iters=20, warm=3)
    print("%2d frames: encoder v3 %6.1f us" % (NI, 1e3 * ms), flush=True)
