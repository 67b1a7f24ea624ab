"""The decoder's box-level deformable launch at the 640p geometry (17 clips x 4 frames, 196 queries; levels 80x144 .. 10x18): how many
coarse levels should a (clip frame, head) block stage?  150 KB budget = levels 2 + 3 (115 KB, one block per CU), 60 KB = level 3 only
(23 KB), 4 KB = nothing staged (the gather form).  Same for the 360p geometry (38 KB for levels 2 + 3).  python tools/msda_dec_640p.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms
for name, shapes, Bc, T, D in (("640p", [(80, 144), (40, 72), (20, 36), (10, 18)], 17, 4, 32), ("360p", [(48, 80), (24, 40), (12, 20), (6, 10)], 37, 4, 32),
                              ("swinl 480p", [(60, 108), (30, 54), (15, 27), (8, 14)], 34, 2, 24)):
    Q, M, L, P = 196, 8, 4, 4
    F = Bc + T - 1
    N = sum(h * w for h, w in shapes)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    levels = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
    g = torch.Generator().manual_seed(0)
    BT = Bc * T
    vals = torch.randn(F * N, M * D, generator=g).cuda()
    pr = torch.randn(BT * Q, 3 * M * L * P, generator=g).cuda()
    boxes = torch.rand(BT, Q, 4, generator=g).cuda() * torch.tensor([1, 1, 0.3, 0.3]).cuda()
    grid = torch.randn(M * L * P * 2, generator=g).cuda()
    vidx = torch.tensor([[c + t for t in range(T)] for c in range(Bc)], dtype=torch.int32).reshape(-1).cuda()
    nq = 2 * M * L * P
    out = torch.empty(BT * Q, M * D, device="cuda")
    ref = None
    for kb in (150, 85, 60, 4):
        lib.mdqe_debug_msda_stage_kb(kb)
        ms = time_ms(lambda: ops.msda_fused(vals, pr[:, :nq], pr[:, nq:], boxes, levels, BT, Q, M, D, L, P, mode=1, grid=grid, v_brows=N, vidx=vidx, out=out), iters=30, warm=5)
        same = True if ref is None else bool(torch.equal(out, ref))
        ref = out.clone() if ref is None else ref
        print("%s box-level launch, %d clips: staging budget %3d KB: %.1f us  (same bits as the first: %s)" % (name, Bc, kb, 1e3 * ms, same), flush=True)
    lib.mdqe_debug_msda_stage_kb(150)
