"""Round 6: does the decoder's box-level launch pay for its staging?  The same launch (360p, 40 clips) with each (clip frame, head)'s 196 queries
in 7 / 4 / 2 (shipped) / 1 blocks, i.e. the head's two coarse levels (38 KB) staged 7 / 4 / 2 / 1 times per 196 queries (variant bits of
mdqe_debug_msda_variant: query run of 32 / 64 / shipped 98 / 256).  Equal bits checked.     python tools/msda_dec_staging_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms

shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
Q, M, L, P, D, T, C, Bc = 196, 8, 4, 4, 32, 4, 256, 40
N = sum(h * w for h, w in shapes)
starts = [0]
for h, w in shapes[:-1]:
    starts.append(starts[-1] + h * w)
g = torch.Generator().manual_seed(0)
F, BT = Bc + T - 1, Bc * T
vals = torch.randn(F * N, 12 * C, generator=g).cuda()
boxes = (torch.rand(BT, Q, 4, generator=g) * torch.tensor([1, 1, 0.3, 0.3])).cuda()
grid = torch.randn(M * L * P * 2, generator=g).cuda()
pr = torch.randn(BT * Q, 3 * M * L * P, generator=g).cuda()
vidx = torch.tensor([[c + t for t in range(T)] for c in range(Bc)], dtype=torch.int32).reshape(-1).cuda()
nq = 2 * M * L * P
lv = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
out = torch.empty(BT * Q, C, device="cuda")
box = lambda: ops.msda_fused(vals[:, :C], pr[:, :nq], pr[:, nq:], boxes, lv, BT, Q, M, D, L, P, mode=1, grid=grid, v_brows=N, vidx=vidx, out=out)
ref = None
for rep in range(2):
    for name, var in (("shipped: 2 blocks x 98 queries", -1), ("7 blocks x 32 queries", 8 | (1 << 4)), ("4 blocks x 64 queries", 8 | (2 << 4)),
                      ("1 block x 196 queries (2 iterations, ONE staging)", 8 | (4 << 4))):
        lib.mdqe_debug_msda_variant(var)
        us = 1e3 * time_ms(box, iters=30, warm=5)
        if ref is None:
            ref = out.clone()
        print("%-52s %6.1f us   (same bits: %s)" % (name, us, bool(torch.equal(out, ref))), flush=True)
lib.mdqe_debug_msda_variant(-1)
