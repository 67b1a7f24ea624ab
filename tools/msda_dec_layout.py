"""Decoder MSDA (mode 1, 27 clips x 4 frames, 196 queries): value read as a 256-column slice of the [rows, 3072] cache row
(what the engine does today) against a contiguous [rows, 256] map."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
Bc, T, Q, M, D, L, P, F = 27, 4, 196, 8, 32, 4, 4, 30
shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
N = sum(h * w for h, w in shapes)
starts = [0]
for h, w in shapes[:-1]:
    starts.append(starts[-1] + h * w)
levels = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
g = torch.Generator().manual_seed(0)
BT = Bc * T
wide = torch.randn(F * N, 3072, generator=g).cuda()
narrow = wide[:, 512:768].contiguous()
pr = torch.randn(BT * Q, 3 * M * L * P, generator=g).cuda()
boxes = torch.rand(BT, Q, 4, generator=g).cuda() * torch.tensor([1, 1, 0.3, 0.3]).cuda()
grid = torch.randn(M * L * P * 2, generator=g).cuda()
vidx = torch.tensor([[c + t for t in range(T)] for c in range(Bc)], dtype=torch.int32).reshape(-1).cuda()
nq = 2 * M * L * P
out = torch.empty(BT * Q, 256, device="cuda")
for name, v in (("slice of [rows,3072]", wide[:, 512:768]), ("contiguous [rows,256]", narrow)):
    ms = time_ms(lambda: ops.msda_fused(v, pr[:, :nq], pr[:, nq:], boxes, levels, BT, Q, M, D, L, P, mode=1, grid=grid, v_brows=N, vidx=vidx, out=out), iters=30, warm=5)
    print("%-24s %.1f us" % (name, 1e3 * ms))
# round 2: the same launch with the two coarse levels of every (clip frame, head) staged in LDS (msda_fused_v3_kernel, mode 1) against v2
from mdqe_cvpr2023_amd._lib import lib
v = wide[:, 512:768]
outs = []
for var, name in ((0, "gather form (v2)"), (8, "coarse levels in LDS (v3)")):
    lib.mdqe_debug_msda_variant(var)
    ms = time_ms(lambda: ops.msda_fused(v, pr[:, :nq], pr[:, nq:], boxes, levels, BT, Q, M, D, L, P, mode=1, grid=grid, v_brows=N, vidx=vidx, out=out), iters=30, warm=5)
    outs.append(out.clone())
    print("%-28s %.1f us" % (name, 1e3 * ms))
lib.mdqe_debug_msda_variant(-1)
print("equal bits:", bool(torch.equal(outs[0], outs[1])))
