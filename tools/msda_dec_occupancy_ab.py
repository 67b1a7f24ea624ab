"""Round 6: the decoder's two deformable launches (360p geometry, the pipeline's strides) in register-capped builds: the shipped 832-thread
blocks take 72 (box level) / 82 (temporal) VGPRs = 7 / 5 waves per SIMD, so a CU holds two 13-wave blocks only just (box) or ONE (temporal);
knob 1 = both in the 8-waves-per-SIMD build (61 / 64 VGPRs, the temporal one spills 10), knob 2 = the temporal launch at 7 waves (72 VGPRs,
no spill).  Microseconds per launch, alternated, equal bits checked.     python tools/msda_dec_occupancy_ab.py"""
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
for Bc in (20, 40):
    F, BT = Bc + T - 1, Bc * T
    vals = torch.randn(F * N, 12 * C, generator=g).cuda()
    boxes = (torch.rand(BT, Q, 4, generator=g) * torch.tensor([1, 1, 0.3, 0.3])).cuda()
    grid = torch.randn(M * L * P * 2, generator=g).cuda()
    pr = torch.randn(BT * Q, 3 * M * L * P, generator=g).cuda()
    vidx = torch.tensor([[c + t for t in range(T)] for c in range(Bc)], dtype=torch.int32).reshape(-1).cuda()
    nq = 2 * M * L * P
    lv = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
    pr2 = torch.randn(Bc * Q, 3 * M * T * P, generator=g).cuda()
    ibox = (torch.rand(Bc, Q, 4, generator=g) * torch.tensor([1, 1, 0.3, 0.3])).cuda()
    vidx2 = torch.arange(Bc, dtype=torch.int32).cuda()
    lv_tp = ([s[0] for s in shapes for _ in range(T)], [s[1] for s in shapes for _ in range(T)], [f * N + starts[gq] for gq in range(4) for f in range(T)])
    out, out2 = torch.empty(BT * Q, C, device="cuda"), torch.empty(Bc * Q, C, device="cuda")
    box = lambda: ops.msda_fused(vals[:, :C], pr[:, :nq], pr[:, nq:], boxes, lv, BT, Q, M, D, L, P, mode=1, grid=grid, v_brows=N, vidx=vidx, out=out)
    tp = lambda: ops.msda_fused(vals[:, C:2 * C], pr2[:, :nq], pr2[:, nq:], ibox, lv_tp, Bc, Q, M, D, T, P, mode=1, grid=grid, groups=4, scale=0.25,
                                v_brows=N, vidx=vidx2, out=out2)
    ref = {}
    for rep in range(2):
        for knob in (0, 1, 2):
            lib.mdqe_debug_msda_dec_wpe8(knob)
            tb, tt = 1e3 * time_ms(box, iters=30, warm=5), 1e3 * time_ms(tp, iters=30, warm=5)
            same = True
            if knob == 0 and rep == 0:
                ref = {"b": out.clone(), "t": out2.clone()}
            else:
                same = bool(torch.equal(out, ref["b"]) and torch.equal(out2, ref["t"]))
            print("%2d clips, knob %d: box level %6.1f us   temporal %6.1f us   (same bits as knob 0: %s)" % (Bc, knob, tb, tt, same), flush=True)
    lib.mdqe_debug_msda_dec_wpe8(0)
