"""Differential fuzz of the native op `ms_deform_attn_forward` (the reference ABI, csrc/msda.hip) against the oracle's restatement of
ms_deform_attn_core_pytorch: random batch / level tables / heads / head widths / points, locations from far outside to inside the maps.
python tools/fuzz_msda.py [n]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import mdqe_oracle as O
import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for it in range(n):
    g = torch.Generator().manual_seed(it)
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
    B, M, D, L, P, Q = ri(1, 5), ri(1, 8), [4, 8, 16, 24, 32, 40, 64][it % 7], ri(1, 4), ri(1, 8), ri(1, 300)
    shapes = [(ri(1, 20), ri(1, 30)) for _ in range(L)]
    starts = [0]
    for hh, ww in shapes[:-1]:
        starts.append(starts[-1] + hh * ww)
    S = starts[-1] + shapes[-1][0] * shapes[-1][1]
    v = torch.randn(B, S, M, D, generator=g)
    spread = [0.2, 1.0, 3.0][it % 3]
    loc = 0.5 + spread * (torch.rand(B, Q, M, L, P, 2, generator=g) - 0.5)
    if it % 5 == 0:
        loc[..., 0] = torch.round(loc[..., 0] * 8) / 8                           # exactly on pixel borders / centres
    at = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P)
    out = MSDA.ms_deform_attn_forward(v.cuda(), torch.tensor(shapes, dtype=torch.int64).cuda(), torch.tensor(starts, dtype=torch.int64).cuda(),
                                      loc.cuda(), at.cuda(), 64).cpu()
    ref = O.msda_forward(v, shapes, starts, loc, at)
    d = float((out - ref).abs().max())
    if out.shape != ref.shape or d > 2e-5 * max(1.0, float(ref.abs().max())):
        bad += 1
        print("case %d B=%d S=%d M=%d D=%d L=%d P=%d Q=%d spread %.1f: max diff %.2e" % (it, B, S, M, D, L, P, Q, spread, d), flush=True)
print("msda fuzz: %d cases, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
