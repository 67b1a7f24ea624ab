"""Body of __graft_entry__.smoke(): tiny hot-path invocation on cuda:0 checked against the oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run_smoke():
    import mdqe_oracle as O
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    g = torch.Generator().manual_seed(0)
    shapes = [(12, 20), (6, 10), (3, 5), (2, 3)]
    starts = [0, 240, 300, 315]
    S = 321
    B, M, D, L, P, Q = 2, 8, 32, 4, 4, 50
    v = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Q, M, L, P, 2, generator=g) * 1.2 - 0.1
    at = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P)
    out = MSDA.ms_deform_attn_forward(v.cuda(), torch.tensor(shapes).cuda(), torch.tensor(starts).cuda(),
                                      loc.cuda(), at.cuda(), 64).cpu()
    ref = O.msda_forward(v, shapes, starts, loc, at)
    err = float((out - ref).abs().max())
    assert err < 1e-5, err
    print(f"smoke ok: msda max|diff| = {err:.2e}")
