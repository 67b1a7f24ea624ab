"""The native op's LDS-staged kernel (msda_fwd_v3_kernel: D = 32, 4 levels x 4 points) against msda_fwd_v2_kernel (equal bits) and against
the oracle on a slice, on random level tables (pyramids, unrelated levels, reversed pyramids), batch sizes and query counts.
python tools/fuzz_msda_op_staged.py [n]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import mdqe_oracle as O
import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
from mdqe_cvpr2023_amd._lib import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for it in range(n):
    g = torch.Generator().manual_seed(it)
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
    M, D, L, P = ri(1, 8), 32, 4, 4
    h0, w0 = ri(6, 70), ri(6, 100)
    kind = it % 4
    if kind == 3:
        shapes = [(ri(1, 24), ri(1, 24)) for _ in range(4)]
    else:
        shapes = [(max(1, -(-h0 // 2 ** l)), max(1, -(-w0 // 2 ** l))) for l in range(4)]
        if kind == 2:
            shapes = shapes[::-1]
    S = sum(a * b for a, b in shapes)
    if S < 64:
        continue
    starts = [0]
    for a, b in shapes[:-1]:
        starts.append(starts[-1] + a * b)
    B, Q = ri(1, 20), ri(1, 700)
    sh, st = torch.tensor(shapes, dtype=torch.int64).cuda(), torch.tensor(starts, dtype=torch.int64).cuda()
    v = torch.randn(B, S, M, D, generator=g).cuda()
    loc = 0.5 + 1.8 * (torch.rand(B, Q, M, L, P, 2, generator=g) - 0.5)
    loc[:, ::4] = torch.round(loc[:, ::4] * 8) / 8
    at = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P)
    outs = []
    for staged in (0, 1):
        lib.mdqe_debug_msda_op_staged(staged)
        outs.append(MSDA.ms_deform_attn_forward(v, sh, st, loc.cuda(), at.cuda(), 64))
    lib.mdqe_debug_msda_op_staged(1)
    nq = min(Q, 32)
    ref = O.msda_forward(v[:1].cpu(), shapes, starts, loc[:1, :nq], at[:1, :nq])
    d = float((outs[1][:1, :nq].cpu() - ref).abs().max())
    if not torch.equal(outs[0], outs[1]) or d > 2e-5 * max(1.0, float(ref.abs().max())):
        bad += 1
        print("case %d shapes %s B=%d Q=%d M=%d: equal bits %s, vs oracle %.2e" % (it, shapes, B, Q, M, bool(torch.equal(outs[0], outs[1])), d), flush=True)
print("staged op fuzz: %d cases, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
