"""The LDS-staged fused MSDA kernel (msda_fused_v3_kernel: encoder form and decoder box-level form) against msda_fused_v2_kernel on
random level tables / batch sizes / query counts / head widths / value-row pitches: equal bits required.  python tools/fuzz_msda_fused.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for it in range(n):
    g = torch.Generator().manual_seed(it)
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
    M, D, L, P = 8, (32, 24)[it % 2], 4, 4
    C = M * D
    h0, w0 = ri(4, 60), ri(4, 90)
    if it % 5 == 4:                                                     # not a pyramid
        shapes = [(ri(1, 20), ri(1, 20)) for _ in range(4)]
    else:
        shapes = [(max(1, -(-h0 // 2 ** l)), max(1, -(-w0 // 2 ** l))) for l in range(4)]
    N = sum(a * b for a, b in shapes)
    starts = [0]
    for a, b in shapes[:-1]:
        starts.append(starts[-1] + a * b)
    levels = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
    nq = 2 * M * L * P
    enc = it % 3 != 0
    if enc:
        B, Q = ri(1, 20), N
        pitch = C + 3 * M * L * P + 4 * ri(0, 3)
        proj = (torch.randn(B * Q, pitch, generator=g) * torch.tensor([1.0] * C + [3.0] * nq + [1.0] * (pitch - C - nq))).cuda()
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(a) + 0.5) / a, (torch.arange(c) + 0.5) / c, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                         for a, c in shapes]).float().cuda().contiguous()
        call = lambda out: ops.msda_fused(proj[:, :C], proj[:, C:C + nq], proj[:, C + nq:C + nq + M * L * P], ref, levels, B, Q, M, D, L, P, mode=0, v_brows=N, out=out)
        variants = (1, 9)
    else:
        B, Q, F = ri(1, 24), ri(1, 300), ri(1, 6)
        wide = torch.randn(F * N, 3 * C, generator=g).cuda()
        pr = (2.0 * torch.randn(B * Q, 3 * M * L * P, generator=g)).cuda()
        boxes = (torch.rand(B, Q, 4, generator=g) * torch.tensor([1, 1, 0.6, 0.6])).cuda()
        grid = torch.randn(M * L * P * 2, generator=g).cuda()
        vidx = torch.randint(0, F, (B,), generator=g, dtype=torch.int32).cuda()
        call = lambda out: ops.msda_fused(wide[:, C:2 * C], pr[:, :nq], pr[:, nq:], boxes, levels, B, Q, M, D, L, P, mode=1, grid=grid, v_brows=N, vidx=vidx, out=out)
        variants = (0, 8)
    outs = []
    for var in variants:
        lib.mdqe_debug_msda_variant(var)
        out = torch.full((B * Q, C), float("nan"), device="cuda")
        call(out)
        outs.append(out)
    lib.mdqe_debug_msda_variant(-1)
    if not (torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])):
        bad += 1
        print("case %d %s D=%d shapes %s B=%d Q=%d: differs (max %.3e)" % (it, "enc" if enc else "dec", D, shapes, B, Q, float((outs[0] - outs[1]).abs().nan_to_num(1e9).max())), flush=True)
print("fused msda fuzz: %d cases, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
