"""Is the fused encoder MSDA bound by the texture path (bytes through L1) or by L1 misses?  Same launch, three sample
patterns: offsets like a trained/random model (several pixels), zero offsets (every head of a query samples the query's
own location), one location for everything (all L1 hits)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
B, M, D, L, P = 40, 8, 32, 4, 4
shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
N = sum(h * w for h, w in shapes)
starts = [0]
for h, w in shapes[:-1]:
    starts.append(starts[-1] + h * w)
levels = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
g = torch.Generator().manual_seed(0)
proj = torch.randn(B * N, 256 + 3 * M * L * P, generator=g).cuda()
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                 for h, w in shapes]).float().cuda().contiguous()
out = torch.empty(B * N, 256, device="cuda")
nq = 2 * M * L * P
for name, scale, same in (("offsets ~ N(0, 2 px)", 2.0, False), ("offsets ~ N(0, 0.3 px)", 0.3, False), ("offsets 0", 0.0, False), ("one location", 0.0, True)):
    proj[:, 256:256 + nq].normal_(0, 1, generator=None).mul_(scale)
    r = ref.clone()
    if same:
        r[:] = 0.5
    ms = time_ms(lambda: ops.msda_fused(proj[:, :256], proj[:, 256:256 + nq], proj[:, 256 + nq:], r, levels, B, N, M, D, L, P, mode=0, v_brows=N, out=out),
                 iters=20, warm=3)
    gb = B * N * M * L * P * 4 * 128 / 1e9
    print("%-24s %.1f us   (%.1f TB/s through L1)" % (name, 1e3 * ms, gb / ms))
