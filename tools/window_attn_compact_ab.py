"""Round 4: Swin window attention with the position bias / shift mask in their generating form (2 KB + N bytes per block in LDS) against the
dense [nh, N, N] / [nW, N, N] tables (2 x 83 KB of L2 reads per (window, head) block at N = 144), on the four Swin-L stages of a 40-frame
480 x 864 pass; alternated, equal bits checked.   python tools/window_attn_compact_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd.engine import _swin_rel_tables, _swin_shift_mask, _swin_shift_regions
g = torch.Generator().manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def t_us(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (H, W, ws, nh) in ((120, 216, 12, 6), (60, 108, 12, 12), (30, 54, 12, 24), (15, 27, 6, 48)):
    N, C = ws * ws, nh * 32
    nWy, nWx = (H + ws - 1) // ws, (W + ws - 1) // ws
    nwin = B * nWy * nWx
    qkv = torch.randn(nwin * N, 3 * C, generator=g).cuda()
    scale = (torch.rand(nh, generator=g) * 10 + 1).cuda()
    _, idx = _swin_rel_tables(ws)
    rel = 16 * torch.sigmoid(torch.randn((2 * ws - 1) ** 2, nh, generator=g))
    bias = rel[idx.view(-1)].view(N, N, nh).permute(2, 0, 1).contiguous().cuda()
    rel_t = rel.t().contiguous().cuda()
    mask, region = _swin_shift_mask(H, W, ws).cuda(), _swin_shift_regions(H, W, ws).cuda()
    fl = 4.0 * nwin * nh * N * N * 32
    for shifted in (False, True):
        d = lambda: ops.window_attn(qkv, nwin, N, C, nh, scale, bias, mask if shifted else None, nWy * nWx)
        c = lambda: ops.window_attn_compact(qkv, nwin, ws, C, nh, scale, rel_t, region if shifted else None, nWy * nWx)
        same = bool(torch.equal(d(), c()))
        for rep in range(2):
            td, tc = t_us(d), t_us(c)
            print("%3dx%3d ws %2d heads %2d %s: dense %7.1f us = %5.1f TF   compact %7.1f us = %5.1f TF   equal bits %s"
                  % (H, W, ws, nh, "shifted" if shifted else "plain  ", td, fl / td / 1e6, tc, fl / tc / 1e6, same), flush=True)
