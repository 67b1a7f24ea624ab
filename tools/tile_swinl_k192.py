"""Tile shape for the K = 192 / 384 products of the Swin-L path (encoder FFN1 with hidden 192, Swin stage-1/2 projections): the dispatcher's
rule was measured on the R50 shapes (K = 256 ...).  Interleaved with a copy kernel (no lockstep artefact).  python tools/tile_swinl_k192.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops


def time_us(fn, iters=20, warm=3):
    junk = torch.empty(8 << 20, device="cuda"); junk2 = torch.empty_like(junk)
    for _ in range(warm):
        fn(); junk2.copy_(junk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        junk2.copy_(junk)
    e1.record(); torch.cuda.synchronize()
    base = e0.elapsed_time(e1)
    e0.record()
    for _ in range(iters):
        fn(); junk2.copy_(junk)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * (e0.elapsed_time(e1) - base) / iters


g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, act) in ((301595, 768, 192, "gelu"), (907200, 768, 192, None), (907200, 576, 192, None), (301595, 576, 192, None), (226800, 1536, 384, None),
                       (226800, 1152, 384, None), (301595, 192, 192, None), (907200, 192, 192, None), (301595, 2304, 192, None)):
    x = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5; b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda")
    line = "M=%7d N=%5d K=%4d %-5s" % (M, N, K, act or "")
    for tile, nm in ((0, "auto"), (1, "128x128"), (2, "128x64"), (3, "64x64"), (9, "64x128")):
        us = time_us(lambda: ops.linear(x, w, b, act=act, out=out, tile=tile))
        line += "  %s %7.1f us (%5.1f TF)" % (nm, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
    del x, w, out
