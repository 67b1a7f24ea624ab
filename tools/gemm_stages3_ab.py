"""Round 4: two K-steps of look-ahead (3 LDS stages, 48 KB per 128x128 block: 3 blocks per CU) against one (2 stages, 32 KB: 4 blocks per CU)
on the large K = 256 products of a 40-frame pass -- FFN1 + GELU (N = 1024), the value / offset / logit projection (N = 640), the decoder
value cache (N = 3072).  Timed with another kernel between the launches (no lockstep artefact); bitwise-equal outputs.
python tools/gemm_stages3_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib


def time_us(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


g = torch.Generator(device="cuda").manual_seed(0)
junk = torch.empty(8 << 20, device="cuda"); junk2 = torch.empty_like(junk)
base = time_us(lambda: junk2.copy_(junk))
for M, N, K, act in ((204000, 1024, 256, "gelu"), (204000, 640, 256, None), (204000, 3072, 256, None), (204000, 256, 1024, None), (153600, 512, 128, "relu")):
    x = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g); out = torch.empty(M, N, device="cuda")
    res = {}
    for rep in range(2):
        for st in (2, 3):
            lib.mdqe_debug_gemm_stages(st if st != 2 else 0)
            back = time_us(lambda: ops.linear(x, w, b, act=act, out=out))
            inter = time_us(lambda: (ops.linear(x, w, b, act=act, out=out), junk2.copy_(junk))) - base
            res.setdefault(st, out.clone())
            fl = 2.0 * M * N * K
            print("M=%6d N=%4d K=%4d %-4s  %d stages: back to back %7.1f us = %5.1f TF   interleaved %7.1f us = %5.1f TF"
                  % (M, N, K, act or "", st, back, fl / back / 1e6, inter, fl / inter / 1e6), flush=True)
    print("   bitwise equal:", bool(torch.equal(res[2], res[3])), flush=True)
lib.mdqe_debug_gemm_stages(0)
