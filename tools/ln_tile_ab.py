"""linear + residual + LayerNorm: the one-kernel form (64x256 tile, statistics in the epilogue) against GEMM + LayerNorm kernel, on the
decoder's row counts (instance level 7252 / 3332, box level 13328 / 21168 / 29008) -- where should ops.LINEAR_LN_MIN_ROWS sit?  The two
forms give identical bits since round 3 (tests/test_bench_shapes_gpu.py), so the threshold is a pure timing question.
python tools/ln_tile_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops


def time_us(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


g = torch.Generator(device="cuda").manual_seed(0)
for M in (3332, 7252, 13328, 21168, 29008, 40000):
    for K in (256, 1024):
        h = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(256, K, device="cuda", generator=g) / K ** 0.5
        b = torch.randn(256, device="cuda", generator=g); res = torch.randn(M, 256, device="cuda", generator=g)
        gam = torch.rand(256, device="cuda", generator=g) + 0.5; bet = torch.randn(256, device="cuda", generator=g)
        out = torch.empty_like(res); scratch = torch.empty_like(res)
        t = {}
        for name, thr in (("fused", 0), ("gemm+ln", 1 << 30)):
            ops.LINEAR_LN_MIN_ROWS = thr
            t[name] = time_us(lambda: ops.linear_ln(h, w, b, res, gam, bet, out=out, scratch=scratch))
        # with another kernel between the launches (no lockstep artefact): a 30-MB copy
        junk = torch.empty(8 << 20, device="cuda"); junk2 = torch.empty_like(junk)
        ti = {}
        for name, thr in (("fused", 0), ("gemm+ln", 1 << 30)):
            ops.LINEAR_LN_MIN_ROWS = thr
            base = time_us(lambda: junk2.copy_(junk))
            ti[name] = time_us(lambda: (ops.linear_ln(h, w, b, res, gam, bet, out=out, scratch=scratch), junk2.copy_(junk))) - base
        print("M=%6d K=%4d   back to back: fused %6.1f us  gemm+ln %6.1f us   interleaved: fused %6.1f  gemm+ln %6.1f" % (M, K, t["fused"], t["gemm+ln"], ti["fused"], ti["gemm+ln"]), flush=True)
