"""2-stage vs 4-stage LDS pipeline of the K-step-16 fp32 GEMM on the decoder's small launches: bitwise-equal outputs (same fmaf
chains in the same K order), kernel time from HIP events over back-to-back launches AND, under rocprofv3 --kernel-trace, from
the trace.  python tools/gemm_stages_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms
shapes = [(5292, 256, 256), (7252, 256, 256), (7252, 384, 256), (7252, 512, 256), (7252, 1024, 256), (7252, 256, 1024), (784, 256, 256), (196, 256, 256),
          (21168, 256, 256), (29008, 256, 256), (29008, 384, 256), (777, 130, 48), (3000, 96, 200)]
g = torch.Generator().manual_seed(0)
for M, N, K in shapes:
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    outs, ts = [], []
    for st in (2, 4):
        lib.mdqe_debug_gemm_stages(st)
        outs.append(ops.linear(x, w, b, act="gelu", residual=res, tile=3))
        out = torch.empty(M, N, device="cuda")
        ts.append(time_ms(lambda: ops.linear(x, w, b, out=out, tile=3), iters=30, warm=5))
    lib.mdqe_debug_gemm_stages(0)
    ref = torch.nn.functional.gelu(x.double() @ w.double().t() + b.double()) + res.double()
    err = float((outs[1].double() - ref).abs().max() / ref.abs().max())
    print("M=%5d N=%4d K=%4d  2 stages %.1f us  4 stages %.1f us  (%.0f -> %.0f TF)  bitwise equal %s  rel err %.1e" % (
        M, N, K, 1e3 * ts[0], 1e3 * ts[1], 2.0 * M * N * K / ts[0] / 1e9, 2.0 * M * N * K / ts[1] / 1e9, torch.equal(outs[0], outs[1]), err), flush=True)
    assert torch.equal(outs[0], outs[1]) and err < 3e-6
