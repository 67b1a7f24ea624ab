"""A/B of the start stagger between the blocks that share a CU (mdqe_debug_gemm_stagger, 10-ns ticks) on the encoder's GEMM shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms
ops.set_gemm_precision("f32")
vals = [int(v) for v in sys.argv[1:]] or [0, 300, 600, 1000, 1500, 2500]
for (M, N, K, act, ln) in ((204000, 1024, 256, "gelu", False), (204000, 640, 256, None, False), (204000, 3072, 256, None, False), (204000, 256, 256, None, True),
                           (204000, 256, 1024, None, True), (153600, 256, 2304 // 9, None, False), (38400, 1024, 256, "relu", False)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if ln else None
    g, be = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    row = []
    for st in vals:
        lib.mdqe_debug_gemm_stagger(st)
        f = (lambda: ops.linear_ln(x, w, b, res, g, be)) if ln else (lambda: ops.linear(x, w, b, act=act, out=out))
        ms = time_ms(f, iters=20, warm=5)
        row.append("%d: %.1f us %.1f TF" % (st, 1e3 * ms, 2.0 * M * N * K / ms / 1e9))
    print("M=%d N=%d K=%d %s%s | " % (M, N, K, act or "", " +LN" if ln else "") + " | ".join(row), flush=True)
lib.mdqe_debug_gemm_stagger(0)
