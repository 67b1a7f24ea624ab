"""Power / DVFS check: the same f16x3 GEMM on random vs zero operands (identical instruction stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
for mode in ("f16x3", "f32"):
    ops.set_gemm_precision(mode)
    for (M, N, K) in ((153000, 256, 1024), (153000, 1024, 256)):
        for kind in ("random", "zero", "small-int"):
            if kind == "random":
                x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16
            elif kind == "zero":
                x = torch.zeros(M, K, device="cuda"); w = torch.zeros(N, K, device="cuda")
            else:
                x = torch.randint(-2, 3, (M, K), device="cuda").float(); w = torch.randint(-2, 3, (N, K), device="cuda").float()
            w = ops.const_weight(w)
            out = torch.empty(M, N, device="cuda")
            ms = time_ms(lambda: ops.linear(x, w, None, out=out, tile=1), iters=30, warm=10)
            print(mode, M, N, K, kind, "ms %.4f  TF %.1f" % (ms, 2.0 * M * N * K / ms / 1e9))
