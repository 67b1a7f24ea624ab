"""Does a power-of-two row stride of A hurt?  Same GEMM with lda = K and lda = K + pad."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
for mode in ("f16x3", "f32"):
    ops.set_gemm_precision(mode)
    for (M, N, K) in ((153000, 256, 1024), (153000, 256, 256), (153000, 1024, 256)):
        w = ops.const_weight(torch.randn(N, K, device="cuda") / 16); b = torch.randn(N, device="cuda")
        for pad in (0, 32, 64, 96):
            xf = torch.randn(M, K + pad, device="cuda")
            x = xf[:, :K]
            for opad in (0, 32):
                of = torch.empty(M, N + opad, device="cuda")
                ms = time_ms(lambda: ops.linear(x, w, b, out=of, ldc=N + opad, tile=1), iters=20, warm=5)
                print(mode, M, N, K, "lda", K + pad, "ldc", N + opad, "ms %.4f  TF %.1f" % (ms, 2.0 * M * N * K / ms / 1e9))
