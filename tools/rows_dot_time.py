import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms
for (M, N) in ((31360, 4), (31360, 1), (7840, 4), (21168, 4)):
    x = torch.randn(M, 256, device="cuda"); w = torch.randn(N, 256, device="cuda") / 16; b = torch.randn(N, device="cuda")
    t = []
    for v in (0, 1):
        lib.mdqe_debug_gemm_rows_dot(v)
        t.append(1e3 * time_ms(lambda: ops.linear(x, w, b), iters=50, warm=5))
    print("M=%d N=%d K=256: MFMA tiles %.1f us | rows_dot %.1f us" % (M, N, t[0], t[1]))
