import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
ops.set_gemm_precision("f32")
for (M, N, K, act) in ((204000, 1024, 256, "gelu"), (204000, 640, 256, None), (204000, 3072, 256, None), (38400, 1024, 256, "relu")):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    y = torch.randn(M, 256, device="cuda"); wy = torch.randn(64, 256, device="cuda")
    row = []
    for tile in (1, 5, 2, 9, 4):
        def f():
            ops.linear(y, wy, None)            # a different kernel in between: no lockstep artefact
            ops.linear(x, w, b, act=act, out=out, tile=tile)
        t_pair = time_ms(f, iters=20, warm=5)
        t_other = time_ms(lambda: ops.linear(y, wy, None), iters=20, warm=5)
        ms = t_pair - t_other
        row.append("t%d %.1f us %.1f TF" % (tile, 1e3 * ms, 2.0 * M * N * K / ms / 1e9))
    print("M=%d N=%d K=%d %s | %s" % (M, N, K, act or "", " | ".join(row)), flush=True)
