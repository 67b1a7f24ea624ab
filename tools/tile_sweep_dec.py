"""Tile shape for the decoder's GEMMs by measurement (fp32 K-step-16 kernel): rows = clips*T*196 (box level) or clips*196
(instance level), N in {256, 384, 512, 1024}, K in {256, 1024}; tiles 2 (128x64), 3 (64x64), 7 (32x64), 8 (32x128), 9 (64x128),
and split-K 2 on 64x64.  Prints TFLOP/s per (shape, tile)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
shapes = [(5292, 256, 256), (7252, 256, 256), (7252, 384, 256), (7252, 512, 256), (7252, 1024, 256), (7252, 256, 1024),
          (21168, 256, 256), (29008, 256, 256), (29008, 384, 256), (29008, 512, 256), (29008, 1024, 256), (29008, 4, 256), (29008, 1, 256)]
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    row = []
    for tile, ks in ((0, 0), (2, 0), (3, 0), (7, 0), (8, 0), (9, 0), (3, 2), (7, 2)):
        try:
            t = time_ms(lambda: ops.linear(x, w, b, out=out, tile=tile, ksplit=ks), iters=30, warm=5)
            row.append("t%d%s %.1fus %.0fTF" % (tile, "k2" if ks else "", 1e3 * t, 2.0 * M * N * K / t / 1e9))
        except Exception as e:
            row.append("t%d err" % tile)
    print("M=%5d N=%4d K=%4d | " % (M, N, K) + " | ".join(row), flush=True)
