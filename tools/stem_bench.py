"""Stem A/B on one box: im2col + GEMM against the fused stem kernel (40 frames at 360p; 20 at 640p)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
mean, std = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
g = torch.Generator().manual_seed(0)
for (n, h, w, Hp, Wp) in ((40, 360, 640, 384, 640), (20, 640, 1138, 640, 1152)):
    fr = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8).cuda()
    wt = torch.randn(64, 7, 7, 3, generator=g) / 12
    b = torch.randn(64, generator=g).cuda()
    wp = torch.zeros(64, 160); wp[:, :147] = wt.reshape(64, 147); wp = wp.cuda()
    wk = ops.stem_weight_kmajor(wt).cuda()

    def old():
        col = ops.stem_im2col(fr, Hp, Wp, mean, std)
        return ops.linear(col, wp, b, act="relu")

    def new():
        return ops.stem_conv(fr, Hp, Wp, mean, std, wk, b)

    a, c = old().view(-1), new().view(-1)
    print("max |diff| %.3g (scale %.3g)" % (float((a - c).abs().max()), float(a.abs().max())))
    for name, fn in (("im2col+gemm", old), ("fused", new), ("im2col+gemm", old), ("fused", new)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print("%dx%dx%d %-12s %.3f ms  (%.1f TFLOP/s on 147-tap flops)" % (n, h, w, name, ms, n * (Hp // 2) * (Wp // 2) * 64 * 147 * 2 / ms / 1e9))
