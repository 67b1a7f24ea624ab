"""(kernel form) x (tile) sweep for conv shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms
for name, NI, H, W, Cin, Cout, k, s, p_ in (("res2_3x3", 40, 96, 160, 64, 64, 3, 1, 1), ("res3_3x3", 40, 48, 80, 128, 128, 3, 1, 1),
                                            ("res4_3x3", 40, 24, 40, 256, 256, 3, 1, 1), ("res5_3x3", 40, 12, 20, 512, 512, 3, 1, 1),
                                            ("res3_3x3_s2", 40, 96, 160, 128, 128, 3, 2, 1), ("res4_3x3_s2", 40, 48, 80, 256, 256, 3, 2, 1),
                                            ("res5_3x3_s2", 40, 24, 40, 512, 512, 3, 2, 1),
                                            ("mh_3x3_256_l2", 40, 12, 20, 256, 256, 3, 1, 1), ("mh_3x3_256_l1", 40, 24, 40, 256, 256, 3, 1, 1),
                                            ("mh_3x3_256_l0", 40, 48, 80, 256, 256, 3, 1, 1), ("inproj_3x3_s2", 40, 12, 20, 2048, 256, 3, 2, 1)):
    x = torch.randn(NI, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    res = {}
    for v in (0, 1):
        lib.mdqe_debug_gemm_variant(v)
        for tile in (1, 2, 3):
            res[(v, tile)] = 1e3 * time_ms(lambda: ops.conv2d_nhwc(x, w, b, s, p_, act="relu", tile=tile), iters=20, warm=5)
    best = min(res, key=res.get)
    print("%-16s K=%5d N=%4d M=%7d | k32: %6.1f %6.1f %6.1f | k16: %6.1f %6.1f %6.1f | best form %s tile %d" % (
        name, k * k * Cin, Cout, NI * ((H + 2 * p_ - k) // s + 1) * ((W + 2 * p_ - k) // s + 1), res[(0, 1)], res[(0, 2)], res[(0, 3)], res[(1, 1)], res[(1, 2)], res[(1, 3)],
        "k16" if best[0] else "k32", best[1]))
lib.mdqe_debug_gemm_variant(2)
