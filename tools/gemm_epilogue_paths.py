"""The K-step-16 GEMM's two epilogues -- the few-instruction path of interior tiles and the general one -- must give the SAME BITS:
the tiles of one launch mix them (edge tiles, waves with masked rows), so a difference would make a frame's bits depend on the pass size.
Every form the fast path takes, fast path on against off (mdqe_debug_gemm_fast_epilogue), torch.equal.   python tools/gemm_epilogue_paths.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
g = torch.Generator().manual_seed(3)
bad = 0
for (M, N, K) in ((2560, 256, 64), (12800, 1024, 256), (25600, 640, 256), (29008, 768, 256)):
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / 8).cuda(); b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda(); pos = torch.randn(640, N, generator=g).cuda()
    side = torch.rand(M, 4, generator=g).cuda(); side_w = torch.randn(N, 4, generator=g).cuda()
    rm = (torch.rand(M, generator=g) < 0.05).cuda()
    cases = {"none": {}, "relu": dict(act="relu"), "gelu": dict(act="gelu"), "res after": dict(residual=r), "relu res first": dict(act="relu", residual=r, res_first=True),
             "relu res after": dict(act="relu", residual=r), "periodic res + rowmask": dict(residual=pos, res_mod=640, rowmask=rm, mask_cols=256),
             "rowmask all cols": dict(rowmask=rm, mask_cols=N)}
    for tile in (0, 1, 2, 3):
        for name, kw in cases.items():
            outs = []
            for fast in (1, 0):
                lib.mdqe_debug_gemm_fast_epilogue(fast)
                outs.append(ops.linear(x, w, b, tile=tile, **kw).clone())
            lib.mdqe_debug_gemm_fast_epilogue(1)
            if not torch.equal(outs[0], outs[1]):
                bad += 1
                d = (outs[0] - outs[1]).abs()
                print("DIFF M=%d N=%d K=%d tile %d %-24s max %.3e  elements %d" % (M, N, K, tile, name, float(d.max()), int((d > 0).sum())), flush=True)
        outs = []
        for fast in (1, 0):
            lib.mdqe_debug_gemm_fast_epilogue(fast)
            outs.append(ops.linear_side(x, w, b, side, side_w, (N // 2) // 4 * 4).clone())
        lib.mdqe_debug_gemm_fast_epilogue(1)
        if not torch.equal(outs[0], outs[1]):
            bad += 1
            d = (outs[0] - outs[1]).abs()
            print("DIFF M=%d N=%d K=%d %-24s max %.3e  elements %d" % (M, N, K, "side term", float(d.max()), int((d > 0).sum())), flush=True)
# The CONV and CAT instantiations take the fast path too (ReLU with the res_first residual in every ResNet bottleneck; conv3 + projection
# shortcut as one product): 1x1 and 3x3 convolutions, stride 1 and 2, with / without the residual, and linear_cat2.
for (NI, H, W, Cin, Cout, k, stride) in ((4, 48, 80, 128, 512, 1, 1), (4, 48, 80, 128, 128, 3, 1), (6, 48, 80, 256, 256, 3, 2), (3, 96, 160, 64, 256, 1, 1)):
    x = torch.randn(NI, H, W, Cin, generator=g).cuda(); w = (torch.randn(Cout, k, k, Cin, generator=g) / (k * Cin ** 0.5)).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    pad = k // 2
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = torch.randn(NI, OH, OW, Cout, generator=g).cuda()
    for name, kw in {"conv none": {}, "conv relu": dict(act="relu"), "conv relu res first": dict(act="relu", residual=r, res_first=True),
                     "conv res after": dict(residual=r)}.items():
        for tile in (0, 1, 2, 3):
            outs = []
            for fast in (1, 0):
                lib.mdqe_debug_gemm_fast_epilogue(fast)
                outs.append(ops.conv2d_nhwc(x, w, b, stride, pad, tile=tile, **kw).clone())
            lib.mdqe_debug_gemm_fast_epilogue(1)
            if not torch.equal(outs[0], outs[1]):
                bad += 1
                d = (outs[0] - outs[1]).abs()
                print("DIFF conv NI=%d %dx%d %d->%d k%d s%d tile %d %-22s max %.3e  elements %d" % (NI, H, W, Cin, Cout, k, stride, tile, name, float(d.max()), int((d > 0).sum())), flush=True)
for (NI, H2, W2, K2, K1, N, stride) in ((4, 96, 160, 64, 64, 256, 1), (4, 96, 160, 256, 128, 512, 2), (5, 24, 40, 1024, 512, 2048, 2)):
    OH, OW = (H2 - 1) // stride + 1, (W2 - 1) // stride + 1
    y = torch.randn(NI, OH, OW, K1, generator=g).cuda(); x = torch.randn(NI, H2, W2, K2, generator=g).cuda()
    w = (torch.randn(N, K1 + K2, generator=g) / (K1 + K2) ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    for act in (None, "relu"):
        outs = []
        for fast in (1, 0):
            lib.mdqe_debug_gemm_fast_epilogue(fast)
            outs.append(ops.linear_cat2(y, x, stride, w, b, act=act).clone())
        lib.mdqe_debug_gemm_fast_epilogue(1)
        if not torch.equal(outs[0], outs[1]):
            bad += 1
            d = (outs[0] - outs[1]).abs()
            print("DIFF cat2 NI=%d %dx%d K %d+%d N=%d s%d act %s max %.3e  elements %d" % (NI, H2, W2, K1, K2, N, stride, act, float(d.max()), int((d > 0).sum())), flush=True)
print("epilogue paths: %d differing cases" % bad)
sys.exit(1 if bad else 0)
