"""Times the pieces of the mask-feature head's output layers (40 frames at 360p): depthwise 5x5, its x2-upsampled form,
the pointwise 256 -> 8 product and the GroupNorms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
NI, H, W, C, Md = 40, 48, 80, 256, 32
g = torch.Generator().manual_seed(0)
x = torch.randn(NI, H, W, C, generator=g).cuda()
wt = (torch.randn(25, C, generator=g) / 5).cuda(); db = torch.randn(C, generator=g).cuda()
tw = torch.randn(C, generator=g).cuda(); tb = torch.randn(C, generator=g).cuda()
pw = (torch.randn(Md, C, generator=g) / 16).cuda(); pb = torch.randn(Md, generator=g).cuda()
pw2 = (torch.randn(C, C, generator=g) / 16).cuda(); pb2 = torch.randn(C, generator=g).cuda()
ga = torch.randn(C, generator=g).cuda()
print("dwconv5x5 [40,48,80,256]            %.1f us" % (1e3 * time_ms(lambda: ops.dwconv5x5(x, wt, db), iters=20, warm=3)))
print("pointwise 256->256 on it            %.1f us" % (1e3 * time_ms(lambda: ops.linear(x.view(-1, C), pw2, pb2), iters=20, warm=3)))
print("groupnorm(32)+relu on it            %.1f us" % (1e3 * time_ms(lambda: ops.groupnorm_nhwc(x, 32, ga, ga, act="relu"), iters=20, warm=3)))
z = ops.dwconv5x5(x, wt, db, up2=True, tw=tw, tb=tb)
print("dwconv5x5 up2 -> [40,96,160,256]   %.1f us" % (1e3 * time_ms(lambda: ops.dwconv5x5(x, wt, db, up2=True, tw=tw, tb=tb), iters=20, warm=3)))
print("pointwise 256->32 on it             %.1f us" % (1e3 * time_ms(lambda: ops.linear(z.view(-1, C), pw, pb), iters=20, warm=3)))
z2 = ops.linear(z.view(-1, C), pw, pb).view(NI, 2 * H, 2 * W, Md)
print("groupnorm(32ch)+relu                %.1f us" % (1e3 * time_ms(lambda: ops.groupnorm_nhwc(z2, 32, ga[:Md].contiguous(), ga[:Md].contiguous(), act="relu"), iters=20, warm=3)))
ops.DW_FAST = False
print("generic dwconv5x5 [40,48,80,256]    %.1f us" % (1e3 * time_ms(lambda: ops.dwconv5x5(x, wt, db), iters=20, warm=3)))
print("generic dwconv5x5 up2               %.1f us" % (1e3 * time_ms(lambda: ops.dwconv5x5(x, wt, db, up2=True, tw=tw, tb=tb), iters=20, warm=3)))
ops.DW_FAST = True
