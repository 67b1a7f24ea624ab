"""Workload for rocprofv3 --pmc passes over the MFMA kernels other than the plain GEMM: fused stem, 3x3 implicit-GEMM conv,
GEMM with LayerNorm epilogue, Swin window attention, 196-token decoder attention; and the fused encoder MSDA.  5 launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
g = torch.Generator().manual_seed(0)
mean, std = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
R = 5
# stem: 40 frames 360x640
fr = torch.randint(0, 256, (40, 3, 360, 640), generator=g, dtype=torch.uint8).cuda()
wk = ops.stem_weight_kmajor(torch.randn(64, 7, 7, 3, generator=g) / 12).cuda(); b64 = torch.randn(64, generator=g).cuda()
for _ in range(R):
    ops.stem_conv(fr, 384, 640, mean, std, wk, b64)
# res3 3x3 conv, 40 frames
x = torch.randn(40, 48, 80, 128, generator=g).cuda(); w = (torch.randn(128, 3, 3, 128, generator=g) * 0.05).cuda(); b = torch.randn(128, generator=g).cuda()
for _ in range(R):
    ops.conv2d_nhwc(x, w, b, 1, 1, act="relu")
# mask-head 3x3 conv 256->256 at level 0
x = torch.randn(40, 48, 80, 256, generator=g).cuda(); w = (torch.randn(256, 3, 3, 256, generator=g) * 0.03).cuda(); b = torch.randn(256, generator=g).cuda()
for _ in range(R):
    ops.conv2d_nhwc(x, w, b, 1, 1)
# FFN2 + residual + LayerNorm, M = 204000
M = 204000
h = torch.randn(M, 1024, generator=g).cuda(); w2 = (torch.randn(256, 1024, generator=g) / 32).cuda(); b2 = torch.randn(256, generator=g).cuda()
r = torch.randn(M, 256, generator=g).cuda(); o = torch.empty_like(r)
for _ in range(R):
    ops.linear_ln(h, w2, b2, r, b2, b2, out=o)
del h, r, o
# Swin window attention: stage-1 of Swin-L at 480x853 (40 frames): 7200 windows x 144 tokens, 6 heads
nwin, N, nh = 7200, 144, 6
C = 32 * nh
qkv = torch.randn(nwin * N, 3 * C, generator=g).cuda(); sc = (torch.rand(nh, generator=g) * 10 + 1).cuda(); bias = torch.randn(nh, N, N, generator=g).cuda()
for _ in range(R):
    ops.window_attn(qkv, nwin, N, C, nh, sc, bias, None, 1)
# decoder self-attention: 148 sequences x 196 tokens, 8 heads
B, Q = 148, 196
qk = torch.randn(B * Q, 512, generator=g).cuda(); v = torch.randn(B * Q, 256, generator=g).cuda()
for _ in range(R):
    ops.mha_small(qk, v, B, Q, 256, 8)
# fused encoder MSDA, 40 frames
Bf, Mh, D, L, P = 40, 8, 32, 4, 4
shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
Nq = sum(a * c for a, c in shapes)
starts = [0]
for a, c in shapes[:-1]:
    starts.append(starts[-1] + a * c)
levels = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
proj = torch.randn(Bf * Nq, 256 + 3 * Mh * L * P, generator=g).cuda()
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(a) + 0.5) / a, (torch.arange(c) + 0.5) / c, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                 for a, c in shapes]).float().cuda().contiguous()
out = torch.empty(Bf * Nq, 256, device="cuda")
nq = 2 * Mh * L * P
for _ in range(R):
    ops.msda_fused(proj[:, :256], proj[:, 256:256 + nq], proj[:, 256 + nq:], ref, levels, Bf, Nq, Mh, D, L, P, mode=0, v_brows=Nq, out=out)
torch.cuda.synchronize()
print("done")
