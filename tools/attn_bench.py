"""Kernel-level A/B of the attention cores: scalar form vs fp32-MFMA form."""
import os, sys, torch, functools
print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms
for (B, Q, C, nh) in ((148, 196, 256, 8), (108, 196, 256, 8), (27, 196, 256, 8), (40, 196, 192, 8)):
    qk = torch.randn(B * Q, 2 * C, device="cuda"); v = torch.randn(B * Q, C, device="cuda")
    t = []
    outs = []
    for var in (0, 2, 5, 3, 4, 6, 7, 8):
        lib.mdqe_debug_mha_variant(var)
        outs.append(ops.mha_small(qk, v, B, Q, C, nh))
        t.append(time_ms(lambda: ops.mha_small(qk, v, B, Q, C, nh), iters=30, warm=5))
    fl = 4.0 * B * nh * Q * Q * (C // nh)
    same = all(torch.equal(outs[3], x) for x in outs[5:])
    print("mha B=%d Q=%d D=%d: scalar %.1f us | mfma 3 waves %.1f | 4 waves %.1f us (%.1f TF) | 7 waves %.1f | 13 waves %.1f | V from global: 7 waves %.1f, 4 waves %.1f, 3 waves %.1f (bitwise equal to the LDS form: %s)" % (
        B, Q, C // nh, 1e3 * t[0], 1e3 * t[1], 1e3 * t[2], fl / t[2] / 1e9, 1e3 * t[3], 1e3 * t[4], 1e3 * t[5], 1e3 * t[6], 1e3 * t[7], same))
if len(sys.argv) > 1 and sys.argv[1] == "mha":
    sys.exit(0)
cases = ((7200, 144, 6), (1800, 144, 12), (600, 144, 24), (1200, 36, 48))
if len(sys.argv) > 1 and sys.argv[1] != "win":
    cases = [cases[int(sys.argv[1])]]
for (nwin, N, nh) in cases:
    print('case', nwin, N, nh)
    C = 32 * nh
    qkv = torch.randn(nwin * N, 3 * C, device="cuda"); sc = torch.rand(nh, device="cuda") * 10 + 1; bias = torch.randn(nh, N, N, device="cuda")
    t = []
    for var in (0, 2, 3, 4):
        lib.mdqe_debug_window_attn_variant(var)
        t.append(time_ms(lambda: ops.window_attn(qkv, nwin, N, C, nh, sc, bias, None, 1), iters=10, warm=3))
    fl = 4.0 * nwin * nh * N * N * 32
    print("window nwin=%d N=%d nh=%d: scalar %.1f us | mfma 3 waves %.1f us (%.1f TF) | 5 waves %.1f | 9 waves %.1f" % (
        nwin, N, nh, 1e3 * t[0], 1e3 * t[1], fl / t[1] / 1e9, 1e3 * t[2], 1e3 * t[3]))
lib.mdqe_debug_mha_variant(1); lib.mdqe_debug_window_attn_variant(1)
