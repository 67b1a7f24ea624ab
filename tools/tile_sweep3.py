"""Full-row tiles (64x256, 128x256) against the regular ones on the encoder's N=256 GEMMs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
from mdqe_cvpr2023_amd._lib import check, cur_stream, lib, ptr
for (M, N, K) in ((204000, 256, 256), (204000, 256, 1024), (29008, 256, 256), (29008, 256, 1024)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    ref = None
    line = "M=%6d N=%4d K=%4d " % (M, N, K)
    for tile in (1, 2, 3, 4, 5):
        ms = time_ms(lambda: ops.linear(x, w, b, residual=r, out=out, tile=tile), iters=30, warm=5)
        if ref is None:
            ref = out.clone()
        line += " t%d %.1f us (%.1f TF)" % (tile, 1e3 * ms, 2.0 * M * N * K / ms / 1e9)
        assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4), tile
    ln = time_ms(lambda: ops.layernorm(out, b, b, out=r), iters=30, warm=5)
    print(line + "  | layernorm %.1f us" % (1e3 * ln))

print("linear + residual + LayerNorm: one kernel vs two")
for (M, K) in ((204000, 256), (204000, 1024), (29008, 256), (29008, 1024)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(256, K, device="cuda") / 16; b = torch.randn(256, device="cuda")
    r = torch.randn(M, 256, device="cuda"); y = torch.empty(M, 256, device="cuda"); o = torch.empty(M, 256, device="cuda")
    two = time_ms(lambda: ops.layernorm(ops.linear(x, w, b, residual=r, out=y), b, b, out=o), iters=30, warm=5)
    check(lib.mdqe_gemm_ln_f32(ptr(x), K, ptr(w), ptr(b), ptr(o), 256, M, 256, K, ptr(r), 256, ptr(b), ptr(b), 1e-5, cur_stream()), "ln")
    one = time_ms(lambda: check(lib.mdqe_gemm_ln_f32(ptr(x), K, ptr(w), ptr(b), ptr(o), 256, M, 256, K, ptr(r), 256, ptr(b), ptr(b), 1e-5, cur_stream()), "ln"), iters=30, warm=5)
    print("M=%6d K=%4d  two kernels %.1f us   fused %.1f us" % (M, K, 1e3 * two, 1e3 * one))
