"""The fp32 MFMA GEMM of this repo against the vendor library's fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS) on the bench's
largest shapes: is there headroom left in the kernel, or is ~0.65-0.8 of the 157 TF fp32-matrix peak what gfx950 sustains?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
torch.backends.cuda.matmul.allow_tf32 = False
for name, M, N, K in (("enc_ffn1", 204000, 1024, 256), ("enc_ffn2", 204000, 256, 1024), ("enc_qkv", 204000, 640, 256), ("enc_out", 204000, 256, 256),
                      ("dec_vals", 204000, 3072, 256), ("res4_1x1", 38400, 1024, 256), ("dec_29008_1024", 29008, 1024, 256)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    t_ours = time_ms(lambda: ops.linear(x, w, None, out=out), iters=20, warm=5)
    wt = w.t().contiguous()
    t_lib = time_ms(lambda: torch.mm(x, wt, out=out), iters=20, warm=5)
    t_lib2 = time_ms(lambda: torch.nn.functional.linear(x, w), iters=20, warm=5)
    fl = 2.0 * M * N * K
    print("%-16s M=%6d N=%5d K=%5d  ours %.3f ms = %.1f TF | torch.mm %.3f ms = %.1f TF | F.linear %.3f ms = %.1f TF" % (
        name, M, N, K, t_ours, fl / t_ours / 1e9, t_lib, fl / t_lib / 1e9, t_lib2, fl / t_lib2 / 1e9), flush=True)
