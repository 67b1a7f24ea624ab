"""Workload for tools/pmc_gemm_r05.sh: the round's two dominant fp32 GEMM kernel forms on the largest launch shapes of a 40-frame 360p pass,
5 launches each: (a) FFN1, M = 204000, N = 1024, K = 256, + GELU on the 128 x 128 tile (gemm_nt_f32_k16_kernel<128,128,...>); (b) FFN2 with the
LayerNorm epilogue, M = 204000, N = 256, K = 1024 (gemm_nt_f32_k16_kernel<64,256,...,true,...>: residual + LayerNorm in the epilogue, in place
over the residual as the encoder runs it); (c) the same with the second LayerNorm (mdqe_gemm_ln2_f32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
ops.set_gemm_precision("f32")
M = 204000
x = torch.randn(M, 256, device="cuda"); w1 = torch.randn(1024, 256, device="cuda") / 16; b1 = torch.randn(1024, device="cuda")
hid = torch.empty(M, 1024, device="cuda")
for _ in range(5):
    ops.linear(x, w1, b1, act="gelu", out=hid, tile=1)
torch.cuda.synchronize()
w2 = torch.randn(256, 1024, device="cuda") / 32; b2 = torch.randn(256, device="cuda")
g = torch.rand(256, device="cuda") + 0.5; be = torch.randn(256, device="cuda")
scratch = torch.empty(M, 256, device="cuda")
for _ in range(5):
    ops.linear_ln(hid, w2, b2, x, g, be, out=x, scratch=scratch)
torch.cuda.synchronize()
for _ in range(5):
    ops.linear_ln(hid, w2, b2, x, g, be, out=x, scratch=scratch, second=(g, be))
torch.cuda.synchronize()
