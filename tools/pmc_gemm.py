"""Workload for the PMC passes: the dominant GEMM shape of the bench (encoder FFN1 of a 30-frame chunk) x 10 launches
per precision mode (fp32 MFMA kernel, then f16x3 kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
M, N, K = 153000, 1024, 256
if len(sys.argv) > 3:
    M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(M, K, device="cuda"); w = ops.const_weight(torch.randn(N, K, device="cuda") / 16); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for mode in ("f32", "f16x3"):
    ops.set_gemm_precision(mode)
    for _ in range(10):
        ops.linear(x, w, b, act="gelu", out=out, tile=1)
torch.cuda.synchronize()
print("algorithmic bytes per launch: A %.1f MB + W %.1f MB + C %.1f MB" % (M * K * 4 / 1e6, N * K * 4 / 1e6, M * N * 4 / 1e6))
