"""Workload for tools/pmc_gemm_epilogue.sh: FFN1's shape (M = 204000, N = 1024, K = 256, + GELU) and the 3072-wide decoder-value product on the
128 x 128 fp32 kernel, 5 launches each with the few-instruction epilogue (default) -- or, with FAST_EPI=0, the general one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
lib.mdqe_debug_gemm_fast_epilogue(int(os.environ.get("FAST_EPI", "1")))
ops.set_gemm_precision("f32")
for (M, N, K, act) in ((204000, 1024, 256, "gelu"), (204000, 3072 if os.environ.get("WIDE") else 1024, 256, None)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(5):
        ops.linear(x, w, b, act=act, out=out, tile=1)
    torch.cuda.synchronize()
    del x, w, b, out
