"""What does the fp32 MFMA GEMM reach at ONE block per CU = one wave per SIMD?  That is the occupancy a fused FFN kernel would run at
(the 64-row activation tile 64 KB + the hidden chunk 32 KB + the weight stages 32 KB of LDS leave room for one 4-wave block per CU;
VERDICT r03 item 4a).  The production kernels are launched unchanged with extra dynamic LDS per block
(`mdqe_debug_gemm_lds_pad`), which caps the blocks a CU can hold: 4 (default: 32 KB per block), 3, 2, 1.
  * FFN2 + LayerNorm epilogue (64 x 256 tile, 4 waves: the tile shape a fused FFN would own), M = 204000, K = 1024
  * FFN1 + GELU (128 x 128 tile), M = 204000, N = 1024, K = 256
python tools/gemm_occupancy_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib


def time_us(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


g = torch.Generator(device="cuda").manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 204000
x = torch.randn(M, 256, device="cuda", generator=g)
hid = torch.randn(M, 1024, device="cuda", generator=g)
w1 = torch.randn(1024, 256, device="cuda", generator=g) / 16
b1 = torch.randn(1024, device="cuda", generator=g)
w2 = torch.randn(256, 1024, device="cuda", generator=g) / 32
b2 = torch.randn(256, device="cuda", generator=g)
gam = torch.rand(256, device="cuda", generator=g) + 0.5
bet = torch.randn(256, device="cuda", generator=g)
out = torch.empty(M, 256, device="cuda")
hout = torch.empty(M, 1024, device="cuda")
# blocks per CU by LDS: (32 KB + pad) per block in 160 KB
for label, pad in (("4 blocks/CU (default)", 0), ("3 blocks/CU", 12 * 1024), ("2 blocks/CU", 36 * 1024), ("1 block/CU", 64 * 1024)):
    lib.mdqe_debug_gemm_lds_pad(pad)
    t2 = time_us(lambda: ops.linear_ln(hid, w2, b2, x, gam, bet, out=out))
    t1 = time_us(lambda: ops.linear(x, w1, b1, act="gelu", out=hout))
    f = 2.0 * M * 256 * 1024
    print("%-22s FFN2+LN (64x256 tile, K=1024): %7.1f us = %6.1f TF   FFN1+GELU (128x128 tile, K=256): %7.1f us = %6.1f TF   pair %7.1f us = %6.1f TF"
          % (label, t2, f / t2 / 1e6, t1, f / t1 / 1e6, t1 + t2, 2 * f / (t1 + t2) / 1e6), flush=True)
lib.mdqe_debug_gemm_lds_pad(0)
