"""In-process A/B of the two fp32 GEMM kernel forms (K-step 32 vs K-step 16): correctness vs fp64 and timing."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
from kbench import time_ms
g = torch.Generator().manual_seed(5)
# correctness (variant 1 vs fp64): ragged M/N/K, epilogue options, conv
lib.mdqe_debug_gemm_variant(1)
for (M, N, K, tile) in ((1000, 300, 136, 1), (777, 130, 48, 3), (20000, 640, 256, 1), (5000, 64, 160, 2), (300, 256, 4096, 3)):
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    res = torch.randn(100, N, generator=g) if M % 100 == 0 else None
    rm = torch.rand(M, generator=g) < 0.2
    ref = x.double() @ w.double().t() + b.double()
    ref[:, :N // 2] = F.gelu(ref[:, :N // 2])
    if res is not None:
        ref = ref + res.double().repeat(M // 100, 1)
    ref[:, :N // 3][rm] = 0
    out = ops.linear(x.cuda(), w.cuda(), b.cuda(), act="gelu", act_cols=N // 2, residual=None if res is None else res.cuda(),
                     res_mod=100 if res is not None else 0, rowmask=rm.cuda(), mask_cols=N // 3, tile=tile, ksplit=8 if K == 4096 else 0)
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    print("linear", M, N, K, tile, "rel err %.2e" % err)
    assert err < 3e-6
for (Cin, Cout, k, s, p_) in ((64, 64, 3, 1, 1), (256, 256, 3, 1, 1), (512, 512, 3, 2, 1), (1024, 256, 1, 1, 0), (32, 96, 3, 2, 1)):
    xi = torch.randn(3, Cin, 24, 40, generator=g); wc = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5; bc = torch.randn(Cout, generator=g)
    ref = F.conv2d(xi.double(), wc.double(), bc.double(), s, p_)
    rs = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + rs.double())
    out = ops.conv2d_nhwc(xi.permute(0, 2, 3, 1).contiguous().cuda(), wc.permute(0, 2, 3, 1).contiguous().cuda(), bc.cuda(), s, p_, act="relu",
                          residual=rs.permute(0, 2, 3, 1).contiguous().cuda(), res_first=True)
    err = float((out.cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max())
    print("conv", Cin, Cout, k, s, "rel err %.2e" % err)
    assert err < 3e-6
# timing
cases = [("enc_qkv_30f", 153000, 640, 256, None), ("enc_ffn1_30f", 153000, 1024, 256, None), ("enc_ffn1_gelu", 153000, 1024, 256, "gelu"),
         ("enc_ffn2_30f", 153000, 256, 1024, None), ("enc_out_30f", 153000, 256, 256, None), ("dec_21168_256", 21168, 256, 256, None),
         ("dec_21168_1024", 21168, 1024, 256, None), ("dec_5292_256", 5292, 256, 256, None)]
convs = [("res2_3x3", 30, 96, 160, 64, 64, 3, 1, 1), ("res3_3x3", 30, 48, 80, 128, 128, 3, 1, 1), ("res4_3x3", 30, 24, 40, 256, 256, 3, 1, 1),
         ("res5_3x3", 30, 12, 20, 512, 512, 3, 1, 1), ("res4_1x1", 30, 24, 40, 1024, 256, 1, 1, 0), ("res2_1x1_64_256", 30, 96, 160, 64, 256, 1, 1, 0)]
for rep in range(2):
    for name, M, N, K, act in cases:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        t = []
        for v in (0, 1):
            lib.mdqe_debug_gemm_variant(v)
            t.append(time_ms(lambda: ops.linear(x, w, b, out=out, act=act), iters=20, warm=5))
        print("%-18s k32 %.4f ms (%.1f TF)  k16 %.4f ms (%.1f TF)  speedup %.3f" % (name, t[0], 2.0 * M * N * K / t[0] / 1e9, t[1], 2.0 * M * N * K / t[1] / 1e9, t[0] / t[1]))
    for name, NI, H, W, Cin, Cout, k, s, p_ in convs:
        x = torch.randn(NI, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
        OH, OW = (H + 2 * p_ - k) // s + 1, (W + 2 * p_ - k) // s + 1
        fl = 2.0 * NI * OH * OW * Cout * Cin * k * k
        t = []
        for v in (0, 1):
            lib.mdqe_debug_gemm_variant(v)
            t.append(time_ms(lambda: ops.conv2d_nhwc(x, w, b, s, p_, act="relu"), iters=20, warm=5))
        print("%-18s k32 %.4f ms (%.1f TF)  k16 %.4f ms (%.1f TF)  speedup %.3f" % (name, t[0], fl / t[0] / 1e9, t[1], fl / t[1] / 1e9, t[0] / t[1]))
lib.mdqe_debug_gemm_variant(2)
