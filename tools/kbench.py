"""Per-kernel micro-benchmarks on one MI355X (HIP events on the launch stream).

    python tools/kbench.py msda            # encoder-shaped MSDA at R50_ovis_360 sizes
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_ms(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_msda(args):
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    res = []
    for name, B, shapes, Q in (("enc360_B4", 4, [(48, 80), (24, 40), (12, 20), (6, 10)], None),
                               ("enc360_B30", 30, [(48, 80), (24, 40), (12, 20), (6, 10)], None),
                               ("dec360_B4_Q196", 4, [(48, 80), (24, 40), (12, 20), (6, 10)], 196),
                               ("enc640_B4", 4, [(80, 144), (40, 72), (20, 36), (10, 18)], None)):
        S = sum(h * w for h, w in shapes)
        Q = Q or S
        M, D, L, P = 8, 32, 4, 4
        g = torch.Generator().manual_seed(0)
        v = torch.randn(B, S, M, D, generator=g).cuda()
        ref = torch.rand(B, Q, 1, 1, 1, 2, generator=g)
        loc = (ref + 0.05 * torch.randn(B, Q, M, L, P, 2, generator=g)).cuda()
        at = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P).cuda()
        sh = torch.tensor(shapes).cuda()
        st = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1])).cuda()
        ms = time_ms(lambda: MSDA.ms_deform_attn_forward(v, sh, st, loc, at, 64))
        comp = (v.numel() + loc.numel() + at.numel() + B * Q * M * D) * 4
        gath = B * Q * M * L * P * 4 * D * 4
        res.append(dict(case=name, ms=ms, compulsory_MB=comp / 1e6, GBps_compulsory=comp / ms / 1e6,
                        gathered_MB=gath / 1e6, GBps_gathered=gath / ms / 1e6))
        print(json.dumps(res[-1]))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["msda", "gemm", "gemm32", "all"])
    a = ap.parse_args()
    if a.what in ("msda", "all"):
        bench_msda(a)


def bench_gemm(args, const=False):
    from mdqe_cvpr2023_amd import ops
    cw = ops.const_weight if const else (lambda t: t)
    res = []
    for name, M, N, K, tile in (("enc_qkv_30f", 153000, 640, 256, 1), ("enc_ffn1_30f", 153000, 1024, 256, 1), ("enc_ffn1_30f_gelu", 153000, 1024, 256, 1),
                                ("enc_ffn2_30f", 153000, 256, 1024, 1), ("enc_out_30f", 153000, 256, 256, 1),
                                ("enc_ffn1_4f", 20400, 1024, 256, 1), ("dec_q_784", 784, 256, 256, 3),
                                ("dec_val_4f", 20400, 256, 256, 1), ("dec_val_4f_t3", 20400, 256, 256, 3)):
        x = torch.randn(M, K, device="cuda"); w = cw(torch.randn(N, K, device="cuda") / K ** 0.5); b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        ms = time_ms(lambda: ops.linear(x, w, b, out=out, tile=tile, act="gelu" if name.endswith("gelu") else None), iters=20, warm=5)
        tf = 2.0 * M * N * K / ms / 1e9
        res.append(dict(case=name, M=M, N=N, K=K, tile=tile, ms=ms, TFLOPs=tf, frac_f32_mfma_peak=tf / 157.3))
        print(json.dumps(res[-1]))
    # backbone-style convs (NHWC implicit GEMM), 30 frames at 384x640
    for name, NI, H, W, Cin, Cout, k, s, p in (("res2_3x3", 30, 96, 160, 64, 64, 3, 1, 1), ("res3_3x3", 30, 48, 80, 128, 128, 3, 1, 1),
                                               ("res4_3x3", 30, 24, 40, 256, 256, 3, 1, 1), ("res5_3x3", 30, 12, 20, 512, 512, 3, 1, 1),
                                               ("res4_1x1", 30, 24, 40, 1024, 256, 1, 1, 0)):
        x = torch.randn(NI, H, W, Cin, device="cuda"); w = cw(torch.randn(Cout, k, k, Cin, device="cuda") * 0.05); b = torch.randn(Cout, device="cuda")
        ms = time_ms(lambda: ops.conv2d_nhwc(x, w, b, s, p, act="relu"), iters=20, warm=5)
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        tf = 2.0 * NI * OH * OW * Cout * Cin * k * k / ms / 1e9
        res.append(dict(case=name, ms=ms, TFLOPs=tf, frac_f32_mfma_peak=tf / 157.3))
        print(json.dumps(res[-1]))
    return res


if __name__ == "__main__" and a.what in ("gemm", "gemm32", "all"):
    bench_gemm(a)


if __name__ == "__main__" and a.what in ("gemm", "all"):
    from mdqe_cvpr2023_amd import ops as _ops
    _ops.set_gemm_precision("f16x3")
    print("--- f16x3 (in-kernel split of both operands) ---")
    bench_gemm(a)
    print("--- f16x3 (pre-split constant weights) ---")
    bench_gemm(a, const=True)
    _ops.set_gemm_precision("f32")
