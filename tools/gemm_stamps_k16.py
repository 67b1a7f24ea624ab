"""In-kernel phase timing of the fp32 K-step-16 GEMM (wall-clock stamps per block, 10 ns ticks): prologue / K loop / epilogue.
python tools/gemm_stamps_k16.py M N K [tile]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib, ptr
shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))] if len(sys.argv) > 3 else [(5292, 256, 256), (7252, 256, 256), (21168, 256, 256), (1024, 256, 256)]
tiles = [int(sys.argv[4])] if len(sys.argv) > 4 else [3, 7]
ops.set_gemm_precision("f32")
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for tile in tiles:
        for _ in range(5):
            ops.linear(x, w, b, out=out, tile=tile)
        buf = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        lib.mdqe_debug_gemm_stamps(ptr(buf))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.linear(x, w, b, out=out, tile=tile); e1.record()
        torch.cuda.synchronize()
        lib.mdqe_debug_gemm_stamps(None)
        s = buf.view(-1, 4).cpu()
        s = s[s[:, 0] > 0]
        t0 = s[:, 0].min()
        d = (s - t0).double() * 0.01          # us
        print("M=%d N=%d K=%d tile %d: %d blocks, events %.1f us, first start -> last end %.2f us" % (M, N, K, tile, len(s), 1e3 * e0.elapsed_time(e1), float(d[:, 3].max())))
        for name, a, b_ in (("prologue", 0, 1), ("kloop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
            v = d[:, b_] - d[:, a]
            print("   %-9s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, v.mean(), v.quantile(0.1), v.quantile(0.5), v.quantile(0.9), v.max()))
        st = d[:, 0].sort().values
        print("   block start times: p1 %.2f p50 %.2f p99 %.2f max %.2f us" % tuple(float(st[int(q * (len(st) - 1))]) for q in (0.01, 0.5, 0.99, 1.0)))
