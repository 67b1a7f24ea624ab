"""Tile shape by measurement, with ANOTHER kernel between the timed launches: back-to-back launches of one GEMM keep the blocks of a CU
in lockstep (all in their epilogue at once), which penalises the 128x128 tile in a way the pipeline never sees -- round 1's sweeps had
that artefact.  Shapes = the plain GEMMs of a 40-frame pass at 360p (+ the 3x3 convs with `conv`).  python tools/tile_sweep_interleaved.py [conv]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mdqe_cvpr2023_amd import ops
from kbench import time_ms
ops.set_gemm_precision("f32")
y = torch.randn(204000, 256, device="cuda"); wy = torch.randn(64, 256, device="cuda")
other = lambda: ops.linear(y, wy, None)
t_other = time_ms(other, iters=30, warm=5)
if len(sys.argv) > 1 and sys.argv[1] == "conv":
    for (NI, H, W, Cin, Cout, s) in ((40, 24, 40, 256, 256, 1), (40, 48, 80, 128, 128, 1), (40, 96, 160, 64, 64, 1), (40, 12, 20, 512, 512, 1), (40, 96, 160, 128, 128, 2), (40, 48, 80, 256, 256, 2)):
        x = torch.randn(NI, H, W, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / 48; b = torch.randn(Cout, device="cuda")
        M = NI * ((H + 2 - 3) // s + 1) * ((W + 2 - 3) // s + 1)
        row = []
        for tile in (0, 1, 2, 3, 9):
            def f():
                other(); ops.conv2d_nhwc(x, w, b, s, 1, act="relu", tile=tile)
            ms = time_ms(f, iters=15, warm=4) - t_other
            row.append("t%d %.0f us %.0f TF" % (tile, 1e3 * ms, 2.0 * M * Cout * 9 * Cin / ms / 1e9))
        print("conv3x3/%d M=%d N=%d K=%d | %s" % (s, M, Cout, 9 * Cin, " | ".join(row)), flush=True)
    sys.exit(0)
for (M, N, K, act, res) in ((204000, 1024, 256, "gelu", False), (204000, 640, 256, None, False), (204000, 3072, 256, None, False), (38400, 1024, 256, "relu", True),
                            (614400, 256, 64, "relu", True), (153600, 512, 128, "relu", True), (38400, 256, 1024, "relu", False), (153600, 256, 512, None, False),
                            (153600, 256, 256, None, False), (153600, 128, 512, "relu", False), (9600, 2048, 512, "relu", True), (614400, 64, 256, "relu", False),
                            (614400, 128, 256, "relu", False), (9600, 512, 2048, "relu", False), (38400, 512, 1024, "relu", False), (31360, 256, 256, None, False),
                            (31360, 1024, 256, "gelu", False), (31360, 512, 256, None, False), (7840, 256, 256, None, False)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    row = []
    for tile in (0, 1, 2, 3, 9):
        def f():
            other(); ops.linear(x, w, b, act=act, residual=r, res_first=True, out=out, tile=tile)
        ms = time_ms(f, iters=15, warm=4) - t_other
        row.append("t%d %.0f us %.0f TF" % (tile, 1e3 * ms, 2.0 * M * N * K / ms / 1e9))
    print("M=%d N=%d K=%d %s%s | %s" % (M, N, K, act or "", " +res" if res else "", " | ".join(row)), flush=True)
