"""Workload for the rocprofv3 --pmc passes over the fused encoder MSDA (msda_fused_v2_kernel<4,4>): the 40-frame encoder
launch of R50_ovis_360 (5100 queries per frame, 8 heads x 32 channels, 4 levels x 4 points), offsets ~ N(0, 1) image-eighths
like the model's (ms_deform_attn.py:155: one unit = W/8 pixels).  5 launches.  `python tools/pmc_msda.py time` prints timings."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
g = torch.Generator().manual_seed(0)
Bf, Mh, D, L, P = 40, 8, 32, 4, 4
shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
GEO = os.environ.get("MSDA_GEO", "360p")          # 360p (default) | 640p (R50_ovis_720, 20-frame passes) | swinl (480x864, D = 24)
if os.environ.get("MSDA_STAGE_KB"):                # LDS budget of the staged levels (how many coarse levels a block takes)
    from mdqe_cvpr2023_amd._lib import lib as _l
    _l.mdqe_debug_msda_stage_kb(int(os.environ["MSDA_STAGE_KB"]))
if GEO == "240p":                                  # a geometry whose level 1 fits the LDS beside levels 2 + 3: what staging THREE levels buys
    Bf, shapes = 80, [(30, 52), (15, 26), (8, 13), (4, 7)]
if GEO == "640p":
    Bf, shapes = 20, [(80, 144), (40, 72), (20, 36), (10, 18)]
elif GEO == "swinl":
    Bf, D, shapes = 10, 24, [(60, 108), (30, 54), (15, 27), (8, 14)]
C = Mh * D
Nq = sum(a * c for a, c in shapes)
starts = [0]
for a, c in shapes[:-1]:
    starts.append(starts[-1] + a * c)
levels = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
proj = torch.randn(Bf * Nq, C + 3 * Mh * L * P, generator=g).cuda()
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(a) + 0.5) / a, (torch.arange(c) + 0.5) / c, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                 for a, c in shapes]).float().cuda().contiguous()
out = torch.empty(Bf * Nq, C, device="cuda")
nq = 2 * Mh * L * P
COMP = Nq * (2 * C + 3 * Mh * L * P) * 4.0         # compulsory bytes per frame: value + offsets + logits + output (18.3 MB at 360p)
run = lambda: ops.msda_fused(proj[:, :C], proj[:, C:C + nq], proj[:, C + nq:], ref, levels, Bf, Nq, Mh, D, L, P, mode=0, v_brows=Nq, out=out)
if os.environ.get("MSDA_FORM") == "temporal":      # round 3: the decoder's instance-level launch of a 37-clip batch (msda_fused_tp_kernel; MDQE_TP_STAGED=0: v2)
    from mdqe_cvpr2023_amd._lib import lib
    lib.mdqe_debug_msda_tp_staged(int(os.environ.get("MDQE_TP_STAGED", "1")))
    Bc, Q, Tc = 37, 196, 4
    vals = torch.randn((Bc + Tc - 1) * Nq, C, generator=g).cuda()
    pr = (2.0 * torch.randn(Bc * Q, 3 * Mh * Tc * P, generator=g)).cuda()
    ibox = (torch.rand(Bc, Q, 4, generator=g) * torch.tensor([1, 1, 0.5, 0.5])).cuda()
    grid = torch.randn(Mh * Tc * P * 2, generator=g).cuda()
    lv_tp = ([s[0] for s in shapes for _ in range(Tc)], [s[1] for s in shapes for _ in range(Tc)], [f * Nq + starts[gi] for gi in range(L) for f in range(Tc)])
    vidx = torch.arange(Bc, dtype=torch.int32).cuda()
    out_t = torch.empty(Bc * Q, C, device="cuda")
    nqt = 2 * Mh * Tc * P
    # compulsory bytes of the launch: the 40 frames' value maps once + offsets / logits / output of the Bc*Q queries
    COMP = ((Bc + Tc - 1) * Nq * C + Bc * Q * (3 * Mh * Tc * P + C)) * 4.0 / Bf
    run = lambda: ops.msda_fused(vals, pr[:, :nqt], pr[:, nqt:], ibox, lv_tp, Bc, Q, Mh, D, Tc, P, mode=1, grid=grid, groups=L, scale=0.25, v_brows=Nq, vidx=vidx, out=out_t)
if len(sys.argv) > 1 and sys.argv[1] == "variants":        # block-to-query maps x waves-per-SIMD hint, same box, output checked against variant 0
    from mdqe_cvpr2023_amd._lib import lib
    ref_out = None
    for v in [int(a) for a in sys.argv[2:]] or (0, 1, 2, 4, 5, 6, 9):
        lib.mdqe_debug_msda_variant(v)
        run(); torch.cuda.synchronize()
        if ref_out is None:
            ref_out = out.clone()
        same = bool(torch.equal(out, ref_out))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("variant %d (map %d, waves hint %s%s): %.1f us = %.2f TB/s compulsory, identical to variant 0: %s" % (v, v & 3, "8" if v & 4 else "-", ", coarse levels in LDS" if v & 8 else "", us, COMP * Bf / us / 1e6, same), flush=True)
    lib.mdqe_debug_msda_variant(-1)
elif len(sys.argv) > 1 and sys.argv[1] == "time":
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    comp = COMP * Bf
    print("msda_fused enc 40 frames: %.1f us per launch = %.2f TB/s of the %.0f MB compulsory bytes" % (us, comp / us / 1e6, comp / 1e6))
else:
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print("done")
