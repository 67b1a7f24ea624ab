"""Round 4: the LDS-staged deformable gathers with the first staged level as a compile-time constant (straight-line gather, a level's 16
corner loads in flight together) against round 3's form (runtime level test = a basic block per sample, 1-4 loads in flight per wave),
alternated in one process; equal bits checked against the first variant.  Forms: the encoder launch (MSDA_GEO = 360p | 640p | swinl as
tools/pmc_msda.py), the decoder's temporal launch (MSDA_FORM=temporal) and the decoder's box-level launch (`dec`).
python tools/msda_r04_ab.py [enc|dec]"""
import os, runpy, sys, torch
form = sys.argv[1] if len(sys.argv) > 1 else "enc"
sys.argv = [sys.argv[0]]
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from mdqe_cvpr2023_amd._lib import lib
if form == "dec":
    from mdqe_cvpr2023_amd import ops
    g = torch.Generator().manual_seed(0)
    GEO = os.environ.get("MSDA_GEO", "360p")
    shapes = {"360p": [(48, 80), (24, 40), (12, 20), (6, 10)], "640p": [(80, 144), (40, 72), (20, 36), (10, 18)],
              "swinl": [(60, 108), (30, 54), (15, 27), (8, 14)]}[GEO]
    M, D = 8, (24 if GEO == "swinl" else 32)
    L = P = 4
    C = M * D
    N = sum(a * b for a, b in shapes)
    starts = [0]
    for a, b in shapes[:-1]:
        starts.append(starts[-1] + a * b)
    lv = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
    Bc, T, Q = int(os.environ.get("MSDA_CLIPS", "37")), (2 if GEO == "swinl" else 4), 196
    NF = Bc + T - 1
    LP = L * P
    grid = torch.randn(M * LP * 2, generator=g).cuda()
    fidx = torch.tensor([[b + t for t in range(T)] for b in range(Bc)], dtype=torch.int32).cuda()
    boxes = (torch.rand(Bc * T, Q, 4, generator=g) * torch.tensor([1, 1, 0.5, 0.5])).cuda()
    pr = (2.0 * torch.randn(Bc * T * Q, 3 * M * LP, generator=g)).cuda()
    vals = torch.randn(NF * N, C, generator=g).cuda()
    out = torch.empty(Bc * T * Q, C, device="cuda")
    run = lambda: ops.msda_fused(vals, pr[:, :2 * M * LP], pr[:, 2 * M * LP:], boxes, lv, Bc * T, Q, M, D, L, P, mode=1, grid=grid, v_brows=N,
                                 vidx=fidx.view(-1), out=out)
    COMP, Bf = (NF * N * C + Bc * T * Q * (3 * M * LP + C)) * 4.0, 1
    what = "decoder box level, %s, %d clips x %d frames" % (GEO, Bc, T)
else:
    ns = runpy.run_path(os.path.join(here, "pmc_msda.py"))
    run, COMP, Bf = ns["run"], ns["COMP"], ns["Bf"]
    out = ns["out_t"] if os.environ.get("MSDA_FORM") == "temporal" else ns["out"]
    what = ("decoder temporal launch" if os.environ.get("MSDA_FORM") == "temporal" else "encoder launch") + ", " + os.environ.get("MSDA_GEO", "360p")


def t_us(n=20):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


base = 1 | 8 if form != "dec" else 0 | 8
variants = (("r04 default (dispatcher's choice)", base), ("compile-time level, a level's 16 loads in flight, 1 block/CU", base | 1024 | 256),
            ("r03 as shipped (runtime level, natural allocation)", base | 256))
print(what, flush=True)
ref = None
for rep in range(3):
    for name, var in variants:
        lib.mdqe_debug_msda_variant(var)
        run(); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        us = t_us()
        print("  %-62s %7.1f us = %.2f TB/s of %.0f MB, equal bits: %s" % (name, us, COMP * Bf / us / 1e6, COMP * Bf / 1e6, same), flush=True)
lib.mdqe_debug_msda_variant(-1)
