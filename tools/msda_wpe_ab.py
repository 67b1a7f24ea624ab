"""Encoder MSDA: the 8-waves-per-SIMD build of msda_fused_v3_kernel<.., 1024> (64 VGPRs: two 1024-thread blocks per CU) against the natural
allocation (68 VGPRs = 7 waves per SIMD: ONE block per CU), alternated in one process on the 40-frame 360p launch; equal bits checked.
python tools/msda_wpe_ab.py   (MSDA_GEO=640p|swinl as tools/pmc_msda.py)"""
import os, runpy, sys, torch
sys.argv = [sys.argv[0]]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_msda.py"))     # builds the workload, runs it 5 times
from mdqe_cvpr2023_amd._lib import lib
run, out, COMP, Bf = ns["run"], ns["out"], ns["COMP"], ns["Bf"]


def t_us(n=20):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = None
for rep in range(3):
    for name, var in (("8 waves/SIMD build (2 blocks per CU)", 1 | 8), ("natural allocation (1 block per CU)", 1 | 8 | 256)):
        lib.mdqe_debug_msda_variant(var)
        run(); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        us = t_us()
        print("%-40s %7.1f us = %.2f TB/s of the %.0f MB algorithmic bytes, equal bits: %s" % (name, us, COMP * Bf / us / 1e6, COMP * Bf / 1e6, same), flush=True)
lib.mdqe_debug_msda_variant(-1)
