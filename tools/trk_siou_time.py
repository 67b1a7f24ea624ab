"""The tracker's sign-intersection launch (mdqe_trk_siou_f32: hipMemset + kernel) at the bench's typical size -- 7 saved tracks x 4 clip
instances over 3 overlapping frames of 96 x 160 -- under different block-count targets (the pixel range is cut into chunks so that few
pairs still fill the chip; partial counts meet in float atomics).  python tools/trk_siou_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd._lib import lib, ptr, cur_stream, check
g = torch.Generator().manual_seed(0)
for n_saved, n_in in ((7, 4), (3, 2), (20, 8), (1, 1)):
    n = 3 * 96 * 160
    a = torch.randn(n_saved, n, generator=g).cuda(); b = torch.randn(n_in, n, generator=g).cuda()
    out = torch.empty(n_saved * n_in * 3, device="cuda")
    ref = None
    for blocks in (512, 0, 256, 128, 64, 1):
        lib.mdqe_debug_trk_siou_blocks(blocks)
        run = lambda: check(lib.mdqe_trk_siou_f32(ptr(a), n, n_saved, ptr(b), n, n_in, n, ptr(out), cur_stream()), "siou")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            run()
        e1.record(); torch.cuda.synchronize()
        print("%2d x %2d pairs, target %3d blocks: %6.2f us per launch (memset + kernel), equal: %s" % (n_saved, n_in, blocks, e0.elapsed_time(e1) / 200 * 1e3, bool(torch.equal(out, ref))), flush=True)
lib.mdqe_debug_trk_siou_blocks(0)
