"""Round 6: the encoder's deformable launch with a block iteration = an 8 x 16 PATCH of queries (MDQE_MSDA_PATCH=1, off by default: measured slower) against a run of 128 consecutive
tokens (rounds 2-5), on the bench's own model and video (real offsets: what decides the texture path's hit rate), all three configs:
microseconds per launch (HIP events around every launch of 3 encoder passes, alternated), equal bits of the encoder output.
    python tools/msda_patch_ab.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_bench_shapes_gpu import _bench_model
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib, load_library

L = load_library()
raw = L.mdqe_msda_fused_f32
rec = []


class Wrapped:
    def __getattr__(self, name):
        return getattr(L, name)

    def mdqe_msda_fused_f32(self, *a):
        if a[11] != 0:
            return raw(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = raw(*a); e1.record()
        rec.append((e0, e1))
        return rc


ops.lib = Wrapped()
for config, frames in (("R50_ovis_360", 40), ("R50_ovis_720", 20), ("swinl_ovis", 10)):
    bench, cfg, model, fh, fw = _bench_model(config)
    eng = model.engine
    geo = eng.geometry(fh, fw)
    video = bench.synth_video(0, frames, seed=0, h=fh, w=fw).cuda()
    outs, res = {}, {0: [], 1: []}
    with torch.no_grad():
        feats = eng.backbone(video, geo)
        for rep in range(4):
            for patch in (1, 0):
                lib.mdqe_debug_msda_patch(patch)
                rec.clear()
                enc = eng.encode(feats, geo)
                torch.cuda.synchronize()
                if rep:
                    res[patch] += [a.elapsed_time(b) * 1e3 for a, b in rec]
                outs[patch] = enc.clone()
    lib.mdqe_debug_msda_patch(1)
    same = bool(torch.equal(outs[0], outs[1]))
    m = {k: sum(v) / len(v) for k, v in res.items()}
    print("%-13s %2d frames %dx%d: encoder MSDA launch  patches %.1f us   token runs %.1f us   (%d launches each; encoder output identical: %s)"
          % (config, frames, fh, fw, m[1], m[0], len(res[1]), same), flush=True)
    del model, eng, video, feats, outs
    torch.cuda.empty_cache()
