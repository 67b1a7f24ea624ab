"""Where does a 120-frame step go?  Times (same box) the per-frame stages alone (3 passes of 40 frames), the per-clip
stages alone (decoder + inference_clip over the cached frames), and the full overlapped pipeline.
Corresponds to `python bench.py` (R50_ovis_360, 120 frames, exact fp32, default schedule) with two differences that make its numbers
a little better than the bench's: the video is already in HBM (`frames_resident` in the bench line) and the "alone" figures use three
uniform 40-frame passes / three 40-clip decoder batches instead of the bench's 20/40/40/20 passes and 17/37/37/27 batches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
eng = model.engine
video = synth_video(0, 120, seed=0).cuda()


def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


with torch.no_grad():
    geo = eng.geometry(360, 640)

    def frames_only():
        for a in range(0, 120, 40):
            model._frame_cache(video[a:a + 40], geo)

    caches = [model._frame_cache(video[a:min(120, a + 43)], geo) for a in range(0, 120, 40)]

    def clips_only():
        for c in caches:
            n = c["mf"].shape[0] - 3
            outs = eng.decode_clips(c, list(range(n)), 4, geo)
            eng.inference_clips(outs, [c["mf"][i:i + 4] for i in range(n)])

    def full():
        model([{"image": video, "height": 360, "width": 640}])

    print("frame stages alone   %.1f ms / 120 frames" % timed(frames_only))
    print("clip stages alone    %.1f ms / 120 frames" % timed(clips_only))
    print("full pipeline        %.1f ms / 120 frames" % timed(full))
    # round 3: the decoder with folded positions / fused box-head + time-fuse kernels (engine.DEC_FUSED) against the round-2 form,
    # alternated in one process
    from mdqe_cvpr2023_amd import engine as E
    from mdqe_cvpr2023_amd._lib import lib
    for rep in range(3):
        for fused, two, tp in ((False, False, 0), (True, False, 0), (True, False, 1), (True, True, 1)):
            E.DEC_FUSED, E.DEC_TWO_STREAMS = fused, two
            lib.mdqe_debug_msda_tp_staged(tp)
            print("DEC_FUSED=%d TP_STAGED=%d TWO_STREAMS=%d  clip stages alone %.2f ms   full pipeline %.2f ms"
                  % (fused, tp, two, timed(clips_only, 6), timed(full, 6)), flush=True)
    E.DEC_FUSED, E.DEC_TWO_STREAMS = True, True
    lib.mdqe_debug_msda_tp_staged(1)
