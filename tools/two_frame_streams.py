"""Are the frame stream's kernel boundaries (every launch waits for the last blocks of the one before) worth filling with a SECOND frame
pass?  R50_ovis_360, 120 resident frames: the per-frame stages as 4 x 30 frames on one stream, as 2 + 2 passes on two streams side by side,
and as 8 x 15 frames on one / two streams.   python tools/two_frame_streams.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
model = MDQE(cfg, state_dict=random_state(cfg, seed=0)).eval()
video = synth_video(0, 120, seed=0).cuda()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


with torch.no_grad():
    geo = model.engine.geometry(360, 640)

    def run(size, streams):
        keep = []
        for i, a in enumerate(range(0, 120, size)):
            with torch.cuda.stream(streams[i % len(streams)]):
                keep.append(model._frame_cache(video[a:a + size], geo))
        return keep

    for rnd in range(3):
        print("4 x 30 one stream %.1f ms | 4 x 30 two streams %.1f ms | 8 x 15 one stream %.1f ms | 8 x 15 two streams %.1f ms | 2 x 60 one %.1f ms | 2 x 60 two %.1f ms"
              % (timed(lambda: run(30, [sA])), timed(lambda: run(30, [sA, sB])), timed(lambda: run(15, [sA])), timed(lambda: run(15, [sA, sB])),
                 timed(lambda: run(60, [sA])), timed(lambda: run(60, [sA, sB]))), flush=True)
