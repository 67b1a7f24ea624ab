"""How long does a video go on after its LAST frame pass has finished?  (The last decoder batch, its inference_clip, the tracker, the last
window's final masks and their copy run with an idle frame stream.)  An event on the frame stream behind the last pass against the end
of `model(inputs)`.   python tools/tail_time.py [config] [frames]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
config = sys.argv[1] if len(sys.argv) > 1 else "R50_ovis_360"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cfg = PRESETS[config]
fh, fw = {"R50_ovis_360": (360, 640), "R50_ovis_720": (640, 1138), "swinl_ovis": (480, 853)}[config]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
bench.calibrate_synthetic_scores(model, sd, cfg, fh, fw)
video = bench.synth_video(0, frames, seed=0, h=fh, w=fw).pin_memory()
inp = [{"image": list(video), "height": fh, "width": fw}]
orig = model.iter_clip_results
marks = {}


def patched(frames_dev, clips, frame_offset=0, trace=None, primed=False, on_frames_queued=None, h2d=None, **kw):
    def cb():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(model._frame_stream)
        marks["frames_done"] = ev
        if on_frames_queued is not None:
            on_frames_queued()
    return orig(frames_dev, clips, frame_offset, trace, primed=primed, on_frames_queued=cb, h2d=h2d, **kw)


model.iter_clip_results = patched
# host-side marks of the tail: when does each stage of the LAST group start / end on the host, relative to the call's start
import mdqe_cvpr2023_amd.meta_arch as MA
log = []
T0 = [0.0]


def stamp(name, fn):
    def f(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        log.append((name, 1e3 * (t - T0[0]), 1e3 * (time.perf_counter() - T0[0])))
        return r
    return f


eng = model.engine
eng.decode_clips = stamp("decode_clips (launch)", eng.decode_clips)
eng.inference_clips = stamp("inference_clips (2 syncs)", eng.inference_clips)
MA.ClipMerger._consume = stamp("tracker run (+flush)", MA.ClipMerger._consume)
MA.ClipMerger._early_masks = stamp("  early masks of the window", MA.ClipMerger._early_masks)
MA.ClipMerger.finish = stamp("finish / inference_video", MA.ClipMerger.finish)
with torch.no_grad():
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T0[0] = t0; del log[:]
        e0.record()
        out = model(inp)
        e1.record()
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        if it:
            if it == 5:
                for name, a, b in log[-14:]:
                    print("   host %7.2f .. %7.2f ms  %s" % (a, b, name))
            print("video %d: wall %.1f ms; last frame pass finished at %.1f ms; tail %.1f ms" % (it, wall, e0.elapsed_time(marks["frames_done"]), marks["frames_done"].elapsed_time(e1)), flush=True)
