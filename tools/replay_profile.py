"""Where the time of rank 0's tracker replay goes (the serial part of the sharded schedule): the clip results of the bench's 120-frame video,
repeated W times under shifted frame indices (a W*120-frame video, as sharding.expand_root_load builds it), fed to a ClipMerger on an
otherwise idle GPU.  Host seconds inside the native update (counts launch / counts wait / decision / accumulate launch: mdqe_debug_trk_times),
the Python around it, and the window flushes (get_result + final masks + D2H).      python tools/replay_profile.py [W] [fast=1|0]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd._lib import lib
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE, ClipMerger
from mdqe_cvpr2023_amd.params import random_state
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib.mdqe_debug_trk_fast(fast)
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
L, T = 120, cfg.n_frames_test
video = synth_video(0, L + T - 1, seed=0).cuda()
with torch.no_grad():
    clips = [(s, s + T, False) for s in range(L)]                 # 120 full clips (a middle chunk of a long video)
    res = list(model.iter_clip_results(video, clips, 0))
    torch.cuda.synchronize()
    items = []
    for r in range(W):
        for s, e, l, d in res:
            d2 = {k: v for k, v in d.items() if k != "rows"}
            s2, e2 = s + r * L, e + r * L
            last = r == W - 1 and s == L - 1
            items.append((s2, min(e2, W * L), last, d2 if e2 <= W * L else dict(d2, pred_masks=d2["pred_masks"][:, :W * L - s2].contiguous())))
    items = [(s, e, l, d) for s, e, l, d in items if e - s == T or l]
    geo = model.engine.geometry(360, 640)
    ms = cfg.match_stride
    for rep in range(3):
        m = ClipMerger(model, (360, 640), (360, 640), (geo.Hp // ms, geo.Wp // ms), n_frames=W * L)
        t_flush = [0.0]
        import types, weakref
        wm = weakref.ref(m)                         # (no cycle merger -> wrapper -> merger: the pinned pool re-issues a video's buffers when the
        def timed_early(mm, wm=wm):                 #  LAST reference to its result is gone, and a cycle would keep them until a gc pass)
            t0 = time.perf_counter(); ClipMerger._early_masks(wm(), mm); t_flush[0] += time.perf_counter() - t0
        m._early_masks = timed_early
        lib.mdqe_debug_trk_times(None, 1)
        torch.cuda.synchronize()
        pr = None
        if os.environ.get("REPLAY_CPROFILE") == "2":
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
        t0 = time.perf_counter()
        m.feed_many(items)
        t1 = time.perf_counter()
        if pr is not None:
            pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tt = (ctypes.c_double * 5)()
        lib.mdqe_debug_trk_times(tt, 0)
        out = m.finish()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        n = len(items)
        nat = sum(tt[:4])
        print("rep %d: W=%d fast=%d  %d clips, %d tracks: feed %.1f ms = %.1f us/clip | native %.1f ms (counts launch %.1f, counts wait %.1f, decision %.1f, accumulate launch %.1f; "
              "%d updates) | final masks + D2H queueing %.1f ms | python + get_result %.1f ms | drain %.1f ms, finish %.1f ms"
              % (rep, W, fast, n, m.tracker.num_inst, 1e3 * (t1 - t0), 1e6 * (t1 - t0) / n, 1e3 * nat, 1e3 * tt[0], 1e3 * tt[1], 1e3 * tt[2], 1e3 * tt[3], int(tt[4]),
                 1e3 * t_flush[0], 1e3 * (t1 - t0 - nat - t_flush[0]), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
    if os.environ.get("REPLAY_CPROFILE"):
        import cProfile, pstats
        m = ClipMerger(model, (360, 640), (360, 640), (geo.Hp // ms, geo.Wp // ms), n_frames=W * L)
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        m.feed_many(items)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
