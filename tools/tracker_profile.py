"""cProfile of the tracker replay (the serial tail of the multi-GPU schedule) on the bench workload."""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE, ClipMerger
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
L = 120
video = synth_video(0, L, seed=0).cuda()
with torch.no_grad():
    clips = model.clip_schedule(L, cfg.n_frames_test, cfg.clip_stride)
    res = list(model.iter_clip_results(video, clips, 0))
    torch.cuda.synchronize()
    geo = model.engine.geometry(360, 640)
    ms = cfg.match_stride
    for rep in range(3):
        m = ClipMerger(model, (360, 640), (360, 640), (geo.Hp // ms, geo.Wp // ms))
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        if rep == 2:
            pr.enable()
        for item in res:
            m.feed(*item)
        torch.cuda.synchronize()
        if rep == 2:
            pr.disable()
        print("replay %.2f ms for %d clips" % (1e3 * (time.perf_counter() - t0), len(res)))
        m.finish()
    from mdqe_cvpr2023_amd.sharding import ReplayThread
    for rep in range(2):
        m = ClipMerger(model, (360, 640), (360, 640), (geo.Hp // ms, geo.Wp // ms))
        t0 = time.perf_counter()
        rt = ReplayThread(m, m.dev)
        rt.put(res)
        rt.t.join(0)                                   # (finish() below joins)
        rt.q.put(None); rt.t.join(); torch.cuda.synchronize()
        print("replay on a ReplayThread %.2f ms for %d clips" % (1e3 * (time.perf_counter() - t0), len(res)))
        m.side_is_current = False
        m.finish()
    pstats.Stats(pr).sort_stats("tottime").print_stats(34)
