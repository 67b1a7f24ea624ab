"""Round 4: `decoder_norm` as a second LayerNorm in norm3's GEMM epilogue (mdqe_gemm_ln2_f32) against its own launch -- the per-clip stage
alone (decoder + inference_clip over cached frames, tools/stream_split.py's method), alternated in one process.
python tools/ln2_ab.py [rounds]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
bench.calibrate_synthetic_scores(model, sd, cfg, 360, 640)
eng = model.engine
video = bench.synth_video(0, 120, seed=0).cuda()
with torch.no_grad():
    geo = eng.geometry(360, 640)
    caches = [model._frame_cache(video[a:min(120, a + 43)], geo) for a in range(0, 120, 40)]

    def clips_only():
        for c in caches:
            n = c["mf"].shape[0] - 3
            outs = eng.decode_clips(c, list(range(n)), 4, geo)
            eng.inference_clips(outs, c["mf"], list(range(n)), 4)

    def timed(reps=6):
        clips_only(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            clips_only()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    for r in range(rounds):
        for fused in (True, False):
            ops.LINEAR_LN2_FUSED = fused
            print("decoder_norm in norm3's epilogue=%d   per-clip stage alone %.2f ms / 120 frames" % (fused, timed()), flush=True)
ops.LINEAR_LN2_FUSED = True
