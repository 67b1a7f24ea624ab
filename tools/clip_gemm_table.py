"""Every GEMM-type launch of the per-clip stage ALONE (one batch of 40 clips over a cached 43-frame pass, decoder on ONE stream), timed
with an event pair each: shape -> launches, average microseconds, TFLOP/s, share of the batch's GEMM time.  Then the same shapes as plain
products on every tile shape of the fp32 K-step-16 kernel (ops.linear(tile=...)): is the dispatcher's choice the fastest one?
    python tools/clip_gemm_table.py [frames]"""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
from mdqe_cvpr2023_amd import ops, _lib
from kbench import time_ms
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
eng = model.engine
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 43
T = cfg.n_frames_test
video = synth_video(0, NF, seed=0, h=360, w=640).cuda()
L = _lib.load_library()
rec = []
on = [False]


class Wrapped:
    def __getattr__(self, name):
        fn = getattr(L, name)
        kinds = {"mdqe_gemm_nt_f32": "gemm", "mdqe_gemm_nt_side_f32": "gemm+side", "mdqe_gemm_ln_f32": "gemm+LN", "mdqe_gemm_ln2_f32": "gemm+LN2"}
        if name not in kinds:
            return fn

        def call(*a):
            if not on[0]:
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*a); e1.record()
            rec.append((kinds[name], a[6], a[7], a[8], e0, e1))
            return rc
        return call


ops.lib = Wrapped()
with torch.no_grad():
    geo = eng.geometry(360, 640)
    c = model._frame_cache(video, geo)
    starts = list(range(NF - T + 1))
    for rep in range(5):
        on[0] = rep >= 2
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = eng.decode_clips(c, starts, T, geo, two_streams=False)
        eng.inference_clips(outs, c["mf"], starts, T)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("rep %d: %.3f ms per batch of %d clips (one stream%s)" % (rep, 1e3 * (t1 - t0), len(starts), ", an event pair per GEMM" if on[0] else ""))
    on[0] = False
    acc = collections.OrderedDict()
    for kind, M, N, K, e0, e1 in rec:
        k = (kind, M, N, K)
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1)
    tot = sum(v[1] for v in acc.values())
    reps = 3
    print("GEMM-type launches: %d per batch, %.3f ms per batch" % (len(rec) // reps, tot / reps))
    for (kind, M, N, K), (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%5.1f %%  %3d x %7.1f us  %6.1f TF  %-9s M=%6d N=%5d K=%5d" % (100 * ms / tot, n // reps, 1e3 * ms / n, 2.0 * M * N * K * n / ms / 1e9, kind, M, N, K))
    print()
    shapes = sorted({(M, N, K) for (kind, M, N, K) in acc if kind == "gemm" and N > 8}, key=lambda s: (-s[0], s[1], s[2]))
    for M, N, K in shapes:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        row = []
        for tile in (0, 1, 2, 3, 7, 8, 9):
            try:
                t = time_ms(lambda: ops.linear(x, w, b, out=out, tile=tile), iters=30, warm=5)
                row.append("t%d %.1fus %.0fTF" % (tile, 1e3 * t, 2.0 * M * N * K / t / 1e9))
            except Exception as e:
                row.append("t%d err" % tile)
        print("M=%5d N=%4d K=%4d | " % (M, N, K) + " | ".join(row), flush=True)
