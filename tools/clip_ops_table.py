"""Every op launch of ONE decoder + inference_clip batch (the bench's 37-clip batch by default), timed with an event pair around the
call on the launch stream and grouped by (op, shapes): where the per-clip stage's time goes, launch by launch.
python tools/clip_ops_table.py [frames=40] [reps=5]     (MDQE_DEC_FUSED=0 for the round-2 form)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib
if os.environ.get("MDQE_TP_STAGED"):
    lib.mdqe_debug_msda_tp_staged(int(os.environ["MDQE_TP_STAGED"]))
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 40
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
eng = model.engine
video = synth_video(0, NF, seed=0).cuda()
rec = []
on = [False]


def wrap(name, fn):
    def f(*a, **k):
        if not on[0]:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        shp = tuple(tuple(x.shape) for x in a[:3] if torch.is_tensor(x))
        rec.append((name, shp, e0, e1))
        return r
    return f


for n in ("linear", "linear_side", "linear_ln", "layernorm", "msda_fused", "mha_small", "box_refine", "box_head_refine", "add_rows", "time_fuse",
          "time_fuse_dot", "clip_assoc", "clip_gather_init", "clip_select", "dyn_mask_nms", "clip_finalize", "rows_gather"):
    setattr(ops, n, wrap(n, getattr(ops, n)))

with torch.no_grad():
    geo = eng.geometry(360, 640)
    c = model._frame_cache(video, geo)
    tot = []
    for rep in range(REPS + 1):
        on[0] = rep > 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = eng.decode_clips(c, list(range(NF - 3)), 4, geo)
        res = eng.inference_clips(outs, [c["mf"][i:i + 4] for i in range(NF - 3)])
        e1.record()
        torch.cuda.synchronize()
        if rep > 0:
            tot.append(e0.elapsed_time(e1))
agg = collections.OrderedDict()
for name, shp, a, b in rec:
    k = (name, shp)
    t = agg.setdefault(k, [0, 0.0])
    t[0] += 1; t[1] += a.elapsed_time(b)
total = sum(v[1] for v in agg.values()) / REPS
print("%d clips: %.2f ms per batch wall (events), %.2f ms summed over %d op launches per batch" % (NF - 3, sum(tot) / len(tot), total, len(rec) // REPS))
for (name, shp), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%5.1f %%  %3d x %8.1f us  %-16s %s" % (100 * ms / REPS / total, n // REPS, 1e3 * ms / n, name, " ".join("x".join(map(str, s)) for s in shp)))
