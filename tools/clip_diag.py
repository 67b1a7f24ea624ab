"""How many queries survive each step of inference_clip on the synthetic bench workload?"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
name = sys.argv[1] if len(sys.argv) > 1 else "R50_ovis_360"
cfg = PRESETS[name]
seed = 0
fh, fw = {"R50_ovis_360": (360, 640), "R50_ovis_720": (640, 1138), "swinl_ovis": (480, 853)}[name]
model = MDQE(cfg, state_dict=random_state(cfg, seed=seed)).eval()
eng = model.engine
nf = cfg.n_frames_window_test
T = cfg.n_frames_test
video = synth_video(0, nf, seed=0, h=fh, w=fw).cuda()
with torch.no_grad():
    geo = eng.geometry(fh, fw)
    c = model._frame_cache(video, geo)
    nc = nf - T + 1
    outs = eng.decode_clips(c, list(range(nc)), T, geo)
    cls, emb = outs["cls"], outs["query_embed"]
    ss, si = cls.max(-1)[0].sort(descending=True, dim=1)
    thr = cfg.apply_cls_thres
    keep = ss >= torch.clamp(ss[:, :1], max=thr)
    print("thr", thr, "scores: top1 mean %.4f  median-query mean %.4f  min %.4f" % (float(ss[:, 0].mean()), float(ss[:, ss.shape[1] // 2].mean()), float(ss.min())))
    lg = torch.logit(ss.clamp(1e-6, 1 - 1e-6))
    qs = torch.tensor([0.5, 0.8, 0.9, 0.95, 0.98, 1.0])
    print("max-class logit quantiles over queries (mean over clips):", [round(float(v), 3) for v in torch.quantile(lg, qs.to(lg.device), dim=1).mean(1)])
    print("pass threshold per clip:", keep.sum(1).tolist()[:10])
    e = F.normalize(torch.gather(emb, 1, si[..., None].expand(-1, -1, emb.shape[-1])), dim=-1)
    sim = torch.bmm(e, e.transpose(1, 2))
    off = sim[:, ~torch.eye(sim.shape[1], dtype=torch.bool, device=sim.device)]
    print("pairwise cos of query embeds: mean %.4f min %.4f  frac>0.99 %.3f" % (float(off.mean()), float(off.min()), float((off > 0.99).float().mean())))
    res = eng.inference_clips(outs, [c["mf"][i:i + T] for i in range(nc)])
    print("final instances per clip:", [len(r["scores"]) for r in res][:10])
    print("mask coeff abs mean %.4f; mask feats abs mean %.4f" % (float(outs["mask_coeff"].abs().mean()), float(c["mf"].abs().mean())))
