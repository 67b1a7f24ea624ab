"""Which stage's bits depend on the number of frames / clips in a launch?  Runs the per-frame stages of the bench model on frames
[0, nb) and on [0, ns) and compares the first ns frames of every intermediate (backbone maps, encoder tokens, mask features, query
initialisation, decoder values); then the decoder + inference_clip on all clips of the big cache against the first clips alone.
python tools/pass_size_diag.py [config] [big] [small]      (config: R50_ovis_360 | R50_ovis_720 | swinl_ovis)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state

config = sys.argv[1] if len(sys.argv) > 1 else "R50_ovis_360"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = PRESETS[config]
fh, fw = {"R50_ovis_360": (360, 640), "R50_ovis_720": (640, 1138), "swinl_ovis": (480, 853)}[config]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
bench.calibrate_synthetic_scores(model, sd, cfg, fh, fw)
eng = model.engine
frames = bench.synth_video(0, nb, seed=0, h=fh, w=fw).cuda()
geo = eng.geometry(fh, fw)


def cmp(name, a, b):
    a, b = a[:ns].contiguous(), b[:ns].contiguous()
    same = torch.equal(a, b)
    print("%-28s %s%s" % (name, "same bits" if same else "DIFFERS", "" if same else "  max |d| %.3e of %.3e" % (float((a - b).abs().max()), float(b.abs().max()))), flush=True)


with torch.no_grad():
    outs = []
    for n in (nb, ns):
        fr = frames[:n]
        feats = eng.backbone(fr, geo)
        enc = eng.encode(feats, geo)
        mf = eng.mask_features(enc, geo)
        coords, content, emb = eng.frame_queries(enc, geo)
        vals = eng.dec_values(enc, geo)
        outs.append(dict({"feat%d" % i: f for i, f in enumerate(feats)}, enc=enc, mf=mf, coords=coords, content=content, emb=emb, vals=vals))
    for k in outs[0]:
        cmp(k, outs[0][k], outs[1][k])
    # encoder on IDENTICAL inputs (the big pass's backbone maps): isolates the encoder from backbone differences
    enc_s = eng.encode([f[:ns].contiguous() for f in (outs[0]["feat%d" % i] for i in range(len(feats)))], geo)
    cmp("encode(big feats[:ns])", outs[0]["enc"], enc_s)
    cmp("mask_features(big enc[:ns])", outs[0]["mf"], eng.mask_features(outs[0]["enc"][:ns].contiguous(), geo))
    cmp("dec_values(big enc[:ns])", outs[0]["vals"], eng.dec_values(outs[0]["enc"][:ns].contiguous(), geo))
    c, co, e = eng.frame_queries(outs[0]["enc"][:ns].contiguous(), geo)
    cmp("frame_queries.content", outs[0]["content"], co); cmp("frame_queries.emb", outs[0]["emb"], e)
    # decoder: all clips of the big cache in one batch against the first clips alone (same cache)
    T = cfg.n_frames_test
    cache = {k: outs[0][k] for k in ("mf", "coords", "content", "emb", "vals")}
    starts_all = list(range(nb - T + 1)); starts_few = starts_all[:max(ns - T + 1, 1)]
    d_all = eng.decode_clips(cache, starts_all, T, geo)
    d_few = eng.decode_clips(cache, starts_few, T, geo)
    for k in d_all:
        if torch.is_tensor(d_all[k]):
            a, b = d_all[k][:len(starts_few)], d_few[k]
            same = torch.equal(a, b)
            print("decode_clips[%-14s] %s%s" % (k, "same bits" if same else "DIFFERS", "" if same else "  max |d| %.3e" % float((a - b).abs().max())), flush=True)
    r_all = eng.inference_clips(d_all, cache["mf"], starts_all, T)
    r_few = eng.inference_clips(d_few, cache["mf"], starts_few, T)
    for i, (x, y) in enumerate(zip(r_all, r_few)):
        for k in x:
            if torch.is_tensor(x[k]) and not (x[k].shape == y[k].shape and torch.equal(x[k], y[k])):
                print("inference_clips clip %d [%s] DIFFERS" % (i, k), flush=True)
                break
    print("inference_clips compared on %d clips" % len(r_few))
