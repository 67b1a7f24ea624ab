"""GPU parity AT FULL SIZE on BASELINE.json's configurations, product (HIP kernels behind the C ABI) against the CPU oracle:

  * R50_ovis_360 (360x640 -> 384x640, N=5100, 196 queries, 96x160 mask maps): the whole path from uint8 frames to the
    video output -- per-clip mask logits, scores, labels, tracker windows and final boolean masks (mdqe/mdqe.py:291-471);
  * R50_ovis_720 geometry (640x1138 -> 640x1152, N=15300, levels 80x144 .. 10x18, 160x288 mask maps;
    configs/R50_ovis_720.yaml): the same chain stage by stage + size-independent MSDA properties on the 640p level table;
  * swinl_ovis at 480x853 (-> 480x864, C=192, D=24, N=8617, 2-frame clips): the same chain with the Swin-L backbone.

Two kinds of comparison.  `direct`: product and oracle each run from the frames.  `chained`: every product stage is fed the
ORACLE's output of the stage before it, so that a discrete decision that falls the other way on a near-tie (arg-max of a
query cell, a score at a threshold) cannot hide or fake a numerical difference downstream.  Bar: 1e-3 (north star) on
logits / masks, relative to the activation scale where the values are not O(1).

The oracle passes are cached per process: both GEMM precision modes compare against the same CPU run."""
import dataclasses
import functools
import os
import sys

import numpy as np
import pytest
import torch

import mdqe_oracle as O
from _golden import maxdiff, record_margin

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(params=["f32", "f16x3"])
def gemm_precision(request):
    from mdqe_cvpr2023_amd import ops
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision("f32")


def _hyper(cfg):
    return O.Hyper(hidden_dim=cfg.hidden_dim, n_frames=cfg.n_frames, n_frames_test=cfg.n_frames_test,
                   n_frames_window_test=cfg.n_frames_window_test, apply_cls_thres=cfg.apply_cls_thres, n_max_inst=cfg.n_max_inst,
                   clip_stride=cfg.clip_stride)


def _backbone_fn(cfg, sd):
    if cfg.backbone == "SwinV2":
        return lambda im: O.swinv2(sd, "detr.backbone.0.backbone", im, O.SwinHyper())
    return lambda im: O.resnet(sd, "detr.backbone.0.backbone", im, 50)


def _heavy_tails(sd, seed=5, sigma=1.15):
    """Trained-like statistics for the synthetic weights: the FrozenBatchNorm scale AND variance of every backbone
    channel are spread log-normally over three decades (sigma 1.15 in ln: 0.1 % .. 99.9 % = 1 : 1200) -- the folded per-channel scale
    w / sqrt(var + eps) becomes heavy-tailed (a few channels hundreds of times the typical one), its layer-wise RMS is kept."""
    g = torch.Generator().manual_seed(seed)
    n = 0
    for k in sorted(sd):
        if k.endswith(".norm.weight") and ".backbone." in k and (k[:-len("weight")] + "running_var") in sd:
            c = sd[k].numel()
            s = torch.exp(torch.randn(c, generator=g) * sigma)
            s = s / s.pow(2).mean().sqrt()
            t = torch.exp(torch.randn(c, generator=g) * sigma)
            base = k[:-len("weight")]
            sd[base + "running_var"] = sd[base + "running_var"] * t
            sd[k] = sd[k] * s * t.sqrt()
            n += 1
    assert n >= 50                                                   # every FrozenBN of the ResNet-50
    return sd


@functools.lru_cache(maxsize=None)
def _workload(name, fh, fw, n_frames, window, max_inst=120, tails=False):
    """Weights (random reference-style init, zero-init trap removed, class logits calibrated on the synthetic video so that
    several instances per clip survive -- bench.py's workload), OVIS-like synthetic frames, and ONE oracle pass with every
    intermediate kept."""
    from bench import calibrate_synthetic_scores, synth_video
    from mdqe_cvpr2023_amd import ops
    from mdqe_cvpr2023_amd.config import PRESETS
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    from mdqe_cvpr2023_amd.params import random_state
    base = PRESETS[name]
    cfg = dataclasses.replace(base, n_frames_window_test=window, n_max_inst=max_inst)   # (the oracle's tracker bank is [clips, max_inst, frames, h, w] on the host)
    sd = random_state(base, seed=0)
    if tails:
        sd = _heavy_tails(sd)
    prec = ops.get_gemm_precision()
    ops.set_gemm_precision("f32")
    model = MDQE(base, state_dict=sd).eval()
    calibrate_synthetic_scores(model, sd, base, fh, fw)              # shifts the class bias in sd (and in this throw-away model)
    del model
    torch.cuda.empty_cache()
    ops.set_gemm_precision(prec)
    frames = list(synth_video(0, n_frames, seed=0, h=fh, w=fw))
    hp = _hyper(cfg)
    bb = _backbone_fn(cfg, sd)
    ref = {"frames": frames, "sd": sd, "cfg": cfg, "hp": hp}
    with torch.no_grad():
        x, sizes = O.pad_frames(O.preprocess(hp, frames), 32)
        enc, mask, shapes, mf = O.frame_features(sd, hp, x, sizes, bb)
        ref.update(enc=enc, mask=mask, shapes=shapes, mf=mf)
        L, T = n_frames, cfg.n_frames_test
        clips, saved, tracker = [], 0, None
        cls_w, mask_w, win_logits = [], [], []
        fq = {}                                                      # frame -> the oracle's per-frame query initialisation (a11, before association)
        for start in range(0, L, cfg.clip_stride):
            end, last = start + T, False
            if end > L:
                last, end = True, L
            idx = list(range(start, end))
            dbg = {}
            out = O.transformer_dec(sd, hp, enc[idx], mask[idx], shapes, dbg=dbg)
            clip = O.inference_clip(hp, out, mf[:, idx])
            clip["frame_idx"] = idx
            clips.append({"start": start, "end": end, "last": last, "out": out, "coords0": dbg["coords0"], "clip": clip})
            for t, f in enumerate(idx):
                if f not in fq:
                    fq[f] = {"coords": dbg["coords0"][t], "content": dbg["content0"][t], "emb": dbg["track_emb"][t], "score_up": dbg["score_up"][t].reshape(
                        dbg["score_up"].shape[-2:])}
            if tracker is None:
                tracker = O.Tracker(hp, mf.shape[-2:])
            tracker.update(clip)
            if last or (start + cfg.clip_stride >= window * (saved + 1)):
                c, m = tracker.get_result(last)
                win_logits.append(m.clone())
                cls_w.append(c)
                mask_w.append(O.aligned_bilinear(m, hp.match_stride).sigmoid()[..., :fh, :fw])
                saved += 1
            if last:
                break
        ref.update(clips=clips, cls_w=cls_w, win_logits=win_logits, video=O.inference_video(hp, (fh, fw), cls_w, mask_w), frame_queries=fq)
    return ref


# Discrete decisions at an fp32 NEAR-TIE.  A query cell's location is the argmax of the up-sampled objectness score over the cell
# (transformer_dec.py:83-109, first max wins).  Round 6's 34-frame test met a cell whose best and second-best scores differ by ONE ulp
# (5.96e-8) in the oracle: mathematically tied candidates that fp32 rounding orders, in the reference as much as here -- no implementation can
# be held to the oracle's pick there, and the pick moves one of a clip's 196 queries (decoder heads then differ by 1e-2).  What CAN be held:
# the product's pick scores within TIE_BAND of the oracle's maximum ON THE ORACLE'S OWN SCORE MAP.  The chained tests then continue on the
# oracle's pick (their rule: every stage on the oracle's input, so that no discrete decision amplifies a rounding difference); the direct
# tests accept a clip that fails its bars only if it holds such a cell, and say so in the recorded margins.
TIE_BAND = 2e-6


def _tie_cells(ref, coords, band=TIE_BAND, frames=None):
    """coords: the product's query coordinates [frames, Q, 2] (CPU).  -> {frame: [cells]} where they differ from the oracle's; asserts that
    every such pick is within TIE_BAND of the oracle's maximum of that cell on the oracle's score map."""
    out = {}
    for f, fq in sorted(ref["frame_queries"].items()):
        if frames is not None and f not in frames:
            continue
        d = (coords[f] - fq["coords"]).abs().amax(-1) > 1e-6
        for q in d.nonzero().flatten().tolist():
            su = fq["score_up"]
            Hu, Wu = su.shape
            col, row = int(round(float(coords[f, q, 0]) * Wu)), int(float(coords[f, q, 1]) * Hu + 1e-4)
            co, ro = int(round(float(fq["coords"][q, 0]) * Wu)), int(float(fq["coords"][q, 1]) * Hu + 1e-4)
            gap = float(su[ro, co] - su[row, col])
            assert 0.0 <= gap < band, ("query cell decided differently without a near-tie", f, q, gap)
            record_margin(GROUP["name"], "a11 query cell picked at an fp32 near-tie: oracle score gap", gap, 1.0, band)
            out.setdefault(f, []).append(q)
    return out


GROUP = {"name": "?"}          # the parity-margin group of the running test: "<config> <frames> frames, <gemm mode>, <direct|chained>"


def _m(stage, got, want, tol, scale=1.0):
    """max |got - want| recorded under the running group + asserted against tol * scale."""
    d = record_margin(GROUP["name"], stage, maxdiff(got, want), scale, tol)
    assert d < tol * scale, (GROUP["name"], stage, d, tol * scale)
    return d


def _model(ref):
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    return MDQE(ref["cfg"], state_dict=ref["sd"]).eval()


def _check_video(out, ref_video, tol=1e-3):
    assert out["pred_labels"] == ref_video["pred_labels"]
    _m("video scores", torch.tensor(out["pred_scores"]), torch.tensor(ref_video["pred_scores"]), 1e-3)
    got, want = torch.stack(out["pred_masks"]), torch.stack(ref_video["pred_masks"])
    assert got.shape == want.shape and got.dtype == torch.bool
    mis = record_margin(GROUP["name"], "final masks: mismatching pixel fraction", float((got != want).float().mean()), 1.0, tol)
    assert mis < tol


def _check_clip(res, rc, scale=1.0, tag="clip"):
    assert res["pred_masks"].shape == rc["pred_masks"].shape, (res["pred_masks"].shape, rc["pred_masks"].shape)
    assert res["pred_classes"].tolist() == rc["pred_classes"].tolist()
    if rc["pred_masks"].numel():
        _m(tag + " mask logits", res["pred_masks"].cpu(), rc["pred_masks"], 1e-3, scale)
        _m(tag + " scores", res["scores"].cpu(), rc["scores"], 1e-3)
        _m(tag + " cls_probs", res["cls_probs"].cpu(), rc["cls_probs"], 1e-3)
        _m(tag + " query_embeds", res["query_embeds"].cpu(), rc["query_embeds"], 1e-3, max(1.0, float(rc["query_embeds"].abs().max())))


def _chain(ref, model, fh, fw):
    """Stage by stage, each product stage on the oracle's input: a1-a10 from the frames; a11-a14 on the oracle's encoder
    output; a15 on the oracle's decoder heads + mask features; a16-a18 on the oracle's clip results."""
    from mdqe_cvpr2023_amd.meta_arch import ClipMerger
    eng, cfg = model.engine, ref["cfg"]
    geo = eng.geometry(fh, fw)
    enc_r, mf_r = ref["enc"], ref["mf"]
    assert ref["shapes"] == geo.shapes
    with torch.no_grad():
        fd = torch.stack(ref["frames"]).cuda()
        enc = eng.encode(eng.backbone(fd, geo), geo)
        mf = eng.mask_features(enc, geo)
        _m("a1-a8 encoder tokens", enc.cpu(), enc_r, 1e-3, float(enc_r.abs().max()))
        _m("a10 mask features", mf.cpu(), mf_r.permute(1, 2, 3, 0), 1e-3, max(1.0, float(mf_r.abs().max())))
        del enc, mf
        enc_d = enc_r.cuda().contiguous()
        mf_d = mf_r.permute(1, 2, 3, 0).contiguous().cuda()                  # [frames, Hm, Wm, M] channels-last, as the engine keeps it
        coords, content, emb = eng.frame_queries(enc_d, geo)
        # a11 per frame on the oracle's encoder tokens; a cell picked differently at an fp32 near-tie (asserted) continues on the oracle's pick
        ties = _tie_cells(ref, coords.cpu())
        for f, cells in ties.items():
            fqo = ref["frame_queries"][f]
            for q in cells:
                coords[f, q], content[f, q], emb[f, q] = fqo["coords"][q].cuda(), fqo["content"][q].cuda(), fqo["emb"][q].cuda()
        for f, fqo in ref["frame_queries"].items():
            _m("a11 query coordinates", coords[f].cpu(), fqo["coords"], 1e-5)
            _m("a11 query content", content[f].cpu(), fqo["content"], 1e-3, max(1.0, float(fqo["content"].abs().max())))
            _m("a11 track embedding", emb[f].cpu(), fqo["emb"], 1e-3, max(1.0, float(fqo["emb"].abs().max())))
        vals = eng.dec_values(enc_d, geo)
        cache = {"coords": coords, "content": content, "emb": emb, "vals": vals}
        lscale = max(1.0, max(float(c["clip"]["pred_masks"].abs().max()) for c in ref["clips"] if c["clip"]["pred_masks"].numel()))
        items = []
        for c in ref["clips"]:
            s, e = c["start"], c["end"]
            # a11-a14: decoder of this clip on the oracle's encoder tokens
            out = eng.decode_clips(cache, [s], e - s, geo)
            for k in ("cls", "mask_coeff", "query_embed"):
                _m("a11-a14 decoder " + k, out[k][0].cpu(), c["out"][k][0], 1e-3, max(1.0, float(c["out"][k].abs().max())))
            # a15: inference_clip on the ORACLE's decoder heads
            outs_r = {k: c["out"][k].cuda().contiguous() for k in ("cls", "mask_coeff", "query_embed")}
            res = eng.inference_clips(outs_r, [mf_d[s:e]])[0]
            _check_clip(res, c["clip"], lscale, "a15 inference_clip (oracle heads)")
            # ... and end to end within the clip stage (decoder -> inference_clip on product values)
            res2 = eng.inference_clips(out, [mf_d[s:e]])[0]
            _check_clip(res2, c["clip"], lscale, "a11-a15 decoder + inference_clip")
            rc = c["clip"]
            items.append((s, e, c["last"], {"scores": rc["scores"].cuda(), "pred_classes": rc["pred_classes"].cuda(),
                                            "cls_probs": rc["cls_probs"].cuda(), "query_embeds": rc["query_embeds"].cuda(),
                                            "pred_masks": rc["pred_masks"].cuda().contiguous()}))
        torch.cuda.synchronize()
        # a16-a18: tracker + window flushes + final masks on the ORACLE's clip results, both mask-merge modes
        ms = cfg.match_stride
        for merge_on_cpu, early in ((True, True), (False, True), (False, False), (True, False)):
            model.merge_on_cpu, model.early_masks = merge_on_cpu, early    # (early: masks per flushed window; late: one pass at the end)
            m = ClipMerger(model, (fh, fw), (fh, fw), (geo.Hp // ms, geo.Wp // ms), n_frames=len(ref["frames"]))
            m.feed_many([(s, e, l, dict(r)) for s, e, l, r in items])
            assert len(m.cls_clips) == len(ref["cls_w"])
            for a, b in zip(m.cls_clips, ref["cls_w"]):
                _m("a16 tracker window cls", a, b, 1e-5)
            _check_video(m.finish(), ref["video"])
        model.merge_on_cpu, model.early_masks = None, True


def _direct(ref, model, fh, fw):
    trace = []
    with torch.no_grad():
        out = model.inference_vis([{"image": ref["frames"], "height": fh, "width": fw}], trace=trace)
        # the product's own a11 picks on ITS encoder tokens: where they differ from the oracle's they must be near-ties (TIE_BAND)
        eng = model.engine
        geo = eng.geometry(fh, fw)
        enc = eng.encode(eng.backbone(torch.stack(ref["frames"]).cuda(), geo), geo)
        # (on the product's own tokens the score map carries the backbone's and the encoder's rounding too: a ten times wider band)
        ties = _tie_cells(ref, eng.frame_queries(enc, geo)[0].cpu(), band=10 * TIE_BAND)
        del enc
    assert len(trace) == len(ref["clips"])
    lscale = max(1.0, max(float(c["clip"]["pred_masks"].abs().max()) for c in ref["clips"] if c["clip"]["pred_masks"].numel()))
    tied = 0
    for res, c in zip(trace, ref["clips"]):
        if any(f in ties for f in range(c["start"], c["end"])):
            # a clip that holds a near-tie cell: one of its 196 queries sits elsewhere than the oracle's -- not held to the bars (the chained
            # test holds the same clip on the oracle's pick); counted, so the margins file says how many
            tied += 1
            continue
        _check_clip(res, c["clip"], lscale, "a1-a15 frames -> clip")
    record_margin(GROUP["name"], "clips not held to the oracle (a near-tie query cell)", float(tied), 1.0, float(len(ref["clips"])))
    assert tied <= max(1, len(ref["clips"]) // 8), tied
    if tied == 0:
        _check_video(out, ref["video"])
    else:                                            # the tracker's inputs differ in those clips: the video result is checked for form only
        assert len(out["pred_masks"]) == len(out["pred_scores"]) == len(out["pred_labels"]) >= 1
    assert out["pred_masks"][0].shape == (len(ref["frames"]), fh, fw)
    return out


def test_r50_ovis_360_full_size_end_to_end(gemm_precision):
    """6 frames of 360x640, 4-frame clips, 4-frame tracker windows (two flushes, one carry): frames -> boolean masks."""
    ref = _workload("R50_ovis_360", 360, 640, 6, 4)
    assert sum(int(c["clip"]["scores"].numel()) for c in ref["clips"]) > len(ref["clips"])      # the workload keeps >1 instance per clip
    model = _model(ref)
    GROUP["name"] = "R50_ovis_360 6x360x640 %s chained" % gemm_precision
    _chain(ref, model, 360, 640)
    GROUP["name"] = "R50_ovis_360 6x360x640 %s direct" % gemm_precision
    _direct(ref, model, 360, 640)


def test_r50_ovis_360_shipped_schedule_full_window_flush_and_carry():
    """The SHIPPED R50_ovis_360 schedule at full size against the CPU oracle (until now reached only transitively):
    34 frames of 360x640, 4-frame clips stride 1, WINDOW_FRAME_NUM_TEST = 30 -- one full 30-frame tracker window flush, the carry of the
    T-1 overlapping frames across it and a short last window (mdqe/mdqe.py:301-364, tracking/OverTracker.py:195-225).  Chained (every
    product stage on the oracle's input) and direct (frames -> boolean masks in the default multi-pass, multi-stream schedule)."""
    from mdqe_cvpr2023_amd.config import PRESETS
    base = PRESETS["R50_ovis_360"]
    assert base.n_frames_window_test == 30 and base.n_frames_test == 4 and base.clip_stride == 1       # the shipped values, not a test's
    # the oracle's tracker bank is [clips, max_inst, window + T frames, 96, 160] fp32 on the HOST, twice at a flush: 2 x 7.8 GB at the
    # shipped MAX_NUM_INSTANCES = 120 -- taken when the box has the memory, else 60 (a capacity only: ~10 instances live here)
    avail_gb = next((int(ln.split()[1]) / 1e6 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")), 0.0)
    max_inst = base.n_max_inst if avail_gb > 48 else 60
    ref = _workload("R50_ovis_360", 360, 640, 34, base.n_frames_window_test, max_inst=max_inst)
    assert len(ref["clips"]) == 32 and len(ref["cls_w"]) == 2 and ref["clips"][-1]["last"]          # 31 full clips + the short last one, two flushes
    assert ref["clips"][-1]["end"] - ref["clips"][-1]["start"] == 3
    assert ref["win_logits"][0].shape[1] == 30 and ref["win_logits"][1].shape[1] == 4               # a full window, then the short last one
    assert sum(int(c["clip"]["scores"].numel()) for c in ref["clips"]) > len(ref["clips"])
    model = _model(ref)
    assert model.cfg.n_frames_window_test == 30 and model.cfg.n_max_inst == max_inst
    GROUP["name"] = "R50_ovis_360 34x360x640 window 30 (shipped schedule) f32 chained"
    _chain(ref, model, 360, 640)
    GROUP["name"] = "R50_ovis_360 34x360x640 window 30 (shipped schedule) f32 direct"
    _direct(ref, model, 360, 640)
    # ... and the fast mode (f16x3 split precision: what bench.py's `fast_mode` times) on the same video and oracle pass, same bars
    from mdqe_cvpr2023_amd import ops
    ops.set_gemm_precision("f16x3")
    try:
        GROUP["name"] = "R50_ovis_360 34x360x640 window 30 (shipped schedule) f16x3 direct"
        _direct(ref, _model(ref), 360, 640)
    finally:
        ops.set_gemm_precision("f32")
    _workload.cache_clear()                                          # (34 frames of oracle intermediates: not kept for the later tests)


def test_r50_ovis_360_full_size_heavy_tailed_weights():
    """The same full-size chain with trained-like weight statistics: FrozenBN scales and variances of the whole backbone spread over three
    decades (`_heavy_tails`), class logits re-calibrated on that state.  Exact fp32 mode, chained (every product stage on the oracle's input,
    so no discrete decision can amplify a rounding difference) and direct, the same 1e-3 bars."""
    ref = _workload("R50_ovis_360", 360, 640, 6, 4, tails=True)
    model = _model(ref)
    # the state really is heavy-tailed: folded per-channel scales of one layer span more than two decades
    sd = ref["sd"]
    k = "detr.backbone.0.backbone.res3.0.conv2.norm."
    eff = (sd[k + "weight"] * (sd[k + "running_var"] + 1e-5).rsqrt()).abs()
    assert float(eff.max() / eff.min()) > 100.0
    GROUP["name"] = "R50_ovis_360 6x360x640 f32 heavy-tailed BN chained"
    _chain(ref, model, 360, 640)
    GROUP["name"] = "R50_ovis_360 6x360x640 f32 heavy-tailed BN direct"
    _direct(ref, model, 360, 640)


def test_r50_ovis_360_full_size_reference_precision_map():
    """`precision_map = "reference"`: the regions the reference's harness runs under fp16 autocast on a GPU (backbone, input_proj, the
    embed MLPs, the mask head) on the f16x3 split-precision kernels, the forced-fp32 regions exact -- frames -> boolean masks
    against the fp32 CPU oracle under the same 1e-3 bars, and the regions really switch kernels (the thread-scoped mode is visible inside
    an autocast region only)."""
    from mdqe_cvpr2023_amd import ops
    ref = _workload("R50_ovis_360", 360, 640, 6, 4)
    model = _model(ref)
    model.engine.precision_map = "reference"
    assert ops.get_gemm_precision() == "f32"
    with model.engine.amp():
        assert ops.get_gemm_precision() == "f16x3"
    assert ops.get_gemm_precision() == "f32"
    GROUP["name"] = "R50_ovis_360 6x360x640 reference precision map direct"
    out = _direct(ref, model, 360, 640)
    model.engine.precision_map = ""
    with torch.no_grad():
        exact = model.inference_vis([{"image": ref["frames"], "height": 360, "width": 640}])
    assert out["pred_labels"] == exact["pred_labels"]
    assert out["pred_scores"] != exact["pred_scores"]            # a different arithmetic ran (not a silently ignored switch)


def test_r50_ovis_360_full_size_autocast_f16_margins():
    """`precision_map = "autocast_f16"` (round 5): the regions the reference's harness runs under fp16 autocast on a GPU on ONE f16 MFMA pass
    (operands rounded to nearest f16, fp32 accumulation and results) -- what the reference actually executes there, up to its fp16
    activations.  This arithmetic is NOT held to the 1e-3 bar (the reference's own op test accepts rtol 1e-2 for half inputs, ops/test.py:
    46-60): the achieved margins against the fp32 CPU oracle are RECORDED per stage (parity_margins: group "... autocast f16") and bounded
    loosely (5e-2 of the activation scale) so that a broken kernel cannot hide; stages whose inputs are the oracle's (chained), so that a
    discrete decision cannot amplify a rounding difference."""
    from mdqe_cvpr2023_amd import ops
    ref = _workload("R50_ovis_360", 360, 640, 6, 4)
    model = _model(ref)
    eng = model.engine
    eng.precision_map = "autocast_f16"
    with eng.amp():
        assert ops.get_gemm_precision() == "f16"
    assert ops.get_gemm_precision() == "f32"
    GROUP["name"] = "R50_ovis_360 6x360x640 autocast f16 (one f16 MFMA pass in the reference's autocast regions) chained"
    geo = eng.geometry(360, 640)
    TOL = 5e-2
    with torch.no_grad():
        fd = torch.stack(ref["frames"]).cuda()
        enc = eng.encode(eng.backbone(fd, geo), geo)                       # backbone + input_proj in f16, encoder exact
        _m("a1-a8 encoder tokens", enc.cpu(), ref["enc"], TOL, float(ref["enc"].abs().max()))
        enc_d = ref["enc"].cuda().contiguous()
        mf = eng.mask_features(enc_d, geo)                                 # the mask head on the ORACLE's tokens
        _m("a10 mask features (oracle tokens)", mf.cpu(), ref["mf"].permute(1, 2, 3, 0), TOL, max(1.0, float(ref["mf"].abs().max())))
        coords, content, emb = eng.frame_queries(enc_d, geo)
        cache = {"coords": coords, "content": content, "emb": emb, "vals": eng.dec_values(enc_d, geo)}
        # the query selection (arg-max of the rpn scores per grid cell, transformer_dec.py:81-109) is a DISCRETE decision downstream of an
        # autocast region: in f16 arithmetic some cells pick another location, and such a query has nothing in common with the oracle's.
        # Recorded: the share of cells that agree with the exact-fp32 selection, and the decoder heads' error (not asserted: a flipped
        # cell is a different query, and through self-attention it perturbs the others)
        eng.precision_map = ""
        coords_x, _, _ = eng.frame_queries(enc_d, geo)
        eng.precision_map = "autocast_f16"
        agree = float((coords == coords_x).all(-1).float().mean())
        record_margin(GROUP["name"], "a11 query cells that pick another location than exact fp32 (fraction)", 1.0 - agree, 1.0, None)
        assert agree > 0.9
        for c in ref["clips"]:
            s, e = c["start"], c["end"]
            out = eng.decode_clips(cache, [s], e - s, geo)
            for k in ("cls", "mask_coeff", "query_embed"):
                assert bool(torch.isfinite(out[k]).all())
                record_margin(GROUP["name"], "a11-a14 decoder " + k + " (oracle tokens; incl. queries whose cell flipped)",
                              maxdiff(out[k][0].cpu(), c["out"][k][0]), max(1.0, float(c["out"][k].abs().max())), None)
        out = model.inference_vis([{"image": ref["frames"], "height": 360, "width": 640}])
    eng.precision_map = ""
    with torch.no_grad():
        exact = model.inference_vis([{"image": ref["frames"], "height": 360, "width": 640}])
    assert len(out["pred_masks"]) >= 1 and out["pred_masks"][0].shape == exact["pred_masks"][0].shape
    assert out["pred_scores"] != exact["pred_scores"]                      # a different arithmetic ran
    # end to end: how much of the final boolean masks differs from the exact-fp32 run where the same (label) instances came out
    if out["pred_labels"] == exact["pred_labels"]:
        got, want = torch.stack(out["pred_masks"]), torch.stack(exact["pred_masks"])
        record_margin(GROUP["name"], "final masks vs the exact-fp32 run: mismatching pixel fraction", float((got != want).float().mean()), 1.0, TOL)
        record_margin(GROUP["name"], "video scores vs the exact-fp32 run", maxdiff(torch.tensor(out["pred_scores"]), torch.tensor(exact["pred_scores"])), 1.0, TOL)


def test_r50_ovis_720_geometry_full_size(gemm_precision):
    """configs/R50_ovis_720.yaml geometry: 640x1138 -> 640x1152, N=15300; 3 frames (one short clip -> last-frame repeat in the
    temporal attention, transformer_dec.py:382-386), APPLY_CLS_THRES 0.2, MERGE_ON_CPU both ways."""
    ref = _workload("R50_ovis_720", 640, 1138, 3, 20, 40)
    model = _model(ref)
    geo = model.engine.geometry(640, 1138)
    assert (geo.Hp, geo.Wp, geo.N) == (640, 1152, 15300) and geo.shapes == [(80, 144), (40, 72), (20, 36), (10, 18)]
    GROUP["name"] = "R50_ovis_720 3x640x1138 %s chained" % gemm_precision
    _chain(ref, model, 640, 1138)
    GROUP["name"] = "R50_ovis_720 3x640x1138 %s direct" % gemm_precision
    _direct(ref, model, 640, 1138)


def test_r50_ovis_720_shipped_schedule_window_flush_and_carry():
    """configs[2] (R50_ovis_720) on ITS shipped schedule at full size: 24 frames of 640x1138, 4-frame clips stride 1, WINDOW_FRAME_NUM_TEST =
    20, MERGE_ON_CPU, APPLY_CLS_THRES 0.2 -- one full 20-frame window flush, the carry across it and a short last window (22 clips) --
    chained and direct against the CPU oracle (round 6; until now 3 frames).  Near-tie query cells as in the 360p test (`_tie_cells`)."""
    from mdqe_cvpr2023_amd.config import PRESETS
    base = PRESETS["R50_ovis_720"]
    assert base.n_frames_window_test == 20 and base.n_frames_test == 4 and base.clip_stride == 1 and base.merge_on_cpu
    # (the oracle's tracker bank: [clips, max_inst, window + T frames, 160, 288] fp32 on the host, twice at a flush: 2 x 5 GB at 60 instances)
    ref = _workload("R50_ovis_720", 640, 1138, 24, base.n_frames_window_test, max_inst=60)
    assert len(ref["clips"]) == 22 and len(ref["cls_w"]) == 2 and ref["clips"][-1]["last"]
    assert ref["win_logits"][0].shape[1] == 20 and ref["win_logits"][1].shape[1] == 4
    model = _model(ref)
    GROUP["name"] = "R50_ovis_720 24x640x1138 window 20 (shipped schedule) f32 chained"
    _chain(ref, model, 640, 1138)
    GROUP["name"] = "R50_ovis_720 24x640x1138 window 20 (shipped schedule) f32 direct"
    _direct(ref, model, 640, 1138)
    _workload.cache_clear()


def test_msda_640p_level_table_properties():
    """The native op on the 640p level table (S = Q = 15300, B = 2): linearity in value, partition of unity on a constant map,
    agreement with the oracle on a slice of queries of every level."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    g = torch.Generator().manual_seed(1)
    shapes = [(80, 144), (40, 72), (20, 36), (10, 18)]
    starts = [0, 11520, 14400, 15120]
    S = 15300
    B, M, D, L, P = 2, 8, 32, 4, 4
    sh, st = torch.tensor(shapes, dtype=torch.int64).cuda(), torch.tensor(starts, dtype=torch.int64).cuda()
    v1, v2 = torch.randn(B, S, M, D, generator=g), torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, S, 1, 1, 1, 2, generator=g) + 0.05 * torch.randn(B, S, M, L, P, 2, generator=g)
    at = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P)
    f = lambda v: MSDA.ms_deform_attn_forward(v.cuda(), sh, st, loc.cuda(), at.cuda(), 64)
    o1, o2, o12 = f(v1), f(v2), f(2 * v1 - 3 * v2)
    assert maxdiff((2 * o1 - 3 * o2).cpu(), o12.cpu()) < 1e-4
    loc_in = 0.25 + 0.5 * torch.rand(B, S, M, L, P, 2, generator=g)
    oc = MSDA.ms_deform_attn_forward(torch.full((B, S, M, D), -0.75).cuda(), sh, st, loc_in.cuda(), at.cuda(), 64).cpu()
    assert maxdiff(oc, torch.full_like(oc, -0.75)) < 1e-5
    for q0 in (0, 11500, 14390, 15100):                                      # slices that straddle the level boundaries
        ref = O.msda_forward(v1[1:2], shapes, starts, loc[1:2, q0:q0 + 200], at[1:2, q0:q0 + 200])
        assert maxdiff(o1[1:2, q0:q0 + 200].cpu(), ref) < 2e-5


def test_swinl_ovis_480p_full_size(gemm_precision):
    """swinl_ovis.yaml at its own size: 3 frames of 480x853 -> 480x864, Swin-L (window 12/6, 195 M parameters), hidden 192 (head
    dim 24, mask dim 24, N = 8617), 2-frame clips: the same chained + direct comparison as the R50 configurations -- stage3/4/5
    maps through the encoder and mask head, decoder, inference_clip, tracker, final masks."""
    ref = _workload("swinl_ovis", 480, 853, 3, 20, 40)
    model = _model(ref)
    geo = model.engine.geometry(480, 853)
    assert (geo.Hp, geo.Wp, geo.N) == (480, 864, 8617) and geo.shapes == [(60, 108), (30, 54), (15, 27), (8, 14)]
    GROUP["name"] = "swinl_ovis 3x480x853 %s chained" % gemm_precision
    _chain(ref, model, 480, 853)
    GROUP["name"] = "swinl_ovis 3x480x853 %s direct" % gemm_precision
    _direct(ref, model, 480, 853)


def test_swinl_ovis_shipped_schedule_window_flush_and_carry():
    """configs[3] (swinl_ovis) on ITS shipped schedule at full size: 22 frames of 480x853, 2-frame clips stride 1, WINDOW_FRAME_NUM_TEST = 20,
    MERGE_ON_CPU, APPLY_CLS_THRES 0.1 -- a full window flush, the carry of the overlapping frame across it, a short last window, a
    one-frame last clip (22 clips) -- chained and direct against the CPU oracle (round 6; until now 3 frames)."""
    from mdqe_cvpr2023_amd.config import PRESETS
    base = PRESETS["swinl_ovis"]
    assert base.n_frames_window_test == 20 and base.n_frames_test == 2 and base.clip_stride == 1 and base.merge_on_cpu
    ref = _workload("swinl_ovis", 480, 853, 22, base.n_frames_window_test, max_inst=60)
    assert len(ref["clips"]) == 22 and len(ref["cls_w"]) == 2 and ref["clips"][-1]["last"]
    assert ref["clips"][-1]["end"] - ref["clips"][-1]["start"] == 1 and ref["win_logits"][0].shape[1] == 20 and ref["win_logits"][1].shape[1] == 2
    model = _model(ref)
    GROUP["name"] = "swinl_ovis 22x480x853 window 20 (shipped schedule) f32 chained"
    _chain(ref, model, 480, 853)
    GROUP["name"] = "swinl_ovis 22x480x853 window 20 (shipped schedule) f32 direct"
    _direct(ref, model, 480, 853)
    _workload.cache_clear()


def test_checkpoint_load_path_equals_constructor_path():
    """B-model: weights through `load_state_dict` of a released-checkpoint-shaped dict (aliases, buffers, criterion.*) give the
    same engine outputs, bit for bit, as weights through the constructor."""
    from bench import synth_video
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    from test_bmodel_contract_cpu import SMALL, _released_checkpoint
    cfg = MDQEConfig(**SMALL, n_frames_test=3, n_frames_window_test=4, n_max_inst=40)
    sd, ckpt = _released_checkpoint(cfg, seed=4)
    a = MDQE(cfg, state_dict=sd).eval()
    b = MDQE(cfg, seed=9).eval()
    _ = b.engine                                                              # an engine built from the OLD weights must be dropped
    b.load_state_dict(ckpt, strict=True)
    frames = synth_video(0, 7, seed=3, h=64, w=96, n_obj=3).cuda()
    inp = [{"image": frames, "height": 64, "width": 96}]
    oa, ob = a(inp), b(inp)
    assert oa["pred_labels"] == ob["pred_labels"] and oa["pred_scores"] == ob["pred_scores"]
    assert all(bool((x == y).all()) for x, y in zip(oa["pred_masks"], ob["pred_masks"]))
