"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE (build container only).

    python oracle/make_golden.py            # writes tests/golden/

The reference python under /root/reference is imported from where it lies through
oracle/refshim.py (stand-ins for detectron2/torchvision/timm and for the CUDA-only extension).
Only *data* is written: seeded inputs, a manifest (names + shapes) of the weights, and the outputs
the reference produced.  No reference source text is stored.

Weights: every float parameter/buffer of a reference module is overwritten *before it runs* by
oracle/synth.py (frozen numpy RandomState keyed by parameter name), so tests can re-create the
identical state dict without the fixture carrying tens of MB.  This also removes the zero-init trap
(SURVEY.md §8d): attention_weights / sampling_offsets.weight / sampling_grid_offsets are zero and
cls_embed's last bias is -4.6 at the reference's init, which would leave data-dependent paths idle.
"""
import os
import sys
import math
import types
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402
from synth import apply_synth, manifest_to_arrays  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def npy(t):
    if torch.is_tensor(t):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(arrs)} arrays)")




# --------------------------------------------------------------------------------------------
def gen_msda():
    func = refshim.ref("mdqe.models.ops.functions.ms_deform_attn_func")
    core = func.ms_deform_attn_core_pytorch
    # (1) the reference's own known-answer recipe, mdqe/models/ops/test.py:21-60
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    arrs = {}
    for tag in ("double", "float"):      # same draw order as the script: double check first, then float
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "double":
            out = core(value.double(), shapes, loc.double(), attn.double())
        else:
            out = core(value, shapes, loc, attn)
        arrs.update({f"{tag}_value": value, f"{tag}_loc": loc, f"{tag}_attn": attn, f"{tag}_out": out})
    save("msda_reftest", shapes=shapes, level_start=lsi, **arrs)

    # (2) path-shaped cases (reduced sizes), incl. out-of-range locations and non-pow2 head dim
    g = torch.Generator().manual_seed(11)
    cases = {
        "enc": dict(B=2, shapes=[(12, 20), (6, 10), (3, 5), (2, 3)], M=8, D=32, Q=None, P=4, spread=0.15),
        "dec_spatial": dict(B=3, shapes=[(12, 20), (6, 10), (3, 5), (2, 3)], M=8, D=32, Q=49, P=4, spread=0.6),
        "dec_temporal": dict(B=1, shapes=[(6, 10)] * 4, M=8, D=32, Q=49, P=4, spread=0.6),
        "swin_d24": dict(B=2, shapes=[(8, 14), (4, 7), (2, 4), (1, 2)], M=8, D=24, Q=33, P=4, spread=0.3),
        "tiny_d8": dict(B=1, shapes=[(5, 7), (3, 4)], M=4, D=8, Q=5, P=3, spread=1.0),
    }
    arrs = {}
    for name, c in cases.items():
        sh = torch.as_tensor(c["shapes"], dtype=torch.long)
        S = int(sh.prod(1).sum())
        Q = c["Q"] or S
        L = len(c["shapes"])
        value = torch.randn(c["B"], S, c["M"], c["D"], generator=g)
        ref = torch.rand(c["B"], Q, 1, 1, 1, 2, generator=g)
        loc = ref + torch.randn(c["B"], Q, c["M"], L, c["P"], 2, generator=g) * c["spread"]
        attn = torch.softmax(torch.randn(c["B"], Q, c["M"], L * c["P"], generator=g), -1).view(c["B"], Q, c["M"], L, c["P"])
        out = core(value, sh, loc, attn)
        st = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        arrs.update({f"{name}::value": value, f"{name}::shapes": sh, f"{name}::level_start": st,
                     f"{name}::loc": loc, f"{name}::attn": attn, f"{name}::out": out})
    save("msda_cases", **arrs)


# --------------------------------------------------------------------------------------------
HID = 256   # the reference's mask head only accepts widths whose /8 is a multiple of 24 or 32


def small_pyramid_inputs(g, T, chans=(16, 24, 32), hw=(60, 90)):
    """Synthetic backbone outputs for a (60x90 -> padded 64x96) frame: strides 8/16/32."""
    H, W = 64, 96
    feats = [torch.randn(T, c, H // s, W // s, generator=g) for c, s in zip(chans, (8, 16, 32))]
    return feats, [hw] * T


def gen_encoder():
    util = refshim.ref("mdqe.util.misc")
    mdqe_mod = refshim.ref("mdqe.models.mdqe")
    enc_mod = refshim.ref("mdqe.models.transformer_enc")
    dec_mod = refshim.ref("mdqe.models.transformer_dec")
    pe = refshim.ref("mdqe.models.position_encoding")
    bb = refshim.ref("mdqe.models.backbone")
    top = refshim.ref("mdqe.mdqe")
    g = torch.Generator().manual_seed(21)
    torch.manual_seed(21)
    T = 3
    feats, sizes = small_pyramid_inputs(g, T)

    class FakeBackbone(nn.Module):
        feature_strides = [8, 16, 32]
        num_channels = [16, 24, 32]

    enc = enc_mod.Transformer_Enc(dim=HID, n_heads=8, n_feature_levels=4, n_enc_points=4, n_enc_layers=2, n_frames=3)
    dec = dec_mod.Transformer_Dec(5, dim=HID, n_heads=8, n_feature_levels=4, n_frames=3, n_dec_points=4,
                                  n_dec_layers=2, mlp_ratio=4, dec_temporal=True, n_query=16, fpn_dims=[HID, HID],
                                  window_inter_frame_asso=5, query_embed_dim=16, is_coco=False, mask_on=True)
    joiner = bb.Joiner(FakeBackbone(), pe.PositionEmbeddingSine(HID // 2, normalize=True))
    joiner.num_channels = FakeBackbone.num_channels
    joiner.feature_strides = FakeBackbone.feature_strides
    model = mdqe_mod.mdqe(joiner, enc, dec, n_frames=3, num_feature_levels=4).eval()
    manifest = apply_synth(model, seed=1, prefix="detr.")

    masks = top.MaskedBackbone.mask_out_padding(types.SimpleNamespace(feature_strides=[8, 16, 32]),
                                                [f.shape for f in feats], sizes, torch.device("cpu"))
    nts = [util.NestedTensor(f, m) for f, m in zip(feats, masks)]
    pos = [joiner[1](nt) for nt in nts]
    per_layer = []
    hooks = [l.register_forward_hook(lambda m, i, o: per_layer.append(o.detach().clone()))
             for l in model.transformer_enc.encoder.layers]
    with torch.no_grad():
        enc_out, enc_masks, shapes = model.forward_deformable_enc(nts, pos, is_training=False)
        mf = model.forward_mask_head_inference(enc_out, shapes)[0]
    for h in hooks:
        h.remove()
    arrs = {f"feat{i}": f for i, f in enumerate(feats)}
    arrs.update({f"pos{i}": p for i, p in enumerate(pos)})
    arrs.update({f"mask{i}": m for i, m in enumerate(masks)})
    arrs.update({f"enc_layer{i}": o for i, o in enumerate(per_layer)})
    arrs.update(manifest_to_arrays(manifest))
    save("encoder_small", synth_seed=1, image_sizes=np.array(sizes), enc_out=enc_out, enc_masks=enc_masks, shapes=shapes,
         mask_feats=mf, **arrs)
    return model, enc_out, enc_masks, shapes


def gen_decoder(model, enc_out, enc_masks, shapes):
    """Transformer_Dec on the encoder output above; T=3 (full clip), T=2 (short last clip), T=1."""
    dec = model.transformer_dec
    arrs = {}
    for T in (3, 2, 1):
        inter = {}
        # capture query init through the module's own methods
        with torch.no_grad():
            lsi = torch.cat([shapes.new_zeros(1), shapes.prod(-1).cumsum(0)]).long()
            q, qc, _, _ = dec.query_initialization(enc_out[:T], None, shapes, lsi, False)
            xs, xinst, boxes = dec.decoder(q, qc, enc_out[:T], shapes, enc_masks[:T])
            out = dec(enc_out[:T], enc_masks[:T], shapes)
        arrs.update({f"T{T}::query0": q, f"T{T}::coords": qc, f"T{T}::x_stack": xs, f"T{T}::x_inst_stack": xinst,
                     f"T{T}::boxes_stack": boxes, f"T{T}::cls": out["cls"], f"T{T}::mask_coeff": out["mask_coeff"],
                     f"T{T}::query_embed": out["query_embed"]})
    save("decoder_small", **arrs)


def gen_misc():
    util = refshim.ref("mdqe.util.misc")
    pe = refshim.ref("mdqe.models.position_encoding")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 2, 7, 9, generator=g)
    m = torch.zeros(2, 6, 8, dtype=torch.bool)
    m[0, 5:, :] = True
    m[0, :, 6:] = True
    m[1, :, 7:] = True
    pos = pe.PositionEmbeddingSine(16, normalize=True)(util.NestedTensor(torch.zeros(2, 4, 6, 8), m))
    save("misc_small", ab_in=x, ab_out4=util.aligned_bilinear(x, 4), ab_out2=util.aligned_bilinear(x, 2),
         pos_mask=m, pos_out=pos,
         invsig_in=torch.tensor([-0.1, 0.0, 1e-6, 0.3, 0.999999, 1.0, 1.3]),
         invsig_out=util.inverse_sigmoid(torch.tensor([-0.1, 0.0, 1e-6, 0.3, 0.999999, 1.0, 1.3])))


# --------------------------------------------------------------------------------------------
class TinyPyramid(nn.Module):
    """Deterministic stand-in for detectron2's backbone in the end-to-end capture: three strided
    convs (strides 8/16/32).  Not reference code; its weights are stored in the fixture."""

    def __init__(self, chans=(16, 24, 32)):
        super().__init__()
        self.c1 = nn.Conv2d(3, chans[0], 8, 8)
        self.c2 = nn.Conv2d(chans[0], chans[1], 2, 2)
        self.c3 = nn.Conv2d(chans[1], chans[2], 2, 2)
        self.chans = chans

    def output_shape(self):
        from detectron2.layers import ShapeSpec
        return {"res3": ShapeSpec(channels=self.chans[0], stride=8), "res4": ShapeSpec(channels=self.chans[1], stride=16),
                "res5": ShapeSpec(channels=self.chans[2], stride=32)}

    def forward(self, x):
        a = torch.tanh(self.c1(x))
        b = torch.tanh(self.c2(a))
        c = torch.tanh(self.c3(b))
        return {"res3": a, "res4": b, "res5": c}


def ns(**kw):
    return types.SimpleNamespace(**kw)


def small_cfg(thr, window=4, T=3, max_inst=40):
    return ns(
        INPUT=ns(SAMPLING_FRAME_NUM=3),
        DATASETS=ns(TEST=("ytvis_ovis_val",)),
        TEST=ns(DETECTIONS_PER_IMAGE=15),
        MODEL=ns(DEVICE="cpu", MASK_ON=True, PIXEL_MEAN=[123.675, 116.280, 103.530], PIXEL_STD=[58.395, 57.120, 57.375],
                 MDQE=ns(NUM_CLASSES=5, MASK_STRIDE=4, MATCH_STRIDE=4, HIDDEN_DIM=HID, NUM_OBJECT_QUERIES=16,
                         WINDOW_INTER_FRAME_ASSOCIATION=5, QUERY_EMBED_DIM=16, INTERINST_MASK_THRESHOLD=0.1,
                         INTERINST_MASK_LOSS_ENABLED=True, NHEADS=8, DROPOUT=0.0, ENC_LAYERS=2, DEC_LAYERS=2,
                         NUM_FEATURE_LEVELS=4, DEC_NUM_POINTS=4, ENC_NUM_POINTS=4, DEC_TEMPORAL=True, MLP_RATIO=4,
                         BOX_WEIGHT=2.0, MASK_WEIGHT=4.0, DICE_WEIGHT=4.0, DEEP_SUPERVISION=True, NO_OBJECT_WEIGHT=1,
                         CLIP_STRIDE=1, MERGE_ON_CPU=False, MULTI_CLS_ON=True, APPLY_CLS_THRES=thr,
                         SAMPLING_FRAME_NUM_TEST=T, WINDOW_FRAME_NUM_TEST=window, MAX_NUM_INSTANCES=max_inst)))


THR = 0.12


def gen_video():
    """End-to-end MDQE.inference_vis on a 9-frame 60x90 video with the tiny pyramid backbone.
    Records per-clip inference_clip results and tracker outputs by wrapping reference methods."""
    top = refshim.ref("mdqe.mdqe")
    torch.manual_seed(33)
    g = torch.Generator().manual_seed(33)
    refshim.BACKBONE_BUILDER["fn"] = lambda cfg: TinyPyramid()
    cfg = small_cfg(thr=THR)
    model = top.MDQE(cfg).eval()
    manifest = apply_synth(model.detr, seed=2, prefix="detr.")

    L = 9
    frames = [torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8) for _ in range(L)]
    # temporally smooth video so the tracker has something to match: blend with a common base
    base = torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8).float()
    frames = [(0.85 * base + 0.15 * f.float()).round().to(torch.uint8) for f in frames]

    clip_log, trk_log = [], []
    orig_clip = model.inference_clip

    def wrapped_clip(output, mask_feats, image_size):
        res, valid = orig_clip(output, mask_feats, image_size)
        clip_log.append(dict(cls=output["cls"], mask_coeff=output["mask_coeff"], query_embed=output["query_embed"],
                             mask_feats=mask_feats, scores=res.scores, pred_classes=res.pred_classes,
                             cls_probs=res.cls_probs, pred_masks=res.pred_masks, query_embeds=res.query_embeds))
        return res, valid

    model.inference_clip = wrapped_clip
    trk = refshim.ref("mdqe.tracking.OverTracker")
    orig_get = trk.OverTracker.get_result

    def wrapped_get(self, is_last_clip=False):
        c, m = orig_get(self, is_last_clip)
        trk_log.append(dict(cls=c.clone(), masks=m.clone(), num_inst=self.num_inst))
        return c, m

    trk.OverTracker.get_result = wrapped_get
    with torch.no_grad():
        out = model([{"image": frames, "height": 120, "width": 180, "file_names": ["v/000/0.jpg"]}])
    trk.OverTracker.get_result = orig_get

    arrs = {"frames": torch.stack(frames), "n_clips": len(clip_log), "n_windows": len(trk_log),
            "out_scores": np.array(out["pred_scores"], dtype=np.float32),
            "out_labels": np.array(out["pred_labels"], dtype=np.int64),
            "out_masks": torch.stack(out["pred_masks"]) if len(out["pred_masks"]) else np.zeros((0,)),
            "out_image_size": np.array(out["image_size"])}
    for i, c in enumerate(clip_log):
        for k, v in c.items():
            arrs[f"clip{i}::{k}"] = v
    for i, c in enumerate(trk_log):
        for k, v in c.items():
            arrs[f"win{i}::{k}"] = v
    arrs.update(manifest_to_arrays(manifest))
    for k, v in model.detr.backbone[0].backbone.state_dict().items():
        pass  # tiny pyramid weights are part of the manifest (synthesised like everything else)
    print("video: clips", len(clip_log), "windows", len(trk_log), "instances/clip",
          [int(c["scores"].shape[0]) for c in clip_log], "final", len(out["pred_scores"]),
          "num_inst", [c["num_inst"] for c in trk_log])
    save("video_small", thr=THR, synth_seed=2, **arrs)


def gen_coco_image():
    """MDQE.forward on the COCO single-image branch (mdqe/mdqe.py:213-236 -> inference_image :486-556; decoder is_coco
    branch transformer_dec.py:247-255): a 3-frame pseudo clip of 60x90 images, tiny pyramid backbone."""
    top = refshim.ref("mdqe.mdqe")
    g = torch.Generator().manual_seed(44)
    refshim.BACKBONE_BUILDER["fn"] = lambda cfg: TinyPyramid()
    arrs = {}
    manifest = None
    for tag, multi in (("multi", True), ("single", False)):
        cfg = small_cfg(thr=THR)
        cfg.DATASETS.TEST = ("coco_2017_val_fake",)
        cfg.MODEL.MDQE.MULTI_CLS_ON = multi
        model = top.MDQE(cfg).eval()
        assert model.is_coco
        manifest = apply_synth(model.detr, seed=4, prefix="detr.")
        if tag == "multi":
            base = torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8).float()
            frames = [(0.9 * base + 0.1 * torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8).float()).round().to(torch.uint8)
                      for _ in range(3)]
            arrs["frames"] = torch.stack(frames)
        log = {}
        orig = model.inference_image

        def wrapped(output, batched_inputs, images, log=log, orig=orig):
            log["cls"], log["masks"] = output["cls"].clone(), output["masks"].clone()
            return orig(output, batched_inputs, images)

        model.inference_image = wrapped
        with torch.no_grad():
            res = model([{"image": frames, "height": 100, "width": 140}])[0]["instances"]
        print("coco image", tag, ": instances", len(res.scores), "labels", res.pred_classes.tolist()[:8])
        arrs.update({f"{tag}::cls": log["cls"], f"{tag}::masks": log["masks"], f"{tag}::scores": res.scores,
                     f"{tag}::pred_classes": res.pred_classes, f"{tag}::pred_masks": res.pred_masks,
                     f"{tag}::pred_boxes": res.pred_boxes.tensor})
    arrs.update(manifest_to_arrays(manifest))
    save("coco_image_small", thr=THR, synth_seed=4, **arrs)


def gen_tracker():
    """Crafted clip sequence -> reference OverTracker (mdqe/tracking/OverTracker.py): persistent objects,
    one that appears late (new ID), one that disappears, and a duplicate detection."""
    trk = refshim.ref("mdqe.tracking.OverTracker")
    g = torch.Generator().manual_seed(77)
    T, WIN, K, E, HW, MAXI, THR, L = 3, 4, 5, 32, (12, 16), 12, 0.1, 11
    base = torch.randn(5, E, generator=g) * 1.5
    centers = torch.tensor([[3., 3.], [8., 11.], [5., 8.], [9., 3.], [2., 13.]])
    life = [(0, 11), (0, 11), (4, 11), (0, 6), (7, 11)]          # [first, last) frame of each object
    yy, xx = torch.meshgrid(torch.arange(HW[0]).float(), torch.arange(HW[1]).float(), indexing="ij")

    def blob(obj, f):
        cy, cx = centers[obj] + 0.2 * f * torch.tensor([1.0, -1.0 if obj % 2 else 1.0])
        return 4.0 - ((yy - cy) ** 2 + (xx - cx) ** 2) * 0.8

    tracker = trk.OverTracker(MAXI, T, WIN, 1, K, 32, E, HW, "cpu", THR)
    arrs, results, saved, n_clip = {}, [], 0, 0
    for start in range(0, L):
        end = min(start + T, L)
        last = start + T > L
        fi = list(range(start, end))
        objs = [o for o in range(5) if any(life[o][0] <= f < life[o][1] for f in fi)]
        if start == 5:
            objs = objs + [objs[0]]                                  # duplicate detection of object 0
        perm = torch.randperm(len(objs), generator=g).tolist()
        objs = [objs[i] for i in perm]
        masks = torch.stack([torch.stack([blob(o, f) if life[o][0] <= f < life[o][1] else torch.full(HW, -3.0)
                                          for f in fi]) for o in objs]) + 0.05 * torch.randn(len(objs), len(fi), *HW, generator=g)
        emb = torch.stack([base[o] for o in objs]) + 0.15 * torch.randn(len(objs), E, generator=g)
        cls = torch.rand(len(objs), K, generator=g) * 0.5
        for i, o in enumerate(objs):
            cls[i, o % K] = 0.55 + 0.4 * torch.rand(1, generator=g)
        sc, lab = cls.max(-1)
        order = sc.sort(descending=True)[1]
        res = refshim.Instances(HW, scores=sc[order], pred_classes=lab[order], cls_probs=cls[order],
                                pred_masks=masks[order], query_embeds=emb[order])
        for k in ("scores", "pred_classes", "cls_probs", "pred_masks", "query_embeds"):
            arrs[f"clip{n_clip}::{k}"] = getattr(res, k)
        arrs[f"clip{n_clip}::frame_idx"] = np.array(fi)
        tracker.update(trk.Clips(fi, res))
        arrs[f"clip{n_clip}::num_inst_after"] = tracker.num_inst
        n_clip += 1
        if last or (start + 1 >= WIN * (saved + 1)):
            c, m = tracker.get_result(is_last_clip=last)
            arrs[f"win{saved}::cls"] = c.clone()
            arrs[f"win{saved}::masks"] = m.clone()
            saved += 1
        if last:
            break
    print("tracker: clips", n_clip, "windows", saved, "num_inst", tracker.num_inst)
    save("tracker_seq", n_clips=n_clip, n_windows=saved, T=T, WIN=WIN, K=K, E=E, HW=np.array(HW), MAXI=MAXI, THR=THR, **arrs)


def gen_tracker_long():
    """A LONG crafted clip sequence -> reference OverTracker (mdqe/tracking/OverTracker.py:65-90,115-225): 44 frames, 8-frame windows (six
    window flushes, five re-basings of the bank), 4-frame clips.  Object 2 leaves the picture for 11 frames (longer than a window and longer
    than the short memory of 5 clips, shorter than the long memory of 15) and RETURNS -- it must get its old ID back through the long-memory
    similarity alone (no overlapping masks); object 3 leaves for 20 frames (longer than the long memory) and returns -- a NEW ID; object 5
    appears late; a duplicate detection and a detection that drops below 2*thr are thrown in."""
    trk = refshim.ref("mdqe.tracking.OverTracker")
    g = torch.Generator().manual_seed(91)
    T, WIN, K, E, HW, MAXI, THR, L = 4, 8, 5, 32, (12, 16), 16, 0.1, 44
    base = torch.randn(6, E, generator=g) * 1.5
    centers = torch.tensor([[3., 3.], [8., 11.], [5., 8.], [9., 3.], [2., 13.], [10., 8.]])
    life = [[(0, 44)], [(0, 44)], [(0, 12), (23, 44)], [(0, 8), (28, 44)], [(0, 30)], [(17, 44)]]     # [first, last) frame intervals
    alive = lambda o, f: any(a <= f < b for a, b in life[o])
    yy, xx = torch.meshgrid(torch.arange(HW[0]).float(), torch.arange(HW[1]).float(), indexing="ij")

    def blob(obj, f):
        cy, cx = centers[obj] + 0.05 * f * torch.tensor([1.0, -1.0 if obj % 2 else 1.0])
        return 4.0 - ((yy - cy) ** 2 + (xx - cx) ** 2) * 0.8

    tracker = trk.OverTracker(MAXI, T, WIN, 1, K, 32, E, HW, "cpu", THR)
    arrs, saved, n_clip = {}, 0, 0
    for start in range(0, L):
        end = min(start + T, L)
        last = start + T > L
        fi = list(range(start, end))
        objs = [o for o in range(6) if any(alive(o, f) for f in fi)]
        if start in (5, 26):
            objs = objs + [objs[0]]                                  # duplicate detection of the first object
        perm = torch.randperm(len(objs), generator=g).tolist()
        objs = [objs[i] for i in perm]
        masks = torch.stack([torch.stack([blob(o, f) if alive(o, f) else torch.full(HW, -3.0) for f in fi]) for o in objs]) \
            + 0.05 * torch.randn(len(objs), len(fi), *HW, generator=g)
        emb = torch.stack([base[o] for o in objs]) + 0.15 * torch.randn(len(objs), E, generator=g)
        cls = torch.rand(len(objs), K, generator=g) * 0.5
        for i, o in enumerate(objs):
            cls[i, o % K] = 0.55 + 0.4 * torch.rand(1, generator=g)
        if start == 33:
            cls[0] *= 0.3                                            # a weak detection (score < 2*thr after the scaling of its row)
        sc, lab = cls.max(-1)
        order = sc.sort(descending=True)[1]
        res = refshim.Instances(HW, scores=sc[order], pred_classes=lab[order], cls_probs=cls[order],
                                pred_masks=masks[order], query_embeds=emb[order])
        for k in ("scores", "pred_classes", "cls_probs", "pred_masks", "query_embeds"):
            arrs[f"clip{n_clip}::{k}"] = getattr(res, k)
        arrs[f"clip{n_clip}::frame_idx"] = np.array(fi)
        arrs[f"clip{n_clip}::objects"] = np.array([objs[i] for i in order.tolist()])
        tracker.update(trk.Clips(fi, res))
        arrs[f"clip{n_clip}::num_inst_after"] = tracker.num_inst
        n_clip += 1
        if last or (start + 1 >= WIN * (saved + 1)):
            c, m = tracker.get_result(is_last_clip=last)
            arrs[f"win{saved}::cls"] = c.clone()
            arrs[f"win{saved}::masks"] = m.clone()
            saved += 1
        if last:
            break
    print("tracker_long: clips", n_clip, "windows", saved, "num_inst", tracker.num_inst)
    save("tracker_long", n_clips=n_clip, n_windows=saved, T=T, WIN=WIN, K=K, E=E, HW=np.array(HW), MAXI=MAXI, THR=THR, **arrs)


def gen_swin():
    """Small SwinV2 (embed 32, heads 2/4/8/16 -> head dim 16, window 4 / last stage 2) on 2 x 64x96: exercises window
    padding (4x6 and 2x3 maps), cyclic shift masks, cosine attention with CPB bias, res-post-norm, patch merging."""
    sw = refshim.ref("mdqe.backbone.swin_transformer_v2")
    g = torch.Generator().manual_seed(66)
    torch.manual_seed(66)
    m = sw.SwinTransformerV2(patch_size=4, in_chans=3, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], window_size=4,
                             mlp_ratio=4, drop_path_rate=0.0, out_features=["stage3", "stage4", "stage5"])
    m.eval()                                  # SwinTransformerV2.train() returns None (swin_transformer_v2.py:661-664)
    manifest = apply_synth(m, seed=6, prefix="bb.")
    with torch.no_grad():                     # logit_scale / q_bias etc. got synthetic values; keep logit_scale in a sane range
        for n, p_ in m.named_parameters():
            if n.endswith("logit_scale"):
                p_.copy_(torch.log(torch.full_like(p_, 10.0)) + 0.3 * torch.randn(p_.shape, generator=g))
    x = torch.randn(2, 3, 64, 96, generator=g)
    with torch.no_grad():
        outs = m(x)
    arrs = {"x": x}
    arrs.update({k: v for k, v in outs.items()})
    arrs.update({"ls::" + n: p_ for n, p_ in m.named_parameters() if n.endswith("logit_scale")})
    arrs.update(manifest_to_arrays(manifest))
    save("swin_small", synth_seed=6, **arrs)


def gen_layer256():
    """One encoder layer + one decoder layer at the real width (C=256, D=32) on a small map."""
    enc_mod = refshim.ref("mdqe.models.transformer_enc")
    dec_mod = refshim.ref("mdqe.models.transformer_dec")
    g = torch.Generator().manual_seed(44)
    torch.manual_seed(44)
    shapes_l = [(12, 20), (6, 10), (3, 5), (2, 3)]
    shapes = torch.as_tensor(shapes_l, dtype=torch.long)
    N = int(shapes.prod(1).sum())
    T = 2
    layer = enc_mod.EncoderLayer(256, 8, 4, 4, n_frames=4, pred_offsets=True).eval()
    manifest = apply_synth(layer, seed=3, prefix="layer.")
    x = torch.randn(T, N, 256, generator=g)
    pos = torch.randn(T, N, 256, generator=g) * 0.5
    mask = torch.zeros(T, N, dtype=torch.bool)
    mask[1, 200:240] = True
    misc = refshim.ref("mdqe.models.misc")
    ref = torch.cat([misc.make_reference_points(s) for s in shapes])[None].expand(T, -1, -1)
    boxes = torch.cat([ref, torch.ones_like(ref) * 0.1], -1)
    with torch.no_grad():
        y = layer(x, pos, boxes, shapes, mask, False)
    arrs = dict(x=x, pos=pos, mask=mask, shapes=shapes, out=y)
    arrs.update(manifest_to_arrays(manifest))
    save("enc_layer256", synth_seed=3, **arrs)


def gen_msda_backward():
    """Gradients of the native op through the reference's own pure-PyTorch core (func.py:45-65) in float64 -- the
    function the reference's gradcheck script holds its CUDA backward to (mdqe/models/ops/test.py:63-86)."""
    func = refshim.ref("mdqe.models.ops.functions.ms_deform_attn_func")
    core = func.ms_deform_attn_core_pytorch
    g = torch.Generator().manual_seed(17)
    cases = {
        "enc": dict(B=1, shapes=[(6, 10), (3, 5), (2, 3), (1, 2)], M=8, D=32, Q=None, P=4, spread=0.2),
        "dec": dict(B=2, shapes=[(6, 10), (3, 5), (2, 3), (1, 2)], M=8, D=32, Q=21, P=4, spread=0.6),
        "swin_d24": dict(B=1, shapes=[(8, 14), (4, 7)], M=4, D=24, Q=33, P=2, spread=0.3),
        "tiny_d8": dict(B=1, shapes=[(5, 7), (3, 4)], M=4, D=8, Q=5, P=3, spread=1.0),
    }
    arrs = {}
    for name, c in cases.items():
        sh = torch.as_tensor(c["shapes"], dtype=torch.long)
        S = int(sh.prod(1).sum())
        Q = c["Q"] or S
        L = len(c["shapes"])
        value = torch.randn(c["B"], S, c["M"], c["D"], generator=g)
        ref = torch.rand(c["B"], Q, 1, 1, 1, 2, generator=g)
        loc = ref + torch.randn(c["B"], Q, c["M"], L, c["P"], 2, generator=g) * c["spread"]
        attn = torch.softmax(torch.randn(c["B"], Q, c["M"], L * c["P"], generator=g), -1).view(c["B"], Q, c["M"], L, c["P"])
        gout = torch.randn(c["B"], Q, c["M"] * c["D"], generator=g)
        with torch.enable_grad():
            v, lo, at = (t.double().requires_grad_(True) for t in (value, loc, attn))
            out = core(v, sh, lo, at)
            gv, gl, ga = torch.autograd.grad(out, (v, lo, at), gout.double())
        st = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        arrs.update({f"{name}::value": value, f"{name}::shapes": sh, f"{name}::level_start": st, f"{name}::loc": loc,
                     f"{name}::attn": attn, f"{name}::grad_out": gout, f"{name}::grad_value": gv.float(),
                     f"{name}::grad_loc": gl.float(), f"{name}::grad_attn": ga.float()})
    save("msda_backward", **arrs)


if __name__ == "__main__":
    refshim.install()
    if len(sys.argv) > 1 and sys.argv[1] == "msda_backward":
        gen_msda_backward()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tracker_long":
        with torch.no_grad():
            gen_tracker_long()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coco_image":
        with torch.no_grad():
            gen_coco_image()
        sys.exit(0)
    with torch.no_grad():
        gen_msda()
        gen_msda_backward()
        gen_misc()
        model, enc_out, enc_masks, shapes = gen_encoder()
        gen_decoder(model, enc_out, enc_masks, shapes)
        gen_layer256()
        gen_video()
        gen_tracker()
        gen_tracker_long()
        gen_swin()
        gen_coco_image()
