"""GPU parity of the product pipeline (HIP kernels behind the C ABI) against reference-generated goldens
and the CPU oracle.  Bar: 1e-3 on logits/masks (north star); stage tolerances below are tighter."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mdqe_oracle as O
from _golden import Fixture, maxdiff, record_margin

REFG = "vs the REFERENCE's own output (golden fixtures, reduced size)"


def _rm(stage, got, want, tol):
    """max |got - want| against a reference-run fixture: recorded (profiles/r04_parity_margins.txt) and asserted."""
    d = record_margin(REFG, stage, maxdiff(got, want), 1.0, tol)
    assert d < tol, (stage, d, tol)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"], autouse=True)
def gemm_precision(request):
    """Every pipeline parity test runs in both GEMM modes with the SAME tolerances: exact fp32 MFMA and the
    split-precision f16x3 kernel."""
    from mdqe_cvpr2023_amd import ops
    ops.set_gemm_precision(request.param)
    yield request.param
    ops.set_gemm_precision("f32")


def small_cfg(**kw):
    from mdqe_cvpr2023_amd.config import MDQEConfig
    d = dict(backbone="custom", backbone_channels=(16, 24, 32), enc_layers=2, dec_layers=2, n_frames=3, num_classes=5,
             num_queries=16, query_embed_dim=16)
    d.update(kw)
    return MDQEConfig(**d)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def test_encoder_and_mask_head_vs_reference(ln_rows):
    from mdqe_cvpr2023_amd.engine import Engine
    fx = Fixture("encoder_small")
    eng = Engine(small_cfg(), fx.state())
    geo = eng.geometry(60, 90)
    assert geo.shapes == fx.shapes() and torch.equal(geo.mask_flat.cpu(), fx.t("enc_masks")[0])
    feats = [nhwc(fx.t(f"feat{i}")) for i in range(3)]
    enc = eng.encode(feats, geo)
    _rm("encoder_small: input_proj + 6 encoder layers", enc.cpu(), fx.t("enc_out"), 2e-4)
    mf = eng.mask_features(enc, geo)                      # [NI,Hm,Wm,M]
    ref = fx.t("mask_feats").permute(1, 2, 3, 0)          # [M,T,H,W] -> [T,H,W,M]
    assert mf.shape == ref.shape
    _rm("encoder_small: mask features", mf.cpu(), ref, 5e-4)


@pytest.fixture(params=["default", "ln_epilogue_everywhere"])
def ln_rows(request):
    """The encoder/decoder `norm(x + linear(..))` steps take the LayerNorm-epilogue GEMM only from 16384 rows up; the second
    setting forces it for every row count, so the small fixtures go through it too (ragged 64-row tiles)."""
    from mdqe_cvpr2023_amd import ops
    old = ops.LINEAR_LN_MIN_ROWS
    ops.LINEAR_LN_MIN_ROWS = old if request.param == "default" else 0
    yield request.param
    ops.LINEAR_LN_MIN_ROWS = old


@pytest.mark.parametrize("T", [3, 2, 1])
def test_decoder_vs_reference(T, ln_rows):
    from mdqe_cvpr2023_amd.engine import Engine
    enc_fx, fx = Fixture("encoder_small"), Fixture("decoder_small")
    eng = Engine(small_cfg(), enc_fx.state())
    geo = eng.geometry(60, 90)
    enc = enc_fx.t("enc_out")[:T].cuda().contiguous()
    coords, content, emb = eng.frame_queries(enc, geo)
    vals = eng.dec_values(enc, geo)
    out = eng.decode_clip(coords, content, emb, vals, geo)
    for k in ("cls", "mask_coeff", "query_embed"):
        ref = fx.t(f"T{T}::{k}")[0]
        assert out[k].shape == ref.shape
        _rm("decoder_small T=%d: %s" % (T, k), out[k].cpu(), ref, 5e-4)


def tiny_pyramid_gpu(sd):
    p = "detr.backbone.0.backbone."
    w = {k: sd[p + k].cuda() for k in ("c1.weight", "c1.bias", "c2.weight", "c2.bias", "c3.weight", "c3.bias")}
    mean = torch.tensor([123.675, 116.280, 103.530]).view(1, 3, 1, 1).cuda()
    std = torch.tensor([58.395, 57.120, 57.375]).view(1, 3, 1, 1).cuda()

    def fn(frames, geo):       # test-only stand-in for detectron2's backbone (as in the golden capture)
        x = (frames.float() - mean) / std
        xp = torch.zeros(x.shape[0], 3, geo.Hp, geo.Wp, device=x.device)
        xp[:, :, :x.shape[2], :x.shape[3]] = x
        a = torch.tanh(F.conv2d(xp, w["c1.weight"], w["c1.bias"], 8))
        b = torch.tanh(F.conv2d(a, w["c2.weight"], w["c2.bias"], 2))
        c = torch.tanh(F.conv2d(b, w["c3.weight"], w["c3.bias"], 2))
        return [t.permute(0, 2, 3, 1).contiguous() for t in (a, b, c)]
    return fn


@pytest.mark.parametrize("frame_batch", [4, 30, 1])
def test_video_end_to_end_vs_reference(frame_batch):
    """Reference MDQE.inference_vis output (9 frames, short last clip, two tracker windows) reproduced by the
    compute-once engine, for several frame-cache chunk sizes."""
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    fx = Fixture("video_small")
    sd = fx.state()
    cfg = small_cfg(apply_cls_thres=fx.f("thr"), n_frames_test=3, n_frames_window_test=4, n_max_inst=40)
    model = MDQE(cfg, state_dict=sd, backbone_fn=tiny_pyramid_gpu(sd)).eval()
    model.frame_batch = frame_batch
    trace = []
    frames = list(fx.t("frames"))
    with torch.no_grad():
        out = model.inference_vis([{"image": frames, "height": 120, "width": 180}], trace=trace)
    assert len(trace) == fx.i("n_clips")
    for i, c in enumerate(trace):
        ref = fx.t(f"clip{i}::pred_masks")
        assert c["pred_masks"].shape == ref.shape, i
        _rm("video_small: clip mask logits", c["pred_masks"].cpu(), ref, 1e-3)
        _rm("video_small: clip scores", c["scores"].cpu(), fx.t(f"clip{i}::scores"), 1e-3)
    assert out["pred_labels"] == fx.t("out_labels").tolist()
    _rm("video_small: video scores", torch.tensor(out["pred_scores"]), torch.from_numpy(np.asarray(fx.z["out_scores"])).float(), 1e-3)
    got, ref = torch.stack(out["pred_masks"]), fx.t("out_masks")
    assert got.shape == ref.shape and got.dtype == torch.bool and got.device.type == "cpu"
    assert record_margin(REFG, "video_small: final masks, mismatching pixel fraction", float((got != ref).float().mean()), 1.0, 1e-3) < 1e-3


def test_resnet50_vs_oracle():
    """detectron2's ResNet-50 is third-party/unpinned: the HIP backbone is checked against the oracle's
    restatement (same synthetic weights)."""
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.engine import Engine
    from mdqe_cvpr2023_amd.params import full_manifest
    from synth import synth_tensor
    cfg = MDQEConfig(enc_layers=1, dec_layers=1)
    sd = {k: synth_tensor(k, s, 7) for k, s in full_manifest(cfg).items()}
    eng = Engine(cfg, sd)
    g = torch.Generator().manual_seed(1)
    frames = torch.randint(0, 256, (2, 3, 90, 120), generator=g, dtype=torch.uint8)
    geo = eng.geometry(90, 120)
    outs = eng.backbone(frames.cuda(), geo)
    x = O.pad_frames(O.preprocess(O.Hyper(), list(frames)), 32)[0]
    ref = O.resnet(sd, "detr.backbone.0.backbone", x, 50)
    for o, r in zip(outs, ref):
        o = o.permute(0, 3, 1, 2).cpu()
        assert o.shape == r.shape
        assert float((o - r).abs().max() / r.abs().max()) < 1e-4


# (full-size R50_ovis_360 / R50_ovis_720 / swinl_ovis parity: tests/test_fullsize_gpu.py)


def test_swinv2_backbone_vs_reference():
    """HIP SwinV2 backbone against the reference's SwinTransformerV2 output (fixture swin_small)."""
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.engine import Engine
    from mdqe_cvpr2023_amd.params import head_manifest
    from synth import synth_tensor
    fx = Fixture("swin_small")
    cfg = MDQEConfig(backbone="SwinV2", swin_embed_dim=32, swin_depths=(2, 2, 2, 2), swin_heads=(2, 4, 8, 16), swin_window=4,
                     backbone_channels=(64, 128, 256), enc_layers=1, dec_layers=1, pixel_mean=(0., 0., 0.), pixel_std=(1., 1., 1.))
    sd = {k.replace("bb.", "detr.backbone.0.backbone.", 1): v for k, v in fx.state().items()}
    for k in fx.z.files:
        if k.startswith("ls::"):
            sd["detr.backbone.0.backbone." + k[4:]] = fx.t(k)
    sd.update({k: synth_tensor(k, s, 9) for k, s in head_manifest(cfg).items()})
    eng = Engine(cfg, sd)
    x = fx.t("x")                                      # [2,3,64,96] already "normalised" (mean 0, std 1)
    geo = eng.geometry(64, 96)
    outs = eng.backbone(x.cuda().contiguous(), geo)
    for o, name in zip(outs, ("stage3", "stage4", "stage5")):
        ref = fx.t(name)
        o = o.permute(0, 3, 1, 2).cpu()
        assert o.shape == ref.shape
        _rm("swin_small: " + name, o, ref, 2e-4)


def test_swin_window_partition_fused_into_gemm_and_norm_is_bit_identical():
    """Round 4: the window partition as row addressing of the qkv product's A loads (mdqe_gemm_nt_swin_f32) and its reverse + residual as
    the stores of norm1 (mdqe_layernorm_swin_scatter_f32) against the copy kernels they replace: the same products on the same rows, the
    same LayerNorm, one commutative add -- equal bits, on maps whose sizes are NOT multiples of the window (zero-padded windows), with
    and without the cyclic shift, and through the whole reduced backbone."""
    from mdqe_cvpr2023_amd import engine as E, ops
    from mdqe_cvpr2023_amd._lib import MdqeError
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.params import head_manifest
    from synth import synth_tensor
    g = torch.Generator().manual_seed(3)
    if ops.get_gemm_precision() != "f32":              # the split-precision kernels do not know the row map: the op refuses, the engine
        with pytest.raises(MdqeError):                 # partitions first (both backbone runs below then take the copy kernels)
            ops.linear_swin(torch.zeros(1, 8, 8, 32).cuda(), torch.zeros(96, 32).cuda(), torch.zeros(96).cuda(), 4, 0)
    for (B, H, W, C, ws, shift) in ((3, 15, 27, 96, 6, 0), (2, 15, 27, 96, 6, 3), (2, 30, 54, 192, 12, 6), (1, 8, 14, 48, 4, 2)):
        if ops.get_gemm_precision() != "f32":
            break
        x = torch.randn(B, H, W, C, generator=g).cuda()
        w = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda()
        b = torch.randn(3 * C, generator=g).cuda()
        ref = ops.linear(ops.swin_window_gather(x, ws, shift), w, b)
        got = ops.linear_swin(x, w, b, ws, shift)
        assert got.shape == ref.shape and torch.equal(got, ref), (B, H, W, C, ws, shift)
        rows = torch.randn(ref.shape[0], C, generator=g).cuda()
        gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
        want = ops.swin_window_scatter_add(ops.layernorm(rows, gam, bet), x, ws, shift)
        assert torch.equal(ops.layernorm_swin_scatter(rows, gam, bet, x, ws, shift, out=torch.empty_like(x)), want)
        xc = x.clone()
        assert torch.equal(ops.layernorm_swin_scatter(rows, gam, bet, xc, ws, shift), want) and xc.data_ptr() != x.data_ptr()   # in place
    fx = Fixture("swin_small")
    cfg = MDQEConfig(backbone="SwinV2", swin_embed_dim=32, swin_depths=(2, 2, 2, 2), swin_heads=(2, 4, 8, 16), swin_window=4,
                     backbone_channels=(64, 128, 256), enc_layers=1, dec_layers=1, pixel_mean=(0., 0., 0.), pixel_std=(1., 1., 1.))
    sd = {k.replace("bb.", "detr.backbone.0.backbone.", 1): v for k, v in fx.state().items()}
    for k in fx.z.files:
        if k.startswith("ls::"):
            sd["detr.backbone.0.backbone." + k[4:]] = fx.t(k)
    sd.update({k: synth_tensor(k, s, 9) for k, s in head_manifest(cfg).items()})
    eng = E.Engine(cfg, sd)
    x = fx.t("x").cuda().contiguous()
    geo = eng.geometry(64, 96)
    old = E.SWIN_FUSED
    try:
        E.SWIN_FUSED = True
        a = eng.backbone(x, geo)
        E.SWIN_FUSED = False
        b = eng.backbone(x, geo)
    finally:
        E.SWIN_FUSED = old
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_build_swinv2_backbone_surface_vs_reference():
    """The detectron2-facing builder (`build_swinv2_backbone(cfg, input_shape)`, swin_transformer_v2.py:675-702): reference parameter
    names through `load_state_dict`, `forward(x)` -> {"stage3","stage4","stage5"} NCHW equal to the reference run (fixture swin_small),
    `output_shape()` as the meta-architecture reads it (mdqe/mdqe.py:28-30)."""
    from types import SimpleNamespace as NS
    from mdqe_cvpr2023_amd import build_swinv2_backbone
    fx = Fixture("swin_small")
    cfg = NS(MODEL=NS(DEVICE="cuda", SWIN=NS(EMBED_DIM=32, DEPTHS=[2, 2, 2, 2], NUM_HEADS=[2, 4, 8, 16], WINDOW_SIZE=4, MLP_RATIO=4,
                                             OUT_FEATURES=["stage3", "stage4", "stage5"])))
    bb = build_swinv2_backbone(cfg, NS(channels=3)).eval()
    sd = {k.replace("bb.", "", 1): v for k, v in fx.state().items()}
    for k in fx.z.files:
        if k.startswith("ls::"):
            sd[k[4:]] = fx.t(k)
    missing, unexpected = bb.load_state_dict(sd, strict=False)
    assert not missing, missing[:5]
    outs = bb(fx.t("x").cuda())
    assert list(outs) == ["stage3", "stage4", "stage5"]
    shp = bb.output_shape()
    for name, stride, ch in (("stage3", 8, 64), ("stage4", 16, 128), ("stage5", 32, 256)):
        ref = fx.t(name)
        assert outs[name].shape == ref.shape and shp[name].stride == stride and shp[name].channels == ch
        assert maxdiff(outs[name].cpu(), ref) < 2e-4, name
    with pytest.raises(RuntimeError):
        bb(torch.zeros(1, 3, 60, 96, device="cuda"))                     # not padded to 32: loud


def test_swinl_ovis_runs_end_to_end():
    """swinl_ovis (config 4) at a reduced frame size: Swin-L backbone (195 M params, window 12/6), hidden 192 (head dim 24,
    mask dim 24, GroupNorm 24), 2-frame clips.  Encoder output vs the CPU oracle; whole video well-formed."""
    from mdqe_cvpr2023_amd.config import SWINL_OVIS
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    from mdqe_cvpr2023_amd.params import random_state
    cfg = SWINL_OVIS
    sd = random_state(cfg, seed=1)
    g = torch.Generator().manual_seed(3)
    frames = [torch.randint(0, 256, (3, 120, 216), generator=g, dtype=torch.uint8) for _ in range(3)]
    model = MDQE(cfg, state_dict=sd).eval()
    eng = model.engine
    geo = eng.geometry(120, 216)
    with torch.no_grad():
        feats = eng.backbone(torch.stack(frames).cuda(), geo)
        enc = eng.encode(feats, geo)
    hp = O.Hyper(hidden_dim=192, n_frames=2, n_frames_test=2, n_frames_window_test=20)
    x, sizes = O.pad_frames(O.preprocess(hp, frames), 32)
    with torch.no_grad():
        ref_feats = O.swinv2(sd, "detr.backbone.0.backbone", x, O.SwinHyper())
        masks = O.padding_masks(3, [tuple(f.shape[-2:]) for f in ref_feats], (8, 16, 32), sizes)
        xr, mr, pr, shapes = O.input_proj_and_flatten(sd, hp, ref_feats, masks)
        enc_r = O.encoder(sd, hp, xr, mr, pr, shapes)
    for o, r in zip(feats, ref_feats):
        assert maxdiff(o.permute(0, 3, 1, 2).cpu(), r) < 1e-3 * max(1.0, float(r.abs().max()))
    assert maxdiff(enc.cpu(), enc_r) < 1e-3 * max(1.0, float(enc_r.abs().max()))
    with torch.no_grad():
        res = model([{"image": frames, "height": 120, "width": 216}])
    assert len(res["pred_masks"]) == len(res["pred_scores"]) >= 10 and res["pred_masks"][0].shape == (3, 120, 216)


@pytest.mark.parametrize("tag", ["multi", "single"])
def test_coco_image_branch_vs_reference(tag):
    """COCO single-image branch (MDQE.forward with a COCO test set -> inference_image) vs the reference run
    (fixture coco_image_small: 3-frame pseudo clip, MULTI_CLS_ON on / off)."""
    import dataclasses
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    fx = Fixture("coco_image_small")
    sd = fx.state()
    cfg = dataclasses.replace(small_cfg(apply_cls_thres=fx.f("thr"), n_frames_test=3, n_frames_window_test=4, n_max_inst=40),
                              is_coco=True, multi_cls=(tag == "multi"))
    model = MDQE(cfg, state_dict=sd, backbone_fn=tiny_pyramid_gpu(sd)).eval()
    with torch.no_grad():
        res = model([{"image": list(fx.t("frames")), "height": 100, "width": 140}])[0]["instances"]
    assert res.pred_classes.tolist() == fx.t(f"{tag}::pred_classes").tolist()
    assert maxdiff(res.scores.cpu(), fx.t(f"{tag}::scores")) < 1e-3
    ref_m = fx.t(f"{tag}::pred_masks")
    got_m = res.pred_masks.cpu()
    assert got_m.dtype == torch.bool and got_m.shape == ref_m.shape
    assert float((got_m != ref_m).float().mean()) < 1e-3
    assert maxdiff(res.pred_boxes.tensor.cpu(), fx.t(f"{tag}::pred_boxes")) <= 2.0
    assert res.image_size == (100, 140) and len(res) == len(ref_m)
