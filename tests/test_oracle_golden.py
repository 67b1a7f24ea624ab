"""CPU tests: the oracle (oracle/mdqe_oracle.py) against outputs of the reference itself.

Fixtures were produced by oracle/make_golden.py executing the reference python in the build
container.  Tolerances: both sides are fp32 CPU torch, so 1e-5 absolute on O(1) activations
(the north-star bar for the product is 1e-3)."""
import numpy as np
import pytest
import torch

import mdqe_oracle as O
from _golden import Fixture, maxdiff

TOL = 2e-5


def test_msda_reference_known_answer():
    """The reference's own test recipe (mdqe/models/ops/test.py:21-60): double allclose default,
    float rtol 1e-2 / atol 1e-3 -- we hold 1e-7."""
    fx = Fixture("msda_reftest")
    shapes = fx.shapes()
    st = fx.t("level_start").tolist()
    for tag, tol in (("double", 1e-12), ("float", 1e-7)):
        v, loc, at, ref = fx.t(f"{tag}_value"), fx.t(f"{tag}_loc"), fx.t(f"{tag}_attn"), fx.t(f"{tag}_out")
        if tag == "double":
            v, loc, at = v.double(), loc.double(), at.double()
        out = O.msda_forward(v, shapes, st, loc, at)
        assert out.shape == ref.shape
        assert torch.allclose(out, ref) and maxdiff(out, ref) < tol


@pytest.mark.parametrize("case", ["enc", "dec_spatial", "dec_temporal", "swin_d24", "tiny_d8"])
def test_msda_cases(case):
    fx = Fixture("msda_cases")
    out = O.msda_forward(fx.t(f"{case}::value"), fx.shapes(f"{case}::shapes"), fx.t(f"{case}::level_start").tolist(),
                         fx.t(f"{case}::loc"), fx.t(f"{case}::attn"))
    assert maxdiff(out, fx.t(f"{case}::out")) < 5e-6


@pytest.mark.parametrize("case", ["enc", "dec", "swin_d24", "tiny_d8"])
def test_msda_backward_cases(case):
    """Oracle gradients vs the float64 gradients of the reference's own PyTorch core (fixture made by make_golden.py)."""
    fx = Fixture("msda_backward")
    g = lambda k: fx.t(f"{case}::{k}")
    for dt, tol in ((torch.float64, 1e-6), (torch.float32, 2e-5)):
        gv, gl, ga = O.msda_backward(g("value").to(dt), fx.shapes(f"{case}::shapes"), [int(v) for v in g("level_start")],
                                     g("loc").to(dt), g("attn").to(dt), g("grad_out").to(dt))
        for got, name in ((gv, "grad_value"), (gl, "grad_loc"), (ga, "grad_attn")):
            ref = g(name)
            assert maxdiff(got, ref) <= tol * max(1.0, float(ref.abs().max())), (case, name, dt)


def test_misc():
    fx = Fixture("misc_small")
    assert maxdiff(O.aligned_bilinear(fx.t("ab_in"), 4), fx.t("ab_out4")) < 1e-6
    assert maxdiff(O.aligned_bilinear(fx.t("ab_in"), 2), fx.t("ab_out2")) < 1e-6
    assert maxdiff(O.pos_sine(fx.t("pos_mask"), 16), fx.t("pos_out")) < 1e-6
    assert maxdiff(O.inverse_sigmoid(fx.t("invsig_in")), fx.t("invsig_out")) == 0.0


def small_hyper(**kw):
    d = dict(hidden_dim=256, nheads=8, enc_layers=2, dec_layers=2, n_levels=4, enc_points=4, dec_points=4,
             n_frames=3, num_classes=5, num_queries=16, query_embed_dim=16, window_inter_frame_asso=5)
    d.update(kw)
    return O.Hyper(**d)


def test_encoder_and_mask_head():
    fx = Fixture("encoder_small")
    sd, hp = fx.state(), small_hyper()
    feats = [fx.t(f"feat{i}") for i in range(3)]
    sizes = [tuple(r) for r in fx.z["image_sizes"]]
    masks = O.padding_masks(feats[0].shape[0], [tuple(f.shape[-2:]) for f in feats], (8, 16, 32), sizes)
    for i, m in enumerate(masks):
        assert torch.equal(m, fx.t(f"mask{i}"))
        assert maxdiff(O.pos_sine(m, 128), fx.t(f"pos{i}")) < 1e-6
    x, mask, pos, shapes = O.input_proj_and_flatten(sd, hp, feats, masks)
    assert shapes == fx.shapes()
    assert torch.equal(mask, fx.t("enc_masks"))
    layers = []
    enc = O.encoder(sd, hp, x, mask, pos, shapes, collect=layers)
    for i, l in enumerate(layers):
        assert maxdiff(l, fx.t(f"enc_layer{i}")) < TOL
    assert maxdiff(enc, fx.t("enc_out")) < TOL
    mf = O.mask_head(sd, fx.t("enc_out"), shapes)
    assert mf.shape == fx.t("mask_feats").shape
    assert maxdiff(mf, fx.t("mask_feats")) < TOL


def test_encoder_layer_256():
    fx = Fixture("enc_layer256")
    sd = {k.replace("layer.", "L.encoder.layers.0."): v for k, v in fx.state().items()}
    # identity final norm so O.encoder returns the layer output
    sd["L.encoder.norm.weight"] = torch.ones(256)
    sd["L.encoder.norm.bias"] = torch.zeros(256)
    hp = O.Hyper(enc_layers=1)
    layers = []
    O.encoder(sd, hp, fx.t("x"), fx.t("mask"), fx.t("pos"), fx.shapes(), p="L", collect=layers)
    assert maxdiff(layers[0], fx.t("out")) < TOL


@pytest.mark.parametrize("T", [3, 2, 1])
def test_decoder(T):
    enc_fx, fx = Fixture("encoder_small"), Fixture("decoder_small")
    sd, hp = enc_fx.state(), small_hyper()
    enc, mask, shapes = enc_fx.t("enc_out")[:T], enc_fx.t("enc_masks")[:T], enc_fx.shapes()
    dbg = {}
    out = O.transformer_dec(sd, hp, enc, mask, shapes, dbg=dbg)
    assert maxdiff(dbg["coords"], fx.t(f"T{T}::coords")) < 1e-6
    assert maxdiff(dbg["query0"], fx.t(f"T{T}::query0")) < TOL
    xs, xi, bx = fx.t(f"T{T}::x_stack"), fx.t(f"T{T}::x_inst_stack"), fx.t(f"T{T}::boxes_stack")
    for i in range(hp.dec_layers):
        assert maxdiff(dbg["x"][i], xs[i + 1]) < TOL
        assert maxdiff(dbg["x_inst"][i], xi[i + 1]) < TOL
        assert maxdiff(dbg["boxes"][i], bx[i + 1]) < TOL
    for k in ("cls", "mask_coeff", "query_embed"):
        assert maxdiff(out[k], fx.t(f"T{T}::{k}")) < TOL


def test_decoder_fixed_grid_matches_formula():
    """The decoder's `sampling_offsets` buffer is not in the synthetic manifest: the oracle derives it
    from the formula (ms_deform_attn.py:81-87).  Passing test_decoder proves equality; spot-check values."""
    g = O.msda_dir_grid(8, 4, 4)
    assert g.shape == (8, 4, 4, 2)
    assert torch.allclose(g[0, 0, :, 0], torch.tensor([2., 4., 6., 8.]))
    assert torch.allclose(g[2, 1, 3], torch.tensor([0., 8.]), atol=1e-6)


def _clip_inputs(fx, i):
    return {k: fx.t(f"clip{i}::{k}") for k in ("cls", "mask_coeff", "query_embed")}, fx.t(f"clip{i}::mask_feats")


def test_inference_clip():
    fx = Fixture("video_small")
    hp = small_hyper(apply_cls_thres=fx.f("thr"), n_frames_test=3, n_frames_window_test=4, n_max_inst=40)
    for i in range(fx.i("n_clips")):
        out, mf = _clip_inputs(fx, i)
        r = O.inference_clip(hp, out, mf)
        assert torch.equal(r["pred_classes"], fx.t(f"clip{i}::pred_classes"))
        for k in ("scores", "cls_probs", "pred_masks", "query_embeds"):
            assert r[k].shape == fx.t(f"clip{i}::{k}").shape
            assert maxdiff(r[k], fx.t(f"clip{i}::{k}")) < TOL


@pytest.mark.parametrize("fixture", ["tracker_seq", "tracker_long"])
def test_tracker_sequence(fixture):
    """tracker_long (round 5): 44 frames, six window flushes; an object leaves for 11 frames -- longer than a window and than the short
    memory -- and gets its OLD id back from the long memory alone, another leaves for longer than the long memory and returns as a NEW id
    (mdqe/tracking/OverTracker.py:65-90,124-134)."""
    fx = Fixture(fixture)
    hp = O.Hyper(hidden_dim=fx.i("E"), num_classes=fx.i("K"), n_frames_test=fx.i("T"), n_frames_window_test=fx.i("WIN"),
                 n_max_inst=fx.i("MAXI"), apply_cls_thres=fx.f("THR"), clip_stride=1)
    trk = O.Tracker(hp, tuple(int(v) for v in fx.z["HW"]))
    saved, L = 0, None
    n = fx.i("n_clips")
    for i in range(n):
        clip = {k: fx.t(f"clip{i}::{k}") for k in ("scores", "pred_classes", "cls_probs", "pred_masks", "query_embeds")}
        clip["frame_idx"] = fx.t(f"clip{i}::frame_idx").tolist()
        trk.update(clip)
        assert trk.num_inst == fx.i(f"clip{i}::num_inst_after")
        start = clip["frame_idx"][0]
        last = i == n - 1
        if last or (start + 1 >= hp.n_frames_window_test * (saved + 1)):
            c, m = trk.get_result(last)
            assert maxdiff(c, fx.t(f"win{saved}::cls")) < 1e-6
            assert m.shape == fx.t(f"win{saved}::masks").shape
            assert maxdiff(m, fx.t(f"win{saved}::masks")) < 1e-5
            saved += 1
    assert saved == fx.i("n_windows")


def tiny_pyramid(sd):
    import torch.nn.functional as F
    p = "detr.backbone.0.backbone."

    def fn(x):
        a = torch.tanh(F.conv2d(x, sd[p + "c1.weight"], sd[p + "c1.bias"], 8))
        b = torch.tanh(F.conv2d(a, sd[p + "c2.weight"], sd[p + "c2.bias"], 2))
        c = torch.tanh(F.conv2d(b, sd[p + "c3.weight"], sd[p + "c3.bias"], 2))
        return [a, b, c]
    return fn


@pytest.mark.parametrize("schedule", ["compute-once", "as-reference"])
def test_video_end_to_end(schedule):
    """MDQE.inference_vis end to end (mdqe/mdqe.py:291-366) incl. the short last clip and a window flush.
    'compute-once' (each frame through backbone/encoder/mask head exactly once) must equal the
    reference's per-clip window recompute."""
    fx = Fixture("video_small")
    sd = fx.state()
    hp = small_hyper(apply_cls_thres=fx.f("thr"), n_frames_test=3, n_frames_window_test=4, n_max_inst=40)
    frames = list(fx.t("frames"))
    trace = []
    out = O.inference_vis(sd, hp, frames, tiny_pyramid(sd), out_size=(120, 180), schedule=schedule, trace=trace)
    assert len(trace) == fx.i("n_clips")
    for i, c in enumerate(trace):
        assert c["pred_masks"].shape == fx.t(f"clip{i}::pred_masks").shape
        assert maxdiff(c["pred_masks"], fx.t(f"clip{i}::pred_masks")) < 1e-4
        assert maxdiff(c["scores"], fx.t(f"clip{i}::scores")) < 1e-5
    assert out["pred_labels"] == fx.t("out_labels").tolist()
    assert np.allclose(out["pred_scores"], fx.z["out_scores"], atol=1e-5)
    ref_masks = fx.t("out_masks")
    got = torch.stack(out["pred_masks"])
    assert got.shape == ref_masks.shape
    assert (got != ref_masks).float().mean() < 1e-4


def test_resnet50_structure():
    """detectron2's ResNet is third-party and absent: parity UNPINNED.  Structural checks only:
    output strides/channels as consumed at mdqe/mdqe.py:28-30 and models/mdqe.py:31-38."""
    from synth import synth_tensor
    shapes = resnet50_manifest()
    sd = {k: synth_tensor(k, s, 7) for k, s in shapes.items()}
    x = torch.randn(1, 3, 64, 96)
    outs = O.resnet(sd, "bb", x, 50)
    assert [tuple(o.shape) for o in outs] == [(1, 512, 8, 12), (1, 1024, 4, 6), (1, 2048, 2, 3)]
    assert all(torch.isfinite(o).all() for o in outs)


def resnet50_manifest(p="bb"):
    m = {}

    def conv(name, cout, cin, k):
        m[f"{name}.weight"] = (cout, cin, k, k)
        for s in ("weight", "bias", "running_mean", "running_var"):
            m[f"{name}.norm.{s}"] = (cout,)
    conv(f"{p}.stem.conv1", 64, 3, 7)
    cin = 64
    for si, nb in enumerate((3, 4, 6, 3)):
        mid, cout = 64 * 2 ** si, 256 * 2 ** si
        for b in range(nb):
            q = f"{p}.res{si + 2}.{b}"
            if b == 0:
                conv(q + ".shortcut", cout, cin, 1)
            conv(q + ".conv1", mid, cin, 1)
            conv(q + ".conv2", mid, mid, 3)
            conv(q + ".conv3", cout, mid, 1)
            cin = cout
    return m


def swin_small_state(fx):
    sd = fx.state()
    for k in fx.z.files:
        if k.startswith("ls::"):
            sd["bb." + k[4:]] = fx.t(k)
    return sd


def test_swinv2_backbone():
    """SwinTransformerV2.forward (mdqe/backbone/swin_transformer_v2.py:639-659) incl. padded windows and shift masks."""
    fx = Fixture("swin_small")
    sd = swin_small_state(fx)
    sh = O.SwinHyper(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), window_size=4)
    outs = O.swinv2(sd, "bb", fx.t("x"), sh)
    for o, name in zip(outs, ("stage3", "stage4", "stage5")):
        ref = fx.t(name)
        assert o.shape == ref.shape
        assert maxdiff(o, ref) < TOL


@pytest.mark.parametrize("tag", ["multi", "single"])
def test_coco_image_branch(tag):
    """Oracle inference_image (COCO single-image branch, MULTI_CLS_ON on/off) vs the reference run of MDQE.forward."""
    fx = Fixture("coco_image_small")
    sd = fx.state()
    hp = small_hyper(apply_cls_thres=fx.f("thr"), n_frames_test=3, n_frames_window_test=4, n_max_inst=40)
    frames = list(fx.t("frames"))
    with torch.no_grad():
        got = O.inference_image(sd, hp, frames, tiny_pyramid(sd), out_size=(100, 140), multi_cls=(tag == "multi"))
    assert maxdiff(got["cls"], fx.t(f"{tag}::cls")[0]) < 2e-5
    assert maxdiff(got["masks"], fx.t(f"{tag}::masks")[0]) < 2e-4
    assert got["pred_classes"].tolist() == fx.t(f"{tag}::pred_classes").tolist()
    assert maxdiff(got["scores"], fx.t(f"{tag}::scores")) < 2e-5
    ref_m = fx.t(f"{tag}::pred_masks")
    assert got["pred_masks"].shape == ref_m.shape and float((got["pred_masks"] != ref_m).float().mean()) < 1e-4
    assert maxdiff(got["pred_boxes"], fx.t(f"{tag}::pred_boxes")) <= 1.0
