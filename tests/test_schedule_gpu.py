"""The stream schedule, the early-mask path and the two fp32 GEMM kernel forms must not change results."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _small_model():
    from mdqe_cvpr2023_amd.config import PRESETS
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    from mdqe_cvpr2023_amd.params import random_state
    import dataclasses
    cfg = dataclasses.replace(PRESETS["R50_ovis_360"], n_frames_window_test=6)
    return cfg, MDQE(cfg, state_dict=random_state(cfg, seed=3)).eval()


def _video(L, h=96, w=160):
    from bench import synth_video
    return synth_video(0, L, seed=1, h=h, w=w, n_obj=4)


def _same(a, b):
    assert a["image_size"] == b["image_size"]
    assert a["pred_labels"] == b["pred_labels"]
    assert torch.allclose(torch.tensor(a["pred_scores"]), torch.tensor(b["pred_scores"]), atol=0, rtol=0)
    assert len(a["pred_masks"]) == len(b["pred_masks"])
    for x, y in zip(a["pred_masks"], b["pred_masks"]):
        assert x.dtype == torch.bool and x.shape == y.shape and bool((x == y).all())


@pytest.mark.parametrize("frame_batch", [3, 6, 0])
def test_two_stream_schedule_is_bit_identical_to_single_stream(frame_batch):
    """Frame stages on their own stream + double-buffered caches + early masks vs everything on one stream: the kernels and
    their inputs are the same, so every output bit is."""
    cfg, model = _small_model()
    frames = _video(17).cuda()
    model.frame_batch = frame_batch
    inp = [{"image": frames, "height": 96, "width": 160}]
    model.overlap_streams = True
    a = model(inp)
    a2 = model(inp)                                    # a second call reuses streams / rings / pinned blocks
    model.overlap_streams = False
    b = model(inp)
    _same(a, b)
    _same(a, a2)
    assert len(a["pred_masks"]) > 0 and a["pred_masks"][0].shape == (17, 96, 160)


@pytest.mark.parametrize("frame_batch,dec_batch,cache_frames", [(3, 0, 0), (3, 0, 1), (2, 7, 1), (6, 20, 0), (0, 40, 0), (4, 1, 1)])
def test_linear_frame_cache_segments_and_decoder_batch_do_not_change_a_bit(frame_batch, dec_batch, cache_frames, monkeypatch):
    """Round 5's frame cache: one linear buffer per kind, results stored in place, the decoder batch independent of the pass size
    (`dec_batch` clips per batch).  A video longer than the cache continues in a second buffer (segments; MDQE_CACHE_FRAMES=1 forces the
    smallest capacity the schedule allows, so a 41-frame video switches buffers several times).  Same kernels on the same rows whatever
    the grouping: every output bit equals the default schedule's, and the clip results equal it clip by clip."""
    cfg, model = _small_model()
    frames = _video(41).cuda()
    inp = [{"image": frames, "height": 96, "width": 160}]
    ref = model(inp)
    clips = model.clip_schedule(41, cfg.n_frames_test, cfg.clip_stride)
    with torch.no_grad():
        ref_clips = [(s, e, l, {k: v.clone() for k, v in r.items() if torch.is_tensor(v)}) for s, e, l, r in model.iter_clip_results(frames, clips, 0)]
    model.frame_batch, model.dec_batch = frame_batch, dec_batch
    if cache_frames:
        monkeypatch.setenv("MDQE_CACHE_FRAMES", str(cache_frames))
    out = model(inp)
    _same(ref, out)
    _same(out, model(inp))
    with torch.no_grad():
        got = list(model.iter_clip_results(frames, clips, 0))
    assert [(s, e, l) for s, e, l, _ in got] == [(s, e, l) for s, e, l, _ in ref_clips]
    for (_, _, _, a), (_, _, _, b) in zip(ref_clips, got):
        for k in ("scores", "pred_classes", "cls_probs", "query_embeds", "pred_masks"):
            assert torch.equal(a[k], b[k]), k
    if dec_batch:                                       # the planner waited for dec_batch clips: the first batch holds at least that many
        first = next(i for i, (_, _, _, r) in enumerate(got) if r["batch_end"]) + 1
        assert first >= min(dec_batch, len(clips) - 1)


def test_pinned_mask_buffers_are_pooled_but_never_shared():
    """`MDQE.pinned_mask_buffer` (round 5): the per-track pinned read-back buffers of a video are kept by the model and re-issued once
    nothing references them any more.  Two requests while the first is held give DIFFERENT storage (a video's tracks never share a
    buffer); a buffer whose holders -- the merger and the result views handed to the caller -- are gone comes back; results of two
    consecutive calls held at the same time do not alias."""
    cfg, model = _small_model()
    a = model.pinned_mask_buffer((5, 8, 12))
    b = model.pinned_mask_buffer((5, 8, 12))
    assert a.is_pinned() and b.is_pinned() and a.data_ptr() != b.data_ptr()
    pa = a.data_ptr()
    va = a.view(torch.bool)[:3]                        # what a result holds: a view
    del a
    c = model.pinned_mask_buffer((5, 8, 12))
    assert c.data_ptr() not in (pa, b.data_ptr())      # `va` still holds the first buffer
    del va
    d = model.pinned_mask_buffer((5, 8, 12))
    assert d.data_ptr() == pa                          # ... now it is free again
    frames = _video(14).cuda()
    inp = [{"image": frames, "height": 96, "width": 160}]
    r1 = model(inp)
    r2 = model(inp)                                    # r1 is still alive: r2 must not write into its buffers
    _same(r1, r2)
    assert len(r1["pred_masks"]) > 1
    ptrs = [m.data_ptr() for m in r1["pred_masks"]] + [m.data_ptr() for m in r2["pred_masks"]]
    assert len(set(ptrs)) >= len(set(m.data_ptr() for m in r1["pred_masks"])) * 2
    del r2
    r3 = model(inp)
    _same(r1, r3)


def test_streams_are_shared_by_models_and_touched_once():
    """One set of pipeline streams per device and process: a second model instance runs on the SAME stream objects (a second set would take
    another deal of hardware queues), `touch_streams()` is idempotent, and none of it changes a bit."""
    cfg, m1 = _small_model()
    _, m2 = _small_model()
    frames = _video(10).cuda()
    inp = [{"image": frames, "height": 96, "width": 160}]
    a = m1(inp)
    m2.touch_streams(); m2.touch_streams()
    b = m2(inp)
    _same(a, b)
    for name in ("_frame_stream", "_copy_stream", "_trk_stream", "_work_stream"):
        s1, s2 = getattr(m1, name), getattr(m2, name)
        assert s1 is not None and s1 is s2, name


def test_early_masks_equal_the_direct_path():
    """ClipMerger with n_frames (masks produced per window into pinned memory) vs without (one pass at the end)."""
    cfg, model = _small_model()
    frames = _video(15).cuda()
    with torch.no_grad():
        geo = model.engine.geometry(96, 160)
        ms = cfg.match_stride
        clips = model.clip_schedule(15, cfg.n_frames_test, cfg.clip_stride)
        res = list(model.iter_clip_results(frames, clips, 0))
        torch.cuda.synchronize()
        outs = []
        for n_frames in (15, None):
            outs.append(model.merge_clips(iter(res), (96, 160), (80, 130), (geo.Hp // ms, geo.Wp // ms), n_frames=n_frames))
    _same(outs[0], outs[1])
    assert outs[0]["pred_masks"][0].shape == (15, 80, 130)


def test_fp32_gemm_kernel_forms_agree():
    """K-step 32 (gemm.hip) and K-step 16 (gemm_k16.hip) are the same fmaf chains in the same K order: bitwise equal."""
    from mdqe_cvpr2023_amd import ops
    from mdqe_cvpr2023_amd._lib import lib
    g = torch.Generator().manual_seed(9)
    try:
        for (M, N, K, tile) in ((3000, 640, 256, 1), (777, 130, 48, 3), (5000, 64, 160, 2), (20000, 256, 1024, 0)):
            x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
            r = torch.randn(M, N, generator=g).cuda()
            outs = []
            for v in (0, 1):
                lib.mdqe_debug_gemm_variant(v)
                outs.append(ops.linear(x, w, b, act="gelu", residual=r, tile=tile))
            assert torch.equal(outs[0], outs[1]), (M, N, K, tile)
            ref = F.gelu(x.double() @ w.double().t() + b.double()) + r.double()
            assert float((outs[1].double() - ref).abs().max() / ref.abs().max()) < 2e-6
        xi = torch.randn(3, 24, 40, 128, generator=g).cuda(); wc = (torch.randn(96, 3, 3, 128, generator=g) / 34).cuda(); bc = torch.randn(96, generator=g).cuda()
        outs = []
        for v in (0, 1):
            lib.mdqe_debug_gemm_variant(v)
            outs.append(ops.conv2d_nhwc(xi, wc, bc, 2, 1, act="relu"))
        assert torch.equal(outs[0], outs[1])
    finally:
        lib.mdqe_debug_gemm_variant(2)


def test_const_weight_registry():
    from mdqe_cvpr2023_amd import ops
    w = torch.randn(256, 256, device="cuda")
    ops.const_weight(w)
    assert ops._wsplit(w) is not None
    assert ops._wsplit(w.view(256, 16, 16)) is not None          # a full reshaped view is the same weight
    assert ops._wsplit(w[:128]) is None                          # a partial view at the same address is not
    small = ops.const_weight(torch.randn(64, 256, device="cuda"))
    assert ops._wsplit(small) is None                            # too narrow for the 128-wide tiles: left alone
    ptr = w.data_ptr()
    del w
    import gc
    gc.collect()
    assert ptr not in ops._split                                  # the planes are released with the weight


def test_rle_output_equals_encoding_of_the_dense_masks():
    """model.rle_output: run boundaries from the device (the dense masks are never written) -> the same COCO RLE strings as
    encoding the dense output on the host (oracle restatement of cocoapi rleEncode/rleToString), and they decode back to
    the dense masks; the drop-in result writer produces the reference's record layout."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import rle_oracle as RO
    from mdqe_cvpr2023_amd import rle as R
    cfg, model = _small_model()
    frames = _video(14).cuda()
    inp = [{"image": frames, "height": 90, "width": 150, "video_id": 7, "length": 14}]
    dense = model(inp)
    model.rle_output = True
    out = model(inp)
    model.rle_output = False
    assert "pred_masks" not in out and out["pred_labels"] == dense["pred_labels"] and out["pred_scores"] == dense["pred_scores"]
    assert len(out["pred_rles"]) == len(dense["pred_masks"]) > 0
    for rl, dm in zip(out["pred_rles"], dense["pred_masks"]):
        assert len(rl) == dm.shape[0] == 14
        for f in range(14):
            ref = RO.encode(dm[f].numpy())
            assert rl[f]["size"] == ref["size"] == [90, 150]
            assert rl[f]["counts"].encode() == ref["counts"]
            assert (RO.rle_decode(RO.rle_from_string(rl[f]["counts"].encode()), 90, 150) == dm[f].numpy()).all()
    recs = R.instances_to_coco_json_video(inp, out)
    recs_dense = R.instances_to_coco_json_video(inp, dense)
    assert recs == recs_dense and recs[0]["video_id"] == 7 and set(recs[0]) == {"video_id", "score", "category_id", "segmentations"}


def test_resize_on_device_equals_resizing_on_the_host_first():
    """model.resize_on_device: native-size frames in, the mapper's ResizeShortestEdge applied on the GPU == feeding frames the
    oracle (== Pillow) resized on the host; height/width default to the ORIGINAL size like the mapper's dataset dict."""
    import dataclasses
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import resize_oracle as RO
    cfg, model = _small_model()
    model.cfg = cfg = dataclasses.replace(cfg, min_size_test=48, max_size_test=100)
    big = _video(9, h=120, w=200)                                       # -> 48 x 80
    small = torch.stack([torch.from_numpy(RO.resize_bilinear_u8(f.permute(1, 2, 0).numpy(), 48, 80)).permute(2, 0, 1) for f in big])
    ref = model([{"image": small.cuda(), "height": 120, "width": 200}])
    model.resize_on_device = True
    got = model([{"image": big.cuda()}])
    model.resize_on_device = False
    _same(got, ref)
    assert got["image_size"] == (120, 200)


@pytest.mark.parametrize("geo_cache", [16, 1])
def test_videos_of_different_sizes_back_to_back(geo_cache, monkeypatch):
    """An eval loop feeds one model videos of varying resolution and length: the per-resolution constants, cache rings,
    workspaces and pinned blocks of one video must not leak into the next.  A (96x160) -> B (72x104, other length, other
    output size) -> A again: both A results and a fresh model's are bit-identical -- also when the per-resolution cache
    holds a single entry and every switch rebuilds the constants."""
    import mdqe_cvpr2023_amd.engine as E
    monkeypatch.setattr(E, "GEO_CACHE", geo_cache)
    cfg, model = _small_model()
    A = [{"image": _video(11).cuda(), "height": 96, "width": 160}]
    B = [{"image": _video(7, h=72, w=104).cuda(), "height": 144, "width": 208}]
    a1 = model(A)
    b1 = model(B)
    a2 = model(A)
    _same(a1, a2)
    assert len(model.engine._geo) == min(2, geo_cache)
    assert b1["image_size"] == (144, 208) and all(m.shape == (7, 144, 208) for m in b1["pred_masks"])
    _, fresh = _small_model()
    _same(fresh(B), b1)
    _same(fresh(A), a1)


def test_forward_stream_equals_separate_calls():
    """forward_stream (the next video's first pass queued under the current video's tail) over videos of different length and
    resolution -- one of them shorter than a clip, one a single pass -- against one forward() per video: bit-identical, in order."""
    cfg, model = _small_model()
    vids = [[{"image": _video(11).cuda(), "height": 96, "width": 160}],
            [{"image": _video(2, h=72, w=104).cuda(), "height": 72, "width": 104}],
            [{"image": _video(7, h=72, w=104).cuda(), "height": 144, "width": 208}],
            [{"image": _video(25).cuda(), "height": 96, "width": 160}]]
    model.frame_batch = 6
    ref = [model(v) for v in vids]
    got = list(model.forward_stream(iter(vids)))
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        _same(a, b)
    assert list(model.forward_stream(iter([]))) == []
    assert torch.is_grad_enabled()                     # the generator leaves no no_grad / autocast state behind
    # a bad input in the stream surfaces at ITS position: the video before it is still delivered
    bad = [{"image": _video(5).cuda(), "height": 96, "width": 160}, {"image": _video(5).cuda(), "height": 96, "width": 160}]
    gen = model.forward_stream(iter([vids[0], bad, vids[2]]))
    _same(next(gen), ref[0])
    with pytest.raises(RuntimeError):
        next(gen)


def test_clip_stride_larger_than_the_clip_vs_oracle():
    """CLIP_STRIDE > SAMPLING_FRAME_NUM_TEST (the reference then skips frames, mdqe/mdqe.py:308-312): a clip that starts past
    the frames already cached must read ITS frames (frame passes of 5: clip (12,15) starts beyond the first two passes' carry)."""
    import dataclasses
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mdqe_oracle as O
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    from mdqe_cvpr2023_amd.params import random_state
    kw = dict(enc_layers=1, dec_layers=2, n_frames=3, num_classes=12, num_queries=16, query_embed_dim=16)
    ev = dict(n_frames_test=3, n_frames_window_test=6, n_max_inst=40, apply_cls_thres=0.12, clip_stride=5)
    cfg = MDQEConfig(**kw, **ev)
    sd = random_state(cfg, seed=11)
    frames = list(_video(14, h=64, w=96))
    model = MDQE(cfg, state_dict=sd).eval()
    ref_trace = []
    with torch.no_grad():
        ref = O.inference_vis(sd, O.Hyper(**kw, **ev), frames, lambda im: O.resnet(sd, "detr.backbone.0.backbone", im, 50),
                              out_size=(64, 96), trace=ref_trace)
    assert [c["frame_idx"][0] for c in ref_trace] == [0, 5, 10]
    for fb in (4, 2, 0):
        model.frame_batch = fb
        trace = []
        with torch.no_grad():
            out = model.inference_vis([{"image": frames, "height": 64, "width": 96}], trace=trace)
        assert len(trace) == len(ref_trace)
        for a, b in zip(trace, ref_trace):
            assert a["pred_masks"].shape == b["pred_masks"].shape
            assert float((a["pred_masks"].cpu() - b["pred_masks"]).abs().max()) < 1e-3
        assert out["pred_labels"] == ref["pred_labels"]
        got, want = torch.stack(out["pred_masks"]), torch.stack(ref["pred_masks"])
        assert got.shape == want.shape and (got != want).float().mean() < 2e-3


def test_decode_ahead_of_trailing_short_clips_is_bit_identical():
    """MDQE.decode_ahead: the short clips at the end of a video (no new frame pass needed) are decoded on an auxiliary stream beside the
    last full-length group instead of after its tracker run -- same kernels, same inputs: identical outputs; videos whose length leaves
    1, 2 and 3 trailing short clips (stride 1: one; stride 2 / other lengths: none), repeated calls."""
    cfg, model = _small_model()
    for L in (17, 9, 6, 5):
        frames = _video(L).cuda()
        inp = [{"image": frames, "height": 96, "width": 160}]
        model.decode_ahead = True
        a = model(inp)
        a2 = model(inp)
        model.decode_ahead = False
        b = model(inp)
        model.decode_ahead = True
        _same(a, b)
        _same(a, a2)
