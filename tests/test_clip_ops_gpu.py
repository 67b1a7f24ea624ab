"""GPU: the per-clip batch kernels (csrc/clip_ops.hip) one by one against plain torch restatements of the reference lines
they replace (transformer_dec.py:111-145,374-376,473-503; mdqe/mdqe.py:368-428), and the batched inference_clip against
the oracle's per-clip `inference_clip` on crafted batches: ties at the thresholds, blank masks, near-duplicate embeddings,
clips that keep one / all / more than max_keep queries."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mdqe_oracle as O
from _golden import maxdiff

pytestmark = pytest.mark.gpu


def test_assoc_and_gather_init_vs_torch():
    from mdqe_cvpr2023_amd import ops
    g = torch.Generator().manual_seed(0)
    for (nb, E, C, T, NF, starts) in ((14, 64, 256, 4, 9, [0, 3, 5]), (4, 16, 32, 3, 5, [2, 0]), (14, 64, 192, 2, 3, [1])):
        Q = nb * nb
        emb = torch.randn(NF, Q, E, generator=g)
        content = torch.randn(NF, Q, C, generator=g)
        coords = torch.rand(NF, Q, 2, generator=g)
        fidx = torch.tensor([[a + t for t in range(T)] for a in starts], dtype=torch.int32)
        ct = int((T - 1) / 2)
        idx = ops.clip_assoc(emb.cuda(), fidx.cuda(), ct, 2.5, nb)
        x, ref, xi = ops.clip_gather_init(content.cuda(), coords.cuda(), fidx.cuda(), idx, ct)
        relpos = O.query_relpos_grid(nb)
        for b, a in enumerate(starts):
            fr = list(range(a, a + T))
            q_r, c_r, _ = O.inter_frame_query_association(content[fr], coords[fr], emb[fr], relpos, 5.0)    # (the oracle halves the window itself)
            got = x.view(len(starts), T, Q, C)[b].cpu()
            assert torch.equal(got, q_r)
            assert torch.equal(ref.view(len(starts), T, Q, 4)[b, ..., :2].cpu(), c_r)
            assert bool((ref.view(len(starts), T, Q, 4)[b, ..., 2:].cpu() == 0.1).all())
            assert torch.equal(xi.view(len(starts), Q, C)[b].cpu(), q_r[ct])
    # T == 1: identity
    fidx = torch.tensor([[1], [0]], dtype=torch.int32).cuda()
    x, ref, xi = ops.clip_gather_init(content.cuda(), coords.cuda(), fidx, None, 0)
    assert torch.equal(x.view(2, Q, C)[0].cpu(), content[1]) and torch.equal(xi.view(2, Q, C)[1].cpu(), content[0])


def test_assoc_near_ties_follow_the_fp32_softmax_of_the_reference():
    """`softmax(-2).argmax(-2)` (transformer_dec.py:142-143) picks the FIRST index whose fp32 softmax value equals the largest: a cell
    1-3 ulps below the maximum can tie with it through the rounding of exp() and of the division by the column sum.  Similarities are
    made exact (centre embedding = a unit vector, so sim(q) = the first component of q's embedding); all cells but the two contenders
    sit at -100 (exp underflows to 0), so the column sum has two terms and no summation-order freedom.  The kernel must agree with
    torch's CPU softmax on every case."""
    from mdqe_cvpr2023_amd import ops
    nb, E, T = 4, 16, 2
    Q = nb * nb
    cases = []
    for best in (0.3, 0.11, -0.27, 0.9, 0.6, 0.051, -0.73, 1.7, 3.0):
        for ulps in (1, 2, 3, 5):
            for first in (2, 9):                                             # the lower contender comes BEFORE the maximum (index 12)
                lo = np.float32(best)
                for _ in range(ulps):
                    lo = np.nextafter(lo, np.float32(-np.inf), dtype=np.float32)
                cases.append((np.float32(best), lo, first))
    emb = torch.zeros(2 * len(cases), Q, E)
    for i, (best, lo, first) in enumerate(cases):
        emb[2 * i, :, 0] = 1.0                                               # centre frame: every query is the unit vector e0
        emb[2 * i + 1, :, 0] = -100.0
        emb[2 * i + 1, 12, 0] = float(best)
        emb[2 * i + 1, first, 0] = float(lo)
    fidx = torch.tensor([[2 * i, 2 * i + 1] for i in range(len(cases))], dtype=torch.int32)
    idx = ops.clip_assoc(emb.cuda(), fidx.cuda(), 0, 100.0, nb).cpu()        # window wide open: every cell admissible
    n_tie = 0
    for i, (best, lo, first) in enumerate(cases):
        sim = emb[2 * i + 1, :, 0].view(Q, 1).expand(Q, Q).contiguous()       # [q, k]
        want = sim.softmax(-2).argmax(-2)                                     # the reference's expression on the CPU
        n_tie += int(want[0]) == first
        assert torch.equal(idx[i, 1].long(), want), (float(best), float(lo), first, idx[i, 1, :4].tolist(), want[:4].tolist())
    assert n_tie > 0                                                          # the construction does produce softmax ties below the maximum


def test_box_refine_time_fuse_add_vs_torch():
    from mdqe_cvpr2023_amd import ops
    g = torch.Generator().manual_seed(1)
    Bc, T, Q, C = 3, 4, 50, 64
    delta = torch.randn(Bc * T * Q, 4, generator=g)
    prev = torch.rand(Bc * T * Q, 4, generator=g)
    prev[0] = torch.tensor([0.0, 1.0, 1e-7, 0.5])                       # the clamps of inverse_sigmoid
    for (t0, t1) in ((0, 4), (1, 5), (0, 2)):
        boxes, ibox = ops.box_refine(delta.cuda(), prev.cuda(), Bc, T, Q, t0, t1)
        ref = (delta + O.inverse_sigmoid(prev)).sigmoid()
        assert maxdiff(boxes.cpu(), ref) < 2e-6
        b = O.box_cxcywh_to_xyxy(ref.view(Bc, T, Q, 4).transpose(1, 2)[:, :, t0:t1]).clamp(0, 1)
        b = torch.cat([b[..., :2].min(-2)[0], b[..., 2:].max(-2)[0]], -1)
        assert maxdiff(ibox.cpu(), O.box_xyxy_to_cxcywh(b).reshape(-1, 4)) < 2e-6
    w = torch.randn(Bc * T * Q, 1, generator=g) * 3
    x = torch.randn(Bc * T * Q, C, generator=g)
    pos = torch.randn(Bc * Q, C, generator=g)
    out, out2 = ops.time_fuse(w.cuda(), x.cuda(), Bc, T, Q, pos=pos.cuda())
    ref = (torch.softmax(w.view(Bc, T, Q, 1), 1) * x.view(Bc, T, Q, C)).sum(1).reshape(Bc * Q, C)
    assert maxdiff(out.cpu(), ref) < 2e-6 and maxdiff(out2.cpu(), ref + pos) < 2e-6
    assert maxdiff(ops.time_fuse(w.cuda(), x.cuda(), Bc, T, Q).cpu(), ref) < 2e-6
    a, b2 = torch.randn(77, 96, generator=g), torch.randn(77, 200, generator=g)
    assert torch.equal(ops.add_rows(a.cuda(), b2.cuda()[:, 8:104]).cpu(), a + b2[:, 8:104])


def _crafted_batch(g, B, Q, K, C, M, T, Hm, Wm):
    cls = torch.rand(B, Q, K, generator=g) * 0.3
    emb = torch.randn(B, Q, C, generator=g)
    coef = torch.tanh(torch.randn(B, Q, M, generator=g))
    mf = torch.relu(torch.randn(B + T - 1, Hm, Wm, M, generator=g))
    return cls, emb, coef, mf


@pytest.mark.parametrize("shape", [(5, 16, 5, 32, 32, 3, 16, 24), (3, 196, 25, 256, 32, 4, 24, 40), (2, 196, 25, 192, 24, 2, 16, 28),
                                   (2, 36, 7, 64, 8, 5, 12, 20)])
def test_batched_inference_clip_vs_oracle(shape):
    """The whole of inference_clip for a batch of clips against the oracle clip by clip."""
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.engine import Engine
    B, Q, K, C, M, T, Hm, Wm = shape
    g = torch.Generator().manual_seed(sum(shape))
    cls, emb, coef, mf = _crafted_batch(g, B, Q, K, C, M, T, Hm, Wm)
    thr = 0.25
    # clip 0: several queries well above the threshold, two of them near-duplicates, one with a blank mask
    cls[0, 3, 1] = 0.9; cls[0, 5, 2] = 0.8; cls[0, 7, 0] = 0.7; cls[0, 9, 1] = 0.6; cls[0, 11, 2] = 0.5
    emb[0, 5] = emb[0, 3] * 1.5 + 1e-3 * torch.randn(C, generator=g)        # cosine > 0.99 with query 3 -> dropped
    coef[0, 7] = -coef[0, 7].abs()                                           # features >= 0 -> all logits <= 0: blank
    # clip 1: nothing reaches the threshold -> the best one alone (min(thr, top))
    cls[1] *= 0.5
    # clip 2: two identical masks (NMS suppresses the lower-scored one)
    if B > 2:
        cls[2, 1, 0] = 0.95; cls[2, 2, 1] = 0.85
        coef[2, 2] = coef[2, 1]
    cfg = MDQEConfig(backbone="custom", hidden_dim=C, num_classes=K, num_queries=Q, n_frames=T, n_frames_test=T, apply_cls_thres=thr,
                     detections_per_image=2 if Q == 36 else 15)
    eng = Engine.__new__(Engine)
    eng.cfg, eng.dev = cfg, torch.device("cuda")
    outs = {"cls": cls.cuda(), "mask_coeff": coef.cuda(), "query_embed": emb.cuda()}
    mfd = mf.cuda()
    res = eng.inference_clips(outs, mfd, list(range(B)), T)
    res_v = eng.inference_clips(outs, [mfd[b:b + T] for b in range(B)])                       # the list-of-views form
    hp = O.Hyper(hidden_dim=C, num_classes=K, n_frames=T, n_frames_test=T, apply_cls_thres=thr, detections_per_image=cfg.detections_per_image)
    for b in range(B):
        ref = O.inference_clip(hp, {"cls": cls[b:b + 1], "mask_coeff": coef[b:b + 1], "query_embed": emb[b:b + 1]},
                               mf[b:b + T].permute(3, 0, 1, 2))
        for r in (res[b], res_v[b]):
            assert r["pred_masks"].shape == ref["pred_masks"].shape, (b, r["pred_masks"].shape, ref["pred_masks"].shape)
            assert r["pred_classes"].tolist() == ref["pred_classes"].tolist()
            assert maxdiff(r["pred_masks"].cpu(), ref["pred_masks"]) < 1e-4
            assert maxdiff(r["scores"].cpu(), ref["scores"]) < 1e-5 and maxdiff(r["cls_probs"].cpu(), ref["cls_probs"]) < 1e-5
            assert torch.equal(r["query_embeds"].cpu(), ref["query_embeds"])
            assert np.array_equal(r["host"]["scores"], r["scores"].cpu().numpy())
            assert np.array_equal(r["host"]["query_embeds"], r["query_embeds"].cpu().numpy())
    assert res[1]["scores"].numel() == 1
    assert 5 not in [int(q) for q in (emb[0] == res[0]["query_embeds"].cpu()[:, None]).all(-1).nonzero()[:, 1]]   # the duplicate is gone


def test_batched_inference_clip_is_deterministic_and_handles_empty_batches():
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.engine import Engine
    g = torch.Generator().manual_seed(5)
    B, Q, K, C, M, T, Hm, Wm = 70, 16, 5, 32, 32, 2, 8, 12                  # more than 64 clips: two launches of the batch kernels
    cls, emb, coef, mf = _crafted_batch(g, B, Q, K, C, M, T, Hm, Wm)
    cls[:, 0, 0] = 0.9; cls[:, 1, 1] = 0.8
    cfg = MDQEConfig(backbone="custom", hidden_dim=C, num_classes=K, num_queries=Q, n_frames=T, n_frames_test=T, apply_cls_thres=0.25)
    eng = Engine.__new__(Engine)
    eng.cfg, eng.dev = cfg, torch.device("cuda")
    outs = {"cls": cls.cuda(), "mask_coeff": coef.cuda(), "query_embed": emb.cuda()}
    a = eng.inference_clips(outs, mf.cuda(), list(range(B)), T)
    b = eng.inference_clips(outs, mf.cuda(), list(range(B)), T)
    hp = O.Hyper(hidden_dim=C, num_classes=K, n_frames=T, n_frames_test=T, apply_cls_thres=0.25)
    for i, (x, y) in enumerate(zip(a, b)):
        assert all(torch.equal(x[k], y[k]) for k in ("scores", "pred_classes", "cls_probs", "pred_masks", "query_embeds"))
        ref = O.inference_clip(hp, {"cls": cls[i:i + 1], "mask_coeff": coef[i:i + 1], "query_embed": emb[i:i + 1]}, mf[i:i + T].permute(3, 0, 1, 2))
        assert x["pred_classes"].tolist() == ref["pred_classes"].tolist() and maxdiff(x["pred_masks"].cpu(), ref["pred_masks"]) < 1e-4


# ---- round 3: fewer launches between the decoder's GEMMs ---------------------------------------------------------------------------

@pytest.mark.parametrize("M,N,K,cols", [(29008, 768, 256, 512), (7252, 384, 256, 384), (5000, 576, 192, 384), (37, 768, 256, 512), (70000, 768, 256, 512)])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_linear_side_is_the_projection_of_x_plus_pos(M, N, K, cols, precision):
    """mdqe_gemm_nt_side_f32: x W^T + b with the rank-4 side term on the first `cols` columns == the projection of (x + pos) for
    pos = Linear(4 -> K)(box) on those columns and of x alone on the rest (float64 checker); every tile form the dispatcher picks
    (M from 37 to 70 000 rows) and the split-precision mode, where the term is a pass of its own."""
    from mdqe_cvpr2023_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g); box = torch.rand(M, 4, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    Pw = torch.randn(K, 4, generator=g); Pb = torch.randn(K, generator=g)
    sw = torch.zeros(N, 4, dtype=torch.float64); sw[:cols] = W[:cols].double() @ Pw.double()
    bb = b.double().clone(); bb[:cols] += W[:cols].double() @ Pb.double()
    ops.set_gemm_precision(precision)
    try:
        Wd = ops.const_weight(W.cuda())
        out = ops.linear_side(x.cuda(), Wd, bb.float().cuda(), box.cuda(), sw.float().cuda(), cols).cpu().double()
    finally:
        ops.set_gemm_precision("f32")
    pos = box.double() @ Pw.double().t() + Pb.double()
    ref = torch.cat([(x.double() + pos) @ W[:cols].double().t() + b[:cols].double(), x.double() @ W[cols:].double().t() + b[cols:].double()], 1)
    assert float((out - ref).abs().max() / ref.abs().max()) < (2e-6 if precision == "f32" else 8e-6)


@pytest.mark.parametrize("Bc,T,Q", [(37, 4, 196), (3, 4, 196), (5, 1, 49), (2, 5, 100)])
def test_box_head_refine_equals_the_two_kernel_form(Bc, T, Q):
    """One kernel for bbox_embed's last Linear(256 -> 4) + refinement + clip boxes: the bits of rows_dot_kernel<4> followed by box_refine_kernel."""
    from mdqe_cvpr2023_amd import ops
    g = torch.Generator().manual_seed(Bc * T + Q)
    h = torch.randn(Bc * T * Q, 256, generator=g).cuda(); w = (torch.randn(4, 256, generator=g) / 16).cuda(); b = torch.randn(4, generator=g).cuda()
    prev = torch.rand(Bc * T * Q, 4, generator=g).cuda()
    prev[::17] = 0.0; prev[5::23] = 1.0                                      # inverse_sigmoid's clamps
    ct = int((T - 1) / 2)
    t0, t1 = max(ct - 1, 0), ct + 4
    b1, i1 = ops.box_head_refine(h, w, b, prev, Bc, T, Q, t0, t1)
    b2, i2 = ops.box_refine(ops.linear(h, w, b), prev, Bc, T, Q, t0, t1)
    assert torch.equal(b1, b2) and torch.equal(i1, i2)
    ref = torch.sigmoid((h.double() @ w.double().t() + b.double()) + torch.logit(prev.double().clamp(0, 1), eps=1e-5))
    assert float((b1.double() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("Bc,T,Q", [(37, 4, 196), (3, 2, 196), (4, 1, 49), (2, 5, 100)])
def test_time_fuse_dot_equals_the_two_kernel_form(Bc, T, Q):
    """One kernel for time_weights Linear(256 -> 1) + softmax over the frames + weighted sum: the bits of rows_dot_kernel<1> + time_fuse_kernel."""
    from mdqe_cvpr2023_amd import ops
    g = torch.Generator().manual_seed(Bc + T + Q)
    xw = torch.randn(Bc * T * Q, 256, generator=g).cuda(); src = torch.randn(Bc * T * Q, 256, generator=g).cuda()
    wt = (torch.randn(1, 256, generator=g) / 4).cuda(); bt = torch.randn(1, generator=g).cuda()
    a = ops.time_fuse_dot(xw, wt, bt, src, Bc, T, Q)
    b = ops.time_fuse(ops.linear(xw, wt, bt), src, Bc, T, Q)
    assert torch.equal(a, b)
    p = torch.softmax((xw.double() @ wt.double().t() + bt.double()).view(Bc, T, Q), 1)
    ref = (p[..., None] * src.double().view(Bc, T, Q, 256)).sum(1).view(Bc * Q, 256)
    assert float((a.double() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("config", ["R50_ovis_360", "swinl_ovis"])
def test_decoder_with_folded_positions_equals_the_materialised_form(config):
    """engine.DEC_FUSED: `(x + pos) W^T` as `x W^T + box (W P)^T` (one [3C, C] product for q, k, v; no position tensor; the fused box-head
    and time-fuse kernels) against the form that materialises pos and projects q/k and v apart -- the same function up to fp32
    reassociation: 2e-5 of the output scale on random weights, both hidden sizes (256: all fused kernels; 192: the side-term GEMM only)."""
    from mdqe_cvpr2023_amd import engine as E
    from mdqe_cvpr2023_amd.config import PRESETS
    from mdqe_cvpr2023_amd.params import random_state
    cfg = PRESETS[config]
    sd = random_state(cfg, seed=5, remove_zero_init_trap=True)
    eng = E.Engine(cfg, sd)
    C, T, Q = cfg.hidden_dim, cfg.n_frames_test, cfg.n_bins ** 2
    geo = eng.geometry(96, 160)
    g = torch.Generator().manual_seed(1)
    F_ = T + 3
    cache = {"coords": torch.rand(F_, Q, 2, generator=g).cuda(), "content": torch.randn(F_, Q, C, generator=g).cuda(),
             "emb": torch.randn(F_, Q, cfg.query_embed_dim, generator=g).cuda(),
             "vals": torch.randn(F_, geo.N, eng.P.n_val * C, generator=g).cuda()}
    outs = []
    try:
        for flag in (True, False):
            E.DEC_FUSED = flag
            with torch.no_grad():
                outs.append(eng.decode_clips(cache, [0, 1, 2, 3], T, geo))
    finally:
        E.DEC_FUSED = True
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), k


def test_decoder_instance_chain_on_a_side_stream_is_bit_identical():
    """engine.DEC_TWO_STREAMS: the instance-level chain of layer l on a side stream beside the box level of layer l + 1 -- the same
    kernels on the same inputs, so every output bit is the same; repeated calls reuse the stream and the allocator's blocks."""
    from mdqe_cvpr2023_amd import engine as E
    from mdqe_cvpr2023_amd.config import PRESETS
    from mdqe_cvpr2023_amd.params import random_state
    cfg = PRESETS["R50_ovis_360"]
    eng = E.Engine(cfg, random_state(cfg, seed=6, remove_zero_init_trap=True))
    C, T, Q = cfg.hidden_dim, cfg.n_frames_test, cfg.n_bins ** 2
    geo = eng.geometry(96, 160)
    g = torch.Generator().manual_seed(2)
    F_ = T + 8
    cache = {"coords": torch.rand(F_, Q, 2, generator=g).cuda(), "content": torch.randn(F_, Q, C, generator=g).cuda(),
             "emb": torch.randn(F_, Q, cfg.query_embed_dim, generator=g).cuda(),
             "vals": torch.randn(F_, geo.N, eng.P.n_val * C, generator=g).cuda()}
    outs = []
    try:
        for flag in (True, False, True):
            E.DEC_TWO_STREAMS = flag
            with torch.no_grad():
                for _ in range(3):                                  # back-to-back calls: blocks recycled across the two streams
                    o = eng.decode_clips(cache, list(range(9)), T, geo)
                    junk = torch.randn(1 << 22, device="cuda")      # allocator pressure between calls
                    del junk
            torch.cuda.synchronize()
            outs.append(o)
    finally:
        E.DEC_TWO_STREAMS = True
    for k in outs[0]:
        assert torch.isfinite(outs[0][k]).all()
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k]), k
