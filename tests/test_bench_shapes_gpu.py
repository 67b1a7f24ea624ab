"""GPU: correctness AT THE BENCH'S OWN LAUNCH SHAPES.

The full-size parity tests hold 3-6 frames (one frame pass, <= 30 600 GEMM rows) to the oracle; `bench.py` runs 20/40/40/20-frame passes
(204 000-row GEMMs on the 128x128 tile, a 2.5-GB decoder-value product, 40-frame MSDA launches with the XCD-aware block order, 37-clip
decoder batches, three ring buffers, tapered passes).  Here:

 (a) the bench's own video through the default schedule against the same video in 6-frame passes (the size held to the oracle):
     labels, scores and masks must be IDENTICAL -- every kernel picks its arithmetic from (N, K), never from the number of rows;
 (b) the large-M GEMM / LayerNorm-epilogue / conv / fused-MSDA launches against float64 products and per-frame slices;
 (c) short seeds of the four differential fuzzers of tools/ (the long runs are under profiles/).
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench_model(config):
    import bench
    from mdqe_cvpr2023_amd.config import PRESETS
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    from mdqe_cvpr2023_amd.params import random_state
    cfg = PRESETS[config]
    fh, fw = {"R50_ovis_360": (360, 640), "R50_ovis_720": (640, 1138), "swinl_ovis": (480, 853)}[config]
    sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
    model = MDQE(cfg, state_dict=sd).eval()
    bench.calibrate_synthetic_scores(model, sd, cfg, fh, fw)
    return bench, cfg, model, fh, fw


def _same(a, b):
    assert a["image_size"] == b["image_size"]
    assert a["pred_labels"] == b["pred_labels"], (a["pred_labels"], b["pred_labels"])
    assert a["pred_scores"] == b["pred_scores"]
    assert len(a["pred_masks"]) == len(b["pred_masks"])
    for x, y in zip(a["pred_masks"], b["pred_masks"]):
        assert x.dtype == torch.bool and x.shape == y.shape and torch.equal(x, y)


@pytest.mark.parametrize("config,frames,small", [("R50_ovis_360", 120, 6), ("R50_ovis_720", 60, 3), ("swinl_ovis", 40, 3)])
def test_bench_video_default_schedule_equals_oracle_sized_passes(config, frames, small):
    """The exact video, weights, calibration and call of `bench.py --config <config>` (pinned host frames, H2D inside the call, default
    pass sizes -- 20/40/40/20 at 360p --, look-ahead 2, tapered passes) against the same call with `frame_batch = small` frames per
    pass and no look-ahead games (the per-pass size tests/test_fullsize_gpu.py holds to the oracle): bit-identical outputs."""
    bench, cfg, model, fh, fw = _bench_model(config)
    video = bench.synth_video(0, frames, seed=0, h=fh, w=fw).pin_memory()
    inp = [{"image": list(video), "height": fh, "width": fw}]
    with torch.no_grad():
        a = model(inp)
        a2 = model(inp)                                   # (ring / stream / pinned-block reuse, as the bench's steps 2..K)
        model.frame_batch = small
        model.taper_passes = False
        b = model(inp)
    assert len(a["pred_scores"]) >= 1 and a["pred_masks"][0].shape == (frames, fh, fw)
    _same(a, a2)
    _same(a, b)


def _f64_linear(x, w, b):
    return x.double() @ w.double().t() + b.double()


@pytest.mark.parametrize("N,K,act", [(1024, 256, "gelu"), (3072, 256, None), (256, 1024, None), (640, 256, None)])
def test_gemm_at_the_40_frame_pass_rows(N, K, act):
    """M = 204 000 rows (40 frames x 5100 encoder tokens): FFN1 + GELU (N = 1024), the 12-layer decoder value projection (N = 3072, a
    2.5-GB output), FFN2 (K = 1024) and the encoder's value/offset/weight projection (N = 640) -- against a float64 product on the GPU."""
    from mdqe_cvpr2023_amd import ops
    M = 204000
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    out = ops.linear(x, w, b, act=act)
    worst = 0.0
    for r0 in range(0, M, 51000):                         # float64 checker in four row blocks (bounded memory)
        ref = _f64_linear(x[r0:r0 + 51000], w, b)
        if act == "gelu":
            ref = torch.nn.functional.gelu(ref)
        worst = max(worst, float((out[r0:r0 + 51000].double() - ref).abs().max() / ref.abs().max()))
        del ref
    assert worst < 2e-6, worst
    # the same rows through a small launch (another tile form): identical bits -- a row's arithmetic does not depend on M
    small = ops.linear(x[100000:105100].contiguous(), w, b, act=act)
    assert torch.equal(small, out[100000:105100])


@pytest.mark.parametrize("K", [256, 1024])
def test_layernorm_epilogue_gemm_at_the_40_frame_pass_rows(K):
    """`x = LayerNorm(x + linear(h))` as one kernel (64x256 tile, statistics in the epilogue) on 204 000 rows, in place over the
    residual, against float64; and against the 6-frame launch of the same rows (identical bits)."""
    from mdqe_cvpr2023_amd import ops
    M, N = 204000, 256
    g = torch.Generator(device="cuda").manual_seed(K)
    h = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    gam = torch.rand(N, device="cuda", generator=g) + 0.5
    bet = torch.randn(N, device="cuda", generator=g)
    x = res.clone()
    ops.linear_ln(h, w, b, x, gam, bet, out=x)
    worst = 0.0
    for r0 in range(0, M, 51000):
        ref = torch.nn.functional.layer_norm(_f64_linear(h[r0:r0 + 51000], w, b) + res[r0:r0 + 51000].double(), (N,), gam.double(), bet.double(), 1e-5)
        worst = max(worst, float((x[r0:r0 + 51000].double() - ref).abs().max()))
        del ref
    assert worst < 2e-5, worst
    r0, r1 = 5100 * 17, 5100 * 23                         # six frames' rows as their own launch
    y = res[r0:r1].clone()
    ops.linear_ln(h[r0:r1].contiguous(), w, b, y, gam, bet, out=y)
    assert torch.equal(y, x[r0:r1])


def test_res2_convs_on_a_40_frame_pass():
    """res2 on 40 x 96 x 160 pixels (614 400 rows): the 3x3 64 -> 64 implicit-GEMM conv + ReLU, the 1x1 64 -> 256, and conv3 + projection
    shortcut as ONE product (CAT mode) -- against float64 (nine shifted products for the 3x3)."""
    from mdqe_cvpr2023_amd import ops
    NI, H, W, C = 40, 96, 160, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(NI, H, W, C, device="cuda", generator=g)
    w3 = torch.randn(C, 3, 3, C, device="cuda", generator=g) / 24.0
    b3 = torch.randn(C, device="cuda", generator=g)
    y = ops.conv2d_nhwc(x, w3, b3, 1, 1, act="relu")
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
    ref = b3.double().expand(NI, H, W, C).clone()
    for kh in range(3):
        for kw in range(3):
            ref += xp[:, kh:kh + H, kw:kw + W] @ w3[:, kh, kw].double().t()
    ref.relu_()
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    y6 = ops.conv2d_nhwc(x[11:17].contiguous(), w3, b3, 1, 1, act="relu")                 # six frames as their own launch: same bits
    assert torch.equal(y6, y[11:17])
    del xp, ref
    w1 = torch.randn(256, C, device="cuda", generator=g) / 8.0
    b1 = torch.randn(256, device="cuda", generator=g)
    ws = torch.randn(256, C, device="cuda", generator=g) / 8.0
    bs = torch.randn(256, device="cuda", generator=g)
    z = ops.linear(y.view(-1, C), w1, b1)
    refz = _f64_linear(y.view(-1, C), w1, b1)
    assert float((z.double() - refz).abs().max() / refz.abs().max()) < 2e-6
    cat = ops.linear_cat2(y, x, 1, torch.cat([w1, ws], 1).contiguous(), b1 + bs, act="relu")
    refc = (refz + _f64_linear(x.view(-1, C), ws, bs)).relu_()
    assert float((cat.view(-1, 256).double() - refc).abs().max() / refc.abs().max()) < 2e-6


@pytest.mark.parametrize("shapes,B", [([(48, 80), (24, 40), (12, 20), (6, 10)], 40), ([(80, 144), (40, 72), (20, 36), (10, 18)], 20)])
def test_encoder_msda_at_pass_size_equals_per_frame_launches(shapes, B):
    """The fused encoder MSDA on a whole pass (B = 40 at 360p / 20 at 640p: the XCD-aware block order of the B >= 16 branch, coarse levels in
    LDS) against the same frames launched one at a time: identical bits, every row written."""
    from mdqe_cvpr2023_amd import ops
    M, D, L, P = 8, 32, 4, 4
    C = M * D
    Nq = sum(h * w for h, w in shapes)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    levels = ([s[0] for s in shapes], [s[1] for s in shapes], starts)
    nq = 2 * M * L * P
    g = torch.Generator(device="cuda").manual_seed(B)
    proj = torch.randn(B * Nq, C + 3 * M * L * P, device="cuda", generator=g)
    proj[:, C:C + nq] *= 2.0
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(a) + 0.5) / a, (torch.arange(c) + 0.5) / c, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                     for a, c in shapes]).float().cuda().contiguous()
    out = torch.full((B * Nq, C), float("nan"), device="cuda")
    ops.msda_fused(proj[:, :C], proj[:, C:C + nq], proj[:, C + nq:], ref, levels, B, Nq, M, D, L, P, mode=0, v_brows=Nq, out=out)
    assert torch.isfinite(out).all()
    for b in (0, 1, 7, 8, 15, 16, B // 2, B - 1):
        p1 = proj[b * Nq:(b + 1) * Nq]
        o1 = ops.msda_fused(p1[:, :C], p1[:, C:C + nq], p1[:, C + nq:], ref, levels, 1, Nq, M, D, L, P, mode=0, v_brows=Nq)
        assert torch.equal(o1, out[b * Nq:(b + 1) * Nq]), b


@pytest.mark.parametrize("tool,args", [("fuzz_msda.py", ["21"]), ("fuzz_msda_fused.py", ["20"]), ("fuzz_inference_clip.py", ["20"]),
                                       ("fuzz_tracker.py", ["20", "--gpu"]), ("fuzz_pipeline.py", ["8"])])
def test_fuzz_seeds(tool, args):
    """The first seeds of tools/fuzz_*.py (the long runs on every round's final code: profiles/rNN_fuzz_*.txt -- 60 pipeline cases, 0 mismatches; 8 of
    them here: the suite has a 900-s budget on the driver's box and the per-seed cost is 8-12 s): native MSDA op vs the oracle, fused MSDA forms vs each
    other, batched inference_clip vs the oracle's per-clip restatement, the tracker on the HIP bank vs the oracle's, the whole driver vs the
    oracle's at random small configurations.  One child process per tool (they are scripts that exit non-zero on a mismatch)."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "oracle") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.setdefault("OMP_NUM_THREADS", "16")                      # the oracle's GEMMs: 16 threads, not the 256 logical CPUs a GPU box shows (conftest.py)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
