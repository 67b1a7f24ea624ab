"""GPU parity: HIP MSDA (through the C ABI) vs the oracle and the reference-generated goldens."""
import pytest
import torch

import mdqe_oracle as O
from _golden import Fixture, maxdiff, record_margin

REFG = "native MSDA op vs the REFERENCE's own output (golden fixtures)"

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda().contiguous()


def run_hip(value, shapes, starts, loc, attn):
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    sh = torch.as_tensor(shapes, dtype=torch.int64).cuda()
    st = torch.as_tensor(starts, dtype=torch.int64).cuda()
    return MSDA.ms_deform_attn_forward(dev(value), sh, st, dev(loc), dev(attn), 64).cpu()


def test_reference_known_answer_float():
    """mdqe/models/ops/test.py:46-60 asks rtol 1e-2 / atol 1e-3; we hold 1e-6."""
    fx = Fixture("msda_reftest")
    out = run_hip(fx.t("float_value"), fx.shapes(), fx.t("level_start").tolist(), fx.t("float_loc"), fx.t("float_attn"))
    ref = fx.t("float_out")
    assert torch.allclose(out, ref, rtol=1e-2, atol=1e-3)
    assert record_margin(REFG, "ops/test.py known-answer recipe, float", maxdiff(out, ref), 1.0, 1e-6) < 1e-6


@pytest.mark.parametrize("case", ["enc", "dec_spatial", "dec_temporal", "swin_d24", "tiny_d8"])
def test_golden_cases(case):
    fx = Fixture("msda_cases")
    out = run_hip(fx.t(f"{case}::value"), fx.shapes(f"{case}::shapes"), fx.t(f"{case}::level_start").tolist(),
                  fx.t(f"{case}::loc"), fx.t(f"{case}::attn"))
    assert record_margin(REFG, "msda_cases: " + case, maxdiff(out, fx.t(f"{case}::out")), 1.0, 2e-5) < 2e-5     # O(1) values, fp32


def test_edge_cases():
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    sh = torch.tensor([[4, 6]], dtype=torch.int64).cuda()
    st = torch.tensor([0], dtype=torch.int64).cuda()
    # empty query set
    out = MSDA.ms_deform_attn_forward(torch.zeros(1, 24, 2, 4).cuda(), sh, st, torch.zeros(1, 0, 2, 1, 1, 2).cuda(),
                                      torch.zeros(1, 0, 2, 1, 1).cuda(), 64)
    assert out.shape == (1, 0, 8)
    # all samples outside -> exact zeros; exactly-on-border samples follow the reference's strict inequalities
    v = torch.randn(1, 24, 2, 4)
    loc = torch.tensor([[-0.5, 0.5], [1.5, 0.5], [0.5, -0.3], [0.5, 1.4]]).view(1, 4, 1, 1, 1, 2).repeat(1, 1, 2, 1, 1, 1)
    at = torch.ones(1, 4, 2, 1, 1)
    out = MSDA.ms_deform_attn_forward(v.cuda(), sh, st, loc.cuda(), at.cuda(), 64).cpu()
    assert torch.equal(out, torch.zeros_like(out))
    loc = torch.tensor([[0.0, 0.0], [1.0, 1.0], [-1.0 / 12, 0.5], [1.0 + 1.0 / 12, 0.5]]).view(1, 4, 1, 1, 1, 2).repeat(1, 1, 2, 1, 1, 1)
    out = MSDA.ms_deform_attn_forward(v.cuda(), sh, st, loc.cuda().contiguous(), at.cuda(), 64).cpu()
    ref = O.msda_forward(v, [(4, 6)], [0], loc, at)
    assert maxdiff(out, ref) < 1e-6
    # contract errors are RuntimeErrors like the reference's AT_ASSERTM
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v, sh, st, loc, at, 64)             # CPU tensor
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v.cuda().transpose(1, 2), sh, st, loc.cuda(), at.cuda(), 64)   # non-contiguous


def test_full_size_properties():
    """R50_ovis_360 encoder shape (B=4,S=Q=5100,M=8,D=32,L=P=4): the oracle is too slow at this size, so
    check size-independent properties: linearity in value, partition of unity (constant value map +
    interior samples -> sum of weights), and agreement with the oracle on a slice."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    g = torch.Generator().manual_seed(0)
    shapes = [(48, 80), (24, 40), (12, 20), (6, 10)]
    S = sum(h * w for h, w in shapes)
    starts = [0, 3840, 4800, 5040]
    B, M, D, L, P = 4, 8, 32, 4, 4
    sh = torch.tensor(shapes, dtype=torch.int64).cuda()
    st = torch.tensor(starts, dtype=torch.int64).cuda()
    v1 = torch.randn(B, S, M, D, generator=g)
    v2 = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, S, 1, 1, 1, 2, generator=g) + 0.08 * torch.randn(B, S, M, L, P, 2, generator=g)
    at = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P)
    f = lambda v: MSDA.ms_deform_attn_forward(v.cuda(), sh, st, loc.cuda(), at.cuda(), 64)
    o1, o2, o12 = f(v1), f(v2), f(2 * v1 - 3 * v2)
    assert maxdiff((2 * o1 - 3 * o2).cpu(), o12.cpu()) < 1e-4
    # constant map, interior samples: every sample returns the constant, weights sum to 1
    loc_in = 0.25 + 0.5 * torch.rand(B, S, M, L, P, 2, generator=g)
    oc = MSDA.ms_deform_attn_forward(torch.full((B, S, M, D), 1.5).cuda(), sh, st, loc_in.cuda(), at.cuda(), 64).cpu()
    assert maxdiff(oc, torch.full_like(oc, 1.5)) < 1e-5
    # oracle on a slice of queries
    ref = O.msda_forward(v1[:1], shapes, starts, loc[:1, :600], at[:1, :600])
    assert maxdiff(o1[:1, :600].cpu(), ref) < 2e-5


@pytest.mark.parametrize("shapes,B,Q", [
    ([(48, 80), (24, 40), (12, 20), (6, 10)], 2, 5100),       # R50_ovis_360 encoder: the two coarse levels fit the S/16 budget
    ([(48, 80), (24, 40), (12, 20), (6, 10)], 17, 196),       # decoder-like: few queries per element; 17 elements: XCD-aware order
    ([(8, 12), (4, 6), (2, 3), (1, 2)], 3, 128),              # tiny maps: three levels staged
    ([(10, 10), (10, 10), (10, 10), (10, 10)], 2, 400),       # not a pyramid: only the last level fits
    ([(6, 10), (12, 20), (24, 40), (48, 80)], 2, 300),        # fine level LAST: nothing fits, every corner through the texture path
    ([(100, 160), (50, 80), (25, 40), (13, 20)], 1, 300),     # 21260 tokens: the budget is capped (960 rows), only the coarsest level fits
])
def test_op_with_coarse_levels_in_lds_equals_the_gather_form(shapes, B, Q):
    """msda_fwd_v3_kernel (the native op with the coarse levels of a (batch element, head) staged in LDS; the level table is device
    memory, so every block sizes the staging area itself) against msda_fwd_v2_kernel: equal bits, whatever suffix of levels fits."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    from mdqe_cvpr2023_amd._lib import lib
    g = torch.Generator().manual_seed(Q)
    M, D, L, P = 8, 32, 4, 4
    S = sum(h * w for h, w in shapes)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    sh, st = torch.tensor(shapes, dtype=torch.int64).cuda(), torch.tensor(starts, dtype=torch.int64).cuda()
    v = torch.randn(B, S, M, D, generator=g).cuda()
    loc = (0.5 + 1.6 * (torch.rand(B, Q, M, L, P, 2, generator=g) - 0.5))
    loc[:, ::5] = torch.round(loc[:, ::5] * 8) / 8                             # pixel borders
    at = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P)
    outs = []
    try:
        for staged in (0, 1):
            lib.mdqe_debug_msda_op_staged(staged)
            outs.append(MSDA.ms_deform_attn_forward(v, sh, st, loc.cuda(), at.cuda(), 64))
    finally:
        lib.mdqe_debug_msda_op_staged(1)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    ref = O.msda_forward(v[:1].cpu(), shapes, starts, loc[:1, :64], at[:1, :64])
    assert maxdiff(outs[1][:1, :64].cpu(), ref) < 2e-5


@pytest.mark.parametrize("case", ["enc", "dec", "swin_d24", "tiny_d8"])
def test_backward_vs_reference_core_gradients(case):
    """ms_deform_attn_backward (HIP, fp32, float atomics) vs float64 autograd through the reference's own PyTorch core --
    the check the reference's ops/test.py:63-86 applies to its CUDA kernel -- and vs the oracle."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    fx = Fixture("msda_backward")
    g = lambda k: fx.t(f"{case}::{k}")
    args = [g(k).cuda() for k in ("value", "shapes", "level_start", "loc", "attn", "grad_out")]
    gv, gl, ga = MSDA.ms_deform_attn_backward(*args, 64)
    ov, ol, oa = O.msda_backward(g("value"), fx.shapes(f"{case}::shapes"), [int(v) for v in g("level_start")], g("loc"), g("attn"),
                                 g("grad_out"))
    for got, name, orc in ((gv, "grad_value", ov), (gl, "grad_loc", ol), (ga, "grad_attn", oa)):
        ref = g(name)
        scale = max(1.0, float(ref.abs().max()))
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert maxdiff(got.cpu(), ref) <= 3e-5 * scale, (case, name, maxdiff(got.cpu(), ref), scale)
        assert maxdiff(got.cpu(), orc) <= 5e-5 * scale


def test_backward_contract():
    """Inputs are borrowed, outputs are new tensors, grad_value is fully overwritten (zero where nothing samples), wrong
    inputs raise like the forward."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    B, S, M, D, Q, L, P = 1, 12, 2, 4, 3, 1, 2
    value = torch.randn(B, S, M, D).cuda()
    shapes = torch.tensor([[3, 4]]).cuda(); st = torch.tensor([0]).cuda()
    loc = torch.full((B, Q, M, L, P, 2), 5.0).cuda()            # every sample outside the map
    attn = torch.rand(B, Q, M, L, P).cuda()
    go = torch.randn(B, Q, M * D).cuda()
    v0 = value.clone()
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, st, loc, attn, go, 64)
    assert torch.equal(value, v0)
    assert float(gv.abs().max()) == 0 and float(gl.abs().max()) == 0 and float(ga.abs().max()) == 0
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(value, shapes, st, loc, attn, go.cpu(), 64)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(value, shapes, st, loc.transpose(1, 2), attn, go, 64)


# ---- float64: the reference dispatches float AND double (ms_deform_attn_cuda.cu:64,134), and two of the three checks of its own test script
# (mdqe/models/ops/test.py) run in double.  Twins of all three checks, through the same drop-in module.

class MSDeformAttnFunction(torch.autograd.Function):
    """What mdqe/models/ops/functions/ms_deform_attn_func.py:22-42 wraps around the extension's two exports (kept by the reference; the
    drop-in replaces only the extension module it imports)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
        ctx.im2col_step = im2col_step
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return output

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
        value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights = ctx.saved_tensors
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


def test_reference_known_answer_double():
    """check_forward_equal_with_pytorch_double (ops/test.py:32-44): the op in float64 on the script's own draw (seed 3) against the
    reference's PyTorch core run in double (fixture `double_out`, produced by executing the reference) -- `torch.allclose` with its
    default tolerances, as the script asks; and 1e-12 on top."""
    fx = Fixture("msda_reftest")
    sh = torch.as_tensor(fx.shapes(), dtype=torch.int64).cuda()
    st = fx.t("level_start").cuda()
    args = [fx.t(k).double().cuda() for k in ("double_value", "double_loc", "double_attn")]
    out = MSDeformAttnFunction.apply(args[0], sh, st, args[1], args[2], 2).detach().cpu()
    ref = fx.t("double_out")
    assert out.dtype == torch.float64 and ref.dtype == torch.float64
    assert torch.allclose(out, ref)
    assert maxdiff(out, ref) < 1e-12 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("case", ["enc", "dec_spatial", "dec_temporal", "swin_d24", "tiny_d8"])
def test_golden_cases_double(case):
    """The path-shaped cases in float64: against the oracle run in double (1e-12) and against the reference-run fp32 golden (2e-5)."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    fx = Fixture("msda_cases")
    v, loc, at = (fx.t(f"{case}::{k}").double() for k in ("value", "loc", "attn"))
    shapes, starts = fx.shapes(f"{case}::shapes"), fx.t(f"{case}::level_start")
    out = MSDA.ms_deform_attn_forward(v.cuda(), torch.as_tensor(shapes, dtype=torch.int64).cuda(), starts.cuda(), loc.cuda(), at.cuda(), 64).cpu()
    ref = O.msda_forward(v, shapes, starts.tolist(), loc, at)
    assert out.dtype == torch.float64
    assert maxdiff(out, ref) < 1e-12 * max(1.0, float(ref.abs().max()))
    assert maxdiff(out.float(), fx.t(f"{case}::out")) < 2e-5


@pytest.mark.parametrize("case", ["enc", "dec", "swin_d24", "tiny_d8"])
def test_backward_double_vs_float64_autograd(case):
    """ms_deform_attn_backward in float64 (double atomics) against float64 autograd through the oracle's restatement of the reference's
    PyTorch core (itself pinned to the reference): 1e-10; and against the reference-run gradients of the fixture."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    fx = Fixture("msda_backward")
    g = lambda k: fx.t(f"{case}::{k}")
    v, loc, at, go = (g(k).double() for k in ("value", "loc", "attn", "grad_out"))
    shapes, starts = fx.shapes(f"{case}::shapes"), g("level_start")
    gv, gl, ga = MSDA.ms_deform_attn_backward(v.cuda(), torch.as_tensor(shapes, dtype=torch.int64).cuda(), starts.cuda(), loc.cuda(), at.cuda(), go.cuda(), 64)
    with torch.enable_grad():
        v_, l_, a_ = (t.clone().requires_grad_(True) for t in (v, loc, at))
        O.msda_forward(v_, shapes, starts.tolist(), l_, a_).backward(go)
    for got, want, name in ((gv, v_.grad, "grad_value"), (gl, l_.grad, "grad_loc"), (ga, a_.grad, "grad_attn")):
        scale = max(1.0, float(want.abs().max()))
        assert got.dtype == torch.float64 and got.shape == want.shape
        assert maxdiff(got.cpu(), want) <= 1e-10 * scale, (case, name, maxdiff(got.cpu(), want))
        assert maxdiff(got.cpu().float(), g(name)) <= 3e-5 * scale


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])
def test_gradient_numerical_like_the_reference(channels):
    """check_gradient_numerical (ops/test.py:63-86), every channel count of its `__main__`: `torch.autograd.gradcheck` of the autograd
    function around the two exports, double inputs, the script's shapes (N, M = 1, 2; Lq, L, P = 2, 2, 2; levels 6x4 and 3x2)."""
    from torch.autograd import gradcheck
    N, M = 1, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    level_start_index = torch.cat((shapes.new_zeros((1, )), shapes.prod(1).cumsum(0)[:-1]))
    S = sum([(H * W).item() for H, W in shapes])
    torch.manual_seed(3)
    value = torch.rand(N, S, M, channels).cuda() * 0.01
    sampling_locations = torch.rand(N, Lq, M, L, P, 2).cuda()
    attention_weights = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
    attention_weights /= attention_weights.sum(-1, keepdim=True).sum(-2, keepdim=True)
    im2col_step = 2
    value.requires_grad = True
    sampling_locations.requires_grad = True
    attention_weights.requires_grad = True
    # (the three biggest channel counts use gradcheck's fast mode -- one random projection instead of a perturbation per element,
    # 10^5 forward launches otherwise; the small ones run the full element-wise check the script runs)
    assert gradcheck(MSDeformAttnFunction.apply, (value.double(), shapes, level_start_index, sampling_locations.double(), attention_weights.double(), im2col_step),
                     fast_mode=channels > 100)


def test_dtype_contract():
    """float and double only, all floating operands alike -- anything else raises RuntimeError like AT_DISPATCH_FLOATING_TYPES."""
    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    sh = torch.tensor([[3, 4]], dtype=torch.int64).cuda(); st = torch.tensor([0], dtype=torch.int64).cuda()
    v = torch.randn(1, 12, 2, 4).cuda(); loc = torch.rand(1, 3, 2, 1, 2, 2).cuda(); at = torch.rand(1, 3, 2, 1, 2).cuda()
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v.half(), sh, st, loc.half(), at.half(), 64)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v.double(), sh, st, loc, at, 64)                       # mixed
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(v.double(), sh, st, loc.double(), at.double(), torch.randn(1, 3, 8).cuda(), 64)
    o32 = MSDA.ms_deform_attn_forward(v, sh, st, loc, at, 64)
    o64 = MSDA.ms_deform_attn_forward(v.double(), sh, st, loc.double(), at.double(), 64)
    assert o32.dtype == torch.float32 and o64.dtype == torch.float64 and maxdiff(o32.cpu().double(), o64.cpu()) < 1e-5
