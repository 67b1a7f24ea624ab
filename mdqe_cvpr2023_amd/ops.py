"""Host wrappers over the C ABI (include/mdqe_hip.h): torch owns memory + stream, HIP does the math.

All tensors are CUDA fp32 contiguous unless stated; images are NHWC.  Nothing here falls back to
torch arithmetic: a missing library or an unsupported shape raises.
"""
import contextlib
import os
import threading

import torch

from ._lib import check, cur_stream, lib, ptr, raw_stream

ACT = {"none": 0, None: 0, "relu": 1, "gelu": 2, "sigmoid": 3, "tanh": 4}

if os.environ.get("MDQE_MSDA_VARIANT"):                 # tools/ A/B of the fused MSDA's block-to-query map (csrc/msda_fused.hip)
    check(lib.mdqe_debug_msda_variant(int(os.environ["MDQE_MSDA_VARIANT"])), "msda_variant")

if os.environ.get("MDQE_MSDA_DEC_STAGED"):              # tools/ A/B: decoder box-level MSDA on the LDS-staged kernel (1, default) or v2 (0)
    check(lib.mdqe_debug_msda_dec_staged(int(os.environ["MDQE_MSDA_DEC_STAGED"])), "msda_dec_staged")

if os.environ.get("MDQE_GEMM_TILE_RULE"):               # tools/ A/B of the auto tile rule (csrc/gemm.hip dispatch_gemm)
    check(lib.mdqe_debug_gemm_tile_rule(int(os.environ["MDQE_GEMM_TILE_RULE"])), "gemm_tile_rule")
if os.environ.get("MDQE_GEMM_STAGGER"):                 # tools/ A/B: start stagger between the blocks of a CU (csrc/gemm_k16.hip), 10-ns ticks
    check(lib.mdqe_debug_gemm_stagger(int(os.environ["MDQE_GEMM_STAGGER"])), "gemm_stagger")

if os.environ.get("MDQE_MSDA_DEC_STAGE_KB"):           # tools/ A/B: LDS staging budget (KB) of the decoder's box-level deformable launch
    check(lib.mdqe_debug_msda_dec_stage_kb(int(os.environ["MDQE_MSDA_DEC_STAGE_KB"])), "msda_dec_stage_kb")
if os.environ.get("MDQE_MSDA_TP_STAGED"):              # tools/ A/B: 0 = the decoder's temporal launch on the gather form (no LDS staging)
    check(lib.mdqe_debug_msda_tp_staged(int(os.environ["MDQE_MSDA_TP_STAGED"])), "msda_tp_staged")
if os.environ.get("MDQE_GEMM_LDS_PAD"):                # tools/ A/B: extra LDS bytes per K-step-16 GEMM block (caps the GEMM blocks per CU)
    check(lib.mdqe_debug_gemm_lds_pad(int(os.environ["MDQE_GEMM_LDS_PAD"])), "gemm_lds_pad")

_ws = {}


GEMM_MODES = {"f32": 0, "f16x3": 1, "f16": 2}


def set_gemm_precision(mode):
    """'f32' (exact fp32 MFMA), 'f16x3' (split-precision f16 MFMA, ~1e-6 rel. to fp32) or 'f16' (ONE f16 MFMA pass, operands rounded to
    nearest f16, fp32 accumulate / out: the reference's autocast arithmetic) for the large-tile GEMM/conv."""
    check(lib.mdqe_set_gemm_precision(GEMM_MODES[mode]), "set_gemm_precision")


def get_gemm_precision():
    """The mode this thread's launches use (its `gemm_precision` region if inside one, else the process-wide mode)."""
    return {v: k for k, v in GEMM_MODES.items()}[lib.mdqe_get_gemm_precision()]


_tl_prec = threading.local()


@contextlib.contextmanager
def gemm_precision(mode):
    """`with gemm_precision("f16x3"):` -- the GEMM mode of the CALLING thread's launches inside the block (C ABI
    mdqe_set_gemm_precision_thread); nests; other host threads and the process-wide mode are untouched."""
    prev = getattr(_tl_prec, "v", -1)
    cur = GEMM_MODES[mode]
    check(lib.mdqe_set_gemm_precision_thread(cur), "set_gemm_precision_thread")
    _tl_prec.v = cur
    try:
        yield
    finally:
        check(lib.mdqe_set_gemm_precision_thread(prev), "set_gemm_precision_thread")
        _tl_prec.v = prev


# ---- constant weights: pre-split f16 planes for the f16x3 mode (gemm_f16x3w.hip) ---------------------
_split = {}          # data_ptr -> (weakref to the weight tensor, its _version, planes tensor)


def const_weight(w):
    """Declare `w` ([N, ...] CUDA fp32, contiguous) a constant GEMM/conv weight: its f16 hi / lo planes are computed
    once and handed to the kernels as `w_split`.  Returns w.  Tensors the fast kernel cannot take are left alone."""
    import weakref
    if not (torch.is_tensor(w) and w.is_cuda and w.dtype == torch.float32 and w.dim() >= 2 and w.is_contiguous()):
        return w
    N = w.shape[0]
    K = w.numel() // max(N, 1)
    if N < 128 or K % 32 != 0 or float(w.abs().max()) >= 32752.0:
        return w
    planes = torch.empty(3 * w.numel(), dtype=torch.float16, device=w.device)     # hi | lo (f16x3) | round-to-nearest (f16)
    check(lib.mdqe_f16x3_split_f32(ptr(w), w.numel(), ptr(planes), cur_stream()), "f16x3_split")
    key = w.data_ptr()
    _split[key] = (weakref.ref(w), w._version, planes)
    weakref.finalize(w, _drop_split, key)               # the planes go when the weight tensor goes
    return w


def _drop_split(key):
    ent = _split.get(key)
    if ent is not None and ent[0]() is None:            # (a newer tensor may have been registered at the same address)
        del _split[key]


def _wsplit(w):
    ent = _split.get(w.data_ptr())
    if ent is None:
        return None
    ref, ver, planes = ent
    base = ref()
    if base is None or base._version != ver:         # freed (the address may be reused) or modified in place
        del _split[w.data_ptr()]
        return None
    if w.numel() != base.numel() or w._version != ver or not w.is_contiguous():
        return None                                  # some other tensor at the same address (a partial view): not ours
    return planes                                    # w is the registered tensor or a full reshaped view of it


def _workspace(nbytes, device):
    """Scratch buffer (split-K partials, GroupNorm statistics), one per (device, stream): the per-frame stages and the
    per-clip stages run on different streams and must not share it."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, raw_stream(dev))
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected contiguous CUDA {dtype} tensor, got {t.dtype} {t.device} "
                           f"contiguous={t.is_contiguous()}")


def linear(x, weight, bias=None, act=None, residual=None, res_mod=0, rowmask=None, mask_cols=0, act_cols=0,
           out=None, ldc=None, tile=0, res_first=False, ksplit=0):
    """out[..., n] = epilogue(x[..., :] @ weight[n, :]).  x may be a 2-D row-strided view (last dim contiguous)."""
    K = x.shape[-1]
    N = weight.shape[0]
    if x.dim() != 2:
        x2 = x.reshape(-1, K)
    else:
        x2 = x
    if x2.stride(-1) != 1:
        raise RuntimeError("linear: last dim of x must be contiguous")
    M = x2.shape[0]
    lda = x2.stride(0) if M > 1 else K
    _chk(weight, "weight"); _chk(bias, "bias")
    if rowmask is not None and rowmask.dtype == torch.bool:
        rowmask = rowmask.view(torch.uint8)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
        ldc_ = N
    else:
        ldc_ = ldc if ldc is not None else out.stride(0) if out.dim() == 2 else N
    ldr = 0
    if residual is not None:
        ldr = residual.stride(0) if residual.dim() == 2 else residual.shape[-1]
    ws = None
    # (no automatic split-K here: a split changes the order of a row's K sum, and which arithmetic a row gets must not depend on
    # how many rows share its launch -- the schedule tests demand identical bits across pass sizes; callers pass ksplit explicitly)
    if ksplit > 1:
        ws = _workspace(ksplit * M * N * 4 + 64, x.device)
    check(lib.mdqe_gemm_nt_f32(ptr(x2), lda, ptr(weight), ptr(bias), ptr(out), ldc_, M, N, K, ACT[act], act_cols,
                               ptr(residual), ldr, res_mod, int(res_first), ptr(rowmask), mask_cols, tile, ksplit, ptr(ws),
                               ptr(_wsplit(weight)), cur_stream()), "gemm_nt_f32")
    if x.dim() != 2 and ldc is None and out.dim() == 2 and out.shape == (M, N):
        return out.view(*x.shape[:-1], N)
    return out


def conv2d_nhwc(x, w_packed, bias=None, stride=1, pad=0, act=None, residual=None, out=None, tile=0, res_first=False, ksplit=0):
    """x [NI,H,W,Cin] (dense, or a view whose images are x.stride(0) apart with dense pixels);
    w_packed [Cout,KH,KW,Cin]; returns [NI,OH,OW,Cout]."""
    _chk(w_packed, "w"); _chk(bias, "bias"); _chk(residual, "residual")
    NI, H, W, Cin = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and x.stride(3) == 1 and x.stride(2) == Cin and x.stride(1) == W * Cin):
        raise RuntimeError("conv2d_nhwc: x must be CUDA fp32 NHWC with dense pixels")
    xis = x.stride(0) if NI > 1 else 0
    Cout, KH, KW, _ = w_packed.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty((NI, OH, OW, Cout), dtype=torch.float32, device=x.device)
    ldy = out.stride(-2)
    ldr = residual.stride(-2) if residual is not None else 0
    M, K = NI * OH * OW, KH * KW * Cin
    ws = None
    if ksplit == 0 and K >= 4096 and ((OH * OW + 63) // 64) * ((Cout + 63) // 64) <= 16:
        # few output pixels PER IMAGE, very deep K (input_proj's 3x3/s2 on res5: 60 pixels, K = 18432): spread K over the CUs.  Decided
        # from one image's tiles, never from the batch: a split changes the order of the K sum, and a frame's bits must not depend on
        # how many frames share its pass
        ksplit = min(16, K // 1024)
    if ksplit > 1:
        ws = _workspace(ksplit * M * Cout * 4 + 64, x.device)
    check(lib.mdqe_conv2d_nhwc_f32(ptr(x), xis, ptr(w_packed), ptr(bias), ptr(out), ldy, NI, H, W, Cin, Cout, KH, KW, stride,
                                   pad, ACT[act], ptr(residual), ldr, int(res_first), tile, ptr(_wsplit(w_packed)), ksplit, ptr(ws),
                                   cur_stream()),
          "conv2d_nhwc_f32")
    return out


LINEAR_LN_FUSED = os.environ.get("MDQE_LINEAR_LN_FUSED", "1") != "0"     # 0: GEMM then LayerNorm kernel (debug / A-B)
LINEAR_LN_MIN_ROWS = 16384        # below this the 64x256 tile leaves CUs idle: 64x64 GEMM + LayerNorm kernel is faster


def linear_ln(x, weight, bias, residual, gamma, beta, eps=1e-5, out=None, scratch=None, second=None):
    """out = LayerNorm(x @ weight^T + bias + residual) * gamma + beta over N == 256 columns; `out` may be `residual`.
    One kernel (64x256 tile, statistics in the epilogue) in exact-fp32 mode with enough rows to fill the chip; otherwise
    the GEMM (own tile / f16x3 arithmetic) into `scratch` followed by the LayerNorm kernel.
    second=(gamma2, beta2): also returns out2 = LayerNorm(out) * gamma2 + beta2 -- in the same epilogue where the one-kernel form runs
    (mdqe_gemm_ln2_f32), by one more LayerNorm launch otherwise; the two give the same bits."""
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    # one block per 64 rows: a grid just over a multiple of the 256 CUs runs its last round nearly empty (21 168 rows = 331 blocks take two
    # rounds: 139 us against 117 for GEMM + LayerNorm, tools/ln_tile_ab.py) -- the two forms give identical bits, so the choice is free
    nb = (M + 63) // 64
    full = nb >= 1024 or nb / (256.0 * ((nb + 255) // 256)) >= 0.8
    small = M * max(out.stride(0), residual.stride(0) if residual is not None else 0) * 4 < 0xFFFF0000       # (32-bit buffer offsets in the epilogue)
    if LINEAR_LN_FUSED and N == 256 and M >= LINEAR_LN_MIN_ROWS and full and small and get_gemm_precision() == "f32" and x.stride(1) == 1:
        _chk(weight, "weight"); _chk(bias, "bias"); _chk(gamma, "gamma"); _chk(beta, "beta"); _chk(residual, "residual"); _chk(out, "out")
        if second is not None and LINEAR_LN2_FUSED:
            _chk(second[0], "gamma2"); _chk(second[1], "beta2")
            out2 = torch.empty((M, N), dtype=torch.float32, device=x.device)
            check(lib.mdqe_gemm_ln2_f32(ptr(x), x.stride(0) if M > 1 else K, ptr(weight), ptr(bias), ptr(out), out.stride(0), M, N, K,
                                        ptr(residual), residual.stride(0) if residual is not None else 0, ptr(gamma), ptr(beta),
                                        ptr(second[0]), ptr(second[1]), ptr(out2), out2.stride(0), eps, cur_stream()), "gemm_ln2_f32")
            return out, out2
        check(lib.mdqe_gemm_ln_f32(ptr(x), x.stride(0) if M > 1 else K, ptr(weight), ptr(bias), ptr(out), out.stride(0), M, N, K,
                                   ptr(residual), residual.stride(0) if residual is not None else 0, ptr(gamma), ptr(beta), eps,
                                   cur_stream()), "gemm_ln_f32")
        return out if second is None else (out, layernorm(out, second[0], second[1], eps=eps))
    y = linear(x, weight, bias, residual=residual, out=scratch)
    out = layernorm(y, gamma, beta, eps=eps, out=out)
    return out if second is None else (out, layernorm(out, second[0], second[1], eps=eps))


LINEAR_LN2_FUSED = os.environ.get("MDQE_LINEAR_LN2_FUSED", "1") != "0"   # 0: the second LayerNorm as its own launch (debug / A-B)


def linear_side(x, weight, bias, side, side_w, side_cols, out=None):
    """out = x @ weight^T + bias, and out[:, :side_cols] += side [M,4] @ side_w[:side_cols, :4]^T (mdqe_gemm_nt_side_f32): a projection
    of `x + pos` for a pos that is linear in four numbers per row, without materialising pos."""
    M, K = x.shape
    N = weight.shape[0]
    _chk(weight, "weight"); _chk(bias, "bias"); _chk(side, "side"); _chk(side_w, "side_w")
    if x.stride(1) != 1 or side.shape != (M, 4) or side_w.shape != (N, 4):
        raise RuntimeError("linear_side: x rows contiguous, side [M,4], side_w [N,4] required")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(lib.mdqe_gemm_nt_side_f32(ptr(x), x.stride(0) if M > 1 else K, ptr(weight), ptr(bias), ptr(out), out.stride(0), M, N, K,
                                    ptr(side), ptr(side_w), side_cols, ptr(_wsplit(weight)), cur_stream()), "gemm_nt_side_f32")
    return out


def linear_cat2(y, x_nhwc, stride, weight, bias, act=None, out=None):
    """act([y | x'] @ weight^T + bias): a bottleneck's conv3 and its projection shortcut in one product (mdqe_gemm_nt_cat2_f32).
    y [NI, OH, OW, K1] (contiguous), x_nhwc [NI, H2, W2, K2] read at (oh*stride, ow*stride), weight [N, K1 + K2] -> [NI, OH, OW, N]."""
    _chk(y, "y"); _chk(x_nhwc, "x"); _chk(weight, "weight"); _chk(bias, "bias")
    NI, OH, OW, K1 = y.shape
    _, H2, W2, K2 = x_nhwc.shape
    N = weight.shape[0]
    if not (y.is_contiguous() and x_nhwc.is_contiguous() and weight.is_contiguous() and weight.shape[1] == K1 + K2):
        raise RuntimeError("linear_cat2: contiguous operands and a [N, K1 + K2] weight required")
    if out is None:
        out = torch.empty((NI, OH, OW, N), dtype=torch.float32, device=y.device)
    check(lib.mdqe_gemm_nt_cat2_f32(ptr(y), K1, K1, ptr(x_nhwc), K2, K2, NI, OH, OW, H2, W2, stride, ptr(weight), ptr(bias), ptr(out), N, N,
                                    ACT[act], cur_stream()), "gemm_nt_cat2")
    return out


def layernorm(x, gamma, beta, res=None, eps=1e-5, out=None):
    _chk(x, "x"); _chk(res, "res"); _chk(gamma, "gamma"); _chk(beta, "beta")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    check(lib.mdqe_layernorm_f32(ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(out), rows, C, eps, cur_stream()), "layernorm")
    return out


def groupnorm_nhwc(x, groups, gamma, beta, act=None, eps=1e-5, out=None):
    """x [NI, HW, C] (pixel stride x.stride(1), image stride x.stride(0)); `out` may be a strided view
    (e.g. a level slice of the encoder token buffer) or x itself (in place)."""
    if x.dim() == 4:
        x = x.view(x.shape[0], -1, x.shape[-1])
    NI, HW, C = x.shape
    if out is None:
        out = torch.empty((NI, HW, C), dtype=torch.float32, device=x.device)
    elif out.dim() == 4:
        out = out.view(NI, HW, C)
    ws = _workspace(lib.mdqe_groupnorm_workspace_bytes(NI, groups), x.device)
    check(lib.mdqe_groupnorm_nhwc_f32(ptr(x), x.stride(1), x.stride(0) if NI > 1 else 0, ptr(out), out.stride(1),
                                      out.stride(0) if NI > 1 else 0, NI, HW, C, groups, ptr(gamma),
                                      ptr(beta), eps, ACT[act], ptr(ws), cur_stream()), "groupnorm_nhwc")
    return out


def stem_im2col(frames, Hp, Wp, mean, std):
    """frames [NI,3,h,w] uint8 or fp32 CUDA -> [NI*Hp/2*Wp/2, 160] normalised+padded im2col."""
    import ctypes
    if not frames.is_cuda or not frames.is_contiguous() or frames.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError("stem_im2col: frames must be contiguous CUDA uint8/float32 [NI,3,h,w]")
    NI, _, h, w = frames.shape
    out = torch.empty((NI * (Hp // 2) * (Wp // 2), 160), dtype=torch.float32, device=frames.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(lib.mdqe_stem_im2col_f32(ptr(frames), int(frames.dtype == torch.uint8), 3 * h * w, NI, h, w, Hp, Wp, m, s, ptr(out),
                                   cur_stream()), "stem_im2col")
    return out


def stem_weight_kmajor(w_krsc):
    """[64,7,7,3] (BN-folded, k = (kh,kw,c)) -> the [154,64] k-major layout of mdqe_stem_conv_f32 (row kh*22+21 zero)."""
    wk = torch.zeros(7, 22, 64, dtype=torch.float32)
    wk[:, :21] = w_krsc.reshape(64, 7, 21).permute(1, 2, 0)
    return wk.reshape(154, 64).contiguous()


def stem_conv(frames, Hp, Wp, mean, std, wk, bias):
    """frames [NI,3,h,w] uint8 or fp32 CUDA -> relu(conv7x7/s2(pad(normalise(frames))) + bias) as NHWC [NI,Hp/2,Wp/2,64]."""
    import ctypes
    if not frames.is_cuda or not frames.is_contiguous() or frames.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError("stem_conv: frames must be contiguous CUDA uint8/float32 [NI,3,h,w]")
    _chk(wk, "wk"); _chk(bias, "bias")
    if tuple(wk.shape) != (154, 64) or bias.numel() != 64:
        raise RuntimeError("stem_conv: wk must be [154,64] (stem_weight_kmajor) and bias [64]")
    NI, _, h, w = frames.shape
    out = torch.empty((NI, Hp // 2, Wp // 2, 64), dtype=torch.float32, device=frames.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(lib.mdqe_stem_conv_f32(ptr(frames), int(frames.dtype == torch.uint8), 3 * h * w, NI, h, w, Hp, Wp, m, s, ptr(wk),
                                 ptr(bias), ptr(out), cur_stream()), "stem_conv")
    return out


def maxpool3x3s2(x):
    _chk(x, "x")
    NI, H, W, C = x.shape
    out = torch.empty((NI, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    check(lib.mdqe_maxpool3x3s2_nhwc_f32(ptr(x), ptr(out), NI, H, W, C, cur_stream()), "maxpool")
    return out


def upsample_nearest_add(a, b, out=None):
    _chk(a, "a"); _chk(b, "b")
    NI, H, W, C = a.shape
    if out is None:
        out = torch.empty_like(a)
    check(lib.mdqe_upsample_nearest_add_nhwc_f32(ptr(a), ptr(b), ptr(out), NI, H, W, b.shape[1], b.shape[2], C, cur_stream()),
          "upsample_nearest_add")
    return out


def dwconv5x5(x, wt, bias, up2=False, tw=None, tb=None):
    """x [NI,H,W,C]; wt [25,C]; if up2: output is [NI,2H,2W,C] over the virtual transposed-conv output."""
    _chk(x, "x"); _chk(wt, "wt"); _chk(bias, "bias")
    NI, H, W, C = x.shape
    OH, OW = (2 * H, 2 * W) if up2 else (H, W)
    out = torch.empty((NI, OH, OW, C), dtype=torch.float32, device=x.device)
    if C == 256 and DW_FAST:                              # wave = the 64 channel groups of a pixel, taps in registers
        if up2:
            _chk(tw, "tw"); _chk(tb, "tb")
            check(lib.mdqe_dwconv5x5_up2_c256_f32(ptr(x), ptr(wt), ptr(bias), ptr(tw), ptr(tb), ptr(out), NI, H, W, cur_stream()),
                  "dwconv5x5_up2_c256")
        else:
            check(lib.mdqe_dwconv5x5_c256_f32(ptr(x), ptr(wt), ptr(bias), ptr(out), NI, H, W, cur_stream()), "dwconv5x5_c256")
        return out
    check(lib.mdqe_dwconv5x5_nhwc_f32(ptr(x), ptr(wt), ptr(bias), ptr(out), NI, OH, OW, C, int(up2), ptr(tw), ptr(tb),
                                      cur_stream()), "dwconv5x5")
    return out


DW_FAST = os.environ.get("MDQE_DW_FAST", "1") != "0"        # 0: generic depthwise kernel (debug / A-B)


def msda(value, shapes_dev, starts_dev, loc, attn, groups=1, scale=1.0, out=None):
    """value [B,S,M,D]; shapes_dev [G*L,2] int64; loc [B,Q,M,L,P,2]; attn [B,Q,M,L,P] -> [B,Q,M*D]."""
    B, S, M, D = value.shape
    Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    if out is None:
        out = torch.empty((B, Q, M * D), dtype=torch.float32, device=value.device)
    check(lib.mdqe_msda_forward_grouped_f32(ptr(value), ptr(shapes_dev), ptr(starts_dev), ptr(loc), ptr(attn), B, S, M, D,
                                            groups, L, Q, P, scale, ptr(out), cur_stream()), "msda")
    return out


def msda_fused(value, offs, logits, ref, levels, B, Q, M, D, L, P, mode=0, grid=None, groups=1, scale=1.0,
               v_brows=None, out=None, vidx=None):
    """Fused MSDeformAttn core.  value/offs/logits: 2-D row-strided fp32 views (last dim contiguous):
    value [B*v_brows, M*D], offs [B*Q, M*L*P*2], logits [B*Q, M*L*P].  ref: [Q,2|4] (broadcast) or [B,Q,2|4].
    levels: (H list, W list, start-row list) of length groups*L."""
    import ctypes
    Hs, Ws, Ss = levels
    n = groups * L
    assert len(Hs) == n and len(Ws) == n and len(Ss) == n
    if ref.dim() == 2:
        ref_b, ref_dim = 0, ref.shape[-1]
    else:
        ref_b, ref_dim = ref.stride(0), ref.shape[-1]
    _chk(ref, "ref")
    if out is None:
        out = torch.empty((B * Q, M * D), dtype=torch.float32, device=value.device)
    arr = lambda v: (ctypes.c_int * n)(*[int(x) for x in v])
    if v_brows is None:
        v_brows = value.shape[0] // max(B, 1)
    if vidx is not None and (vidx.dtype != torch.int32 or not vidx.is_cuda or vidx.numel() != B):
        raise RuntimeError("msda_fused: vidx must be a CUDA int32 tensor with B entries")
    check(lib.mdqe_msda_fused_f32(ptr(value), value.stride(0), v_brows, ptr(vidx), ptr(offs), offs.stride(0), ptr(logits),
                                  logits.stride(0), ptr(ref), ref_b, ref_dim, mode, ptr(grid), arr(Hs), arr(Ws), arr(Ss),
                                  B, M, D, groups, L, Q, P, scale, ptr(out), out.stride(0), value.shape[0], cur_stream()), "msda_fused")
    return out


def mask_row_stats(logits):
    """logits [n,T,H,W] -> (stats [n,5], soft_h [n,Ph], hard_h [n,Ph]); see mdqe_mask_row_stats_f32."""
    _chk(logits, "logits")
    n, T, H, W = logits.shape
    t_step = 2 if T >= 5 else 1
    Th = (T + t_step - 1) // t_step
    Ph = Th * (H // 2) * (W // 2)
    stats = torch.empty(n, 5, device=logits.device)
    soft_h = torch.empty(n, Ph, device=logits.device)
    hard_h = torch.empty(n, Ph, device=logits.device)
    check(lib.mdqe_mask_row_stats_f32(ptr(logits), n, T, H, W, t_step, ptr(stats), ptr(soft_h), ptr(hard_h), cur_stream()),
          "mask_row_stats")
    return stats, soft_h, hard_h


def mha_small(qk, v, B, Q, C, nh, out=None):
    """qk [B*Q, 2C] (q | k), v [B*Q, C] -> [B*Q, C]."""
    if out is None:
        out = torch.empty((B * Q, C), dtype=torch.float32, device=v.device)
    check(lib.mdqe_mha_small_f32(ptr(qk), qk.stride(0), ptr(v), v.stride(0), ptr(out), out.stride(0), B, Q, C, nh, cur_stream()),
          "mha_small")
    return out


def query_select(conf, nb, out=None):
    """conf [NI,H,W,K] -> coords [NI, nb*nb, 2] (out: a contiguous tensor of that shape to store into)."""
    _chk(conf, "conf")
    NI, H, W, K = conf.shape
    ws = torch.empty(NI * H * W, dtype=torch.float32, device=conf.device)
    if out is not None:
        _chk(out, "out")
        if tuple(out.shape) != (NI, nb * nb, 2):
            raise RuntimeError("query_select: out must be [NI, nb*nb, 2]")
    coords = out if out is not None else torch.empty(NI, nb * nb, 2, dtype=torch.float32, device=conf.device)
    check(lib.mdqe_query_select_f32(ptr(conf), NI, H, W, K, nb, ptr(ws), ptr(coords), cur_stream()), "query_select")
    return coords


def sample_levels_mean(tokens, coords, shapes, starts, out=None):
    """tokens [NI,N,C], coords [NI,Q,2] -> [NI,Q,C] (out: a contiguous tensor of that shape to store into)."""
    import ctypes
    _chk(tokens, "tokens"); _chk(coords, "coords")
    NI, N, C = tokens.shape
    Qn = coords.shape[1]
    n = len(shapes)
    arr = lambda v: (ctypes.c_int * n)(*[int(x) for x in v])
    if out is None:
        out = torch.empty(NI, Qn, C, dtype=torch.float32, device=tokens.device)
    else:
        _chk(out, "out")
        if tuple(out.shape) != (NI, Qn, C):
            raise RuntimeError("sample_levels_mean: out must be [NI, Q, C]")
    check(lib.mdqe_sample_levels_mean_f32(ptr(tokens), NI, N, C, ptr(coords), Qn, arr([s[0] for s in shapes]), arr([s[1] for s in shapes]),
                                          arr(starts), n, ptr(out), cur_stream()), "sample_levels_mean")
    return out


def final_masks(logits, inst_idx, factor, h, w, Ho, Wo, out, f_off):
    """logits [n,Fw,Hm,Wm]; inst_idx int32 CUDA [n_sel]; out uint8 [n_sel_total, L, Ho, Wo] (rows 0..n_sel-1 written)."""
    _chk(logits, "logits")
    n, Fw, Hm, Wm = logits.shape
    check(lib.mdqe_final_masks_u8(ptr(logits), int(inst_idx.numel()), ptr(inst_idx), Fw, Hm, Wm, factor, h, w, Ho, Wo, ptr(out),
                                  out.stride(0), f_off, cur_stream()), "final_masks")
    return out


def layernorm_post(x, gamma, beta, post, eps=1e-5, out=None):
    """out = LN(x)*gamma + beta + post."""
    _chk(x, "x"); _chk(post, "post")
    C = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    check(lib.mdqe_layernorm_post_f32(ptr(x), ptr(gamma), ptr(beta), ptr(post), ptr(out), x.numel() // C, C, eps, cur_stream()),
          "layernorm_post")
    return out


def patch4_im2col(frames, Hp, Wp, mean, std):
    import ctypes
    NI, _, h, w = frames.shape
    out = torch.empty((NI * (Hp // 4) * (Wp // 4), 48), dtype=torch.float32, device=frames.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(lib.mdqe_patch4_im2col_f32(ptr(frames), int(frames.dtype == torch.uint8), 3 * h * w, NI, h, w, Hp, Wp, m, s, ptr(out),
                                     cur_stream()), "patch4_im2col")
    return out


def swin_window_gather(x, ws, shift):
    """x [B,H,W,C] -> window rows [B*nW*ws*ws, C] (zero pad + cyclic shift + partition)."""
    _chk(x, "x")
    B, H, W, C = x.shape
    Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
    out = torch.empty((B * Hp * Wp, C), dtype=torch.float32, device=x.device)
    check(lib.mdqe_swin_window_f32(ptr(x), None, ptr(out), B, H, W, C, ws, shift, 0, cur_stream()), "swin_window_gather")
    return out


def swin_window_scatter_add(rows, shortcut, ws, shift, out=None):
    """out[b,y,x] = shortcut[b,y,x] + rows[window order] (reverse + un-shift + crop)."""
    _chk(rows, "rows"); _chk(shortcut, "shortcut")
    B, H, W, C = shortcut.shape
    if out is None:
        out = torch.empty_like(shortcut)
    check(lib.mdqe_swin_window_f32(ptr(rows), ptr(shortcut), ptr(out), B, H, W, C, ws, shift, 1, cur_stream()), "swin_window_scatter")
    return out


def linear_swin(x, weight, bias, ws, shift, out=None):
    """x [B,H,W,C] (contiguous NHWC) -> window_partition(roll(pad(x))) @ weight^T + bias as rows [B*Hp*Wp, N] in window order, without
    the partitioned copy (mdqe_gemm_nt_swin_f32; exact-fp32 GEMM mode only -- callers check get_gemm_precision())."""
    _chk(x, "x"); _chk(weight, "weight"); _chk(bias, "bias")
    B, H, W, C = x.shape
    N = weight.shape[0]
    Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
    if out is None:
        out = torch.empty((B * Hp * Wp, N), dtype=torch.float32, device=x.device)
    check(lib.mdqe_gemm_nt_swin_f32(ptr(x), C, ptr(weight), ptr(bias), ptr(out), out.stride(0), B, H, W, ws, shift, N, C, cur_stream()),
          "gemm_nt_swin")
    return out


def layernorm_swin_scatter(rows, gamma, beta, shortcut, ws, shift, eps=1e-5, out=None):
    """out[b,y,x] = shortcut[b,y,x] + LN(rows[window-order row of (b,y,x)]) * gamma + beta; rows [B*Hp*Wp, C], shortcut [B,H,W,C].
    out=None writes over `shortcut` (every pixel is written by exactly one row)."""
    _chk(rows, "rows"); _chk(shortcut, "shortcut"); _chk(gamma, "gamma"); _chk(beta, "beta")
    B, H, W, C = shortcut.shape
    if out is None:
        out = shortcut
    check(lib.mdqe_layernorm_swin_scatter_f32(ptr(rows), ptr(gamma), ptr(beta), ptr(shortcut), ptr(out), B, H, W, C, ws, shift, eps,
                                              cur_stream()), "layernorm_swin_scatter")
    return out


def window_attn(qkv, n_windows, N, C, nh, scale, bias, mask=None, nW=1):
    _chk(qkv, "qkv"); _chk(scale, "scale"); _chk(bias, "bias"); _chk(mask, "mask")
    out = torch.empty((n_windows * N, C), dtype=torch.float32, device=qkv.device)
    check(lib.mdqe_window_attn_f32(ptr(qkv), qkv.stride(0), ptr(out), C, n_windows, N, C, nh, ptr(scale), ptr(bias), ptr(mask), nW,
                                   cur_stream()), "window_attn")
    return out


def patch_merge_gather(x):
    _chk(x, "x")
    B, H, W, C = x.shape
    out = torch.empty((B * ((H + 1) // 2) * ((W + 1) // 2), 4 * C), dtype=torch.float32, device=x.device)
    check(lib.mdqe_patch_merge_gather_f32(ptr(x), ptr(out), B, H, W, C, cur_stream()), "patch_merge_gather")
    return out


def image_mask_stats(logits, factor, h, w):
    """logits [n,Hm,Wm] -> [n,6] = (sum soft*hard, count hard, xmin, ymin, xmax, ymax of logit>0) over the x`factor` crop [:h,:w]."""
    _chk(logits, "logits")
    n, Hm, Wm = logits.shape
    out = torch.empty(n, 6, device=logits.device)
    check(lib.mdqe_image_mask_stats_f32(ptr(logits), n, Hm, Wm, factor, h, w, ptr(out), cur_stream()), "image_mask_stats")
    return out


def image_final_masks(logits, idx, factor, h, w, Ho, Wo):
    """logits [n,Hm,Wm], idx int32 CUDA [k] -> uint8 [k,Ho,Wo]: bilinear resize of the cropped up-sampled logits, > 0."""
    _chk(logits, "logits")
    n, Hm, Wm = logits.shape
    out = torch.empty(int(idx.numel()), Ho, Wo, dtype=torch.uint8, device=logits.device)
    check(lib.mdqe_image_final_masks_u8(ptr(logits), int(idx.numel()), ptr(idx), Hm, Wm, factor, h, w, Ho, Wo, ptr(out), cur_stream()),
          "image_final_masks")
    return out


def final_masks_rle(logits, inst_idx, factor, h, w, Ho, Wo, cap):
    """logits [n,Fw,Hm,Wm]; inst_idx int32 CUDA [n_sel] -> (pos int32 [n_sel*Fw, cap], n_pos int32 [n_sel*Fw]): column-major
    positions at which each final mask changes value (ops.final_masks never materialised)."""
    _chk(logits, "logits")
    n, Fw, Hm, Wm = logits.shape
    k = int(inst_idx.numel())
    pos = torch.empty(k * Fw, cap, dtype=torch.int32, device=logits.device)
    n_pos = torch.empty(k * Fw, dtype=torch.int32, device=logits.device)
    check(lib.mdqe_final_masks_rle(ptr(logits), k, ptr(inst_idx), Fw, Hm, Wm, factor, h, w, Ho, Wo, cap, ptr(pos), ptr(n_pos),
                                   cur_stream()), "final_masks_rle")
    return pos, n_pos


# ---- per-clip stages (csrc/clip_ops.hip) -----------------------------------------------------------------------------
def clip_assoc(emb, fidx, ct, wdw, nb):
    """emb [frames, Q, E]; fidx [Bc, T] int32 CUDA -> idx [Bc, T, Q] int32 (inter-frame query association)."""
    _chk(emb, "emb"); _chk(fidx, "fidx", torch.int32)
    Bc, T = fidx.shape
    Q, E = emb.shape[1], emb.shape[2]
    idx = torch.empty(Bc, T, Q, dtype=torch.int32, device=emb.device)
    check(lib.mdqe_clip_assoc_f32(ptr(emb), Q, E, ptr(fidx), Bc, T, ct, float(wdw), nb, ptr(idx), cur_stream()), "clip_assoc")
    return idx


def clip_gather_init(content, coords, fidx, idx, ct):
    """-> x [Bc*T*Q, C], ref [Bc*T*Q, 4], x_inst [Bc*Q, C]."""
    _chk(content, "content"); _chk(coords, "coords"); _chk(fidx, "fidx", torch.int32); _chk(idx, "idx", torch.int32)
    Bc, T = fidx.shape
    Q, C = content.shape[1], content.shape[2]
    x = torch.empty(Bc * T * Q, C, device=content.device)
    ref = torch.empty(Bc * T * Q, 4, device=content.device)
    xi = torch.empty(Bc * Q, C, device=content.device)
    check(lib.mdqe_clip_gather_init_f32(ptr(content), ptr(coords), ptr(fidx), ptr(idx), Bc, T, Q, C, ct, ptr(x), ptr(ref), ptr(xi),
                                        cur_stream()), "clip_gather_init")
    return x, ref, xi


def box_refine(delta, prev, Bc, T, Q, t0, t1):
    """boxes = sigmoid(delta + inverse_sigmoid(prev)) [Bc*T*Q, 4]; clip boxes [Bc*Q, 4] over frames [t0, t1)."""
    _chk(delta, "delta"); _chk(prev, "prev")
    boxes = torch.empty(Bc * T * Q, 4, device=delta.device)
    ibox = torch.empty(Bc * Q, 4, device=delta.device)
    check(lib.mdqe_box_refine_f32(ptr(delta), ptr(prev), Bc, T, Q, t0, t1, ptr(boxes), ptr(ibox), cur_stream()), "box_refine")
    return boxes, ibox


def box_head_refine(h, w, b, prev, Bc, T, Q, t0, t1):
    """box_refine(h @ w^T + b, prev) with the 4-column product inside the kernel: h [Bc*T*Q, K] (K % 256 == 0), w [4, K], b [4]."""
    _chk(h, "h"); _chk(w, "w"); _chk(b, "b"); _chk(prev, "prev")
    boxes = torch.empty(Bc * T * Q, 4, device=h.device)
    ibox = torch.empty(Bc * Q, 4, device=h.device)
    check(lib.mdqe_box_head_refine_f32(ptr(h), h.stride(0), ptr(w), ptr(b), ptr(prev), Bc, T, Q, h.shape[1], t0, t1, ptr(boxes), ptr(ibox),
                                       cur_stream()), "box_head_refine")
    return boxes, ibox


def time_fuse_dot(xw, wt, bt, src, Bc, T, Q):
    """time_fuse(xw @ wt^T + bt, src): the time weights are computed inside the kernel.  xw, src [Bc*T*Q, 256], wt [1, 256], bt [1]."""
    _chk(xw, "xw"); _chk(wt, "wt"); _chk(bt, "bt"); _chk(src, "src")
    C = src.shape[-1]
    out = torch.empty(Bc * Q, C, device=src.device)
    check(lib.mdqe_time_fuse_dot_f32(ptr(xw), ptr(wt), ptr(bt), ptr(src), Bc, T, Q, C, ptr(out), cur_stream()), "time_fuse_dot")
    return out


def add_rows(a, b, out=None):
    """out = a + b for 2-D row-strided fp32 operands of the same shape."""
    rows, C = a.shape
    if out is None:
        out = torch.empty(rows, C, device=a.device)
    check(lib.mdqe_add_rows_f32(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), rows, C, cur_stream()), "add_rows")
    return out


def time_fuse(w, x, Bc, T, Q, pos=None):
    """out[b,q,:] = sum_t softmax_t(w[b,t,q]) x[b,t,q,:]  (-> (out, out + pos) when pos is given)."""
    _chk(w, "w"); _chk(x, "x"); _chk(pos, "pos")
    C = x.shape[-1]
    out = torch.empty(Bc * Q, C, device=x.device)
    out2 = torch.empty(Bc * Q, C, device=x.device) if pos is not None else None
    check(lib.mdqe_time_fuse_f32(ptr(w), ptr(x), Bc, T, Q, C, ptr(out), ptr(pos), ptr(out2), cur_stream()), "time_fuse")
    return (out, out2) if pos is not None else out


def clip_select(cls, emb, thr, max_keep):
    """cls [B,Q,K], emb [B,Q,C] -> kept [B,Q] int32 (query indices of the kept ranks, score order), n_keep [B] int32."""
    _chk(cls, "cls"); _chk(emb, "emb")
    B, Q, K = cls.shape
    C = emb.shape[-1]
    dev = cls.device
    order = torch.empty(B, Q, dtype=torch.int32, device=dev)
    n_thr = torch.empty(B, dtype=torch.int32, device=dev)
    inv = torch.empty(B, Q, device=dev)
    sim = torch.empty(B, Q, Q, device=dev)
    kept = torch.empty(B, Q, dtype=torch.int32, device=dev)
    n_keep = torch.empty(B, dtype=torch.int32, device=dev)
    check(lib.mdqe_clip_select_f32(ptr(cls), ptr(emb), B, Q, K, C, float(thr), int(max_keep), ptr(order), ptr(n_thr), ptr(inv), ptr(sim),
                                   ptr(kept), ptr(n_keep), cur_stream()), "clip_select")
    return kept, n_keep


def dyn_mask_nms(coef, kept, feats, row0, n, f0, T):
    """The fused dynamic-mask kernel + NMS for a batch of clips.  coef [B,Q,M]; kept [B,Q] int32; feats [frames,H,W,M];
    row0 / n / f0: host int32 numpy arrays [B].  -> logits [n_rows,T,H,W], stats [n_rows,5], mi [n_rows]."""
    import numpy as np
    _chk(coef, "coef"); _chk(kept, "kept", torch.int32); _chk(feats, "feats")
    B, Q, M = coef.shape
    H, W = feats.shape[1], feats.shape[2]
    n_rows = int(n.sum())
    dev = coef.device
    logits = torch.empty(n_rows, T, H, W, device=dev)
    stats = torch.empty(n_rows, 5, device=dev)
    mi = torch.empty(n_rows, device=dev)
    if n_rows == 0:
        return logits, stats, mi
    t_step = 2 if T >= 5 else 1
    Ph = ((T + t_step - 1) // t_step) * (H // 2) * (W // 2)
    soft_h = torch.empty(n_rows, Ph, device=dev)
    hard_h = torch.empty(n_rows, Ph, device=dev)
    gram = torch.empty(lib.mdqe_nms_workspace_floats(int(n.max())), device=dev)
    part = torch.empty(lib.mdqe_dyn_mask_workspace_floats(n_rows, T, H, W), device=dev)
    row0 = np.ascontiguousarray(row0, dtype=np.int32); n = np.ascontiguousarray(n, dtype=np.int32); f0 = np.ascontiguousarray(f0, dtype=np.int32)
    check(lib.mdqe_dyn_mask_nms_f32(ptr(coef), ptr(kept), ptr(feats), B, Q, M, T, H, W, row0.ctypes.data, n.ctypes.data, f0.ctypes.data,
                                    ptr(logits), ptr(soft_h), ptr(hard_h), ptr(part), ptr(gram), ptr(stats), ptr(mi), cur_stream()), "dyn_mask_nms")
    return logits, stats, mi


def clip_finalize(cls, emb, kept, stats, mi, thr, row0, n):
    """-> sel [n_rows] int32, n_sel [B] int32, out [n_rows, 2+K+C] (rows row0[b] .. row0[b]+n_sel[b] of clip b are valid)."""
    import numpy as np
    B, Q, K = cls.shape
    C = emb.shape[-1]
    n_rows = int(n.sum())
    dev = cls.device
    sel = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
    n_sel = torch.empty(B, dtype=torch.int32, device=dev)
    out = torch.empty(max(n_rows, 1), 2 + K + C, device=dev)
    row0 = np.ascontiguousarray(row0, dtype=np.int32); n = np.ascontiguousarray(n, dtype=np.int32)
    check(lib.mdqe_clip_finalize_f32(ptr(cls), ptr(emb), ptr(kept), ptr(stats), ptr(mi), B, Q, K, C, float(thr), row0.ctypes.data,
                                     n.ctypes.data, ptr(sel), ptr(n_sel), ptr(out), cur_stream()), "clip_finalize")
    return sel, n_sel, out


def rows_gather(src, idx, out=None):
    """out[i] = src[idx[i]] over the leading dim (idx int32 CUDA)."""
    _chk(src, "src"); _chk(idx, "idx", torch.int32)
    n = int(idx.numel())
    row_len = src.numel() // max(src.shape[0], 1)
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), device=src.device)
    check(lib.mdqe_rows_gather_f32(ptr(src), ptr(idx), n, row_len, ptr(out), cur_stream()), "rows_gather")
    return out
