"""Drop-in for the reference's native extension module `MultiScaleDeformableAttention`
(imported at mdqe/models/ops/functions/ms_deform_attn_func.py:19; exports at
mdqe/models/ops/src/vision.cpp:13-16).

    import mdqe_cvpr2023_amd.MultiScaleDeformableAttention as MSDA
    out = MSDA.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)

Same argument meaning and error behaviour as ms_deform_attn_cuda_forward
(src/cuda/ms_deform_attn_cuda.cu:20-80): tensors must be contiguous and on the GPU, float32 or float64 (the reference's
AT_DISPATCH_FLOATING_TYPES, .cu:64,134; the eval path's callers force fp32, func.py:24; the reference's own test script runs
double, ops/test.py:32-44,63-86), `batch % min(batch, im2col_step) == 0`; returns a NEW [B,Q,M*D] tensor;
asynchronous on the current stream.  Violations raise RuntimeError (AT_ASSERTM -> RuntimeError there).
The arithmetic runs in libmdqe_hip.so (csrc/msda.hip); torch only owns the memory and the stream.
"""
import torch

from ._lib import check, cur_stream, lib, ptr


def _req(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        _req(t.is_contiguous(), f"{name} tensor has to be contiguous")
        _req(t.is_cuda, f"{name} must be a CUDA tensor")
    _req(value.dtype in (torch.float32, torch.float64), f'"ms_deform_attn_forward_cuda" not implemented for \'{value.dtype}\'')
    _req(sampling_loc.dtype == value.dtype and attn_weight.dtype == value.dtype,
         "ms_deform_attn_forward: value, sampling_loc and attn_weight must have the same dtype")
    _req(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
         "spatial_shapes / level_start_index must be int64")
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Q, P = sampling_loc.shape[1], sampling_loc.shape[4]
    _req(tuple(sampling_loc.shape) == (B, Q, M, L, P, 2), "sampling_loc must be [B,Q,M,L,P,2]")
    _req(tuple(attn_weight.shape) == (B, Q, M, L, P), "attn_weight must be [B,Q,M,L,P]")
    step = min(B, int(im2col_step))
    _req(B == 0 or (step > 0 and B % step == 0), f"batch({B}) must divide im2col_step({step})")
    out = torch.empty((B, Q, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        fn = lib.mdqe_msda_forward_f32 if value.dtype == torch.float32 else lib.mdqe_msda_forward_f64
        check(fn(ptr(value), ptr(spatial_shapes), ptr(level_start_index), ptr(sampling_loc),
                 ptr(attn_weight), B, S, M, D, L, Q, P, ptr(out), cur_stream()),
              "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight], as ms_deform_attn_cuda_backward
    (src/cuda/ms_deform_attn_cuda.cu:83-153): same checks as the forward plus a contiguous CUDA grad_output."""
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)):
        _req(t.is_contiguous(), f"{name} tensor has to be contiguous")
        _req(t.is_cuda, f"{name} must be a CUDA tensor")
    _req(value.dtype in (torch.float32, torch.float64), f'"ms_deform_attn_backward_cuda" not implemented for \'{value.dtype}\'')
    _req(all(t.dtype == value.dtype for t in (sampling_loc, attn_weight, grad_output)),
         "ms_deform_attn_backward: value, sampling_loc, attn_weight and grad_output must have the same dtype")
    _req(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
         "spatial_shapes / level_start_index must be int64")
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Q, P = sampling_loc.shape[1], sampling_loc.shape[4]
    _req(tuple(sampling_loc.shape) == (B, Q, M, L, P, 2), "sampling_loc must be [B,Q,M,L,P,2]")
    _req(tuple(attn_weight.shape) == (B, Q, M, L, P), "attn_weight must be [B,Q,M,L,P]")
    _req(tuple(grad_output.shape) == (B, Q, M * D), "grad_output must be [B,Q,M*D]")
    step = min(B, int(im2col_step))
    _req(B == 0 or (step > 0 and B % step == 0), f"batch({B}) must divide im2col_step({step})")
    gv = torch.empty_like(value)
    gl = torch.empty_like(sampling_loc)
    ga = torch.empty_like(attn_weight)
    with torch.cuda.device(value.device):
        fn = lib.mdqe_msda_backward_f32 if value.dtype == torch.float32 else lib.mdqe_msda_backward_f64
        check(fn(ptr(value), ptr(spatial_shapes), ptr(level_start_index), ptr(sampling_loc),
                 ptr(attn_weight), ptr(grad_output), B, S, M, D, L, Q, P, ptr(gv), ptr(gl), ptr(ga),
                 cur_stream()), "ms_deform_attn_backward")
    return [gv, gl, ga]
