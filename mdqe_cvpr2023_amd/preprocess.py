"""Eval-time input side on the device.

The reference's YTVISDatasetMapper resizes every decoded frame on the host before the model sees it
(mdqe/data/dataset_mapper.py:252-258: ResizeShortestEdgeClip -> detectron2 ResizeTransform -> PIL
`Image.resize(BILINEAR)` for uint8 images; size rule mdqe/data/augmentation.py:376-389).  Here the decoded frames go to
the GPU at their native size and are resized there, bit-identical to Pillow: the same separable fixed-point resampling
(csrc/spatial.hip: resize_pil_bilinear_kernel), with the coefficient tables Pillow would build computed on the host in
double precision.  Normalisation and zero padding stay fused into the stem's im2col kernel.
"""
import numpy as np
import torch

from ._lib import check, cur_stream, lib, ptr

PRECISION_BITS = 32 - 8 - 2
_tables = {}


def shortest_edge_size(h, w, size, max_size):
    """ResizeShortestEdgeClip.get_transform (mdqe/data/augmentation.py:376-389): output (h, w)."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc (Resample.c), bilinear filter, whole axis; vectorised.
    -> xmin int32 [out], cnt int32 [out], k int32 [out, ksize]."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    x0 = np.maximum((center - support + 0.5).astype(np.int64), 0)
    x1 = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = x1 - x0
    t = np.arange(ksize, dtype=np.float64)[None]
    arg = np.abs((t + x0[:, None] - center[:, None] + 0.5) / filterscale)
    w = np.where((arg < 1.0) & (t < n[:, None]), 1.0 - arg, 0.0)
    ww = w.sum(1, keepdims=True)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    q = (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)           # weights are >= 0 for the triangle filter
    return x0.astype(np.int32), n.astype(np.int32), q.astype(np.int32)


def _dev_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _tables:
        while len(_tables) >= 64:                        # a few KB each; bounded all the same (oldest first)
            _tables.pop(next(iter(_tables)))
        x0, n, k = pil_bilinear_coeffs(in_size, out_size)
        _tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (x0, n, k)) + (int(k.shape[1]),)
    return _tables[key]


def resize_frames(frames, out_h, out_w):
    """frames: CUDA uint8 [NI, C, H, W] (contiguous) -> [NI, C, out_h, out_w] uint8, == PIL Image.resize(BILINEAR) per frame."""
    if not (frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.is_contiguous()):
        raise RuntimeError("resize_frames: expected a contiguous CUDA uint8 [NI,C,H,W] tensor")
    NI, C, H, W = frames.shape
    if (H, W) == (out_h, out_w):
        return frames
    xm, xc, xk, kxs = _dev_tables(W, out_w, frames.device)
    ym, yc, yk, kys = _dev_tables(H, out_h, frames.device)
    out = torch.empty(NI, C, out_h, out_w, dtype=torch.uint8, device=frames.device)
    check(lib.mdqe_resize_pil_bilinear_u8(ptr(frames), C * H * W, NI, C, H, W, out_h, out_w, ptr(xm), ptr(xc), ptr(xk), kxs,
                                          ptr(ym), ptr(yc), ptr(yk), kys, ptr(out), cur_stream()), "resize_pil_bilinear")
    return out


def resize_shortest_edge(frames, min_size, max_size):
    """The eval augmentation of the reference on device-resident frames: [NI,C,H,W] uint8 -> resized uint8 frames."""
    oh, ow = shortest_edge_size(int(frames.shape[-2]), int(frames.shape[-1]), min_size, max_size)
    return resize_frames(frames, oh, ow)
