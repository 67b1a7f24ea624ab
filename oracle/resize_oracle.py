"""CPU restatement of the eval-time frame resize (TEST INFRASTRUCTURE): ResizeShortestEdgeClip.get_transform
(mdqe/data/augmentation.py:364-389, eval branch of build_augmentation :463-481) -> detectron2 ResizeTransform.apply_image,
which for uint8 images is PIL `Image.fromarray(img).resize((w, h), Image.BILINEAR)` (third party: Pillow, Resample.c).

Pillow's algorithm, restated: separable two-pass convolution, horizontal first, each pass rounded to uint8;
per output coordinate the taps are a triangle filter stretched by max(scale, 1) (antialiasing when shrinking), normalised in
double precision, then quantised to 22-bit fixed point; accumulation starts at 1 << 21 and is shifted right by 22, clipped
to [0, 255].  PINNED: Pillow itself is installed in the build container, tests/test_resize_cpu.py compares this file with
`PIL.Image.resize` bit for bit.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def shortest_edge_size(h, w, size, max_size):
    """ResizeShortestEdgeClip.get_transform (mdqe/data/augmentation.py:376-389)."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1) over the whole axis.
    -> (xmin int32 [out], n int32 [out], k int32 [out, ksize])"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32); cnt = np.zeros(out_size, np.int32); kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        x0 = int(center - support + 0.5)
        if x0 < 0:
            x0 = 0
        x1 = int(center + support + 0.5)
        if x1 > in_size:
            x1 = in_size
        n = x1 - x0
        w = np.zeros(ksize, np.float64)
        for x in range(n):
            t = abs((x + x0 - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        ww = w[:n].sum()
        if ww != 0.0:
            w[:n] = w[:n] / ww
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        xmin[xx], cnt[xx] = x0, n
        kk[xx] = q.astype(np.int32)
    return xmin, cnt, kk


def _pass(img, xmin, cnt, kk, axis):
    """img uint8 [..]; resample along `axis` (0 = rows / vertical, 1 = columns / horizontal) of an [H, W, C] array."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(xmin),) + img.shape[1:], np.int64)
    for i in range(len(xmin)):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(cnt[i]):
            acc += img[xmin[i] + t] * int(kk[i, t])
        out[i] = acc >> PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [H, W, C] -> [out_h, out_w, C], as PIL.Image.resize((out_w, out_h), BILINEAR) (horizontal pass first)."""
    h, w = img.shape[:2]
    if w != out_w:
        img = _pass(img, *coeffs(w, out_w), axis=1)
    if h != out_h:
        img = _pass(img, *coeffs(h, out_h), axis=0)
    return img
