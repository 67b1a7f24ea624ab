"""CPU restatement of the COCO mask RLE used by the reference's result writer (TEST INFRASTRUCTURE).

Call site: instances_to_coco_json_video (mdqe/data/ytvis_eval.py:288-324) -> pycocotools `encode`
(mdqe/data/pycocotools/_mask.pyx:42,137-140 -> rleEncode, rleToString).  The C core (common/maskApi.c of cocoapi, third
party) is NOT in the reference tree and pycocotools is not installed here, so this file restates the published algorithm
and is **parity unpinned** for the string packing (no reference-generated vectors; the tests check the device path against
this file and through encode -> decode round trips).  The run lengths (rleEncode) ARE held to an independent implementation:
`transformers`' SAM post-processing `_mask_to_rle` (tests/test_rle_cpu.py::test_run_lengths_pinned_against_an_independent_implementation):

  rleEncode:   the mask is read in COLUMN-major order; counts = lengths of alternating runs, the first run counts zeros
               (it is 0 when the mask starts with a one).
  rleToString: counts[i] is written as x = counts[i] - (i > 2 ? counts[i-2] : 0) in a LEB128-like code of 5-bit groups,
               bit 5 = continuation, sign handled by stopping when the remaining value is all sign bits, each byte + 48.
"""
import numpy as np


def rle_counts(mask):
    """mask [H,W] (bool / 0-1) -> list of run lengths, column-major, starting with the zeros run."""
    v = np.asarray(mask, dtype=np.uint8).flatten(order="F")
    counts, prev, run = [], 0, 0
    for x in v:
        if x != prev:
            counts.append(run)
            run, prev = 0, x
        run += 1
    counts.append(run)
    return counts


def rle_to_string(counts):
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def rle_from_string(s):
    counts, p, k = [], 0, 0
    s = bytes(s)
    while p < len(s):
        x, sh, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * sh)
            more = bool(c & 0x20)
            p += 1
            sh += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * sh)
        if k > 2:
            x += counts[k - 2]
        counts.append(x)
        k += 1
    return counts


def rle_decode(counts, h, w):
    v = np.zeros(h * w, dtype=np.uint8)
    p, val = 0, 0
    for c in counts:
        v[p:p + c] = val
        p += c
        val ^= 1
    return v.reshape((h, w), order="F").astype(bool)


def encode(mask):
    """-> {"size": [h, w], "counts": bytes} like pycocotools.mask.encode on one [H,W] mask."""
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": rle_to_string(rle_counts(mask))}
