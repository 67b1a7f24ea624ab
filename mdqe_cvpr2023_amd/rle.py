"""COCO mask RLE for the result writer.

The reference turns every [L,H,W] boolean video mask into per-frame COCO RLEs on the host
(instances_to_coco_json_video, mdqe/data/ytvis_eval.py:288-324: pycocotools `encode` of a Fortran-ordered uint8 copy per
frame).  Here the run boundaries come straight from the device (ops.final_masks_rle: the dense mask is never written nor
copied) and only the LEB128-like string packing (cocoapi rleToString) runs on the host, vectorised over all runs of a
video.  `instances_to_coco_json_video` is the drop-in for the reference function: it takes the model output with
`pred_rles` (model.rle_output = True) or, as a fallback, dense `pred_masks`.
"""
import numpy as np


def counts_to_strings(counts, lengths):
    """counts: int64 [sum(lengths)] run lengths of consecutive masks; lengths: runs per mask -> list of bytes (rleToString).
    Every run i > 2 of a mask is stored as the difference to run i-2; values are cut into 5-bit groups, bit 5 flags a
    continuation, a group sequence ends when the rest is all sign bits; each byte is offset by 48."""
    counts = np.asarray(counts, dtype=np.int64)
    lengths = np.asarray(lengths, dtype=np.int64)
    n = int(counts.shape[0])
    if n == 0:
        return [b"" for _ in lengths]
    starts = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    idx_in_mask = np.arange(n, dtype=np.int64) - np.repeat(starts, lengths)
    x = counts.copy()
    d = np.zeros_like(counts)
    d[2:] = counts[:-2]
    x = np.where(idx_in_mask > 2, counts - d, counts)
    chars = np.zeros((n, 8), dtype=np.uint8)
    valid = np.zeros((n, 8), dtype=bool)
    alive = np.ones(n, dtype=bool)
    for r in range(8):                                   # 8 groups cover 40 bits
        c = x & 0x1f
        x = x >> 5                                        # arithmetic shift
        more = np.where((c & 0x10) != 0, x != -1, x != 0)
        ch = (c | (more.astype(np.int64) << 5)) + 48
        chars[:, r] = np.where(alive, ch, 0)
        valid[:, r] = alive
        alive = alive & more
        if not alive.any():
            break
    per_run = valid.sum(1)
    flat = chars[valid]                                   # row-major: groups of a run stay together, runs stay in order
    run_off = np.concatenate([[0], np.cumsum(per_run)])
    out = []
    buf = flat.tobytes()
    for s, l in zip(starts, lengths):
        out.append(buf[run_off[s]:run_off[s + l]])
    return out


def positions_to_counts(pos, n_pos, total):
    """pos [n_masks, cap] change positions (column-major pixel indices), n_pos [n_masks] -> (counts, lengths) of all masks."""
    pos = np.asarray(pos, dtype=np.int64)
    n_pos = np.asarray(n_pos, dtype=np.int64)
    n_masks, cap = pos.shape
    if (n_pos > cap).any():
        raise OverflowError("RLE position buffer too small")
    lengths = n_pos + 1
    col = np.arange(cap + 1, dtype=np.int64)[None]
    ext = np.concatenate([pos, np.zeros((n_masks, 1), dtype=np.int64)], 1)
    ext = np.where(col < n_pos[:, None], ext, total)              # the position after the last change is the mask's end
    prev = np.concatenate([np.zeros((n_masks, 1), dtype=np.int64), ext[:, :-1]], 1)
    runs = ext - prev
    keep = col <= n_pos[:, None]
    return runs[keep], lengths


def encode_dense(mask):
    """Host fallback: one [H,W] boolean mask (numpy / CPU tensor) -> {"size", "counts": str}, pycocotools-style."""
    m = np.asarray(mask).astype(np.uint8)
    v = m.flatten(order="F")
    change = np.flatnonzero(np.diff(np.concatenate([[0], v])) != 0)
    counts, lengths = positions_to_counts(change[None], [len(change)], v.shape[0])
    return {"size": [int(m.shape[0]), int(m.shape[1])], "counts": counts_to_strings(counts, lengths)[0].decode("utf-8")}


def instances_to_coco_json_video(inputs, outputs):
    """Drop-in for mdqe/data/ytvis_eval.py:288-324: list of {"video_id", "score", "category_id", "segmentations"}."""
    assert len(inputs) == 1, "More than one inputs are loaded for inference!"
    video_id = inputs[0]["video_id"]
    res = []
    rles = outputs.get("pred_rles")
    for i, (s, l) in enumerate(zip(outputs["pred_scores"], outputs["pred_labels"])):
        segms = rles[i] if rles is not None else [encode_dense(m) for m in outputs["pred_masks"][i]]
        res.append({"video_id": video_id, "score": s, "category_id": l, "segmentations": segms})
    return res
