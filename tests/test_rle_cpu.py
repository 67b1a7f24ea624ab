"""Host side of the COCO RLE path (mdqe_cvpr2023_amd/rle.py) against the oracle restatement of cocoapi's rleEncode /
rleToString (oracle/rle_oracle.py; parity unpinned -- pycocotools' C core is neither in the reference tree nor installed)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rle_oracle as RO  # noqa: E402
from mdqe_cvpr2023_amd import rle as R  # noqa: E402


def test_known_small_vectors():
    assert RO.encode(np.zeros((4, 4), bool))["counts"] == b"`0"            # one run of 16 zeros
    m = np.zeros((5, 7), bool); m[1:4, 2:5] = 1
    assert RO.rle_counts(m) == [11, 3, 2, 3, 2, 3, 11]
    assert R.encode_dense(m)["counts"].encode() == RO.encode(m)["counts"]
    one = np.ones((3, 2), bool)
    assert RO.rle_counts(one) == [0, 6]                                     # starts with a one: empty zeros run first


def test_vectorised_packing_matches_the_scalar_restatement():
    rng = np.random.RandomState(3)
    for _ in range(100):
        h, w = rng.randint(1, 60), rng.randint(1, 60)
        m = rng.rand(h, w) < rng.rand()
        a, b = R.encode_dense(m), RO.encode(m)
        assert a["size"] == b["size"] and a["counts"].encode() == b["counts"]
        assert (RO.rle_decode(RO.rle_from_string(b["counts"]), h, w) == m).all()
    big = np.zeros((360, 640), bool); big[100:300, 50:600] = 1; big[120:130, :] = 0; big[0, 0] = 1; big[-1, -1] = 1
    assert R.encode_dense(big)["counts"].encode() == RO.encode(big)["counts"]     # long runs: multi-group codes, negative deltas


def test_positions_to_counts_and_overflow():
    pos = np.array([[3, 5, 0, 0], [0, 0, 0, 0]]); n_pos = np.array([2, 0])
    counts, lengths = R.positions_to_counts(pos, n_pos, 12)
    assert lengths.tolist() == [3, 1] and counts.tolist() == [3, 2, 7, 12]
    import pytest
    with pytest.raises(OverflowError):
        R.positions_to_counts(pos, np.array([5, 0]), 12)


def test_result_writer_layout():
    out = {"pred_scores": [0.9], "pred_labels": [3], "pred_masks": [np.zeros((2, 4, 4), bool)]}
    rec = R.instances_to_coco_json_video([{"video_id": 11, "length": 2}], out)
    assert rec == [{"video_id": 11, "score": 0.9, "category_id": 3,
                    "segmentations": [{"size": [4, 4], "counts": "`0"}, {"size": [4, 4], "counts": "`0"}]}]


def test_run_lengths_pinned_against_an_independent_implementation():
    """The rleEncode half (column-major run lengths, zeros run first) held to an implementation that is neither the reference's nor
    ours: `_mask_to_rle` of the `transformers` package in this image (its SAM post-processing, 'in the format expected by pycoco
    tools').  The string packing of rleToString has no independent implementation here and stays parity-unpinned."""
    import pytest
    import torch
    sam = pytest.importorskip("transformers.models.sam.image_processing_pil_sam")      # (the torchvision-free twin)
    rng = np.random.RandomState(5)
    masks = [np.zeros((6, 9), bool), np.ones((6, 9), bool)]
    for _ in range(60):
        h, w = rng.randint(1, 50), rng.randint(1, 50)
        masks.append(rng.rand(h, w) < rng.rand())
    blob = np.zeros((40, 64), bool); blob[5:30, 10:50] = 1; blob[0, 0] = 1
    masks.append(blob)
    for m in masks:
        ref = sam._mask_to_rle(torch.from_numpy(m)[None])[0]
        assert ref["size"] == list(m.shape)
        assert RO.rle_counts(m) == [int(c) for c in ref["counts"]]
        # ... and the product's packing decodes back to those run lengths
        assert RO.rle_from_string(R.encode_dense(m)["counts"].encode()) == [int(c) for c in ref["counts"]]
        assert (np.asarray(sam._rle_to_mask(ref)) == m).all()
