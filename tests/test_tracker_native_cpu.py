"""CPU: the native rectangular assignment (csrc/tracker_native.hip, mdqe_lsap_f64) against the reference's own dependency,
scipy.optimize.linear_sum_assignment (mdqe/tracking/OverTracker.py:159) -- identical pairs, including tie-heavy costs --
and the native bi-softmax / decision core on hand-made cases."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from mdqe_cvpr2023_amd.tracking import lsap


@pytest.mark.parametrize("maximize", [False, True])
def test_lsap_equals_scipy_on_random_and_tied_costs(maximize):
    rng = np.random.RandomState(0)
    shapes = [(1, 1), (1, 5), (5, 1), (3, 3), (7, 4), (4, 7), (20, 20), (34, 15), (15, 34), (120, 60), (2, 150)]
    for nr, nc in shapes:
        for kind in range(6):
            if kind == 0:
                c = rng.rand(nr, nc)
            elif kind == 1:
                c = rng.randint(0, 3, (nr, nc)).astype(np.float64)            # heavy ties
            elif kind == 2:
                c = rng.rand(nr, nc) * (rng.rand(nr, nc) > 0.7)                 # mostly zeros, as the tracker's thresholded scores
            elif kind == 3:
                c = np.zeros((nr, nc))
            elif kind == 4:
                c = rng.rand(nr, nc).astype(np.float32).astype(np.float64)      # fp32 values, as the tracker feeds them
            else:
                c = np.round(rng.rand(nr, nc), 1)
            r0, c0 = linear_sum_assignment(c, maximize=maximize)
            r1, c1 = lsap(c, maximize=maximize)
            assert r0.tolist() == r1.tolist() and c0.tolist() == c1.tolist(), (nr, nc, kind)


def test_lsap_empty_and_infeasible():
    r, c = lsap(np.zeros((0, 4)))
    assert len(r) == 0 and len(c) == 0
    from mdqe_cvpr2023_amd._lib import MdqeError
    with pytest.raises(MdqeError):
        lsap(np.array([[np.nan, 1.0], [1.0, 2.0]]))


def test_native_core_first_clip_new_ids_and_duplicates():
    """Two clips by hand: first clip -> ids 0..n-1; second clip re-detects both (matched by embedding + IoU), plus a confident
    new object (new id) and an unconfident one (discarded)."""
    import torch
    from _standins import Clips, TorchBankTracker
    E, K, hw = 8, 3, (4, 4)
    trk = TorchBankTracker(10, 2, 4, 1, K, 4, E, hw, torch.device("cpu"), 0.1)
    emb = np.eye(E, dtype=np.float32)[:4] * 6

    def masks(rows, T):
        m = torch.full((len(rows), T) + hw, -3.0)
        for i, r in enumerate(rows):
            m[i, :, r, :] = 3.0
        return m
    c0 = {"scores": np.array([0.9, 0.8], np.float32), "cls_probs": np.full((2, K), 0.3, np.float32), "query_embeds": emb[:2]}
    trk.update(Clips([0, 1], {"pred_masks": masks([0, 1], 2), "host": c0}))
    assert trk.num_inst == 2
    c1 = {"scores": np.array([0.9, 0.8, 0.7, 0.15], np.float32), "cls_probs": np.full((4, K), 0.3, np.float32),
          "query_embeds": np.stack([emb[1], emb[0], emb[2], emb[3]])}
    trk.update(Clips([1, 2], {"pred_masks": masks([1, 0, 2, 3], 2), "host": c1}))
    assert trk.num_inst == 3                                  # object 2 is new; object 3 (score 0.15 <= 2*thr) is dropped
    c, m = trk.get_result(is_last_clip=True)
    assert m.shape == (3, 3) + hw and c.shape == (3, K)
    assert bool((m[0, :, 0] > 0).all()) and bool((m[1, :, 1] > 0).all())      # ids kept their rows across the swap
    assert bool((m[2, 1:, 2] > 0).all()) and bool((m[2, 0] == 0).all())


def test_native_core_equals_the_oracle_tracker_on_random_sequences():
    """40 random clip sequences (persistent / appearing / vanishing / duplicated objects, random clip length and window): the native
    core with a torch stand-in bank against the oracle's restatement of OverTracker (itself held to the reference's recorded sequence in
    test_oracle_golden.py) -- instance counts after every clip, class scores and mean logits of every window."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_tracker
    for seed in range(40):
        assert fuzz_tracker.run(seed) is None, seed


def test_a_refused_clip_leaves_the_tracker_untouched():
    """`decide` validates before it mutates.  A first clip with more instances than MAX_NUM_INSTANCES, and a clip beyond
    the window's memory, are refused (the reference raises there too) -- and the object, which the product reuses, is exactly as
    before: a valid clip afterwards behaves as if the refused one had never been offered."""
    import torch
    from _standins import Clips, TorchBankTracker
    from mdqe_cvpr2023_amd._lib import MdqeError
    E, K, hw = 8, 3, (4, 4)

    def clip(frames, n, seed):
        rng = np.random.RandomState(seed)
        host = {"scores": np.full(n, 0.9, np.float32), "cls_probs": rng.rand(n, K).astype(np.float32),
                "query_embeds": (np.eye(E, dtype=np.float32)[np.arange(n) % E] * 6 + rng.rand(n, E).astype(np.float32) * 0.01)}
        return Clips(frames, {"pred_masks": torch.from_numpy(rng.randn(n, len(frames), *hw).astype(np.float32)), "host": host})

    a = TorchBankTracker(3, 2, 4, 1, K, 4, E, hw, torch.device("cpu"), 0.1)
    b = TorchBankTracker(3, 2, 4, 1, K, 4, E, hw, torch.device("cpu"), 0.1)
    with pytest.raises(MdqeError):
        a.update(clip([0, 1], 5, 0))                            # 5 first-clip instances > max_inst 3
    assert a.num_inst == 0 and a.num_clip == 0
    for t in (a, b):
        t.update(clip([0, 1], 2, 1))
    with pytest.raises(MdqeError):
        a.update(clip([40, 41], 2, 2))                          # frames far beyond the window's memory
    assert (a.num_inst, a.num_clip) == (b.num_inst, b.num_clip) == (2, 1)
    for t in (a, b):
        t.update(clip([1, 2], 3, 3))
    ca, ma = a.get_result(is_last_clip=True)
    cb, mb = b.get_result(is_last_clip=True)
    assert torch.equal(ca, cb) and torch.equal(ma, mb) and a.num_inst == b.num_inst
