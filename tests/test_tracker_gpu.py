"""GPU: the product tracker (native host core + the HIP bank kernels trk_siou / trk_accumulate / trk_window_mean / trk_carry)
against the reference OverTracker's recorded behaviour on the crafted sequence (late, vanishing and duplicate objects,
3 windows; fixture tracker_seq made by running mdqe/tracking/OverTracker.py:115-225), clip by clip and as runs of clips
through ONE native call (update_many)."""
import pytest
import torch

from _golden import Fixture, maxdiff

pytestmark = pytest.mark.gpu


def _clips(fx):
    from mdqe_cvpr2023_amd.tracking import Clips
    out = []
    for i in range(fx.i("n_clips")):
        clip = {k: fx.t(f"clip{i}::{k}") for k in ("scores", "pred_classes", "cls_probs", "pred_masks", "query_embeds")}
        clip["pred_masks"] = clip["pred_masks"].cuda()
        fi = fx.t(f"clip{i}::frame_idx").tolist()
        out.append((fi, Clips(fi, clip)))
    return out


def _tracker(fx):
    from mdqe_cvpr2023_amd.tracking import OverTracker
    return OverTracker(fx.i("MAXI"), fx.i("T"), fx.i("WIN"), 1, fx.i("K"), 4, fx.i("E"), tuple(int(v) for v in fx.z["HW"]),
                       torch.device("cuda"), fx.f("THR"))


@pytest.mark.parametrize("fixture", ["tracker_seq", "tracker_long"])
@pytest.mark.parametrize("mode", ["per_clip", "runs"])
def test_hip_tracker_matches_reference_sequence(mode, fixture):
    """tracker_long: 44 frames, six window flushes, an instance that leaves for longer than a window and returns under its old id, one
    that returns under a new id (reference-run golden, oracle/make_golden.py gen_tracker_long)."""
    fx = Fixture(fixture)
    trk = _tracker(fx)
    clips = _clips(fx)
    saved, n = 0, len(clips)
    run = []
    for i, (fi, clip) in enumerate(clips):
        last = i == n - 1
        flush = last or (fi[0] + 1 >= fx.i("WIN") * (saved + 1))
        if mode == "per_clip":
            trk.update(clip)
            assert trk.num_inst == fx.i(f"clip{i}::num_inst_after")
        else:
            run.append(clip)
            if flush:
                trk.update_many(run)                       # every clip since the previous flush in one native call
                run = []
                assert trk.num_inst == fx.i(f"clip{i}::num_inst_after")
        if flush:
            c, m = trk.get_result(last)
            assert maxdiff(c, fx.t(f"win{saved}::cls")) < 1e-6
            ref = fx.t(f"win{saved}::masks")
            assert m.shape == ref.shape and maxdiff(m.cpu(), ref) < 1e-5
            saved += 1
    assert saved == fx.i("n_windows")


def test_hip_tracker_more_than_128_new_tracks_in_one_clip():
    """A clip may deliver up to min(n_query, 10*DETECTIONS_PER_IMAGE) = 150 instances and on the first clip each becomes a track
    (the accumulate kernel takes 128 pairs per launch)."""
    import numpy as np
    from mdqe_cvpr2023_amd.tracking import Clips, OverTracker
    n, hw = 150, (8, 12)
    trk = OverTracker(160, 2, 4, 1, 3, 4, 16, hw, torch.device("cuda"), 0.1)
    g = torch.Generator().manual_seed(0)
    masks = torch.randn(n, 2, *hw, generator=g).cuda()
    host = {"scores": np.full(n, 0.9, np.float32), "cls_probs": np.full((n, 3), 0.5, np.float32),
            "query_embeds": torch.randn(n, 16, generator=g).numpy()}
    trk.update(Clips([0, 1], {"pred_masks": masks, "host": host}))
    assert trk.num_inst == n
    c, m = trk.get_result(True)
    assert m.shape == (n, 2) + hw and torch.equal(m, masks)


@pytest.mark.parametrize("many", [False, True])
def test_hip_tracker_equals_the_oracle_on_random_sequences(many):
    """Random clip sequences (tools/fuzz_tracker.py) on the HIP bank, clip by clip and through update_many -- among them the two edge
    cases the fuzz found in round 2: a first clip without any instance (seeds 284, 286) and a window flushed without any track (82, 99)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_tracker
    for seed in (82, 99, 284, 286, 0, 1, 2, 3, 4, 5, 6, 7):
        assert fuzz_tracker.run(seed, gpu=True, many=many) is None, seed


@pytest.mark.parametrize("fast,spin_us", [(1, 2000), (1, 0), (0, 0)])
def test_counts_fast_path_classic_path_and_sync_fallback_agree(fast, spin_us):
    """Round 5: an update takes its sign-intersection counts through ONE kernel that delivers them to host-coherent memory and raises a
    flag the host polls (mdqe_trk_siou_host_f32 / mdqe_trk_wait_flag) instead of memset + kernel + copy + synchronize.  The same recorded
    reference sequence and random sequences through (a) the fast path, (b) the fast path with a zero spin budget -- the wait falls back to
    a stream synchronize at once -- and (c) the classic path: identical decisions and window results; and the counts themselves equal the
    classic kernel's on a crafted pair of banks, launch after launch on the same accumulator (left zero by the kernel)."""
    import os, sys
    import ctypes
    from mdqe_cvpr2023_amd._lib import check, cur_stream, lib, ptr
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_tracker
    lib.mdqe_debug_trk_fast(fast); lib.mdqe_debug_trk_spin_us(spin_us)
    try:
        for mode in ("per_clip", "runs"):
            for fixture in ("tracker_seq", "tracker_long"):
                test_hip_tracker_matches_reference_sequence(mode, fixture)
        for seed in (82, 284, 0, 1, 2):
            assert fuzz_tracker.run(seed, gpu=True, many=True) is None, seed
    finally:
        lib.mdqe_debug_trk_fast(1); lib.mdqe_debug_trk_spin_us(2000)
    if fast and spin_us:
        import numpy as np
        g = torch.Generator(device="cuda").manual_seed(3)
        acc = torch.zeros(4096, device="cuda"); ticket = torch.zeros(16, dtype=torch.int32, device="cuda")
        words = torch.zeros(4096, dtype=torch.int64, pin_memory=True)
        for seq, (ns, ni, n) in enumerate([(3, 5, 4 * 96 * 160), (7, 7, 3 * 96 * 160), (1, 1, 1024), (12, 2, 4 * 1000)], start=1):
            a = torch.randn(ns, n, device="cuda", generator=g); b = torch.randn(ni, n, device="cuda", generator=g)
            want = torch.zeros(ns * ni * 3, device="cuda")
            check(lib.mdqe_trk_siou_f32(ptr(a), n, ns, ptr(b), n, ni, n, ptr(want), cur_stream()), "siou")
            check(lib.mdqe_trk_siou_host_f32(ptr(a), n, ns, ptr(b), n, ni, n, ptr(acc), ptr(ticket), words.data_ptr(), seq, cur_stream()), "siou_host")
            got = np.zeros(ns * ni * 3, dtype=np.float32)
            check(lib.mdqe_trk_wait_counts(words.data_ptr(), ns * ni * 3, seq, 100000, got.ctypes.data, cur_stream()), "wait_counts")
            assert np.array_equal(got, want.cpu().numpy())
            torch.cuda.synchronize()
            assert not bool(acc.any()) and int(ticket[0]) == 0          # left clean for the next launch
