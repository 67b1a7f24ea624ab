"""CPU, world_size 2 over gloo: sharded clip production + variable-length all-gather + tracker replay must
equal the single-process schedule (the N>1 path of bench.py / SURVEY.md §8e)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from mdqe_cvpr2023_amd import sharding
from mdqe_cvpr2023_amd.config import MDQEConfig
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _standins import Clips, TorchBankTracker as OverTracker   # the native host core + a torch stand-in for the HIP bank

CFG = MDQEConfig(backbone="custom", n_frames_test=3, n_frames_window_test=4, num_classes=5, hidden_dim=32, n_max_inst=16,
                 apply_cls_thres=0.1)
HW = (6, 8)
L = 13


def clip_schedule(L, T, stride):
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    return MDQE.clip_schedule(L, T, stride)


def fake_result(start, end):
    """Deterministic per-clip 'inference_clip' output keyed by the clip start (3 persistent objects)."""
    g = torch.Generator().manual_seed(1000 + start)
    n = 2 + start % 2
    base = torch.eye(CFG.hidden_dim)[:3] * 4
    emb = base[:n] + 0.05 * torch.randn(n, CFG.hidden_dim, generator=g)
    masks = torch.full((n, end - start) + HW, -3.0)
    for i in range(n):
        masks[i, :, i * 2:i * 2 + 2, :] = 3.0
    cls = torch.rand(n, CFG.num_classes, generator=g) * 0.3
    cls[torch.arange(n), torch.arange(n)] = 0.8
    sc, lab = cls.max(-1)
    return {"scores": sc, "pred_classes": lab, "cls_probs": cls, "query_embeds": emb, "pred_masks": masks}


def replay(results):
    trk = OverTracker(CFG.n_max_inst, CFG.n_frames_test, CFG.n_frames_window_test, 1, CFG.num_classes, 4, CFG.hidden_dim, HW,
                      torch.device("cpu"), CFG.apply_cls_thres)
    outs, saved = [], 0
    for s, e, last, r in results:
        trk.update(Clips(range(s, e), r))
        if last or (s + 1 >= CFG.n_frames_window_test * (saved + 1)):
            c, m = trk.get_result(last)
            outs.append((c.clone(), m.clone()))
            saved += 1
    return outs


def worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = CFG.n_frames_test
    clips = clip_schedule(L, T, 1)
    mine = sharding.owned_clips(clips, L, world, rank)
    f0, f1 = sharding.frame_range(L, world, rank, T)
    assert all(f0 <= s and e <= f1 for s, e, _ in mine)            # halo covers every owned clip
    local = [(s, e, l, fake_result(s, e)) for s, e, l in mine]
    proto = {"scores": ((), torch.float32), "pred_classes": ((), torch.int64), "cls_probs": ((CFG.num_classes,), torch.float32),
             "query_embeds": ((CFG.hidden_dim,), torch.float32), "pred_masks": ((T,) + HW, torch.float32)}
    merged = sharding.all_gather_clips(local, T, dist, world, torch.device("cpu"), proto)
    rooted = sharding.all_gather_clips(local, T, dist, world, torch.device("cpu"), proto, root=0, rank=rank)   # gather-to-root form
    assert (rooted is None) == (rank != 0)
    if rank == 0:
        assert [(s, e, l) for s, e, l, _ in rooted] == [(s, e, l) for s, e, l, _ in merged]
        for (_, _, _, a), (_, _, _, b) in zip(rooted, merged):
            assert all(torch.equal(a[f], b[f]) for f in sharding.FIELDS)
    outs = replay(merged)
    torch.save((rank, [(s, e, l) for s, e, l, _ in merged], outs), os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    got = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(2)]
    clips = clip_schedule(L, CFG.n_frames_test, 1)
    ref = replay([(s, e, l, fake_result(s, e)) for s, e, l in clips])
    for rank, order, outs in got:
        assert order == clips
        assert len(outs) == len(ref)
        for (c, m), (cr, mr) in zip(outs, ref):
            assert torch.allclose(c, cr) and torch.equal(m, mr)


def test_ranges_cover_all_clips_once():
    for L_, world in ((13, 2), (120, 8), (7, 4), (4, 8), (240, 3)):
        clips = clip_schedule(L_, 4, 1)
        seen = []
        for r in range(world):
            seen += sharding.owned_clips(clips, L_, world, r)
        assert sorted(seen) == clips


def test_round_robin_plan_covers_all_clips_in_order():
    for L_, T, chunk, world in ((13, 3, 4, 2), (120, 4, 30, 8), (960, 4, 30, 8), (7, 4, 30, 4), (61, 4, 30, 3)):
        plan = sharding.chunk_plan(L_, T, 1, chunk)
        clips = clip_schedule(L_, T, 1)
        flat = [c for ch in plan for c in ch[0]]
        assert flat == clips                                         # global order preserved, every clip once
        for cl, f0, f1 in plan:
            assert all(f0 <= s and e <= f1 for s, e, _ in cl)       # halo covers every clip of the chunk
        owned = sorted(g for r in range(world) for g in sharding.owned_chunks(plan, world, r))
        assert owned == list(range(len(plan)))


def test_halo_exchange_plan_partitions_frames_and_clips():
    """chunk_plan(halo_exchange=True): the chunks partition the frames (no frame twice), every clip belongs to the chunk of its
    LAST frame, a chunk's clips start at most T-1 frames before it, and every chunk holds a whole clip of its own."""
    for L_, T, chunk in ((13, 3, 4), (120, 4, 30), (960, 4, 60), (61, 4, 30), (7, 4, 30), (33, 4, 30), (11, 3, 4), (14, 3, 4), (5, 3, 4)):
        plan = sharding.chunk_plan(L_, T, 1, chunk, halo_exchange=True)
        clips = clip_schedule(L_, T, 1)
        assert [c for ch in plan for c in ch[0]] == clips
        assert plan[0][1] == 0 and plan[-1][2] == L_ and all(a[2] == b[1] for a, b in zip(plan[:-1], plan[1:]))
        for cl, f0, f1 in plan:
            assert all(f0 <= e - 1 < f1 and s >= f0 - (T - 1) for s, e, _ in cl)
            assert any(s >= f0 and e - s == min(T, L_) for s, e, _ in cl)


def test_halo_recompute_fraction():
    T = 4
    plan = sharding.chunk_plan(120, T, 1, 30)
    assert abs(sharding.halo_recompute_frac(plan, 120) - 9 / 120) < 1e-9           # three interior chunk edges x (T-1) frames
    assert sharding.halo_recompute_frac(sharding.chunk_plan(120, T, 1, 30, halo_exchange=True), 120) == 0.0


def test_decreasing_rounds_plan():
    """round_sizes: per-rank frames split into rounds of decreasing chunks (the last round's replay is the one nobody hides);
    chunk_plan with per-round sizes: round q = chunks q*world .. q*world+world-1 of that size; all clips once, in order, both forms."""
    assert sharding.round_sizes(120, 4) == [69, 34, 17]
    assert sharding.round_sizes(30, 4) == [20, 10] or sum(sharding.round_sizes(30, 4)) == 30
    for per in (5, 12, 24, 62, 120, 240, 1000):
        sz = sharding.round_sizes(per, 4)
        assert sum(sz) == per and sz == sorted(sz, reverse=True) and (len(sz) == 1 or sz[-1] >= 12)
    for L_, T, world in ((960, 4, 8), (240, 4, 2), (500, 4, 8), (123, 3, 4)):
        sz = sharding.round_sizes(-(-L_ // world), T)
        for halo in (False, True):
            plan = sharding.chunk_plan(L_, T, 1, sz, halo_exchange=halo, world=world)
            assert [c for ch in plan for c in ch[0]] == clip_schedule(L_, T, 1)
            if halo:
                assert plan[0][1] == 0 and plan[-1][2] == L_ and all(a[2] == b[1] for a, b in zip(plan[:-1], plan[1:]))
                lens = [b - a for _, a, b in plan]
            else:
                lens = [b - a - (T - 1) for _, a, b in plan]
            for q in range(len(sz)):                                   # whole rounds of equal chunks (the video's end may cut the last ones)
                assert all(x == sz[q] for x in lens[q * world:(q + 1) * world][:-1] if q * world + world < len(plan))
    assert sharding.chunk_plan(13, 3, 1, 4) == sharding.chunk_plan(13, 3, 1, [4], world=2)


def test_replay_thread_keeps_order_and_surfaces_errors():
    class Merger:
        def __init__(self):
            self.seen = []

        def feed(self, s, e, last, res):
            if res == "boom":
                raise ValueError("boom")
            self.seen.append(s)

        def finish(self):
            return self.seen

    m = Merger()
    t = sharding.ReplayThread(m, torch.device("cpu"))
    t.put([(0, 3, False, None), (1, 4, False, None)])
    t.put([(2, 5, True, None)])
    assert t.finish() == [0, 1, 2]
    t = sharding.ReplayThread(Merger(), torch.device("cpu"))
    t.put([(0, 3, False, "boom")])
    import pytest
    with pytest.raises(ValueError):
        t.finish()


# ---- run_round_robin_stream with a stand-in model (the schedule, the collectives and the hand-out order; no kernels) ----------
class _FakeGeo:
    Hp, Wp = HW[0] * 4, HW[1] * 4


class _FakeEngine:
    def geometry(self, h, w):
        return _FakeGeo()


class _FakeModel:
    cfg = CFG
    engine = _FakeEngine()

    def __init__(self, log):
        self.log = log

    def iter_clip_results(self, frames, clips, f0, trace=None, primed=False, on_frames_queued=None, side_streams=True, split_small=False):
        self.log.append(("frames_queued", clips[0][0]))
        if primed:
            yield None
        for s, e, l in clips:
            yield s, e, l, fake_result(s, e)


class _FakeMerger:
    """Stands in for meta_arch.ClipMerger: records the clip order it is fed and replays the tracker on the CPU."""
    use_side = False

    def __init__(self, model, frame_hw, out_size, mask_hw, n_frames=None, emit_masks=True):
        self.items, self.dev = [], torch.device("cpu")

    def feed(self, s, e, l, res):
        self.items.append((s, e, l, res))
        return bool(l)

    def finish(self):
        return [(s, e, l) for s, e, l, _ in self.items], replay(self.items)


STREAM_L = (13, 5, 22)            # 4, 1 and 6 chunks of 4 frames


def stream_worker(rank, world, port, outdir, root_only, lengths=None):
    import torch.distributed as dist
    import mdqe_cvpr2023_amd.meta_arch as MA
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    MA.ClipMerger = _FakeMerger
    log = []
    model = _FakeModel(log)
    jobs = []
    for Lv in (lengths or STREAM_L):
        plan = sharding.chunk_plan(Lv, CFG.n_frames_test, 1, 4)
        frames = {g: torch.zeros(plan[g][2] - plan[g][1], 3, HW[0] * 4, HW[1] * 4) for g in sharding.owned_chunks(plan, world, rank)}
        jobs.append((frames, plan, torch.zeros(0, 3, HW[0] * 4, HW[1] * 4)))
    outs, order, stats = [], [], []
    for k, out in enumerate(sharding.run_round_robin_stream(model, jobs, rank, world, dist, (HW[0] * 4, HW[1] * 4), root_only=root_only,
                                                            stats=stats)):
        outs.append(out)
        order.append((k, len(log)))               # how much per-frame work had been queued when video k's result came out
    # the per-video timing breakdown bench.py gathers into `scaling_breakdown`: one entry per video on EVERY rank
    assert len(stats) == len(jobs)
    for st, (_, plan, _) in zip(stats, jobs):
        assert set(st) == {"compute", "pack", "gather_wait", "gather_payload", "feed", "replay_exposed", "replay_busy", "rounds"}
        assert st["rounds"] == -(-len(plan) // world) and all(v >= 0 for v in st.values())
    torch.save((outs, order, log), os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _run_stream(tmp_path, root_only, world=2, lengths=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=stream_worker, args=(r, world, port, str(tmp_path), root_only, lengths)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(world)]


def test_round_robin_stream_two_ranks_all_replay(tmp_path):
    got = _run_stream(tmp_path, False)
    for outs, order, log in got:
        assert len(outs) == len(STREAM_L)
        for Lv, (clip_order, tracks) in zip(STREAM_L, outs):
            clips = clip_schedule(Lv, CFG.n_frames_test, 1)
            assert clip_order == clips                                  # global clip order on every rank
            ref = replay([(s, e, l, fake_result(s, e)) for s, e, l in clips])
            assert len(tracks) == len(ref)
            for (c, m), (cr, mr) in zip(tracks, ref):
                assert torch.allclose(c, cr) and torch.equal(m, mr)
    # look-ahead: when video 0's result is handed out, this rank has already queued per-frame work of video 1 (rank 0 owns
    # its only chunk) or of video 2 (rank 1 owns nothing of video 1 and moves straight on)
    outs, order, log = got[0]
    n_video0 = len(sharding.owned_chunks(sharding.chunk_plan(STREAM_L[0], CFG.n_frames_test, 1, 4), 2, 0))
    assert order[0][1] > n_video0


def test_round_robin_stream_two_ranks_root_only(tmp_path):
    got = _run_stream(tmp_path, True)
    outs0, outs1 = got[0][0], got[1][0]
    assert all(o is None for o in outs1) and len(outs1) == len(STREAM_L)
    for Lv, (clip_order, tracks) in zip(STREAM_L, outs0):
        assert clip_order == clip_schedule(Lv, CFG.n_frames_test, 1)
        ref = replay([(s, e, l, fake_result(s, e)) for s, e, l in clip_schedule(Lv, CFG.n_frames_test, 1)])
        for (c, m), (cr, mr) in zip(tracks, ref):
            assert torch.allclose(c, cr) and torch.equal(m, mr)


def test_round_robin_stream_eight_ranks_root_only(tmp_path):
    """The bench's N = 8 form with a stand-in model: videos of 4, 1 and 18 chunks on eight ranks (idle ranks in a round, a whole video
    without work for most ranks, three rounds of the long one): rank 0 sees every clip once, in global order, and replays to the
    single-process tracker result; the other ranks hand out None."""
    lengths = (13, 5, 70)
    got = _run_stream(tmp_path, True, world=8, lengths=lengths)
    assert all(o is None for r in range(1, 8) for o in got[r][0]) and all(len(got[r][0]) == len(lengths) for r in range(8))
    for Lv, (clip_order, tracks) in zip(lengths, got[0][0]):
        clips = clip_schedule(Lv, CFG.n_frames_test, 1)
        assert clip_order == clips
        ref = replay([(s, e, l, fake_result(s, e)) for s, e, l in clips])
        assert len(tracks) == len(ref)
        for (c, m), (cr, mr) in zip(tracks, ref):
            assert torch.allclose(c, cr) and torch.equal(m, mr)


class _Dist1:
    """world_size 1 stand-in for torch.distributed (the collectives copy)."""

    @staticmethod
    def all_gather(outs, t):
        outs[0].copy_(t)

    @staticmethod
    def gather(t, outs, dst=0):
        outs[0].copy_(t)


def test_producer_failure_stops_the_replay_workers():
    """A producer that raises mid-stream (or a caller that drops the generator) must not leave replay threads blocked in
    q.get() holding their mergers."""
    import threading
    import pytest
    import mdqe_cvpr2023_amd.meta_arch as MA

    class Boom(_FakeModel):
        def iter_clip_results(self, frames, clips, f0, trace=None, primed=False, on_frames_queued=None, side_streams=True, split_small=False):
            if frames.shape[0] == 7:
                raise ValueError("bad video")
            yield from super().iter_clip_results(frames, clips, f0, trace, primed, on_frames_queued)

    old = MA.ClipMerger
    MA.ClipMerger = _FakeMerger
    try:
        base = threading.active_count()

        def jobs(second_len):
            for Lv in (13, second_len):
                plan = sharding.chunk_plan(Lv, CFG.n_frames_test, 1, 40)
                yield ({0: torch.zeros(Lv, 3, HW[0] * 4, HW[1] * 4)}, plan)
        it = sharding.run_round_robin_stream(Boom([]), jobs(7), 0, 1, _Dist1, (HW[0] * 4, HW[1] * 4), root_only=True)
        with pytest.raises(ValueError, match="bad video"):
            list(it)
        assert threading.active_count() == base
        it = sharding.run_round_robin_stream(Boom([]), jobs(9), 0, 1, _Dist1, (HW[0] * 4, HW[1] * 4), root_only=True)
        first = next(it)
        assert first[0] == clip_schedule(13, CFG.n_frames_test, 1)
        it.close()                                       # the caller walks away after the first video
        assert threading.active_count() == base
    finally:
        MA.ClipMerger = old


def test_chunk_plan_properties_on_random_inputs():
    """chunk_plan over random video lengths, clip lengths, strides, world sizes and per-round chunk sizes: every clip exactly once and in
    global order; the recompute form's frame range covers its clips; the halo form partitions the frames, puts a clip into the chunk of
    its last frame, keeps a whole clip in every chunk (sizes >= T) and never reaches further left than T-1 frames."""
    rng = np.random.RandomState(11)
    for _ in range(400):
        L_, T, stride, world = int(rng.randint(1, 400)), int(rng.randint(1, 6)), int(rng.randint(1, 4)), int(rng.randint(1, 9))
        sizes = [int(v) for v in rng.randint(max(T, 1), 60, size=rng.randint(1, 4))]
        chunk = sizes if rng.rand() < 0.7 else sizes[0]
        clips = clip_schedule(L_, T, stride)
        plan = sharding.chunk_plan(L_, T, stride, chunk, world=world)
        assert [c for ch in plan for c in ch[0]] == clips
        assert all(f0 <= s and e <= f1 for cl, f0, f1 in plan for s, e, _ in cl)
        assert sorted(g for r in range(world) for g in sharding.owned_chunks(plan, world, r)) == list(range(len(plan)))
        if stride == 1 and L_ >= T:
            hp = sharding.chunk_plan(L_, T, stride, chunk, halo_exchange=True, world=world)
            assert [c for ch in hp for c in ch[0]] == clips
            assert hp[0][1] == 0 and hp[-1][2] == L_ and all(a[2] == b[1] for a, b in zip(hp[:-1], hp[1:]))
            for cl, f0, f1 in hp:
                assert f1 - f0 >= T or len(hp) == 1
                assert all(f0 <= e - 1 < f1 and s >= f0 - (T - 1) for s, e, _ in cl)


# ---- the halo exchange's ring on its own process group (gloo, 3 ranks, 2 rounds) ------------------------------------------------------
def halo_worker(rank, world, port, outdir, rounds=2, rest=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T1, N, C, Hm, Wm, M = 3, 5, 4, 2, 3, 2
    job = sharding._Job.__new__(sharding._Job)                 # only what _halo() reads
    job.world, job.rank, job.dist, job.T, job.device = world, rank, dist, T1 + 1, torch.device("cpu")
    job.plan = [([g], g, g + 1) for g in range(rounds * world)]      # `rounds` rounds of `world` chunks (only emptiness is read)
    if rest:
        job.plan[(rounds - 1) * world] = ([], 0, 0)            # rank 0 rests in the last round (rest_root_sizes)
    job.halo_dims, job.halos, job.local_halo = (N, C, Hm, Wm, M), {}, False
    job.halo_pg = sharding.halo_group(dist, world)
    assert job.halo_pg is not None and job.halo_pg is not dist.group.WORLD        # its own communicator
    assert sharding.halo_group(dist, world) is job.halo_pg                          # created once
    got = {}
    for q in range(rounds):
        g = q * world + rank
        if not job.plan[g][0]:
            t = torch.ones(1)
            dist.all_reduce(t)                                 # (the resting rank still takes part in the round's default-group collective)
            continue
        h = job._halo(q, g)
        # a collective on the DEFAULT group between the exchanges, issued on different sides of it by even and odd ranks: harmless now
        t = torch.ones(1)
        if rank % 2 == 0:
            dist.all_reduce(t)
        enc = torch.full((T1, N, C), float(g)); mf = torch.full((T1, Hm, Wm, M), float(g) + 0.5)
        h.on_tail(enc, mf)
        if rank % 2 == 1:
            dist.all_reduce(t)
        if g > 0:                                              # chunk g reads the tail of chunk g - 1
            e, m = h.head()
            got[g] = (float(e.mean()), float(m.mean()), tuple(e.shape), tuple(m.shape))
    torch.save(got, os.path.join(outdir, f"halo{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_ring_on_its_own_process_group(tmp_path):
    """sharding._Halo / halo_group over gloo with three ranks and two rounds: chunk g receives exactly the tail of chunk g - 1 (rank 0's
    comes from the LAST rank of the previous round), on a process group of its own -- a default-group collective issued before the exchange
    on some ranks and after it on others does not disturb it."""
    world = 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=halo_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    seen = {}
    for r in range(world):
        seen.update(torch.load(os.path.join(str(tmp_path), f"halo{r}.pt"), weights_only=False))
    assert sorted(seen) == [1, 2, 3, 4, 5]
    for g, (e, m, es, ms_) in seen.items():
        assert e == float(g - 1) and m == float(g - 1) + 0.5 and es == (3, 5, 4) and ms_ == (3, 2, 3, 2), (g, e, m)


def test_halo_ring_skips_the_resting_root(tmp_path):
    """The ring over the NON-EMPTY chunks of a plan (sharding._Job._halo) with rank 0 resting in the last of three rounds, four ranks: the
    last rank of round 1 sends its tail to rank 1, which posted the receive in round 1 beside its own chunk's and reads it a round later;
    rank 0 posts nothing for a round it does not compute."""
    world, rounds = 4, 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=halo_worker, args=(r, world, port, str(tmp_path), rounds, True)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    seen = {}
    for r in range(world):
        seen.update(torch.load(os.path.join(str(tmp_path), f"halo{r}.pt"), weights_only=False))
    g_rest = (rounds - 1) * world
    assert sorted(seen) == [g for g in range(1, rounds * world) if g != g_rest]
    for g, (e, m, es, ms_) in seen.items():
        src = g - 2 if g == g_rest + 1 else g - 1                 # chunk g_rest + 1 reads the tail of the chunk BEFORE the empty one
        assert e == float(src) and m == float(src) + 0.5, (g, e, m)


def test_root_load_expansion_is_the_round_rank_0_of_an_n_rank_job_would_gather():
    """sharding.expand_root_load (bench.py MDQE_BENCH_ROOT_LOAD / the line's `root_load`): rank 0's own chunk of round q of an 8-rank plan is
    expanded to the 8 chunks of that round -- every clip of the plan exactly once, in global order, with the plan's frame ranges and `last`
    flags; foreign clips carry rank 0's instances (the short clips at the end of the video trimmed in time); replayed by the tracker core,
    the expanded rounds consume the whole virtual video."""
    T, W, per = CFG.n_frames_test, 8, 12
    sizes = sharding.round_sizes(per, T, smallest=3)
    assert len(sizes) >= 2 and sum(sizes) == per
    Lv = per * W
    plan = sharding.chunk_plan(Lv, T, 1, sizes, world=W)
    rounds = -(-len(plan) // W)
    seen = []
    for q in range(rounds):
        g0 = q * W
        own = [(s, e, l, fake_result(s, e)) for s, e, l in plan[g0][0]]
        out = sharding.expand_root_load(own, q, plan, W, T)
        want = [c for g in range(g0, min(g0 + W, len(plan))) for c in plan[g][0]]
        assert [(s, e, l) for s, e, l, _ in out] == want
        for (s, e, l, r), k in zip(out, range(len(out))):
            assert r["pred_masks"].shape[1] == e - s and r["pred_masks"].shape[0] == len(r["scores"])
        assert all(a[3] is b[3] for a, b in zip(out, own))            # rank 0's own clips are handed on untouched
        seen += out
    assert [(s, e, l) for s, e, l, _ in seen] == clip_schedule(Lv, T, 1)
    outs = replay(seen)
    assert sum(m.shape[1] for _, m in outs) == Lv                      # every frame of the virtual video came out of a window flush
    import pytest
    with pytest.raises(RuntimeError):
        sharding.expand_root_load([(1, 4, False, fake_result(1, 4))], 0, plan, W, T)


def test_resting_root_plan_keeps_every_clip_once_and_rank_0_free_in_the_last_round():
    """sharding.rest_root_sizes: the last round's frames go to ranks 1 .. N-1; rank 0's chunk of that round is an EMPTY placeholder (chunk g
    still belongs to rank g % world), every clip is owned exactly once and in order, two ranks / one round are left alone."""
    T = 4
    for per, W in ((120, 8), (120, 4), (60, 3)):
        base = sharding.round_sizes(per, T, ratio=0.6)
        sizes = sharding.rest_root_sizes(base, W)
        assert isinstance(sizes[-1], list) and sizes[-1][0] == 0 and sum(sizes[-1]) == base[-1] * W and max(sizes[-1]) - min(sizes[-1][1:]) <= 1
        Lv = per * W
        plan = sharding.chunk_plan(Lv, T, 1, sizes, world=W)
        assert [c for ch in plan for c in ch[0]] == clip_schedule(Lv, T, 1)
        rounds = -(-len(plan) // W)
        assert rounds == len(sizes)
        g_rest = (rounds - 1) * W
        assert plan[g_rest][0] == [] and all(plan[g][0] for g in range(len(plan)) if g != g_rest)
        assert sharding.owned_chunks(plan, W, 0)[-1] == g_rest
        for g, (cl, f0, f1) in enumerate(plan):
            assert all(f0 <= c[0] and c[1] <= f1 for c in cl)
    assert sharding.rest_root_sizes([69, 34, 17], 2) == [69, 34, 17] and sharding.rest_root_sizes([120], 8) == [120]
    import pytest
    with pytest.raises(ValueError):
        sharding.chunk_plan(100, T, 1, [10, [0, 5, 5]], world=2)                      # one entry per rank
    with pytest.raises(ValueError):
        sharding.chunk_plan(100, T, 1, [[2, 10]], halo_exchange=True, world=2)       # frames but no whole clip
    # the halo-exchange form with a resting root: the frames are partitioned, the empty chunk keeps its slot, every clip once and in order
    for per, W in ((120, 8), (60, 3)):
        sizes = sharding.rest_root_sizes(sharding.round_sizes(per, T, ratio=0.6), W)
        plan = sharding.chunk_plan(per * W, T, 1, sizes, halo_exchange=True, world=W)
        g_rest = (len(sizes) - 1) * W
        assert plan[g_rest][0] == [] and plan[g_rest][1] == plan[g_rest][2]
        assert [c for ch in plan for c in ch[0]] == clip_schedule(per * W, T, 1)
        assert all(a[2] == b[1] for a, b in zip(plan[:-1], plan[1:])) and plan[0][1] == 0 and plan[-1][2] == per * W


def test_root_load_expansion_with_a_resting_root_repeats_its_last_own_round():
    T, W, per = CFG.n_frames_test, 4, 24
    sizes = sharding.rest_root_sizes(sharding.round_sizes(per, T, ratio=0.5, smallest=3), W)
    assert isinstance(sizes[-1], list)
    Lv = per * W
    plan = sharding.chunk_plan(Lv, T, 1, sizes, world=W)
    rounds = -(-len(plan) // W)
    seen, template = [], None
    for q in range(rounds):
        own = [(s, e, l, fake_result(s, e)) for s, e, l in plan[q * W][0]]
        out = sharding.expand_root_load(own, q, plan, W, T, template=template)
        if own:
            template = out[:len(own)]
        assert [(s, e, l) for s, e, l, _ in out] == [c for g in range(q * W, min(q * W + W, len(plan))) for c in plan[g][0]]
        seen += out
    assert plan[(rounds - 1) * W][0] == [] and [(s, e, l) for s, e, l, _ in seen] == clip_schedule(Lv, T, 1)
    assert sum(m.shape[1] for _, m in replay(seen)) == Lv


def rest_stream_worker(rank, world, port, outdir):
    import torch.distributed as dist
    import mdqe_cvpr2023_amd.meta_arch as MA
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    MA.ClipMerger = _FakeMerger
    model = _FakeModel([])
    Lv = 72
    sizes = sharding.rest_root_sizes([10, 5, 3], world)
    plan = sharding.chunk_plan(Lv, CFG.n_frames_test, 1, sizes, world=world)
    frames = {g: torch.zeros(plan[g][2] - plan[g][1], 3, HW[0] * 4, HW[1] * 4) for g in sharding.owned_chunks(plan, world, rank) if plan[g][0]}
    out = sharding.run_round_robin(model, frames, plan, rank, world, dist, (HW[0] * 4, HW[1] * 4), root_only=True,
                                   like=torch.zeros(0, 3, HW[0] * 4, HW[1] * 4))
    torch.save((out, plan), os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_round_robin_with_a_resting_root_four_ranks(tmp_path):
    """Four gloo ranks, rank 0 without a chunk in the last round: it still takes part in that round's gather (zero clips), sees every clip
    once in global order and replays to the single-process tracker result."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=rest_stream_worker, args=(r, 4, port, str(tmp_path))) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(4)]
    out0, plan = got[0]
    assert all(g[0] is None for g in got[1:])
    assert plan[8][0] == [] and len(plan) == 12
    clip_order, tracks = out0
    clips = clip_schedule(72, CFG.n_frames_test, 1)
    assert clip_order == clips
    ref = replay([(s, e, l, fake_result(s, e)) for s, e, l in clips])
    assert len(tracks) == len(ref)
    for (c, m), (cr, mr) in zip(tracks, ref):
        assert torch.allclose(c, cr) and torch.equal(m, mr)


# ---- measured plan: rank 0's share from gathered per-rank times, not from constants --------------------------------
def test_tune_root_share_balances_a_linear_cost_model():
    """Pure function: with per-frame costs c_0 (root, its replay included) and c_o, two iterations land on the share that equalises the
    ranks' busy times, from any start; equal costs keep the share; clamped."""
    world, per = 8, 103.0
    for c0, co in ((1.30, 1.0), (1.08, 1.0), (1.0, 1.0), (0.9, 1.0)):
        share = 1.0
        for _ in range(3):
            f0 = per * share
            fo = (per * world - f0) / (world - 1)
            share = sharding.tune_root_share([c0 * f0] + [co * fo] * (world - 1), [f0] + [fo] * (world - 1), share)
        f0 = per * share
        fo = (per * world - f0) / (world - 1)
        assert abs(c0 * f0 - co * fo) < 0.01 * co * fo, (c0, share)
    assert sharding.tune_root_share([400.0, 100.0], [100, 100], 1.0) == 0.5            # clamped below
    assert sharding.tune_root_share([0.0, 100.0], [0, 100], 0.93) == 0.93                # nothing measured on rank 0: unchanged


class _SlowRootModel(_FakeModel):
    """The stand-in with a per-frame cost: `ms_per_frame` of sleep per clip it hands out (a chunk of n frames holds ~n stride-1 clips);
    rank 0 is given a larger one -- its replay.  The cost falls where the real model's does: while the clip results are consumed."""
    def __init__(self, log, ms_per_frame):
        super().__init__(log)
        self.ms = ms_per_frame

    def iter_clip_results(self, frames, clips, f0, **kw):
        import time
        for item in super().iter_clip_results(frames, clips, f0, **kw):
            if item is not None:
                time.sleep(self.ms * 1e-3)
            yield item


def tune_worker(rank, world, port, outdir):
    import torch.distributed as dist
    import mdqe_cvpr2023_amd.meta_arch as MA
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    MA.ClipMerger = _FakeMerger
    torch.set_num_threads(1)
    model = _SlowRootModel([], 30.0 if rank == 0 else 20.0)            # rank 0 costs 1.5x per frame (sleeps long enough to dominate a loaded host)
    Lv, base, share, hist = 4 * 60, [30, 20, 10], 1.0, []
    like = torch.zeros(0, 3, HW[0] * 4, HW[1] * 4)
    out = None
    for it in range(3):
        sizes = sharding.rest_root_sizes(base, world, share=share)
        plan = sharding.chunk_plan(Lv, CFG.n_frames_test, 1, sizes, world=world)
        frames = {g: torch.zeros(plan[g][2] - plan[g][1], 3, HW[0] * 4, HW[1] * 4) for g in sharding.owned_chunks(plan, world, rank) if plan[g][0]}
        stats = []
        for _ in range(2):
            out = sharding.run_round_robin(model, frames, plan, rank, world, dist, (HW[0] * 4, HW[1] * 4), root_only=True, like=like, stats=stats)
        mine = sum(plan[g][2] - plan[g][1] for g in sharding.owned_chunks(plan, world, rank))
        share, info = sharding.measured_root_share(stats, mine, share, rank, world, dist)
        hist.append(info)
    torch.save((out, hist), os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_measured_plan_converges_with_a_slowed_root_four_ranks(tmp_path):
    """gloo world 4, rank 0 1.5x slower per frame: after warm videos every rank gathers the per-rank busy times and derives the same
    smaller share for rank 0; the measurement on the re-dealt plan is balanced (busy times within 25 % on a loaded host, 15 % on an idle
    one), the result stays the single-process one."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=tune_worker, args=(r, 4, port, str(tmp_path))) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(4)]
    hists = [g[1] for g in got]
    assert all(h == hists[0] for h in hists)                             # every rank derived the same plan from the same numbers
    h = hists[0]
    assert h[0]["share_before"] == 1.0 and h[0]["busy_ms"][0] > 1.15 * max(h[0]["busy_ms"][1:])     # measured: the root is the slow rank (it already rests in the last round)
    assert 0.5 < h[0]["share"] < 0.95 and h[1]["frames"][0] < h[0]["frames"][0]
    b = h[2]["busy_ms"]
    assert max(b) / min(b) < 1.25, h                                       # balanced on the measured plan (first measurement: > 1.15 the other way round)
    clip_order, tracks = got[0][0]
    clips = clip_schedule(240, CFG.n_frames_test, 1)
    assert clip_order == clips and all(g[0] is None for g in got[1:])


def test_resting_plan_falls_back_inside_the_library_when_a_halo_chunk_would_hold_no_clip():
    """the per-rank halo-exchange deal can leave a chunk with frames but no whole clip (ValueError) -- only bench.py caught it.
    sharding.chunk_plan_resting keeps the per-rank deal where it is valid and falls back to the uniform per-round sizes where it is not;
    a video longer than its sizes were planned for does not rest rank 0 in every overflow round."""
    T = CFG.n_frames_test
    plan, used = sharding.chunk_plan_resting(8 * 60, T, 1, [30, 20, 10], 8, halo_exchange=True)
    assert isinstance(used[-1], list) and used[-1][0] == 0 and sum(len(p[0]) for p in plan) == len(clip_schedule(480, T, 1))
    # tiny last round: rank 0's share leaves the others chunks shorter than a clip -> uniform sizes, every clip still exactly once
    plan2, used2 = sharding.chunk_plan_resting(4 * 14, T, 1, [8, 4, 2], 4, halo_exchange=True, share=0.5)
    assert sum(len(p[0]) for p in plan2) == len(clip_schedule(56, T, 1))
    # overflow rounds (video longer than planned): the resting entry is not repeated for ever -- rank 0 owns frames again behind it
    sizes = sharding.rest_root_sizes([10, 5], 4)
    planL = sharding.chunk_plan(4 * 15 + 40, T, 1, sizes, world=4)
    own0 = [g for g in sharding.owned_chunks(planL, 4, 0) if planL[g][0]]
    assert len(planL) > 8 and any(g >= 8 for g in own0), (len(planL), own0)


def test_overflow_rounds_of_a_halo_exchange_plan_keep_the_root_resting():
    """Round 6 (found by the 4-rank bench rehearsal): a rank that rests in round q cannot receive the tail its chunk of round q + 1 would
    start from, so in the halo-exchange form the resting entry repeats in overflow rounds (the recompute form deals them evenly)."""
    T = CFG.n_frames_test
    sizes = sharding.rest_root_sizes([7, 7], 4, halo_exchange=True)
    plan = sharding.chunk_plan(60, T, 1, sizes, halo_exchange=True, world=4)          # 56 planned frames, 4 in overflow
    assert len(plan) > 8 and all(not plan[g][0] for g in sharding.owned_chunks(plan, 4, 0) if g >= 4)
    assert sum(len(p[0]) for p in plan) == len(clip_schedule(60, T, 1))
    plan_r = sharding.chunk_plan(60, T, 1, sharding.rest_root_sizes([7, 7], 4), world=4)
    assert any(plan_r[g][0] for g in sharding.owned_chunks(plan_r, 4, 0) if g >= 8)   # recompute form: rank 0 works again behind its rest
