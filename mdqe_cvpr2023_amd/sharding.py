"""Multi-GPU: one long video sharded across ranks.

Frames are independent through the per-frame stages and clips are independent through the decoder and
inference_clip, so rank r owns the clips that START in its contiguous frame range and holds those
frames plus a (T-1)-frame halo; there is no collective on that path.  The single exchange step is one
variable-length all-gather (RCCL over xGMI; `gloo` in the CPU tests) of the per-clip tracker inputs,
padded to the global maximum instance count; every rank then replays the tracker in global clip
order, which is bit-identical to the single-GPU schedule.
"""
import contextlib
import datetime
import os
import time

import torch

# Every process group this module creates gets a finite timeout: a rank that dies before a collective must not leave the others waiting
# for the driver's kill (bench.py gives the default group the same one).
COLLECTIVE_TIMEOUT_S = float(os.environ.get("MDQE_COLLECTIVE_TIMEOUT_S", "120"))

# the decoder's instance-chain side stream inside a sharded chunk (round 3: off -- with HIP's default 4 hardware queues it landed behind the
# frame stream's GEMMs; tools A/B with GPU_MAX_HW_QUEUES=8)
SHARD_SIDE_STREAMS = os.environ.get("MDQE_SHARD_SIDE_STREAMS", "0") == "1"

# MDQE_SHARD_SPLIT_PASS=1: a chunk of a round behind the first that fits one frame pass is run as two half passes (meta_arch.MDQE.pass_bounds
# split_small) -- the previous round's clip work, and with it that round's gather and the start of its replay on rank 0, trails the pass
# queued behind it.  Measured in the N = 8 / N = 4 root-load rehearsal: 179.2 against 177.3 ms, 169.6 against 166.5 ms per step -- no gain, so
# off by default (profiles/r05_ab_split_pass.txt)
SHARD_SPLIT_PASS = os.environ.get("MDQE_SHARD_SPLIT_PASS", "0") == "1"
HALO_LOCAL = os.environ.get("MDQE_HALO_LOCAL", "0") == "1"

FIELDS = ("scores", "pred_classes", "cls_probs", "query_embeds", "pred_masks")


def root_share(halo_exchange=False, world=8):
    """Rank 0's share of a round's chunk size in the rounds it computes in (rest_root_sizes; MDQE_SHARD_ROOT_SHARE overrides).  Measured in
    the root-load rehearsal (profiles/r05_ab_root_share.txt): with the halo's recompute the OTHER ranks' delivery of the last round is what
    rank 0 waits for at N = 4 and 8 and a smaller root chunk only slows them (1.0); with the halo exchange they are 5 % faster, rank 0's
    own load decides and 0.93 balances the two (N = 8: 164.0 -> 153.2 ms per step)."""
    v = os.environ.get("MDQE_SHARD_ROOT_SHARE", "")
    return float(v) if v else (0.93 if halo_exchange and world >= 3 else 1.0)     # (two ranks: 0.97 left rank 1 the slower one, 153.0 / 146.1 ms)


def owned_range(L, world, rank):
    """Frames whose clips this rank owns: [a, b)."""
    per = (L + world - 1) // world
    return min(L, rank * per), min(L, (rank + 1) * per)


def frame_range(L, world, rank, T, stride=1):
    """Frames this rank must hold: owned range + halo so every owned clip is complete."""
    a, b = owned_range(L, world, rank)
    return a, min(L, b + T - 1)


def owned_clips(clips, L, world, rank):
    a, b = owned_range(L, world, rank)
    return [c for c in clips if a <= c[0] < b]


def pack(results, T, device, proto_shapes):
    """list of (start,end,last,res) -> (meta [n_clips,4] int64, vec [n_inst, 2+K+C] fp32, masks [n_inst, T, h, w] fp32): the
    instances of all clips back to back (no padding per clip).  vec columns = score | class | cls_probs | query_embeds -- the layout
    the engine already holds them in (`res["rows"]`, engine.inference_clips), so a round is packed by two concatenations."""
    n = len(results)
    K, C = proto_shapes["cls_probs"][0][0], proto_shapes["query_embeds"][0][0]
    hw = tuple(proto_shapes["pred_masks"][0][1:])
    meta = torch.tensor([[s, e, int(l), len(r["scores"])] for s, e, l, r in results], dtype=torch.int64, device=device).view(n, 4)
    rows, masks = [], []
    for s, e, _, r in results:
        k = len(r["scores"])
        if k == 0:
            continue
        v = r.get("rows")
        if v is None:
            v = torch.cat([r["scores"].reshape(k, 1).float(), r["pred_classes"].reshape(k, 1).float(), r["cls_probs"].reshape(k, K).float(),
                           r["query_embeds"].reshape(k, C).float()], 1)
        rows.append(v)
        m = r["pred_masks"]
        if m.shape[1] != T:                                # the short last clip is padded in time
            m = torch.cat([m, m.new_zeros((k, T - m.shape[1]) + hw)], 1)
        masks.append(m)
    vec = torch.cat(rows).to(device) if rows else torch.zeros(0, 2 + K + C, device=device)
    msk = torch.cat(masks).to(device) if masks else torch.zeros((0, T) + hw, device=device)
    return meta, vec.contiguous(), msk.contiguous()


def all_gather_clips(local, T, dist, world, device, proto_shapes, root=None, rank=0, timing=None):
    """Variable-length gather: sizes first (all-gather, 16 B per rank), then three payloads -- clip table, per-instance vectors,
    per-instance mask logits -- padded to the global maxima of clips and of INSTANCES per rank (not clips x max instances).
    root=None: all-gather, every rank gets the merged list.  root=r: payloads go to rank r only (`dist.gather`; the other
    ranks return None) -- 1/world of the all-gather's traffic into every non-root rank.
    timing: a dict that accumulates host seconds -- "pack" (the two concatenations), "wait" (the size all-gather + its host sync: the
    time this rank waits for the slowest rank of the round), "payload" (issuing the three gathers + reading the clip table and the
    per-instance vectors back, ONE device->host copy and sync for the whole round)."""
    t_in = time.perf_counter()
    meta_l, vec_l, msk_l = pack(local, T, device, proto_shapes)
    K = proto_shapes["cls_probs"][0][0]
    sizes = torch.tensor([len(local), vec_l.shape[0]], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    t_pack = time.perf_counter()
    dist.all_gather(all_sizes, sizes)
    sizes_h = torch.stack(all_sizes).cpu().tolist()               # one sync
    t_sizes = time.perf_counter()
    cmax = max(1, max(s_[0] for s_ in sizes_h))                   # never hand RCCL a zero-size buffer
    imax = max(1, max(s_[1] for s_ in sizes_h))
    mine = root is None or rank == root

    def exchange(part, rows):
        """-> [world, rows, ...] (one allocation; its `world` slices are the collective's output list)."""
        buf = part
        if part.shape[0] != rows:
            buf = torch.zeros((rows,) + tuple(part.shape[1:]), dtype=part.dtype, device=device)
            buf[:part.shape[0]] = part
        whole = torch.empty((world, rows) + tuple(part.shape[1:]), dtype=part.dtype, device=device) if mine else None
        outs = list(whole.unbind(0)) if mine else None
        if root is None:
            dist.all_gather(outs, buf)
        else:
            dist.gather(buf, outs, dst=root)
        return whole

    g_meta = exchange(meta_l, cmax)
    g_vec = exchange(vec_l, imax)
    g_msk = exchange(msk_l, imax)
    if timing is not None:
        timing["pack"] = timing.get("pack", 0.0) + t_pack - t_in
        timing["wait"] = timing.get("wait", 0.0) + t_sizes - t_pack
    if not mine:
        if timing is not None:
            timing["payload"] = timing.get("payload", 0.0) + time.perf_counter() - t_sizes
        return None
    ready = None
    if device.type == "cuda":
        ready = torch.cuda.Event()
        ready.record()                                 # the gathered payloads are complete once this event fires
        # the clip tables and the per-instance vectors of ALL ranks in one device->host transfer each and ONE sync (round 3: two
        # blocking copies per rank -- 2N host syncs per round on rank 0's main thread)
        meta_h = torch.empty(g_meta.shape, dtype=g_meta.dtype, pin_memory=True)
        vec_h = torch.empty(g_vec.shape, dtype=g_vec.dtype, pin_memory=True)
        meta_h.copy_(g_meta, non_blocking=True)
        vec_h.copy_(g_vec, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    else:
        meta_h, vec_h = g_meta, g_vec
    meta_l_, host_all = meta_h.tolist(), vec_h.numpy()
    merged = []
    for r in range(world):
        nclips, ninst = sizes_h[r]
        if nclips == 0:
            continue
        vec = g_vec[r][:ninst]
        host = host_all[r]
        classes = vec[:, 1].long()
        msk = g_msk[r]
        o = 0
        for s, e, l, n in meta_l_[r][:nclips]:
            pm = msk[o:o + n]
            if e - s != T:
                pm = pm[:, :e - s].contiguous()        # the short last clip was padded in time
            h = host[o:o + n]
            merged.append((s, e, bool(l), {"scores": vec[o:o + n, 0], "pred_classes": classes[o:o + n], "cls_probs": vec[o:o + n, 2:2 + K],
                                           "query_embeds": vec[o:o + n, 2 + K:], "pred_masks": pm, "ready": ready,
                                           "host": {"scores": h[:, 0], "cls_probs": h[:, 2:2 + K], "query_embeds": h[:, 2 + K:]}}))
            o += n
    merged.sort(key=lambda c: c[0])
    if timing is not None:
        timing["payload"] = timing.get("payload", 0.0) + time.perf_counter() - t_sizes
    return merged


class ReplayThread:
    """Tracker replay off the main thread (rank 0 of the root-only schedule): the Hungarian matching, the numpy bookkeeping
    and the per-clip device->host sync of the tracker run while the main thread keeps queueing the next round's kernels
    (the GIL is released during the syncs and inside the HIP / numpy calls)."""

    def __init__(self, merger, device):
        import queue
        import threading
        self.merger, self.device, self.err = merger, device, None
        # the device a new thread starts on is 0, not the creator's: pin it ("cuda" without an index = the creator's current one)
        self.dev_index = None
        if device.type == "cuda":
            self.dev_index = device.index if device.index is not None else torch.cuda.current_device()
        self.q = queue.Queue()
        self.busy_s = 0.0                          # host seconds this worker spent replaying (tracker + window flushes), for `stats`
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        try:
            if self.dev_index is not None:
                torch.cuda.set_device(self.dev_index)
                side = getattr(self.merger, "side", None)
                if side is not None:               # this thread only ever feeds this merger: its tracker stream becomes the
                    torch.cuda.set_stream(side)    # thread's current stream once (a stream context per clip costs ~30 us)
                    self.merger.side_is_current = True
            with torch.no_grad():
                while True:
                    items = self.q.get()
                    if items is None:
                        return
                    t0 = time.perf_counter()
                    fm = getattr(self.merger, "feed_many", None)
                    if fm is not None:
                        fm(items)                  # runs of clips between window flushes: one native tracker call each
                    else:
                        for it in items:
                            self.merger.feed(*it)
                    self.busy_s += time.perf_counter() - t0
        except BaseException as e:                     # surfaced by finish()
            self.err = e

    def put(self, items):
        if self.err is not None:                   # a worker-side failure surfaces at the next hand-over, not only at finish()
            raise self.err
        self.q.put(items)

    def finish(self):
        self.q.put(None)
        self.t.join()
        if hasattr(self.merger, "side_is_current"):
            self.merger.side_is_current = False    # finish() runs on the caller's thread and stream
        if self.err is not None:
            raise self.err
        return self.merger.finish()

    def abort(self):
        """The producer failed: stop the worker, keep the producer's exception."""
        self.q.put(None)
        self.t.join()


def expand_root_load(merged, q, plan, vworld, T, template=None):
    """The N = `vworld` ROOT LOAD on one GPU (bench.py MDQE_BENCH_ROOT_LOAD): `merged` holds the clip results of chunk q*vworld -- rank
    0's own chunk of round q in the plan of a `vworld`-rank job -- and the return value is the round as rank 0 of that job would have
    gathered it: the clips of chunks q*vworld .. q*vworld+vworld-1 in global clip order, where a foreign chunk repeats rank 0's own
    results under its own frame indices (clip k of chunk g takes the instances of clip k of rank 0's chunk; a chunk at the end of the
    video with fewer / shorter clips takes the first ones, trimmed in time).  The tracker replay, the bank updates, the window flushes,
    the final-mask kernel and the device->host copies of the masks then carry the volume of the N-rank job while rank 0 computes its own
    chunks; what the rehearsal cannot show is the wire (the foreign payloads never cross xGMI) and the other ranks' pace."""
    g0 = q * vworld
    own = {s: (s, e, l, r) for s, e, l, r in merged}
    own_clips = plan[g0][0]
    if [c[0] for c in own_clips] != sorted(own):
        raise RuntimeError("expand_root_load: the gathered round is not rank 0's chunk %d of the plan" % g0)
    if not own_clips:
        # rank 0 rests in this round (rest_root_sizes): the foreign chunks repeat the results of its LAST own round (`template`)
        if not template:
            raise RuntimeError("expand_root_load: rank 0 has no chunk in round %d and no earlier round to repeat" % q)
        srcs = [r for _, _, _, r in template]
    else:
        srcs = [own[c[0]][3] for c in own_clips]
    out = []
    for g in range(g0, min(g0 + vworld, len(plan))):
        for k, (s, e, l) in enumerate(plan[g][0]):
            src = srcs[min(k, len(srcs) - 1)]
            if g == g0:
                out.append((s, e, l, src))
                continue
            r = dict(src)
            if e - s != src["pred_masks"].shape[1]:
                r["pred_masks"] = src["pred_masks"][:, :e - s].contiguous()
            r.pop("rows", None)
            out.append((s, e, bool(l), r))
    return out


# ------------------------------------------------------------------------------------------------
# Round-robin chunks: overlap the (inherently sequential) tracker replay with compute
# ------------------------------------------------------------------------------------------------
def round_sizes(frames_per_rank, T, ratio=0.5, smallest=None, max_rounds=3):
    """Chunk sizes (frames) of the rounds of one video, DECREASING: the tracker replay of round q (world x the clips of a chunk, on
    rank 0, sequential) runs under the compute of round q+1, so the only replay nobody hides is the LAST round's -- which should
    therefore be short -- while a replay stays hidden as long as the next round is not much shorter than it (ratio ~ replay time /
    compute time per frame x world, about 0.4 at N = 8).  120 frames per rank -> 69 / 34 / 17."""
    smallest = max(3 * T, 12) if smallest is None else smallest
    per = int(frames_per_rank)
    for k in range(max_rounds, 1, -1):
        s0 = per * (1 - ratio) / (1 - ratio ** k)
        if s0 * ratio ** (k - 1) >= smallest:
            sizes = [max(int(round(s0 * ratio ** i)), 1) for i in range(k)]
            sizes[0] += per - sum(sizes)
            return sizes
    return [per]


def chunk_plan(L, T, stride, chunk, halo_exchange=False, world=1):
    """Global chunks in clip order: [(clips, f0, f1)] with clips = those that START in the chunk's frames and [f0, f1) the frames
    they need (chunk + (T-1)-frame halo, which the owner computes again).  `chunk`: frames per chunk, or a list of per-ROUND chunk
    sizes (round q = the `world` chunks q*world .. q*world+world-1; the last entry repeats) -- see round_sizes.
    halo_exchange=True: no frame is computed twice.  A chunk holds exactly its own frames and owns the clips whose LAST frame falls
    there; its first clips start up to T-1 frames earlier, in the left neighbour's chunk, whose encoder tokens + mask features of
    those frames arrive by send/recv (_Halo).  A last chunk shorter than T frames is merged into its neighbour (every chunk needs
    a whole clip of its own)."""
    from .meta_arch import MDQE
    clips = MDQE.clip_schedule(L, T, stride)
    sizes = [chunk] if isinstance(chunk, int) else list(chunk)
    per_rank = any(isinstance(c, (list, tuple)) for c in sizes)       # a round as a list of per-RANK sizes (rest_root_sizes): zeros allowed
    w = max(world, 1)
    if per_rank and any(isinstance(c, (list, tuple)) and len(c) != w for c in sizes):
        raise ValueError("chunk_plan: per-rank chunk sizes need one entry per rank")

    def size_of(g):
        c = sizes[min(g // w, len(sizes) - 1)]
        if g // w >= len(sizes) and isinstance(c, (list, tuple)) and min(c) == 0 and not halo_exchange:
            # a video longer than the sizes were planned for: a RESTING entry (a rank with 0 frames) is not repeated -- the overflow
            # rounds deal the same round total evenly, so no rank idles in every remaining round.  (Not in the halo-exchange form: its
            # ring needs a rank to have run in round q to receive the tail its chunk of round q + 1 starts from, _Job._halo -- there
            # the resting entry repeats, as before.)
            return max(1, -(-sum(int(v) for v in c) // w))
        return int(c[g % w]) if isinstance(c, (list, tuple)) else int(c)
    if any((min(c) < 0 or max(c) < 1) if isinstance(c, (list, tuple)) else int(c) < 1 for c in sizes):
        raise ValueError("chunk_plan: chunk sizes must be positive")
    edges, g = [0], 0
    while edges[-1] < L:
        edges.append(min(L, edges[-1] + size_of(g)))
        g += 1
    if halo_exchange:
        if len(edges) > 2 and edges[-1] - edges[-2] < T:
            del edges[-2]
        plan = [([c for c in clips if a <= c[1] - 1 < b], a, b) for a, b in zip(edges[:-1], edges[1:])]
        if per_rank and any(b > a and not cl for cl, a, b in plan):
            raise ValueError("chunk_plan: a chunk of the halo-exchange form holds frames but no whole clip")
        return plan                                # (an EMPTY chunk, a == b, keeps its slot: chunk g belongs to rank g % world)
    plan = []
    for a, b in zip(edges[:-1], edges[1:]):
        cl = [c for c in clips if a <= c[0] < b]
        if cl:
            plan.append((cl, cl[0][0], max(c[1] for c in cl)))
        elif per_rank:
            plan.append(([], a, a))                # an EMPTY chunk keeps its slot: chunk g still belongs to rank g % world
    return plan


def chunk_plan_resting(L, T, stride, sizes, world, halo_exchange=False, share=None):
    """The plan of a job whose root rests (rest_root_sizes applied to the per-round `sizes`), with the library's own fallback: in the
    halo-exchange form a per-rank deal can leave a chunk with frames but no whole clip of its own (chunk_plan raises ValueError) -- the
    uniform per-round sizes are used then.  Returns (plan, the sizes the plan was made with)."""
    per_rank = rest_root_sizes(sizes, world, share=share, halo_exchange=halo_exchange)
    try:
        return chunk_plan(L, T, stride, per_rank, halo_exchange=halo_exchange, world=world), per_rank
    except ValueError:
        if not halo_exchange:
            raise
        return chunk_plan(L, T, stride, list(sizes), halo_exchange=True, world=world), list(sizes)


def rest_root_sizes(sizes, world, share=None, halo_exchange=False):
    """Per-round chunk sizes that take load off rank 0, the only rank that replays the tracker (and runs the window flushes, the final
    masks and their read-back) beside its own compute.
    (a) Rank 0 RESTS in the last round: that round's frames go to ranks 1 .. world-1 (its entry becomes a per-rank list [0, a, a, .., b]).
    Rank 0 is the only rank with work after the last gather -- the replay of the last round, the last window flushes, the video merge;
    with no chunk of its own in the last round it has caught up with the replay when the last gather arrives and runs the last round's
    updates on an idle GPU.  (world >= 3 and more than one round.)
    (b) `share` < 1 (default: root_share(halo_exchange)): in the rounds before, rank 0's chunk is `share` of the round's size and the other ranks split
    the rest -- the replay of N ranks' clips costs rank 0 a fixed slice of its GPU, which it gets back as fewer frames."""
    sizes = list(sizes)
    if world < 2 or any(isinstance(s_, (list, tuple)) for s_ in sizes):
        return sizes
    share = root_share(halo_exchange, world) if share is None else float(share)
    rest = world >= 3 and len(sizes) >= 2
    out = []
    for q, c in enumerate(sizes):
        c = int(c)
        if rest and q == len(sizes) - 1:
            root = 0
        elif share != 1.0:
            root = max(1, int(round(c * share)))
        else:
            out.append(c)
            continue
        base, extra = divmod(c * world - root, world - 1)
        out.append([root] + [base + (1 if r < extra else 0) for r in range(world - 1)])
    return out


def tune_root_share(busy_ms, frames, share, lo=0.5, hi=1.25):
    """Rank 0's chunk share from MEASURED times (the constants of root_share were tuned on a one-GPU stand-in and will
    not survive the first real node).  busy_ms[r]: rank r's own milliseconds per video that do not depend on waiting for the others --
    compute + pack, on rank 0 also feed + the replay that nothing hides; frames[r]: the frames it computed per video under the plan that
    was measured (made with `share`).  Per-frame cost c_r = busy / frames (rank 0's carries its replay); the frames are re-dealt so that
    c_0 f_0 = c_o f_o with the total kept: f_0 = c_o F / (c_0 (N - 1) + c_o), c_o = the slowest other rank's.  Returns the new share
    (old share x f_0 / old f_0), clamped.  Pure and deterministic: every rank that holds the same gathered numbers derives the same plan."""
    world = len(busy_ms)
    if world < 2 or frames[0] <= 0 or busy_ms[0] <= 0:
        return float(share)
    c0 = busy_ms[0] / frames[0]
    co = max(b / f for b, f in zip(busy_ms[1:], frames[1:]) if f > 0)
    if co <= 0:
        return float(share)
    total = float(sum(frames))
    f0 = co * total / (c0 * (world - 1) + co)
    return float(min(hi, max(lo, share * f0 / frames[0])))


def measured_root_share(stats, frames_mine, share, rank, world, dist):
    """All ranks: gather every rank's busy milliseconds per video (from the `stats` run_round_robin_stream filled over a few warm videos)
    and its frames per video, and derive the SAME new share on every rank (tune_root_share).  Returns (share, info for the bench line)."""
    n = max(len(stats), 1)
    busy = sum(v.get("compute", 0.0) + v.get("pack", 0.0) + ((v.get("feed", 0.0) + v.get("replay_exposed", 0.0)) if rank == 0 else 0.0) for v in stats) / n
    allr = [None] * world
    dist.all_gather_object(allr, (float(busy), int(frames_mine)))
    busy_ms, frames = [a[0] for a in allr], [a[1] for a in allr]
    new = tune_root_share(busy_ms, frames, share)
    return new, {"share_before": round(float(share), 4), "share": round(new, 4), "busy_ms": [round(b, 2) for b in busy_ms], "frames": frames}


def owned_chunks(plan, world, rank):
    """Chunk g goes to rank g % world: after round q the chunks q*world .. q*world+world-1 -- the NEXT ones in global clip
    order -- are complete, so their all-gather and tracker replay run while round q+1 computes."""
    return [g for g in range(len(plan)) if g % world == rank]


def run_round_robin(model, chunk_frames, plan, rank, world, dist, out_size, emit_masks=True, root_only=False, halo_exchange=False, like=None,
                    stats=None, vworld=None, as_rank=0, rest_until_ms=0.0):
    """chunk_frames: {g: device tensor of frames plan[g].f0 .. plan[g].f1} for the chunks this rank owns.
    Default: every rank all-gathers each round and replays the tracker (all ranks return the video result;
    emit_masks=False skips the mask production on ranks that only keep the tracker in step).
    root_only=True (bench.py): the rounds are gathered to rank 0 only, which replays the tracker on a worker thread while
    its main thread goes on with the next round; the other ranks only compute and send, and return None.
    `like`: any [.., h, w] tensor on the frames' device -- needed by a rank that owns NO chunk of this video (more ranks than chunks)."""
    return next(run_round_robin_stream(model, [(chunk_frames, plan, like) if like is not None else (chunk_frames, plan)], rank, world, dist, out_size,
                                       emit_masks=emit_masks, root_only=root_only, halo_exchange=halo_exchange, stats=stats, vworld=vworld,
                                       as_rank=as_rank, rest_until_ms=rest_until_ms))


_HALO_GROUPS = {}


def halo_group(dist, world):
    """A process group (its own RCCL communicator) for the halo exchange's send/recv.  The per-round gathers run on the default group;
    with the point-to-point traffic on a communicator of its own, the order in which a rank issues a halo exchange relative to a
    gather can no longer pair it with the wrong operation of a peer -- RCCL matches operations per communicator, in issue order --
    so the exchange does not depend on every rank reaching it at the same program point.  Created once per (backend, world), by all
    ranks together (the first halo-exchange job of a process: `new_group` is itself collective)."""
    if dist is None or not hasattr(dist, "new_group") or not dist.is_initialized():
        return None
    default = getattr(getattr(dist, "group", None), "WORLD", None)     # a re-initialised process group is a new object: no stale handle
    hit = _HALO_GROUPS.get(world)
    if hit is None or hit[0] is not default:
        try:
            pg = dist.new_group(ranks=list(range(world)), timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
        except TypeError:                              # (a stand-in `dist` of the CPU tests)
            pg = dist.new_group(ranks=list(range(world)))
        hit = _HALO_GROUPS[world] = (default, pg)
    return hit[1]


class _Halo:
    """The halo exchange of one (rank, round): this chunk's last T-1 frames go to the right neighbour as [encoder tokens | mask
    features] (7.5 MB per frame at 360p instead of 1.1 ms of per-frame work each), the left neighbour's arrive the same way.
    Every rank of a round issues ONE grouped send/recv (`batch_isend_irecv`: RCCL runs the pair concurrently, so the ring
    of sends cannot deadlock) when its last frame pass is queued -- the same program point on every rank, between the gathers
    of two rounds.  Rank 0's left neighbour is the LAST rank of the previous round: what rank 0 receives in round q it uses in
    round q+1 (`carry`).  gloo (the 1-GPU tests) moves host copies."""

    def __init__(self, dist, send_to, recv_from, dims, device, carry_src=None, group=None, carry_from=None, local=False):
        self.dist, self.send_to, self.recv_from, self.dims, self.device = dist, send_to, recv_from, dims, device
        self.group = group                         # the exchange's own communicator (halo_group); None = the default group
        self.tail_sent = False
        self.works, self.recv_buf, self.send_buf, self.carry_buf = [], None, None, None
        # recv_from: the owner of the chunk in front of this one when it runs in the SAME round (this chunk's head).  carry_from: the
        # owner of the chunk in front of this rank's NEXT chunk, when that one runs in this round -- the message is posted here, where
        # every rank of the round issues its grouped send/recv, and read a round later through `carry_src` (the _Halo that posted it)
        self.carry_from = carry_from
        self.carry_src = carry_src
        # local: the rehearsal forms on one GPU (HALO_LOCAL at world 1; the root-load rehearsal of an N-rank plan): this chunk's own
        # tail stands in for every message it would receive -- what the exchange costs WITHOUT the communicator
        self.local = bool(local)
        self.host = dist is not None and getattr(dist, "get_backend", lambda: "")() == "gloo"

    def on_tail(self, enc_tail, mf_tail):
        """Called on the frame stream once the chunk's last pass is queued."""
        self.tail_sent = True
        T1, N, C, Hm, Wm, M = self.dims
        ops = []
        if self.local:
            k = enc_tail.shape[0]
            flat = torch.cat([enc_tail.reshape(k, -1), mf_tail.reshape(k, -1)], 1).contiguous()
            if self.recv_from is not None:
                self.recv_buf = flat
            if self.carry_from is not None:
                self.carry_buf = flat
            self.local_ev = torch.cuda.Event() if flat.is_cuda else None
            if self.local_ev is not None:
                self.local_ev.record()
            return
        if self.send_to is not None:
            k = enc_tail.shape[0]
            flat = torch.cat([enc_tail.reshape(k, -1), mf_tail.reshape(k, -1)], 1).contiguous()
            self.send_buf = flat.cpu() if self.host else flat
            ops.append(self.dist.P2POp(self.dist.isend, self.send_buf, self.send_to, self.group))
        if self.recv_from is not None:
            self.recv_buf = torch.empty(T1, N * C + Hm * Wm * M, dtype=torch.float32, device="cpu" if self.host else self.device)
            ops.append(self.dist.P2POp(self.dist.irecv, self.recv_buf, self.recv_from, self.group))
        if self.carry_from is not None:
            self.carry_buf = torch.empty(T1, N * C + Hm * Wm * M, dtype=torch.float32, device="cpu" if self.host else self.device)
            ops.append(self.dist.P2POp(self.dist.irecv, self.carry_buf, self.carry_from, self.group))
        if ops:
            self.works = self.dist.batch_isend_irecv(ops)

    def received(self, carry=False):
        """The message this rank received in this round (complete on the current stream), or None: the head of this round's chunk, or
        (carry=True) the head of this rank's next chunk."""
        for w in self.works:
            w.wait()
        self.works = []
        if getattr(self, "local_ev", None) is not None:
            torch.cuda.current_stream().wait_event(self.local_ev)
        buf = self.carry_buf if carry else self.recv_buf
        if buf is None:
            return None
        if self.host:
            return buf.to(self.device)                 # (pageable host memory: a plain, blocking copy -- this is the gloo rehearsal path)
        # the buffer was allocated under the FRAME stream (on_tail) and is read on the caller's stream: tell the caching allocator,
        # or the block could be handed to a later frame-stream allocation while the reader's kernels are still queued
        buf.record_stream(torch.cuda.current_stream(buf.device))
        if self.send_buf is not None and self.send_buf.is_cuda:
            self.send_buf.record_stream(torch.cuda.current_stream(self.send_buf.device))
        return buf

    def head(self):
        """(encoder tokens [T-1, N, C], mask features [T-1, Hm, Wm, M]) of the T-1 frames before this chunk."""
        flat = self.carry_src.received(carry=True) if self.carry_src is not None else self.received()
        if flat is None:
            raise RuntimeError("halo exchange: no message from the left neighbour")
        T1, N, C, Hm, Wm, M = self.dims
        return flat[:, :N * C].reshape(T1, N, C).contiguous(), flat[:, N * C:].reshape(T1, Hm, Wm, M).contiguous()


class _Job:
    """One video of the round-robin schedule on this rank: its chunks, its merger and (root-only form) its replay thread."""

    def __init__(self, model, chunk_frames, plan, rank, world, out_size, emit_masks, root_only, like=None, dist=None,
                 halo_exchange=False, local_halo=False):
        from .meta_arch import ClipMerger
        cfg = model.cfg
        self.model, self.chunk_frames, self.plan, self.rank, self.world = model, chunk_frames, plan, rank, world
        self.T = cfg.n_frames_test
        any_fr = next(iter(chunk_frames.values()), like)
        if any_fr is None:
            raise ValueError("a rank that owns no chunk of a video must pass `like` (any [.., h, w] tensor on the frames' device)")
        self.device = any_fr.device if any_fr.is_cuda else torch.device(getattr(model, "device", any_fr.device))   # host frames: uploaded per chunk
        h, w = int(any_fr.shape[-2]), int(any_fr.shape[-1])
        geo = model.engine.geometry(h, w)
        ms = cfg.match_stride
        mask_hw = (geo.Hp // ms, geo.Wp // ms)
        self.proto = {"scores": ((), torch.float32), "pred_classes": ((), torch.int64), "cls_probs": ((cfg.num_classes,), torch.float32),
                      "query_embeds": ((cfg.hidden_dim,), torch.float32), "pred_masks": ((self.T,) + tuple(mask_hw), torch.float32)}
        self.dist, self.halo_exchange = dist, bool(halo_exchange)
        self.local_halo = bool(local_halo)         # the root-load rehearsal: no peer exists, a chunk's own tail stands in (_Halo.local)
        self.halo_carry = None                     # rank 0: the last rank's tail of the previous round
        self.halos = {}
        self.halo_dims = (geo.N, cfg.hidden_dim, mask_hw[0], mask_hw[1], cfg.mask_dim) if halo_exchange else None
        self.halo_pg = halo_group(dist, world) if halo_exchange and not local_halo else None
        self.merger = self.replay = None
        if not root_only or rank == 0:
            self.merger = ClipMerger(model, (h, w), out_size, mask_hw, n_frames=max(c[2] for c in plan), emit_masks=emit_masks)
            if getattr(model, "merge_on_cpu", None) is None and hasattr(self.merger, "merge_on_cpu"):
                # a video long enough to be sharded is the case MERGE_ON_CPU exists for (mdqe/mdqe.py:455-456): its windows' final
                # masks stream to the host as they are flushed, under the next rounds' compute, whatever the config's default says
                # (an explicit model.merge_on_cpu wins); at N = 8 the one-shot copy of 960 frames x 7 tracks would be 1.5 GB at the end
                self.merger.merge_on_cpu = True
            if root_only:
                self.replay = ReplayThread(self.merger, self.device)
        self.rounds = (len(plan) + world - 1) // world
        self.t0 = time.perf_counter()              # start of this video on this rank
        self.own_last = None                       # root-load rehearsal: rank 0's clip results of its last non-empty round
        self.tm = {"compute": 0.0, "pack": 0.0, "gather_wait": 0.0, "gather_payload": 0.0, "feed": 0.0, "replay_exposed": 0.0, "replay_busy": 0.0}
        self.replay_busy = 0.0

    def start(self, q):
        """Queue the per-frame work of this rank's chunk of round q (async); the returned generator yields its clip results."""
        g = q * self.world + self.rank
        if q >= self.rounds or g >= len(self.plan) or not self.plan[g][0]:      # (an empty chunk: this rank rests in round q)
            return None
        fr, h2d = self.chunk_frames[g], None
        if not fr.is_cuda and self.device.type == "cuda":          # a1's host->device copy of this chunk, chunked on the copy stream
            fr, h2d = self.model.upload_frames(fr)
        kw = {"h2d": h2d} if h2d else {}
        if self.halo_exchange:
            kw["halo"] = self._halo(q, g)
        # (no side streams inside the decoder here: the sharded schedule already runs RCCL's stream and the replay thread's tracker
        # stream beside the frame / clip / copy streams, and HIP has 4 hardware queues for all of them -- meta_arch.iter_clip_results)
        if SHARD_SPLIT_PASS and q > 0:
            kw["split_small"] = True                   # (rounds behind the first: their gather trails the NEXT round's first pass)
        gen = self.model.iter_clip_results(fr, self.plan[g][0], self.plan[g][1], primed=True, side_streams=SHARD_SIDE_STREAMS, **kw)
        next(gen)
        return gen

    def _halo(self, q, g):
        """The exchange of this rank's chunk g (round q) in the ring of the plan's NON-EMPTY chunks: its tail goes to the owner of the
        next one; its head comes from the owner of the one before -- in the same grouped send/recv when that chunk runs in this round,
        else from the message this rank posted a round earlier (`carry`: rank 0, whose left neighbour is the last rank of the round
        before; with a resting root, rank 1 in the last round)."""
        world, rank, plan = self.world, self.rank, self.plan

        def step(i, d):
            i += d
            while 0 <= i < len(plan) and not plan[i][0]:
                i += d
            return i if 0 <= i < len(plan) else None
        nx, pv = step(g, 1), step(g, -1)
        send_to = nx % world if nx is not None else None
        recv_from = carry_src = carry_from = None
        if pv is not None:
            if pv // world == q:
                recv_from = pv % world
            else:
                carry_src = self.halos.get(pv // world)
                if carry_src is None or carry_src.carry_from is None:
                    raise ValueError("halo exchange: chunk %d's left neighbour ran in round %d, where rank %d posted no receive for it"
                                     % (g, pv // world, rank))
        g2 = next((i for i in range(g + world, len(plan), world) if plan[i][0]), None)      # this rank's next non-empty chunk ...
        if g2 is not None:
            p2 = step(g2, -1)
            if p2 is not None and p2 // world == q and g2 // world > q:
                carry_from = p2 % world            # ... whose left neighbour runs in THIS round: its tail is received now
        local = self.local_halo or (HALO_LOCAL and world == 1)
        if world == 1 and self.dist is None and not local:
            send_to = recv_from = carry_from = None    # (with a backend, one rank sends to itself: the 1-GPU RCCL rehearsal)
        h = _Halo(self.dist, send_to, recv_from, (self.T - 1,) + self.halo_dims, self.device, carry_src=carry_src, group=self.halo_pg,
                  carry_from=carry_from, local=local)
        self.halos[q] = h
        self.halos.pop(q - 2, None)
        return h

    def feed(self, merged):
        if self.replay is not None:
            self.replay.put(merged)                # global clip order within the round: chunk q*world, q*world+1, ...
        elif self.merger is not None:
            fm = getattr(self.merger, "feed_many", None)
            if fm is not None:
                fm(merged)
            else:
                for item in merged:
                    self.merger.feed(*item)

    def finish(self):
        if self.replay is not None:
            out = self.replay.finish()
            self.replay_busy = self.replay.busy_s
            return out
        return self.merger.finish() if self.merger is not None else None

    def abort(self):
        """The producer failed (or the caller dropped the generator): stop the worker so that it does not stay blocked in
        q.get() holding the merger, the tracker bank and its pinned buffers."""
        if self.replay is not None:
            self.replay.abort()
            self.replay = None


def halo_recompute_frac(plan, L):
    """Share of the per-frame work that is done twice: frames held by the chunks of a plan / frames of the video - 1 (0 with the halo
    exchange, whose chunks partition the frames)."""
    return sum(f1 - f0 for _, f0, f1 in plan) / float(max(L, 1)) - 1.0


def run_round_robin_stream(model, jobs, rank, world, dist, out_size, emit_masks=True, root_only=False, halo_exchange=False, stats=None,
                           vworld=None, as_rank=0, rest_until_ms=0.0):
    """Videos as a stream through the round-robin schedule.  jobs: iterable of (chunk_frames, plan[, like]) as for
    run_round_robin (`like`: any [.., h, w] tensor on the device, for a rank that owns no chunk of a short video); yields each video's result in order (None on the ranks that do not replay).  Within a video the next round's per-frame
    work is queued before this round's clip work; ACROSS videos the first round of video k+1 is queued before the last round's
    clip work of video k, and video k's result is handed out only after that first round has been gathered -- so the replay of
    video k's last round (N x clips of a chunk on rank 0, which has no next round of its own to hide under) and its mask
    read-back run under video k+1's compute.  Every rank walks the same sequence of collectives.
    stats: a list; one dict of host milliseconds per video is appended on every rank -- `compute` (queueing a round's per-frame work +
    consuming its clip results: ends with the host sync behind the round's last clip kernel), `pack` / `gather_wait` / `gather_payload`
    (all_gather_clips: `gather_wait` is the wait for the slowest rank of a round), `feed` (handing the round to the tracker or its
    replay thread), `replay_exposed` (joining the replay + the video merge after the last gather: what no later round hides),
    `replay_busy` (the replay worker's busy time: tracker + window flushes of every round), `rounds`.
    vworld (one rank only): the root-load rehearsal -- `plan` is the plan of a `vworld`-rank job, this rank computes rank 0's chunks of it
    and every gathered round is expanded to the `vworld` chunks rank 0 of that job would replay (expand_root_load); with the halo
    exchange a chunk's own tail stands in for the neighbour's message (no peer exists: the wire is not rehearsed).
    rest_until_ms (rehearsal of a resting root only; bench.py supplies the other ranks' measured compute + pack): a round in which
    this rank has no chunk is gathered no earlier than that many milliseconds after the start of the video."""
    if vworld is not None and (world != 1 or rank != 0):
        raise ValueError("the root-load rehearsal runs on ONE rank")
    pworld = vworld if vworld is not None else world   # the world the chunks are dealt to
    # as_rank (rehearsal only): play rank `as_rank` of the vworld-rank job instead of rank 0 -- its chunks, no replay (a non-root rank of
    # the root-only schedule computes and sends): what the OTHER ranks' step costs when rank 0 rests in the last round
    prank = as_rank if vworld is not None else rank
    it = iter(jobs)
    touch = getattr(model, "touch_streams", None)
    if touch is not None:
        touch()                                        # the model's streams take their hardware queues before a job creates a communicator
    ws = getattr(model, "work_stream", contextlib.nullcontext)      # the model's high-priority stream (no context is held across a yield)

    def open_next():
        j = next(it, None)
        return None if j is None else _Job(model, j[0], j[1], prank, pworld, out_size, emit_masks, root_only,
                                           like=j[2] if len(j) > 2 else None, dist=dist, halo_exchange=halo_exchange,
                                           local_halo=halo_exchange and vworld is not None)

    def finish(j):
        t0 = time.perf_counter()
        with ws():
            out = j.finish()
        if stats is not None:
            j.tm["replay_exposed"] = time.perf_counter() - t0
            j.tm["replay_busy"] = j.replay_busy
            stats.append({k: 1e3 * v for k, v in j.tm.items() if k != "rounds"} | {"rounds": j.rounds})
        return out

    job = pending = nxt_job = None                 # pending: the previous video, all rounds fed, result not yet collected
    try:
        with ws():
            job = open_next()
            gen = job.start(0) if job is not None else None
        while job is not None:
            nxt_job = None
            for q in range(job.rounds):
                t_r = time.perf_counter()
                with ws():
                    if q + 1 < job.rounds:
                        nxt_gen = job.start(q + 1)     # the next round's per-frame work goes to the frame stream first ...
                    else:
                        nxt_job = open_next()          # ... or the first round of the next video
                        nxt_gen = nxt_job.start(0) if nxt_job is not None else None
                    local = [r for r in gen] if gen is not None else []     # ... and runs under this round's decoder + clip inference
                    t_c = time.perf_counter()
                    if vworld is not None and prank == 0 and rest_until_ms > 0 and not job.plan[q * vworld][0]:
                        # rehearsal of a resting root: the round's gather completes when the OTHER ranks deliver -- not before
                        # rest_until_ms after the start of the video (their measured compute + pack); the replay thread works on meanwhile
                        while (time.perf_counter() - job.t0) * 1e3 < rest_until_ms:
                            time.sleep(2e-4)
                    tg = {}
                    merged = all_gather_clips(local, job.T, dist, world, job.device, job.proto, root=0 if root_only else None, rank=rank, timing=tg)
                    if vworld is not None and prank == 0:
                        merged = expand_root_load(merged, q, job.plan, vworld, job.T, template=job.own_last)
                        if job.plan[q * vworld][0]:
                            job.own_last = merged[:len(job.plan[q * vworld][0])]
                    t_g = time.perf_counter()
                    job.feed(merged)
                    tm = job.tm
                    tm["compute"] += t_c - t_r
                    tm["pack"] += tg.get("pack", 0.0); tm["gather_wait"] += tg.get("wait", 0.0); tm["gather_payload"] += tg.get("payload", 0.0)
                    tm["feed"] += time.perf_counter() - t_g
                if q == 0 and pending is not None:     # the previous video's tail has had this whole round to finish
                    p, pending = pending, None
                    yield finish(p)
                gen = nxt_gen
            if pending is not None:                    # (a video without rounds cannot occur: a plan has at least one chunk)
                p, pending = pending, None
                yield finish(p)
            pending, job, nxt_job = job, nxt_job, None
        if pending is not None:
            p, pending = pending, None
            yield finish(p)
    finally:                                           # an exception above, or the caller dropped the generator (GeneratorExit)
        for j in (job, pending, nxt_job):
            if j is not None:
                j.abort()


def run_sharded(model, shard_frames, f0, L, rank, world, dist, out_size):
    """shard_frames: device tensor of this rank's frames, first one is global frame f0."""
    cfg = model.cfg
    T = cfg.n_frames_test
    h, w = int(shard_frames.shape[-2]), int(shard_frames.shape[-1])
    geo = model.engine.geometry(h, w)
    ms = cfg.match_stride
    mask_hw = (geo.Hp // ms, geo.Wp // ms)
    clips = model.clip_schedule(L, T, cfg.clip_stride)
    mine = owned_clips(clips, L, world, rank)
    local = list(model.iter_clip_results(shard_frames, mine, f0))
    proto = {"scores": ((), torch.float32), "pred_classes": ((), torch.int64), "cls_probs": ((cfg.num_classes,), torch.float32),
             "query_embeds": ((cfg.hidden_dim,), torch.float32), "pred_masks": ((T,) + tuple(mask_hw), torch.float32)}
    merged = all_gather_clips(local, T, dist, world, shard_frames.device, proto)
    return model.merge_clips(iter(merged), (h, w), out_size, mask_hw, n_frames=L)
