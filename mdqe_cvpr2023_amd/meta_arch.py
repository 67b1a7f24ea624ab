"""MDQE meta-architecture: the drop-in for the reference's `MDQE` (mdqe/mdqe.py:60-471) on the
eval-only VIS path.

    model = MDQE(cfg)                      # cfg: the reference's CfgNode, or an MDQEConfig
    model.load_state_dict(ckpt["model"])   # reference checkpoint names (prefix `detr.`)
    out = model([{"image": [uint8 [3,h,w], ...], "height": H, "width": W}])
    # -> {"image_size", "pred_scores", "pred_labels", "pred_masks": [BoolTensor[L,H,W] (CPU)]}

Same call contract and output dict as MDQE.forward -> inference_vis -> inference_video.  When
detectron2 is importable the class is registered in its META_ARCH_REGISTRY under the names "MDQE" (taking the slot over from the
reference's model if `mdqe` was imported first) and "MDQE_MI355X" -- `register_with_detectron2` at the end of this file.
COCO image sets (`DATASETS.TEST[0]` = coco*) take the single-image branch (`inference_image`); training is out of scope and raises.
"""
import contextlib
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import MDQEConfig, from_d2_cfg
from .engine import Engine
from .params import ALIASES, full_manifest, random_state
from .tracking import Clips, OverTracker


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer=False):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


# The pipeline's HIP streams, ONE set per device and process, shared by every MDQE instance (a process runs one model at a time, as the
# reference's harness does).  HIP deals streams onto its hardware queues in the order they are first used and a queue is served in order,
# so WHICH streams share a queue decides how well the pipeline overlaps (tools/stream_map_ab.py): a second model in the same process -- the
# bench's extra configurations, an evaluator that rebuilds the model -- used to create a second set of streams and land on another, usually
# worse, deal (Swin-L as the third model of a bench process: 135 against 150 frames/s in a process of its own).
_STREAMS = {}
HALO_EARLY_DECODE = os.environ.get("MDQE_HALO_EARLY_DECODE", "1") != "0"     # (A/B knob: 0 = round 4's behaviour)
HALO_OWN_STREAM = os.environ.get("MDQE_HALO_OWN_STREAM", "1") != "0"         # (A/B knob: 0 = the halo hand-over on the frame stream)


def _shared_stream(name):
    def key(self):
        d = self.device
        return ("cpu",) if d.type != "cuda" else (d.index if d.index is not None else torch.cuda.current_device(),)

    def get(self):
        return _STREAMS.get(key(self) + (name,))

    def put(self, stream):
        if stream is not None:
            _STREAMS[key(self) + (name,)] = stream
    return property(get, put)


class MDQE(nn.Module):
    _trk_stream = _shared_stream("tracker")
    _frame_stream = _shared_stream("frame")
    _copy_stream = _shared_stream("copy")
    _ahead_stream = _shared_stream("ahead")
    _work_stream = _shared_stream("work")

    def __init__(self, cfg, state_dict=None, backbone_fn=None, seed=0):
        super().__init__()
        self.cfg = cfg if isinstance(cfg, MDQEConfig) else from_d2_cfg(cfg)
        self.device = torch.device(self.cfg.device)
        self._backbone_fn = backbone_fn
        self._engine = None
        sd = state_dict if state_dict is not None else random_state(self.cfg, seed)
        man = full_manifest(self.cfg)
        if state_dict is not None:
            sd = dict(sd)
            for k in list(sd):                       # released checkpoints carry the decoder's shared modules under an alias too
                for a, b in ALIASES.items():
                    if k.startswith(a) and (b + k[len(a):]) not in sd:
                        sd[b + k[len(a):]] = sd[k]
            missing = [n for n in man if n not in sd]
            if missing:                              # a partial / mis-prefixed checkpoint must not become silent all-zero layers
                raise KeyError("MDQE(cfg, state_dict=...): %d parameters of the manifest are missing, e.g. %s" % (len(missing), missing[:5]))
        for name, shape in man.items():
            t = sd[name]
            _register(self, name, t.detach().clone().float(), buffer=name.endswith(("running_mean", "running_var")))
        self._extra = {k: v for k, v in sd.items() if k not in man}      # e.g. custom-backbone weights
        # frames per pass of the per-frame stages: 0 = by resolution (~300k encoder tokens per pass: at most 40 frames: 40 at 360p, 20 at 640p)
        self.frame_batch = int(os.environ.get("MDQE_FRAME_BATCH", "0"))
        # priority of the tracker's stream (0 normal, -1 high): its kernels are small and sit on the per-clip critical path of the replay
        self.trk_priority = int(os.environ.get("MDQE_TRK_PRIORITY", "0"))
        self.resize_on_device = False               # True: frames arrive at native size and get the mapper's ResizeShortestEdge here
        self.rle_output = False                     # True: forward() returns per-frame COCO RLEs ("pred_rles") instead of dense masks
        self.merge_on_cpu = None                    # None: cfg.merge_on_cpu (MODEL.MDQE.MERGE_ON_CPU); True / False override it
        # Final masks of a tracker window leave the device when the window is flushed (pinned host buffers, copied under the later
        # windows' compute) instead of in one pass + one 100-MB copy after the last window.  Independent of MERGE_ON_CPU, which in the
        # reference only picks the device the window results wait on (mdqe/mdqe.py:185-186,337,354-355) and never changes an output.
        self.early_masks = os.environ.get("MDQE_EARLY_MASKS", "1") != "0"
        self.decode_ahead = os.environ.get("MDQE_DECODE_AHEAD", "1") != "0"         # trailing short clips decoded beside the last full group
        # the decoder of a group starts as soon as ITS inputs of the group's last frame pass exist (queries + value projections); the
        # mask-feature head of that pass, which only inference_clip reads, runs beside the decoder's first layers (round 4)
        self.early_decode = os.environ.get("MDQE_EARLY_DECODE", "1") != "0"
        self.overlap_streams = os.environ.get("MDQE_OVERLAP_STREAMS", "1") != "0"   # frame stages on their own stream
        self.clip_priority = os.environ.get("MDQE_CLIP_PRIORITY", "1") != "0"       # per-clip stages on a high-priority stream
        self.stage_times = None
        self.taper_passes = os.environ.get("MDQE_TAPER_PASSES", "1") != "0"   # half-size first / last frame pass (pipeline fill / drain)
        self.taper_tail = int(os.environ.get("MDQE_TAPER_TAIL", "0"))          # frames of the last pass (0: half a pass)
        self.lookahead = max(1, int(os.environ.get("MDQE_LOOKAHEAD", "2")))   # groups whose frame passes are queued ahead of the one being decoded
        # clips per decoder batch the planner waits for before it fixes a group (0: the clips ONE frame pass completes, rounds 2-4): the
        # cache is per frame and the decoder reads it through index tables, so its batch is independent of the pass size
        self.dec_batch = int(os.environ.get("MDQE_DEC_BATCH", "0"))

    # ---- checkpoint contract ---------------------------------------------------------------------
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # accept (and drop) the reference checkpoint's aliased / non-eval keys
        for k in list(state_dict):
            kk = k[len(prefix):]
            # (+ the SwinV2 blocks' fixed buffers -- `relative_coords_table`, `relative_position_index`, swin_transformer_v2.py:120,133 -- and
            # the shift masks some exports carry: functions of the window size, rebuilt here)
            if any(kk.startswith(a) for a in ALIASES) or kk.endswith((".sampling_offsets", "lvl_spatial_scales",
                                                                      "query_relpos_grid", "num_batches_tracked",
                                                                      "relative_coords_table", "relative_position_index", "attn_mask")) \
                    or kk.startswith("criterion."):
                if not (kk.endswith(".sampling_offsets.weight") or kk.endswith(".sampling_offsets.bias")):
                    state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        self._engine = None

    def train(self, mode=True):
        if mode:
            raise RuntimeError("mdqe_cvpr2023_amd.MDQE implements the eval-only path (SURVEY.md §8); training is out of scope -- to train, "
                               "select the reference's model: MODEL.META_ARCHITECTURE MDQE_REFERENCE, or MDQE_MI355X_REGISTER=alias in the "
                               "environment so that 'MDQE' stays the reference's")
        return super().train(False)

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            sd = OrderedDict((k, v) for k, v in self.state_dict().items())
            sd.update(self._extra)
            with self._on_device():
                self._engine = Engine(self.cfg, sd, self.device, backbone_fn=self._backbone_fn)
        return self._engine

    @property
    def precision_map(self):
        """"" = exact fp32 everywhere (default); "reference" = the reference harness's own precision map on a GPU with its fp16 regions on the
        f16x3 split-precision kernels (engine.Engine.precision_map; env MDQE_PRECISION_MAP)."""
        return self.engine.precision_map

    @precision_map.setter
    def precision_map(self, v):
        if v not in ("", "reference", "autocast_f16"):
            raise ValueError("precision_map: '', 'reference' or 'autocast_f16'")
        self.engine.precision_map = v

    # ---- forward (mdqe/mdqe.py:194-242) ------------------------------------------------------------
    def _make_streams(self):
        """The pipeline's normal-priority streams, created TOGETHER and in a fixed order (copy, frame, tracker) the first time the model
        runs on a GPU.  HIP deals its streams onto 4 hardware queues in creation order and a queue is served in order, so WHICH streams
        share a queue is decided here -- and it matters: with one / three foreign normal-priority streams created before these, the
        same pipeline runs 5 % / 11 % slower (tools/stream_map_ab.py: 760 -> 719 / 679 frames/s; two: 754).  `MDQE_STREAM_PAD=k` shifts
        the deal by k queues for a process that has created streams of its own before the model's first call."""
        pad = int(os.environ.get("MDQE_STREAM_PAD", "0"))
        self._pad_streams = [torch.cuda.Stream(self.device) for _ in range(max(pad, 0))]
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        if self._frame_stream is None:
            self._frame_stream = torch.cuda.Stream(self.device)
        if self._trk_stream is None:
            self._trk_stream = torch.cuda.Stream(self.device, priority=self.trk_priority)
        # a stream gets its hardware queue when it is first USED, not when it is created (unused foreign streams change nothing,
        # tools/stream_pad_ab.sh).  The pipeline's natural first-use order (copy, frame, tracker, between the high-priority streams'
        # first uses) is the best deal measured; MDQE_STREAM_TOUCH=1 instead touches [pads,] copy, frame, tracker here, in that order --
        # a knob for a host process whose own streams have shifted the deal (tools/stream_touch_ab.sh: 731-735 against 737-745 frames/s)
        order = os.environ.get("MDQE_STREAM_ORDER", "")          # tools/stream_order_ab.sh: first-use order of ALL the pipeline's streams
        if order:
            if self._ahead_stream is None:
                self._ahead_stream = torch.cuda.Stream(self.device, priority=-1)
            from .engine import INST_STREAMS as pool
            inst = pool.setdefault(self._work_stream.cuda_stream, torch.cuda.Stream(self.device, priority=-1))
            table = {"w": self._work_stream, "i": inst, "a": self._ahead_stream, "c": self._copy_stream, "f": self._frame_stream, "t": self._trk_stream}
            for ch in order:
                st = table.get(ch) or torch.cuda.Stream(self.device, priority=-1 if ch == "X" else 0)
                if ch not in table:
                    self._pad_streams.append(st)
                with torch.cuda.stream(st):
                    torch.zeros(1, device=self.device)
        elif os.environ.get("MDQE_STREAM_TOUCH", "0") == "1" or getattr(self, "_touch_on_create", False):
            for st in self._pad_streams + [self._copy_stream, self._frame_stream, self._trk_stream]:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=self.device)

    def touch_streams(self):
        """Give the pipeline's normal-priority streams their hardware queues NOW, in a fixed order (copy, frame, tracker): a stream gets its
        queue when it is first used, and whatever creates streams of its own in between -- a second RCCL communicator (the halo exchange's),
        a host application -- shifts the deal.  The sharded schedule calls this before it creates anything else (one-rank halo-exchange
        rehearsal under 8 queues: 180.9 ms per step without, 153.5 with; profiles/r05_ab_halo_exchange_queues.txt).  Once per process."""
        if self.device.type != "cuda" or os.environ.get("MDQE_NO_STREAM_TOUCH") == "1":
            return
        key = ("touched", self.device.index if self.device.index is not None else torch.cuda.current_device())
        if _STREAMS.get(key):
            return
        _STREAMS[key] = True
        if self._work_stream is not None:
            return                                     # the streams are in use already: their queues are taken
        # the same order as MDQE_STREAM_TOUCH=1: copy, frame, tracker are touched inside `_make_streams`, right when the streams are
        # created and BEFORE the work stream's own first use (touching them behind it measured no better than not touching at all)
        self._touch_on_create = True
        try:
            with self._on_device(), self.work_stream():
                pass
        finally:
            self._touch_on_create = False

    @contextlib.contextmanager
    def work_stream(self):
        """The model's own HIGH-PRIORITY stream for the per-clip stages (decoder, inference_clip: hundreds of small kernels
        and three host syncs per pass).  Their blocks are then dispatched ahead of the queued blocks of the frame stream's
        large GEMMs instead of waiting for whole waves of them to drain: +4-7 % frames/s (tools/prio_ab.py; the opposite,
        a high-priority frame stream, costs 7 %).  Entering makes it wait for the caller's stream (the inputs), leaving makes
        the caller's stream wait for it (device-side outputs); re-entrant; a no-op on CPU or with MDQE_CLIP_PRIORITY=0."""
        if self.device.type != "cuda" or not self.clip_priority:
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        if self._work_stream is None:
            self._work_stream = torch.cuda.Stream(self.device, priority=-1)
            self._make_streams()
        ws = self._work_stream
        if cur == ws:
            yield
            return
        ws.wait_stream(cur)
        try:
            with torch.cuda.stream(ws):
                yield
        finally:
            cur.wait_stream(ws)

    PIN_POOL_GB = float(os.environ.get("MDQE_PIN_POOL_GB", "8"))       # pinned host memory the model keeps for mask read-back between calls

    def pinned_mask_buffer(self, shape):
        """A pinned uint8 host buffer for one track's final masks.  A video's result hands VIEWS of these buffers to the caller, so a
        buffer is free again when the caller has dropped that result: the pool keeps the buffers it has made (up to PIN_POOL_GB) and
        re-issues one as soon as nothing but the pool references its storage.  A fresh pinned allocation costs ~75 us per MB (7 tracks of a
        960-frame 360p video: 120 ms; tools/pinned_probe.py), and the framework's own host allocator recycles only every other video."""
        pool = self.__dict__.setdefault("_pin_pool", {})
        free = pool.setdefault(tuple(shape), [])
        # What is handed out is a VIEW (its own tensor object on the pooled storage): the storage's use count then stays above the pool's
        # own for as long as a merger, or a result the caller still holds, references the buffer -- the pool's tensor itself is never
        # given away (two Python references to ONE tensor object count once, and a second track of the same video would get the same
        # buffer: tests/test_fullsize_gpu.py caught exactly that).
        use_count, idle = self._pin_pool_probe()
        if use_count is not None:
            for t in free:
                if use_count(t.untyped_storage()._cdata) == idle:      # exactly what an unshared pooled buffer shows (calibrated, not assumed)
                    return t[:]         # (every frame row is rewritten by the windows' read-backs before the result is handed out)
        t = torch.empty(tuple(shape), dtype=torch.uint8, pin_memory=True)
        held = sum(b.numel() for bufs in pool.values() for b in bufs)
        if use_count is not None and held + t.numel() <= self.PIN_POOL_GB * 2 ** 30:
            free.append(t)
            return t[:]
        return t

    _PIN_PROBE = None

    @classmethod
    def _pin_pool_probe(cls):
        """(use-count accessor, the count an UNSHARED pooled buffer shows) or (None, None): pooling off.  The accessor is a private torch
        call whose baseline depends on the torch version's storage wrapper handling, so it is CALIBRATED once per process on a scratch
        tensor -- the count with only the owner alive, and that a view raises it by exactly one and dropping the view restores it -- instead
        of assumed; if the self-check fails the pool is disabled (fresh pinned buffers per call: slower, never wrong)."""
        if cls._PIN_PROBE is None:
            fn = getattr(torch._C, "_storage_Use_Count", None)
            probe = (None, None)
            if fn is not None:
                try:
                    t = torch.empty(16, dtype=torch.uint8)
                    idle = fn(t.untyped_storage()._cdata)
                    v = t[:]
                    shared = fn(t.untyped_storage()._cdata)
                    del v
                    if shared == idle + 1 and fn(t.untyped_storage()._cdata) == idle:
                        probe = (fn, idle)
                except Exception:
                    pass
            cls._PIN_PROBE = probe
        return cls._PIN_PROBE

    def _on_device(self):
        """Every ctypes launch goes to the CURRENT device's current stream: make the model's device current for the call."""
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    @torch.no_grad()
    def forward(self, batched_inputs):
        if len(batched_inputs) != 1:
            raise RuntimeError("MDQE eval takes exactly one video per call (mdqe/mdqe.py:292)")
        with self._on_device(), torch.autocast(device_type="cuda", enabled=False), self.work_stream():   # neutralise ambient autocast
            if self.cfg.is_coco:                                           # DATASETS.TEST[0] is a COCO set (mdqe/mdqe.py:213,233-236)
                return self.inference_image(batched_inputs)
            return self.inference_vis(batched_inputs)

    def alloc_cache(self, frames, geo, keep_enc=False):
        """An empty frame cache for `frames` frames: name -> [frames, ...] device buffer (engine.cache_shapes)."""
        return {k: torch.empty((int(frames),) + tuple(sh), dtype=torch.float32, device=self.device)
                for k, sh in self.engine.cache_shapes(geo, keep_enc).items()}

    def _frame_cache(self, frames, geo, ring=None, at=0, keep_enc=False, dec_ready=None):
        """Per-frame stages a1-a11 for a batch of frames: everything a clip needs, computed once, stored IN PLACE into rows
        [at, at+n) of the frame cache `ring` (name -> [capacity, ...] buffer; None: a cache of exactly these frames is allocated and
        returned) -- the 63 MB/frame decoder value cache by its GEMM, the mask features by the mask head's last product, the query
        coordinates / contents / embeddings by their kernels: no device-to-device copy of any result (round 5; rounds 2-4 copied the
        mask features and the small tensors into the ring and carried T-1 frames from ring to ring).  keep_enc: the encoder tokens go
        into the cache too (5.2 MB per frame; a sharded video ships the tokens of a chunk's last T-1 frames to the neighbour,
        sharding._Halo).
        Order (round 4): what the DECODER reads -- query selection / content / embeddings and the value projections -- comes first and
        `dec_ready` (an event) is recorded behind it; the mask-feature head, which only `inference_clip` reads, runs after.  The decoder
        of a group can then start while the mask head of its last frame pass is still running: at the end of a video that pass has
        nothing else to overlap with, and the decoder's chain of small dependent launches is latency-bound."""
        eng = self.engine
        n = frames.shape[0]
        ret = ring is None
        if ring is None:
            ring, at = self.alloc_cache(n, geo, keep_enc), 0
        feats = eng.backbone(frames, geo)
        enc = eng.encode(feats, geo)
        del feats
        eng.frame_queries(enc, geo, out=(ring["coords"][at:at + n], ring["content"][at:at + n], ring["emb"][at:at + n]))
        eng.dec_values(enc, geo, out=ring["vals"][at:at + n])
        if "enc" in ring:
            ring["enc"][at:at + n].copy_(enc)
        if dec_ready is not None:
            dec_ready.record(torch.cuda.current_stream(self.device))
        eng.mask_features(enc, geo, out=ring["mf"][at:at + n])
        return ring if ret else None

    def _cache_from_tokens(self, enc, mf, geo, ring, at):
        """Cache entries of frames whose encoder tokens and mask features were computed elsewhere (a neighbour rank's halo):
        only query selection / content sampling and the decoder value projections are redone here."""
        eng = self.engine
        n = enc.shape[0]
        eng.frame_queries(enc, geo, out=(ring["coords"][at:at + n], ring["content"][at:at + n], ring["emb"][at:at + n]))
        eng.dec_values(enc, geo, out=ring["vals"][at:at + n])
        ring["mf"][at:at + n].copy_(mf)
        if "enc" in ring:
            ring["enc"][at:at + n].copy_(enc)

    @staticmethod
    def pass_bounds(n_frames, fbatch, taper=True, tail=0, split_small=False):
        """End frames of the passes of the per-frame stages over a chunk of `n_frames`: uniform passes of `fbatch` frames, or
        (taper) a half-size first pass -- the clip stream starts after half a pass instead of a whole one -- and a last pass of
        at most `tail` frames (0: half a pass) -- the tail that runs with an idle frame stream (last decoder batch, tracker,
        mask read-back) is shorter.  Measured: 20/40/40/20 at 360p +0.2-1.1 %; shorter tails lose to the extra pass."""
        bounds = list(range(fbatch, n_frames, fbatch)) + [n_frames]
        if split_small and taper and 16 <= n_frames <= fbatch:
            # a chunk of a sharded video that fits ONE pass goes in two: a round's clip work trails its frames by about the pass queued
            # behind them (the next round's first), and the round is gathered -- and its tracker replay can start on rank 0 -- only then
            return [n_frames // 2, n_frames]
        if taper and n_frames > fbatch:
            h = max(fbatch // 2, 1)
            t = max(min(tail, h), 1) if tail > 0 else h
            bounds = [h] + list(range(h + fbatch, n_frames, fbatch)) + [n_frames]
            if len(bounds) >= 2 and bounds[-1] - bounds[-2] > t and n_frames - t - bounds[-2] >= 4:
                bounds.insert(-1, n_frames - t)
            if len(bounds) >= 3 and bounds[-1] - bounds[-2] < max(min(t, h) // 2, 1):
                # a last pass of a handful of frames (a 60-frame chunk + its 3-frame halo: 20 / 40 / 3) runs every GEMM of the
                # per-frame stages on a sliver: split what lies behind the first pass evenly instead (20 / 22 / 21)
                r = n_frames - h
                k = -(-r // fbatch)
                bounds = [h] + [h + (r * (i + 1)) // k for i in range(k)]
        return bounds

    @staticmethod
    def stacked_view(imgs):
        """A list of per-frame host tensors that are consecutive views of ONE buffer (a loader that decoded into one block,
        `list(video)` of a stacked tensor) as a single [L, ...] view of that buffer, else None."""
        f0 = imgs[0]
        if f0.is_cuda or not f0.is_contiguous() or len(imgs) < 2:
            return None
        nb = f0.numel() * f0.element_size()
        if not all(f.dtype == f0.dtype and f.shape == f0.shape and f.is_contiguous() and f.data_ptr() == f0.data_ptr() + i * nb
                   for i, f in enumerate(imgs)):
            return None
        try:                                               # (as_strided checks the storage bounds)
            return f0.as_strided((len(imgs),) + tuple(f0.shape), (f0.numel(),) + tuple(f0.stride()))
        except RuntimeError:
            return None

    @staticmethod
    def clip_schedule(L, T, stride):
        """Clip list of mdqe/mdqe.py:308-312: (start, end, is_last); the loop ends at the first clip that
        reaches past the video (it is clamped and may be shorter than T)."""
        clips = []
        for start in range(0, L, stride):
            end = start + T
            last = end > L
            clips.append((start, min(end, L), last))
            if last:
                break
        return clips

    CACHE_GB = float(os.environ.get("MDQE_CACHE_GB", "24"))       # HBM budget of ONE frame-cache buffer (a long video uses two)

    def iter_clip_results(self, frames_dev, clips, frame_offset=0, trace=None, primed=False, on_frames_queued=None, h2d=None,
                          halo=None, side_streams=True, prime_all=True, split_small=False):
        """Per-frame features (computed once, streamed in passes of `frame_batch` frames) + decoder + inference_clip for
        `clips` (global frame indices; frames_dev[0] is global frame `frame_offset`).  Yields (start, end, last, res).

        Two HIP streams: the per-frame stages (backbone .. decoder value cache: large GEMMs) of the next passes run on a frame
        stream while the decoder + inference_clip of group k (small kernels and two host syncs) run on the caller's
        stream, so the matrix cores stay fed through the data-dependent part.

        The frame cache (round 5) is ONE linear buffer per kind, row = frame: every pass stores its results in place at the rows of its
        frames (`_frame_cache`) and a group of clips reads them through index tables (`starts`), so nothing is copied from buffer to
        buffer -- no carried T-1 frames, no first fill -- and the decoder batch is free of the pass size (`dec_batch`).  A video
        longer than the cache budget (MDQE_CACHE_GB per buffer, 24 GB = 360 frames at 360p) continues in a second buffer, to which the
        frames still to be read are copied once per segment; the buffers then alternate, events order their reuse."""
        eng, cfg = self.engine, self.cfg
        h, w = int(frames_dev.shape[-2]), int(frames_dev.shape[-1])
        geo = eng.geometry(h, w)
        n_local = frames_dev.shape[0]
        Tmax = max((c[1] - c[0] for c in clips), default=1)
        Tn = cfg.n_frames_test
        fbatch = self.frame_batch if self.frame_batch > 0 else max(8, min(40, 306000 // max(geo.N, 1)))
        # halo exchange (sharded videos, sharding._Halo): `clips` may START up to T-1 frames before this chunk; those clips read
        # the LEFT neighbour's last T-1 frames, which arrive as encoder tokens + mask features and get their cache entries in the
        # `lead` rows IN FRONT of this chunk's first frame (so a straddling clip's frames are consecutive rows); in return the tokens
        # of this chunk's last T-1 frames are handed to `halo.on_tail` as soon as the last pass is queued.
        strad, lead = [], 0
        if halo is not None:
            strad = [c for c in clips if c[0] < frame_offset]
            clips = [c for c in clips if c[0] >= frame_offset]
            if any(c[1] - c[0] != Tn or c[0] < frame_offset - (Tn - 1) for c in strad) or not clips:
                raise RuntimeError("halo exchange: a chunk must hold at least one whole clip and its straddling clips T frames")
            lead = Tn - 1 if strad else 0
        bounds = self.pass_bounds(n_local, fbatch, self.taper_passes, self.taper_tail, split_small=split_small)
        cuda = frames_dev.is_cuda
        clip_stream = torch.cuda.current_stream(frames_dev.device) if cuda else None
        if cuda and self._frame_stream is None:
            self._frame_stream = torch.cuda.Stream(frames_dev.device)
        fstream = (self._frame_stream if self.overlap_streams else clip_stream) if cuda else None
        fctx = (lambda: torch.cuda.stream(fstream)) if cuda else contextlib.nullcontext
        dec_batch = max(0, int(self.dec_batch))   # clips per decoder batch the planner waits for (0: whatever one frame pass completes)
        NR = max(2, self.lookahead + 1)           # groups prepared (their frame passes queued) ahead of the one being decoded + 1
        # ---- cache capacity (frames): the whole chunk if the budget allows, else segments in two alternating buffers
        shapes = eng.cache_shapes(geo, keep_enc=halo is not None)
        per_frame = 4 * sum(int(np.prod(sh)) for sh in shapes.values())
        forced = int(os.environ.get("MDQE_CACHE_FRAMES", "0"))                 # (tests / A-B: force short segments)
        gframes = max(fbatch, dec_batch) + Tmax - 1 + fbatch                    # frames one group can span, passes rounded up
        floor = lead + (NR + 1) * gframes                                       # what the groups in flight need at the very least
        budget = forced if forced > 0 else int(self.CACHE_GB * 2 ** 30 // max(per_frame, 1))
        want = n_local + lead
        capf = min(want, max(budget, floor))
        single = capf >= want
        if single:
            capf = -(-capf // 8) * 8              # (rounded: videos of similar length reuse the allocator's blocks)
        elif halo is not None:
            raise RuntimeError("halo exchange: a chunk (%d frames) must fit one frame-cache buffer (%d frames; MDQE_CACHE_GB)" % (n_local, capf))
        bufs = [self.alloc_cache(capf, geo, keep_enc=halo is not None), None]
        readers = [[], []]                        # buffer -> group states that read it
        seg = {"b": 0, "base": 0, "lead": lead}   # row of local frame f in buffer b: f - base + lead
        nxt = 0                                   # local frames [0, nxt) have been through the per-frame stages
        started = False

        def switch_segment(ls):
            """Continue in the other buffer, whose row 0 becomes local frame `ls` (the first frame a clip not yet planned reads):
            frames [ls, nxt) are copied over once; the groups that read that buffer two segments ago have finished (events)."""
            nb = 1 - seg["b"]
            if bufs[nb] is None:
                bufs[nb] = self.alloc_cache(capf, geo)
            keep = nxt - ls
            with fctx():
                for st in readers[nb]:
                    if "done" not in st:
                        raise RuntimeError("frame cache: a buffer is needed again while a group still reads it (capacity %d frames)" % capf)
                    if cuda and st["done"] is not None:
                        fstream.wait_event(st["done"])
                readers[nb] = []
                if keep > 0:
                    o = ls - seg["base"] + seg["lead"]
                    for k, v in bufs[nb].items():
                        v[:keep].copy_(bufs[seg["b"]][k][o:o + keep])
            seg.update(b=nb, base=ls, lead=0)

        def queue_pass():
            """The next frame pass, asynchronous on the frame stream, stored in place at the rows of its frames."""
            nonlocal nxt, started
            c1 = next(b for b in bounds if b > nxt)
            row = nxt - seg["base"] + seg["lead"]
            with fctx():
                if cuda:
                    if not started and fstream is not clip_stream:
                        fstream.wait_stream(clip_stream)               # the frames (and weights) were produced on the caller's stream
                    if h2d:                                            # the upload chunks this pass reads (upload_frames)
                        for end, ev in h2d:
                            if end > nxt:
                                fstream.wait_event(ev)
                            if end >= c1:
                                break
                started = True
                # what the DECODER reads of this pass is complete (the mask features may still be on their way).  (Also with the halo
                # exchange since round 5: the neighbour's rows are written on the clip stream itself, in front of the group that reads them.)
                dr = torch.cuda.Event() if cuda and self.early_decode and (halo is None or HALO_EARLY_DECODE) else None
                self._frame_cache(frames_dev[nxt:c1], geo, ring=bufs[seg["b"]], at=row, dec_ready=dr)
                ready = None
                if cuda:
                    ready = torch.cuda.Event()
                    ready.record(fstream)
            nxt = c1
            return {"end": c1, "ready": ready, "dec_ready": dr}

        def frames_queued():
            """Once, when the LAST frame of this video has been queued on the frame stream: a caller that streams videos
            queues the next video's first pass now, under this video's remaining clip / tracker work."""
            nonlocal on_frames_queued
            if on_frames_queued is not None and nxt >= n_local:
                cb, on_frames_queued = on_frames_queued, None
                cb()

        tail_state = {"views": None, "ev": None, "consuming": False}

        def send_tail():
            """The grouped send/recv of the halo exchange, behind the chunk's last pass.  Never while the
            generator is being primed: every rank must issue it at the same point of its program -- between the gathers of two
            rounds -- or a rank whose chunk is a single pass would queue it BEFORE the previous round's gather and RCCL, which runs
            a rank's operations in issue order, would deadlock against a rank that queued it after."""
            if halo is None or halo.tail_sent or tail_state["views"] is None:
                return
            if not tail_state["consuming"]:       # (guard for future edits: calling this from plan_next / the priming loop hangs ranks)
                raise RuntimeError("halo exchange: send_tail() while the generator is being primed -- the grouped send/recv must be "
                                   "issued between the gathers of two rounds on every rank")
            if not cuda or not HALO_OWN_STREAM:
                with fctx():
                    if cuda:
                        fstream.wait_event(tail_state["ev"])
                    halo.on_tail(*tail_state["views"])
                return
            # On the model's COPY stream, behind the event of the chunk's last pass only.  (Rounds 3-4 issued it on the frame stream --
            # which by now holds the NEXT round's primed passes: the message, and with it the last decoder group of every rank of this
            # round, waited for 1-3 passes of the next round.)
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(frames_dev.device)
            hs = self._copy_stream
            hs.wait_event(tail_state["ev"])
            with torch.cuda.stream(hs):
                halo.on_tail(*tail_state["views"])
            for v in tail_state["views"]:
                v.record_stream(hs)

        from collections import deque
        states = deque()                          # prepared (their frames queued on the frame stream) and not yet decoded, in clip order
        plan = {"ci": 0}

        def plan_next():
            """Fix the next group of clips and queue the frame passes it still needs: the group starts at the first clip not yet
            planned, the planner queues passes until `dec_batch` clips (at least one) are complete, and every further clip of the same
            length whose frames are then cached joins the batch -- clips are independent through the decoder, so they run as ONE pass
            (M = clips*T*Q rows)."""
            nonlocal nxt
            ci = plan["ci"]
            if ci >= len(clips):
                return False
            off = frame_offset
            T = clips[ci][1] - clips[ci][0]
            ls = clips[ci][0] - off
            if ls > nxt:
                nxt = ls                          # CLIP_STRIDE > clip length: the frames between two clips are never read
            k = ci
            while k + 1 < len(clips) and k + 1 - ci < max(dec_batch, 1) and clips[k + 1][1] - clips[k + 1][0] == T:
                k += 1
            target = clips[k][1] - off
            end = nxt
            while end < target:
                end = next(b for b in bounds if b > end)
            if end - seg["base"] + seg["lead"] > capf:
                switch_segment(ls)
            nxt_before = nxt
            passes = []
            while nxt < end:
                passes.append(queue_pass())
            rows = nxt - seg["base"] + seg["lead"]
            b = seg["b"]
            if halo is not None and nxt >= n_local and tail_state["views"] is None and not halo.tail_sent:
                k_t = min(Tn - 1, n_local)
                tail_state["views"] = (bufs[b]["enc"][rows - k_t:rows], bufs[b]["mf"][rows - k_t:rows])
                tail_state["ev"] = passes[-1]["ready"] if passes else None
                if cuda and tail_state["ev"] is None:
                    tail_state["ev"] = torch.cuda.Event()
                    tail_state["ev"].record(fstream)
            j = ci
            while j < len(clips) and clips[j][1] - off <= nxt and clips[j][1] - clips[j][0] == T:
                j += 1
            ready = passes[-1]["ready"] if passes else None
            if cuda and ready is None:            # no new pass (rows of earlier passes, maybe copied by a segment switch): all that is queued so far
                ready = torch.cuda.Event()
                with fctx():
                    ready.record(fstream)
            st = {"b": b, "base": seg["base"], "lead": seg["lead"], "i": ci, "j": j, "T": T, "ready": ready,
                  "dec_ready": passes[-1]["dec_ready"] if passes else None, "new_frames": nxt - nxt_before, "rows": rows,
                  "cache": {k_: v[:rows] for k_, v in bufs[b].items()}}
            readers[b].append(st)
            plan["ci"] = j
            states.append(st)
            frames_queued()
            return True

        def rows_of(st, group):
            return [c[0] - frame_offset - st["base"] + st["lead"] for c in group]

        plan_next()
        if primed:
            # a primed generator (sharded videos) is resumed only after the caller has gathered the previous round -- host syncs,
            # collectives: the frame stream gets its whole look-ahead now so that it does not run dry meanwhile.  prime_all=False
            # (forward_stream): only the first pass -- the host has the previous video's last decoder batch to launch right now, and
            # three passes of launches (13 ms of host time) in front of it cost more than they hide
            while prime_all and len(states) < NR and plan_next():
                pass
            yield None                            # per-frame work of the first passes is queued; the caller resumes later
        tail_state["consuming"] = True            # from here on the caller has gathered the previous round: the exchange may be issued
        while states:
            # `lookahead` groups' passes are queued BEFORE this group's clip work: with one, the frame stream ran dry at every group
            # boundary -- the clip kernels share the chip with the pass queued behind them and finish together with it, and
            # the next pass was only queued after the host had consumed the group (2-3 ms of whole-GPU idle per boundary in
            # the HIP trace, tools/trace_gaps.py)
            while len(states) < NR and plan_next():
                pass
            send_tail()
            cur = states.popleft()
            cache, T = cur["cache"], cur["T"]
            group = clips[cur["i"]:cur["j"]]
            dr = cur.get("dec_ready") if cuda else None
            if cuda:
                clip_stream.wait_event(dr if dr is not None else cur["ready"])
            starts = rows_of(cur, group)
            if strad and T == Tn and (cur["j"] >= len(clips) or clips[cur["j"]][1] - clips[cur["j"]][0] != Tn):
                # the last full-length group of the chunk: the straddling clips join it.  Cache rows [0, T-1) = the left neighbour's
                # last T-1 frames, directly in front of this chunk's first frame
                enc_h, mf_h = halo.head()                      # (waits for the neighbour's message on this stream)
                self._cache_from_tokens(enc_h, mf_h, geo, bufs[cur["b"]], 0)
                starts = starts + rows_of(cur, strad)
                group = group + strad
                strad = []
            outs = cur.pop("outs", None)
            if outs is None:
                # side_streams=False (the sharded schedule): HIP multiplexes its streams onto 4 hardware queues, and with RCCL's stream
                # and the replay thread's tracker stream in play the decoder's instance-chain stream lands on a queue behind the frame
                # stream's long GEMMs -- 616 against 708 frames/s per rank (tools/hwq_ab2.sh; GPU_MAX_HW_QUEUES=8 recovers most of it,
                # 697, but costs the single-GPU path 1.6 %)
                outs = eng.decode_clips(cache, starts, T, geo, two_streams=side_streams)
            elif cuda:
                clip_stream.wait_stream(self._ahead_stream)        # decoded ahead (below) on the auxiliary stream
                for v in outs.values():
                    v.record_stream(clip_stream)
            # Decode-ahead: a following group that needs NO new frame pass -- the short clips at the end of a video, whose frames the
            # last pass has already produced -- is decoded now, on an auxiliary stream beside this group's decoder, instead of after
            # this group's inference_clip syncs and tracker run.  Both decoders are chains of small dependent launches (latency-bound
            # at these sizes), so the second one is nearly free: the tail of a 120-frame video drops by the 2.6 ms the lone 3-frame
            # clip took (tools/tail_time.py).  Same kernels, same inputs, same bits.
            nx = states[0] if states else None
            if (cuda and self.decode_ahead and side_streams and nx is not None and nx["new_frames"] == 0 and not strad and trace is None
                    and "outs" not in nx):
                if self._ahead_stream is None:
                    self._ahead_stream = torch.cuda.Stream(frames_dev.device, priority=-1)
                aux = self._ahead_stream
                aux.wait_event(nx["ready"])                        # its frames (the pass this group waits for too)
                with torch.cuda.stream(aux):
                    nx["outs"] = eng.decode_clips(nx["cache"], rows_of(nx, clips[nx["i"]:nx["j"]]), nx["T"], geo)
            if dr is not None:
                clip_stream.wait_event(cur["ready"])               # the mask features of the group's last pass (inference_clip reads them)
            ress = eng.inference_clips(outs, cache["mf"], starts, T)
            ready = None
            if cuda:
                ready = torch.cuda.Event()
                ready.record(clip_stream)         # the clip results are complete once this event fires
            cur["done"] = ready                   # ... and the group no longer reads its buffer
            for gi, ((start, end, last), res) in enumerate(zip(group, ress)):
                if trace is not None:
                    trace.append({k_: v.clone() for k_, v in res.items() if torch.is_tensor(v)})
                res["ready"] = ready
                res["batch_end"] = gi == len(group) - 1
                yield start, end, last, res
        if strad:
            raise RuntimeError("halo exchange: the chunk has no full-length group for its straddling clips (chunk_plan(..., halo_exchange=True) "
                               "merges a short last chunk into its neighbour)")

    def merge_clips(self, results, frame_hw, out_size, mask_hw, n_frames=None):
        """Tracker + window flushes + video merge (mdqe/mdqe.py:337-366) over clip results in global order."""
        m = ClipMerger(self, frame_hw, out_size, mask_hw, n_frames)
        buf = []
        for item in results:                       # the clips of one decoder batch go to the tracker together
            buf.append(item)
            if item[3].get("batch_end", True):
                done = m.feed_many(buf)
                buf = []
                if done:
                    break
        if buf:
            m.feed_many(buf)
        return m.finish()

    H2D_CHUNK = 10                                  # frames per host->device copy (one event each)

    def upload_frames(self, imgs):
        """a1's host->device step (mdqe/mdqe.py:473-484 copies every frame inside the call): frames that arrive in host
        memory go to ONE device buffer in chunks of a few frames on the model's copy stream, an event per chunk; the
        per-frame stages wait only for the chunks they read, so the upload of the rest of the video runs under the first
        pass (asynchronous when the frames are pinned; pageable frames are staged by the runtime).  A list whose tensors are
        consecutive views of one buffer (a loader that decoded into one block) is copied chunk-wise too.
        Returns (frames [L,3,h,w] on the device, [(end_frame, event), ...] or None when nothing is in flight)."""
        if torch.is_tensor(imgs):
            stack = imgs
        else:
            imgs = list(imgs)
            if len(imgs) == 0:
                raise RuntimeError("MDQE: a video needs at least one frame")
            f0 = imgs[0]
            stack = self.stacked_view(imgs)
            if stack is None and f0.is_cuda:
                stack = torch.stack(imgs)
        if self.device.type != "cuda" and stack is None:
            stack = torch.stack(imgs)
        if stack is not None and (stack.is_cuda or self.device.type != "cuda"):
            if stack.dtype not in (torch.uint8, torch.float32):
                stack = stack.float()
            return stack.to(self.device).contiguous(), None
        # host frames -> device, chunked on the copy stream
        first = stack[0] if stack is not None else imgs[0]
        L = stack.shape[0] if stack is not None else len(imgs)
        dt = first.dtype if first.dtype in (torch.uint8, torch.float32) else torch.float32
        dev = torch.empty((L,) + tuple(first.shape), dtype=dt, device=self.device)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        cs = self._copy_stream
        cs.wait_stream(torch.cuda.current_stream(self.device))     # (the buffer may be a recycled block still read upstream)
        dev.record_stream(cs)
        events = []
        with torch.cuda.stream(cs):
            for a in range(0, L, self.H2D_CHUNK):
                b = min(L, a + self.H2D_CHUNK)
                if stack is not None:
                    dev[a:b].copy_(stack[a:b], non_blocking=True)
                else:
                    for i in range(a, b):
                        dev[i].copy_(imgs[i], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cs)
                events.append((b, ev))
        return dev, events

    def to_device_frames(self, imgs):
        """All frames on the device, visible to the current stream."""
        dev, events = self.upload_frames(imgs)
        if events:
            torch.cuda.current_stream(self.device).wait_event(events[-1][1])
        return dev

    def _frames_for(self, video):
        """(frames on the device, upload events, original (h0, w0)) of one input dict; the optional device-side resize
        (the mapper's eval augmentation) needs the whole upload."""
        frames_dev, h2d = self.upload_frames(video["image"])
        h0, w0 = int(frames_dev.shape[-2]), int(frames_dev.shape[-1])
        if self.resize_on_device and frames_dev.dtype == torch.uint8:
            from .preprocess import resize_shortest_edge
            if h2d:
                torch.cuda.current_stream(self.device).wait_event(h2d[-1][1])
                h2d = None
            frames_dev = resize_shortest_edge(frames_dev, self.cfg.min_size_test, self.cfg.max_size_test)
        return frames_dev, h2d, h0, w0

    def inference_vis(self, batched_inputs, trace=None):
        """mdqe/mdqe.py:291-366 with the compute-once schedule (same clips, same flush points)."""
        cfg = self.cfg
        video = batched_inputs[0]
        frames_dev, h2d, h0, w0 = self._frames_for(video)
        L, h, w = frames_dev.shape[0], int(frames_dev.shape[-2]), int(frames_dev.shape[-1])
        out_size = (video.get("height", h0), video.get("width", w0))       # the mapper reports the ORIGINAL size as height/width
        geo = self.engine.geometry(h, w)
        ms = cfg.match_stride
        clips = self.clip_schedule(L, cfg.n_frames_test, cfg.clip_stride)
        return self.merge_clips(self.iter_clip_results(frames_dev, clips, 0, trace, h2d=h2d), (h, w), out_size,
                                (geo.Hp // ms, geo.Wp // ms), n_frames=L)

    def forward_stream(self, batches):
        """An eval loop over videos: yields `forward(b)` for every b of the iterable `batches`, in order, bit-identical to
        separate calls -- but with one video of look-ahead: as soon as the last frame of video k is on the frame stream,
        the first pass of the per-frame stages of video k+1 is queued behind it, so it runs under video k's last decoder
        batch, tracker updates and mask read-back instead of after them (the chip is half idle there).  The reference's
        evaluator (`inference_on_dataset`) hands videos over one by one; this is that loop with the hand-over moved
        forward.  COCO image sets and CPU runs fall back to plain calls."""
        it = iter(batches)

        def start(b):
            """Frames to the device, geometry, clip schedule; queues the first pass (the generator is primed)."""
            if len(b) != 1:
                raise RuntimeError("MDQE eval takes exactly one video per call (mdqe/mdqe.py:292)")
            cfg = self.cfg
            video = b[0]
            frames_dev, h2d, h0, w0 = self._frames_for(video)
            L, h, w = frames_dev.shape[0], int(frames_dev.shape[-2]), int(frames_dev.shape[-1])
            st = {"done_frames": False, "out_size": (video.get("height", h0), video.get("width", w0)), "hw": (h, w), "L": L}
            geo = self.engine.geometry(h, w)
            st["mask_hw"] = (geo.Hp // cfg.match_stride, geo.Wp // cfg.match_stride)
            clips = self.clip_schedule(L, cfg.n_frames_test, cfg.clip_stride)

            def cb():
                st["done_frames"] = True
                look_ahead(st)
            st["gen"] = self.iter_clip_results(frames_dev, clips, 0, None, primed=True, on_frames_queued=cb, h2d=h2d, prime_all=False)
            next(st["gen"])
            return st

        state = {"cur": None, "next": None}

        def look_ahead(st):
            # only the video being consumed may pull the next one in (a short video whose frames are all queued by its own
            # start() must not cascade through the whole stream)
            if st is state["cur"] and state["next"] is None:
                b = next(it, None)
                if b is not None:
                    try:
                        state["next"] = start(b)
                    except Exception as e:                 # belongs to the NEXT video: raised when its turn comes
                        state["next"] = {"error": e}

        first = next(it, None)
        if first is None:
            return
        if self.cfg.is_coco or self.device.type != "cuda":
            yield self.forward(first)
            for b in it:
                yield self.forward(b)
            return

        def guarded(fn):                                   # no context manager is held across a yield
            with self._on_device(), torch.autocast(device_type="cuda", enabled=False), torch.no_grad(), self.work_stream():
                return fn()

        def step():
            st = state["cur"]
            if "error" in st:
                state["cur"] = None
                raise st["error"]
            if st["done_frames"]:
                look_ahead(st)
            out = self.merge_clips(st["gen"], st["hw"], st["out_size"], st["mask_hw"], n_frames=st["L"])
            if state["next"] is None:
                look_ahead(st)                             # (the generator ends only after its callback; belt and braces)
            state["cur"], state["next"] = state["next"], None
            return out

        state["cur"] = guarded(lambda: start(first))
        while state["cur"] is not None:
            yield guarded(step)

    def inference_image(self, batched_inputs):
        """COCO single-image branch: MDQE.forward :213-236 -> mdqe.forward (models/mdqe.py:62-70) -> decoder
        eval branch `is_coco` (transformer_dec.py:247-255) -> MDQE.inference_image (mdqe/mdqe.py:486-556).  The pseudo clip
        (cfg.n_frames images) goes through the same per-frame stages and decoder as a video clip; the centre frame's masks
        are scored, box-IoU-suppressed and resized by two kernels that evaluate aligned_bilinear in closed form.
        Returns [{"instances": Instances(scores, pred_classes, pred_masks, pred_boxes)}]."""
        from . import ops
        from .d2_compat import Boxes, Instances
        cfg, eng = self.cfg, self.engine
        item = batched_inputs[0]
        frames = self.to_device_frames(item["image"])
        T, h, w = int(frames.shape[0]), int(frames.shape[-2]), int(frames.shape[-1])
        geo = eng.geometry(h, w)
        cache = self._frame_cache(frames, geo)
        out = eng.decode_clips(cache, [0], T, geo)
        cls, coef = out["cls"][0], out["mask_coeff"][0]                      # [Q,K] (sigmoid), [Q,M] (tanh)
        thr = cfg.apply_cls_thres
        score = cls.max(-1)[0]
        idx = torch.nonzero(score >= torch.clamp(score.max(), max=thr)).reshape(-1)       # :504
        ct = int((cfg.n_frames - 1) / 2)
        mf = cache["mf"][ct]                                                 # [Hm,Wm,M] channels-last
        Hm, Wm, Md = mf.shape
        lg = ops.linear(coef[idx].contiguous(), mf.reshape(-1, Md)).view(-1, Hm, Wm)     # einsum 'qm,mhw->qhw' of the centre frame
        st = ops.image_mask_stats(lg, cfg.match_stride, h, w)
        mc = cls[idx] * (st[:, 0] / (st[:, 1] + 1e-6))[:, None]             # mask-quality rescoring :512-516
        n = int(idx.numel())
        order = torch.arange(n, device=self.device)
        if n > 0:                                                            # box-IoU NMS :519-531
            order = mc.max(-1)[0].sort(descending=True)[1]
            mc, sb = mc[order], st[order]
            empty = sb[:, 2] > sb[:, 4]
            bx = torch.stack([sb[:, 2], sb[:, 3], sb[:, 4] + 1, sb[:, 5] + 1], 1)
            bx = torch.where(empty[:, None], torch.zeros_like(bx), bx) / torch.tensor([w, h, w, h], device=self.device, dtype=torch.float32)
            area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
            lt = torch.max(bx[:, None, :2], bx[None, :, :2])
            rb = torch.min(bx[:, None, 2:], bx[None, :, 2:])
            inter = (rb - lt).clamp(min=0).prod(-1)
            iou = inter / (area[:, None] + area[None] - inter).clamp(min=1e-3)
            mc = mc * (1 - torch.triu(iou, diagonal=1).max(0)[0])[:, None]
        if cfg.multi_cls:                                                    # :534-540
            ls = torch.nonzero(mc > thr)
            ii, label = ls[:, 0], ls[:, 1]
            sc = mc[ii, label]
        else:
            sc, label = mc.max(-1)
            ii = torch.arange(n, device=self.device)
        Ho, Wo = int(item.get("height", h)), int(item.get("width", w))
        sel = order[ii].to(torch.int32).contiguous()                         # rows of lg, in output order
        pm = ops.image_final_masks(lg, sel, cfg.match_stride, h, w, Ho, Wo).view(torch.bool)
        # BitMasks(mask).get_bounding_boxes() of the final masks :554
        xa, ya = pm.any(1), pm.any(2)
        has = xa.any(1)
        ar_x, ar_y = torch.arange(Wo, device=self.device), torch.arange(Ho, device=self.device)
        x0 = torch.where(xa, ar_x, Wo).min(1)[0]; x1 = torch.where(xa, ar_x, -1).max(1)[0] + 1
        y0 = torch.where(ya, ar_y, Ho).min(1)[0]; y1 = torch.where(ya, ar_y, -1).max(1)[0] + 1
        boxes = torch.stack([x0, y0, x1, y1], 1).float() * has[:, None]
        res = Instances((Ho, Wo))
        res.scores, res.pred_classes, res.pred_masks, res.pred_boxes = sc, label, pm, Boxes(boxes)
        return [{"instances": res}]

    def inference_video(self, image_size, cls_clips, windows, frame_hw, n_frames, early=None, emit_masks=True):
        """mdqe/mdqe.py:430-471.  The x4 aligned-bilinear up-sampling, sigmoid, crop (:357-358), nearest resize to the
        original size and the 0.5 threshold (:458-462) run as ONE kernel per window; windows in which an instance did not
        exist yet stay zero (:442).  `early` (ClipMerger, CUDA): the masks of every tracked instance were already produced
        and copied to pinned host memory window by window, under the later windows' compute; only the selection is left."""
        from . import ops
        K = self.cfg.num_classes
        total = cls_clips[-1].shape[0]
        self.last_num_tracks = int(total)                      # tracks of the video just merged (diagnostics; bench.py reports it)
        cc = torch.stack([torch.cat([c, c.new_zeros(total - c.shape[0], c.shape[1])]) for c in cls_clips])
        out_cls = (0.75 * cc.mean(0) + 0.25 * cc.max(0)[0]).flatten().cpu()
        k = min(max(int(out_cls.gt(0.05).sum()), 10), out_cls.numel())   # (the reference's topk(max(.,10)), :449-450, assumes >= 10 scores)
        sc, ti = out_cls.topk(k, sorted=False)
        labels = (ti % K).tolist()
        inst = torch.div(ti, K, rounding_mode="floor").tolist()
        sel = sorted(set(inst))
        Ho, Wo = int(image_size[0]), int(image_size[1])
        if not emit_masks:
            return {"image_size": (Ho, Wo), "pred_scores": sc.tolist(), "pred_labels": labels, "pred_masks": []}
        if early is not None and self.rle_output:
            from . import rle as R
            empty = {"size": [Ho, Wo], "counts": R.counts_to_strings([Ho * Wo], [1])[0].decode("utf-8")}
            per_inst = {i: [dict(empty) for _ in range(n_frames)] for i in sel}     # before an instance's first window: empty masks (:442)
            for f_off, nf, n_w, pos, n_pos in early["rle"]:
                counts, lengths = R.positions_to_counts(pos, n_pos, Ho * Wo)
                strs = R.counts_to_strings(counts, lengths)
                for i in sel:
                    if i < n_w:
                        for f in range(nf):
                            per_inst[i][f_off + f] = {"size": [Ho, Wo], "counts": strs[i * nf + f].decode("utf-8")}
            return {"image_size": (Ho, Wo), "pred_scores": sc.tolist(), "pred_labels": labels,
                    "pred_rles": [per_inst[i] for i in inst]}
        if early is not None:
            early["done"].synchronize()
            hosts = early["host"]                                  # per instance: [n_frames, Ho, Wo] uint8, pinned
            return {"image_size": (Ho, Wo), "pred_scores": sc.tolist(), "pred_labels": labels,
                    "pred_masks": [hosts[i].view(torch.bool)[:n_frames] for i in inst]}
        out = torch.zeros(len(sel), n_frames, Ho, Wo, dtype=torch.uint8, device=self.device)
        sel_dev = torch.tensor(sel, dtype=torch.int32, device=self.device)
        for f_off, m in windows:
            cnt = sum(1 for i in sel if i < m.shape[0])       # sel is ascending: these are its first `cnt` entries
            if cnt:
                ops.final_masks(m, sel_dev[:cnt], self.cfg.match_stride, frame_hw[0], frame_hw[1], Ho, Wo, out, f_off)
        if out.is_cuda:                                            # one D2H into pinned memory (pageable copies run at a fraction of PCIe)
            hbuf = torch.empty(out.shape, dtype=torch.uint8, pin_memory=True)
            hbuf.copy_(out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            host = hbuf.view(torch.bool)
        else:
            host = out.view(torch.bool)
        pos = {i: p for p, i in enumerate(sel)}
        if self.rle_output:                                        # no early path (CPU device / unknown length): encode on the host
            from . import rle as R
            enc = {i: [R.encode_dense(fm.numpy()) for fm in host[pos[i]]] for i in sel}
            return {"image_size": (Ho, Wo), "pred_scores": sc.tolist(), "pred_labels": labels, "pred_rles": [enc[i] for i in inst]}
        return {"image_size": (Ho, Wo), "pred_scores": sc.tolist(), "pred_labels": labels,
                "pred_masks": [host[pos[i]] for i in inst]}


class ClipMerger:
    """Incremental form of the clip loop's second half (mdqe/mdqe.py:337-366): tracker update per clip, window flushes,
    final video merge.  The tracker runs on its own HIP stream so that its small kernels and per-clip host syncs overlap
    with per-frame work the producer has already queued on the main stream."""

    tracker_cls = OverTracker               # (tests without a GPU substitute a stand-in bank, tests/_standins.py)
    EARLY_TRACKS = 48                       # tracks per video the early-mask path budgets pinned memory for

    def __init__(self, model, frame_hw, out_size, mask_hw, n_frames=None, emit_masks=True):
        self.model, self.frame_hw, self.out_size, self.mask_hw = model, frame_hw, out_size, mask_hw
        self.emit_masks = emit_masks                # False: scores / labels only (ranks > 0 of a sharded video)
        self.n_frames = n_frames                    # total frames of the video when known: enables the early mask path
        self.early = None
        # MODEL.MDQE.MERGE_ON_CPU (mdqe/mdqe.py:185-186,337,354-355; True in R50_ovis_720 / swinl_ovis): the device the window results
        # wait on for the end of the video -- a memory-placement switch, the outputs are the same.  WHEN the final masks are produced is
        # a separate choice (`model.early_masks`, default on for both settings since round 3): per flushed window, into pinned host
        # buffers under the later windows' compute -- the window's stride-4 logits are then dropped at once under EITHER setting (nothing
        # reads them again) -- or, off, in one pass + one copy at the end, which needs the logits of every window and keeps them in HBM
        # whatever MERGE_ON_CPU says.
        self.merge_on_cpu = bool(model.cfg.merge_on_cpu if model.merge_on_cpu is None else model.merge_on_cpu)
        self.early_on = bool(getattr(model, "early_masks", True))
        # The early path holds one pinned [n_frames, Ho, Wo] buffer per TRACK (the late path: per selected output).  Budget: an estimate
        # of EARLY_TRACKS tracks must fit into MDQE_EARLY_PINNED_GB (default 24) of pinned host memory, else the late path is taken for
        # this video (a 120-frame 360p video: 27.6 MB per track; one rank's view of a 1920-frame one: 442 MB per track).
        if self.early_on and n_frames is not None:
            per_track = int(n_frames) * int(out_size[0]) * int(out_size[1])
            if per_track * self.EARLY_TRACKS > float(os.environ.get("MDQE_EARLY_PINNED_GB", "24")) * 2 ** 30:
                self.early_on = False
        self.dev = model.device
        self.use_side = self.dev.type == "cuda"
        self.main = torch.cuda.current_stream(self.dev) if self.use_side else None
        if self.use_side and model._trk_stream is None:
            model._trk_stream = torch.cuda.Stream(self.dev, priority=getattr(model, "trk_priority", 0))
        self.side = model._trk_stream if self.use_side else None
        self.side_is_current = False                # set by sharding.ReplayThread in its own thread
        self.saved, self.tracker = 0, None
        self.cls_clips, self.windows, self.f_off = [], [], 0
        self.done = False

    def feed(self, start, end, last, res):
        """Returns True once the last clip has been consumed."""
        return self.feed_many([(start, end, last, res)])

    def feed_many(self, items):
        """Clip results in global order.  The clips between two window flushes go to the tracker as ONE native call
        (`OverTracker.update_many`: no Python between clips -- what keeps rank 0's replay of a gathered round off the critical
        path of a sharded video).  Returns True once the last clip has been consumed."""
        cfg = self.model.cfg
        stride, win = cfg.clip_stride, cfg.n_frames_window_test
        run = []
        for it in items:
            run.append(it)
            start, last = it[0], it[2]
            if last or (start + stride >= win * (self.saved + 1)):
                self._consume(run, True, last)
                run = []
                if last:
                    break
        if run:
            self._consume(run, False, False)
        return self.done

    def _consume(self, run, flush, last):
        cfg = self.model.cfg
        T, stride, win = cfg.n_frames_test, cfg.clip_stride, cfg.n_frames_window_test
        # (a replay thread makes the tracker stream its current stream once instead of entering a stream context per clip)
        ctx = torch.cuda.stream(self.side) if self.use_side and not self.side_is_current else contextlib.nullcontext()
        with ctx:
            clips, seen = [], set()
            for start, end, _, res in run:
                if self.use_side:
                    ev = res.get("ready")
                    if ev is None:
                        self.side.wait_stream(self.main)
                    elif id(ev) not in seen:            # the clips of one decoder batch share their event
                        seen.add(id(ev))
                        self.side.wait_event(ev)
                    res["pred_masks"].record_stream(self.side)
                clips.append(Clips(range(start, end), res))
            if self.tracker is None:
                self.tracker = self.tracker_cls(cfg.n_max_inst, T, win, stride, cfg.num_classes, cfg.mask_dim, cfg.hidden_dim,
                                                self.mask_hw, self.dev, cfg.apply_cls_thres)
            self.tracker.update_many(clips)
            if flush:
                c, m = self.tracker.get_result(is_last_clip=last)   # m: mean logits [n, F, Hm, Wm] of this window
                self.cls_clips.append(c)
                m = m.contiguous()
                if not self.emit_masks:
                    self.windows.append((self.f_off, None))
                elif self.use_side and self.n_frames is not None and (self.early_on or self.model.rle_output):
                    self._early_masks(m)
                    # inference_video returns from its `early` branch and never reads `windows` then: the stride-4 logits of a flushed
                    # window are not kept for the rest of the video under either MERGE_ON_CPU setting (round 3 held them for nothing)
                    self.windows.append((self.f_off, None))
                else:
                    self.windows.append((self.f_off, m))
                self.f_off += m.shape[1]
                self.saved += 1
        self.done = self.done or bool(last)

    def _early_masks(self, m):
        """Final masks of EVERY instance tracked so far for the window just flushed (m: [n, F, Hm, Wm] mean logits), copied to
        pinned host memory on a copy stream while later windows compute; finish() then only selects rows.  A few rows may
        be produced in vain (instances that miss the final top-k).  One pinned buffer per track (no re-allocation as tracks
        appear; the caching host allocator recycles the blocks of the previous call)."""
        from . import ops
        model = self.model
        n, nf = int(m.shape[0]), int(m.shape[1])
        Ho, Wo = int(self.out_size[0]), int(self.out_size[1])
        if model._copy_stream is None:
            model._copy_stream = torch.cuda.Stream(self.dev)
        cs = model._copy_stream
        if self.early is None:
            self.early = {"host": [], "windows": [], "done": torch.cuda.Event(), "rle": []}
        if model.rle_output:                        # run boundaries instead of dense masks: KBs instead of MBs per window
            if n:
                idx = torch.arange(n, dtype=torch.int32, device=self.dev)
                cap = 4 * (Ho + Wo) + 64                # a blob crosses a column twice: generous for anything mask-like
                while True:
                    pos, n_pos = ops.final_masks_rle(m, idx, model.cfg.match_stride, self.frame_hw[0], self.frame_hw[1], Ho, Wo, cap)
                    mx = int(n_pos.max())               # (sync on the tracker stream; the window's logits are final here)
                    if mx <= cap:
                        break
                    cap = mx
                self.early["rle"].append((self.f_off, nf, n, pos[:, :max(mx, 1)].cpu().numpy(), n_pos.cpu().numpy()))
            self.early["windows"].append((self.f_off, nf, n))
            return
        hosts = self.early["host"]
        while len(hosts) < n:                       # a new track: its own pinned [L, Ho, Wo] buffer, zero before its first window (:442)
            hbuf = model.pinned_mask_buffer((int(self.n_frames), Ho, Wo))
            if self.f_off > 0:
                hbuf[:self.f_off].zero_()
            hosts.append(hbuf)
        if n:
            dev = torch.empty(n, nf, Ho, Wo, dtype=torch.uint8, device=self.dev)
            idx = torch.arange(n, dtype=torch.int32, device=self.dev)
            ops.final_masks(m, idx, model.cfg.match_stride, self.frame_hw[0], self.frame_hw[1], Ho, Wo, dev, 0)
            cs.wait_stream(self.side)
            with torch.cuda.stream(cs):
                for i in range(n):
                    hosts[i][self.f_off:self.f_off + nf].copy_(dev[i], non_blocking=True)
                dev.record_stream(cs)
                self.early["done"].record(cs)
        self.early["windows"].append((self.f_off, nf, n))

    def finish(self):
        if self.use_side:
            self.main.wait_stream(self.side)
            for _, m in self.windows:
                if m is not None:
                    m.record_stream(self.main)
        return self.model.inference_video(self.out_size, self.cls_clips, self.windows, self.frame_hw, self.f_off, early=self.early,
                                          emit_masks=self.emit_masks)


class MDQE_MI355X(MDQE):
    """The same model under a name of its own: `MODEL.META_ARCHITECTURE MDQE_MI355X` on the reference's command line selects this
    implementation whatever else is registered as "MDQE"."""


_REGISTRATION = {"state": "detectron2 not imported"}


def register_with_detectron2(registry=None, takeover=None):
    """Put this implementation into detectron2's `META_ARCH_REGISTRY` BESIDE the reference's `mdqe` package.

    The reference registers its own torch model under the name "MDQE" the moment `mdqe` is imported (`mdqe/__init__.py:3` ->
    `@META_ARCH_REGISTRY.register()` at `mdqe/mdqe.py:60-61`), and `train_net.py:40-43` / `demo/demo.py:16` import `mdqe` for
    `add_mdqe_config` and the data loaders; fvcore's `Registry` asserts on a second registration of a name.  So:

    * the alias `MDQE_MI355X` is always registered (no collision possible);
    * the name "MDQE" -- what every config of the reference selects (`configs/*.yaml: META_ARCHITECTURE: "MDQE"`) -- is TAKEN OVER
      (`takeover=True`, the default; `MDQE_MI355X_REGISTER=alias` in the environment turns it off): a reference model registered earlier
      is moved to "MDQE_REFERENCE" and this class takes its slot, with a warning in the log; a registration of another "MDQE" that
      comes LATER (this package imported before the reference's) is diverted to "MDQE_REFERENCE" instead of tripping fvcore's
      assertion.  Both import orders end with `build_model(cfg)` constructing this class and the reference's model still selectable.

    Returns a dict describing what was done (also kept in `registration_state()`).  A missing detectron2 is silent; a detectron2 that
    is installed but fails to import (ImportError / OSError) is a logged warning recorded in `registration_state()` -- the d2-free
    `MDQE(cfg)` keeps working -- unless `MDQE_MI355X_REGISTER=strict`; a failure of the registration itself propagates."""
    import logging
    log = logging.getLogger("mdqe_cvpr2023_amd")
    if registry is None:
        try:
            from detectron2.modeling import META_ARCH_REGISTRY as registry
        except (ImportError, OSError) as e:
            # ModuleNotFoundError of detectron2 itself: absent in this image, the d2-free entry points are all there is (silent).
            # Any other ImportError / OSError: a detectron2 that is installed but does not import here -- typically `from detectron2
            # import _C` on a ROCm box without its compiled ops.  That must not take the d2-free `MDQE(cfg)` entry point down with it:
            # warn, record, go on.  `MDQE_MI355X_REGISTER=strict` raises instead.  (A failure of the REGISTRATION below -- a registry
            # that misbehaves -- still propagates.)
            absent = isinstance(e, ModuleNotFoundError) and (e.name or "").split(".")[0] == "detectron2"
            if not absent:
                if os.environ.get("MDQE_MI355X_REGISTER", "") == "strict":
                    raise
                log.warning("detectron2 is installed but `detectron2.modeling` does not import (%s: %s): mdqe_cvpr2023_amd.MDQE is NOT in "
                            "META_ARCH_REGISTRY; the detectron2-free entry point MDQE(cfg) works (MDQE_MI355X_REGISTER=strict makes this "
                            "an error)", type(e).__name__, e)
            _REGISTRATION.clear()
            _REGISTRATION.update(state="detectron2 not importable" if absent else "detectron2 import failed",
                                 error=None if absent else "%s: %s" % (type(e).__name__, e))
            return dict(_REGISTRATION)
    if takeover is None:
        takeover = os.environ.get("MDQE_MI355X_REGISTER", "replace") != "alias"
    objs = registry._obj_map
    if objs.get("MDQE_MI355X") is not MDQE_MI355X:
        registry.register(MDQE_MI355X)            # (a foreign class of that name would assert here, as it should)
    done = {"state": "alias only", "alias": "MDQE_MI355X", "MDQE": None, "MDQE_REFERENCE": None}
    if takeover:
        prev = objs.get("MDQE")
        if prev is None:
            registry.register(MDQE)
        elif prev is not MDQE:
            objs["MDQE_REFERENCE"] = prev
            objs["MDQE"] = MDQE
            log.warning("META_ARCH_REGISTRY['MDQE'] now builds mdqe_cvpr2023_amd.meta_arch.MDQE (MI355X); the class registered before "
                        "(%s.%s) stays selectable as 'MDQE_REFERENCE'", getattr(prev, "__module__", "?"), getattr(prev, "__qualname__", "?"))
        if not getattr(registry, "_mdqe_mi355x_guard", False):
            inner = registry._do_register

            def _do_register(name, obj, _inner=inner, _objs=objs):
                if name == "MDQE" and _objs.get("MDQE") is MDQE and obj is not MDQE:
                    _objs["MDQE_REFERENCE"] = obj
                    log.warning("a second 'MDQE' (%s.%s) was registered after mdqe_cvpr2023_amd's: kept as 'MDQE_REFERENCE'; "
                                "'MDQE' keeps building the MI355X implementation", getattr(obj, "__module__", "?"), getattr(obj, "__qualname__", "?"))
                    return
                _inner(name, obj)
            registry._do_register = _do_register
            registry._mdqe_mi355x_guard = True
        done.update(state="MDQE taken over", MDQE="mdqe_cvpr2023_amd.meta_arch.MDQE")
    ref = objs.get("MDQE_REFERENCE")
    done["MDQE_REFERENCE"] = None if ref is None else "%s.%s" % (getattr(ref, "__module__", "?"), getattr(ref, "__qualname__", "?"))
    _REGISTRATION.clear()
    _REGISTRATION.update(done)
    return dict(done)


def registration_state():
    """What `register_with_detectron2` did at import (for logs and tests)."""
    return dict(_REGISTRATION)


def _register_backbone():
    """The Swin builder goes into BACKBONE_REGISTRY whenever the meta-architecture went into META_ARCH_REGISTRY (same policy)."""
    if _REGISTRATION.get("state") in ("MDQE taken over", "alias only"):
        from .backbone import register_backbone_with_detectron2
        try:
            from detectron2.modeling import BACKBONE_REGISTRY  # noqa: F401
        except (ImportError, OSError, AttributeError):
            return                                   # (a stand-in detectron2 without a backbone registry)
        register_backbone_with_detectron2()


register_with_detectron2()                        # silent without detectron2, a warning if it is there but does not import, loud otherwise
_register_backbone()
