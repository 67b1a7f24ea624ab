"""Cross-clip tracker: the decisions of the reference's OverTracker
(mdqe/tracking/OverTracker.py:10-242) on a state split between device and host, both native.

Reference state: saved_logits [clips, instances, frames, h, w] (8 GB at R50_ovis_360, re-zeroed per
window) that is only ever consumed through sum-over-clips, plus ~70 tiny device ops and several
host syncs per clip.  Here:
  * device: a running per-(instance, frame) SUM of mask logits + a count (250 MB), touched by the HIP
    kernels of csrc/tracker.hip: sign-intersection counts for the hard-mask IoU, an indexed
    accumulate, the per-window mean and the re-basing of the bank at a window flush;
  * host, C++ (csrc/tracker_native.hip): everything that is O(instances x 256): embeddings, class
    probabilities, presence flags, untracked counters, the bi-softmax similarity, the rectangular
    assignment (scipy's algorithm restated; the reference calls scipy at :159) and the ID bookkeeping;
  * one small device->host copy + stream sync per update (the [n_saved, n_in, 3] count tensor),
    inside the native call (the GIL is released for its duration);
  * `update_many`: a run of clips in ONE native call -- the replay of a gathered round on rank 0 of a
    sharded video never returns to Python between clips.
There is one path: the bank lives on a HIP device.  (CPU tests drive the same native host core with a
torch stand-in for the bank, tests/_standins.py.)
"""
import ctypes

import numpy as np
import torch

from ._lib import check, cur_stream, lib


def lsap(cost, maximize=False):
    """scipy.optimize.linear_sum_assignment's algorithm as restated in csrc/tracker_native.hip -> (rows, cols) int arrays."""
    c = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = c.shape
    k = min(nr, nc)
    r = np.empty(max(k, 1), dtype=np.int32)
    cc = np.empty(max(k, 1), dtype=np.int32)
    n = ctypes.c_int(0)
    check(lib.mdqe_lsap_f64(c.ctypes.data, nr, nc, int(bool(maximize)), r.ctypes.data, cc.ctypes.data, ctypes.byref(n)), "lsap")
    return r[:n.value].copy(), cc[:n.value].copy()


class Clips:
    """OverTracker.py:245-256: per-clip detections.  mask logits stay on the device; the small per-instance
    vectors are taken as host arrays (`results["host"]`, one batched copy made by the engine) when present."""

    def __init__(self, frame_idx, results):
        self.frame_idx = list(frame_idx)
        self.mask_logits = results["pred_masks"]
        h = results.get("host")
        if h is None:
            h = {k: results[k].detach().float().cpu().numpy() for k in ("scores", "cls_probs", "query_embeds")}
        self.scores = np.ascontiguousarray(h["scores"], dtype=np.float32)
        self.cls_probs = np.ascontiguousarray(h["cls_probs"], dtype=np.float32)
        self.query_embeds = np.ascontiguousarray(h["query_embeds"], dtype=np.float32)
        self.num_instance = int(self.scores.shape[0])


class TrackerCore:
    """The native host core (mdqe_tracker_*): decisions and bookkeeping, no device state."""

    def __init__(self, num_max_inst, num_frames, window_frames, clip_stride, num_classes, embed_dim, apply_cls_thres):
        self.T, self.win, self.stride = int(num_frames), int(window_frames), int(clip_stride)
        self.K, self.E = int(num_classes), int(embed_dim)
        self.max_inst = int(num_max_inst)
        self.mem_len = self.win + self.T
        h = ctypes.c_void_p()
        check(lib.mdqe_tracker_create(self.max_inst, self.T, self.win, self.stride, self.K, self.E, float(apply_cls_thres),
                                      ctypes.byref(h)), "tracker_create")
        self._h = h
        self._destroy = lib.mdqe_tracker_destroy

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._destroy(h)

    def _state(self):
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.mdqe_tracker_state(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "tracker_state")
        return a.value, b.value, c.value

    @property
    def num_inst(self):
        return self._state()[0]

    @property
    def num_clip(self):
        return self._state()[1]

    @property
    def start_frame(self):
        return self._state()[2]

    # the two-phase form (a stand-in bank, or a caller that wants the counts): overlap -> counts -> decide
    def overlap(self, clip):
        ni, s0, a, nf = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.mdqe_tracker_overlap(self._h, clip.frame_idx[0], len(clip.frame_idx), ctypes.byref(ni), ctypes.byref(s0),
                                       ctypes.byref(a), ctypes.byref(nf)), "tracker_overlap (overlapping frames must be contiguous)")
        return ni.value, s0.value, a.value, nf.value

    def decide(self, clip, counts3):
        n_in = clip.num_instance
        r = np.empty(n_in + 1, dtype=np.int32)
        c = np.empty(n_in + 1, dtype=np.int32)
        n, s0, a, nf = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        c3 = None
        if counts3 is not None:
            counts3 = np.ascontiguousarray(counts3, dtype=np.float32)
            c3 = counts3.ctypes.data
        check(lib.mdqe_tracker_decide(self._h, clip.frame_idx[0], len(clip.frame_idx), n_in, clip.scores.ctypes.data,
                                      clip.cls_probs.ctypes.data, clip.query_embeds.ctypes.data, c3, r.ctypes.data, c.ctypes.data,
                                      ctypes.byref(n), ctypes.byref(s0), ctypes.byref(a), ctypes.byref(nf)),
              "tracker_decide (tracker memory exceeded: MAX_NUM_INSTANCES / clip slots, as the reference would)")
        return r[:n.value], c[:n.value], s0.value, a.value, nf.value

    def result_host(self, is_last):
        out_c = np.empty((self.max_inst, self.K), dtype=np.float32)
        cv = np.zeros((self.max_inst, self.mem_len - self.win), dtype=np.uint8)
        n, ln = ctypes.c_int(), ctypes.c_int()
        check(lib.mdqe_tracker_result(self._h, int(bool(is_last)), out_c.ctypes.data, ctypes.byref(n), ctypes.byref(ln), cv.ctypes.data),
              "tracker_result")
        return out_c[:n.value].copy(), n.value, ln.value, cv[:n.value].astype(bool)


class OverTracker(TrackerCore):
    def __init__(self, num_max_inst, num_frames, window_frames, clip_stride, num_classes, mask_dim, embed_dim,
                 image_size, device, apply_cls_thres):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("OverTracker: the tracker bank lives on a HIP device (there is no CPU path)")
        super().__init__(num_max_inst, num_frames, window_frames, clip_stride, num_classes, embed_dim, apply_cls_thres)
        self.size = tuple(int(v) for v in image_size)
        self.hw = int(self.size[0] * self.size[1])
        if self.hw % 4 != 0:
            raise RuntimeError("OverTracker: mask map size must be a multiple of 4 pixels")
        self.device = device
        self.sum_logits = torch.zeros(self.max_inst, self.mem_len, *self.size, device=device)
        self.cnt = torch.zeros(self.max_inst, self.mem_len, device=device)
        self.carry = torch.empty(self.max_inst, self.mem_len - self.win, *self.size, device=device)
        self._counts_dev = self._counts_host = None
        self._cap = 0

    def _scratch(self, n_in):
        if self._counts_dev is None or n_in > self._cap:         # (also for a first clip without instances)
            self._cap = max(64, 2 * n_in)
            self._counts_dev = torch.empty(self.max_inst * self._cap * 3, device=self.device)
            self._counts_host = torch.empty(self.max_inst * self._cap * 3, pin_memory=True)
        return self._counts_dev.data_ptr(), self._counts_host.data_ptr()

    @staticmethod
    def _masks(clip):
        m = clip.mask_logits
        if m.dtype != torch.float32 or m.stride(-1) != 1 or (m.dim() == 4 and (m.stride(2) != m.shape[3] or m.stride(1) != m.shape[2] * m.shape[3])):
            m = m.float().contiguous()
            clip.mask_logits = m
        return m

    # ---- OverTracker.update (:115-193) ------------------------------------------------------------
    def update(self, clip: Clips):
        m = self._masks(clip)
        n_in = clip.num_instance
        cd, ch = self._scratch(n_in)
        check(lib.mdqe_tracker_update(self._h, self.sum_logits.data_ptr(), self.cnt.data_ptr(), self.hw, clip.frame_idx[0],
                                      len(clip.frame_idx), n_in, clip.scores.ctypes.data, clip.cls_probs.ctypes.data,
                                      clip.query_embeds.ctypes.data, m.data_ptr() if n_in else None, m.stride(0) if n_in else 0,
                                      cd, ch, cur_stream()), "tracker_update")

    def update_many(self, clips):
        """A run of clips in one native call (no Python between them)."""
        n = len(clips)
        if n == 0:
            return
        if n == 1:
            return self.update(clips[0])
        ms = [self._masks(c) for c in clips]
        n_in = np.array([c.num_instance for c in clips], dtype=np.int32)
        row0 = np.zeros(n, dtype=np.int32)
        row0[1:] = np.cumsum(n_in)[:-1]
        f0 = np.array([c.frame_idx[0] for c in clips], dtype=np.int32)
        nf = np.array([len(c.frame_idx) for c in clips], dtype=np.int32)
        sc = np.concatenate([c.scores for c in clips]).astype(np.float32, copy=False)
        cp = np.concatenate([c.cls_probs.reshape(-1, self.K) for c in clips]).astype(np.float32, copy=False)
        em = np.concatenate([c.query_embeds.reshape(-1, self.E) for c in clips]).astype(np.float32, copy=False)
        ptrs = (ctypes.c_void_p * n)(*[(m.data_ptr() if k else None) for m, k in zip(ms, n_in)])
        strides = np.array([(m.stride(0) if k else 0) for m, k in zip(ms, n_in)], dtype=np.int64)
        cd, ch = self._scratch(int(n_in.max()))
        check(lib.mdqe_tracker_update_many(self._h, self.sum_logits.data_ptr(), self.cnt.data_ptr(), self.hw, n, f0.ctypes.data,
                                           nf.ctypes.data, n_in.ctypes.data, row0.ctypes.data, sc.ctypes.data, cp.ctypes.data,
                                           em.ctypes.data, ptrs, strides.ctypes.data, cd, ch, cur_stream()), "tracker_update_many")

    # ---- OverTracker.get_result (:195-225) --------------------------------------------------------
    def get_result(self, is_last_clip=False):
        n = self.num_inst
        out_c = np.empty((max(n, 1), self.K), dtype=np.float32)
        cap_f = self.mem_len if is_last_clip else self.win
        out_m = torch.empty(n, cap_f, *self.size, device=self.device)
        n_, ln = ctypes.c_int(), ctypes.c_int()
        check(lib.mdqe_tracker_get_result(self._h, int(bool(is_last_clip)), self.sum_logits.data_ptr(), self.cnt.data_ptr(), self.hw,
                                          out_m.data_ptr(), self.carry.data_ptr(), out_c.ctypes.data, ctypes.byref(n_), ctypes.byref(ln),
                                          cur_stream()), "tracker_get_result")
        ln = ln.value
        if ln != cap_f:                      # last window: only the frames seen; rows were written ln frames apart
            out_m = out_m.view(-1)[:n * ln * self.hw].view(n, ln, *self.size)
        return torch.from_numpy(out_c[:n].copy()), out_m
