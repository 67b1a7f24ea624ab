"""Cross-clip tracker: the decisions of the reference's OverTracker
(mdqe/tracking/OverTracker.py:10-242) on a state split between device and host.

Reference state: saved_logits [clips, instances, frames, h, w] (8 GB at R50_ovis_360, re-zeroed per
window) that is only ever consumed through sum-over-clips, plus ~70 tiny device ops and several
host syncs per clip.  Here:
  * device: a running per-(instance, frame) SUM of mask logits + a count (250 MB), touched by two
    fused HIP kernels per update (csrc/tracker.hip): sign-intersection counts for the hard-mask IoU
    and an indexed accumulate;
  * host (numpy fp32): everything that is O(instances x 256): embeddings, class probabilities,
    presence flags, untracked counters, the bi-softmax similarity, the Hungarian assignment (scipy,
    exactly like the reference :159) and the ID bookkeeping;
  * one small device->host copy per update (the [n_saved, n_in, 3] count tensor).
A tracker built on a CPU device (gloo tests) takes the torch route for the two device steps.
"""
import ctypes

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def ctt_similarity(saved, inp):
    """bi-softmax similarity, OverTracker.py:228-242 (numpy fp32)."""
    f = saved @ inp.T
    Ns, Ni = f.shape
    if Ns == 0 or Ni == 0:
        return f
    Ws, Wi = (1 if Ns > 1 else 0), (1 if Ni > 1 else 0)
    d2t, t2d = _softmax(f, 0), _softmax(f, 1)
    if Ns == 1 and Ni == 1:
        return (0.5 * (d2t + t2d)).astype(np.float32)
    return ((Ws * d2t + Wi * t2d) / max(Ws + Wi, 1)).astype(np.float32)


class Clips:
    """OverTracker.py:245-256: per-clip detections.  mask logits stay on the device; the small per-instance
    vectors are taken as host arrays (`results["host"]`, one batched copy made by the engine) when present."""

    def __init__(self, frame_idx, results):
        self.frame_idx = list(frame_idx)
        self.frame_set = set(frame_idx)
        self.mask_logits = results["pred_masks"]
        h = results.get("host")
        if h is None:
            h = {k: results[k].detach().float().cpu().numpy() for k in ("scores", "cls_probs", "query_embeds")}
        self.scores = np.asarray(h["scores"], dtype=np.float32)
        self.cls_probs = np.asarray(h["cls_probs"], dtype=np.float32)
        self.query_embeds = np.asarray(h["query_embeds"], dtype=np.float32)
        self.num_instance = int(self.scores.shape[0])


class OverTracker:
    def __init__(self, num_max_inst, num_frames, window_frames, clip_stride, num_classes, mask_dim, embed_dim,
                 image_size, device, apply_cls_thres):
        self.T, self.win, self.stride = num_frames, window_frames, clip_stride
        self.K, self.E = num_classes, embed_dim
        self.size = tuple(image_size)
        self.hw = int(self.size[0] * self.size[1])
        self.device = torch.device(device)
        self.thr = apply_cls_thres
        self.max_inst = num_max_inst
        self.num_inst = 0
        self.mem_len = window_frames + num_frames
        self.num_clips = window_frames // clip_stride + 2
        self.saved_idx = set()
        self.start_frame = 0
        self.sum_logits = torch.zeros(self.max_inst, self.mem_len, *self.size, device=self.device)
        self.cnt = torch.zeros(self.max_inst, self.mem_len, device=self.device)
        self.cnt_h = np.zeros((self.max_inst, self.mem_len), dtype=np.float32)       # host mirror of cnt
        self.clip_valid = np.zeros((self.num_clips, self.max_inst), dtype=bool)
        self.cls = np.zeros((self.num_clips, self.max_inst, self.K), dtype=np.float32)
        self.embeds = np.zeros((self.num_clips, self.max_inst, self.E), dtype=np.float32)
        self._init_memory(True)
        self.n_long = 15 // clip_stride
        self.n_short = max(num_frames, 5) // clip_stride
        self.w_mem = np.exp(np.arange(self.n_long, dtype=np.float32) * np.float32(0.25)).astype(np.float32)
        self.untracked = np.zeros(self.max_inst, dtype=np.float32)
        self.embed_mem = np.zeros((self.max_inst, self.E), dtype=np.float32)

    def _init_memory(self, first=False):
        self.num_clip = 0 if first else 1
        self.start_frame = 0 if first else self.start_frame + self.win
        self.saved_idx.difference_update(range(self.start_frame))
        if not first:
            self.sum_logits.zero_(); self.cnt.zero_()
            self.cnt_h[:] = 0; self.clip_valid[:] = False; self.cls[:] = 0; self.embeds[:] = 0
        self.frame_idx = range(self.start_frame, self.start_frame + self.mem_len)

    # ---- device steps ----------------------------------------------------------------------------
    def _siou_counts(self, ni, s0, clip, a, nf):
        """[ni, n_in, 3] = (|A&B|, |A|, |B|) over frames s0..s0+nf-1 of the bank vs frames a..a+nf-1 of the clip."""
        n_in = clip.num_instance
        m = clip.mask_logits
        if self.device.type == "cuda":
            from ._lib import check, cur_stream, lib
            if not (m.is_contiguous() and m.dtype == torch.float32):
                m = m.float().contiguous()
            out = torch.empty(ni, n_in, 3, device=self.device)
            check(lib.mdqe_trk_siou_f32(ctypes.c_void_p(self.sum_logits.data_ptr() + 4 * s0 * self.hw), self.mem_len * self.hw, ni,
                                        ctypes.c_void_p(m.data_ptr() + 4 * a * self.hw), m.shape[1] * self.hw, n_in, nf * self.hw,
                                        ctypes.c_void_p(out.data_ptr()), cur_stream()), "trk_siou")
            return out.cpu().numpy()                                        # the one host sync of an update
        A = (self.sum_logits[:ni, s0:s0 + nf] > 0).flatten(1).float()        # CPU-device route (tests)
        B = (m[:, a:a + nf].float() > 0).flatten(1).float()
        inter = A @ B.t()
        return torch.stack([inter, A.sum(1)[:, None].expand_as(inter), B.sum(1)[None].expand_as(inter)], -1).numpy()

    def _accumulate(self, r_idx, c_idx, s0, clip, a, nf):
        m = clip.mask_logits
        if self.device.type == "cuda":
            from ._lib import check, cur_stream, lib
            if not (m.is_contiguous() and m.dtype == torch.float32):
                m = m.float().contiguous()
            n = len(r_idx)
            arr = lambda v: (ctypes.c_int * n)(*[int(x) for x in v])
            check(lib.mdqe_trk_accumulate_f32(ctypes.c_void_p(self.sum_logits.data_ptr() + 4 * s0 * self.hw), self.mem_len * self.hw,
                                              ctypes.c_void_p(self.cnt.data_ptr() + 4 * s0), self.mem_len,
                                              ctypes.c_void_p(m.data_ptr() + 4 * a * self.hw), m.shape[1] * self.hw, nf * self.hw, nf,
                                              arr(r_idx), arr(c_idx), n, cur_stream()), "trk_accumulate")
            return
        r = torch.as_tensor(r_idx, dtype=torch.long)
        c = torch.as_tensor(c_idx, dtype=torch.long)
        self.sum_logits[r, s0:s0 + nf] += m[c, a:a + nf].float()
        self.cnt[r, s0:s0 + nf] += 1

    # ---- OverTracker._update_memory (:65-90) -----------------------------------------------------
    def _update_memory(self, n_clip, r_idx, c_idx, clip):
        if n_clip >= self.num_clips or (len(r_idx) and max(r_idx) >= self.max_inst):
            raise IndexError("tracker memory exceeded (MAX_NUM_INSTANCES / clip slots), as the reference would")
        fi = clip.frame_idx
        s0 = max(min(fi) - self.start_frame, 0)
        s1 = max(fi) - self.start_frame
        a, b = fi.index(self.frame_idx[s0]), fi.index(self.frame_idx[s1])
        self.untracked += 1
        if not len(r_idx):
            return
        self._accumulate(r_idx, c_idx, s0, clip, a, b - a + 1)
        self.cnt_h[r_idx, s0:s1 + 1] += 1
        self.clip_valid[n_clip, r_idx] = True
        self.cls[n_clip, r_idx] = clip.cls_probs[c_idx]
        self.embeds[n_clip, r_idx] = clip.query_embeds[c_idx]
        self.untracked[r_idx] = 0
        if n_clip > 0:
            st = max(n_clip - 2, 0)
            qm = self.embeds[st:n_clip + 1][:, r_idx]
            w = self.w_mem[:qm.shape[0]].reshape(-1, 1, 1)
            vm = (qm != 0).any(-1)[..., None]
            self.embed_mem[r_idx] = (qm * w).sum(0) / np.maximum((vm * w).sum(0), 1)
        else:
            self.embed_mem[r_idx] = clip.query_embeds[c_idx]

    # ---- OverTracker.update (:115-193) ------------------------------------------------------------
    def update(self, clip: Clips):
        n_in = clip.num_instance
        if self.num_inst == 0:
            mid, midx = list(range(n_in)), list(range(n_in))
            self.num_inst += n_in
            siou = sm = np.zeros((0, n_in), dtype=np.float32)
        else:
            ni = self.num_inst
            qm = self.embed_mem[:ni]
            lo = np.nonzero(self.untracked[:ni] < self.n_long)[0]
            sh = np.nonzero(self.untracked[:ni] < self.n_short)[0]
            sm = np.zeros((ni, n_in), dtype=np.float32)
            sm[lo] = ctt_similarity(qm[lo], clip.query_embeds)
            if not (len(sh) == len(lo) and self.n_short <= self.n_long):
                sm[sh] = 0.5 * (sm[sh] + ctt_similarity(qm[sh], clip.query_embeds))
            # (else: sh is the same index set as lo -- the usual case, every track seen recently -- and
            #  0.5 * (a + a) == a bit for bit, so the second similarity is skipped)
            ii, si_ = [], []
            for o, f in enumerate(clip.frame_idx):
                if f in self.saved_idx and f >= self.start_frame:
                    ii.append(o)
                    si_.append(self.frame_idx.index(f))
            siou = np.zeros((ni, n_in), dtype=np.float32)
            if len(si_) > 0 and n_in > 0:
                if si_ != list(range(si_[0], si_[-1] + 1)) or ii != list(range(ii[0], ii[-1] + 1)):
                    raise RuntimeError("tracker: overlapping frames must be contiguous (clip_stride <= clip length)")
                c3 = self._siou_counts(ni, si_[0], clip, ii[0], len(si_))
                inter, sa, ia = c3[..., 0], c3[..., 1], c3[..., 2]
                v = (sa > 0) & (ia > 0)
                siou = np.where(v, inter / (sa + ia - inter + np.float32(1e-6)), np.float32(0)).astype(np.float32)
            scores = siou + sm
            above = scores > 0.6
            scores = scores * above
            r, c = linear_sum_assignment(scores, maximize=True)
            mid, midx = [], []
            for ri, ci in zip(r, c):
                if not above[ri, ci]:
                    continue
                midx.append(int(ci))
                mid.append(int(ri))
                siou[ri, ci] = -1
                sm[ri, ci] = 0
        un = [i for i in range(n_in) if i not in midx]
        rep = []
        if siou.shape[0] > 0:
            rep = [i for i in un if siou[:, i].max() > 0.4 or sm[:, i].max() > 0.6]
        un = [i for i in range(n_in) if i not in midx + rep and clip.scores[i] > 2 * self.thr]
        new = list(range(self.num_inst, self.num_inst + len(un)))
        mid, midx = list(mid) + new, list(midx) + un
        self._update_memory(self.num_clip, mid, midx, clip)
        self.saved_idx.update(clip.frame_set)
        self.num_clip += 1
        self.num_inst += len(new)

    # ---- OverTracker.get_result (:195-225) --------------------------------------------------------
    def get_result(self, is_last_clip=False):
        n = self.num_inst
        lg = self.sum_logits[:n] / self.cnt[:n].clamp(min=1)[..., None, None]
        nv = max(self.saved_idx) - self.start_frame + 1
        ln = self.win if not is_last_clip else int(nv)
        out_m = lg[:, :ln]
        vc = self.clip_valid[:self.num_clip, :n][..., None].astype(np.float32)
        cl = self.cls[:self.num_clip, :n]
        qe = self.embeds[:self.num_clip, :n]
        out_c = ((cl * vc).sum(0) / np.maximum(vc.sum(0), 1)).astype(np.float32)
        nc = min(max(3, (self.T - 1) // self.stride), self.num_clip)
        qw = vc[-nc:] * self.w_mem[:nc].reshape(-1, 1, 1)
        oq = ((qe[-nc:] * qw).sum(0) / np.maximum(qw.sum(0), 1)).astype(np.float32)
        if not is_last_clip:
            carry_l = lg[:, self.win:]                       # lg is a fresh tensor, safe across the re-zeroing
            carry_v = self.cnt_h[:n, self.win:] > 0
            self._init_memory(False)
            k = self.mem_len - self.win
            cv = torch.from_numpy(carry_v.astype(np.float32)).to(self.device)
            self.sum_logits[:n, :k] = carry_l * cv[..., None, None]
            self.cnt[:n, :k] = cv
            self.cnt_h[:n, :k] = carry_v
            self.clip_valid[0, :n] = carry_v.any(-1)
            self.cls[0, :n] = out_c
            self.embeds[0, :n] = oq
        return torch.from_numpy(out_c), out_m
