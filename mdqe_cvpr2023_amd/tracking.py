"""Cross-clip tracker: the decisions of the reference's OverTracker
(mdqe/tracking/OverTracker.py:10-242) on a leaner state.

Reference state: saved_logits [clips, instances, frames, h, w] (8 GB at R50_ovis_360, re-zeroed per
window) that is only ever consumed through sum-over-clips.  Here: a running per-(instance, frame)
SUM of logits + a count (250 MB), per-clip class/embedding/presence tables (tiny), and the
host-side bookkeeping (untracked counters, frame sets) on the host.  One device->host transfer per
update (the Hungarian assignment runs on the host with scipy exactly like the reference, :159).
Works on CUDA (HIP GEMM for the hard-mask intersections) and on CPU tensors (gloo tests).
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


def ctt_similarity(saved, inp):
    """bi-softmax similarity, OverTracker.py:228-242."""
    f = saved @ inp.t()
    Ns, Ni = f.shape
    Ws, Wi = (1 if Ns > 1 else 0), (1 if Ni > 1 else 0)
    d2t, t2d = f.softmax(0), f.softmax(1)
    if Ns == 1 and Ni == 1:
        return 0.5 * (d2t + t2d)
    return (Ws * d2t + Wi * t2d) / max(Ws + Wi, 1)


def _nt(a, b):
    """a @ b.T for 0/1 fp32 matrices."""
    if a.is_cuda:
        from . import ops
        return ops.linear(a.contiguous(), b.contiguous())
    return a @ b.t()


class Clips:
    """OverTracker.py:245-256 (dict-backed)."""

    def __init__(self, frame_idx, results):
        self.frame_idx = list(frame_idx)
        self.frame_set = set(frame_idx)
        self.classes = results["pred_classes"]
        self.scores = results["scores"]
        self.cls_probs = results["cls_probs"]
        self.mask_logits = results["pred_masks"]
        self.query_embeds = results["query_embeds"]
        self.num_instance = len(self.scores)


class OverTracker:
    def __init__(self, num_max_inst, num_frames, window_frames, clip_stride, num_classes, mask_dim, embed_dim,
                 image_size, device, apply_cls_thres):
        self.T, self.win, self.stride = num_frames, window_frames, clip_stride
        self.K, self.E = num_classes, embed_dim
        self.size = tuple(image_size)
        self.device = device
        self.thr = apply_cls_thres
        self.max_inst = num_max_inst
        self.num_inst = 0
        self.mem_len = window_frames + num_frames
        self.num_clips = window_frames // clip_stride + 2
        self.saved_idx = set()
        self.start_frame = 0
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        self.sum_logits = z(self.max_inst, self.mem_len, *self.size)       # sum over clips of saved logits
        self.cnt = z(self.max_inst, self.mem_len)                          # number of clips that wrote (inst, frame)
        self.clip_valid = z(self.num_clips, self.max_inst, dt=torch.bool)  # instance present in clip slot
        self.cls = z(self.num_clips, self.max_inst, self.K)
        self.embeds = z(self.num_clips, self.max_inst, self.E)
        self._init_memory(True)
        self.n_long = 15 // clip_stride
        self.n_short = max(num_frames, 5) // clip_stride
        self.w_mem = torch.exp(torch.arange(self.n_long, device=device) * 0.25)
        self.untracked = np.zeros(self.max_inst, dtype=np.float64)         # host-side counters (:45, :77-78)
        self.embed_mem = z(self.max_inst, self.E)

    def _init_memory(self, first=False):
        self.num_clip = 0 if first else 1
        self.start_frame = 0 if first else self.start_frame + self.win
        self.saved_idx.difference_update(range(self.start_frame))
        if not first:
            self.sum_logits.zero_(); self.cnt.zero_(); self.clip_valid.zero_(); self.cls.zero_(); self.embeds.zero_()
        self.frame_idx = range(self.start_frame, self.start_frame + self.mem_len)

    def _update_memory(self, n_clip, r_idx, c_idx, clip):
        """OverTracker.py:65-90."""
        if n_clip >= self.num_clips or (len(r_idx) and max(r_idx) >= self.max_inst):
            raise IndexError("tracker memory exceeded (MAX_NUM_INSTANCES / clip slots), as the reference would")
        fi = clip.frame_idx
        s0 = max(min(fi) - self.start_frame, 0)
        s1 = max(fi) - self.start_frame
        a, b = fi.index(self.frame_idx[s0]), fi.index(self.frame_idx[s1])
        r = torch.as_tensor(r_idx, dtype=torch.long, device=self.device)
        c = torch.as_tensor(c_idx, dtype=torch.long, device=self.device)
        if len(r_idx):
            self.sum_logits[r, s0:s1 + 1] += clip.mask_logits[c, a:b + 1].float()
            self.cnt[r, s0:s1 + 1] += 1
            self.clip_valid[n_clip, r] = True
            self.cls[n_clip, r] = clip.cls_probs[c]
            self.embeds[n_clip, r] = clip.query_embeds[c].float()
        self.untracked += 1
        self.untracked[r_idx] = 0
        if not len(r_idx):
            return
        if n_clip > 0:
            st = max(n_clip - 2, 0)
            qm = self.embeds[st:n_clip + 1][:, r]
            w = self.w_mem[:qm.shape[0]].reshape(-1, 1, 1)
            vm = (qm != 0).any(-1)[..., None]
            self.embed_mem[r] = (qm * w).sum(0) / (vm * w).sum(0).clamp(min=1)
        else:
            self.embed_mem[r] = clip.query_embeds[c].float()

    @staticmethod
    def _siou(saved_logits, inp_logits):
        """hard-mask IoU over the overlapping frames (OverTracker.py:92-113): sigmoid(x) > 0.5 <=> x > 0;
        |A & B| as an NT GEMM of 0/1 rows (exact in fp32), |A | B| = |A| + |B| - |A & B|."""
        i = inp_logits.flatten(1).gt(0).float()
        s = saved_logits.flatten(1).gt(0).float()
        inter = _nt(s, i)
        si, ii = s.sum(1), i.sum(1)
        v = (si[:, None] > 0) & (ii[None] > 0)
        union = si[:, None] + ii[None] - inter
        return torch.where(v, inter / (union + 1e-6), torch.zeros_like(inter))

    def update(self, clip: Clips):
        n_in = clip.num_instance
        if self.num_inst == 0:
            mid = midx = list(range(n_in))
            self.num_inst += n_in
            siou = sm = np.zeros((0, n_in))
            sc = None
        else:
            ni = self.num_inst
            qm = self.embed_mem[:ni]
            lo = np.nonzero(self.untracked[:ni] < self.n_long)[0].tolist()
            sh = np.nonzero(self.untracked[:ni] < self.n_short)[0].tolist()
            sm_d = torch.zeros(ni, n_in, device=self.device)
            if lo:
                sm_d[lo] = ctt_similarity(qm[lo], clip.query_embeds)
            if sh:
                sm_d[sh] = 0.5 * (sm_d[sh] + ctt_similarity(qm[sh], clip.query_embeds))
            ii, si_ = [], []
            for o, f in enumerate(clip.frame_idx):
                if f in self.saved_idx and f >= self.start_frame:
                    ii.append(o)
                    si_.append(self.frame_idx.index(f))
            siou_d = torch.zeros(ni, n_in, device=self.device)
            if len(si_) > 0 and n_in > 0:
                contiguous = si_ == list(range(si_[0], si_[-1] + 1)) and ii == list(range(ii[0], ii[-1] + 1))
                im = (clip.mask_logits[:, ii[0]:ii[-1] + 1] if contiguous else clip.mask_logits[:, ii]).float()
                n_present = self.clip_valid[:self.num_clip, :ni].sum(0).clamp(min=1).reshape(-1, 1, 1, 1)
                s = (self.sum_logits[:ni, si_[0]:si_[-1] + 1] if contiguous else self.sum_logits[:ni][:, si_]) / n_present
                siou_d = self._siou(s, im)
            host = torch.cat([siou_d, sm_d, clip.scores.reshape(1, -1).float()], 0).cpu().numpy()   # the one host sync
            siou, sm, sc = host[:ni].copy(), host[ni:2 * ni].copy(), host[2 * ni]
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
        if sc is None:
            sc = clip.scores.float().cpu().numpy() if n_in else np.zeros(0)
        un = [i for i in range(n_in) if i not in midx]
        rep = []
        if siou.shape[0] > 0:
            rep = [i for i in un if siou[:, i].max() > 0.4 or sm[:, i].max() > 0.6]
        un = [i for i in range(n_in) if i not in midx + rep and sc[i] > 2 * self.thr]
        new = list(range(self.num_inst, self.num_inst + len(un)))
        mid, midx = list(mid) + new, list(midx) + un
        self._update_memory(self.num_clip, mid, midx, clip)
        self.saved_idx.update(clip.frame_set)
        self.num_clip += 1
        self.num_inst += len(new)

    def get_result(self, is_last_clip=False):
        """OverTracker.py:195-225."""
        n = self.num_inst
        lg = self.sum_logits[:n] / self.cnt[:n].clamp(min=1)[..., None, None]
        nv = max(self.saved_idx) - self.start_frame + 1
        ln = self.win if not is_last_clip else int(nv)
        out_m = lg[:, :ln]
        vc = self.clip_valid[:self.num_clip, :n][..., None]
        cl = self.cls[:self.num_clip, :n]
        qe = self.embeds[:self.num_clip, :n]
        out_c = (cl * vc).sum(0) / vc.sum(0).clamp(min=1)
        nc = min(max(3, (self.T - 1) // self.stride), self.num_clip)
        qw = vc[-nc:] * self.w_mem[:nc].reshape(-1, 1, 1)
        oq = (qe[-nc:] * qw).sum(0) / qw.sum(0).clamp(min=1)
        if not is_last_clip:
            carry_l = lg[:, self.win:]                       # lg is a fresh tensor, safe across the re-zeroing
            carry_v = (self.cnt[:n, self.win:] > 0)
            self._init_memory(False)
            k = self.mem_len - self.win
            self.sum_logits[:n, :k] = carry_l * carry_v[..., None, None]
            self.cnt[:n, :k] = carry_v.float()
            self.clip_valid[0, :n] = carry_v.any(-1)
            self.cls[0, :n] = out_c
            self.embeds[0, :n] = oq
        return out_c, out_m
