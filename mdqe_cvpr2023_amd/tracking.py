"""Cross-clip tracker: same decisions as the reference's OverTracker
(mdqe/tracking/OverTracker.py:10-242), state kept on the device.

The memory bank [clips, instances, frames, h, w] is allocated once and re-zeroed per window; the
Hungarian assignment stays on the host (n <= 120) exactly like the reference (scipy, :159).
INTERIM-TORCH: the soft-IoU / averaging reductions use torch device ops; they are HBM-bound
reductions scheduled to become HIP kernels."""
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
        self.logits = z(self.num_clips, self.max_inst, self.mem_len, *self.size)
        self.valid = z(self.num_clips, self.max_inst, self.mem_len, dt=torch.bool)
        self.cls = z(self.num_clips, self.max_inst, self.K)
        self.embeds = z(self.num_clips, self.max_inst, self.E)
        self._init_memory(True)
        self.n_long = 15 // clip_stride
        self.n_short = max(num_frames, 5) // clip_stride
        self.w_mem = torch.exp(torch.arange(self.n_long, device=device) * 0.25)
        self.untracked = z(self.max_inst)
        self.embed_mem = z(self.max_inst, self.E)

    def _init_memory(self, first=False):
        self.num_clip = 0 if first else 1
        self.start_frame = 0 if first else self.start_frame + self.win
        self.saved_idx.difference_update(range(self.start_frame))
        if not first:
            self.logits.zero_(); self.valid.zero_(); self.cls.zero_(); self.embeds.zero_()
        self.frame_idx = range(self.start_frame, self.start_frame + self.mem_len)

    def _update_memory(self, n_clip, r_idx, c_idx, clip):
        fi = clip.frame_idx
        s0 = max(min(fi) - self.start_frame, 0)
        s1 = max(fi) - self.start_frame
        a, b = fi.index(self.frame_idx[s0]), fi.index(self.frame_idx[s1])
        self.logits[n_clip, r_idx, s0:s1 + 1] = clip.mask_logits[c_idx, a:b + 1].float()
        self.valid[n_clip, r_idx, s0:s1 + 1] = True
        self.cls[n_clip, r_idx] = clip.cls_probs[c_idx]
        self.embeds[n_clip, r_idx] = clip.query_embeds[c_idx].float()
        self.untracked += 1
        self.untracked[r_idx] = 0
        if n_clip > 0:
            st = max(n_clip - 2, 0)
            qm = self.embeds[st:n_clip + 1][:, r_idx]
            w = self.w_mem[:qm.shape[0]].reshape(-1, 1, 1)
            vm = (qm != 0).any(-1)[..., None]
            self.embed_mem[r_idx] = (qm * w).sum(0) / (vm * w).sum(0).clamp(min=1)
        else:
            self.embed_mem[r_idx] = clip.query_embeds[c_idx].float()

    @staticmethod
    def _siou(saved, inp):
        """hard-mask IoU over overlapping frames, OverTracker.py:92-113 (same arithmetic, GEMM form:
        |A & B| = A.B^T on 0/1 rows, |A | B| = |A| + |B| - |A & B|)."""
        i = inp.flatten(1).gt(0.5).float()
        s = saved.flatten(1).gt(0.5).float()
        inter = s @ i.t()
        si, ii = s.sum(1), i.sum(1)
        v = (si[:, None] > 0) & (ii[None] > 0)
        union = si[:, None] + ii[None] - inter
        return torch.where(v, inter / (union + 1e-6), torch.zeros_like(inter))

    def update(self, clip: Clips):
        n_in = clip.num_instance
        siou = sm = None
        if self.num_inst == 0:
            mid = midx = list(range(n_in))
            self.num_inst += n_in
            siou = torch.zeros(0, n_in, device=self.device)
            sm = torch.zeros(0, n_in, device=self.device)
        else:
            qm = self.embed_mem[:self.num_inst]
            lo = (self.untracked[:self.num_inst] < self.n_long).nonzero().reshape(-1)
            sh = (self.untracked[:self.num_inst] < self.n_short).nonzero().reshape(-1)
            sm = torch.zeros(self.num_inst, n_in, device=self.device)
            sm[lo] = ctt_similarity(qm[lo], clip.query_embeds)
            sm[sh] = 0.5 * (sm[sh] + ctt_similarity(qm[sh], clip.query_embeds))
            ii, si_ = [], []
            for o, f in enumerate(clip.frame_idx):
                if f in self.saved_idx and f >= self.start_frame:
                    ii.append(o)
                    si_.append(self.frame_idx.index(f))
            siou = torch.zeros(self.num_inst, n_in, device=self.device)
            if len(si_) > 0:
                im = clip.mask_logits[:, ii].float()
                s = self.logits[:self.num_clip, :self.num_inst][:, :, si_]
                sv = self.valid[:self.num_clip, :self.num_inst].any(-1)
                s = s.sum(0) / sv.sum(0).clamp(min=1).reshape(-1, 1, 1, 1)
                siou = self._siou(s.sigmoid(), im.sigmoid())
            scores = siou + sm
            above = scores > 0.6
            scores = scores * above.float()
            r, c = linear_sum_assignment(scores.cpu().numpy(), maximize=True)       # host sync, as the reference (:159)
            above_c = above.cpu().numpy()
            mid, midx = [], []
            for ri, ci in zip(r, c):
                if not above_c[ri, ci]:
                    continue
                midx.append(int(ci))
                mid.append(int(ri))
            if mid:
                siou[mid, midx] = -1
                sm[mid, midx] = 0
        un = [i for i in range(n_in) if i not in midx]
        rep = []
        if un and siou.shape[0] > 0:
            ms = siou[:, un].max(0)[0].cpu()
            mc = sm[:, un].max(0)[0].cpu()
            rep = [i for j, i in enumerate(un) if ms[j] > 0.4 or mc[j] > 0.6]
        sc = clip.scores.cpu()
        un = [i for i in range(n_in) if i not in midx + rep and sc[i] > 2 * self.thr]
        new = list(range(self.num_inst, self.num_inst + len(un)))
        mid, midx = list(mid) + new, list(midx) + un
        self._update_memory(self.num_clip, mid, midx, clip)
        self.saved_idx.update(clip.frame_set)
        self.num_clip += 1
        self.num_inst += len(new)

    def get_result(self, is_last_clip=False):
        lg = self.logits[:self.num_clip, :self.num_inst]
        va = self.valid[:self.num_clip, :self.num_inst]
        cl = self.cls[:self.num_clip, :self.num_inst]
        qe = self.embeds[:self.num_clip, :self.num_inst]
        lg = lg.sum(0) / va.sum(0).clamp(min=1)[..., None, None]
        nv = max(self.saved_idx) - self.start_frame + 1
        ln = self.win if not is_last_clip else int(nv)
        out_m = lg[:, :ln]
        vc = va.any(-1)[..., None]
        out_c = (cl * vc).sum(0) / vc.sum(0).clamp(min=1)
        nc = min(max(3, (self.T - 1) // self.stride), self.num_clip)
        qw = vc[-nc:] * self.w_mem[:nc].reshape(-1, 1, 1)
        oq = (qe[-nc:] * qw).sum(0) / qw.sum(0).clamp(min=1)
        if not is_last_clip:
            n = self.num_inst
            carry_v = va[:, :n, self.win:].any(0)
            carry_l = lg[:n, self.win:].clone()
            self._init_memory(False)
            self.logits[0, :n, :self.mem_len - self.win] = carry_l
            self.valid[0, :n, :self.mem_len - self.win] = carry_v
            self.cls[0, :n] = out_c
            self.embeds[0, :n] = oq
        return out_c, out_m
