"""Test-only stand-ins (never imported by the product).

TorchBankTracker: the product's native tracker HOST core (mdqe_tracker_overlap / decide / result in
csrc/tracker_native.hip, through mdqe_cvpr2023_amd.tracking.TrackerCore) driven with a torch CPU stand-in for the
device bank, so that the decisions and the bookkeeping can be held to the reference's recorded tracker sequence and
exercised by the gloo sharding tests in a container without a GPU.  The product's OverTracker pairs the same core with the
HIP kernels of csrc/tracker.hip and has no such path."""
import torch

from mdqe_cvpr2023_amd.tracking import Clips, TrackerCore  # noqa: F401


class TorchBankTracker(TrackerCore):
    def __init__(self, num_max_inst, num_frames, window_frames, clip_stride, num_classes, mask_dim, embed_dim, image_size,
                 device, apply_cls_thres):
        super().__init__(num_max_inst, num_frames, window_frames, clip_stride, num_classes, embed_dim, apply_cls_thres)
        self.size = tuple(int(v) for v in image_size)
        self.sum_logits = torch.zeros(self.max_inst, self.mem_len, *self.size)
        self.cnt = torch.zeros(self.max_inst, self.mem_len)

    def update(self, clip):
        ni, s0, a, nf = self.overlap(clip)
        m = clip.mask_logits.float()
        c3 = None
        if ni > 0 and nf > 0 and clip.num_instance > 0:
            A = (self.sum_logits[:ni, s0:s0 + nf] > 0).flatten(1).float()
            B = (m[:, a:a + nf] > 0).flatten(1).float()
            inter = A @ B.t()
            c3 = torch.stack([inter, A.sum(1)[:, None].expand_as(inter), B.sum(1)[None].expand_as(inter)], -1).numpy()
        r, c, s0, a, nf = self.decide(clip, c3)
        if len(r):
            r, c = torch.as_tensor(r, dtype=torch.long), torch.as_tensor(c, dtype=torch.long)
            self.sum_logits[r, s0:s0 + nf] += m[c, a:a + nf]
            self.cnt[r, s0:s0 + nf] += 1

    def get_result(self, is_last_clip=False):
        n = self.num_inst
        lg = self.sum_logits[:n] / self.cnt[:n].clamp(min=1)[..., None, None]
        out_c, n_, ln, cv = self.result_host(is_last_clip)
        assert n_ == n
        out_m = lg[:, :ln].clone()
        if not is_last_clip:
            k = self.mem_len - self.win
            carry = lg[:, self.win:].clone()
            self.sum_logits.zero_(); self.cnt.zero_()
            cvf = torch.from_numpy(cv.astype("float32"))
            self.sum_logits[:n, :k] = carry * cvf[..., None, None]
            self.cnt[:n, :k] = cvf
        return torch.from_numpy(out_c), out_m
