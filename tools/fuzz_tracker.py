"""Differential fuzz of the native tracker core (csrc/tracker_native.hip, with the torch stand-in bank of tests/_standins.py -- no GPU
needed) against the oracle's restatement of OverTracker: random sequences of clips with persistent, appearing, vanishing and duplicated
objects, random windows / clip lengths.  python tools/fuzz_tracker.py [n_sequences]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import mdqe_oracle as O
from _standins import Clips, TorchBankTracker


def sequence(seed):
    g = torch.Generator().manual_seed(seed)
    T = int(torch.randint(2, 5, (1,), generator=g)); win = int(torch.randint(T, 9, (1,), generator=g)); E, K = 16, 4
    hw = (6, 8)
    L = int(torch.randint(win + 1, 3 * win + 2, (1,), generator=g))
    n_obj = int(torch.randint(2, 7, (1,), generator=g))
    proto = torch.randn(n_obj, E, generator=g) * 3
    life = [(int(torch.randint(0, L // 2 + 1, (1,), generator=g)), int(torch.randint(L // 2, L + 1, (1,), generator=g))) for _ in range(n_obj)]
    cells = [(int(torch.randint(0, hw[0] - 1, (1,), generator=g)), int(torch.randint(0, hw[1] - 1, (1,), generator=g))) for _ in range(n_obj)]
    clips = []
    for s in range(0, L):
        e, last = s + T, False
        if e > L:
            e, last = L, True
        objs = [o for o in range(n_obj) if life[o][0] <= s < life[o][1]]
        if float(torch.rand(1, generator=g)) < 0.25 and objs:            # a duplicate detection of one object
            objs = objs + [objs[0]]
        n = len(objs)
        emb = proto[objs] + 0.1 * torch.randn(n, E, generator=g) if n else torch.zeros(0, E)
        masks = torch.full((n, e - s) + hw, -2.0)
        for i, o in enumerate(objs):
            y, x = cells[o]
            masks[i, :, y:y + 2, x:x + 2] = 2.0 + 0.1 * float(torch.rand(1, generator=g))
        cls = torch.rand(n, K, generator=g) * 0.2
        for i, o in enumerate(objs):
            cls[i, o % K] = 0.5 + 0.4 * float(torch.rand(1, generator=g))
        sc = cls.max(-1)[0] if n else torch.zeros(0)
        clips.append((s, e, last, {"scores": sc, "pred_classes": cls.argmax(-1) if n else torch.zeros(0, dtype=torch.long), "cls_probs": cls,
                                   "query_embeds": emb, "pred_masks": masks}))
        if last:
            break
    return T, win, E, K, hw, clips


def run(seed, gpu=False, many=False):
    """gpu: the product tracker (HIP bank kernels) instead of the torch stand-in; many: the clips between two flushes go through
    update_many (one native call)."""
    T, win, E, K, hw, clips = sequence(seed)
    hp = O.Hyper(hidden_dim=E, num_classes=K, n_frames_test=T, n_frames_window_test=win, n_max_inst=24, apply_cls_thres=0.1)
    ref = O.Tracker(hp, hw)
    if gpu:
        from mdqe_cvpr2023_amd.tracking import OverTracker
        trk = OverTracker(24, T, win, 1, K, 4, E, hw, torch.device("cuda"), 0.1)
    else:
        trk = TorchBankTracker(24, T, win, 1, K, 4, E, hw, torch.device("cpu"), 0.1)
    saved, run_ = 0, []
    for s, e, last, r in clips:
        c = dict(r); c["frame_idx"] = list(range(s, e))
        ref.update(c)
        rr = dict(r, pred_masks=r["pred_masks"].cuda()) if gpu else r
        flush = last or (s + 1 >= win * (saved + 1))
        if gpu and many:
            run_.append(Clips(range(s, e), rr))
            if flush:
                trk.update_many(run_)
                run_ = []
        else:
            trk.update(Clips(range(s, e), rr))
        if (flush or not (gpu and many)) and trk.num_inst != ref.num_inst:
            return "num_inst %d vs %d after clip %d" % (trk.num_inst, ref.num_inst, s)
        if flush:
            c0, m0 = ref.get_result(last)
            c1, m1 = trk.get_result(last)
            m1 = m1.cpu()
            if m0.shape != m1.shape or float((m0 - m1).abs().max() if m0.numel() else 0) > 1e-5 or float((c0 - c1).abs().max() if c0.numel() else 0) > 1e-6:
                return "window %d differs" % saved
            saved += 1
    return None


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    gpu = "--gpu" in sys.argv
    bad = 0
    for seed in range(n):
        try:
            r = run(seed, gpu=gpu, many=gpu and seed % 2 == 1)
        except Exception as ex:                                      # both sides raise on the same overflow conditions or neither
            r = "exception %r" % (ex,)
        if r:
            bad += 1
            print("seed %d: %s" % (seed, r), flush=True)
    print("tracker fuzz: %d sequences, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)
