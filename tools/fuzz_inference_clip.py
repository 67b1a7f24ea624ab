"""Differential fuzz of the batched inference_clip (clip_ops.hip) against the oracle's per-clip restatement: random batches with
random thresholds, duplicated embeddings, copied masks, blank rows and score ties.  Reports every clip whose decisions differ.
python tools/fuzz_inference_clip.py [n_batches]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mdqe_oracle as O
from mdqe_cvpr2023_amd.config import MDQEConfig
from mdqe_cvpr2023_amd.engine import Engine

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = bad_tie = bad_swap = checked = 0
for it in range(first, n_batches):
    g = torch.Generator().manual_seed(1000 + it)
    B = int(torch.randint(1, 9, (1,), generator=g)); Q = [16, 36, 100, 196][it % 4]; K = [5, 25][it % 2]; C = [32, 64, 256][it % 3]
    M = [8, 24, 32][it % 3]; T = [1, 2, 4, 5][it % 4]; Hm, Wm = [(8, 12), (16, 24), (24, 40)][it % 3]
    thr = float(torch.rand(1, generator=g) * 0.5 + 0.05)
    cls = torch.rand(B, Q, K, generator=g) ** 3
    emb = torch.randn(B, Q, C, generator=g)
    coef = torch.tanh(torch.randn(B, Q, M, generator=g) * 1.5)
    mf = torch.relu(torch.randn(B + T - 1, Hm, Wm, M, generator=g) - 0.3)
    tied = set()
    for b in range(B):                                    # structure: duplicates, copied masks, blanks, exact score ties
        for _ in range(int(torch.randint(0, 4, (1,), generator=g))):
            i, j = torch.randint(0, Q, (2,), generator=g).tolist()
            emb[b, j] = emb[b, i] * float(torch.rand(1, generator=g) + 0.5) + 1e-3 * torch.randn(C, generator=g)
        for _ in range(int(torch.randint(0, 3, (1,), generator=g))):
            i, j = torch.randint(0, Q, (2,), generator=g).tolist()
            coef[b, j] = coef[b, i]
        for _ in range(int(torch.randint(0, 3, (1,), generator=g))):
            coef[b, int(torch.randint(0, Q, (1,), generator=g))] = -coef[b, 0].abs()
        if it % 5 == 0:
            i, j = torch.randint(0, Q, (2,), generator=g).tolist()
            cls[b, j] = cls[b, i]                                               # an exact tie of two queries' scores: the reference's
            tied.add(b)                                                         # `sort(descending=True)` (mdqe.py:373) is unstable, the order is open
    cfg = MDQEConfig(backbone="custom", hidden_dim=C, num_classes=K, num_queries=Q, n_frames=T, n_frames_test=T, apply_cls_thres=thr,
                     detections_per_image=[2, 15][it % 2])
    eng = Engine.__new__(Engine); eng.cfg, eng.dev = cfg, torch.device("cuda")
    res = eng.inference_clips({"cls": cls.cuda(), "mask_coeff": coef.cuda(), "query_embed": emb.cuda()}, mf.cuda(), list(range(B)), T)
    hp = O.Hyper(hidden_dim=C, num_classes=K, n_frames=T, n_frames_test=T, apply_cls_thres=thr, detections_per_image=cfg.detections_per_image)
    for b in range(B):
        ref = O.inference_clip(hp, {"cls": cls[b:b + 1], "mask_coeff": coef[b:b + 1], "query_embed": emb[b:b + 1]}, mf[b:b + T].permute(3, 0, 1, 2))
        r = res[b]
        checked += 1
        ok = r["pred_masks"].shape == ref["pred_masks"].shape and r["pred_classes"].tolist() == ref["pred_classes"].tolist()
        ok = ok and float((r["pred_masks"].cpu() - ref["pred_masks"]).abs().max() if ref["pred_masks"].numel() else 0.0) < 1e-4
        ok = ok and float((r["scores"].cpu() - ref["scores"]).abs().max() if ref["scores"].numel() else 0.0) < 1e-5
        swap = False
        if not ok and r["pred_masks"].shape == ref["pred_masks"].shape and ref["scores"].numel():
            # the same SET of instances in another order?  (two final scores closer than the fp32 noise of the rescoring product)
            gm, gs, gc = r["pred_masks"].cpu().flatten(1), r["scores"].cpu(), r["pred_classes"].cpu()
            used, swap = set(), True
            for k in range(ref["scores"].numel()):
                d = (gm - ref["pred_masks"][k].flatten()[None]).abs().amax(1)
                cand = [j for j in range(len(d)) if j not in used and float(d[j]) < 1e-4 and int(gc[j]) == int(ref["pred_classes"][k])
                        and abs(float(gs[j]) - float(ref["scores"][k])) < 1e-5]
                if not cand:
                    swap = False
                    break
                used.add(cand[0])
        if not ok:
            bad += 1
            bad_tie += b in tied
            bad_swap += swap and b not in tied
            print("MISMATCH%s batch %d clip %d (Q=%d K=%d C=%d M=%d T=%d thr=%.3f): got %d instances %s / oracle %d %s" % (
                " (injected score tie)" if b in tied else (" (same instances, order of near-equal scores)" if swap else ""), it, b, Q, K, C, M, T, thr, r["scores"].numel(), [round(float(v), 4) for v in r["scores"].cpu()][:6],
                ref["scores"].numel(), [round(float(v), 4) for v in ref["scores"]][:6]), flush=True)
print("fuzz: %d clips checked, %d mismatches: %d in clips with an injected exact score tie (order left open by the reference's unstable "
      "sort; torch.sort(stable=True) and the kernel both break ties by query index), %d with the same instances in another order of "
      "near-equal final scores, %d other" % (checked, bad, bad_tie, bad_swap, bad - bad_tie - bad_swap))
sys.exit(1 if bad != bad_tie + bad_swap else 0)
