"""Differential fuzz of the whole driver (MDQE.inference_vis on the HIP path) against the oracle at small sizes: random frame sizes (odd,
padded on every level), video lengths (shorter than a clip .. several windows), clip length / stride / tracker window, frame-pass size,
look-ahead depth, output size, uint8 / float frames.  python tools/fuzz_pipeline.py [n_cases] [first]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import mdqe_oracle as O
from mdqe_cvpr2023_amd.config import MDQEConfig
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for it in range(first, n_cases):
    g = torch.Generator().manual_seed(500 + it)
    ri = lambda a, b: int(torch.randint(a, b + 1, (1,), generator=g))
    T = ri(2, 4); stride = 1 if it % 3 else ri(1, 2); win = ri(T, 7); L = ri(1, 15)
    h, w = ri(33, 100), ri(40, 130)
    kw = dict(enc_layers=1, dec_layers=ri(1, 2), n_frames=T, num_classes=ri(3, 9), num_queries=[16, 25, 36][it % 3], query_embed_dim=16)
    ev = dict(n_frames_test=T, n_frames_window_test=win, n_max_inst=60, apply_cls_thres=[0.08, 0.12, 0.2][it % 3], clip_stride=stride)
    cfg = MDQEConfig(**kw, **ev)
    sd = random_state(cfg, seed=100 + it)
    base = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8).float()
    frames = [(0.8 * base + 0.2 * torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8).float()).round() for _ in range(L)]
    frames = [f.to(torch.uint8) if it % 4 else f for f in frames]
    out_size = (h, w) if it % 2 else (ri(30, 150), ri(30, 150))
    model = MDQE(cfg, state_dict=sd).eval()
    model.frame_batch = [0, 2, 5, 9][it % 4]
    model.lookahead = 1 + it % 3
    model.merge_on_cpu = bool(it % 2)
    model.early_masks = bool((it // 2) % 2)
    trace, ref_trace = [], []
    try:
        with torch.no_grad():
            out = model.inference_vis([{"image": frames, "height": out_size[0], "width": out_size[1]}], trace=trace)
            ref = O.inference_vis(sd, O.Hyper(**kw, **ev), frames, lambda im: O.resnet(sd, "detr.backbone.0.backbone", im, 50), out_size=out_size,
                                  trace=ref_trace)
        msg = None
        if len(trace) != len(ref_trace):
            msg = "clip count %d vs %d" % (len(trace), len(ref_trace))
        else:
            for ci, (a, b) in enumerate(zip(trace, ref_trace)):
                if a["pred_masks"].shape != b["pred_masks"].shape:
                    msg = "clip %d: %d vs %d instances" % (ci, a["pred_masks"].shape[0], b["pred_masks"].shape[0]); break
                if b["pred_masks"].numel() and float((a["pred_masks"].cpu() - b["pred_masks"]).abs().max()) > 1e-3 * max(1.0, float(b["pred_masks"].abs().max())):
                    msg = "clip %d: mask logits differ by %.2e" % (ci, float((a["pred_masks"].cpu() - b["pred_masks"]).abs().max())); break
        if msg is None and out["pred_labels"] != ref["pred_labels"]:
            msg = "labels %s vs %s" % (out["pred_labels"][:8], ref["pred_labels"][:8])
        if msg is None and len(out["pred_masks"]):
            got, want = torch.stack(out["pred_masks"]), torch.stack(ref["pred_masks"])
            if got.shape != want.shape or float((got != want).float().mean()) > 2e-3:
                msg = "final masks differ (%s vs %s)" % (tuple(got.shape), tuple(want.shape))
    except Exception as ex:
        msg = "exception %r" % (ex,)
    if msg:
        bad += 1
        print("case %d (L=%d T=%d stride=%d win=%d %dx%d -> %s, fb=%d la=%d): %s" % (it, L, T, stride, win, h, w, out_size, model.frame_batch, model.lookahead, msg), flush=True)
print("pipeline fuzz: %d cases, %d mismatches" % (n_cases - first, bad))
