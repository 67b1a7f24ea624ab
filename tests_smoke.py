"""Body of __graft_entry__.smoke(): one small invocation of the whole hot path on cuda:0 (R50 backbone, 2+2
transformer layers, 5 frames of 64x96, 3-frame clips, 2 tracker windows) checked against the CPU oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run_smoke():
    import mdqe_oracle as O
    from mdqe_cvpr2023_amd.config import MDQEConfig
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    from mdqe_cvpr2023_amd.params import random_state
    kw = dict(enc_layers=2, dec_layers=2, n_frames=3, num_classes=5, num_queries=16, query_embed_dim=16)
    ev = dict(n_frames_test=3, n_frames_window_test=4, n_max_inst=40, apply_cls_thres=0.12)
    cfg = MDQEConfig(**kw, **ev)
    sd = random_state(cfg, seed=3)
    g = torch.Generator().manual_seed(1)
    base = torch.randint(0, 256, (3, 64, 96), generator=g, dtype=torch.uint8).float()
    frames = [(0.8 * base + 0.2 * torch.randint(0, 256, (3, 64, 96), generator=g, dtype=torch.uint8).float()).round().to(torch.uint8)
              for _ in range(5)]
    model = MDQE(cfg, state_dict=sd).eval()
    trace = []
    with torch.no_grad():
        out = model.inference_vis([{"image": frames, "height": 64, "width": 96}], trace=trace)
    hp = O.Hyper(**kw, **ev)
    ref_trace = []
    with torch.no_grad():
        ref = O.inference_vis(sd, hp, frames, lambda im: O.resnet(sd, "detr.backbone.0.backbone", im, 50), out_size=(64, 96),
                              trace=ref_trace)
    assert len(trace) == len(ref_trace)
    err = 0.0
    for a, b in zip(trace, ref_trace):
        assert a["pred_masks"].shape == b["pred_masks"].shape, (a["pred_masks"].shape, b["pred_masks"].shape)
        err = max(err, float((a["pred_masks"].cpu() - b["pred_masks"]).abs().max()))
    assert err < 1e-3, err
    assert out["pred_labels"] == ref["pred_labels"]
    diff = torch.stack(out["pred_masks"]) != torch.stack(ref["pred_masks"])
    assert diff.float().mean() < 1e-3
    print(f"smoke ok: {len(trace)} clips, max |mask logit diff| vs oracle = {err:.2e}, "
          f"{len(out['pred_scores'])} outputs, mask mismatch {diff.float().mean():.1e}")
