"""Round 6 diagnosis: the shipped R50_ovis_360 schedule at full size (34 frames, window 30) -- per clip, the decoder on the oracle's encoder
tokens against the oracle's; for the clips over the bar: which discrete decision differs (query-cell argmax, inter-frame association) and by
what margin the oracle's own decision was taken.    python tools/diag_shipped.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
torch.set_num_threads(16)
import mdqe_oracle as O
import test_fullsize_gpu as TF
from mdqe_cvpr2023_amd import ops

ref = TF._workload("R50_ovis_360", 360, 640, 34, 30, max_inst=120)
model = TF._model(ref)
eng, cfg, hp, sd = model.engine, ref["cfg"], ref["hp"], ref["sd"]
geo = eng.geometry(360, 640)
with torch.no_grad():
    enc_d = ref["enc"].cuda().contiguous()
    coords, content, emb = eng.frame_queries(enc_d, geo)
    vals = eng.dec_values(enc_d, geo)
    cache = {"coords": coords, "content": content, "emb": emb, "vals": vals}
    bad = []
    for ci, c in enumerate(ref["clips"]):
        s, e = c["start"], c["end"]
        out = eng.decode_clips(cache, [s], e - s, geo)
        d = {k: float((out[k][0].cpu() - c["out"][k][0]).abs().max()) for k in ("cls", "mask_coeff", "query_embed")}
        flag = d["cls"] > 1e-4
        print("clip %2d frames %2d-%2d  cls %.2e  mask_coeff %.2e  query_embed %.2e %s" % (ci, s, e, d["cls"], d["mask_coeff"], d["query_embed"], "<<<" if flag else ""), flush=True)
        if flag:
            bad.append(ci)
    for ci in bad[:4]:
        c = ref["clips"][ci]
        s, e = c["start"], c["end"]
        T = e - s
        dbg = {}
        O.query_initialization(sd, hp, ref["enc"][s:e], ref["shapes"], dbg=dbg)
        co = dbg["coords0"]                                                   # [T,Q,2]
        cp = coords[s:e].cpu()
        dc = (co - cp).abs().amax(-1)                                           # [T,Q]
        print("clip %d: query cells whose coordinates differ: %d of %d (max |d| %.3e)" % (ci, int((dc > 1e-6).sum()), dc.numel(), float(dc.max())))
        for t, q in (dc > 1e-6).nonzero().tolist()[:8]:
            # the oracle's margin in that cell: best minus second best of the up-sampled score inside the cell
            sc = dbg["score_up"][t]
            nb = hp.n_bins
            Hu, Wu = sc.shape[-2:]
            r, w_ = Hu // nb, Wu // nb
            cell = sc.reshape(nb, r, nb, w_).permute(0, 2, 1, 3).reshape(nb * nb, r * w_)[q]
            top = torch.topk(cell, 2).values
            print("   frame %d cell %3d: oracle (%.6f, %.6f) product (%.6f, %.6f); oracle's best - second best score in the cell = %.3e" %
                  (s + t, q, co[t, q, 0], co[t, q, 1], cp[t, q, 0], cp[t, q, 1], float(top[0] - top[1])))
        # association
        ct = int((T - 1) / 2)
        fidx = torch.tensor([[s + t for t in range(T)]], dtype=torch.int32).cuda()
        idx_p = ops.clip_assoc(emb, fidx, ct, cfg.window_inter_frame_asso / 2, cfg.n_bins).view(T, -1).cpu() if T > 1 else None
        idx_o = dbg.get("assoc_idx")
        if idx_p is not None and idx_o is not None:
            dif = (idx_p.long() != idx_o.long())
            print("clip %d: association entries that differ: %d of %d" % (ci, int(dif.sum()), dif.numel()))
            em = dbg["track_emb"]
            sim = torch.einsum("tqc,kc->tqk", em, em[ct])
            rel = O.query_relpos_grid(hp.n_bins)
            for t, k in dif.nonzero().tolist()[:8]:
                itv = max(t - ct, ct - t)
                m = (rel > (hp.window_inter_frame_asso / 2) * itv).any(-1)
                col = sim[t].masked_fill(m, float("-inf")).softmax(-2)[:, k]
                top = torch.topk(col, 2)
                print("   t %d key %3d: oracle picks %d, product %d; oracle's softmax column top-2 = %.9f (q %d), %.9f (q %d); product emb diff on that frame %.2e" %
                      (t, k, int(idx_o[t, k]), int(idx_p[t, k]), float(top.values[0]), int(top.indices[0]), float(top.values[1]), int(top.indices[1]),
                       float((emb[s + t].cpu() - em[t]).abs().max())))
