"""Per-stage wall-time breakdown (synchronised timers; diagnostic only, never inside a timed bench region)."""
import time

import torch


def _t(fn, n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0) / n


def stage_breakdown(model, frames_dev, chunk=30, n_clips=8):
    from .tracking import Clips, OverTracker
    eng, cfg = model.engine, model.cfg
    fr = frames_dev[:chunk]
    h, w = int(fr.shape[-2]), int(fr.shape[-1])
    geo = eng.geometry(h, w)
    out = {}
    feats, out["backbone"] = _t(lambda: eng.backbone(fr, geo))
    enc, out["encoder"] = _t(lambda: eng.encode(feats, geo))
    mf, out["mask_head"] = _t(lambda: eng.mask_features(enc, geo))
    q, out["frame_queries"] = _t(lambda: eng.frame_queries(enc, geo))
    vals, out["dec_values"] = _t(lambda: eng.dec_values(enc, geo))
    out = {k + f"_per_{fr.shape[0]}f": v for k, v in out.items()}
    T = cfg.n_frames_test
    coords, content, emb = q
    dec, out["decode_clip"] = _t(lambda: eng.decode_clip(coords[:T], content[:T], emb[:T], vals[:T], geo), n_clips)
    res, out["inference_clip"] = _t(lambda: eng.inference_clip(dec, mf[:T]), n_clips)
    nb = fr.shape[0] - T + 1
    cache = {"coords": coords, "content": content, "emb": emb, "vals": vals}
    decs, out[f"decode_clips_x{nb}"] = _t(lambda: eng.decode_clips(cache, list(range(nb)), T, geo), 2)
    _, out[f"inference_clips_x{nb}"] = _t(lambda: eng.inference_clips(decs, [mf[i:i + T] for i in range(nb)]), 2)
    ms = cfg.match_stride
    trk, out["tracker_alloc"] = _t(lambda: OverTracker(cfg.n_max_inst, T, cfg.n_frames_window_test, cfg.clip_stride, cfg.num_classes,
                                                       cfg.mask_dim, cfg.hidden_dim, (geo.Hp // ms, geo.Wp // ms), model.device,
                                                       cfg.apply_cls_thres))
    t_up = 0.0
    for i in range(n_clips):
        _, dt = _t(lambda: trk.update(Clips(range(i, i + T), res)))
        t_up += dt
    out["tracker_update"] = t_up / n_clips
    (c, m), out["tracker_get_result"] = _t(lambda: trk.get_result(True))
    cls_clips, windows = [c], [(0, m.contiguous())]
    _, out["inference_video"] = _t(lambda: model.inference_video((h, w), cls_clips, windows, (h, w), m.shape[1]))
    out["n_inst_clip"] = int(len(res["scores"]))
    return out
