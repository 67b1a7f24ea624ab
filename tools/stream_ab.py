"""K videos through separate forward() calls against forward_stream() (one video of look-ahead), same box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
model = MDQE(cfg, state_dict=sd).eval()
calibrate_synthetic_scores(model, sd, cfg, 360, 640)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 120
K = 6
inp = [{"image": synth_video(0, L, seed=0).cuda(), "height": 360, "width": 640}]
with torch.no_grad():
    ref = model(inp)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(K):
            o = model(inp)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        outs = list(model.forward_stream(inp for _ in range(K)))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("separate calls %.1f ms/video (%.1f fps)   stream %.1f ms/video (%.1f fps)" % (
            1e3 * (t1 - t0) / K, L * K / (t1 - t0), 1e3 * (t2 - t1) / K, L * K / (t2 - t1)))
    for o in outs:
        assert o["pred_labels"] == ref["pred_labels"] and o["pred_scores"] == ref["pred_scores"]
        assert all(torch.equal(a, b) for a, b in zip(o["pred_masks"], ref["pred_masks"]))
    print("stream outputs identical to separate calls:", len(outs), "videos,", len(ref["pred_scores"]), "instances each")
