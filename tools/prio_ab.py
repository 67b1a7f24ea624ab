"""Does a high-priority caller stream (the clip stages: many small kernels + host syncs) help against the frame stream's GEMMs?"""
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
K, L = 6, 120
inp = [{"image": synth_video(0, L, seed=0).cuda(), "height": 360, "width": 640}]
hp = torch.cuda.Stream(priority=-1)
with torch.no_grad():
    list(model.forward_stream(inp for _ in range(2)))
    for rep in range(3):
        for name, st in (("default-priority caller stream", None), ("high-priority caller stream", hp)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if st is None:
                list(model.forward_stream(inp for _ in range(K)))
            else:
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    list(model.forward_stream(inp for _ in range(K)))
            torch.cuda.synchronize()
            print("%-34s %.1f ms/video  %.1f fps" % (name, 1e3 * (time.perf_counter() - t0) / K, L * K / (time.perf_counter() - t0)), flush=True)
