"""HIP multiplexes streams onto 4 hardware queues in creation order: does the pipeline's throughput depend on WHICH of its six streams
share a queue?  Creates k dummy streams (normal / high priority) before the model creates its own, one process per setting.
python tools/stream_map_ab.py <n_normal> <n_high>"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nn_, nh_ = int(sys.argv[1]), int(sys.argv[2])
pads = [torch.cuda.Stream() for _ in range(nn_)] + [torch.cuda.Stream(priority=-1) for _ in range(nh_)]
for p in pads:                                   # make sure the runtime really creates them
    with torch.cuda.stream(p):
        torch.zeros(1, device="cuda")
import bench
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
model = MDQE(cfg, state_dict=sd).eval()
bench.calibrate_synthetic_scores(model, sd, cfg, 360, 640)
video = bench.synth_video(0, 120, seed=0).pin_memory()
inp = [{"image": list(video), "height": 360, "width": 640}]
with torch.no_grad():
    for _ in range(3):
        model(inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        model(inp)
    torch.cuda.synchronize()
print("dummy streams before the model's: %d normal + %d high priority -> %.1f frames/s" % (nn_, nh_, 1200 / (time.perf_counter() - t0)), flush=True)
