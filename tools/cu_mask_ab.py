"""Does confining the frame stream's large GEMMs to a subset of the CUs (hipExtStreamCreateWithCUMask) leave the per-clip stage's small
kernels somewhere to run, and does that beat letting both streams share every CU?  R50_ovis_360, 120 resident frames, one model per mask
(a model owns its streams), alternated.   python tools/cu_mask_ab.py [masked-out CU counts ...]
Mask bit i = CU i in the runtime's numbering (on a multi-XCD part consecutive bits go to consecutive XCDs), so the first k bits cleared
take k/8 CUs from every XCD; `hi` variants clear the last k bits instead."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video, calibrate_synthetic_scores
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(clear_lo=0, clear_hi=0, only_lo=0):
    bits = [1] * NCU
    for i in range(clear_lo):
        bits[i] = 0
    for i in range(clear_hi):
        bits[NCU - 1 - i] = 0
    if only_lo:
        bits = [1 if i < only_lo else 0 for i in range(NCU)]
    words = (ctypes.c_uint32 * ((NCU + 31) // 32))()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= (1 << (i % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


cfg = PRESETS["R50_ovis_360"]
sd = random_state(cfg, seed=0)
video = synth_video(0, 120, seed=0).cuda()
inp = [{"image": video, "height": 360, "width": 640}]
variants = [("all CUs (default streams)", None)]
for k in [int(a) for a in sys.argv[1:]] or [16, 32, 64]:
    variants.append(("frame stream without the first %d CUs" % k, dict(clear_lo=k)))
variants.append(("frame stream on a masked stream with every CU set", dict()))
models = []
for name, kw in variants:
    m = MDQE(cfg, state_dict=sd).eval()
    calibrate_synthetic_scores(m, sd, cfg, 360, 640)
    if kw is not None:
        m._frame_stream = masked_stream(**kw)
    models.append((name, m))
ref = None
with torch.no_grad():
    for name, m in models:
        o = m(inp); m(inp)
        key = (o["pred_scores"], o["pred_labels"])
        if ref is None:
            ref = key
        assert key == ref, name
    torch.cuda.synchronize()
    for rnd in range(3):
        for name, m in models:
            t0 = time.perf_counter()
            for _ in range(5):
                m(inp)
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / 5
            print("round %d  %-52s %.1f ms  %.1f frames/s" % (rnd, name, 1e3 * d, 120 / d), flush=True)
