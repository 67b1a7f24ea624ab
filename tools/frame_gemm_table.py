"""Every GEMM / conv launch of one pass of the per-frame stages (40 frames of 360p by default), timed with HIP events on the launch
stream, grouped by shape: count, average duration, achieved TFLOP/s, share of the pass.  python tools/frame_gemm_table.py [config] [frames]
Corresponds to ONE full-size frame pass of `python bench.py --config <config>` (40 frames at 360p, 20 at 640p, 35 for Swin-L) run ALONE on
the chip with an event pair around every launch: the figures are the kernels' isolated rates (the bench's `roofline_isolated`), not what
they reach beside the clip stream's kernels (`roofline`)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_video
from mdqe_cvpr2023_amd.config import PRESETS
from mdqe_cvpr2023_amd.meta_arch import MDQE
from mdqe_cvpr2023_amd.params import random_state
from mdqe_cvpr2023_amd import ops, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "R50_ovis_360"
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fh, fw = {"R50_ovis_360": (360, 640), "R50_ovis_720": (640, 1138), "swinl_ovis": (480, 853)}[name]
cfg = PRESETS[name]
model = MDQE(cfg, state_dict=random_state(cfg, seed=0)).eval()
video = synth_video(0, nfr, seed=0, h=fh, w=fw).cuda()
rec = []
L = _lib.load_library()
if os.environ.get("GEMM_VARIANT"):                       # tools/ A/B: 1 = the K-step-16 kernel for every shape
    L.mdqe_debug_gemm_variant(int(os.environ["GEMM_VARIANT"]))
raw = {n: getattr(L, n) for n in ("mdqe_gemm_nt_f32", "mdqe_gemm_ln_f32", "mdqe_conv2d_nhwc_f32", "mdqe_gemm_nt_cat2_f32")}


class Wrapped:
    def __getattr__(self, n):
        return getattr(L, n)

    def _timed(self, n, key, flops, a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = raw[n](*a); e1.record()
        rec.append((key, flops, e0, e1))
        return rc

    def mdqe_gemm_nt_f32(self, *a):
        M, N, K = a[6], a[7], a[8]
        return self._timed("mdqe_gemm_nt_f32", "gemm     M=%7d N=%5d K=%5d" % (M, N, K), 2.0 * M * N * K, a)

    def mdqe_gemm_ln_f32(self, *a):
        M, N, K = a[6], a[7], a[8]
        return self._timed("mdqe_gemm_ln_f32", "gemm+LN  M=%7d N=%5d K=%5d" % (M, N, K), 2.0 * M * N * K, a)

    def mdqe_gemm_nt_cat2_f32(self, *a):
        K1, K2, NI, OH, OW, N = a[2], a[5], a[6], a[7], a[8], a[16]
        M = NI * OH * OW
        return self._timed("mdqe_gemm_nt_cat2_f32", "gemm cat M=%7d N=%5d K=%5d" % (M, N, K1 + K2), 2.0 * M * N * (K1 + K2), a)

    def mdqe_conv2d_nhwc_f32(self, *a):
        NI, H, W, Cin, Cout, KH, KW, stride, pad = a[6:15]
        M = NI * ((H + 2 * pad - KH) // stride + 1) * ((W + 2 * pad - KW) // stride + 1)
        return self._timed("mdqe_conv2d_nhwc_f32", "conv %dx%d/%d M=%7d N=%5d K=%5d" % (KH, KW, stride, M, Cout, KH * KW * Cin), 2.0 * M * Cout * KH * KW * Cin, a)


with torch.no_grad():
    geo = model.engine.geometry(fh, fw)
    model._frame_cache(video, geo)                       # warm
    torch.cuda.synchronize()
    ops.lib = Wrapped()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); model._frame_cache(video, geo); e1.record()
    torch.cuda.synchronize()
    ops.lib = L
tot = e0.elapsed_time(e1)
agg = collections.OrderedDict()
for key, fl, a, b in rec:
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1; d[1] += a.elapsed_time(b); d[2] += fl
gsum = sum(d[1] for d in agg.values())
print("%s, %d frames: pass %.2f ms (with per-launch events), GEMM/conv launches %d = %.2f ms, %.1f TF overall" % (name, nfr, tot, len(rec), gsum, sum(d[2] for d in agg.values()) / gsum / 1e9))
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%5.1f %%  %3d x %8.1f us  %6.1f TF  %s" % (100 * ms / gsum, n, 1e3 * ms / n, fl / ms / 1e9, key))
