"""In-kernel phase timing of the f16x3w GEMM (s_memtime stamps per block): prologue / K loop / epilogue cycles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops
from mdqe_cvpr2023_amd._lib import lib, ptr
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (153000, 1024, 256)
act = sys.argv[4] if len(sys.argv) > 4 else None
x = torch.randn(M, K, device="cuda"); w = ops.const_weight(torch.randn(N, K, device="cuda") / 16); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
ops.set_gemm_precision("f16x3")
for _ in range(5):
    ops.linear(x, w, b, act=act, out=out, tile=1)
nblk = 512
buf = torch.zeros(nblk * 4 + 64, dtype=torch.int64, device="cuda")
lib.mdqe_debug_gemm_stamps(ptr(buf))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.linear(x, w, b, act=act, out=out, tile=1); e1.record()
torch.cuda.synchronize()
lib.mdqe_debug_gemm_stamps(None)
s = buf[:nblk * 4].view(-1, 4).cpu()
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
d = (s - t0).double()
print("blocks", len(s), "kernel ms", e0.elapsed_time(e1), "span cycles", float(d[:, 3].max()), "(100 MHz ticks if memtime is realtime)")
for name, a, b_ in (("prologue", 0, 1), ("kloop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
    v = d[:, b_] - d[:, a]
    print("%-9s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f" % (name, v.mean(), v.quantile(0.1), v.quantile(0.5), v.quantile(0.9)))
st = d[:, 0].sort().values
print("block start times: p1 %.0f p25 %.0f p50 %.0f p75 %.0f p99 %.0f" % tuple(float(st[int(q * (len(st) - 1))]) for q in (0.01, 0.25, 0.5, 0.75, 0.99)))
