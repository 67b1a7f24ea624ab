"""Host cost of one ABI launch: stream lookup old vs new, and a whole tiny op."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdqe_cvpr2023_amd import ops, _lib
torch.cuda.init()
n = 20000
t0 = time.perf_counter()
for _ in range(n):
    torch.cuda.current_stream(None).cuda_stream
t1 = time.perf_counter()
for _ in range(n):
    _lib.raw_stream()
t2 = time.perf_counter()
print("torch.cuda.current_stream().cuda_stream %.2f us   raw_stream() %.2f us" % (1e6 * (t1 - t0) / n, 1e6 * (t2 - t1) / n))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    assert _lib.raw_stream() == s.cuda_stream
assert _lib.raw_stream() == torch.cuda.current_stream().cuda_stream
x = torch.randn(64, 256, device="cuda"); g = torch.ones(256, device="cuda"); o = torch.empty_like(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5000):
    ops.layernorm(x, g, g, out=o)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host time per small ABI op (layernorm 64x256): %.2f us" % (1e6 * (t1 - t0) / 5000))
