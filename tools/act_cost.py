import os, sys, torch
sys.path.insert(0, "/root/repo")
from mdqe_cvpr2023_amd import ops
def time_us(fn, iters=12, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
g = torch.Generator(device="cuda").manual_seed(0)
junk = torch.empty(8 << 20, device="cuda"); junk2 = torch.empty_like(junk)
base = time_us(lambda: junk2.copy_(junk))
M, N, K = 204000, 1024, 256
x = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) / 16
b = torch.randn(N, device="cuda", generator=g); out = torch.empty(M, N, device="cuda")
for rep in range(3):
    for act in (None, "relu", "gelu", "tanh"):
        t = time_us(lambda: (ops.linear(x, w, b, act=act, out=out), junk2.copy_(junk))) - base
        print("FFN1 shape act=%-5s %7.1f us = %5.1f TF" % (act, t, 2.0*M*N*K/t/1e6), flush=True)
    t = time_us(lambda: (ops.linear(x, w, None, act=None, out=out), junk2.copy_(junk))) - base
    print("FFN1 shape no bias     %7.1f us = %5.1f TF" % (t, 2.0*M*N*K/t/1e6), flush=True)
