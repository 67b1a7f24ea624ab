"""Is a device->pinned-host `copy_(non_blocking=True)` asynchronous for the host, and what does the link carry?  (ClipMerger._early_masks: per flushed
window, one copy per track into a slice of that track's pinned [frames, H, W] buffer.)      python tools/d2h_probe.py"""
import time, torch
dev = torch.device("cuda")
n, nf, H, W = 7, 30, 360, 640
src = torch.randint(0, 2, (n, nf, H, W), dtype=torch.uint8, device=dev)
big = [torch.empty(960, H, W, dtype=torch.uint8, pin_memory=True) for _ in range(n)]
flat = torch.empty(n, nf, H, W, dtype=torch.uint8, pin_memory=True)
cs = torch.cuda.Stream()
spin = torch.empty(1 << 28, device=dev)          # a 1-GB fill keeps the stream busy ~0.3 ms


def run(label, fn, reps=5):
    for r in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(cs):
            spin.fill_(1.0)                          # work in front of the copies: an async copy call returns before it is done
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
        cs.synchronize()
        t2 = time.perf_counter()
    print("%-60s host %.3f ms in the calls, %.3f ms until complete (%.1f GB/s over both)" % (label, 1e3 * (t1 - t0), 1e3 * (t2 - t0), src.numel() / (t2 - t0) / 1e9), flush=True)


run("7 slices of 7 pinned [960,H,W] buffers (as _early_masks)", lambda: [big[i][60:60 + nf].copy_(src[i], non_blocking=True) for i in range(n)])
run("7 rows of ONE pinned [7,30,H,W] buffer", lambda: [flat[i].copy_(src[i], non_blocking=True) for i in range(n)])
run("ONE copy of the whole [7,30,H,W] block", lambda: flat.copy_(src, non_blocking=True))
print("is_pinned: slice %s, row %s, whole %s" % (big[0][60:90].is_pinned(), flat[0].is_pinned(), flat.is_pinned()))
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def raw():
    for i in range(n):
        hip.hipMemcpyAsync(ctypes.c_void_p(big[i][60:60 + nf].data_ptr()), ctypes.c_void_p(src[i].data_ptr()), ctypes.c_size_t(src[i].numel()), 2, ctypes.c_void_p(cs.cuda_stream))
run("raw hipMemcpyAsync, 7 slices", raw)
