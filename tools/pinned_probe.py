"""Does the pinned-host caching allocator recycle the per-track mask buffers of a video (ClipMerger._early_masks), and what does a fresh pinned
allocation cost?      python tools/pinned_probe.py"""
import time, torch
src = torch.zeros(30, 360, 640, dtype=torch.uint8, device="cuda")
def alloc(n_frames, n=7):
    t0 = time.perf_counter()
    bufs = [torch.empty(n_frames, 360, 640, dtype=torch.uint8, pin_memory=True) for _ in range(n)]
    return bufs, 1e3 * (time.perf_counter() - t0)
for n_frames in (120, 960):
    keep = []
    for it in range(6):
        bufs, ms = alloc(n_frames)
        for b in bufs:
            b[:30].copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        keep.append([b.view(torch.bool)[:n_frames] for b in bufs])      # what the caller holds: views
        del bufs
        if len(keep) > 1:
            keep.pop(0)                                                  # the previous video's result is dropped
        print("n_frames %d iteration %d: 7 pinned buffers of %.0f MB in %.2f ms" % (n_frames, it, n_frames * 0.2304, ms), flush=True)
