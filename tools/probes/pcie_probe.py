#!/usr/bin/env python
"""Host <-> device copy rates of the box for page-locked buffers of a packed factor's size (2.9 GB at n = 27 000), one direction and both
at once, one and four copies in flight (torch is only the probe's way to page-locked memory and streams)."""
import time, torch
n = int(2.9e9) // 8
h = [torch.empty(n, dtype=torch.float64).pin_memory() for _ in range(4)]
d = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
s = [torch.cuda.Stream() for _ in range(8)]
def run(pairs, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for q, (dst, src) in enumerate(pairs):
            with torch.cuda.stream(s[q]):
                dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return reps * sum(x[0].numel() for x in pairs) * 8 / (time.perf_counter() - t0) / 1e9
run([(d[0], h[0])], 1)
print("H2D one copy      %.1f GB/s" % run([(d[0], h[0])]))
print("D2H one copy      %.1f GB/s" % run([(h[0], d[0])]))
print("H2D four at once  %.1f GB/s" % run([(d[q], h[q]) for q in range(4)]))
print("D2H four at once  %.1f GB/s" % run([(h[q], d[q]) for q in range(4)]))
print("H2D + D2H (2 + 2) %.1f GB/s in all" % run([(d[0], h[0]), (d[1], h[1]), (h[2], d[2]), (h[3], d[3])]))
