#!/usr/bin/env python3
"""On-box HBM ceilings with stock torch kernels (streaming copy / fill / read), buffers >> 256 MB Infinity Cache."""
import torch
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
N = 1 << 30  # 1 GiB per buffer
src = torch.empty(N, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty_like(src)
s4 = src.view(torch.float32); d4 = dst.view(torch.float32)
t = timeit(lambda: d4.copy_(s4)); print("copy  1 GiB -> 1 GiB : %.1f us  %.2f TB/s (read+write)" % (t * 1e6, 2 * N / t / 1e12))
t = timeit(lambda: d4.fill_(1.0)); print("fill  1 GiB         : %.1f us  %.2f TB/s (write)" % (t * 1e6, N / t / 1e12))
t = timeit(lambda: s4.sum()); print("sum   1 GiB         : %.1f us  %.2f TB/s (read)" % (t * 1e6, N / t / 1e12))
big = torch.empty(3 * N, dtype=torch.uint8, device="cuda").view(torch.float32)
t = timeit(lambda: big.mul_(1.0001)); print("scale 3 GiB in place: %.1f us  %.2f TB/s (read+write)" % (t * 1e6, 6 * N / t / 1e12))
