#!/usr/bin/env python3
"""PCIe rates of this box with pinned memory (GPU box): H2D alone, D2H alone, both at once on two streams, and a device
kernel reading pinned host memory (torch copy_ of a mapped tensor is an SDMA transfer; `clone` on the device side of a
pinned tensor is not available, so the kernel path is probed with an elementwise add).  Sizes: one 1080p uint8 frame up
(6.2 MB), one 1024^2 BEV frame down (3.1 MB)."""
import time

import torch

up_h = torch.empty(1080 * 1920 * 3, dtype=torch.uint8, pin_memory=True)
dn_h = torch.empty(1024 * 1024 * 3, dtype=torch.uint8, pin_memory=True)
up_d, dn_d = torch.empty_like(up_h, device="cuda"), torch.empty_like(dn_h, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def rate(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def h2d():
    with torch.cuda.stream(s1):
        up_d.copy_(up_h, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        dn_h.copy_(dn_d, non_blocking=True)


def both():
    h2d()
    d2h()


t = rate(h2d)
print("H2D 6.2 MB alone        %.1f us  %.1f GB/s" % (t * 1e6, up_h.numel() / t / 1e9))
t = rate(d2h)
print("D2H 3.1 MB alone        %.1f us  %.1f GB/s" % (t * 1e6, dn_h.numel() / t / 1e9))
t = rate(both)
print("both, two streams       %.1f us per pair (sum alone would be the serial time)" % (t * 1e6))
