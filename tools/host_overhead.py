#!/usr/bin/env python3
"""Where the host time of one small launch goes (configs[0]: 720p -> 512^2 u8): the C ABI call alone through ctypes, the Python
entry's validated-launch fast path, and the general path.  GPU box: python tools/host_overhead.py"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd import _lib, warp
from tests import workloads as wl

dev = torch.device("cuda", 0)
sw, sh, dw, dh = 1280, 720, 512, 512
M = wl.synth_brno_H(sw, sh, dw, dh)
src = torch.from_numpy(wl.frame(0, sh, sw, np.uint8)).to(dev)
out = torch.empty((dh, dw, 3), dtype=torch.uint8, device=dev)
minv = warp.device_inverse(M, dev)
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
args = (src.data_ptr(), out.data_ptr(), 1, sh, sw, dh, dw, 3, sh * sw * 3, sw * 3, dh * dw * 3, dw * 3, minv.data_ptr(), 1, 0, 1, None, stream)


def host_us(f, n=3000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


def gpu_us(f, n=1000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


fast = lambda: warp.warp_perspective(src, None, (dw, dh), out=out, M_inv_device=minv)  # noqa: E731
general = lambda: warp.warp_perspective(src, M, (dw, dh))  # noqa: E731
raw = lambda: lib.bevwarp_warp(*args)  # noqa: E731
for name, f in (("C ABI through ctypes", raw), ("Python entry, validated-launch cache", fast), ("Python entry, general path (allocates, looks the matrix up)", general)):
    print("%-62s host %6.2f us/call   back-to-back %6.2f us/call" % (name, host_us(f), gpu_us(f)))
print("torch.cuda.current_device()  %.2f us" % host_us(torch.cuda.current_device, 20000))
print("torch._C._cuda_getCurrentRawStream(0)  %.2f us" % host_us(lambda: torch._C._cuda_getCurrentRawStream(0), 20000))
small = torch.zeros(8, device=dev)
print("a trivial torch kernel (x.add_(1))  host %.2f us/call" % host_us(lambda: small.add_(1)))
