#!/usr/bin/env python3
"""A/B of project_points builds (access kinds), interleaved in one process: python tools/ab_points.py label=lib.so ..."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bev_amd import _lib
libs = []
for spec in sys.argv[1:]:
    label, path = spec.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.bevwarp_project_points
    fn.restype, fn.argtypes = _lib.SYMBOLS["bevwarp_project_points"]
    libs.append((label, fn))
H = np.ascontiguousarray(np.array([[0.02, -0.001, -3.0], [0.0004, 0.05, -20.0], [1e-5, 0.0009, 0.4]]))
N = 10_000_000
dev = torch.device("cuda", 0)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for tdt, code, esz in ((torch.float32, 1, 4), (torch.float64, 2, 8)):
    nbuf = max(2, int(np.ceil(600e6 / (N * 2 * esz * 2))))
    ins = [(torch.rand((N, 2), dtype=torch.float64, device=dev) * 1000).to(tdt) for _ in range(nbuf)]
    outs = [torch.empty_like(x) for x in ins]
    times = {l: [] for l, _ in libs}
    it = 0
    for r in range(44):
        for label, fn in (libs if r % 2 == 0 else libs[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert fn(ins[it % nbuf].data_ptr(), outs[it % nbuf].data_ptr(), N, 2, H.ctypes.data_as(ctypes.c_void_p), code, stream) == 0
            e1.record()
            it += 1
            e1.synchronize()
            if r >= 4:
                times[label].append(e0.elapsed_time(e1) * 1e3)
    for label, _ in libs:
        t = np.array(times[label])
        print("%s %-10s median %7.1f us  min %7.1f us   %.2f TB/s (median)" % (str(tdt)[6:], label, np.median(t), t.min(), N * 4 * esz / np.median(t) / 1e6))
    del ins, outs
    torch.cuda.empty_cache()
