#!/usr/bin/env python3
"""Phase breakdown of warp_tiles from the -DBEVWARP_TIMING build (diagnostic; shares, not run time).
GPU box:  BEVWARP_LIB=bev_amd/csrc/variants/libbevwarp_timing.so python tools/phases.py [u8|f32]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd import _lib, warp  # noqa: E402
from tests import workloads as wl  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "u8"
B, sw, sh, dw, dh = 32, 1920, 1080, 1024, 1024
base = wl.keystone_H(sw, sh, dw, dh)
Ms = np.stack([wl.jitter_H(base, i) for i in range(B)])
src = torch.randint(0, 256, (B, sh, sw, 3), dtype=torch.uint8, device="cuda")
if dtype == "f32":
    src = src.float() / 256
lib = _lib.load()
lib.bevwarp_debug_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    warp.warp_perspective(src, Ms, (dw, dh))
torch.cuda.synchronize()
lib.bevwarp_debug_phases(buf, 1)
n = 10
for _ in range(n):
    warp.warp_perspective(src, Ms, (dw, dh))
torch.cuda.synchronize()
lib.bevwarp_debug_phases(buf, 0)
names = ["0 matrix+lane consts", "1 corners+rowtab+barrier", "2 region", "3 stage issue", "4 stage barrier", "5 fast rows", "6 general rows"]
wgs = buf[15]
tot = sum(buf[i] for i in range(7))
print("workgroups %d  mean ticks per workgroup (wave 0) %.0f" % (wgs, tot / wgs))
for i, nm in enumerate(names):
    print("  %-26s %8.0f ticks  %5.1f %%" % (nm, buf[i] / wgs, 100.0 * buf[i] / tot))
