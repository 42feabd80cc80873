#!/usr/bin/env python3
"""Shader clock the chip holds while the warp kernel runs (guide: DVFS give-back item 6).  GPU box, with the diagnostic
build:  make -C bev_amd/csrc variants/clock.so  &&  python tools/clock.py [--dtype u8] [--interp linear] [--seconds 2]
Launches back to back for `seconds`, then reads sum(delta s_memtime) / sum(delta s_memrealtime) x 100 MHz over all
workgroups of the last launches."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BEVWARP_LIB", os.path.join(ROOT, "bev_amd", "csrc", "variants", "clock.so"))

p = argparse.ArgumentParser()
p.add_argument("--dtype", default="u8")
p.add_argument("--interp", default="linear")
p.add_argument("--homography", default="keystone")
p.add_argument("--seconds", type=float, default=2.0)
args = p.parse_args()
from bev_amd import _lib, warp  # noqa: E402
from tests import workloads as wl  # noqa: E402

lib = _lib.load()
dbg = lib.bevwarp_debug_clock
dbg.restype, dbg.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
B, sw, sh, dw, dh = 32, 1920, 1080, 1024, 1024
ndt = np.uint8 if args.dtype == "u8" else np.float32
base = (wl.keystone_H if args.homography == "keystone" else wl.synth_brno_H)(sw, sh, dw, dh)
Ms = np.stack([wl.jitter_H(base, g) for g in range(B)])
f0 = torch.stack([torch.from_numpy(wl.frame(g, sh, sw, ndt)) for g in range(4)]).cuda()
srcs = [torch.cat([f0.roll(s, 0)] * (B // 4)).contiguous() for s in range(4)]
dsts = [torch.empty((B, dh, dw, 3), dtype=srcs[0].dtype, device="cuda") for _ in range(4)]
minv = warp.device_inverse(Ms, torch.device("cuda", 0))
flags = 1 if args.interp == "linear" else 0
t_end = time.time() + args.seconds
n = 0
while time.time() < t_end:
    for _ in range(50):
        warp.warp_perspective(srcs[n % 4], None, (dw, dh), flags=flags, out=dsts[n % 4], M_inv_device=minv)
        n += 1
    torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
assert dbg(out, 1) == 0
for _ in range(20):
    warp.warp_perspective(srcs[n % 4], None, (dw, dh), flags=flags, out=dsts[n % 4], M_inv_device=minv)
    n += 1
torch.cuda.synchronize()
assert dbg(out, 0) == 0
print("%s %s %s: %d workgroups, mean life %.0f shader ticks = %.2f us, clock held %.0f MHz" % (
    args.dtype, args.interp, args.homography, out[2], out[0] / out[2], out[1] / out[2] / 100.0, 100.0 * out[0] / out[1]))
if out[7]:  # staged tiles ran (rows_staged.inc): per-role stamps, shader ticks
    P, C = out[7], out[7] * 3
    print("staged tiles %d: producer life %.0f ticks, %.1f source rows of which %.2f found no free slot at once; consumer life %.0f ticks, %.1f rows each, "
          "waiting for source rows %.0f ticks (%.0f %% of its life), of which before its first row %.0f (%.0f %%)" % (
              P, out[4] / P, out[6] / P, out[5] / P, out[8] / C, out[11] / C, out[9] / C, 100.0 * out[9] / out[8], out[10] / C, 100.0 * out[10] / out[8]))
