#!/usr/bin/env python3
"""The reference's "small" branch (vis_homo.py:73-78,90-91) on 32 x 1080p uint8 frames: resize to 852 x 480 + warp of the small frames
(the reference's two steps, its pixels) against the fused one-pass form (warp_perspective_resized), HIP-event times per launch.
GPU box:  python tools/time_small_branch.py [BEVWARP_LIB=... for a variant]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd import warp  # noqa: E402
from bev_amd.resize import resize  # noqa: E402
from tests import workloads as wl  # noqa: E402


def times(fn, n=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) * 1e3 for a, b in ev])


B, SH, SW, NW, NH, D = 32, 1080, 1920, 852, 480, 1024
dev = torch.device("cuda", 0)
nset = 6  # 6 x 199 MB of sources: past the Infinity Cache
sets = [torch.from_numpy(np.stack([wl.frame(32 * s + i, SH, SW, np.uint8) for i in range(B)])).to(dev) for s in range(nset)]
small = [torch.empty((B, NH, NW, 3), dtype=torch.uint8, device=dev) for _ in range(nset)]
outs = [torch.empty((B, D, D, 3), dtype=torch.uint8, device=dev) for _ in range(nset)]
M_small = np.stack([wl.jitter_H(wl.keystone_H(NW, NH, D, D), i) for i in range(B)])
S = warp.resize_matrix((SW, SH), (NW, NH), False)
M_fused = np.stack([m @ S for m in M_small])
minv_small, minv_fused = warp.device_inverse(M_small, dev), warp.device_inverse(M_fused, dev)
k = [0]


def step_resize():
    i = k[0] % nset
    resize(sets[i], (NW, NH), out=small[i])
    k[0] += 1


def step_warp_small():
    i = k[0] % nset
    warp.warp_perspective(small[i], None, (D, D), out=outs[i], M_inv_device=minv_small)
    k[0] += 1


def step_two():
    i = k[0] % nset
    resize(sets[i], (NW, NH), out=small[i])
    warp.warp_perspective(small[i], None, (D, D), out=outs[i], M_inv_device=minv_small)
    k[0] += 1


def step_fused():
    i = k[0] % nset
    warp.warp_perspective(sets[i], None, (D, D), out=outs[i], M_inv_device=minv_fused)
    k[0] += 1


for name, fn in (("resize 32 x 1080p -> 852 x 480", step_resize), ("warp 32 x (852 x 480) -> 1024^2", step_warp_small), ("two steps", step_two),
                 ("fused (one pass over the 1080p frames)", step_fused)):
    t = times(fn)
    print("%-42s mean %7.1f us  min %7.1f" % (name, t.mean(), t.min()))
rd, wr = B * SH * SW * 3, B * NH * NW * 3
print("resize bytes: %.1f MB read (whole frames) + %.1f MB written" % (rd / 1e6, wr / 1e6))
