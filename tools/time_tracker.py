#!/usr/bin/env python3
"""Time of the one-launch tracker step and of the IoU matrix alone (512 x 512, float64 and float32) with the library
BEVWARP_LIB names (default: the in-tree build).  GPU box:  BEVWARP_LIB=bev_amd/csrc/variants/x.so python tools/time_tracker.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd.iou import rbox_iou  # noqa: E402
from bev_amd.tracker_geom import tracker_geometry_step  # noqa: E402


def times(fn, n=200, warm=20):
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


rng = np.random.default_rng(11)
H_world_bev = np.array([[0.0, 0.0625, -10.0], [-0.0625, 0.0, 40.0], [0, 0, 1.0]])
H_img_world = np.linalg.inv(np.array([[0.02, -0.001, -3.0], [0.0004, 0.05, -20.0], [1e-5, 0.0009, 0.4]]))
for dt in (torch.float64, torch.float32):
    dets = torch.from_numpy(np.column_stack([rng.uniform(0, 1024, (512, 2)), rng.uniform(25, 35, 512), rng.uniform(56, 96, 512), rng.uniform(-np.pi, np.pi, 512)])).cuda().to(dt)
    trks = torch.from_numpy(np.column_stack([rng.uniform(-10, 54, 512), rng.uniform(-24, 40, 512), rng.uniform(1.6, 2.2, 512), rng.uniform(3.5, 6, 512),
                                             rng.uniform(-np.pi, np.pi, 512)])).cuda().to(dt)
    buf = tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world)
    t1 = times(lambda: tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world, out=buf))
    io = torch.empty((512, 512), dtype=dt, device="cuda")
    dw = buf["dets_world"].clone()  # (the same pairs as the step scores)
    t2 = times(lambda: rbox_iou(dw, trks, out=io))
    print("%s: tracker step median %.1f us (min %.1f)   iou alone median %.1f us (min %.1f)   candidates %d" % (
        str(dt).split(".")[1], np.median(t1), t1.min(), np.median(t2), t2.min(), int(buf["candidates"].sum())))
