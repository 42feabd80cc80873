#!/usr/bin/env python3
"""Where the rotated-IoU kernel's time goes, in shader cycles: reads the stamps of a `stamps` build (tools/ablate_geom.py stamps=stamps).
GPU box:  BEVWARP_LIB=bev_amd/csrc/variants/stamps.so python tools/iou_stamps.py
Per wave that has a lane past the rejection test (its first such lane): cycles from the start of pair_iou to the rejection branch, through
the sincos pair, through the contour integral; and the clock those cycles ran at (s_memtime against the 100 MHz s_memrealtime)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd import _lib  # noqa: E402
from bev_amd.iou import rbox_iou  # noqa: E402

lib = _lib.load()
fn = lib.bevwarp_debug_iou_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(11)


def boxes(n, span):
    return np.column_stack([rng.uniform(0, span, (n, 2)), rng.uniform(1.6, 2.2, n), rng.uniform(3.5, 6, n), rng.uniform(-np.pi, np.pi, n)])


for span, label in ((100.0, "config 5 (100 m square)"), (64.0, "64 m square"), (12.0, "dense (12 m square)")):
    a, b = torch.from_numpy(boxes(512, span)).cuda(), torch.from_numpy(boxes(512, span)).cuda()
    out = torch.empty((512, 512), dtype=torch.float64, device="cuda")
    for _ in range(20):
        rbox_iou(a, b, out=out)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8192 * 8))()
    assert fn(buf, 1) == 0
    n = 50
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        rbox_iou(a, b, out=out)
    ev1.record()
    torch.cuda.synchronize()
    assert fn(buf, 1) == 0  # (the records of the LAST launch: one per workgroup, plain stores)
    rec = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
    rec = rec[rec[:, 0] > 0]
    m = rec.mean(axis=0)
    print("%-26s %5d waves past the rejection test (%.1f lanes each); per such wave, mean cycles: kernel start -> pair %5.0f, -> rejection branch %5.0f, sincos pair %5.0f, "
          "frame + contour %5.0f; pair total %5.0f (max %5.0f) = %4.2f us at %4.0f MHz   (launch + kernel, back to back: %.1f us)"
          % (label, len(rec), m[7], m[6], m[1], m[2], m[3], m[4], rec[:, 4].max(), m[5] / 100.0, 100.0 * m[4] / max(m[5], 1), ev0.elapsed_time(ev1) * 1e3 / n))
