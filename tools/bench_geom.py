#!/usr/bin/env python3
"""BASELINE.json configs[2] and configs[4]: point projection (1e7 points) and 512 x 512 rotated IoU on one MI355X.
Prints one JSON object per line.  GPU box:  python tools/bench_geom.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd.iou import rbox_iou  # noqa: E402
from bev_amd.points import project_points  # noqa: E402


def timeit(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e-3
    return float(t.mean()), float(t.min())


H = np.array([[0.02, -0.001, -3.0], [0.0004, 0.05, -20.0], [1e-5, 0.0009, 0.4]])
N = 10_000_000
rng = np.random.default_rng(7)
for dt, tdt in (("f32", torch.float32), ("f64", torch.float64)):
    # rotate over enough distinct buffers to stay out of the 256 MB Infinity Cache
    esz = 4 if dt == "f32" else 8
    nbuf = max(2, int(np.ceil(600e6 / (N * 2 * esz * 2))))
    ins = [torch.from_numpy(rng.uniform(0, [1920, 1080], (N, 2))).to(tdt).cuda() for _ in range(nbuf)]
    outs = [torch.empty_like(x) for x in ins]
    k = [0]

    def step():
        i = k[0] % nbuf
        project_points(ins[i], H, out=outs[i])
        k[0] += 1

    mean, mn = timeit(step)
    nbytes = N * 2 * esz * 2
    print(json.dumps({"config": "configs[2]: 1e7 (u,v) points through a 3x3 H, %s" % dt, "ms": round(mean * 1e3, 4), "ms_min": round(mn * 1e3, 4),
                      "Gpts_per_s": round(N / mean / 1e9, 2), "algorithmic_GB_per_s": round(nbytes / mean / 1e9, 1),
                      "frac_of_8TBs": round(nbytes / mean / 8e12, 3)}))
    del ins, outs
    torch.cuda.empty_cache()


def boxes(n):
    return np.stack([rng.uniform(0, 100, n), rng.uniform(0, 100, n), rng.uniform(1.6, 2.2, n), rng.uniform(3.5, 6, n), rng.uniform(-np.pi, np.pi, n)], axis=1)


for dt, tdt in (("f32", torch.float32), ("f64", torch.float64)):
    a = torch.from_numpy(boxes(512)).to(tdt).cuda()
    b = torch.from_numpy(boxes(512)).to(tdt).cuda()
    mean, mn = timeit(lambda: rbox_iou(a, b), n=200)
    print(json.dumps({"config": "configs[4] IoU half: 512 x 512 rotated-box IoU, %s" % dt, "us": round(mean * 1e6, 2), "us_min": round(mn * 1e6, 2),
                      "Mpairs_per_s": round(512 * 512 / mean / 1e6, 1)}))

# configs[4], whole: one camera frame's BEV warp plus the tracker's geometry step on 512 detections x 512 tracks
# (per-camera sequential state: "replicas only" across GPUs)
from bev_amd import warp as _warp  # noqa: E402
from bev_amd.tracker_geom import tracker_geometry_step  # noqa: E402
from tests import workloads as _wl  # noqa: E402

Hk = _wl.keystone_H(1920, 1080, 1024, 1024)
frames = [torch.from_numpy(_wl.frame(i, 1080, 1920, np.uint8)).cuda() for i in range(8)]
outs = [torch.empty((1024, 1024, 3), dtype=torch.uint8, device="cuda") for _ in range(8)]
minv = _warp.device_inverse(Hk, frames[0].device)
H_world_bev = np.array([[0.0, 0.0625, -10.0], [-0.0625, 0.0, 40.0], [0, 0, 1.0]])
dets_bev = torch.from_numpy(np.column_stack([rng.uniform(0, 1024, (512, 2)), rng.uniform(25, 35, 512), rng.uniform(56, 96, 512), rng.uniform(-np.pi, np.pi, 512)])).cuda()
trks = torch.from_numpy(boxes(512)).cuda()
kk = [0]


def tracker_step():
    i = kk[0] % 8
    _warp.warp_perspective(frames[i], None, (1024, 1024), out=outs[i], M_inv_device=minv)
    tracker_geometry_step(dets_bev, trks, H_world_bev, 0.3)
    kk[0] += 1


mean, mn = timeit(tracker_step, n=100)
print(json.dumps({"config": "configs[4]: one 1080p -> 1024^2 uint8 bilinear warp + tracker geometry step (512 dets x 512 tracks, float64)",
                  "us": round(mean * 1e6, 1), "us_min": round(mn * 1e6, 1), "frames_per_s": round(1 / mean, 1)}))

# the same step captured once in a HIP graph and replayed (the launch-bound form a per-camera loop would run)
try:
    from tools.graphed_step import GraphedStep  # noqa: E402
    kk[0] = 0
    g = GraphedStep(tracker_step)
    mean, mn = timeit(g.replay, n=100)
    print(json.dumps({"config": "configs[4]: the same step replayed from a HIP graph", "us": round(mean * 1e6, 1), "us_min": round(mn * 1e6, 1),
                      "frames_per_s": round(1 / mean, 1)}))
except Exception as e:  # graph capture is an optimisation of the harness, not of the path
    print(json.dumps({"config": "configs[4]: HIP graph replay", "error": "%s: %s" % (type(e).__name__, e)}))

# PCIe-inclusive rate of the numpy drop-in (bev.warp.warpPerspective: host frame up, BEV frame down, pageable memory)
import time as _time  # noqa: E402

img = _wl.frame(0, 1080, 1920, np.uint8)
for _ in range(3):
    _warp.warpPerspective(img, Hk, (1024, 1024))
t0 = _time.perf_counter()
n_rep = 20
for _ in range(n_rep):
    res = _warp.warpPerspective(img, Hk, (1024, 1024))
dt_host = (_time.perf_counter() - t0) / n_rep
print(json.dumps({"config": "numpy drop-in warpPerspective, one 1080p uint8 frame -> 1024^2, host to host (PCIe-inclusive)",
                  "ms": round(dt_host * 1e3, 3), "Mpix_per_s": round(1024 * 1024 / dt_host / 1e6, 1)}))

# the per-camera frame loop of vis_homo.py:85-91 on resident frames, one launch per frame, back to back
kk[0] = 0


def eager_frame():
    i = kk[0] % 8
    _warp.warp_perspective(frames[i], None, (1024, 1024), out=outs[i], M_inv_device=minv)
    kk[0] += 1


torch.cuda.synchronize()
t0 = _time.perf_counter()
for _ in range(2000):
    eager_frame()
torch.cuda.synchronize()
dt1 = (_time.perf_counter() - t0) / 2000
print(json.dumps({"config": "one resident 1080p uint8 frame -> 1024^2 per call, back to back", "us_per_frame": round(dt1 * 1e6, 1),
                  "frames_per_s": round(1 / dt1, 1)}))
