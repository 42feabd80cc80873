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
