#!/usr/bin/env python3
"""The kernels beside the warp, launched N times each for rocprofv3 (tools/prof_geom.sh): project_points_kernel<float,2> and
<double,2> (configs[2]: 1e7 points, buffers rotated past the Infinity Cache), rbox_iou_kernel<double> and
tracker_step_kernel<double> (configs[4]: 512 x 512), warp_composite<3> (f3: 1080p + 1080p + mask -> 1024^2), composite_kernel
(composite_reg_img on 1024^2 x 3), and one u8 warp of the composite's destination for scale.  GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd import warp  # noqa: E402
from bev_amd.compo import composite_bev_img, composite_reg_img  # noqa: E402
from bev_amd.homo import homo_from_KRt  # noqa: E402
from bev_amd.iou import rbox_iou  # noqa: E402
from bev_amd.points import project_points  # noqa: E402
from bev_amd.tracker_geom import tracker_geometry_step  # noqa: E402
from tests import workloads as wl  # noqa: E402

N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)
H = np.array([[0.02, -0.001, -3.0], [0.0004, 0.05, -20.0], [1e-5, 0.0009, 0.4]])
N = 10_000_000
for tdt, esz in ((torch.float32, 4), (torch.float64, 8)):
    nbuf = max(2, int(np.ceil(600e6 / (N * 2 * esz * 2))))
    gen = torch.Generator(device=dev).manual_seed(7)
    ins = [(torch.rand((N, 2), dtype=torch.float64, device=dev, generator=gen) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)).to(tdt) for _ in range(nbuf)]
    outs = [torch.empty_like(x) for x in ins]
    for i in range(N_IT):
        project_points(ins[i % nbuf], H, out=outs[i % nbuf])
    torch.cuda.synchronize()
    del ins, outs
    torch.cuda.empty_cache()


def boxes(n):
    return np.stack([rng.uniform(0, 100, n), rng.uniform(0, 100, n), rng.uniform(1.6, 2.2, n), rng.uniform(3.5, 6, n), rng.uniform(-np.pi, np.pi, n)], axis=1)


a, b = torch.from_numpy(boxes(512)).to(dev), torch.from_numpy(boxes(512)).to(dev)
for _ in range(N_IT):
    rbox_iou(a, b)
H_world_bev = np.array([[0.0, 0.0625, -10.0], [-0.0625, 0.0, 40.0], [0, 0, 1.0]])
H_img_world = np.linalg.inv(H)
dets = torch.from_numpy(np.column_stack([rng.uniform(0, 1024, (512, 2)), rng.uniform(25, 35, 512), rng.uniform(56, 96, 512), rng.uniform(-np.pi, np.pi, 512)])).to(dev)
trks = torch.from_numpy(np.column_stack([rng.uniform(-10, 54, 512), rng.uniform(-24, 40, 512), rng.uniform(1.6, 2.2, 512), rng.uniform(3.5, 6, 512), rng.uniform(-np.pi, np.pi, 512)])).to(dev)
buf = tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world)
for _ in range(N_IT):
    tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world, out=buf)
torch.cuda.synchronize()

K = np.array([[1200.0, 0, 959.5], [0, 1190.0, 539.5], [0, 0, 1.0]])
c, s_ = np.cos(0.9), np.sin(0.9)
RT = np.array([[1, 0, 0, 0.0], [0, c, -s_, 2.0], [0, s_, c, 14.0], [0, 0, 0, 1.0]])
H_world2bev = np.array([[0.0, 24.0, 512.0], [-24.0, 0.0, 900.0], [0.0, 0.0, 1.0]])
H_img2world_fix = np.linalg.inv(homo_from_KRt(K, Rt_homo=RT)) @ np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1]])
sets = [[torch.from_numpy(wl.frame(3 * j + i, 1080, 1920, np.uint8)).to(dev) for i in range(3)] for j in range(4)]
Hb = H_world2bev.dot(H_img2world_fix)
Hc = H_world2bev.dot(np.linalg.inv(homo_from_KRt(K, Rt_homo=RT)))
outs = [torch.empty((1024, 1024, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
for i in range(N_IT):
    bg, fg, mask = sets[i % 4]
    composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, K, RT, 1024, 1024)
torch.cuda.synchronize()
for i in range(N_IT):
    bg, fg, mask = sets[i % 4]
    warp.warp_perspective(bg, Hb, (1024, 1024), out=outs[i % 4])  # one u8 warp of the composite's destination, for scale
torch.cuda.synchronize()
w3 = [warp.warp_perspective(sets[0][0], Hb, (1024, 1024)), warp.warp_perspective(sets[0][1], Hc, (1024, 1024)), warp.warp_perspective(sets[0][2], Hc, (1024, 1024))]
for _ in range(N_IT):
    composite_reg_img(*w3)
torch.cuda.synchronize()
print("prof_geom done")
