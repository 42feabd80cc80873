"""Profiling aid (GPU box): the tracker step and the IoU matrix on 512 x 512 boxes, once with overlapping pairs and once with every pair\ndisjoint -- under rocprofv3 --kernel-trace the difference is the cost of the clip of the few surviving pairs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bev_amd.iou import rbox_iou
from bev_amd.tracker_geom import tracker_geometry_step
rng = np.random.default_rng(11)
H_world_bev = np.array([[0.0, 0.0625, -10.0], [-0.0625, 0.0, 40.0], [0, 0, 1.0]])
H_img_world = np.linalg.inv(np.array([[0.02, -0.001, -3.0], [0.0004, 0.05, -20.0], [1e-5, 0.0009, 0.4]]))
dets = torch.from_numpy(np.column_stack([rng.uniform(0, 1024, (512, 2)), rng.uniform(25, 35, 512), rng.uniform(56, 96, 512), rng.uniform(-np.pi, np.pi, 512)])).cuda()
trk = np.column_stack([rng.uniform(-10, 54, 512), rng.uniform(-24, 40, 512), rng.uniform(1.6, 2.2, 512), rng.uniform(3.5, 6, 512), rng.uniform(-np.pi, np.pi, 512)])
for label, off in (("overlapping", 0.0), ("disjoint", 5000.0)):
    trks = torch.from_numpy(trk + np.array([off, 0, 0, 0, 0])).cuda()
    buf = tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world)
    for _ in range(50):
        tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world, out=buf)
        rbox_iou(dets, trks)
    torch.cuda.synchronize()
    print(label, int(buf["candidates"].sum()), int((buf["iou"] > 0).sum()))
