"""Batched tracker geometry as one device step.

The reference's per-frame tracking loop (bev/tool/rbox_tracking_BrnoCompSpeed.py:88-109 with the association
front-end of bev/tracker/rbox_tracker.py:383-405) moves detections from the BEV raster to the world plane
(`rbox_world_bev`), scores them against the trackers' predicted boxes (`iou_batch_rbox` -> d3d), thresholds the
scores, and projects box centres into the image (`rbox_world_img`).  Those are tens of numpy calls per frame on
tens of boxes; here they are a handful of launches on device tensors with no host synchronisation: the homography
is validated on the host once, boxes stay on the GPU, the IoU matrix comes from the HIP kernel.  The Hungarian
assignment and the per-track Kalman filters stay on the host (out of scope)."""
import numpy as np
import torch

from .iou import rbox_iou


def _similarity_terms(H):
    """Host-side checks and constants of rbox_world_bev (rbox.py:173-219): H must be a similarity."""
    H = np.asarray(H, dtype=np.float64)
    H = H / H[2, 2]
    assert abs(H[2, 0]) + abs(H[2, 1]) < 1e-5, "H must be affine (a similarity between the BEV raster and the world)"
    scale, scale_1 = np.hypot(H[0, 0], H[0, 1]), np.hypot(H[1, 0], H[1, 1])
    assert abs(scale - scale_1) < 1e-5, "H must scale both axes equally"
    return H, float(scale)


def rbox_world_bev_device(rbox_src, H, src):
    """(n, >=5) CUDA tensor of [x, y, w, h, yaw] between the BEV ("bev": yaw = atan2(u, v)) and the world ("world":
    yaw = atan2(y, x)) through the similarity H -- rbox.py:173-219 / rbox_torch.py:123-168 without their device->host
    assertions.  Returns (n, 5)."""
    assert src in ("bev", "world")
    H, scale = _similarity_terms(H)
    r = rbox_src[:, 4]
    # yaw2v (rbox.py:20-36): bev yaw is measured from the v axis, world yaw from the x axis
    vx, vy = (torch.sin(r), torch.cos(r)) if src == "bev" else (torch.cos(r), torch.sin(r))
    tx = H[0, 0] * vx + H[0, 1] * vy
    ty = H[1, 0] * vx + H[1, 1] * vy
    r_tgt = torch.atan2(ty, tx) if src == "bev" else torch.atan2(tx, ty)  # v2yaw of the TARGET convention
    x, y = rbox_src[:, 0], rbox_src[:, 1]
    return torch.stack([H[0, 0] * x + H[0, 1] * y + H[0, 2], H[1, 0] * x + H[1, 1] * y + H[1, 2],
                        rbox_src[:, 2] * scale, rbox_src[:, 3] * scale, r_tgt], dim=1)


def tracker_geometry_step(dets_bev, trks_world, H_world_bev, iou_threshold=0.3, H_img_world=None, device="cuda"):
    """dets_bev (n, >=5) detections in BEV pixels, trks_world (m, >=5) predicted tracker boxes in the world.
    Returns a dict of device tensors: dets_world (n, 5), iou (n, m), candidates (n, m) bool = iou > iou_threshold
    (the gate of rbox_tracker.py:395-405), and, when H_img_world is given, dets_img (n, 2) image pixels of the box
    centres (rbox_world_img, rbox.py:221-226).  No host synchronisation."""
    def dev(x):
        return x.to(device) if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x)).to(device)
    dets_bev, trks_world = dev(dets_bev), dev(trks_world)
    dets_world = rbox_world_bev_device(dets_bev, H_world_bev, "bev")
    iou = rbox_iou(dets_world, trks_world[:, :5].to(dets_world.dtype))
    out = {"dets_world": dets_world, "iou": iou, "candidates": iou > iou_threshold}
    if H_img_world is not None:
        from .points import project_points
        out["dets_img"] = project_points(dets_world[:, :2].contiguous(), H_img_world)
    return out
