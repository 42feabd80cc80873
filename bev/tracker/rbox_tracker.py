"""Drop-in names of /root/reference/bev/tracker/rbox_tracker.py that lie on the hot path.

    iou_batch_rbox(bb_test, bb_gt)              rbox_tracker.py:87-92    numpy (N, >=5) x (M, >=5) -> numpy (N, M)
    associate_candidates(dets, trks, thr)       the gate of :383-405     IoU matrix + `iou > thr` on the device
"""
from bev_amd.iou import iou_batch_rbox  # noqa: F401
from bev_amd.tracker_geom import tracker_geometry_step  # noqa: F401

__all__ = ["iou_batch_rbox", "tracker_geometry_step"]
