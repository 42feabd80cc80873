"""`bev.tracker.rbox_tracker` -- the reference's module with its IoU front-end on the MI355X.

    iou_batch_rbox(bb_test, bb_gt)          /root/reference/bev/tracker/rbox_tracker.py:87-92   numpy (N, >=5) x (M, >=5) -> (N, M)
    tracker_geometry_step(...)              the device half of one tracking step (:383-405 + rbox_tracking_BrnoCompSpeed.py:88-109)

With a reference `bev/` co-installed behind this overlay, the reference's own file is executed into this namespace first:
`Sort`, `KalmanBoxTracker`, `associate_detections_to_trackers`, `linear_assignment` ... are the reference's objects,
unchanged, and the `iou_batch_rbox` that `associate_detections_to_trackers` looks up at call time (:393-394) is the one
bound below -- one HIP launch over all N x M pairs instead of d3d.box.box2d_iou.  Without a reference only the two names
above exist."""
from bev_amd import overlay as _overlay

_overlay.ensure_d3d()  # the reference imports d3d for the one call this module replaces
_overlay.exec_shadowed(__name__, globals())

from bev_amd.iou import iou_batch_rbox  # noqa: E402,F401  (after the reference's definitions: this binding wins)
from bev_amd.tracker_geom import tracker_geometry_step  # noqa: E402,F401


def __getattr__(name):
    raise _overlay.missing_name(__name__, name, globals())
