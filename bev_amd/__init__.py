"""bev_amd -- MI355X-native BEV homography-warp path behind the `bev` package API.

Host algebra (numpy, float64): homo, calib.Calib, bevspec.BEVWorldSpec, rbox, rbox_torch,
constructor.homo_constr -- mirrors of the reference's bev.homo / bev.calib / bev.bev / ...
Device path (HIP, gfx950, through the C ABI of include/bevwarp.h): warp.warp_perspective,
points.project_points, iou.rbox_iou.  The device modules import lazily so the host algebra
works on machines without the built library; the device entry points raise loudly if it is missing.
"""
from .bevspec import BEVWorldSpec
from .calib import Calib
from . import homo, rbox

__all__ = ["BEVWorldSpec", "Calib", "homo", "rbox"]
__version__ = "0.1.0"
