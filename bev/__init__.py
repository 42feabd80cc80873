"""Drop-in `bev` package: the reference's import surface (/root/reference/bev/__init__.py:1-9,
`bev.BEVWorldSpec`, `bev.Calib`, `bev.homo`, `bev.bev`, `bev.calib`, `bev.rbox`, `bev.rbox_torch`,
`bev.frozen_class`, `bev.constructor.homo_constr`) served by bev_amd, plus `bev.warp` -- the HIP
replacement for the cv2.warpPerspective call of vis_homo.py:89."""
from . import constructor  # noqa: F401
from .bev import BEVWorldSpec
from .calib import Calib

__all__ = ["BEVWorldSpec", "Calib", "constructor"]
