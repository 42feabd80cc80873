"""Drop-in `bev` package: an OVERLAY over the reference's package of the same name (/root/reference/bev/__init__.py:1-9).

What lies on the MI355X hot path is served from bev_amd -- `bev.BEVWorldSpec`, `bev.Calib`, `bev.homo`, `bev.bev`,
`bev.calib`, `bev.rbox`, `bev.rbox_torch`, `bev.frozen_class`, `bev.constructor.homo_constr`, `bev.tool.compo`,
`bev.tracker.rbox_tracker.iou_batch_rbox`, plus `bev.warp` / `bev.cv2_compat` (the HIP replacement of the
cv2.warpPerspective call of vis_homo.py:89).  Everything else (`bev.io`, `bev.visualizer`, `bev.converter`,
`bev.evaluator`, `bev.tool.io_vis`, the tracker tools, `Sort`) falls through to a reference `bev/` found further down
sys.path: this package's `__path__` is extended with it, ours first.  `python -m bev_amd.run vis_homo.py ...` sets that
up and leaves the script untouched."""
from bev_amd.overlay import extend as _extend

__path__ = _extend(__path__, __name__)

from . import constructor  # noqa: E402,F401
from .bev import BEVWorldSpec  # noqa: E402
from .calib import Calib  # noqa: E402

__all__ = ["BEVWorldSpec", "Calib", "constructor"]

# the sub-packages the reference's __init__ star-imports eagerly (they pull cv2, tqdm, ...): here they load on first use
_LAZY = ("visualizer", "io", "converter", "tool", "evaluator", "tracker")


def __getattr__(name):
    if name in _LAZY:
        import importlib
        try:
            return importlib.import_module(__name__ + "." + name)
        except ModuleNotFoundError as e:
            if e.name != __name__ + "." + name:
                raise
            raise AttributeError("bev.%s is outside the MI355X hot path and comes from the reference's `bev` package: put the "
                                 "reference on sys.path behind this overlay (python -m bev_amd.run does)" % name) from None
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
