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
# ... and the order in which it star-imports them (/root/reference/bev/__init__.py:1-6): a star-import binds every name of the
# sub-package's __all__ -- sub-modules (bev.homo_constr, bev.homo_vis, bev.homo_io, bev.rbox_cvt, bev.io_vis, bev.compo, bev.kpts_eval)
# and functions (bev.video_generator, bev.read_txt_to_dict, ...) -- as an attribute of `bev` itself
_STAR = ("constructor", "visualizer", "io", "converter", "tool", "evaluator")
_NOT_HERE = ("bev.%s is outside the MI355X hot path and comes from the reference's `bev` package: put the reference on sys.path "
             "behind this overlay (python -m bev_amd.run does)")


def _subpackage(name):
    import importlib
    try:
        return importlib.import_module(__name__ + "." + name)
    except ModuleNotFoundError as e:
        if e.name != __name__ + "." + name:
            raise  # the sub-package exists and one of ITS imports is missing (cv2, tqdm, ...): say so
        return None


def __getattr__(name):
    if name in _LAZY:
        mod = _subpackage(name)
        if mod is None:
            raise AttributeError(_NOT_HERE % name)
        return mod
    if not name.startswith("__"):
        # a name the reference's star-imports would have bound: look it up in the sub-packages' __all__, in the reference's order
        import importlib
        for sub in _STAR:
            try:
                mod = _subpackage(sub)
            except ImportError:
                mod = None  # a sub-package whose own imports fail (no cv2, no tqdm ...) cannot supply the name: `hasattr(bev, x)` must get
                            # an AttributeError, never an ImportError; `bev.<sub-package>` itself still raises the real reason
            if mod is None or name not in getattr(mod, "__all__", ()):
                continue
            if hasattr(mod, name):
                value = getattr(mod, name)
            else:  # a sub-module named in __all__: `from .sub import *` imports it
                try:
                    value = importlib.import_module(mod.__name__ + "." + name)
                except ModuleNotFoundError as e:
                    if e.name != mod.__name__ + "." + name:
                        raise
                    continue
            globals()[name] = value
            return value
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
