"""Rebind the third-party calls of the reference's hot path to the MI355X library, in an unmodified reference.

    import bev_amd.patch; bev_amd.patch.install()

  cv2.warpPerspective                         <- bev_amd.cv2_compat.warpPerspective     vis_homo.py:89,91; bev/tool/compo.py:38,46,47
  bev.tracker.rbox_tracker.iou_batch_rbox     <- bev_amd.iou.iou_batch_rbox             bev/tracker/rbox_tracker.py:87-92 (d3d.box.box2d_iou)

Nothing else of cv2 is touched: `cv2.VideoCapture`, `cv2.resize`, `cv2.imshow`, `cv2.VideoWriter`, `cv2.findHomography` stay
OpenCV's.  The tracker function is rebound in the module object -- `associate_detections_to_trackers` (:383-405) looks the
name up in its module's globals at call time -- whether the module is already imported or gets imported later (a meta-path
hook patches it right after its body has run).  This works both when the reference's own `bev` package is the one on
sys.path and when this repository's overlay `bev/` is in front of it (the overlay's rbox_tracker is already bound; the hook
then finds nothing to do).

`python -m bev_amd.run <script> [args]` calls install() and runs a script of the reference unchanged.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_TRACKER = "bev.tracker.rbox_tracker"
_state = {"cv2": None, "cv2_original": None, "cv2_registered": False, "hook": None, "tracker_originals": {}}


def _warp_entry():
    from . import cv2_compat
    return cv2_compat.warpPerspective


def patch_cv2(cv2_module=None, shim_missing_cv2=False):
    """Rebind `warpPerspective` of the cv2 module (imported here unless given).  With no OpenCV installed: raise, or -- when
    `shim_missing_cv2` -- register bev_amd.cv2_compat under the name `cv2` (warpPerspective, findHomography,
    perspectiveTransform, invert and the flag constants only; a script then fails at the first cv2 name the warp path does not
    own, with an AttributeError that names it)."""
    cv2 = cv2_module
    if cv2 is None:
        try:
            cv2 = importlib.import_module("cv2")
        except ImportError:
            if not shim_missing_cv2:
                raise ImportError("bev_amd.patch: OpenCV (cv2) is not importable; pass shim_missing_cv2=True (runner: --cv2-shim) to "
                                  "register bev_amd.cv2_compat as `cv2` for the names of the warp path") from None
            from . import cv2_compat
            sys.modules["cv2"] = cv2_compat
            _state["cv2"], _state["cv2_registered"] = cv2_compat, True
            return cv2_compat
    entry = _warp_entry()
    if getattr(cv2, "warpPerspective", None) is not entry:
        _state["cv2"], _state["cv2_original"] = cv2, getattr(cv2, "warpPerspective", None)
        cv2.warpPerspective = entry
    return cv2


def patch_tracker(module):
    """Rebind `iou_batch_rbox` in an imported (reference) bev.tracker.rbox_tracker module."""
    from .iou import iou_batch_rbox
    if getattr(module, "iou_batch_rbox", None) is not iou_batch_rbox:
        _state["tracker_originals"][id(module)] = (module, getattr(module, "iou_batch_rbox", None))
        module.iou_batch_rbox = iou_batch_rbox
    return module


class _TrackerHook(importlib.abc.MetaPathFinder):
    """Patches bev.tracker.rbox_tracker right after its body has executed, whoever imports it and whenever."""

    def __init__(self):
        self._busy = False

    def find_spec(self, name, path, target=None):
        if name != _TRACKER or self._busy:
            return None
        self._busy = True
        try:
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            spec = None
        finally:
            self._busy = False
        if spec is None or spec.loader is None or not hasattr(spec.loader, "exec_module"):
            return None
        loader, orig_exec = spec.loader, spec.loader.exec_module

        def exec_module(module):
            orig_exec(module)
            patch_tracker(module)

        loader.exec_module = exec_module
        return spec


def install(cv2_module=None, shim_missing_cv2=False, tracker=True, d3d_stand_in=True):
    """Apply both rebindings (idempotent).  Returns the cv2 module that was patched.  No GPU call is made here: the first
    one happens inside the first patched call."""
    cv2 = patch_cv2(cv2_module, shim_missing_cv2)
    if tracker:
        if d3d_stand_in:
            from .overlay import ensure_d3d
            ensure_d3d()
        mod = sys.modules.get(_TRACKER)
        if mod is not None:
            patch_tracker(mod)
        elif _state["hook"] is None:
            _state["hook"] = _TrackerHook()
            sys.meta_path.insert(0, _state["hook"])
    return cv2


def uninstall():
    """Undo install(): the original cv2.warpPerspective and iou_batch_rbox are put back, the d3d stand-in (if one was
    registered) is removed from sys.modules."""
    if _state["cv2_registered"]:
        if sys.modules.get("cv2") is _state["cv2"]:
            del sys.modules["cv2"]
    elif _state["cv2"] is not None and _state["cv2_original"] is not None:
        _state["cv2"].warpPerspective = _state["cv2_original"]
    _state["cv2"], _state["cv2_original"], _state["cv2_registered"] = None, None, False
    for module, orig in _state["tracker_originals"].values():
        if orig is not None:
            module.iou_batch_rbox = orig
    _state["tracker_originals"].clear()
    if _state["hook"] is not None:
        if _state["hook"] in sys.meta_path:
            sys.meta_path.remove(_state["hook"])
        _state["hook"] = None
    from .overlay import remove_d3d_stand_in
    remove_d3d_stand_in()  # (the restored iou_batch_rbox must not find the stand-in under d3d's name)
