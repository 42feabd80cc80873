"""Overlay mechanics of the drop-in `bev` package.

The reference's `bev` (/root/reference/bev/__init__.py:1-9) star-imports six sub-packages; this repository re-implements only
the ones on the warp path: `bev.homo`, `bev.bev`, `bev.calib`, `bev.rbox`, `bev.rbox_torch`,
`bev.frozen_class`, `bev.constructor`, `bev.tool.compo`, `bev.tracker.rbox_tracker.iou_batch_rbox`.  Everything else --
`bev.io`, `bev.visualizer`, `bev.converter`, `bev.evaluator`, `bev.tool.io_vis`, the tracker tools, `Sort` and its Kalman
filters -- is host I/O, drawing and per-track logic that stays the reference's.  So the shipped `bev/` is an OVERLAY:

  * every package `__init__` of the overlay extends its `__path__` with the same-named package found further down
    `sys.path` (a co-installed reference), ours first: `bev.homo` resolves here, `bev.io.utils` there
    (`extend(__path__, __name__)`);
  * a module the overlay shadows only PARTLY (`bev.tracker.rbox_tracker`: one function of a 646-line file is on the path)
    executes the reference's file of the same name into its own namespace and then rebinds the names it owns
    (`exec_shadowed`), so `Sort`, `KalmanBoxTracker`, `associate_detections_to_trackers` are the reference's objects and the
    `iou_batch_rbox` they look up at call time is the HIP one.

Nothing here imports or needs the reference: without one on `sys.path` the overlay is simply the hot-path subset.
"""
import os
import pkgutil
import sys
import types


def extend(path, name):
    """`__path__` of an overlay package: its own directory first, then every same-named package directory on `sys.path`
    (for a sub-package: on the parent's `__path__`).  pkgutil.extend_path does exactly this for regular packages."""
    return pkgutil.extend_path(path, name)


def shadowed_file(modname):
    """Path of `<leaf>.py` for module `modname` in a LATER portion of its parent package's `__path__` than the overlay's
    own (i.e. the co-installed reference's file this overlay module stands in front of), or None."""
    pkg, _, leaf = modname.rpartition(".")
    parent = sys.modules.get(pkg)
    own = getattr(sys.modules.get(modname), "__file__", None)
    for d in list(getattr(parent, "__path__", []) or []):
        cand = os.path.join(d, leaf + ".py")
        if os.path.isfile(cand) and not (own and os.path.exists(own) and os.path.samefile(cand, own)):
            return cand
    return None


def exec_shadowed(modname, namespace):
    """Execute the reference's file that overlay module `modname` shadows into `namespace` (the overlay module's globals).
    Returns the file's path, or None when no reference is co-installed.  An ImportError raised by the reference's own
    third-party imports (filterpy, skimage, ...) is kept in `namespace['__shadowed_error__']` instead of propagating, so
    that the hot-path names of the overlay module stay importable on machines without the tracker's dependencies."""
    path = shadowed_file(modname)
    namespace["__shadowed_file__"] = path
    namespace["__shadowed_error__"] = None
    if path is None:
        return None
    with open(path, "rb") as f:
        code = compile(f.read(), path, "exec")
    try:
        exec(code, namespace)
    except ImportError as e:
        namespace["__shadowed_error__"] = e
    return path


def missing_name(modname, name, namespace):
    """The AttributeError a partly shadowing overlay module raises for a name only the reference defines."""
    err = namespace.get("__shadowed_error__")
    if err is not None:
        return AttributeError("%s.%s is defined by the co-installed reference (%s), whose import failed: %s: %s"
                              % (modname, name, namespace.get("__shadowed_file__"), type(err).__name__, err))
    if namespace.get("__shadowed_file__") is None:
        return AttributeError("%s.%s is outside the MI355X hot path and comes from the reference's `bev` package: put the "
                              "reference on sys.path behind this overlay (python -m bev_amd.run does)" % (modname, name))
    return AttributeError("module %r has no attribute %r" % (modname, name))


def _swap_wh(boxes):
    """[x, y, w, h, yaw, ...] rows -> [x, y, h, w, yaw]: the same rectangle turned a quarter about its own centre."""
    if not hasattr(boxes, "dim"):  # (anything but a torch tensor: numpy semantics)
        import numpy as np
        boxes = np.asarray(boxes)
    return boxes[:, [0, 1, 3, 2, 4]]


def remove_d3d_stand_in():
    """Take the stand-in out of sys.modules again (bev_amd.patch.uninstall); a real d3d is never touched."""
    mod = sys.modules.get("d3d")
    if mod is not None and getattr(mod, "__bev_amd_stand_in__", False):
        sys.modules.pop("d3d", None)
        sys.modules.pop("d3d.box", None)
        return True
    return False


def ensure_d3d():
    """The reference's tracker imports `d3d` only for `d3d.box.box2d_iou(a, b, method="rbox")`
    (/root/reference/bev/tracker/rbox_tracker.py:40-47, :92), the call the HIP IoU kernel replaces.  When d3d is not
    installed, register a minimal module of that name whose `box.box2d_iou` is the HIP kernel, so the reference's file still
    imports.  A real d3d is never touched.

    Convention: the kernel (like bev.rbox) lays a box's length `h` along its yaw; the reference turns both yaws by pi/2 before it
    calls d3d (rbox_tracker.py:88-91), i.e. d3d lays `w` along the yaw.  The stand-in therefore swaps w and h -- the exact form
    of "yaw - pi/2" -- so that stand_in(a + pi/2, b + pi/2) == bev_amd.iou.iou_batch_rbox(a, b), the identity the rebound
    tracker function relies on (d3d itself is absent here: its convention is assumed, parity unpinned)."""
    try:
        import d3d  # noqa: F401
        return False
    except ImportError:
        pass
    from . import iou as _iou
    d3d = types.ModuleType("d3d")
    d3d.__doc__ = "stand-in registered by bev_amd.overlay.ensure_d3d: only box.box2d_iou(method='rbox'), on the MI355X kernel"
    box = types.ModuleType("d3d.box")

    def box2d_iou(boxes1, boxes2, method="box"):
        if method != "rbox":
            raise NotImplementedError("bev_amd's d3d stand-in implements method='rbox' only")
        return _iou.iou_any(_swap_wh(boxes1), _swap_wh(boxes2))

    box.box2d_iou = box2d_iou
    d3d.box = box
    d3d.__bev_amd_stand_in__ = True
    sys.modules["d3d"], sys.modules["d3d.box"] = d3d, box
    return True
