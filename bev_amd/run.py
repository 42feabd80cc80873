"""Run an UNMODIFIED script of the reference on the MI355X path.

    python -m bev_amd.run [--reference DIR] [--cv2-shim] vis_homo.py --video-tag 6_left --no-small ...
    python -m bev_amd.run [--reference DIR] [--cv2-shim] -m bev.tool.rbox_tracking_BrnoCompSpeed --video-tag ...

What it does, in this one process (runpy -- never an exec of a process that has touched the GPU):

  1. puts this repository's root at the front of sys.path, so `import bev` (vis_homo.py:2) resolves to the overlay package:
     `bev.Calib`, `bev.homo`, `bev.constructor.homo_constr` ... are the MI355X path's, `bev.io`, `bev.visualizer`,
     `bev.tracker.rbox_tracker.Sort` fall through to the reference's package;
  2. puts the script's own directory behind it, as `python script.py` would (that is where the reference keeps its `bev/`:
     vis_homo.py sits next to it), plus `--reference DIR` when the reference lives elsewhere;
  3. bev_amd.patch.install(): `cv2.warpPerspective` (vis_homo.py:89,91) and `iou_batch_rbox` (rbox_tracker.py:87-92) land
     in libbevwarp.so; `cv2.VideoCapture`, `cv2.resize`, `cv2.imshow`, the x264 writer stay OpenCV's;
  4. runs the script as `__main__` with its own argv.

`--cv2-shim`: on a machine without OpenCV, register bev_amd.cv2_compat as `cv2` (enough for scripts that only warp).
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # holds bev/ (the overlay) and bev_amd/

USAGE = "usage: python -m bev_amd.run [--reference DIR] [--cv2-shim] (SCRIPT | -m MODULE) [args ...]"


def parse(argv):
    opts = {"reference": [], "cv2_shim": False, "module": None, "script": None, "rest": []}
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "--reference":
            if i + 1 >= len(argv):
                raise SystemExit(USAGE)
            opts["reference"].append(argv[i + 1])
            i += 2
        elif a.startswith("--reference="):
            opts["reference"].append(a.split("=", 1)[1])
            i += 1
        elif a == "--cv2-shim":
            opts["cv2_shim"] = True
            i += 1
        elif a == "-m":
            if i + 1 >= len(argv):
                raise SystemExit(USAGE)
            opts["module"], opts["rest"] = argv[i + 1], argv[i + 2:]
            return opts
        elif a in ("-h", "--help"):
            raise SystemExit(__doc__)
        else:
            opts["script"], opts["rest"] = a, argv[i + 1:]
            return opts
    raise SystemExit(USAGE)


def set_paths(script, reference_dirs):
    """sys.path for the run: overlay root first, then the script's directory (python's own rule), then --reference."""
    tail = [os.path.abspath(d) for d in reference_dirs]
    if script is not None:
        tail.insert(0, os.path.dirname(os.path.abspath(script)))
    for d in [ROOT] + tail:
        while d in sys.path:
            sys.path.remove(d)
    sys.path[0:0] = [ROOT] + tail
    stale = sys.modules.get("bev")
    if stale is not None and not os.path.abspath(getattr(stale, "__file__", "") or "").startswith(os.path.join(ROOT, "bev") + os.sep):
        for k in [k for k in sys.modules if k == "bev" or k.startswith("bev.")]:  # a reference `bev` imported before the overlay was in front
            del sys.modules[k]


def main(argv=None):
    opts = parse(sys.argv[1:] if argv is None else list(argv))
    set_paths(opts["script"], opts["reference"])
    from bev_amd import patch
    patch.install(shim_missing_cv2=opts["cv2_shim"])
    if opts["module"] is not None:
        sys.argv = [opts["module"]] + opts["rest"]
        runpy.run_module(opts["module"], run_name="__main__", alter_sys=True)
    else:
        sys.argv = [opts["script"]] + opts["rest"]
        runpy.run_path(opts["script"], run_name="__main__")


if __name__ == "__main__":
    main()
