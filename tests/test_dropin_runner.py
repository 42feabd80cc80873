"""The drop-in boundary, demonstrated: scripts with the reference's own import and call lines run UNCHANGED through
`python -m bev_amd.run`, against SYNTHETIC stand-ins built in tmp_path -- a fake `cv2` module and a fake reference `bev/`
package holding only the sub-packages that are outside the hot path (`bev.io`, `bev.visualizer`, `bev.evaluator`,
`bev.tracker.rbox_tracker` with a `Sort`).  None of the reference's files is used, copied or read here: the stubs are
written below, and only the import / call LINES of /root/reference/vis_homo.py:1-9,56-63,86-89 and
/root/reference/bev/tool/rbox_tracking_BrnoCompSpeed.py:1-6 are mirrored (they are the interface under test).

CPU tests check the resolution (who serves which name); the `-m gpu` tests run the same scripts with a device and compare the
pixels / IoUs with the oracle.

Two further CPU tests at the end run only where /root/reference exists (the build container; never the GPU box, and nothing of
the reference is copied): the reference's OWN vis_homo.py, byte for byte, through the runner, and its OWN rbox_tracker.py through
the overlay."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_CV2 = '''
"""synthetic stand-in for OpenCV (test fixture): decode / resize / display names only"""
import numpy as np
__version__ = "0.0-stub"
INTER_LINEAR, INTER_NEAREST = 1, 0
CALLS = []


class VideoCapture:
    def __init__(self, path):
        self.path, self.n = path, 0

    def read(self):
        if self.n >= 2:
            return False, None
        rng = np.random.default_rng(1234 + self.n)
        self.n += 1
        return True, rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)


def warpPerspective(*a, **k):
    raise AssertionError("the stub's own warpPerspective ran: the rebinding did not happen")


def resize(img, dsize):
    CALLS.append("resize")
    ys = (np.arange(dsize[1]) * img.shape[0]) // dsize[1]
    xs = (np.arange(dsize[0]) * img.shape[1]) // dsize[0]
    return np.ascontiguousarray(img[ys][:, xs])


def imshow(*a):
    CALLS.append("imshow")


def waitKey(*a):
    return 0
'''

# the fake reference package: __init__ shaped like the reference's (star-imports of every sub-package), which must NOT run
# when the overlay is in front, a `homo` that must never be resolved, and the out-of-scope sub-packages
FAKE_REF = {
    "bev/__init__.py": '''
        from .constructor import *
        from .visualizer import *
        from .io import *
        from .evaluator import *
        from .bev import BEVWorldSpec
        from .calib import Calib
        REFERENCE_INIT_RAN = True
    ''',
    "bev/homo.py": 'raise AssertionError("the stub reference\'s bev.homo was imported: the overlay must win")\n',
    "bev/bev.py": 'class BEVWorldSpec:\n    STUB = True\n',
    "bev/calib.py": 'class Calib:\n    STUB = True\n',
    "bev/rbox.py": 'def rbox_world_bev(*a, **k):\n    raise AssertionError("stub rbox")\nxy82xywhr = rbox_world_img = rbox_world_bev\n',
    "bev/constructor/__init__.py": '__all__ = ["homo_constr"]\n',
    "bev/constructor/homo_constr.py": 'def load_calib(*a, **k):\n    raise AssertionError("stub constructor")\npreset_bspec = load_bspec = load_calib\n',
    "bev/io/__init__.py": 'from .utils import video_generator, video_parser\n__all__ = ["video_generator", "video_parser"]\n',
    "bev/io/utils.py": '''
        import cv2
        STUB_IO = True


        def video_generator(path, fps, width, height):
            return ("stub-writer", path, fps, width, height)


        def video_parser(path):
            cap = cv2.VideoCapture(path)
            i = 0
            while True:
                ok, frame = cap.read()
                if not ok:
                    return
                yield frame, i
                i += 1
    ''',
    "bev/io/rbox_io.py": 'from ..rbox import rbox_world_bev\n\n\ndef read_txt_yolo_pred(path, mode):\n    return {}\n',
    "bev/visualizer/__init__.py": '__all__ = ["homo_vis", "rbox_vis"]\n',
    "bev/visualizer/homo_vis.py": 'import cv2\n\n\ndef vis_bspec_and_calib_in_grid(img, bspec, calib=None):\n    return img\n',
    "bev/visualizer/rbox_vis.py": 'from ..rbox import xywhr2xyxy, yaw2v\n\n\ndef vis_rbox(img, rboxes, **k):\n    return img\n',
    "bev/evaluator/__init__.py": '__all__ = ["kpts_eval"]\n',
    "bev/evaluator/kpts_eval.py": 'def lin_iou(a, b, c):\n    raise AssertionError("stub lin_iou")\nlin_iou_ellipsoid = lin_iou\n',
    "bev/tracker/__init__.py": '__all__ = ["rbox_tracker"]\n',
    "bev/tracker/rbox_tracker.py": '''
        import numpy as np
        try:
            import d3d
        except ImportError:
            raise ImportError("install d3d")
        from ..evaluator.kpts_eval import lin_iou, lin_iou_ellipsoid
        STUB_TRACKER = True


        def iou_batch_rbox(bb_test, bb_gt):
            raise AssertionError("the stub's d3d-based iou_batch_rbox ran: the rebinding did not happen")


        def associate_detections_to_trackers(detections, trackers, iou_threshold=0.3):
            iou_matrix = iou_batch_rbox(detections, trackers)   # looked up in this module's globals at call time
            return iou_matrix, iou_matrix > iou_threshold


        class Sort(object):
            def __init__(self, mode="rbox", max_age=1, min_hits_init=3):
                self.mode, self.trackers = mode, np.zeros((0, 5))

            def update(self, dets):
                return associate_detections_to_trackers(dets, self.trackers)
    ''',
    "bev/tool/__init__.py": '__all__ = ["tracking_tool"]\n',
    # import lines of rbox_tracking_BrnoCompSpeed.py:1-6 (the interface under test), then a report
    "bev/tool/tracking_tool.py": '''
        from ..tracker.rbox_tracker import Sort
        from ..io.rbox_io import read_txt_yolo_pred
        from ..io.utils import video_parser, video_generator
        from ..constructor.homo_constr import preset_bspec, load_calib
        from ..rbox import rbox_world_bev, xy82xywhr, rbox_world_img
        from ..visualizer.rbox_vis import vis_rbox

        import cv2
        import os
        import numpy as np

        import argparse

        if __name__ == "__main__":
            import json, sys
            from bev.tracker import rbox_tracker
            parser = argparse.ArgumentParser()
            parser.add_argument("--out", type=str)
            parser.add_argument("--run", action="store_true")
            args = parser.parse_args()
            rep = {"Sort": Sort.__module__, "Sort_file": sys.modules[Sort.__module__].__file__, "stub_tracker": getattr(rbox_tracker, "STUB_TRACKER", False),
                   "iou": rbox_tracker.iou_batch_rbox.__module__, "rbox_world_bev": rbox_world_bev.__module__,
                   "load_calib": load_calib.__module__, "vis_rbox": vis_rbox.__module__, "video_parser": video_parser.__module__,
                   "d3d_stand_in": bool(getattr(sys.modules.get("d3d"), "__bev_amd_stand_in__", False))}
            if args.run:
                rng = np.random.default_rng(11)
                def boxes(n):
                    return np.column_stack([rng.uniform(0, 40, (n, 2)), rng.uniform(1.6, 2.2, n), rng.uniform(3.5, 6, n), rng.uniform(-np.pi, np.pi, n)])
                tracker = Sort(mode="rbox")
                tracker.trackers = boxes(37)
                dets = boxes(29)
                iou, cand = tracker.update(dets)
                np.save(args.out + ".iou.npy", iou), np.save(args.out + ".dets.npy", dets), np.save(args.out + ".trks.npy", tracker.trackers)
                rep["ran"] = True
            json.dump(rep, open(args.out, "w"))
    ''',
}

# import lines of vis_homo.py:1-9, set-up lines of :56-63, the hot loop's read + warp (:86-89); then a report
VIS_SCRIPT = '''
import argparse
import bev
import cv2
import numpy as np
import os

from bev.constructor.homo_constr import load_calib, preset_bspec, load_bspec
from bev.io.utils import video_generator
from bev.visualizer.homo_vis import vis_bspec_and_calib_in_grid
if __name__ == "__main__":
    import json, sys, traceback
    parser = argparse.ArgumentParser()
    parser.add_argument("--video-path", type=str)
    parser.add_argument("--calib-path", type=str)
    parser.add_argument("--out", type=str)
    args = parser.parse_args()

    video = cv2.VideoCapture(args.video_path)
    calib = load_calib("BrnoCompSpeed", args.calib_path)
    bspec = load_bspec("BrnoCompSpeed", 6.1, calib)
    H_world_img = calib.gen_H_world_img()
    H_world_bev = bspec.gen_H_world_bev()
    H_bev_img = np.linalg.inv(H_world_bev).dot(H_world_img)

    rep = {"bev_file": bev.__file__, "bev_init_of_reference_ran": hasattr(bev, "REFERENCE_INIT_RAN"), "Calib": bev.Calib.__module__,
           "load_calib": load_calib.__module__, "video_generator": video_generator.__module__, "io_utils_file": sys.modules["bev.io.utils"].__file__,
           "vis": vis_bspec_and_calib_in_grid.__module__, "warpPerspective": cv2.warpPerspective.__module__, "VideoCapture": cv2.VideoCapture.__module__,
           "cv2_file": cv2.__file__, "resize": cv2.resize.__module__, "bev_io_via_attr": bev.io.utils.STUB_IO, "dsize": [bspec.u_size, bspec.v_size]}
    ret, img = video.read()
    try:
        bev_img = cv2.warpPerspective(img, H_bev_img, (bspec.u_size, bspec.v_size))
        img = vis_bspec_and_calib_in_grid(bev_img, bspec)
        np.save(args.out + ".bev.npy", bev_img), np.save(args.out + ".H.npy", H_bev_img)
        rep["warped"] = True
    except Exception as e:
        rep["warped"] = False
        rep["warp_error"] = "%s: %s" % (type(e).__name__, e)
        rep["warp_error_files"] = [f.filename for f in traceback.extract_tb(e.__traceback__)]
    json.dump(rep, open(args.out, "w"))
'''


def build_stubs(tmp_path, golden):
    stubs = tmp_path / "site"
    stubs.mkdir()
    (stubs / "cv2.py").write_text(FAKE_CV2)
    ref = tmp_path / "ref"
    for rel, body in FAKE_REF.items():
        p = ref / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(body))
    (ref / "vis_script.py").write_text(VIS_SCRIPT)  # next to the fake reference's bev/, like vis_homo.py
    calib = tmp_path / "system_dubska_optimal_calib.json"
    calib.write_text(json.dumps(golden["brno_file"]["json"]))
    return stubs, ref, calib


def run_runner(args, stubs, cwd, extra_path=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, str(stubs)] + [str(p) for p in extra_path])
    env.pop("BEVWARP_LIB", None)
    r = subprocess.run([sys.executable, "-m", "bev_amd.run"] + args, cwd=str(cwd), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r


def check_vis_report(rep, ref, stubs):
    assert os.path.samefile(rep["bev_file"], os.path.join(ROOT, "bev", "__init__.py")) and not rep["bev_init_of_reference_ran"]
    assert rep["Calib"] == "bev_amd.calib" and rep["load_calib"] == "bev_amd.constructor.homo_constr"  # the hot path's names are ours
    assert rep["video_generator"] == "bev.io.utils" and os.path.samefile(rep["io_utils_file"], str(ref / "bev" / "io" / "utils.py"))  # the stub bev.io resolved
    assert rep["vis"] == "bev.visualizer.homo_vis" and rep["bev_io_via_attr"] is True
    assert rep["warpPerspective"] == "bev_amd.cv2_compat"  # cv2.warpPerspective landed in bev_amd
    assert rep["VideoCapture"] == "cv2" and rep["resize"] == "cv2" and os.path.samefile(rep["cv2_file"], str(stubs / "cv2.py"))  # the rest of cv2 untouched
    assert rep["dsize"] == [384, 768] or len(rep["dsize"]) == 2


def test_vis_homo_shaped_script_runs_unchanged_through_the_runner(tmp_path, golden):
    stubs, ref, calib = build_stubs(tmp_path, golden)
    out = tmp_path / "vis.json"
    run_runner([str(ref / "vis_script.py"), "--video-path", "none.avi", "--calib-path", str(calib), "--out", str(out)], stubs, tmp_path)
    rep = json.loads(out.read_text())
    check_vis_report(rep, ref, stubs)
    if not rep["warped"]:  # no device here: the call must have died INSIDE bev_amd (upload of the frame), not in the stub
        assert any(f.startswith(os.path.join(ROOT, "bev_amd")) for f in rep["warp_error_files"]), rep
        assert "stub" not in rep["warp_error"]


def test_tracker_tool_imports_resolve_through_the_overlay(tmp_path, golden):
    stubs, ref, _ = build_stubs(tmp_path, golden)
    out = tmp_path / "trk.json"
    run_runner(["--reference", str(ref), "-m", "bev.tool.tracking_tool", "--out", str(out)], stubs, tmp_path)
    rep = json.loads(out.read_text())
    assert rep["Sort"] == "bev.tracker.rbox_tracker" and rep["stub_tracker"] is True  # the reference's class, executed into the overlay module
    assert os.path.samefile(rep["Sort_file"], os.path.join(ROOT, "bev", "tracker", "rbox_tracker.py"))
    assert rep["iou"] == "bev_amd.iou" and rep["rbox_world_bev"] == "bev_amd.rbox" and rep["load_calib"] == "bev_amd.constructor.homo_constr"
    assert rep["vis_rbox"] == "bev.visualizer.rbox_vis" and rep["video_parser"] == "bev.io.utils"
    assert rep["d3d_stand_in"] is True  # d3d is absent from this image: the stand-in carries box2d_iou(method="rbox") only


def test_patch_mode_on_a_reference_package_that_is_in_front(tmp_path, golden):
    """No overlay: the (stub) reference's own `bev` is the package on sys.path and bev_amd.patch.install() rebinds the two
    third-party calls in it -- cv2.warpPerspective now, iou_batch_rbox when the tracker module gets imported."""
    stubs, ref, _ = build_stubs(tmp_path, golden)
    (ref / "bev" / "homo.py").write_text("def homo_from_KRt(*a, **k):\n    return 'stub'\n")
    code = textwrap.dedent('''
        import sys, json
        import bev_amd.patch as patch
        cv2 = patch.install()
        import cv2 as again
        from bev.tracker import rbox_tracker
        rep = {"bev_init_ran": False, "warp": cv2.warpPerspective.__module__, "same": cv2 is again, "cap": cv2.VideoCapture.__module__,
               "iou": rbox_tracker.iou_batch_rbox.__module__, "Sort": rbox_tracker.Sort.__module__, "tracker_file": rbox_tracker.__file__}
        patch.uninstall()
        rep["warp_after_uninstall"] = cv2.warpPerspective.__module__
        rep["iou_after_uninstall"] = rbox_tracker.iou_batch_rbox.__module__
        print(json.dumps(rep))
    ''')
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([str(ref), str(stubs), ROOT])  # the reference's bev shadows the overlay
    (ref / "bev" / "__init__.py").write_text("from .bev import BEVWorldSpec\nfrom .calib import Calib\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["warp"] == "bev_amd.cv2_compat" and rep["same"] and rep["cap"] == "cv2"
    assert rep["iou"] == "bev_amd.iou" and rep["Sort"] == "bev.tracker.rbox_tracker"
    assert os.path.samefile(rep["tracker_file"], str(ref / "bev" / "tracker" / "rbox_tracker.py"))
    assert rep["warp_after_uninstall"] == "cv2" and rep["iou_after_uninstall"] == "bev.tracker.rbox_tracker"


def test_overlay_without_a_reference_names_what_is_missing():
    code = "import bev\nfrom bev.tracker.rbox_tracker import iou_batch_rbox\ntry:\n    bev.io\nexcept AttributeError as e:\n    print('A', e)\n" \
           "import bev.tracker.rbox_tracker as t\ntry:\n    t.Sort\nexcept AttributeError as e:\n    print('B', e)\n"
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd="/", env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "A bev.io is outside the MI355X hot path" in r.stdout and "B bev.tracker.rbox_tracker.Sort is outside the MI355X hot path" in r.stdout


def test_runner_cv2_shim_registers_the_compat_module(tmp_path):
    script = tmp_path / "s.py"
    script.write_text("import cv2, json, sys\nprint(json.dumps({'cv2': cv2.__name__, 'has_imread': hasattr(cv2, 'imread'), 'argv': sys.argv[1:]}))\n")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-m", "bev_amd.run", "--cv2-shim", str(script), "--flag", "1"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    try:
        import cv2  # noqa: F401
        has_real = not getattr(cv2, "__version__", "").startswith("bev_amd")
    except ImportError:
        has_real = False
    assert rep["argv"] == ["--flag", "1"]
    if not has_real:
        assert rep["cv2"] == "bev_amd.cv2_compat" and rep["has_imread"] is False


@pytest.mark.gpu
def test_vis_homo_shaped_script_on_the_device_matches_the_oracle(tmp_path, golden):
    from oracle import cpu_oracle as co
    stubs, ref, calib = build_stubs(tmp_path, golden)
    out = tmp_path / "vis.json"
    run_runner([str(ref / "vis_script.py"), "--video-path", "none.avi", "--calib-path", str(calib), "--out", str(out)], stubs, tmp_path)
    rep = json.loads(out.read_text())
    check_vis_report(rep, ref, stubs)
    assert rep["warped"], rep
    bev_img, H = np.load(str(out) + ".bev.npy"), np.load(str(out) + ".H.npy")
    frame = np.random.default_rng(1234).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)  # what the stub VideoCapture decoded
    np.testing.assert_array_equal(bev_img, co.warp_perspective(frame, H, tuple(rep["dsize"]), 1))


@pytest.mark.gpu
def test_tracker_tool_on_the_device_uses_the_hip_iou(tmp_path, golden):
    from oracle import cpu_oracle as co
    stubs, ref, _ = build_stubs(tmp_path, golden)
    out = tmp_path / "trk.json"
    run_runner(["--reference", str(ref), "-m", "bev.tool.tracking_tool", "--out", str(out), "--run"], stubs, tmp_path)
    rep = json.loads(out.read_text())
    assert rep["ran"] and rep["iou"] == "bev_amd.iou"
    iou, dets, trks = (np.load(str(out) + s) for s in (".iou.npy", ".dets.npy", ".trks.npy"))
    np.testing.assert_allclose(iou, co.rbox_iou(dets, trks), rtol=0, atol=1e-12)


REFERENCE = "/root/reference"  # present in the build container only (never on the GPU box; nothing of it is copied)

REAL_CV2_STUB = '''
"""inert stand-in for OpenCV (harness side): the decode names vis_homo.py touches before / around its warp"""
import numpy as np
CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT = 3, 4
INTER_LINEAR = 1


class VideoCapture:
    def __init__(self, path):
        self.n = 0

    def get(self, prop):
        return {3: 1920.0, 4: 1080.0}[prop]

    def isOpened(self):
        return True

    def read(self):
        if self.n >= 1:
            return False, None
        self.n += 1
        return True, np.random.default_rng(1234).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)

    def release(self):
        pass


def warpPerspective(*a, **k):
    raise AssertionError("the stub's own warpPerspective ran: the rebinding did not happen")
'''


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "vis_homo.py")), reason="the reference is only present in the build container")
def test_the_reference_vis_homo_itself_runs_unchanged_through_the_runner(tmp_path, golden):
    """/root/reference/vis_homo.py, byte for byte, through `python -m bev_amd.run` with an inert cv2 stand-in for the decoder:
    `import bev` is the overlay, `bev.io.utils` / `bev.visualizer.homo_vis` are the reference's own files, load_calib /
    load_bspec / Calib.scale are bev_amd's -- and print the values the reference itself prints for this calibration (SURVEY.md
    8(c): center_world [4.67193551 1.32144698 1.], small [4.64784341 1.31490458 1.]) -- and cv2.warpPerspective (vis_homo.py:89)
    lands in bev_amd.  Without a device the call stops there, inside bev_amd/warp.py; with one it would return the BEV frame."""
    site = tmp_path / "site"
    site.mkdir()
    (site / "cv2.py").write_text(REAL_CV2_STUB)
    calib = tmp_path / "system_dubska_optimal_calib.json"
    calib.write_text(json.dumps(golden["brno_file"]["json"]))
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, str(site)])
    r = subprocess.run([sys.executable, "-m", "bev_amd.run", os.path.join(REFERENCE, "vis_homo.py"), "--video-tag", "6_left", "--video-path", "none.avi",
                        "--calib-path", str(calib), "--no-show", "--no-grid"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "center_world [4.67193551 1.32144698 1." in out and "center_world_small [4.64784341 1.31490458 1." in out, out[-2000:]
    assert "the stub's own warpPerspective ran" not in out
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, out[-2000:]
    else:  # the frame's upload is where a machine without a device stops: inside bev_amd, reached through cv2.warpPerspective
        assert r.returncode != 0 and os.path.join("bev_amd", "warp.py") in out and "in warpPerspective" in out, out[-2000:]


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "bev", "tracker", "rbox_tracker.py")), reason="the reference is only present in the build container")
def test_the_reference_sort_loads_through_the_overlay_with_the_hip_iou_bound(tmp_path):
    """`from bev.tracker.rbox_tracker import Sort` (rbox_tracking_BrnoCompSpeed.py:1) against the reference's own file: its
    third-party imports are stubbed (filterpy, skimage; d3d gets bev_amd's stand-in), `Sort` and
    `associate_detections_to_trackers` are the reference's objects and look up bev_amd's iou_batch_rbox."""
    site = tmp_path / "site"
    (site / "filterpy").mkdir(parents=True)
    (site / "skimage").mkdir()
    (site / "cv2.py").write_text(REAL_CV2_STUB)
    (site / "filterpy" / "__init__.py").write_text("")
    (site / "filterpy" / "kalman.py").write_text("class KalmanFilter:\n    def __init__(self, *a, **k):\n        pass\n\n\nclass ExtendedKalmanFilter(KalmanFilter):\n    pass\n")
    (site / "skimage" / "__init__.py").write_text("from . import io\n")
    (site / "skimage" / "io.py").write_text("")
    code = textwrap.dedent('''
        import json, sys
        import matplotlib
        _use = matplotlib.use
        matplotlib.use = lambda *a, **k: _use("Agg")   # the reference asks for TkAgg at import: no display here
        import bev_amd.patch as patch
        patch.install()
        from bev.tracker.rbox_tracker import Sort, associate_detections_to_trackers
        import bev.tracker.rbox_tracker as rt
        print(json.dumps({"Sort": Sort.__module__, "file": rt.__shadowed_file__, "err": str(rt.__shadowed_error__),
                          "iou": associate_detections_to_trackers.__globals__["iou_batch_rbox"].__module__,
                          "d3d": bool(getattr(sys.modules.get("d3d"), "__bev_amd_stand_in__", False))}))
    ''')
    env = dict(os.environ, MPLBACKEND="Agg")
    env["PYTHONPATH"] = os.pathsep.join([ROOT, str(site), REFERENCE])
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["Sort"] == "bev.tracker.rbox_tracker" and rep["file"] == os.path.join(REFERENCE, "bev", "tracker", "rbox_tracker.py") and rep["err"] == "None"
    assert rep["iou"] == "bev_amd.iou"


def test_star_exported_names_resolve_through_the_overlay(tmp_path, golden):
    """names the reference's `from .io import *` / `from .visualizer import *` bind on `bev` (bev.video_generator,
    bev.homo_vis, bev.kpts_eval, ...) must be readable as attributes of the overlay package too."""
    stubs, ref, _ = build_stubs(tmp_path, golden)
    code = textwrap.dedent('''
        import json
        import bev
        rep = {"video_generator": bev.video_generator.__module__, "homo_vis": bev.homo_vis.__name__, "kpts_eval": bev.kpts_eval.__name__,
               "homo_constr": bev.homo_constr.load_calib.__module__, "init_ran": hasattr(bev, "REFERENCE_INIT_RAN"), "bev_file": bev.__file__}
        try:
            bev.nothing_of_the_kind
            rep["missing"] = "no error"
        except AttributeError:
            rep["missing"] = "AttributeError"
        print(json.dumps(rep))
    ''')
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, str(stubs), str(ref)])  # overlay first, the (stub) reference behind it
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["video_generator"] == "bev.io.utils" and rep["homo_vis"] == "bev.visualizer.homo_vis" and rep["kpts_eval"] == "bev.evaluator.kpts_eval"
    assert rep["homo_constr"] == "bev_amd.constructor.homo_constr"  # a star-exported name the overlay owns stays ours
    assert not rep["init_ran"] and os.path.samefile(rep["bev_file"], os.path.join(ROOT, "bev", "__init__.py")) and rep["missing"] == "AttributeError"


def test_unknown_names_stay_attribute_errors_when_a_reference_subpackage_cannot_import(tmp_path, golden):
    """`hasattr(bev, x)` must answer False, not raise: a co-installed reference whose `bev.io` needs a cv2 that is not there makes the
    star-export search skip that sub-package, while `bev.io` itself keeps raising the real reason."""
    import importlib.util
    if importlib.util.find_spec("cv2") is not None:
        pytest.skip("a cv2 is importable here: the stub reference's bev.io would import")
    stubs, ref, _ = build_stubs(tmp_path, golden)
    (stubs / "cv2.py").unlink()  # no cv2 at all: the stub reference's bev/io/utils.py imports it at the top
    code = textwrap.dedent('''
        import json
        import bev
        rep = {"has_typo": hasattr(bev, "no_such_name"), "has_homo_constr": hasattr(bev, "homo_constr"), "has_video_generator": hasattr(bev, "video_generator")}
        try:
            bev.io
            rep["io"] = "imported"
        except ImportError as e:
            rep["io"] = type(e).__name__ + ":" + str(getattr(e, "name", ""))
        print(json.dumps(rep))
    ''')
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, str(stubs), str(ref)])
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep == {"has_typo": False, "has_homo_constr": True, "has_video_generator": False, "io": "ModuleNotFoundError:cv2"}
