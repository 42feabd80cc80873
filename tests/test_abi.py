"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/bevwarp.h declares,
validates arguments before touching the device, and its host helper matches the oracle.  No GPU work."""
import ctypes
import os
import re

import numpy as np
import pytest

from bev_amd import _lib
from oracle import cpu_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_symbols():
    with open(os.path.join(ROOT, "include", "bevwarp.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(bevwarp_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 8 and "bevwarp_warp" in names
    assert sorted(_lib.SYMBOLS) == names  # the ctypes table and the header agree
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert getattr(raw, n) is not None
    assert lib.bevwarp_version() == _lib.ABI_VERSION == 7


def test_header_cites_the_reference_interfaces():
    with open(os.path.join(ROOT, "include", "bevwarp.h")) as f:
        text = f.read()
    for cite in ("vis_homo.py:89", "bev/tool/compo.py:38", "bev/rbox.py:136-151", "bev/tracker/rbox_tracker.py:87-92", "bev/rbox.py:173-219",
                 "bev/tool/rbox_tracking_BrnoCompSpeed.py:88-109", "bev/tracker/rbox_tracker.py:383-405", "vis_homo.py:90"):
        assert cite in text


def test_strerror_and_argument_validation_without_a_device(lib):
    assert lib.bevwarp_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6):
        assert len(lib.bevwarp_strerror(code)) > 4
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    warp = lib.bevwarp_warp
    far = ctypes.c_void_p(1 << 20)  # a destination that does not overlap the source
    ok_args = [one, far, 1, 8, 8, 8, 8, 3, 192, 24, 192, 24, one, 1, _lib.U8, 1, None, None]

    def call(**patch):
        a = list(ok_args)
        for k, v in patch.items():
            a[int(k[1:])] = v
        return warp(*a)

    assert call(a0=None) == -1          # null src
    assert call(a2=-1) == -1            # negative batch
    assert call(a7=5) == -2             # 5 channels
    assert call(a14=7) == -2            # unknown dtype
    assert call(a15=3) == -2            # unknown interpolation
    assert call(a9=23) == -1            # row stride shorter than a row
    assert call(a13=2) == -1            # 2 matrices for a batch of 1
    assert call(a4=40000, a9=120000, a8=960000) == -3  # source wider than 32767
    assert call(a2=0) == 0              # empty batch is a no-op
    # src and dst must not overlap (include/bevwarp.h): in place, partially, and batch-wise through the frame strides
    nan_border = ctypes.cast((ctypes.c_double * 3)(0.0, float("nan"), 0.0), ctypes.c_void_p)  # (a call that passes the overlap test fails later, on this)
    assert b"overlap" in lib.bevwarp_strerror(-6)
    assert call(a1=one) == -6
    assert call(a1=ctypes.c_void_p(16 + 191)) == -6          # the last source byte
    assert call(a1=ctypes.c_void_p(16 + 192), a16=ctypes.cast((ctypes.c_double * 3)(0.0, float("nan"), 0.0), ctypes.c_void_p)) == -4  # adjacent is fine (fails later, on the border)
    assert call(a2=4, a1=ctypes.c_void_p(16 + 3 * 192 + 100)) == -6   # inside frame 3 of a 4-frame source
    assert call(a2=4, a0=ctypes.c_void_p((1 << 20) + 3 * 192 + 191)) == -6  # source starting on the last byte of destination frame 3
    # two ROIs of ONE image side by side (equal row strides, disjoint byte columns): accepted, as cv2.warpPerspective accepts them --
    # an 8 x 8 RGB source at column 0 and an 8 x 8 destination at column 8 of a 16-pixel-wide image (row stride 48)
    base = 4096
    assert call(a0=ctypes.c_void_p(base), a1=ctypes.c_void_p(base + 24), a9=48, a11=48, a8=384, a10=384, a16=nan_border) == -4
    assert call(a0=ctypes.c_void_p(base), a1=ctypes.c_void_p(base + 23), a9=48, a11=48, a8=384, a10=384, a16=nan_border) == -6  # one byte column shared
    assert call(a0=ctypes.c_void_p(base + 24), a1=ctypes.c_void_p(base), a9=48, a11=48, a8=384, a10=384, a16=nan_border) == -4  # right half into left half
    assert call(a0=ctypes.c_void_p(base), a1=ctypes.c_void_p(base + 24), a9=48, a11=52, a8=384, a10=416, a16=nan_border) == -6  # different strides: conservative
    assert call(a2=3, a0=ctypes.c_void_p(base), a1=ctypes.c_void_p(base + 24), a9=48, a11=48, a8=384, a10=384, a16=nan_border) == -4  # batched, frame strides multiples of 48
    assert call(a2=3, a0=ctypes.c_void_p(base), a1=ctypes.c_void_p(base + 24), a9=48, a11=48, a8=400, a10=384, a16=nan_border) == -6
    # verdict tables (ABI v7): their size follows the launch geometry; the call validates like bevwarp_warp plus its own two arguments
    assert lib.bevwarp_tile_classes_bytes(32, 1080, 1920, 1024, 1024, 3, _lib.U8, 1) % 12 == 0 and lib.bevwarp_tile_classes_bytes(32, 1080, 1920, 1024, 1024, 3, _lib.U8, 1) > 0
    assert lib.bevwarp_tile_classes_bytes(0, 8, 8, 8, 8, 3, _lib.U8, 1) == 0
    assert lib.bevwarp_tile_classes_bytes(1, 8, 8, 8, 8, 5, _lib.U8, 1) == -2   # the status of the warp it describes
    classes = lib.bevwarp_warp_classes
    assert classes(*(ok_args[:17] + [None, 0, None])) == -1                     # no table
    assert classes(*(ok_args[:17] + [one, 2, None])) == -1                      # unknown mode
    assert classes(*(ok_args[:17] + [ctypes.c_void_p(18), 0, None])) == -1      # misaligned table
    assert classes(*([None] + ok_args[1:17] + [one, 1, None])) == -1            # the warp's own checks still apply
    planar = lib.bevwarp_warp_planar
    pargs = [one, ctypes.c_void_p(16 + 100), 1, 8, 8, 8, 8, 3, 192, 24, 768, 256, 32, one, 1, _lib.U8, 1, None, None, None, None]
    assert planar(*pargs) == -6
    bad_border = (ctypes.c_double * 3)(0.0, float("nan"), 0.0)
    assert call(a16=ctypes.cast(bad_border, ctypes.c_void_p)) == -4
    H = (ctypes.c_double * 9)(*[float("inf")] * 9)
    assert lib.bevwarp_project_points(one, one, 4, 2, ctypes.cast(H, ctypes.c_void_p), _lib.F64, None) == -4
    assert lib.bevwarp_project_points(one, one, 4, 5, ctypes.cast(H, ctypes.c_void_p), _lib.F64, None) == -1
    assert lib.bevwarp_rbox_iou(one, 4, 3, one, 4, 5, one, _lib.F64, None) == -1
    assert lib.bevwarp_footprint(None, 1, 8, 8, 8, 8, one, 1, 1, None) == -1
    with pytest.raises(ValueError):
        _lib.check(-1)


def test_host_inverse_matches_oracle_bit_for_bit():
    from bev_amd.warp import invert_homography
    rng = np.random.default_rng(0)
    M = rng.normal(size=(64, 3, 3)) * rng.choice([1e-3, 1.0, 1e3], size=(64, 1, 1))
    got = invert_homography(M)
    for i in range(64):
        np.testing.assert_array_equal(got[i], co.invert3x3(M[i]))
    assert not invert_homography(np.ones((3, 3))).any()  # singular -> zeros, like cv::invert
    with pytest.raises(ValueError):
        invert_homography(np.full((3, 3), np.nan))


def test_device_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from bev_amd import iou, points, warp
    with pytest.raises(ValueError):
        warp.warp_perspective(torch.zeros((8, 8, 3), dtype=torch.uint8), np.eye(3), (8, 8))
    with pytest.raises(ValueError):
        points.project_points(torch.zeros((4, 2)), np.eye(3))
    with pytest.raises(ValueError):
        iou.rbox_iou(torch.zeros((1, 5)), torch.zeros((1, 5)))


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under bev_amd/ or bev/ may reference it."""
    for pkg in ("bev_amd", "bev"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h", ".cpp")):
                    with open(os.path.join(dirpath, fn)) as f:
                        src = f.read()
                    assert "cpu_oracle" not in src and "liboracle" not in src and "from oracle" not in src, os.path.join(dirpath, fn)


def test_tracker_step_validates_before_touching_the_device(lib):
    one = ctypes.c_void_p(16)
    eye = np.eye(3)
    H = eye.ctypes.data_as(ctypes.c_void_p)
    step = lib.bevwarp_tracker_step
    ok = [one, 4, 5, one, 3, 7, H, None, 0.3, one, one, one, None, _lib.F64, None]

    def call(**patch):
        a = list(ok)
        for k, v in patch.items():
            a[int(k[1:])] = v
        return step(*a)

    assert call(a2=4) == -1                       # fewer than 5 values per detection row
    assert call(a0=None) == -1                    # null detections
    assert call(a10=None) == -1                   # m > 0 without an IoU buffer
    assert call(a13=_lib.U8) == -2                # dtype
    assert call(a1=70000) == -3                   # n > 65535
    shear = np.array([[1.0, 0, 0], [0, 2.0, 0], [0, 0, 1]])
    assert call(a6=shear.ctypes.data_as(ctypes.c_void_p)) == -1       # axes scale differently
    proj = np.array([[1.0, 0, 0], [0, 1.0, 0], [1e-3, 0, 1]])
    assert call(a6=proj.ctypes.data_as(ctypes.c_void_p)) == -1        # not affine
    nan = np.full((3, 3), np.nan)
    assert call(a6=nan.ctypes.data_as(ctypes.c_void_p)) == -4
    assert call(a7=eye.ctypes.data_as(ctypes.c_void_p)) == -1         # image centres asked for, no buffer
    assert call(a1=0) == 0                        # nothing to do: no launch, no device
    assert lib.bevwarp_rbox_transform(one, 0, 5, H, 1, one, _lib.F64, None) == 0
    assert lib.bevwarp_rbox_transform(one, 3, 4, H, 1, one, _lib.F64, None) == -1


def test_overlap_guard_has_no_false_negatives(lib):
    """The overlap guard of bevwarp_warp against brute force: 3,000 seeded random layouts (1-channel uint8, tiny images, arbitrary strides,
    1-3 frames) of a source and a destination inside one address range.  Whenever a source byte IS a destination byte the call must be
    refused (-6); it may be refused conservatively when they are not -- except for the side-by-side case the refinement exists for.
    No device is touched: calls that pass the guard fail next on a NaN border value (-4)."""
    rng = np.random.default_rng(5)
    nan_border = ctypes.cast((ctypes.c_double * 1)(float("nan")), ctypes.c_void_p)
    one = ctypes.c_void_p(16)
    accepted = refused_disjoint = 0
    for _ in range(3000):
        sw, sh, dw, dh = (int(v) for v in rng.integers(1, 7, 4))
        batch = int(rng.integers(1, 4))
        if rng.random() < 0.5:  # the refinement's class: one common row stride, frame strides multiples of it
            rs = int(rng.integers(max(sw, dw), 20))
            srs = drs = rs
            sfs, dfs = rs * int(rng.integers(sh, sh + 3)), rs * int(rng.integers(dh, dh + 3))
        else:
            srs, drs = int(rng.integers(sw, 20)), int(rng.integers(dw, 20))
            sfs, dfs = int(rng.integers(sh * srs, sh * srs + 30)), int(rng.integers(dh * drs, dh * drs + 30))
        s0, d0 = 4096 + int(rng.integers(0, 120)), 4096 + int(rng.integers(0, 120))
        sbytes = {s0 + f * sfs + r * srs + c for f in range(batch) for r in range(sh) for c in range(sw)}
        dbytes = {d0 + f * dfs + r * drs + c for f in range(batch) for r in range(dh) for c in range(dw)}
        st = lib.bevwarp_warp(ctypes.c_void_p(s0), ctypes.c_void_p(d0), batch, sh, sw, dh, dw, 1, sfs, srs, dfs, drs, one, 1, _lib.U8, 1, nan_border, None)
        assert st in (-4, -6), st
        if sbytes & dbytes:
            assert st == -6, (s0, d0, sw, sh, dw, dh, batch, srs, drs, sfs, dfs)
        elif st == -4:
            accepted += 1
        else:
            refused_disjoint += 1
    assert accepted > 300  # the guard is not simply refusing everything (bounding ranges apart, or side-by-side columns)
    # a side-by-side pair is accepted even though the bounding ranges interleave
    assert lib.bevwarp_warp(ctypes.c_void_p(4096), ctypes.c_void_p(4096 + 6), 2, 4, 6, 4, 6, 1, 64, 16, 64, 16, one, 1, _lib.U8, 1, nan_border, None) == -4
