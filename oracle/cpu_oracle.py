"""ctypes front-end of oracle/liboracle.so (plain-C restatement, TEST INFRASTRUCTURE ONLY).

What each entry restates (reference file:line):
  warp_perspective   cv2.warpPerspective as called at /root/reference/vis_homo.py:89,91 and
                     /root/reference/bev/tool/compo.py:38,46,47  (parity unpinned -- OpenCV absent)
  project_points     pts_world_bev, /root/reference/bev/rbox.py:136-151  (pinned by tests/golden)
  resize_linear_u8   cv2.resize(img, (w, h)) (INTER_LINEAR) as called at /root/reference/vis_homo.py:90  (parity unpinned)
  rbox_iou           d3d.box.box2d_iou(.., method="rbox") as called at
                     /root/reference/bev/tracker/rbox_tracker.py:87-92   (parity unpinned -- d3d absent)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

U8, F32 = 0, 1
NEAREST, LINEAR = 0, 1

_lib = None


def build(force=False):
    """gcc the oracle (a few hundred ms).  Called by __graft_entry__.build() and lazily by load()."""
    srcs = [os.path.join(_HERE, f) for f in ("warp_oracle.c", "resize_oracle.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def load():
    global _lib
    if _lib is None:
        build()  # (a no-op unless a source is newer than the library)
        lib = ctypes.CDLL(_SO)
        c = ctypes
        lib.oracle_invert3x3.argtypes = [c.c_void_p, c.c_void_p]
        lib.oracle_invert3x3.restype = c.c_int
        lib.oracle_block_width.argtypes = [c.c_int, c.c_int]
        lib.oracle_block_width.restype = c.c_int
        lib.oracle_warp_perspective.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int64, c.c_void_p, c.c_int, c.c_int,
                                                c.c_int64, c.c_int, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p,
                                                c.c_int, c.c_void_p]
        lib.oracle_warp_perspective.restype = c.c_int
        lib.oracle_warp_maps.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_void_p]
        lib.oracle_warp_maps.restype = c.c_int
        lib.oracle_bilinear_tab_i.argtypes = [c.c_void_p]
        lib.oracle_project_points_f64.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_int, c.c_void_p]
        lib.oracle_project_points_f64.restype = c.c_int
        lib.oracle_project_points_f32.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_int, c.c_void_p]
        lib.oracle_project_points_f32.restype = c.c_int
        lib.oracle_rbox_iou.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_int, c.c_int, c.c_void_p]
        lib.oracle_rbox_iou.restype = c.c_int
        lib.oracle_resize_linear_u8.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int64, c.c_void_p, c.c_int, c.c_int, c.c_int64, c.c_int]
        lib.oracle_resize_linear_u8.restype = c.c_int
        lib.oracle_rbox_iou_pair.argtypes = [c.c_void_p, c.c_void_p]
        lib.oracle_rbox_iou_pair.restype = c.c_double
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def invert3x3(M):
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(3, 3)
    out = np.empty((3, 3), np.float64)
    load().oracle_invert3x3(_p(M), _p(out))
    return out


def block_width(dst_w, dst_h):
    return load().oracle_block_width(int(dst_w), int(dst_h))


def warp_perspective(src, M, dsize, interp=LINEAR, m_is_inverse=False, border_value=None, nthreads=1,
                     want_footprint=False):
    """dst = cv2.warpPerspective(src, M, dsize) restated.  src: HxW or HxWxC uint8/float32 (contiguous rows).
    dsize = (width, height) like OpenCV.  Returns dst (and the touched-source byte map if asked)."""
    src = np.asarray(src)
    squeeze = src.ndim == 2
    s3 = src[:, :, None] if squeeze else src
    assert s3.ndim == 3 and s3.dtype in (np.uint8, np.float32), (s3.shape, s3.dtype)
    s3 = np.ascontiguousarray(s3)
    h, w, c = s3.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    dst = np.empty((dh, dw, c), s3.dtype)
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(3, 3)
    bv = None if border_value is None else np.ascontiguousarray(np.broadcast_to(np.asarray(border_value, np.float64), (c,)))
    touched = np.zeros((h, w), np.uint8) if want_footprint else None
    rc = load().oracle_warp_perspective(_p(s3), h, w, s3.strides[0], _p(dst), dh, dw, dst.strides[0], c, _p(M),
                                        int(bool(m_is_inverse)), U8 if s3.dtype == np.uint8 else F32, int(interp),
                                        None if bv is None else _p(bv), int(nthreads),
                                        None if touched is None else _p(touched))
    if rc != 0:
        raise ValueError("oracle_warp_perspective: bad arguments")
    if squeeze:
        dst = dst[:, :, 0]
    return (dst, touched) if want_footprint else dst


def footprint(src_hw, M, dsize, interp=LINEAR, m_is_inverse=False):
    """Number of distinct in-bounds source pixels referenced by any tap."""
    h, w = src_hw
    dummy = np.zeros((h, w, 1), np.uint8)
    _, touched = warp_perspective(dummy, M, dsize, interp, m_is_inverse, want_footprint=True)
    return int(touched.sum()), touched


def warp_maps(dsize, M_inv, interp=LINEAR):
    dw, dh = int(dsize[0]), int(dsize[1])
    sxy = np.empty((dh, dw, 2), np.int16)
    alpha = np.empty((dh, dw), np.int32)
    M_inv = np.ascontiguousarray(M_inv, dtype=np.float64).reshape(3, 3)
    load().oracle_warp_maps(dh, dw, _p(M_inv), int(interp), _p(sxy), _p(alpha))
    return sxy, alpha


def bilinear_tab_i():
    t = np.empty((1024, 4), np.int16)
    load().oracle_bilinear_tab_i(_p(t))
    return t


def project_points(pts, H):
    pts = np.ascontiguousarray(pts)
    assert pts.ndim == 2 and pts.shape[1] in (2, 3) and pts.dtype in (np.float32, np.float64)
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(3, 3)
    out = np.empty_like(pts)
    fn = load().oracle_project_points_f64 if pts.dtype == np.float64 else load().oracle_project_points_f32
    fn(_p(pts), _p(out), pts.shape[0], pts.shape[1], _p(H))
    return out


def rbox_iou(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty((a.shape[0], b.shape[0]), np.float64)
    load().oracle_rbox_iou(_p(a), a.shape[0], a.shape[1], _p(b), b.shape[0], b.shape[1], _p(out))
    return out


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize) for uint8 (H, W) / (H, W, C) images, default INTER_LINEAR (oracle/resize_oracle.c)."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim in (2, 3)
    c = 1 if img.ndim == 2 else img.shape[2]
    dw, dh = int(dsize[0]), int(dsize[1])
    out = np.empty((dh, dw) + (() if img.ndim == 2 else (c,)), np.uint8)
    st = load().oracle_resize_linear_u8(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(out), dh, dw, out.strides[0], c)
    if st != 0:
        raise ValueError("oracle_resize_linear_u8 failed: %d" % st)
    return out
