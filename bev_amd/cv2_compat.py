"""A `cv2`-shaped module over the MI355X path: the handful of OpenCV entry points the reference's warp path calls, with
OpenCV's argument order, defaults, flag values and return shapes, so that an integrator can write

    import bev_amd.cv2_compat as cv2          # or: sys.modules["cv2"] = bev_amd.cv2_compat, before `import bev`

and leave /root/reference/vis_homo.py:89-91, bev/homo.py:36 and bev/tool/compo.py:38,46,47 untouched.

    cv2.warpPerspective(img, H, (w, h)[, dst, flags, borderMode, borderValue])   vis_homo.py:89,91; compo.py:38,46,47
    cv2.findHomography(pts_src, pts_tgt[, method]) -> (H, mask)                  bev/homo.py:36
    cv2.perspectiveTransform(pts (N,1,2), H) -> (N,1,2)                          (OpenCV's name for pts_world_bev, bev/rbox.py:136-151)
    cv2.invert(M) -> (retval, M_inv)                                             the 3x3 step inside warpPerspective

    cv2.resize(img, (w, h)[, dst, fx, fy, interpolation])                        vis_homo.py:90 (uint8, INTER_LINEAR only)

`cv2.resize` here is OpenCV's own bilinear algorithm (sampling at (d + 0.5) * scale - 0.5, 11-bit coefficients, replicated edge, the
2 x 2 box-mean rule) as its own device kernel -- round 3 had none, because a resize routed through the WARP kernel (1/32-px
positions, constant border) returns different pixels and must not carry cv2's name.  Restated from memory like the warp: parity
unpinned.  Under `python -m bev_amd.run` a real cv2 keeps its own resize; on the fast path the "small" branch needs none
(bev_amd.warp.warp_perspective_resized folds the resize into the homography).

Pixel work runs on the GPU through libbevwarp.so (no CPU fallback); numpy images go up and come back per call, which is
what the cv2 call shape implies -- keep frames resident and use bev_amd.warp.warp_perspective / bev_amd.pipeline for
throughput.  Everything OpenCV offers beyond this list is deliberately absent: an AttributeError names what a caller
still needs from a real cv2.
"""
import numpy as np

from . import resize as _resize
from . import warp as _warp
from .homo import homo_from_pts as _homo_from_pts
from .rbox import pts_world_bev as _pts_world_bev

__version__ = "bev_amd.cv2_compat"

# flag values of OpenCV 4.x
INTER_NEAREST = 0
INTER_LINEAR = 1
WARP_INVERSE_MAP = 16
BORDER_CONSTANT = 0
DECOMP_LU = 0
RANSAC, LMEDS, RHO = 8, 4, 16


def warpPerspective(src, M, dsize, dst=None, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0):
    """uint8 / float32 images of 1-4 channels; INTER_LINEAR or INTER_NEAREST, optionally | WARP_INVERSE_MAP;
    BORDER_CONSTANT with cv::Scalar border semantics.  Bit-exact with the classic fixed-point algorithm
    (oracle/warp_oracle.c states which OpenCV code path that is)."""
    return _warp.warpPerspective(src, M, dsize, dst=dst, flags=flags, borderMode=borderMode, borderValue=borderValue)


def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR):
    """uint8 images of 1-4 channels, INTER_LINEAR (cv2.resize's default; the reference passes none).  Bit-exact with
    oracle/resize_oracle.c (classic OpenCV 3.x-4.x bilinear path, restated from memory: parity unpinned)."""
    return _resize.cv2_resize(src, dsize, dst=dst, fx=fx, fy=fy, interpolation=interpolation)


def findHomography(srcPoints, dstPoints, method=0, ransacReprojThreshold=3.0, mask=None, maxIters=2000, confidence=0.995):
    """Least-squares homography (method 0, the only one the reference uses): returns (H, mask) with H[2, 2] = 1 and an
    all-ones (N, 1) uint8 mask like OpenCV.  Points may be (N, 2) or (N, 1, 2)."""
    if method != 0:
        raise NotImplementedError("findHomography: only method=0 (least squares over all points) is implemented")
    a = np.asarray(srcPoints, dtype=np.float64).reshape(-1, 2)
    b = np.asarray(dstPoints, dtype=np.float64).reshape(-1, 2)
    H = _homo_from_pts(a, b)
    return H, np.ones((len(a), 1), dtype=np.uint8)


def perspectiveTransform(src, m):
    """(N, 1, 2) or (N, 2) points through the 3x3 m; same shape and dtype back (float32 / float64)."""
    pts = np.asarray(src)
    if pts.dtype not in (np.float32, np.float64):
        raise TypeError("perspectiveTransform: points must be float32 or float64")
    flat = pts.reshape(-1, 2).astype(np.float64)
    out = _pts_world_bev(flat, np.asarray(m, dtype=np.float64))
    return out.astype(pts.dtype).reshape(pts.shape)


def invert(src, flags=DECOMP_LU):
    """3x3 float64 inverse in OpenCV's closed-form evaluation order; (retval, inverse) with retval 0.0 and a zero matrix
    for a singular input."""
    M = np.asarray(src, dtype=np.float64)
    if M.shape != (3, 3):
        raise NotImplementedError("invert: only 3x3 matrices (the homographies of the warp path)")
    inv = _warp.invert_homography(M)
    ok = bool(np.any(inv != 0))
    return (1.0 if ok else 0.0), inv
