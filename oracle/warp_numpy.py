"""Independent numpy twin of oracle/warp_oracle.c -- TEST INFRASTRUCTURE ONLY.

A second, vectorised restatement of cv2.warpPerspective (call sites
/root/reference/vis_homo.py:89,91; /root/reference/bev/tool/compo.py:38,46,47) written
separately from the C oracle so the two can check each other.  Same status: **parity
unpinned** against OpenCV itself (absent here); semantics follow OpenCV 4.x imgwarp.cpp
(WarpPerspectiveInvoker + remap with fixed-point maps, BORDER_CONSTANT).
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
NEAREST, LINEAR = 0, 1
_I32_MIN, _I32_MAX = -(2 ** 31), 2 ** 31 - 1


def invert3x3(S):
    """cv::invert closed form for 3x3 doubles (cofactors times 1/det)."""
    S = np.asarray(S, np.float64).reshape(3, 3)
    a, b, c, d, e, f, g, h, i = S.ravel().tolist()
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    if det == 0.0:
        return np.zeros((3, 3))
    r = 1.0 / det
    return np.array([[(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r],
                     [(f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r],
                     [(d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r]])


def block_width(dst_w, dst_h):
    bh0 = min(16, dst_h)
    return min(1024 // bh0, dst_w)


def _round_clamped(v):
    out = np.where(np.isnan(v), float(_I32_MAX), np.clip(v, float(_I32_MIN), float(_I32_MAX)))
    return np.rint(out).astype(np.int64)


def fixed_point_maps(dsize, Minv, interp):
    """(sx, sy, fx, fy) int arrays of shape (dst_h, dst_w).  For nearest fx = fy = 0."""
    dw, dh = dsize
    M = np.asarray(Minv, np.float64).ravel()
    bw0 = block_width(dw, dh)
    x = np.arange(dw)
    bx = (x // bw0) * bw0
    x1 = (x - bx).astype(np.float64)[None, :]
    bx = bx.astype(np.float64)[None, :]
    y = np.arange(dh, dtype=np.float64)[:, None]
    X0 = (M[0] * bx + M[1] * y) + M[2]
    Y0 = (M[3] * bx + M[4] * y) + M[5]
    W0 = (M[6] * bx + M[7] * y) + M[8]
    W = W0 + M[6] * x1
    num = 1.0 if interp == NEAREST else float(INTER_TAB_SIZE)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        Wr = np.where(W != 0, num / np.where(W != 0, W, 1.0), 0.0)
        X = _round_clamped((X0 + M[0] * x1) * Wr)
        Y = _round_clamped((Y0 + M[3] * x1) * Wr)
    if interp == NEAREST:
        sx, sy = X, Y
        fx = np.zeros_like(X)
        fy = np.zeros_like(Y)
    else:
        sx, sy = X >> INTER_BITS, Y >> INTER_BITS
        fx, fy = X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)
    sx = np.clip(sx, -32768, 32767)
    sy = np.clip(sy, -32768, 32767)
    return sx, sy, fx, fy


def warp_perspective(src, M, dsize, interp=LINEAR, m_is_inverse=False, border_value=0.0):
    src = np.asarray(src)
    squeeze = src.ndim == 2
    s3 = src[:, :, None] if squeeze else src
    h, w, c = s3.shape
    Minv = np.asarray(M, np.float64).reshape(3, 3) if m_is_inverse else invert3x3(M)
    sx, sy, fx, fy = fixed_point_maps(dsize, Minv, interp)
    bv = np.broadcast_to(np.asarray(border_value, np.float64), (c,))
    if s3.dtype == np.uint8:
        cval = np.clip(np.rint(bv), 0, 255).astype(np.uint8)
    else:
        cval = bv.astype(np.float32)

    def tap(ix, iy):
        inside = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
        v = s3[np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)]
        return np.where(inside[..., None], v, cval[None, None, :])

    if interp == NEAREST:
        out = tap(sx, sy).astype(s3.dtype)
    elif s3.dtype == np.uint8:
        wx1, wy1 = fx.astype(np.int64), fy.astype(np.int64)
        wx0, wy0 = 32 - wx1, 32 - wy1
        # integer table: (32-fy)(32-fx)*32 ... (the {32767,0,0,1} entry at fx=fy=0 yields identical bytes)
        w00, w01, w10, w11 = wy0 * wx0 * 32, wy0 * wx1 * 32, wy1 * wx0 * 32, wy1 * wx1 * 32
        acc = (tap(sx, sy).astype(np.int64) * w00[..., None] + tap(sx + 1, sy).astype(np.int64) * w01[..., None] +
               tap(sx, sy + 1).astype(np.int64) * w10[..., None] + tap(sx + 1, sy + 1).astype(np.int64) * w11[..., None])
        out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    else:
        s = np.float32(1.0 / INTER_TAB_SIZE)
        tx1, ty1 = fx.astype(np.float32) * s, fy.astype(np.float32) * s
        tx0, ty0 = np.float32(1) - tx1, np.float32(1) - ty1
        w00, w01, w10, w11 = ty0 * tx0, ty0 * tx1, ty1 * tx0, ty1 * tx1
        out = ((tap(sx, sy) * w00[..., None] + tap(sx + 1, sy) * w01[..., None]) + tap(sx, sy + 1) * w10[..., None]) + \
            tap(sx + 1, sy + 1) * w11[..., None]
        out = out.astype(np.float32)
    if interp != NEAREST:  # remapBilinear's "fully outside" path stores the border value itself, not a blend of four of them
        all_out = (sx >= w) | (sx + 1 < 0) | (sy >= h) | (sy + 1 < 0)
        out = np.where(all_out[..., None], cval[None, None, :], out).astype(s3.dtype)
    return out[:, :, 0] if squeeze else out
