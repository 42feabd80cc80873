"""oracle/resize_numpy.py -- TEST INFRASTRUCTURE ONLY: an independently written numpy twin of oracle/resize_oracle.c
(cv2.resize(img, (w, h)) with the default INTER_LINEAR on uint8 images, the reference's call at /root/reference/vis_homo.py:90;
classic OpenCV 3.x-4.x bilinear path restated from memory -- parity unpinned, see the C file's header).  Vectorised over whole
images where the C version walks pixels; the two must agree bit for bit (tests/test_oracle_resize.py)."""
import numpy as np

_SCALE = 2048


def _axis(src_n, dst_n, clamp):
    scale = 1.0 / (float(dst_n) / float(src_n))  # two roundings, as cv::resize computes it
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    one_tap = np.zeros(dst_n, dtype=bool)
    if clamp:
        low = s < 0
        s[low], f[low] = 0, np.float32(0)
        far = s + 1 >= src_n
        first = int(np.argmax(far)) if far.any() else dst_n
        one_tap[first:] = True  # every column from the first clamped one on reads a single tap
        end = s >= src_n - 1
        s[end], f[end] = src_n - 1, np.float32(0)
    c0 = np.clip(np.rint((np.float32(1) - f) * np.float32(_SCALE)), -32768, 32767).astype(np.int64)
    c1 = np.clip(np.rint(f * np.float32(_SCALE)), -32768, 32767).astype(np.int64)
    return s, c0, c1, one_tap


def resize_linear_u8(img, dsize):
    img = np.asarray(img)
    assert img.dtype == np.uint8
    src = img if img.ndim == 3 else img[:, :, None]
    sh, sw, _ = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if abs(1.0 / (dw / sw) - 2.0) < np.finfo(np.float64).eps and abs(1.0 / (dh / sh) - 2.0) < np.finfo(np.float64).eps:
        s = src.astype(np.int64)
        out = (s[0:2 * dh:2, 0:2 * dw:2] + s[0:2 * dh:2, 1:2 * dw:2] + s[1:2 * dh:2, 0:2 * dw:2] + s[1:2 * dh:2, 1:2 * dw:2] + 2) >> 2
    else:
        sx, a0, a1, one = _axis(sw, dw, True)
        sy, b0, b1, _ = _axis(sh, dh, False)
        wide = src.astype(np.int64)
        right = np.minimum(sx + 1, sw - 1)  # (never read with a non-zero weight where it is clamped)
        h = wide[:, sx, :] * a0[None, :, None] + np.where(one[None, :, None], 0, wide[:, right, :] * a1[None, :, None])
        h = np.where(one[None, :, None], wide[:, sx, :] * _SCALE, h)
        r0, r1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
        out = (((b0[:, None, None] * (h[r0] >> 4)) >> 16) + ((b1[:, None, None] * (h[r1] >> 4)) >> 16) + 2) >> 2
    out = out.astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]
