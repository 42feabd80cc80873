"""The resize oracle (oracle/resize_oracle.c: cv2.resize(img, (w, h)), INTER_LINEAR, uint8 -- the reference's call at
/root/reference/vis_homo.py:90) against analytic known answers and its independently written numpy twin.  PARITY UNPINNED:
OpenCV is absent from this image and the reference holds no fixture for this call; the algorithm is restated from memory of
OpenCV 3.x-4.x resize.cpp (11-bit coefficients, (dx + 0.5) * scale - 0.5 sampling, replicated edge, the 2 x 2 -> INTER_AREA rule)."""
import numpy as np
import pytest

from oracle import cpu_oracle as co
from oracle import resize_numpy as rn
from tests import workloads as wl


def test_identity_and_constant_images():
    img = wl.frame(3, 37, 53, np.uint8)
    np.testing.assert_array_equal(co.resize_linear_u8(img, (53, 37)), img)  # same size: every output pixel is its source pixel
    for v in (0, 1, 127, 254, 255):
        flat = np.full((19, 23, 3), v, np.uint8)
        for dsize in ((7, 5), (46, 38), (100, 3), (23, 40)):
            assert (co.resize_linear_u8(flat, dsize) == v).all()  # the 11-bit coefficients of a pixel sum to 2048: constants survive


def test_integer_magnification_phases():
    """x2 magnification samples at source positions d / 2 - 1 / 4: phases 3/4 and 1/4, the first and last columns replicate the edge.
    With source values that are multiples of 4 the quarter blends are integers, so the fixed point must return them exactly."""
    row = np.array([0, 64, 128, 192, 252], np.uint8)
    img = np.repeat(row[None, :], 3, 0)
    out = co.resize_linear_u8(img, (10, 3))
    np.testing.assert_array_equal(out[0], [0, 16, 48, 80, 112, 144, 176, 207, 237, 252])
    # (207 = (192 * 3 + 252) / 4, 237 = (192 + 3 * 252) / 4: exact quarters again)
    col = co.resize_linear_u8(img.T.copy(), (3, 10))  # the same along the other axis: rows are clipped, not zero-weighted
    np.testing.assert_array_equal(col[:, 0], out[0])
    # x3: phases 1/3 from (d + 0.5) / 3 - 0.5: float coefficients 682.67 -> 683 and 1365.33 -> 1365
    out3 = co.resize_linear_u8(np.array([[0, 255]], np.uint8), (6, 1))[0]
    assert out3[0] == 0 and out3[-1] == 255 and list(out3) == sorted(out3)
    h = [0 * 2048, 0 * 1365 + 255 * 683, 0 * 683 + 255 * 1365, 0 * 0 + 255 * 2048]  # columns 1..4: sx = 0, fx = 0, 1/3, 2/3, then sx = 1 clamped
    exp = [(((2048 * (v >> 4)) >> 16) + 2) >> 2 for v in h]
    assert list(out3[1:5]) == exp


def test_two_by_two_decimation_is_the_box_mean():
    img = wl.frame(5, 40, 64, np.uint8)
    s = img.astype(np.int64)
    exp = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    np.testing.assert_array_equal(co.resize_linear_u8(img, (32, 20)), exp.astype(np.uint8))
    # ... but only when BOTH scales are exactly 2
    assert not np.array_equal(co.resize_linear_u8(img, (32, 21))[:20], exp.astype(np.uint8))


@pytest.mark.parametrize("shape,dsize", [((1080, 1920, 3), (852, 480)), ((720, 1280, 3), (640, 360)), ((37, 53, 1), (100, 64)), ((64, 48, 4), (31, 17)),
                                         ((5, 7, 3), (1, 1)), ((1, 1, 3), (9, 4)), ((480, 852, 2), (1920, 1080)), ((33, 65), (17, 9))])
def test_numpy_twin_agrees_bit_for_bit(shape, dsize):
    img = wl.frame(11, shape[0], shape[1], np.uint8, shape[2] if len(shape) == 3 else 1)
    if len(shape) == 2:
        img = img[:, :, 0]
    got = co.resize_linear_u8(img, dsize)
    assert got.shape[:2] == (dsize[1], dsize[0]) and got.dtype == np.uint8
    np.testing.assert_array_equal(got, rn.resize_linear_u8(img, dsize))


def test_the_small_branch_of_vis_homo():
    """vis_homo.py:73-78,90: 1920 x 1080 -> 852 x 480.  The result is a low-pass of the frame (within a grey level of the float bilinear
    sample at (dx + 0.5) * scale - 0.5) and stays inside the range of its four taps."""
    img = wl.frame(2, 1080, 1920, np.uint8)
    out = co.resize_linear_u8(img, (852, 480)).astype(np.float64)
    sx, sy = 1.0 / (852 / 1920), 1.0 / (480 / 1080)
    fx = np.clip((np.arange(852) + 0.5) * sx - 0.5, 0, 1919)
    fy = np.clip((np.arange(480) + 0.5) * sy - 0.5, 0, 1079)
    x0, y0 = np.minimum(np.floor(fx).astype(int), 1918), np.minimum(np.floor(fy).astype(int), 1078)
    ax, ay = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - ax) + f[y0][:, x0 + 1] * ax) * (1 - ay) + (f[y0 + 1][:, x0] * (1 - ax) + f[y0 + 1][:, x0 + 1] * ax) * ay
    assert np.abs(out - ref).max() <= 1.0
