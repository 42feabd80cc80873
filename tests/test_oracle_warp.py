"""Pin the CPU oracle (oracle/warp_oracle.c).  OpenCV is absent, so warpPerspective parity is UNPINNED
against cv2 itself; the restated semantics are pinned by analytic known answers and by
an independently written numpy twin (oracle/warp_numpy.py)."""
import numpy as np
import pytest

from oracle import cpu_oracle as co
from oracle import warp_numpy as wn
from tests import workloads as wl


def img_u8(h, w, c=3, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, c), dtype=np.uint8)


def img_f32(h, w, c=3, seed=0):
    return np.random.default_rng(seed).random((h, w, c), dtype=np.float32)


@pytest.mark.parametrize("interp", [co.NEAREST, co.LINEAR])
@pytest.mark.parametrize("mk", [img_u8, img_f32])
def test_identity_is_crop(interp, mk):
    src = mk(40, 70)
    out = co.warp_perspective(src, np.eye(3), (50, 30), interp)
    np.testing.assert_array_equal(out, src[:30, :50])
    big = co.warp_perspective(src, np.eye(3), (90, 60), interp)  # beyond the source: zero fill
    np.testing.assert_array_equal(big[:40, :70], src)
    assert not big[40:].any() and not big[:, 70:].any()


@pytest.mark.parametrize("interp", [co.NEAREST, co.LINEAR])
def test_integer_translation(interp):
    src = img_u8(33, 47)
    M = np.array([[1, 0, 5.0], [0, 1, -3.0], [0, 0, 1]])  # dst(x, y) = src(x - 5, y + 3)
    out = co.warp_perspective(src, M, (47, 33), interp)
    exp = np.zeros_like(src)
    exp[:30, 5:] = src[3:, :42]
    np.testing.assert_array_equal(out, exp)


def test_nearest_ties_round_half_even():
    src = np.arange(1, 9, dtype=np.uint8)[None, :, None].repeat(2, 0)  # values 1..8 along x
    # dst x -> src x = x/2 : 0, .5, 1, 1.5, 2, 2.5 ...  ties go to the even integer: 0,0,1,2,2,2,3,4
    M = np.diag([2.0, 1.0, 1.0])
    out = co.warp_perspective(src, M, (8, 1), co.NEAREST)[0, :, 0]
    np.testing.assert_array_equal(out, [1, 1, 2, 3, 3, 3, 4, 5])
    # x2 minification: dst x -> src 2x exactly
    out = co.warp_perspective(src, np.diag([0.5, 1.0, 1.0]), (4, 1), co.NEAREST)[0, :, 0]
    np.testing.assert_array_equal(out, [1, 3, 5, 7])


@pytest.mark.parametrize("interp", [co.NEAREST, co.LINEAR])
def test_rot90_and_flips(interp):
    src = img_u8(20, 31)
    h, w = src.shape[:2]
    # horizontal flip: dst x = (w-1) - src x
    M = np.array([[-1, 0, w - 1.0], [0, 1, 0], [0, 0, 1]])
    np.testing.assert_array_equal(co.warp_perspective(src, M, (w, h), interp), src[:, ::-1])
    M = np.array([[1, 0, 0], [0, -1, h - 1.0], [0, 0, 1]])
    np.testing.assert_array_equal(co.warp_perspective(src, M, (w, h), interp), src[::-1])
    # transpose: dst (x, y) = src (y, x)
    M = np.array([[0, 1, 0], [1, 0, 0], [0, 0, 1.0]])
    np.testing.assert_array_equal(co.warp_perspective(src, M, (h, w), interp), src.transpose(1, 0, 2))
    # rotate 90: dst(x, y) = src(y, h-1-x)  <=>  forward: x_d = h-1-y_s, y_d = x_s
    M = np.array([[0, -1, h - 1.0], [1, 0, 0], [0, 0, 1]])
    np.testing.assert_array_equal(co.warp_perspective(src, M, (h, w), interp), np.rot90(src, -1))


@pytest.mark.parametrize("k", [1, 7, 16, 31])
def test_subpixel_shift_f32_and_u8(k):
    f = k / 32.0
    src = img_f32(6, 40, 1)
    M = np.array([[1, 0, -f], [0, 1, 0], [0, 0, 1.0]])  # dst x samples src x + k/32
    out = co.warp_perspective(src, M, (39, 6), co.LINEAR)
    a, b = src[:, :39], src[:, 1:40]
    w0, w1 = np.float32(1) - np.float32(f), np.float32(f)
    np.testing.assert_array_equal(out, a * w0 + b * w1)
    u = img_u8(6, 40, 3)
    outu = co.warp_perspective(u, M, (39, 6), co.LINEAR)
    a, b = u[:, :39].astype(np.int64), u[:, 1:40].astype(np.int64)
    exp = (a * (32 - k) * 1024 + b * k * 1024 + (1 << 14)) >> 15
    np.testing.assert_array_equal(outu, exp.astype(np.uint8))
    # vertical
    Mv = np.array([[1, 0, 0], [0, 1, -f], [0, 0, 1.0]])
    outv = co.warp_perspective(u, Mv, (40, 5), co.LINEAR)
    a, b = u[:5].astype(np.int64), u[1:6].astype(np.int64)
    np.testing.assert_array_equal(outv, ((a * (32 - k) * 1024 + b * k * 1024 + (1 << 14)) >> 15).astype(np.uint8))


def test_coordinates_quantised_to_1_32():
    """A shift of 1/64 px is a tie at the 1/32 grid: round-half-even picks the even 1/32 step."""
    src = img_f32(4, 20, 1)
    for shift, k in ((1 / 64, 0), (3 / 64, 2), (1 / 128, 0), (5 / 128, 1)):
        M = np.array([[1, 0, -shift], [0, 1, 0], [0, 0, 1.0]])
        sxy, alpha = co.warp_maps((8, 1), co.invert3x3(M), co.LINEAR)
        assert (alpha[0] & 31 == k).all(), (shift, alpha)


def test_out_of_bounds_taps_blend_with_zero_and_border_value():
    src = np.full((4, 4, 1), 200, np.uint8)
    M = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1.0]])  # dst x samples src x - 0.5
    out = co.warp_perspective(src, M, (6, 4), co.LINEAR)[0, :, 0]
    np.testing.assert_array_equal(out, [100, 200, 200, 200, 100, 0])
    out = co.warp_perspective(src, M, (6, 4), co.LINEAR, border_value=50)[0, :, 0]
    np.testing.assert_array_equal(out, [125, 200, 200, 200, 125, 50])
    out = co.warp_perspective(src, M, (6, 4), co.NEAREST, border_value=[7])[0, :, 0]
    np.testing.assert_array_equal(out, [200, 200, 200, 200, 7, 7])  # x-.5: -.5->-0  .5->0  1.5->2  2.5->2  3.5->4 (out)  4.5->4 (out)
    f = np.ones((3, 3, 2), np.float32)
    M = np.array([[1, 0, 0.25], [0, 1, 0.25], [0, 0, 1.0]])
    out = co.warp_perspective(f, M, (3, 3), co.LINEAR)
    assert out[0, 0, 0] == np.float32(0.75 * 0.75) and out[1, 1, 1] == 1.0


def test_float_pixels_fully_outside_are_the_border_value_itself():
    """remapBilinear's "fully outside" path (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) stores cval[k] directly: a
    float32 pixel whose four taps are all outside is EXACTLY the border value, not cval*w0 + cval*w1 + cval*w2 + cval*w3
    (which is an ulp off for most values and sub-pixel positions).  Partial overlap keeps per-tap substitution."""
    rng = np.random.default_rng(3)
    src = rng.random((9, 11, 3), dtype=np.float32)
    bv = [0.3, 0.7, 1.0 / 3.0]
    # a sub-pixel shift far outside the frame: every destination pixel samples beyond the right / bottom edge
    M = np.array([[1, 0, -40.0 - 13 / 32], [0, 1, -25.0 - 7 / 32], [0, 0, 1.0]])
    for fn in (co.warp_perspective, wn.warp_perspective):
        out = fn(src, M, (16, 12), co.LINEAR, border_value=bv)
        for k in range(3):
            assert (out[..., k] == np.float32(bv[k])).all(), (fn.__module__, k)
    # the 4-term sum the old restatement computed is NOT the border value for this weight set (what the rule is about)
    w = [np.float32((1 - 7 / 32) * (1 - 13 / 32)), np.float32((1 - 7 / 32) * (13 / 32)), np.float32((7 / 32) * (1 - 13 / 32)), np.float32((7 / 32) * (13 / 32))]
    c = np.float32(0.3)
    assert ((c * w[0] + c * w[1]) + c * w[2]) + c * w[3] != c
    # partial overlap: taps substituted one by one (left tap column outside, right inside)
    M = np.array([[1, 0, 1.0 - 8 / 32], [0, 1, 0], [0, 0, 1.0]])  # dst x samples src x - 0.75
    one = np.ones((3, 4, 1), np.float32)
    out = co.warp_perspective(one, M, (2, 1), co.LINEAR, border_value=0.3)
    exp0 = (np.float32(0.3) * np.float32(0.75) + np.float32(1.0) * np.float32(0.25)) + np.float32(0) + np.float32(0)
    assert out[0, 0, 0] == np.float32(exp0)
    np.testing.assert_array_equal(out, wn.warp_perspective(one, M, (2, 1), co.LINEAR, border_value=0.3))


def test_bilinear_table_matches_closed_form_for_all_bytes():
    """BilinearTab_i as OpenCV builds it: (32-fy)(32-fx)*32 ... except {32767,0,0,1} at (0,0), which
    produces the same byte as {32768,0,0,0} for every input."""
    tab = co.bilinear_tab_i().astype(np.int64)
    fy, fx = np.divmod(np.arange(1024), 32)
    closed = np.stack([(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx], 1) * 32
    np.testing.assert_array_equal(tab[1:], closed[1:])
    np.testing.assert_array_equal(tab[0], [32767, 0, 0, 1])
    assert (tab.sum(1) == 32768).all()
    p = np.arange(256)
    for p11 in (0, 255):
        np.testing.assert_array_equal((p * 32767 + p11 + 16384) >> 15, p)


def test_invert_matches_numpy_and_singular():
    rng = np.random.default_rng(1)
    for _ in range(20):
        M = rng.normal(size=(3, 3))
        np.testing.assert_allclose(co.invert3x3(M), np.linalg.inv(M), rtol=1e-9, atol=1e-12)
        np.testing.assert_array_equal(co.invert3x3(M), wn.invert3x3(M))
    assert not co.invert3x3(np.ones((3, 3))).any()
    out = co.warp_perspective(img_u8(8, 8), np.ones((3, 3)), (8, 8), co.LINEAR)  # singular -> M^-1 = 0 -> W = 0
    np.testing.assert_array_equal(out, np.broadcast_to(img_u8(8, 8)[0, 0], (8, 8, 3)))


def test_block_origin_evaluation_order():
    """X0/Y0/W0 are taken at the left edge of each 64-wide block and x1 added afterwards: the maps are
    periodic in that structure, and block widths follow min(1024 // min(16, h), w)."""
    assert co.block_width(1024, 1024) == 64 and co.block_width(40, 100) == 40
    assert co.block_width(500, 8) == 128 and co.block_width(500, 15) == 68 and co.block_width(2000, 1) == 1024
    M = wl.synth_brno_H(1920, 1080, 320, 200)
    Mi = co.invert3x3(M)
    sxy, alpha = co.warp_maps((320, 200), Mi, co.LINEAR)
    # recompute pixel (x=130, y=77) by hand with block origin 128
    x, y, bx = 130, 77, 128
    X0 = Mi[0, 0] * bx + Mi[0, 1] * y + Mi[0, 2]
    Y0 = Mi[1, 0] * bx + Mi[1, 1] * y + Mi[1, 2]
    W0 = Mi[2, 0] * bx + Mi[2, 1] * y + Mi[2, 2]
    W = 32 / (W0 + Mi[2, 0] * (x - bx))
    X = int(np.rint((X0 + Mi[0, 0] * (x - bx)) * W))
    Y = int(np.rint((Y0 + Mi[1, 0] * (x - bx)) * W))
    assert tuple(sxy[y, x]) == (X >> 5, Y >> 5) and alpha[y, x] == (Y & 31) * 32 + (X & 31)


CASES = [
    ("brno", 1280, 720, 512, 512), ("brno", 1920, 1080, 320, 640), ("keystone", 640, 360, 256, 192),
    ("brno", 192, 108, 37, 53), ("keystone", 100, 60, 300, 9), ("brno", 64, 36, 70, 1),
]


@pytest.mark.parametrize("kind,sw,sh,dw,dh", CASES)
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("interp", [co.NEAREST, co.LINEAR])
def test_c_oracle_equals_numpy_twin(kind, sw, sh, dw, dh, dtype, interp):
    M = (wl.synth_brno_H if kind == "brno" else wl.keystone_H)(sw, sh, dw, dh)
    src = wl.frame(0, sh, sw, dtype)
    a = co.warp_perspective(src, M, (dw, dh), interp)
    b = wn.warp_perspective(src, M, (dw, dh), interp)
    np.testing.assert_array_equal(a, b)
    if kind == "brno" and dw > 100:
        assert 0.02 < (a.reshape(-1, 3).any(1)).mean() <= 1.0  # a real, partially in-bounds warp


def test_channels_and_threads_and_strides():
    M = wl.synth_brno_H(320, 180, 96, 80)
    for c in (1, 2, 3, 4):
        src = np.random.default_rng(c).integers(0, 256, (180, 320, c), dtype=np.uint8)
        a = co.warp_perspective(src, M, (96, 80), co.LINEAR)
        np.testing.assert_array_equal(a, wn.warp_perspective(src, M, (96, 80), co.LINEAR))
        np.testing.assert_array_equal(a, co.warp_perspective(src, M, (96, 80), co.LINEAR, nthreads=4))
    g = np.random.default_rng(9).integers(0, 256, (180, 320), dtype=np.uint8)
    assert co.warp_perspective(g, M, (96, 80)).shape == (80, 96)
    # inverse flag
    src = wl.frame(1, 180, 320, np.uint8)
    np.testing.assert_array_equal(co.warp_perspective(src, M, (96, 80)),
                                  co.warp_perspective(src, co.invert3x3(M), (96, 80), m_is_inverse=True))


def test_footprint_counts_distinct_source_pixels():
    n, touched = co.footprint((10, 10), np.eye(3), (4, 3), co.NEAREST)
    assert n == 12 and touched[:3, :4].all()
    n, _ = co.footprint((10, 10), np.eye(3), (4, 3), co.LINEAR)
    assert n == 20  # taps reach one pixel further right and down (weight 0)
    M = wl.keystone_H(1920, 1080, 1024, 1024)
    n, touched = co.footprint((1080, 1920), M, (1024, 1024), co.LINEAR)
    assert 0.75 < n / (1080 * 1920) < 0.85  # ~80 % of the frame
    # every sample point of the keystone lies inside the frame
    sxy, _ = co.warp_maps((1024, 1024), co.invert3x3(M), co.LINEAR)
    assert sxy[..., 0].min() >= 0 and sxy[..., 0].max() <= 1919 and sxy[..., 1].min() >= 0 and sxy[..., 1].max() <= 1079


def test_point_projection_against_reference_vectors(golden):
    g = golden["rbox"]
    Hm = np.array(g["H_world_img"])
    for key, out in (("pts2", "pts_world_bev_2"), ("pts3", "pts_world_bev_3")):
        got = co.project_points(np.array(g[key]), Hm)
        np.testing.assert_allclose(got, np.array(g[out]), rtol=1e-12, atol=0)
    got32 = co.project_points(np.array(g["pts2"], dtype=np.float32), Hm)
    np.testing.assert_allclose(got32, np.array(g["pts_world_bev_2"]), rtol=1e-4, atol=1e-4)  # inputs rounded to f32


def test_rbox_iou_known_answers():
    sq = np.array([[0, 0, 1, 1, 0.0]])
    assert co.rbox_iou(sq, sq)[0, 0] == pytest.approx(1.0, abs=1e-12)
    assert co.rbox_iou(sq, np.array([[5, 5, 1, 1, 0.3]]))[0, 0] == 0.0
    # axis-aligned offset: w=2 (along y), h=4 (along x) shifted by 1 along x: inter 3*2, union 16-6
    a = np.array([[0, 0, 2, 4, 0.0]])
    b = np.array([[1, 0, 2, 4, 0.0]])
    assert co.rbox_iou(a, b)[0, 0] == pytest.approx(6 / 10, abs=1e-12)
    # 90-degree rotated square is the same square; rotated rectangle swaps extents
    assert co.rbox_iou(sq, np.array([[0, 0, 1, 1, np.pi / 2]]))[0, 0] == pytest.approx(1.0, abs=1e-12)
    assert co.rbox_iou(a, np.array([[0, 0, 4, 2, np.pi / 2]]))[0, 0] == pytest.approx(1.0, abs=1e-12)
    # unit square vs itself rotated 45 degrees: intersection is a regular octagon of area 2(sqrt2 - 1)
    inter = 2 * (np.sqrt(2) - 1)
    assert co.rbox_iou(sq, np.array([[0, 0, 1, 1, np.pi / 4]]))[0, 0] == pytest.approx(inter / (2 - inter), abs=1e-12)
    # common rotation (the tracker's +pi/2 on both yaws) does not change the IoU
    rng = np.random.default_rng(11)
    A = np.stack([rng.uniform(0, 20, 16), rng.uniform(0, 20, 16), rng.uniform(1.6, 2.2, 16), rng.uniform(3.5, 6, 16), rng.uniform(-np.pi, np.pi, 16)], 1)
    B = A + rng.normal(0, 0.5, A.shape)
    i0 = co.rbox_iou(A, B)
    A2, B2 = A.copy(), B.copy()
    A2[:, 4] += np.pi / 2
    B2[:, 4] += np.pi / 2
    R = np.array([[0, -1], [1, 0.0]])
    A2[:, :2] = A[:, :2] @ R.T
    B2[:, :2] = B[:, :2] @ R.T
    np.testing.assert_allclose(co.rbox_iou(A2, B2), i0, atol=1e-12)
    np.testing.assert_allclose(co.rbox_iou(A, B), co.rbox_iou(B, A).T, atol=1e-12)
    assert (np.diag(i0) > 0.05).all() and i0.min() >= 0 and i0.max() <= 1
