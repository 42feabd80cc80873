"""The drop-in boundary: the cv2-shaped module, the tracker module, and the reference's own call sequence
(/root/reference/vis_homo.py:57-91) walked through the names it uses -- fixtures only, the reference's files never travel."""
import json

import numpy as np
import pytest

from bev_amd import cv2_compat, rbox
from bev_amd.homo import homo_from_pts


def test_cv2_compat_surface_without_a_device():
    cv2 = cv2_compat
    assert (cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.WARP_INVERSE_MAP, cv2.BORDER_CONSTANT) == (0, 1, 16, 0)
    # findHomography: (H, mask) like OpenCV, exact for four points, h33 = 1, (N, 1, 2) accepted (bev/homo.py:36)
    src = np.array([[0, 0], [639, 0], [639, 359], [0, 359]], dtype=np.float64)
    dst = np.array([[100, 50], [500, 80], [620, 300], [20, 340]], dtype=np.float64)
    H, mask = cv2.findHomography(src, dst)
    assert H.shape == (3, 3) and H[2, 2] == 1.0 and mask.shape == (4, 1) and mask.dtype == np.uint8 and mask.all()
    np.testing.assert_allclose(H, homo_from_pts(src, dst), rtol=0, atol=0)
    H2, _ = cv2.findHomography(src[:, None, :].astype(np.float32), dst[:, None, :].astype(np.float32))
    np.testing.assert_allclose(H2, H, rtol=1e-6)
    with pytest.raises(NotImplementedError):
        cv2.findHomography(src, dst, cv2.RANSAC)
    # perspectiveTransform: OpenCV's (N, 1, 2) shape and dtype preserved; == pts_world_bev (bev/rbox.py:136-151)
    pts = np.random.default_rng(0).uniform(0, 600, (9, 1, 2)).astype(np.float32)
    out = cv2.perspectiveTransform(pts, H)
    assert out.shape == pts.shape and out.dtype == np.float32
    np.testing.assert_allclose(out[:, 0], rbox.pts_world_bev(pts[:, 0].astype(np.float64), H), rtol=1e-6)
    np.testing.assert_allclose(cv2.perspectiveTransform(src, H), dst, atol=1e-9)
    with pytest.raises(TypeError):
        cv2.perspectiveTransform(pts.astype(np.int32), H)
    # invert: the closed form of the warp path
    ok, Hi = cv2.invert(H)
    assert ok == 1.0
    np.testing.assert_allclose(Hi @ H, np.eye(3), atol=1e-9)
    assert cv2.invert(np.zeros((3, 3)))[0] == 0.0
    assert not hasattr(cv2, "imread")  # what the path does not need is absent, not faked


def test_tracker_module_exports_the_reference_names():
    from bev.tracker import rbox_tracker
    import inspect
    sig = inspect.signature(rbox_tracker.iou_batch_rbox)
    assert list(sig.parameters)[:2] == ["bb_test", "bb_gt"]  # /root/reference/bev/tracker/rbox_tracker.py:87
    assert callable(rbox_tracker.tracker_geometry_step)


@pytest.mark.gpu
def test_vis_homo_sequence_through_the_reference_names(golden, tmp_path):
    """vis_homo.py:57-91 -- load_calib(json) -> load_bspec(yaml) -> compose -> cv2.warpPerspective, and the "small" branch
    (a resized frame + Calib.scale + warpPerspective) -- with `cv2` bound to bev_amd.cv2_compat and `bev` to the drop-in package."""
    import torch

    import bev_amd.cv2_compat as cv2
    from bev.constructor.homo_constr import load_bspec, load_calib
    from bev_amd import warp
    from oracle import cpu_oracle as co
    from tests import workloads as wl
    calib_path = tmp_path / "system_dubska_optimal_calib.json"
    calib_path.write_text(json.dumps(golden["brno_file"]["json"]))
    calib = load_calib("BrnoCompSpeed", str(calib_path))
    bspec = load_bspec("BrnoCompSpeed", sub_id=6.1, calib=calib)
    H_world_img = calib.gen_H_world_img()
    H_world_bev = bspec.gen_H_world_bev()
    H_bev_img = np.linalg.inv(H_world_bev).dot(H_world_img)
    img = wl.frame(0, 1080, 1920, np.uint8)
    bev_img = cv2.warpPerspective(img, H_bev_img, (bspec.u_size, bspec.v_size))
    assert bev_img.shape == (bspec.v_size, bspec.u_size, 3) and bev_img.dtype == np.uint8
    np.testing.assert_array_equal(bev_img, co.warp_perspective(img, H_bev_img, (bspec.u_size, bspec.v_size), 1))
    # the small branch, as written in the reference ...
    new_u, new_v = 852, 480
    calib_small = calib.scale(align_corners=False, new_u=new_u, new_v=new_v)
    H_bev_img_small = np.linalg.inv(H_world_bev).dot(calib_small.gen_H_world_img())
    img_small = cv2.resize(img, (new_u, new_v))  # vis_homo.py:90, as written: OpenCV's own bilinear algorithm as a device kernel
    assert img_small.shape == (new_v, new_u, 3) and img_small.dtype == np.uint8
    np.testing.assert_array_equal(img_small, co.resize_linear_u8(img, (new_u, new_v)))
    bev_small = cv2.warpPerspective(img_small, H_bev_img_small, (bspec.u_size, bspec.v_size))
    np.testing.assert_array_equal(bev_small, co.warp_perspective(img_small, H_bev_img_small, (bspec.u_size, bspec.v_size), 1))
    # ... and fused: the full-resolution frame sampled once through H_small @ S (no intermediate image)
    fused = warp.warp_perspective_resized(torch.from_numpy(img).cuda(), H_bev_img_small, (bspec.u_size, bspec.v_size), (new_u, new_v)).cpu().numpy()
    S = warp.resize_matrix((1920, 1080), (new_u, new_v))
    np.testing.assert_array_equal(fused, co.warp_perspective(img, H_bev_img_small @ S, (bspec.u_size, bspec.v_size), 1))
    assert fused.shape == bev_small.shape and fused.dtype == np.uint8
    # the two forms of the small branch see the same scene: the two-step one low-passes through the resize, so they differ by grey levels,
    # not by geometry (mean |difference| on a noise image stays far below the noise's own contrast)
    both = (bev_small > 0).all(axis=2) & (fused > 0).all(axis=2)
    assert both.mean() > 0.3 and np.abs(bev_small[both].astype(np.int32) - fused[both].astype(np.int32)).mean() < 64


def _real_cv2():
    cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed on this box (never a requirement, never installed by us)")
    if not hasattr(cv2, "warpPerspective") or getattr(cv2, "__version__", "").startswith("bev_amd"):
        pytest.skip("`cv2` resolves to a stand-in, not to OpenCV")
    return cv2


def test_opencv_cross_check_of_the_resize_oracle_when_present():
    """The only thing that can pin oracle/resize_oracle.c is a real cv2: when one is importable, cv2.resize (INTER_LINEAR, uint8) must
    agree within 1 LSB (exactly on the classic fixed-point path the oracle restates)."""
    cv2 = _real_cv2()
    from oracle import cpu_oracle as co
    from tests import workloads as wl
    for shape, dsize in (((1080, 1920, 3), (852, 480)), ((64, 48, 1), (31, 17)), ((40, 64, 3), (32, 20)), ((33, 65, 3), (130, 99))):
        img = wl.frame(4, shape[0], shape[1], np.uint8, shape[2])
        got = cv2.resize(img, dsize).reshape(dsize[1], dsize[0], shape[2])
        assert np.abs(got.astype(np.int32) - co.resize_linear_u8(img, dsize).astype(np.int32)).max() <= 1


def test_opencv_cross_check_of_the_oracle_when_present():
    """the only thing that can pin the warp oracle is a real cv2.  When one is importable: nearest must agree
    exactly, 8-bit bilinear within 1 LSB (exactly on the classic fixed-point path the oracle restates), float bilinear within
    the north star's 1e-5 on that path; findHomography within solver tolerance."""
    cv2 = _real_cv2()
    from oracle import cpu_oracle as co
    from tests import workloads as wl
    M = wl.synth_brno_H(640, 360, 256, 192)
    u8, f32 = wl.frame(1, 360, 640, np.uint8), wl.frame(2, 360, 640, np.float32)
    np.testing.assert_array_equal(cv2.warpPerspective(u8, M, (256, 192), flags=cv2.INTER_NEAREST), co.warp_perspective(u8, M, (256, 192), 0))
    d8 = np.abs(cv2.warpPerspective(u8, M, (256, 192)).astype(int) - co.warp_perspective(u8, M, (256, 192), 1).astype(int)).max()
    assert d8 <= 1, "8-bit bilinear differs from OpenCV %s by %d LSB" % (cv2.__version__, d8)
    if d8 == 0:  # the classic 1/32-px path: float must then meet the north-star tolerance
        df = np.abs(cv2.warpPerspective(f32, M, (256, 192)) - co.warp_perspective(f32, M, (256, 192), 1)).max()
        assert df <= 1e-5, df
    src = np.array([[0, 0], [639, 0], [639, 359], [0, 359], [320, 180], [100, 300]], dtype=np.float64)
    dst = rbox.pts_world_bev(src, M) + np.random.default_rng(1).normal(0, 0.05, (6, 2))
    np.testing.assert_allclose(homo_from_pts(src, dst), cv2.findHomography(src, dst)[0], rtol=1e-5, atol=1e-7)
