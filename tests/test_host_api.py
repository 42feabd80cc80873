"""Host algebra (bev.homo / bev.calib / bev.bev / bev.constructor / bev.rbox*) against vectors the
upstream reference itself produced (tests/golden/reference_vectors.json, made by make_golden.py).
Tolerances: these are float64 chains of a few dozen flops -> rtol 1e-12 unless stated."""
import json
import os

import numpy as np
import pytest
import torch

import bev
from bev import homo as H
from bev import rbox, rbox_torch
from bev.constructor import homo_constr as hc
from bev.constructor import homo_constr_utils as hcu

RT = dict(rtol=1e-12, atol=1e-12)

CALIB_FIELDS = ["K", "u_size", "v_size", "T", "vp1", "vp2", "pp", "height", "mode"]
BSPEC_FIELDS = ["u_size", "v_size", "u_axis", "v_axis", "x_size", "y_size", "x_min", "x_max",
                "y_min", "y_max", "u_min", "u_max", "v_min", "v_max"]


def close(a, b, **kw):
    kw = kw or RT
    if b is None or isinstance(b, str):
        assert a == b
    else:
        np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), **kw)


def check_calib(calib, rec):
    for k in CALIB_FIELDS:
        close(getattr(calib, k), rec[k])
    # H_world_img is an inverse of a ~1e3-conditioned matrix: allow 1e-10
    close(calib.gen_H_world_img(), rec["H_world_img"], rtol=1e-10, atol=1e-12)
    close(calib.gen_center_in_world(), rec["center_world"], rtol=1e-10, atol=1e-12)


def check_bspec(b, rec):
    for k in BSPEC_FIELDS:
        close(getattr(b, k), rec[k])
    close(b.gen_bev_corners_in_world(), rec["corners_world"])


def variants(calib):
    return {
        "base": calib,
        "scale_nc_852x480": calib.scale(False, 852, 480),
        "scale_ac_852x480": calib.scale(True, 852, 480),
        "scale_ratio_nc_0.5_0.25": calib.scale(False, scale_ratio_u=0.5, scale_ratio_v=0.25),
        "scale_ratio_ac_0.5_0.25": calib.scale(True, scale_ratio_u=0.5, scale_ratio_v=0.25),
        "pad_8_4_8_4": calib.pad(8, 4, 8, 4),
        "pad_-3_5_0_-2": calib.pad(-3, 5, 0, -2),
        "flip_lr": calib.flip(lr=True),
        "flip_tb": calib.flip(tb=True),
        "flip_lr_tb": calib.flip(lr=True, tb=True),
    }


@pytest.fixture(scope="module")
def calib_vps(golden):
    v = golden["homo"]["vps_in"]
    return bev.Calib(vp1=np.array(v["vp1"]), vp2=np.array(v["vp2"]), pp=np.array(v["pp"]), height=v["height"],
                     u_size=v["u_size"], v_size=v["v_size"])


def test_homo_algebra(golden):
    g = golden["homo"]
    v = g["vps_in"]
    vp1, vp2, pp = np.array(v["vp1"]), np.array(v["vp2"]), np.array(v["pp"])
    close(H.get_focal(vp1, vp2, pp), g["get_focal"])
    K, f = H.get_K_from_vps(vp1, vp2, pp)
    close(K, g["get_K_from_vps"]["K"])
    close(f, g["get_K_from_vps"]["focal"])
    Hiw = H.homo_from_vps(vp1, vp2, v["height"], v["u_size"], v["v_size"], pp)
    close(Hiw, g["homo_from_vps"])
    close(H.homo_from_vps(vp1, vp2, v["height"], v["u_size"], v["v_size"]), g["homo_from_vps_pp_none"])
    a, b = H.get_vps_from_homo(Hiw)
    close(a, g["get_vps_from_homo"]["vp1"])
    close(b, g["get_vps_from_homo"]["vp2"])
    k = g["KRt_in"]
    close(H.homo_from_KRt(np.array(k["K34"]), R=np.array(k["R"]), t=np.array(k["t"])), g["homo_from_KRt_Rt"])
    close(H.homo_from_KRt(np.array(k["K34"])[:, :3], Rt_homo=np.array(k["T"])), g["homo_from_KRt_T"])
    with pytest.raises(AssertionError):
        H.homo_from_KRt(np.eye(3), R=np.eye(3), t=np.zeros(3), Rt_homo=np.eye(4))
    with pytest.raises(AssertionError):
        H.homo_from_KRt(np.eye(3))


def test_KRt_roundtrip_from_homography(golden):
    """get_KRt_from_homo / Rt_from_homo_K invert homo_from_KRt for a camera with square pixels."""
    k = golden["homo"]["KRt_in"]
    K = np.array([[800.0, 0, 640.0], [0, 800.0, 360.0], [0, 0, 1.0]])
    ca, sa, cb, sb = np.cos(0.3), np.sin(0.3), np.cos(1.1), np.sin(1.1)
    R = np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]]) @ np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1.0]])
    t = np.array(k["t"])
    Hiw = H.homo_from_KRt(K, R=R, t=t)
    K2, focal, R2, t2 = H.get_KRt_from_homo(Hiw, np.array([640.0, 360.0]))
    close(K2, K, rtol=1e-9, atol=1e-9)
    close(R2, R, rtol=1e-9, atol=1e-9)
    close(t2, t, rtol=1e-9, atol=1e-9)
    R3, t3 = H.Rt_from_pts_K_dist(np.array([[0, 0, 0], [4, 0, 0], [4, 3, 0], [0, 3, 0], [2, 1, 0.0]]),
                                  rbox.pts_world_bev(np.array([[0, 0], [4, 0], [4, 3], [0, 3], [2, 1.0]]), Hiw), K, None)
    close(R3, R, rtol=1e-8, atol=1e-8)
    close(t3.ravel(), t, rtol=1e-8, atol=1e-8)
    with pytest.raises(NotImplementedError):
        H.Rt_from_pts_K_dist(np.zeros((4, 3)), np.zeros((4, 2)), K, np.array([0.1, 0, 0, 0, 0]))


def test_homo_from_pts():
    rng = np.random.default_rng(0)
    Ht = np.array([[1.2, -0.3, 40.0], [0.2, 0.9, -12.0], [1e-4, -2e-4, 1.0]])
    for n in (4, 6, 9, 16):
        src = rng.uniform(0, 1000, (n, 2))
        tgt = rbox.pts_world_bev(src, Ht)
        close(H.homo_from_pts(src, tgt), Ht, rtol=1e-8, atol=1e-9)
    # noisy over-determined: beats (or ties) the unrefined DLT on reprojection error
    src = rng.uniform(0, 1000, (12, 2))
    tgt = rbox.pts_world_bev(src, Ht) + rng.normal(0, 0.5, (12, 2))
    Hn = H.homo_from_pts(src, tgt)
    assert abs(Hn[2, 2] - 1) < 1e-15
    err = np.linalg.norm(rbox.pts_world_bev(src, Hn) - tgt)
    assert err < np.linalg.norm(rbox.pts_world_bev(src, Ht) - tgt) + 1e-9
    with pytest.raises(AssertionError):
        H.homo_from_pts(np.zeros((4, 3)), np.zeros((4, 2)))
    with pytest.raises(AssertionError):
        H.homo_from_pts(np.zeros((4, 2)), np.zeros((4,)))


def test_calib_from_vps(golden, calib_vps):
    for name, c in variants(calib_vps).items():
        check_calib(c, golden["calib_from_vps"][name])
    v = golden["homo"]["vps_in"]
    c = bev.Calib(vp1=np.array(v["vp1"]), vp2=np.array(v["vp2"]), height=8, u_size=1920, v_size=1080)
    check_calib(c, golden["calib_from_vps_pp_default"])
    assert c.mode == "from_vps"


@pytest.mark.parametrize("sub_id", [1, 4])
def test_calib_KoPER(golden, sub_id):
    c = hc.preset_calib("KoPER", sub_id)
    assert c.mode == "from_KRt" and c.K.dtype == np.float32
    for name, cv in variants(c).items():
        check_calib(cv, golden["calib_KoPER_%d" % sub_id][name])
    g = golden["load_T_KoPER_%d" % sub_id]
    fx, fy, cx, cy, T = hcu.load_T("KoPER", sub_id)
    assert (fx, fy, cx, cy) == (g["fx"], g["fy"], g["cx"], g["cy"])
    close(T, g["T"], rtol=0, atol=0)


def test_calib_modes_and_errors():
    K = np.array([[800.0, 0, 640.0], [0, 800.0, 360.0], [0, 0, 1.0]])
    R = np.eye(3)
    t = np.array([0.0, 0.0, 10.0])
    c = bev.Calib(K=K, R=R, t=t, u_size=1280, v_size=720)  # TypeError in the reference (calib.py:76)
    assert c.T.shape == (4, 4) and c.mode == "from_KRt"
    with pytest.raises(ValueError):
        bev.Calib(K=K, R=R, u_size=1280, v_size=720)
    with pytest.raises(AssertionError):
        bev.Calib(K=K, R=R, t=t, T=np.eye(4), u_size=1280, v_size=720)
    with pytest.raises(TypeError):
        c.not_a_field = 1
    c.height = 3.0  # existing attribute stays writable
    pts_w = np.array([[0, 0, 0], [10, 0, 0], [10, 5, 0], [0, 5, 0.0]])
    pts_i = np.array([[100, 600], [1200, 620], [900, 300], [300, 290.0]])
    cp = bev.Calib(pts_world=pts_w, pts_image=pts_i, u_size=1280, v_size=720)
    assert cp.mode == "from_pts"
    Hwi = cp.gen_H_world_img()
    close(rbox.pts_world_bev(pts_i, Hwi), pts_w[:, :2], rtol=1e-9, atol=1e-9)
    cs = cp.scale(False, 640, 360)
    close(rbox.pts_world_bev((pts_i + 0.5) * 0.5 - 0.5, cs.gen_H_world_img()), pts_w[:, :2], rtol=1e-9, atol=1e-9)
    cf = cp.flip(lr=True).pad(3, 4, 5, 6)
    assert (cf.u_size, cf.v_size) == (1288, 730)
    with pytest.raises(AssertionError):
        cp.gen_H_world_img(mode="nope")


def test_load_pts(golden):
    for key, g in golden["load_pts_852x480"].items():
        name, sub = key.rsplit("_", 1)
        p3, p2 = hcu.load_pts(name, 852, 480, None if sub == "None" else int(sub))
        assert p3.dtype == np.float32 and p2.dtype == np.float32
        close(p3, g["pts_3d"], rtol=0, atol=0)
        close(p2, g["pts_2d"], rtol=0, atol=0)
    c = hc.preset_calib("roundabout")
    assert c.mode == "from_pts" and (c.u_size, c.v_size) == (852, 480)
    Hwi = c.gen_H_world_img()  # 9 surveyed points: least squares, reprojects within metres
    err = np.abs(rbox.pts_world_bev(c.pts_image.astype(np.float64), Hwi) - c.pts_world[:, :2]).max()
    assert err < 3.0


def test_bspec_yaml_and_cfg_modes(golden, calib_vps):
    assert sorted(golden["bspec_yaml"]) == sorted(os.listdir(os.path.dirname(hc.cfg_path_from_dataset_id("BrnoCompSpeed", 0))))
    for fn, g in golden["bspec_yaml"].items():
        sub = fn[len("BrnoCompSpeed_"):-len(".yaml")].replace("_", ".")
        sub = float(sub) if "." in sub else int(sub)
        path = hc.cfg_path_from_dataset_id("BrnoCompSpeed", sub)
        assert os.path.basename(path) == fn
        import yaml
        with open(path) as f:
            assert yaml.safe_load(f) == g["cfg"]  # same schema and numbers as the reference's config
        check_bspec(hc.load_bspec("BrnoCompSpeed", sub, calib_vps), g["bspec"])
    check_bspec(hc.load_bspec_from_cfg(golden["bspec_cfg_abs"]["cfg"]), golden["bspec_cfg_abs"]["bspec"])
    check_bspec(hc.load_bspec_from_cfg(golden["bspec_cfg_centered"]["cfg"], calib_vps), golden["bspec_cfg_centered"]["bspec"])
    for k, v in golden["cfg_path_from_dataset_id"].items():
        assert os.path.basename(hc.cfg_path_from_dataset_id("BrnoCompSpeed", float(k) if "." in k else int(k))) == v
    with pytest.raises(AssertionError):
        hc.load_bspec_from_cfg(golden["bspec_cfg_centered"]["cfg"])
    with pytest.raises(ValueError):
        hc.load_bspec_from_cfg({"mode": "bogus", "spec": {}})
    with pytest.raises(AssertionError):
        hc.cfg_path_from_dataset_id("NoSuchDataset", 1)


def test_preset_bspec(golden, calib_vps):
    for key, g in golden["preset_bspec"].items():
        parts = key.split("|")
        name = parts[0]
        sub = None if parts[1] == "None" else (float(parts[1]) if "." in parts[1] else int(parts[1]))
        calib = calib_vps if (len(parts) > 2 and parts[2] == "calib") else None
        check_bspec(hc.preset_bspec(name, sub, calib), g)


def test_bspec_axes_and_transforms(golden):
    for key, g in golden["bspec_axes"].items():
        ua, va = key.split("|")
        b = bev.BEVWorldSpec(u_size=320, v_size=640, u_axis=ua, v_axis=va, x_min=-3.5, x_size=80.0, y_max=11.25, y_size=40.0)
        check_bspec(b, g)
        # the 4-corner homography is exact and affine (what rbox.rbox_world_bev asserts)
        Hwb = b.gen_H_world_bev()
        corners_px = np.array([[0, 0], [0, 640], [320, 640], [320, 0.0]])
        close(rbox.pts_world_bev(corners_px, Hwb), g["corners_world"], rtol=1e-11, atol=1e-11)
        assert abs(Hwb[2, 0]) + abs(Hwb[2, 1]) < 1e-12
    b0 = bev.BEVWorldSpec(u_size=320, v_size=640, u_axis="y", v_axis="-x", x_min=-23.25, x_size=80.0, y_min=-16.5, y_size=40.0)
    tr = {
        "base": b0,
        "scale_nc_160x320": b0.scale(False, 160, 320),
        "scale_ac_161x321": b0.scale(True, 161, 321),
        "scale_ratio_nc_1.5_0.75": b0.scale(False, scale_ratio_u=1.5, scale_ratio_v=0.75),
        "pad_8_4_16_12": b0.pad(8, 4, 16, 12),
        "pad_then_scale": b0.pad(8, 4, 16, 12).scale(False, 172, 328),
        "scale_then_pad": b0.scale(False, 160, 320).pad(1, 2, 3, 4),
        "flip_lr": b0.flip(lr=True),
        "flip_tb": b0.flip(tb=True),
        "flip_lr_tb": b0.flip(lr=True, tb=True),
    }
    for name, b in tr.items():
        check_bspec(b, golden["bspec_transforms"][name])
    # scaled spec maps the same world rectangle: H_world_bev(scaled) . scale == H_world_bev
    Hs = tr["scale_nc_160x320"].gen_H_world_bev()
    p = np.array([[10.0, 20.0], [300.0, 600.0]])
    close(rbox.pts_world_bev((p + 0.5) * 0.5 - 0.5, Hs), rbox.pts_world_bev(p, b0.gen_H_world_bev()), rtol=1e-10, atol=1e-10)


def test_bspec_errors():
    with pytest.raises(AssertionError):
        bev.BEVWorldSpec(u_size=10, v_size=10, x_min=0, x_max=1, x_size=2, y_min=0, y_size=1)  # inconsistent
    with pytest.raises(AssertionError):
        bev.BEVWorldSpec(u_size=10, v_size=10, x_min=0, y_min=0, y_size=1)  # two of three missing
    with pytest.raises(AssertionError):
        bev.BEVWorldSpec(u_size=10, v_size=10, u_axis="x", v_axis="-x", x_min=0, x_size=1, y_min=0, y_size=1)
    b = bev.BEVWorldSpec(u_size=10, v_size=10, x_min=0, x_size=1, y_min=0, y_size=1)
    with pytest.raises(TypeError):
        b.z_min = 0
    b.set_keep(x_size=4, x_max=None)
    assert b.x_max == 4
    b.u_axis, b.v_axis = "x", "x"
    with pytest.raises(AssertionError):
        b.check_validity()


def test_file_loaders(golden, tmp_path):
    g = golden["carla_file"]
    p = tmp_path / "carla.txt"
    p.write_text(g["text"])
    K, T, us, vs = hcu.load_calib_from_file_carla(str(p))
    close(K, g["K"])
    close(T, g["T_cam_world"])
    assert (us, vs) == (g["u_size"], g["v_size"])
    check_calib(hc.load_calib("CARLA", str(p)), g["calib"])
    g = golden["blender_file"]
    p = tmp_path / "blender.txt"
    p.write_text(g["text"])
    K, Rt, us, vs = hcu.load_calib_from_file_blender(str(p))
    close(K, g["K"])
    close(Rt, g["Rt"])
    assert (int(us), int(vs)) == (g["u_size"], g["v_size"])
    check_calib(hc.load_calib("blender", str(p)), g["calib"])
    g = golden["brno_file"]
    p = tmp_path / "brno.json"
    p.write_text(json.dumps(g["json"]))
    check_calib(hc.load_calib("BrnoCompSpeed", str(p)), g["calib"])
    with pytest.raises(ValueError):
        hc.load_calib("nope", str(p))


def test_rbox_numpy(golden):
    g = golden["rbox"]
    Hwi, Hwb, Hrefl = np.array(g["H_world_img"]), np.array(g["H_world_bev"]), np.array(g["H_world_bev_refl"])
    pts2, pts3, boxes = np.array(g["pts2"]), np.array(g["pts3"]), np.array(g["boxes_bev"])
    close(rbox.pts_world_bev(pts2, Hwi), g["pts_world_bev_2"])
    close(rbox.pts_world_bev(pts3, Hwi), g["pts_world_bev_3"])
    close(rbox.pts_world_bev(pts2[0], Hwi), g["pts_world_bev_1d"])
    bw = rbox.rbox_world_bev(boxes, Hwb, "bev")
    close(bw, g["rbox_world_bev__bev2world"])
    close(rbox.rbox_world_bev(bw, np.linalg.inv(Hwb), "world"), g["rbox_world_bev__world2bev"])
    close(rbox.rbox_world_bev(boxes, Hrefl, "bev"), g["rbox_world_bev__bev2world_refl"])
    close(rbox.rbox_world_img(bw, np.linalg.inv(Hwi)), g["rbox_world_img"])
    for mode in ("bev", "world"):
        xy8 = rbox.xywhr2xyxy(boxes, mode)
        close(xy8, g["xywhr2xyxy_%s" % mode])
        close(rbox.xy82xywhr(xy8, mode), g["xy82xywhr_%s" % mode])
        close(rbox.xywhr2xyvec(boxes, mode), g["xywhr2xyvec_%s" % mode])
        close(rbox.yaw2v(boxes[:, 4], mode), g["yaw2v_%s" % mode])
        close(rbox.v2yaw(rbox.yaw2v(boxes[:, 4], mode), mode), g["v2yaw_%s" % mode])
        close(rbox.yaw2mat(boxes[:, 4], mode), g["yaw2mat_%s" % mode])
        close(rbox.angle_world_bev(boxes[:, 4], Hwb, mode), g["angle_world_bev_src_%s" % mode])
        aa = rbox.xywhr2xyxy(boxes, mode, external_aa=True)  # reference raises; intended hull checked here
        close(aa, np.stack([xy8[:, 0::2].min(1), xy8[:, 1::2].min(1), xy8[:, 0::2].max(1), xy8[:, 1::2].max(1)], 1))
    close(rbox.xy82xyvec(rbox.xywhr2xyxy(boxes, "bev")), g["xy82xyvec"])
    close(rbox.dist_world_bev(boxes[:, 2:4], Hwb), g["dist_world_bev"])
    close(rbox.rbox_zt2tt_world(np.array(g["rboxzt_in"]), np.array(g["K3"]), np.array(g["Rt"])), g["rbox_zt2tt_world"], rtol=1e-9, atol=1e-9)
    close(rbox.rboxtt_world_bev(np.array(g["rboxtt_in"]), np.array(g["H_bev_world_int"]), "world"), g["rboxtt_world_bev"])
    empty = np.zeros((0, 5))
    assert rbox.rbox_world_bev(empty, Hwb, "bev") is empty
    with pytest.raises(AssertionError):
        rbox.rbox_world_bev(boxes, Hwi, "bev")  # perspective H is not a similarity
    with pytest.raises(AssertionError):
        rbox.rbox_world_bev(boxes, Hwb, "image")
    with pytest.raises(NotImplementedError):
        rbox.rboxzt_world_bev(np.zeros((1, 7)), Hwb, np.eye(3), np.eye(4), "bev")


def test_rbox_torch(golden):
    g = golden["rbox"]
    Hwb, boxes = np.array(g["H_world_bev"]), np.array(g["boxes_bev"])
    tb, tH = torch.from_numpy(boxes), torch.from_numpy(Hwb)
    bw = rbox_torch.rbox_world_bev(tb, tH, "bev")
    close(bw.numpy(), g["rbox_world_bev_torch__bev2world"])
    close(rbox_torch.rbox_world_bev(bw, torch.from_numpy(np.linalg.inv(Hwb)), "world").numpy(), g["rbox_world_bev_torch__world2bev"])
    close(rbox_torch.rbox_world_bev(tb.float(), tH.float(), "bev").numpy(), g["rbox_world_bev_torch_f32__bev2world"], rtol=1e-6, atol=1e-5)
    for mode in ("bev", "world"):
        close(rbox_torch.xywhr2xyxy(tb, mode).numpy(), g["xywhr2xyxy_torch_%s" % mode])
        close(rbox_torch.xywhr2xyvec(tb, mode).numpy(), g["xywhr2xyvec_torch_%s" % mode])
        aa = rbox_torch.xywhr2xyxy(tb, mode, external_aa=True)
        assert aa.shape == (len(boxes), 4)
    close(rbox_torch.xy82xyvec(torch.from_numpy(rbox.xywhr2xyxy(boxes, "bev"))).numpy(), g["xy82xyvec_torch"])


def test_resize_matrix_matches_calib_scale():
    """Folding cv2.resize into the homography: resize_matrix follows the pixel conventions of the
    reference's Calib.scale (bev/calib.py:142-198; vis_homo.py:73-78).  For an aspect-preserving resize the scaled
    calibration is exactly the original one seen through S: H_world_img_small @ S == H_world_img up to scale (for
    852x480 the reference's vanishing-point model is only approximately consistent, by ~1e-3)."""
    import bev
    from bev.warp import resize_matrix
    S = resize_matrix((1920, 1080), (852, 480), False)
    np.testing.assert_allclose(S @ [-0.5, -0.5, 1], [-0.5, -0.5, 1], atol=1e-12)          # outer pixel edges map to
    np.testing.assert_allclose(S @ [1919.5, 1079.5, 1], [851.5, 479.5, 1], atol=1e-9)     # outer pixel edges
    S = resize_matrix((1920, 1080), (852, 480), True)
    np.testing.assert_allclose(S @ [1919, 1079, 1], [851, 479, 1], atol=1e-9)             # corner centres to corner centres
    calib = bev.Calib(vp1=np.array([1200.0, -300.0]), vp2=np.array([-2500.0, -150.0]), pp=np.array([959.5, 539.5]), height=8, u_size=1920, v_size=1080)
    H = calib.gen_H_world_img()
    for align in (False, True):
        small = calib.scale(align_corners=align, new_u=960, new_v=540)
        Hs = small.gen_H_world_img() @ resize_matrix((1920, 1080), (960, 540), align)
        if align:  # (1919/959 != 1079/539: not aspect-preserving in this convention)
            continue
        np.testing.assert_allclose(Hs / Hs[2, 2], H / H[2, 2], rtol=1e-9, atol=1e-9)


def test_composite_integer_form_equals_the_float64_expression():
    """composite_reg_img (/root/reference/bev/tool/compo.py:16-23): round(fg * (m / 255) + bg * (1 - m / 255)) in float64, clipped
    to 255.  The HIP kernels evaluate it as ((fg m + bg (255 - m) + 127) * 0x8081) >> 23 -- exact integer arithmetic; here for
    every one of the 2^24 (fg, bg, mask) byte triples against the reference's own numpy expression."""
    f = np.arange(256, dtype=np.float64)[:, None, None]
    b = np.arange(256, dtype=np.float64)[None, :, None]
    m = np.arange(256, dtype=np.float64)[None, None, :] / 255
    ref = (f * m + b * (1 - m)).round()
    ref[ref > 255] = 255
    fi, bi, mi = (np.arange(256, dtype=np.int64).reshape(s) for s in ((256, 1, 1), (1, 256, 1), (1, 1, 256)))
    got = ((fi * mi + bi * (255 - mi) + 127) * 0x8081) >> 23
    np.testing.assert_array_equal(got, ref.astype(np.int64))


def test_d3d_stand_in_is_consistent_with_the_rebound_tracker_iou(monkeypatch):
    """the reference calls d3d.box.box2d_iou(a + pi/2, b + pi/2) (rbox_tracker.py:88-92) and the rebound
    iou_batch_rbox calls the kernel with the yaws as given, so the d3d stand-in must satisfy
    stand_in(a + pi/2, b + pi/2) == iou_batch_rbox(a, b): it swaps w and h.  Checked here with the CPU oracle standing in for
    the kernel (the GPU twin is tests/test_gpu_geom.py::test_d3d_stand_in_identity_on_the_device), on non-square boxes --
    turning a box a quarter about its own centre changes the IoU: two 1 x 4 boxes at (0, 0) and (0, 1) give 0 one way, 0.6 the other."""
    import sys

    from oracle import cpu_oracle as co

    from bev_amd import iou as iou_mod
    from bev_amd import overlay
    monkeypatch.setattr(iou_mod, "iou_any", lambda a, b, device="cuda": co.rbox_iou(np.asarray(a, dtype=np.float64)[:, :5], np.asarray(b, dtype=np.float64)[:, :5]))
    saved = {k: sys.modules.pop(k) for k in ("d3d", "d3d.box") if k in sys.modules}
    try:
        if not overlay.ensure_d3d():
            pytest.skip("a real d3d is installed: the stand-in is not registered")
        import d3d
        rng = np.random.default_rng(4)
        a = np.column_stack([rng.uniform(0, 12, (40, 2)), rng.uniform(1, 2, 40), rng.uniform(3, 6, 40), rng.uniform(-np.pi, np.pi, 40)])
        b = np.column_stack([rng.uniform(0, 12, (30, 2)), rng.uniform(1, 2, 30), rng.uniform(3, 6, 30), rng.uniform(-np.pi, np.pi, 30)])
        turn = np.array([0, 0, 0, 0, np.pi / 2])
        want = co.rbox_iou(a, b)  # what bev_amd.iou.iou_batch_rbox computes (kernel convention: h along the yaw)
        assert (want > 0.05).sum() > 10
        np.testing.assert_allclose(d3d.box.box2d_iou(a + turn, b + turn, method="rbox"), want, rtol=0, atol=1e-12)
        # the advisor's example: under d3d's convention (w along the yaw) two 1 x 4 boxes one unit apart across their length overlap 3/5
        two = np.array([[0.0, 0.0, 4.0, 1.0, 0.0], [1.0, 0.0, 4.0, 1.0, 0.0]])
        np.testing.assert_allclose(d3d.box.box2d_iou(two[:1], two[1:], method="rbox"), [[0.6]], atol=1e-12)
        np.testing.assert_allclose(co.rbox_iou(two[:1], two[1:]), [[0.0]], atol=1e-12)  # ... and not at all with h along the yaw
        with pytest.raises(NotImplementedError):
            d3d.box.box2d_iou(a, b)
        assert overlay.remove_d3d_stand_in() and "d3d" not in sys.modules and not overlay.remove_d3d_stand_in()
    finally:
        sys.modules.pop("d3d", None)
        sys.modules.pop("d3d.box", None)
        sys.modules.update(saved)


def test_patch_uninstall_removes_the_d3d_stand_in():
    import sys

    from bev_amd import overlay, patch
    saved = {k: sys.modules.pop(k) for k in ("d3d", "d3d.box") if k in sys.modules}
    try:
        if not overlay.ensure_d3d():
            pytest.skip("a real d3d is installed")
        assert getattr(sys.modules["d3d"], "__bev_amd_stand_in__", False)
        patch.uninstall()
        assert "d3d" not in sys.modules and "d3d.box" not in sys.modules
    finally:
        sys.modules.update(saved)


def test_bev_serves_the_names_the_reference_star_imports():
    """/root/reference/bev/__init__.py:1-6 star-imports six sub-packages, which binds every name of their __all__ on `bev`
    itself.  The overlay resolves those lazily: the ones it owns here, the rest from a co-installed reference."""
    import bev
    assert bev.homo_constr.__name__ == "bev.constructor.homo_constr" and bev.compo.__name__ == "bev.tool.compo"
    assert "homo_constr" in vars(bev)  # cached after the first lookup
    with pytest.raises(AttributeError):
        bev.no_such_name
