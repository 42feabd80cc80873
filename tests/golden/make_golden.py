#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the upstream reference.

Runs ONLY in the build container (needs /root/reference).  Nothing of the
reference travels: the outputs are plain JSON data (inputs + expected outputs).

The reference imports `cv2` at module level in files whose numpy-only functions
never call it; an inert placeholder module lets those bodies import.  Any path
that would actually reach OpenCV (findHomography, warpPerspective, ...) raises
AttributeError on the placeholder and is therefore NOT captured here -- those
are pinned by analytic known-answer tests instead (tests/test_oracle_warp.py).
`np.float` was removed from numpy>=1.24; the reference still spells it.

Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.modules["cv2"] = types.ModuleType("cv2")  # inert placeholder, no attributes
    np.float = float  # noqa: spelling the reference uses
    sys.path.insert(0, REF)
    import bev  # noqa: F401
    return bev


def tolist(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().tolist()
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, (list, tuple)):
        return [tolist(v) for v in x]
    if isinstance(x, dict):
        return {k: tolist(v) for k, v in x.items()}
    return x


CALIB_FIELDS = ["K", "u_size", "v_size", "T", "vp1", "vp2", "pp", "height", "mode"]
BSPEC_FIELDS = ["u_size", "v_size", "u_axis", "v_axis", "x_size", "y_size", "x_min", "x_max",
                "y_min", "y_max", "u_min", "u_max", "v_min", "v_max"]


def calib_record(calib):
    rec = {k: tolist(getattr(calib, k)) for k in CALIB_FIELDS}
    rec["H_world_img"] = tolist(calib.gen_H_world_img())
    rec["center_world"] = tolist(calib.gen_center_in_world())
    return rec


def bspec_record(bspec):
    rec = {k: tolist(getattr(bspec, k)) for k in BSPEC_FIELDS}
    rec["corners_world"] = tolist(bspec.gen_bev_corners_in_world())
    return rec


def calib_variants(calib):
    out = {"base": calib_record(calib)}
    out["scale_nc_852x480"] = calib_record(calib.scale(False, 852, 480))
    out["scale_ac_852x480"] = calib_record(calib.scale(True, 852, 480))
    out["scale_ratio_nc_0.5_0.25"] = calib_record(calib.scale(False, scale_ratio_u=0.5, scale_ratio_v=0.25))
    out["scale_ratio_ac_0.5_0.25"] = calib_record(calib.scale(True, scale_ratio_u=0.5, scale_ratio_v=0.25))
    out["pad_8_4_8_4"] = calib_record(calib.pad(8, 4, 8, 4))
    out["pad_-3_5_0_-2"] = calib_record(calib.pad(-3, 5, 0, -2))
    out["flip_lr"] = calib_record(calib.flip(lr=True))
    out["flip_tb"] = calib_record(calib.flip(tb=True))
    out["flip_lr_tb"] = calib_record(calib.flip(lr=True, tb=True))
    return out


def main():
    bev = _import_reference()
    from bev import homo as rhomo
    from bev import rbox as rrbox
    from bev import rbox_torch as rrbox_t
    from bev.constructor import homo_constr as rhc
    from bev.constructor import homo_constr_utils as rhcu

    G = {}

    # ---- 1. homography algebra (bev/homo.py) ---------------------------------
    vp1 = np.array([1200.0, -300.0])
    vp2 = np.array([-2500.0, -150.0])
    pp = np.array([959.5, 539.5])
    homo = {}
    homo["vps_in"] = {"vp1": tolist(vp1), "vp2": tolist(vp2), "pp": tolist(pp), "height": 8,
                      "u_size": 1920, "v_size": 1080}
    homo["get_focal"] = rhomo.get_focal(vp1, vp2, pp)
    K, focal = rhomo.get_K_from_vps(vp1, vp2, pp)
    homo["get_K_from_vps"] = {"K": tolist(K), "focal": focal}
    homo["homo_from_vps"] = tolist(rhomo.homo_from_vps(vp1, vp2, 8, 1920, 1080, pp))
    homo["homo_from_vps_pp_none"] = tolist(rhomo.homo_from_vps(vp1, vp2, 8, 1920, 1080))
    H_iw = rhomo.homo_from_vps(vp1, vp2, 8, 1920, 1080, pp)
    v1, v2 = rhomo.get_vps_from_homo(H_iw)
    homo["get_vps_from_homo"] = {"vp1": tolist(v1), "vp2": tolist(v2)}
    rng = np.random.default_rng(3)
    K34 = np.concatenate([np.array([[800.0, 0, 640.0], [0, 790.0, 360.0], [0, 0, 1.0]]), np.zeros((3, 1))], axis=1)
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]]) @ \
        np.array([[1, 0, 0], [0, np.cos(1.1), -np.sin(1.1)], [0, np.sin(1.1), np.cos(1.1)]])
    t = np.array([0.5, -1.5, 12.0])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    homo["KRt_in"] = {"K34": tolist(K34), "R": tolist(R), "t": tolist(t), "T": tolist(T)}
    homo["homo_from_KRt_Rt"] = tolist(rhomo.homo_from_KRt(K34, R=R, t=t))
    homo["homo_from_KRt_T"] = tolist(rhomo.homo_from_KRt(K34[:, :3], Rt_homo=T))
    G["homo"] = homo

    # ---- 2. Calib (bev/calib.py) ----------------------------------------------
    calib_vps = bev.Calib(vp1=vp1.copy(), vp2=vp2.copy(), pp=pp.copy(), height=8, u_size=1920, v_size=1080)
    G["calib_from_vps"] = calib_variants(calib_vps)
    calib_vps_nopp = bev.Calib(vp1=vp1.copy(), vp2=vp2.copy(), height=8, u_size=1920, v_size=1080)
    G["calib_from_vps_pp_default"] = calib_record(calib_vps_nopp)
    for sub_id in (1, 4):
        G["calib_KoPER_%d" % sub_id] = calib_variants(rhc.preset_calib("KoPER", sub_id))
    # fx/fy/cx/cy + T route builds a float32 K (calib.py:74)
    fx, fy, cx, cy, T_k = rhcu.load_T("KoPER", 1)
    G["load_T_KoPER_1"] = {"fx": fx, "fy": fy, "cx": cx, "cy": cy, "T": tolist(T_k)}
    fx, fy, cx, cy, T_k = rhcu.load_T("KoPER", 4)
    G["load_T_KoPER_4"] = {"fx": fx, "fy": fy, "cx": cx, "cy": cy, "T": tolist(T_k)}

    # ---- 3. constants: load_pts ------------------------------------------------
    pts = {}
    for name, sub in (("lturn", 0), ("lturn", None), ("roundabout", None)):
        p3, p2 = rhcu.load_pts(name, 852, 480, sub)
        pts["%s_%s" % (name, sub)] = {"pts_3d": tolist(p3), "pts_2d": tolist(p2)}
    G["load_pts_852x480"] = pts

    # ---- 4. BEV specs from YAML (offset mode) with the from_vps calib ------------
    cfg_dir = os.path.join(REF, "bev", "constructor", "configs_bspec")
    ybs = {}
    for fn in sorted(os.listdir(cfg_dir)):
        with open(os.path.join(cfg_dir, fn)) as f:
            cfg = yaml.safe_load(f)
        ybs[fn] = {"cfg": cfg, "bspec": bspec_record(rhc.load_bspec_from_cfg(cfg, calib_vps))}
    G["bspec_yaml"] = ybs
    # other cfg modes
    cfg_abs = {"mode": "abs", "spec": {"u_size": 200, "v_size": 100, "u_axis": "x", "v_axis": "-y",
                                       "m_per_px": 0.25, "x_min": -3.0, "y_max": 7.5}}
    cfg_cen = {"mode": "centered", "spec": {"u_size": 320, "v_size": 640, "x_size": 80, "y_size": 40,
                                            "u_axis": "y", "v_axis": "-x"}}
    G["bspec_cfg_abs"] = {"cfg": cfg_abs, "bspec": bspec_record(rhc.load_bspec_from_cfg(cfg_abs))}
    G["bspec_cfg_centered"] = {"cfg": cfg_cen, "bspec": bspec_record(rhc.load_bspec_from_cfg(cfg_cen, calib_vps))}
    G["cfg_path_from_dataset_id"] = {
        "6.1": os.path.basename(rhc.cfg_path_from_dataset_id("BrnoCompSpeed", 6.1)),
        "0": os.path.basename(rhc.cfg_path_from_dataset_id("BrnoCompSpeed", 0)),
        "0.5": os.path.basename(rhc.cfg_path_from_dataset_id("BrnoCompSpeed", 0.5)),
    }

    # ---- 5. preset_bspec ----------------------------------------------------------
    pres = {}
    combos = [("lturn", 0), ("lturn", None), ("KoPER", 1), ("KoPER", 4), ("kitti", None),
              ("roundabout", 0), ("roundabout", 1), ("rounD", 0), ("rounD", 2), ("rounD", 5),
              ("rounD_raw", 2)]
    combos += [("CARLA", s) for s in (1, 1.2, 2.1, 2.9, 2.2, 2.8, 3.1, 3.2, 3.8, 3.3, 3.9, 4.2, 4.8, 4.3,
                                      4.4, 4.9, 5.8, 5.9, 6.9, 7.9)]
    for name, sub in combos:
        pres["%s|%s" % (name, sub)] = bspec_record(rhc.preset_bspec(name, sub))
    for sub in (0, 4.1, 4.2, 4.3, 5.1, 5.2, 5.3, 6.1, 6.2, 6.3):
        pres["BrnoCompSpeed|%s|calib" % sub] = bspec_record(rhc.preset_bspec("BrnoCompSpeed", sub, calib_vps))
    pres["BrnoCompSpeed|0|nocalib"] = bspec_record(rhc.preset_bspec("BrnoCompSpeed", 0))
    G["preset_bspec"] = pres

    # ---- 6. BEVWorldSpec: axis combos, scale/pad/flip -------------------------------
    axes = {}
    for ua, va in (("x", "y"), ("x", "-y"), ("-x", "-y"), ("-x", "y"), ("y", "x"), ("y", "-x"), ("-y", "-x"), ("-y", "x")):
        b = bev.BEVWorldSpec(u_size=320, v_size=640, u_axis=ua, v_axis=va, x_min=-3.5, x_size=80.0, y_max=11.25, y_size=40.0)
        axes["%s|%s" % (ua, va)] = bspec_record(b)
    G["bspec_axes"] = axes
    b0 = bev.BEVWorldSpec(u_size=320, v_size=640, u_axis="y", v_axis="-x", x_min=-23.25, x_size=80.0, y_min=-16.5, y_size=40.0)
    tr = {"base": bspec_record(b0)}
    tr["scale_nc_160x320"] = bspec_record(b0.scale(False, 160, 320))
    tr["scale_ac_161x321"] = bspec_record(b0.scale(True, 161, 321))
    tr["scale_ratio_nc_1.5_0.75"] = bspec_record(b0.scale(False, scale_ratio_u=1.5, scale_ratio_v=0.75))
    tr["pad_8_4_16_12"] = bspec_record(b0.pad(8, 4, 16, 12))
    tr["pad_then_scale"] = bspec_record(b0.pad(8, 4, 16, 12).scale(False, 172, 328))
    tr["scale_then_pad"] = bspec_record(b0.scale(False, 160, 320).pad(1, 2, 3, 4))
    tr["flip_lr"] = bspec_record(b0.flip(lr=True))
    tr["flip_tb"] = bspec_record(b0.flip(tb=True))
    tr["flip_lr_tb"] = bspec_record(b0.flip(lr=True, tb=True))
    G["bspec_transforms"] = tr

    # ---- 7. file loaders ----------------------------------------------------------------
    with tempfile.TemporaryDirectory() as td:
        carla_line = "1280 720 90 -45.5 20.25 12.0 0.0 -30.0 115.0\n"
        p = os.path.join(td, "carla.txt")
        with open(p, "w") as f:
            f.write(carla_line)
        Kc, Tc, us, vs = rhcu.load_calib_from_file_carla(p)
        G["carla_file"] = {"text": carla_line, "K": tolist(Kc), "T_cam_world": tolist(Tc), "u_size": us, "v_size": vs,
                           "calib": calib_record(rhc.load_calib("CARLA", p))}
        blender_txt = ("K: 1050.0 0.0 480.0 0.0 0.0 1050.0 270.0 0.0 0.0 0.0 1.0 0.0\n"
                       "cam_pos_inv: 1.0 0.0 0.0 0.5 0.0 -0.5 -0.8660254037844386 1.25 0.0 0.8660254037844386 -0.5 20.0 0.0 0.0 0.0 1.0\n"
                       "note: synthetic\n")
        p = os.path.join(td, "blender.txt")
        with open(p, "w") as f:
            f.write(blender_txt)
        Kb, Tb, us, vs = rhcu.load_calib_from_file_blender(p)
        G["blender_file"] = {"text": blender_txt, "K": tolist(Kb), "Rt": tolist(Tb), "u_size": int(us), "v_size": int(vs),
                             "calib": calib_record(rhc.load_calib("blender", p))}
        brno = {"vp1": [1200.0, -300.0], "vp2": [-2500.0, -150.0], "pp": [959.5, 539.5], "height": 8.0, "scale": 0.02}
        p = os.path.join(td, "brno.json")
        with open(p, "w") as f:
            json.dump({"camera_calibration": 0, **brno}, f)
        G["brno_file"] = {"json": {"camera_calibration": 0, **brno}, "calib": calib_record(rhc.load_calib("BrnoCompSpeed", p))}

    # ---- 8. point / rbox transforms (bev/rbox.py, bev/rbox_torch.py) -----------------------
    rng = np.random.default_rng(5)
    H_wi = np.array(G["calib_from_vps"]["base"]["H_world_img"])
    pts2 = rng.uniform(0, [1920, 1080], (64, 2))
    pts3 = np.concatenate([pts2[:16], rng.uniform(0.5, 2.0, (16, 1))], axis=1)
    rb = {"H_world_img": tolist(H_wi), "pts2": tolist(pts2), "pts3": tolist(pts3)}
    rb["pts_world_bev_2"] = tolist(rrbox.pts_world_bev(pts2, H_wi))
    rb["pts_world_bev_3"] = tolist(rrbox.pts_world_bev(pts3, H_wi))
    rb["pts_world_bev_1d"] = tolist(rrbox.pts_world_bev(pts2[0], H_wi))
    # similarity H (what gen_H_world_bev yields): bev(320x640, u=y, v=-x) -> world
    s = 0.125
    H_wb = np.array([[0.0, -s, 56.5], [s, 0.0, -16.5], [0.0, 0.0, 1.0]])
    H_wb_refl = np.array([[0.0, s, -3.5], [s, 0.0, -16.5], [0.0, 0.0, 1.0]]) * 2.0  # unnormalised + reflection
    boxes_bev = np.stack([rng.uniform(0, 320, 32), rng.uniform(0, 640, 32), rng.uniform(12, 20, 32),
                          rng.uniform(28, 48, 32), rng.uniform(-np.pi, np.pi, 32)], axis=1)
    rb["H_world_bev"] = tolist(H_wb)
    rb["H_world_bev_refl"] = tolist(H_wb_refl)
    rb["boxes_bev"] = tolist(boxes_bev)
    bw = rrbox.rbox_world_bev(boxes_bev, H_wb, "bev")
    rb["rbox_world_bev__bev2world"] = tolist(bw)
    rb["rbox_world_bev__world2bev"] = tolist(rrbox.rbox_world_bev(bw, np.linalg.inv(H_wb), "world"))
    rb["rbox_world_bev__bev2world_refl"] = tolist(rrbox.rbox_world_bev(boxes_bev, H_wb_refl, "bev"))
    rb["rbox_world_bev_torch__bev2world"] = tolist(rrbox_t.rbox_world_bev(torch.from_numpy(boxes_bev), torch.from_numpy(H_wb), "bev"))
    rb["rbox_world_bev_torch__world2bev"] = tolist(rrbox_t.rbox_world_bev(torch.from_numpy(bw), torch.from_numpy(np.linalg.inv(H_wb)), "world"))
    rb["rbox_world_bev_torch_f32__bev2world"] = tolist(rrbox_t.rbox_world_bev(torch.from_numpy(boxes_bev).float(), torch.from_numpy(H_wb).float(), "bev"))
    rb["rbox_world_img"] = tolist(rrbox.rbox_world_img(bw, np.linalg.inv(H_wi)))
    for mode in ("bev", "world"):
        rb["xywhr2xyxy_%s" % mode] = tolist(rrbox.xywhr2xyxy(boxes_bev, mode))
        # external_aa=True raises IndexError in the reference (rbox.py:71,83 index a 4-column array at 4): no vector
        rb["xywhr2xyxy_torch_%s" % mode] = tolist(rrbox_t.xywhr2xyxy(torch.from_numpy(boxes_bev), mode))
        rb["xy82xywhr_%s" % mode] = tolist(rrbox.xy82xywhr(rrbox.xywhr2xyxy(boxes_bev, mode), mode))
        rb["xywhr2xyvec_%s" % mode] = tolist(rrbox.xywhr2xyvec(boxes_bev, mode))
        rb["xywhr2xyvec_torch_%s" % mode] = tolist(rrbox_t.xywhr2xyvec(torch.from_numpy(boxes_bev), mode))
        rb["yaw2v_%s" % mode] = tolist(rrbox.yaw2v(boxes_bev[:, 4], mode))
        rb["v2yaw_%s" % mode] = tolist(rrbox.v2yaw(rrbox.yaw2v(boxes_bev[:, 4], mode), mode))
        rb["yaw2mat_%s" % mode] = tolist(rrbox.yaw2mat(boxes_bev[:, 4], mode))
        rb["angle_world_bev_src_%s" % mode] = tolist(rrbox.angle_world_bev(boxes_bev[:, 4], H_wb, mode))
    rb["xy82xyvec"] = tolist(rrbox.xy82xyvec(rrbox.xywhr2xyxy(boxes_bev, "bev")))
    rb["xy82xyvec_torch"] = tolist(rrbox_t.xy82xyvec(torch.from_numpy(rrbox.xywhr2xyxy(boxes_bev, "bev"))))
    rb["dist_world_bev"] = tolist(rrbox.dist_world_bev(boxes_bev[:, 2:4], H_wb))
    # 3D-tail variants (rbox.py:228-314)
    K3 = np.array([[800.0, 0, 640.0], [0, 790.0, 360.0], [0, 0, 1.0]])
    rboxzt = np.concatenate([bw, rng.uniform(0.0, 0.3, (32, 1)), rng.uniform(1.2, 2.0, (32, 1))], axis=1)
    rb["rboxzt_in"] = tolist(rboxzt)
    rb["K3"] = tolist(K3)
    rb["Rt"] = tolist(T)
    rb["rbox_zt2tt_world"] = tolist(rrbox.rbox_zt2tt_world(rboxzt.copy(), K3, T))
    H_int = np.array([[0.0, -8.0, 452.0], [8.0, 0.0, 132.0], [0.0, 0.0, 1.0]])  # exact-arithmetic similarity
    rboxtt = np.concatenate([np.round(bw * 4) / 4, np.round(rng.uniform(-1, 1, (32, 2)) * 8) / 8], axis=1)
    rb["H_bev_world_int"] = tolist(H_int)
    rb["rboxtt_in"] = tolist(rboxtt)
    rb["rboxtt_world_bev"] = tolist(rrbox.rboxtt_world_bev(rboxtt, H_int, "world"))
    G["rbox"] = rb

    out = os.path.join(HERE, "reference_vectors.json")
    with open(out, "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
