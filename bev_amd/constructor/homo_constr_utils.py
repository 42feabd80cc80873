"""Per-dataset calibration constants and file loaders feeding Calib / BEVWorldSpec.

Same public functions as /root/reference/bev/constructor/homo_constr_utils.py (load_pts :5-80,
load_T :82-106, load_spec_dict_bev :108-474, R_from_euler_carla :476-501,
load_calib_from_file_carla :503-531, load_calib_from_file_blender :533-541,
load_vps_from_file_BrnoCompSpeed :543-551).  The numbers are the reference's measured facts
(surveyed ground points, camera poses, BEV windows); here they live in lookup tables.
"""
import json

import numpy as np

from ..io_utils import read_txt_to_array, read_txt_to_dict

# ---------------------------------------------------------------------------------------------
# ground-plane control points (metres) and their pixels, picked on 1920x1080 screen captures
# ---------------------------------------------------------------------------------------------
_LTURN_LEGACY_WORLD = [[0, 0, 0], [3.7, 0, 0], [7.4, 0, 0], [-0.87, 21.73, 0], [-4.26, 21.73, 0], [-5.22, -2.17, 0]]
_LTURN_LEGACY_PIXEL = [[1575, 611], [1428, 608], [1256, 605], [1066, 876], [1368, 924], [1866, 601]]
# 2020-12-24 survey; stored as (a, b, 0) and used as (x, y) = (-b, -a)
_LTURN_SURVEY = [
    [0, 0, 0],
    [-0.143540669856454, -3.44497607655502, 0],
    [-0.143540669856454, -7.79904306220095, 0],
    [28.6124401913875, -3.58851674641148, 0],
    [25.5502392344497, -7.36842105263157, 0],
    [8.22966507177034, -12.0095693779904, 0],
    [1.05263157894737, -11.9617224880383, 0],
    [-3.15789473684210, -15.7416267942584, 0],
    [-9.52153110047846, -14.8325358851675, 0],
    [-22.5837320574162, -7.12918660287081, 0],
    [-23.1100478468899, 1.29186602870813, 0],
    [-23.0622009569378, 4.64114832535885, 0],
    [-23.2057416267942, 7.84688995215311, 0],
    [-26.9856459330143, -1.91387559808612, 0],
    [-0.191387559808609, 6.60287081339713, 0],
    [2.67942583732058, 5.45454545454545, 0],
]
_LTURN_SURVEY_PIXEL_1BASED = [
    [1572.17701863354, 609.071428571429],
    [1423.10869565217, 608.077639751553],
    [1253.17080745342, 604.102484472050],
    [1680.50000000000, 517.642857142857],
    [1569.19565217391, 521.618012422360],
    [1256.15217391304, 569.319875776398],
    [1139.87888198758, 594.164596273292],
    [911.307453416149, 606.090062111801],
    [755.282608695652, 636.897515527950],
    [514.785714285714, 786.959627329193],
    [1069.31987577640, 872.425465838509],
    [1368.45031055901, 916.152173913044],
    [1704.35093167702, 975.779503105590],
    [398.512422360248, 975.779503105590],
    [1879.25776397516, 614.040372670808],
    [1832.54968944099, 595.158385093168],
]
_ROUNDABOUT_WORLD = [[0, 0, 0], [23.75, 0, 0], [0, 25.75, 0], [0, -29.25, 0], [-18, 0, 0], [18.75, 20, 0], [-17, 10, 0],
                     [32.75, -10, 0], [-17.25, -15.5, 0]]
_ROUNDABOUT_PIXEL = [[1068, 593], [741, 503], [29, 682], [1565, 549], [1552, 730], [293, 555], [1140, 835], [837, 477],
                     [1867, 655]]


def load_pts(name, img_width, img_height, sub_id=None):
    """(pts_3d [n x 3, z = 0], pts_2d [n x 2]) float32, pixels rescaled from 1920x1080 to the given size."""
    if name == "lturn":
        assert sub_id is None or sub_id == 0
        if sub_id == 0:
            pts_3d = np.array(_LTURN_LEGACY_WORLD, dtype=np.float32)
            pts_2d = np.array(_LTURN_LEGACY_PIXEL, dtype=np.float32)
        else:
            surveyed = np.array(_LTURN_SURVEY, dtype=np.float32)
            pts_3d = surveyed.copy()
            pts_3d[:, 0] = -surveyed[:, 1]
            pts_3d[:, 1] = -surveyed[:, 0]
            pts_2d = np.array(_LTURN_SURVEY_PIXEL_1BASED, dtype=np.float32) - 1
    elif name == "roundabout":
        pts_3d = np.array(_ROUNDABOUT_WORLD, dtype=np.float32)
        pts_2d = np.array(_ROUNDABOUT_PIXEL, dtype=np.float32)
    else:
        raise ValueError("cam_name not recognized", name)
    pts_2d[:, 0] = pts_2d[:, 0] / 1920 * img_width
    pts_2d[:, 1] = pts_2d[:, 1] / 1080 * img_height
    return pts_3d, pts_2d


# KoPER intersection cameras: world -> camera pose (row-major 4x4) and pinhole intrinsics
_KOPER = {
    1: dict(T=[-0.998701024990115, -0.0243198486052637, 0.0447750784199287, 12.0101713926222,
               -0.0488171908715811, 0.708471062121443, -0.704049455657713, -3.19544493781655,
               -0.0145994711925239, -0.705320706558607, -0.708738002607851, 18.5953697835002,
               0, 0, 0, 1], fx=336.2903, fy=335.5113, cx=321.3685, cy=251.1326),
    4: dict(T=[0.916927873702706, 0.399046485499693, -0.00227526644131446, -9.32023173352383,
               0.287745260046085, -0.665109208127767, -0.689080841835460, -5.17417993923343,
               -0.276488588820672, 0.631182733979628, -0.724680906729269, 17.1155540514235,
               0, 0, 0, 1], fx=331.2292, fy=330.4413, cx=325.4500, cy=252.1456),
}


def load_T(name, cam_id):
    if name != "KoPER":
        raise ValueError("cam_name not recognized", name)
    assert cam_id in [1, 4]
    cam = _KOPER[cam_id]
    return cam["fx"], cam["fy"], cam["cx"], cam["cy"], np.array(cam["T"], dtype=float).reshape(4, 4)


# ---------------------------------------------------------------------------------------------
# BEV world windows.  Each entry is the keyword dict BEVWorldSpec receives besides u_size / v_size.
# ---------------------------------------------------------------------------------------------
def _win(u_axis, v_axis, **kw):
    kw.update(u_axis=u_axis, v_axis=v_axis)
    return kw


_CARLA_WINDOWS = [
    # (camera ids, x_min, y_min, square side [m], u_axis, v_axis)
    ((1,), -101.76161565095929, 111.53369067537298, 60, "-x", "-y"),
    ((1.2,), -103.76161565095929, 113.53369067537298, 55, "-x", "-y"),
    ((2.1, 2.9), -104.0157, -25.07683, 50, "y", "-x"),
    ((2.2, 2.8), -99.0157, -25.07683, 45, "y", "-x"),
    ((3.1,), 188.47, -339.23, 40, "y", "-x"),
    ((3.2, 3.8), 193.47, -334.23, 40, "y", "-x"),
    ((3.3, 3.9), 193.47, -339.23, 45, "y", "-x"),
    ((4.2, 4.8), -92.72, -150.01, 50, "-x", "-y"),
    ((4.3, 4.4, 4.9), -92.72, -150.01, 60, "-x", "-y"),
    ((5.8, 5.9), -11.43 - 20, 187.22 - 45, 70, "y", "-x"),
    ((6.9,), -92.72, -150.01, 70, "-x", "-y"),
    ((7.9,), -104.0157, -45.07683, 68, "x", "y"),
]

# BrnoCompSpeed: window = ground point under the image centre + offsets (4 m grid)
_BRNO_WINDOWS = {
    # cam id: (x offset of x_min, y offset of y_min, x_size, y_size, u_axis)
    0: (-40, -22, 96, 48, "y"),
    4.1: (-11, -10, 52, 32, "y"),
    4.2: (-12, -10, 48, 32, "y"),
    4.3: (-14, -10, 64, 28, "-y"),
    5.1: (-13, -10, 56, 36, "y"),
    5.2: (-8, -8, 40, 24, "y"),
    5.3: (-8, -6, 56, 24, "-y"),
    6.1: (-28, -18, 80, 40, "y"),
    6.2: (-24, -10, 80, 36, "y"),
    6.3: (-14, -10, 64, 24, "-y"),
}

# rounD drone recordings: raster size and metres per pixel (x 10: scale_down_factor of drone-dataset-tools)
_ROUND_RASTER = {0: (1544, 936, 0.0148098329880904), 1: (1678, 936, 0.0136334127882737), 2: (1678, 936, 0.0101601513616589)}


def load_spec_dict_bev(u_size, v_size, name, cam_id=None, calib=None):
    assert name in ["lturn", "KoPER", "CARLA", "roundabout", "BrnoCompSpeed", "rounD", "rounD_raw"]
    spec = {"u_size": u_size, "v_size": v_size}

    if name == "lturn":
        y_min = -37 if cam_id == 0 else -53  # aspect 17/13 vs 21/13 at 4 px per metre
        spec.update(_win("-x", "y", x_min=-10, x_max=42, y_min=y_min, y_max=31))

    elif name == "KoPER":
        if cam_id == 1:
            x_size, x_max, y_min, axes = 60, 45, -30, ("-x", "y")
        elif cam_id == 4:
            x_size, x_max, y_min, axes = 50, 30, -13, ("x", "-y")
        else:
            raise ValueError("cam_id not recogized")
        spec.update(_win(*axes, x_max=x_max, x_size=x_size, y_min=y_min, y_size=float(v_size) / u_size * x_size))

    elif name == "roundabout":
        if cam_id == 0:
            spec.update(_win("-y", "-x", x_min=-23.97, x_size=60, y_min=-33.57, y_size=60))
        else:
            spec.update(_win("-y", "-x", x_min=-25, x_size=70, y_min=-43, y_size=70))

    elif name in ("rounD", "rounD_raw"):
        if name == "rounD_raw":
            key = 2 if cam_id == 2 else None
        else:
            key = cam_id if cam_id in (0, 1) else (2 if cam_id >= 2 else None)
        if key is None:
            raise ValueError("cam_id {} not recognized. ".format(cam_id))
        raster_u, raster_v, m_per_px = _ROUND_RASTER[key]
        x_size = raster_u * m_per_px * 10
        y_size = raster_v * m_per_px * 10
        if name == "rounD" and key == 1:
            # the reference leaves x_min / y_max / axes unset for recording 1 (NameError there)
            raise ValueError("cam_id {} has no BEV window in the reference. ".format(cam_id))
        if name == "rounD" and key == 2:
            x_min, y_max = 96 - x_size / 2, -17.2 + y_size / 2
        else:
            x_min, y_max = 0, 0
        spec.update(_win("x", "-y", x_min=x_min, x_size=x_size, y_max=y_max, y_size=y_size))

    elif name == "CARLA":
        for ids, x_min, y_min, side, u_axis, v_axis in _CARLA_WINDOWS:
            if cam_id in ids:
                spec.update(_win(u_axis, v_axis, x_min=x_min, y_min=y_min, x_size=side, y_size=side))
                break

    elif name == "BrnoCompSpeed":
        if calib is None:
            spec.update(_win("x", "y", x_min=-34, y_min=-34, x_size=68, y_size=68))
        elif cam_id in _BRNO_WINDOWS:
            center = calib.gen_center_in_world()
            dx, dy, x_size, y_size, u_axis = _BRNO_WINDOWS[cam_id]
            spec.update(_win(u_axis, "-x", x_min=center[0] + dx, y_min=center[1] + dy, x_size=x_size, y_size=y_size))

    return spec


def R_from_euler_carla(roll, pitch, yaw):
    """Rotation for CARLA's left-handed (front, right, up) frame from degrees
    (https://carla.readthedocs.io/en/latest/python_api/#carla.Rotation)."""
    roll, pitch, yaw = (a * np.pi / 180 for a in (roll, pitch, yaw))
    cy, sy = np.cos(yaw), np.sin(yaw)
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    return np.array([[cp * cy, cy * sp * sr - sy * cr, -cy * sp * cr - sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, -sy * sp * cr + cy * sr],
                     [sp, -cp * sr, cp * cr]])


def load_calib_from_file_carla(fpath):
    """One line "u_size v_size fov x y z roll pitch yaw" -> (K, T_cam_world, u_size, v_size)."""
    u_size, v_size, fov_deg, x, y, z, roll, pitch, yaw = read_txt_to_array(fpath).reshape(-1)[:9]
    ux, uy = u_size * 0.5, v_size * 0.5
    fx = ux / np.tan(fov_deg * np.pi / 180 * 0.5)
    K = np.array([[fx, 0, ux], [0, fx, uy], [0, 0, 1]], dtype=float)

    T_world_cam = np.eye(4)
    T_world_cam[:3, :3] = R_from_euler_carla(roll, pitch, yaw)
    T_world_cam[:3, 3] = (x, y, z)
    # CARLA (front, right, up) -> camera (right, down, front)
    axes = np.array([[0, 1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=float)
    return K, axes.dot(np.linalg.inv(T_world_cam)), u_size, v_size


def load_calib_from_file_blender(fpath):
    data = read_txt_to_dict(fpath)
    Rt = data["cam_pos_inv"].reshape(4, 4)
    K = data["K"].reshape(3, 4)[:, :3]
    return K, Rt, (K[0, 2] * 2).round().astype(int), (K[1, 2] * 2).round().astype(int)


def load_vps_from_file_BrnoCompSpeed(fpath):
    """BrnoCompSpeed system_dubska_*.json: lists become arrays (keys used: vp1, vp2, pp, height)."""
    with open(fpath) as f:
        calibration = json.load(f)
    return {k: (np.array(v) if isinstance(v, list) else v) for k, v in calibration.items()}
