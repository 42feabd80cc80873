"""Dataset presets and config loaders producing Calib / BEVWorldSpec objects.

Same entry points as /root/reference/bev/constructor/homo_constr.py: preset_calib :10-27,
load_calib :29-43, preset_bspec :46-154, cfg_path_from_dataset_id :156-168,
load_bspec_from_cfg :170-221, load_bspec :223-230 (the call chain of vis_homo.py:57-59).
YAML is read with yaml.safe_load (the reference's bare yaml.load(f) is an error on PyYAML >= 6).
"""
import copy
import os

import yaml

from ..bevspec import BEVWorldSpec
from ..calib import Calib
from .homo_constr_utils import (load_calib_from_file_blender, load_calib_from_file_carla, load_pts,
                                load_spec_dict_bev, load_T, load_vps_from_file_BrnoCompSpeed)

_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs_bspec")


def preset_calib(dataset_name, sub_id=None):
    assert dataset_name in ["lturn", "KoPER", "roundabout"]
    if dataset_name == "KoPER":
        assert sub_id in [1, 4]
        fx, fy, cx, cy, T = load_T(dataset_name, sub_id)
        return Calib(fx=fx, fy=fy, cx=cx, cy=cy, T=T, u_size=656, v_size=494)
    width, height = 852, 480
    pts_3d, pts_2d = load_pts(dataset_name, width, height, sub_id)
    return Calib(pts_world=pts_3d, pts_image=pts_2d, u_size=width, v_size=height)


def load_calib(dataset_name, fpath):
    if dataset_name in ("CARLA", "blender"):
        loader = load_calib_from_file_carla if dataset_name == "CARLA" else load_calib_from_file_blender
        K, T_cam_world, u_size, v_size = loader(fpath)
        return Calib(K=K, T=T_cam_world, u_size=int(u_size), v_size=int(v_size))
    if dataset_name == "BrnoCompSpeed":
        cal = load_vps_from_file_BrnoCompSpeed(fpath)
        return Calib(vp1=cal["vp1"], vp2=cal["vp2"], pp=cal["pp"], height=cal["height"], u_size=1920, v_size=1080)
    raise ValueError("dataset_name {} not recognized", dataset_name)


# BEV raster sizes (u, v) per dataset / camera
_RASTER = {
    "KoPER": (544, 416), "CARLA": (544, 544), "roundabout": (544, 544), "kitti": (224, 544),
}
_RASTER_BRNO = {0: (384, 768), 4.1: (256, 416), 4.2: (256, 384), 4.3: (224, 512), 5.1: (288, 448), 5.2: (192, 320),
                5.3: (192, 448), 6.1: (320, 640), 6.2: (288, 640), 6.3: (192, 512)}


def preset_bspec(dataset_name, sub_id=None, calib=None):
    assert dataset_name in ["lturn", "KoPER", "kitti", "CARLA", "roundabout", "BrnoCompSpeed", "rounD", "rounD_raw"]
    if dataset_name == "kitti":
        u, v = _RASTER["kitti"]
        return BEVWorldSpec(u_size=u, v_size=v, u_axis="x", v_axis="-y", x_min=-14, x_max=14, y_min=6, y_max=74)
    if dataset_name == "KoPER":
        assert sub_id in [1, 4]
    if dataset_name == "lturn":
        u, v = (416, 544) if sub_id == 0 else (416, 672)
    elif dataset_name in ("rounD", "rounD_raw"):
        if sub_id == 0:
            u, v = 1544, 936
        elif sub_id >= 1:
            u, v = 1678, 936
        else:
            raise ValueError("sub_id {} not recognized. ".format(sub_id))
    elif dataset_name == "BrnoCompSpeed":
        u, v = _RASTER_BRNO[sub_id]
    else:
        u, v = _RASTER[dataset_name]
    kwargs = {"calib": calib} if dataset_name == "BrnoCompSpeed" else {}
    return BEVWorldSpec(**load_spec_dict_bev(u, v, dataset_name, sub_id, **kwargs))


def cfg_path_from_dataset_id(dataset_name, sub_id):
    """configs_bspec/<dataset>_<id with '.' -> '_'>.yaml, falling back to the integer part of the id."""
    for id_str in (str(sub_id).replace(".", "_"), str(sub_id).split(".")[0]):
        fpath = os.path.join(_CFG_DIR, "{}_{}.yaml".format(dataset_name, id_str))
        if os.path.exists(fpath):
            return fpath
    assert os.path.exists(fpath), "Not exist: {}".format(fpath)
    return fpath


def load_bspec_from_cfg(cfg, calib=None):
    """cfg = {"mode": "abs" | "offset" | "centered", "spec": {...BEVWorldSpec kwargs...}}.

    spec may carry `m_per_px` (x_size / y_size derived from the raster size along that axis);
    "offset": x_min_off / x_max_off / y_min_off / y_max_off are relative to the ground point under the
    image centre (needs calib); "centered": the window is centred on that point (needs calib)."""
    mode = cfg["mode"]
    spec = copy.deepcopy(cfg["spec"])

    if "m_per_px" in spec:
        m_per_px = spec.pop("m_per_px")
        spec["x_size"] = (spec["u_size"] if "x" in spec["u_axis"] else spec["v_size"]) * m_per_px
        spec["y_size"] = (spec["u_size"] if "y" in spec["u_axis"] else spec["v_size"]) * m_per_px

    if mode == "abs":
        pass
    elif mode == "offset":
        assert calib is not None, "when using `offset` mode, \
            `calib` must be given to calculate the world coordinate of the center of the original view image"
        center = calib.gen_center_in_world()
        assert ("x_max_off" in spec or "x_min_off" in spec), "when using `offset` mode, \
            either `x_min_off` or `x_max_off` must be given. "
        assert ("y_max_off" in spec or "y_min_off" in spec), "when using `offset` mode, \
            either `y_min_off` or `y_max_off` must be given. "
        for axis, c in (("x", center[0]), ("y", center[1])):
            for bound in ("min", "max"):
                off = "{}_{}_off".format(axis, bound)
                if off in spec:
                    spec["{}_{}".format(axis, bound)] = c + spec.pop(off)
    elif mode == "centered":
        assert calib is not None, "when using `centered` mode, \
            `calib` must be given to calculate the world coordinate of the center of the original view image"
        center = calib.gen_center_in_world()
        spec["x_min"] = center[0] - spec["x_size"] * 0.5
        spec["y_min"] = center[1] - spec["y_size"] * 0.5
    else:
        raise ValueError("mode {} not recognized".format(mode))
    return BEVWorldSpec(**spec)


def load_bspec(dataset_name, sub_id=None, calib=None):
    """BEVWorldSpec from the YAML config of this dataset / camera id."""
    with open(cfg_path_from_dataset_id(dataset_name, sub_id)) as f:
        cfg = yaml.safe_load(f)
    return load_bspec_from_cfg(cfg, calib)
