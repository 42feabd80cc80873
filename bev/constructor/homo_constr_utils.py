from bev_amd.constructor.homo_constr_utils import (R_from_euler_carla, load_calib_from_file_blender,  # noqa: F401
                                                   load_calib_from_file_carla, load_pts, load_spec_dict_bev, load_T,
                                                   load_vps_from_file_BrnoCompSpeed)
