from bev_amd.constructor.homo_constr import (cfg_path_from_dataset_id, load_bspec, load_bspec_from_cfg,  # noqa: F401
                                             load_calib, preset_bspec, preset_calib)
