from bev_amd.calib import Calib  # noqa: F401
