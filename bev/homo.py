from bev_amd.homo import *  # noqa: F401,F403
