from bev_amd.bevspec import BEVWorldSpec  # noqa: F401
