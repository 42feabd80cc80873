from bev_amd.frozen_class import FrozenClass  # noqa: F401
