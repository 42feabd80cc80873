from bev_amd.compo import composite_bev_img, composite_reg_img  # noqa: F401
