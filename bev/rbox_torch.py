from bev_amd.rbox_torch import (rbox_world_bev, v2yaw, xy82xyvec, xywhr2xyvec, xywhr2xyxy, yaw2mat,  # noqa: F401
                                yaw2v)
