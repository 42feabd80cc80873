from bev_amd.rbox import (angle_world_bev, dist_world_bev, pts_world_bev, rbox_world_bev, rbox_world_img,  # noqa: F401
                          rbox_zt2tt_world, rboxtt_world_bev, rboxzt_world_bev, v2yaw, xy82xyvec, xy82xywhr,
                          xywhr2xyvec, xywhr2xyxy, yaw2mat, yaw2v)
