from bev_amd.warp import (BORDER_CONSTANT, INTER_LINEAR, INTER_NEAREST, WARP_INVERSE_MAP, footprint,  # noqa: F401
                          invert_homography, resize_matrix, warp_perspective, warp_perspective_resized, warp_to_planar,
                          warpPerspective)
from bev_amd.resize import cv2_resize, resize  # noqa: F401,E402
