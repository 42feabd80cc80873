from bev_amd.cv2_compat import *  # noqa: F401,F403
from bev_amd.cv2_compat import (BORDER_CONSTANT, INTER_LINEAR, INTER_NEAREST, WARP_INVERSE_MAP, findHomography, invert,  # noqa: F401
                                perspectiveTransform, resize, warpPerspective)
