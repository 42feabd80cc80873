"""Device composites: the second in-repo caller of the warp (reference bev/tool/compo.py:5-49).

composite_reg_img   alpha blend  fg * m + bg * (1 - m), m = mask / 255, rounded and clipped to uint8 (compo.py:5-24)
composite_bev_img   warp background, foreground and mask into the BEV (three warps through the HIP path) and blend
                    (compo.py:26-49).  All pixels stay on the GPU; the blend is one HIP launch in float64 like the reference
                    (bevwarp_composite).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .homo import homo_from_KRt
from .warp import warp_perspective


def _as_cuda(img, device):
    if isinstance(img, torch.Tensor):
        return img.to(device)
    return torch.from_numpy(np.ascontiguousarray(img)).to(device)


def composite_reg_img(bg, fg, fg_mask, bw_mode=False, device="cuda"):
    """uint8 HWC images (numpy or tensors) -> uint8 CUDA tensor.  bw_mode is not supported (the reference needs
    cv2.cvtColor for it)."""
    if bw_mode:
        raise NotImplementedError("bw_mode needs a BGR->gray conversion outside the warp path")
    bg, fg, fg_mask = (_as_cuda(x, device).contiguous() for x in (bg, fg, fg_mask))
    if not (bg.dtype == fg.dtype == fg_mask.dtype == torch.uint8) or not (bg.shape == fg.shape == fg_mask.shape):
        raise ValueError("composite_reg_img needs three uint8 images of one shape")
    out = torch.empty_like(bg)
    stream = torch.cuda.current_stream(bg.device).cuda_stream
    with torch.cuda.device(bg.device):
        _lib.check(_lib.load().bevwarp_composite(bg.data_ptr(), fg.data_ptr(), fg_mask.data_ptr(), out.data_ptr(), bg.numel(),
                                                 ctypes.c_void_p(stream)))
    return out


def composite_bev_img(bg, fg, fg_mask, H_world2bev, H_img2world_fix, K, RT, x_size, y_size, bw_mode=False, device="cuda"):
    """Returns (compo uint8 CUDA tensor of shape (y_size, x_size, C), H_world2img_cam) like the reference."""
    bg, fg, fg_mask = (_as_cuda(x, device) for x in (bg, fg, fg_mask))
    H_img2bev_fix = np.asarray(H_world2bev).dot(H_img2world_fix)
    bg_bev = warp_perspective(bg, H_img2bev_fix, (x_size, y_size))
    H_world2img_cam = homo_from_KRt(np.asarray(K), Rt_homo=np.asarray(RT))
    H_img2bev_cam = np.asarray(H_world2bev).dot(np.linalg.inv(H_world2img_cam))
    # foreground and mask share one homography: warp them as a batch of two
    both = torch.stack([fg, fg_mask]) if fg.shape == fg_mask.shape else None
    if both is not None:
        w = warp_perspective(both, H_img2bev_cam, (x_size, y_size))
        fg_bev, mask_bev = w[0], w[1]
    else:
        fg_bev = warp_perspective(fg, H_img2bev_cam, (x_size, y_size))
        mask_bev = warp_perspective(fg_mask, H_img2bev_cam, (x_size, y_size))
    return composite_reg_img(bg_bev, fg_bev, mask_bev, bw_mode=bw_mode, device=device), H_world2img_cam
