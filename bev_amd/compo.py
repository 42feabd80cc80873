"""Device composites: the second in-repo caller of the warp (reference bev/tool/compo.py:5-49).

composite_reg_img   alpha blend  fg * m + bg * (1 - m), m = mask / 255, rounded and clipped to uint8 (compo.py:5-24):
                    one HIP launch in float64 like the reference's numpy expression (bevwarp_composite)
composite_bev_img   background, foreground and mask warped into the BEV and blended (compo.py:26-49) in ONE launch, bw_mode's
                    grey conversion of the foreground included (tap by tap, inside the kernel)
                    (bevwarp_warp_composite): every BEV pixel samples the background through H_img2bev_fix and the
                    foreground + its mask through H_img2bev_cam and blends in registers; no warped image is ever written.
                    Bit-identical to three bevwarp_warp calls followed by bevwarp_composite.

Both take numpy arrays or CUDA tensors (uint8, HWC).  Like the reference they return numpy when every image came in as
numpy; when any image is a tensor the result stays on the device.  Image paths are not accepted (the reference reads them
with cv2.imread: decoding is outside this path).
"""
import collections
import ctypes

import numpy as np
import torch

from . import _lib
from .homo import homo_from_KRt
from .warp import device_inverse


def _as_cuda(img, device):
    if isinstance(img, str):
        raise ValueError("image paths are not supported: decode the file and pass the array (the reference uses cv2.imread)")
    if isinstance(img, torch.Tensor):
        return img.to(device)
    return torch.from_numpy(np.ascontiguousarray(img)).to(device)


def _hwc(t):
    if t.dtype != torch.uint8 or t.dim() not in (2, 3):
        raise ValueError("images must be uint8 (H, W) or (H, W, C), got %s %s" % (tuple(t.shape), t.dtype))
    t = t if t.dim() == 3 else t[:, :, None]
    return t if (t.stride(2) == 1 and t.stride(1) == t.shape[2]) else t.contiguous()


def gray_bgr(fg):
    """cv2.cvtColor(cv2.cvtColor(fg, COLOR_BGR2GRAY), COLOR_GRAY2BGR) for uint8 BGR (the reference's bw_mode, compo.py:13-14):
    OpenCV's 14-bit fixed point, gray = (1868 B + 9617 G + 4899 R + 8192) >> 14, replicated to three channels.
    (Restated from OpenCV's color conversion; parity unpinned -- the reference holds no fixture for it.)"""
    if fg.dim() != 3 or fg.shape[2] != 3:
        raise ValueError("bw_mode needs a 3-channel BGR foreground (cv2.COLOR_BGR2GRAY), got shape %s" % (tuple(fg.shape),))
    f = fg.to(torch.int32)
    g = ((f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14).to(torch.uint8)
    return g[..., None].expand(-1, -1, 3).contiguous()


def _result(t, as_numpy):
    return t.cpu().numpy() if as_numpy else t


def composite_reg_img(bg, fg, fg_mask, bw_mode=False, device="cuda"):
    """uint8 HWC images of one shape -> their alpha composite (compo.py:5-24)."""
    as_numpy = not any(isinstance(x, torch.Tensor) for x in (bg, fg, fg_mask))
    bg, fg, fg_mask = (_as_cuda(x, device).contiguous() for x in (bg, fg, fg_mask))
    if not (bg.dtype == fg.dtype == fg_mask.dtype == torch.uint8) or not (bg.shape == fg.shape == fg_mask.shape):
        raise ValueError("composite_reg_img needs three uint8 images of one shape")
    if bw_mode:
        fg = gray_bgr(fg)
    out = torch.empty_like(bg)
    stream = torch.cuda.current_stream(bg.device).cuda_stream
    with torch.cuda.device(bg.device):
        _lib.check(_lib.load().bevwarp_composite(bg.data_ptr(), fg.data_ptr(), fg_mask.data_ptr(), out.data_ptr(), bg.numel(),
                                                 ctypes.c_void_p(stream)))
    return _result(out, as_numpy)


_maps_cache = collections.OrderedDict()  # (matrix bytes, device) -> (forward maps, H_world2img_cam), least recently used first
_MAPS_MAX = 64


def _composite_maps(H_world2bev, H_img2world_fix, K, RT, device):
    """The two inverse maps of compo.py:37-44 as one (2, 3, 3) device tensor, and H_world2img_cam (returned to the caller like
    the reference does).  Cached by value in two steps: the matrix algebra here (host), the inversion + upload in
    bev_amd.warp.device_inverse."""
    mats = [np.ascontiguousarray(m, dtype=np.float64) for m in (H_world2bev, H_img2world_fix, K, RT)]
    key = (b"".join(m.tobytes() for m in mats), tuple(m.shape for m in mats), str(device))
    hit = _maps_cache.get(key)
    if hit is None:
        H_world2bev, H_img2world_fix, K, RT = mats
        H_img2bev_fix = H_world2bev.dot(H_img2world_fix)
        H_world2img_cam = homo_from_KRt(K, Rt_homo=RT)
        H_img2bev_cam = H_world2bev.dot(np.linalg.inv(H_world2img_cam))
        fwd = np.stack([H_img2bev_fix, H_img2bev_cam])
        hit = _maps_cache[key] = [fwd, H_world2img_cam, device_inverse(fwd, device)]  # (device_inverse pins or refuses under capture)
        while len(_maps_cache) > _MAPS_MAX:
            _maps_cache.popitem(last=False)  # drops a reference only: a tensor a graph replays is held by device_inverse's pinned set
    else:
        _maps_cache.move_to_end(key)
        capturing = False
        if hit[2].is_cuda:
            with torch.cuda.device(hit[2].device):
                capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # a hipGraph capture bakes the tensor's ADDRESS into the graph: device_inverse's lifetime rules must see this use (it
            # pins the entry for the life of the process, or raises when the matrices are no longer resident)
            hit[2] = device_inverse(hit[0], device)
        elif hit[2].is_cuda:
            hit[2].record_stream(torch.cuda.current_stream(hit[2].device))
    return hit[2], hit[1].copy()


def composite_bev_img(bg, fg, fg_mask, H_world2bev, H_img2world_fix, K, RT, x_size, y_size, bw_mode=False, device="cuda"):
    """Returns (compo, H_world2img_cam) like the reference (compo.py:26-49): compo is (y_size, x_size, C) uint8."""
    as_numpy = not any(isinstance(x, torch.Tensor) for x in (bg, fg, fg_mask))
    bg, fg, fg_mask = (_hwc(_as_cuda(x, device)) for x in (bg, fg, fg_mask))
    if fg.shape != fg_mask.shape or bg.shape[2] != fg.shape[2]:
        raise ValueError("fg and fg_mask must have one shape, and bg their channel count")
    if bw_mode and fg.shape[2] != 3:
        raise ValueError("bw_mode needs a 3-channel BGR foreground (cv2.COLOR_BGR2GRAY), got shape %s" % (tuple(fg.shape),))
    minv, H_world2img_cam = _composite_maps(H_world2bev, H_img2world_fix, K, RT, bg.device)
    C = bg.shape[2]
    out = torch.empty((int(y_size), int(x_size), C), dtype=torch.uint8, device=bg.device)
    stream = torch.cuda.current_stream(bg.device).cuda_stream
    with torch.cuda.device(bg.device):
        st = _lib.load().bevwarp_warp_composite(
            bg.data_ptr(), bg.shape[0], bg.shape[1], bg.stride(0), fg.data_ptr(), fg_mask.data_ptr(), fg.shape[0], fg.shape[1], fg.stride(0),
            fg_mask.stride(0), out.data_ptr(), out.shape[0], out.shape[1], out.stride(0), C, minv.data_ptr(), minv.data_ptr() + 72,
            int(bool(bw_mode)), ctypes.c_void_p(stream))
    _lib.check(st)
    return _result(out, as_numpy), H_world2img_cam
