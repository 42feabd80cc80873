"""Device resize: the HIP replacement for `img_small = cv2.resize(img, (new_u, new_v))` (reference call site vis_homo.py:90, the
"small" branch in front of the warp at :91).

    small = resize(frames, (new_u, new_v))                          # torch tensors on the GPU, batched
    small = cv2_resize(img, (new_u, new_v))                         # numpy in / numpy out, cv2 call shape

uint8 images, the default INTER_LINEAR: OpenCV's classic bilinear path (sampling at (d + 0.5) * scale - 0.5, 11-bit coefficients,
replicated edge, an exact 2 x 2 decimation = the box mean).  Bit-exact with oracle/resize_oracle.c, which restates that path from
memory: parity with an installed cv2 is UNPINNED (OpenCV is absent from this image, the reference holds no fixture).  The fast form
of the "small" branch needs no resize at all (bev_amd.warp.warp_perspective_resized folds it into the homography); this entry exists
for callers that want the reference's two-step pixels.  All pixel work happens in bev_amd/csrc through the C ABI; nothing here falls
back to the CPU.
"""
import ctypes

import numpy as np
import torch

from . import _lib

INTER_LINEAR = _lib.INTER_LINEAR


def resize(src, dsize, interpolation=INTER_LINEAR, out=None):
    """src (B, H, W, C) / (H, W, C) / (H, W) uint8 CUDA tensor, channels-last, rows contiguous; dsize = (width, height) as OpenCV.
    Returns a tensor shaped like src with (height, width) replaced.  Asynchronous on the current stream."""
    if not isinstance(src, torch.Tensor) or not src.is_cuda:
        raise ValueError("resize needs a CUDA (HIP) tensor; use cv2_resize for numpy images")
    if src.dtype != torch.uint8:
        raise ValueError("unsupported dtype %s (uint8)" % src.dtype)
    if int(interpolation) != INTER_LINEAR:
        raise ValueError("only INTER_LINEAR (cv2.resize's default, what the reference uses) is implemented")
    if src.dim() == 2:
        s4 = src[None, :, :, None]
    elif src.dim() == 3:
        s4 = src[None]
    elif src.dim() == 4:
        s4 = src
    else:
        raise ValueError("src must be (B,H,W,C), (H,W,C) or (H,W)")
    B, H, W, C = s4.shape
    if not 1 <= C <= 4:
        raise ValueError("1 to 4 channels, got %d" % C)
    if s4.stride(3) != 1 or s4.stride(2) != C:
        s4 = s4.contiguous()
    dw, dh = int(dsize[0]), int(dsize[1])
    if dw <= 0 or dh <= 0:
        raise ValueError("dsize must be positive, got %s" % (dsize,))
    if out is None:
        d4 = torch.empty((B, dh, dw, C), dtype=torch.uint8, device=s4.device)
    else:
        if not isinstance(out, torch.Tensor) or out.dtype != torch.uint8 or out.device != s4.device or out.numel() != B * dh * dw * C:
            raise ValueError("out must be a uint8 tensor of %d elements on %s" % (B * dh * dw * C, s4.device))
        d4 = out.reshape(B, dh, dw, C)
        if d4.data_ptr() != out.data_ptr() or d4.stride(3) != 1 or d4.stride(2) != C:
            raise ValueError("out must be a contiguous-row channels-last tensor")
    stream = torch.cuda.current_stream(s4.device).cuda_stream
    with torch.cuda.device(s4.device):
        st = _lib.load().bevwarp_resize(s4.data_ptr(), d4.data_ptr(), B, H, W, dh, dw, C, s4.stride(0), s4.stride(1), d4.stride(0), d4.stride(1),
                                        _lib.U8, INTER_LINEAR, ctypes.c_void_p(stream))
    _lib.check(st)
    if out is not None:
        return out
    if src.dim() == 2:
        return d4[0, :, :, 0]
    return d4[0] if src.dim() == 3 else d4


def cv2_resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR, device="cuda"):
    """cv2.resize call shape for numpy uint8 images: uploads, resizes on the GPU, downloads.  dsize (width, height); when it is None
    or (0, 0) the size comes from fx, fy as OpenCV computes it (round(fx * width), round(fy * height))."""
    img = np.asarray(src)
    if img.dtype != np.uint8:
        raise ValueError("unsupported dtype %s (uint8)" % img.dtype)
    if dsize is None or tuple(dsize) == (0, 0):
        if not (fx > 0 and fy > 0):
            raise ValueError("either dsize or both fx and fy must be given")
        dsize = (int(round(fx * img.shape[1])), int(round(fy * img.shape[0])))  # cv::saturate_cast<int>: round half to even, as Python's round
    res = resize(torch.from_numpy(np.ascontiguousarray(img)).to(device), dsize, interpolation).cpu().numpy()
    if dst is not None:
        dst[...] = res
        return dst
    return res
