"""Device point projection: pts_world_bev (reference bev/rbox.py:136-151) for large N on the GPU."""
import ctypes

import numpy as np
import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def project_points(pts, H, out=None):
    """pts: (N, 2) or (N, 3) float32/float64 CUDA tensor, contiguous.  H: 3x3 (numpy / tensor / list).
    N x 2 -> N x 2 (dehomogenised);  N x 3 -> N x 3 (homogeneous in, last column 1 out).
    One pass over memory, float64 arithmetic.  `out` may alias `pts`."""
    if not isinstance(pts, torch.Tensor) or not pts.is_cuda:
        raise ValueError("project_points needs a CUDA (HIP) tensor; bev.rbox.pts_world_bev is the host version")
    if pts.dtype not in _DTYPES or pts.dim() != 2 or pts.shape[1] not in (2, 3):
        raise ValueError("pts must be (N, 2|3) float32/float64, got %s %s" % (tuple(pts.shape), pts.dtype))
    pts = pts.contiguous()
    if out is None:
        out = torch.empty_like(pts)
    elif out.shape != pts.shape or out.dtype != pts.dtype or not out.is_contiguous():
        raise ValueError("out must match pts")
    Hh = np.ascontiguousarray(H.detach().cpu().numpy() if isinstance(H, torch.Tensor) else H, dtype=np.float64).reshape(3, 3)
    stream = torch.cuda.current_stream(pts.device).cuda_stream
    with torch.cuda.device(pts.device):
        st = _lib.load().bevwarp_project_points(pts.data_ptr(), out.data_ptr(), pts.shape[0], pts.shape[1],
                                                Hh.ctypes.data_as(ctypes.c_void_p), _DTYPES[pts.dtype], ctypes.c_void_p(stream))
    _lib.check(st)
    return out
