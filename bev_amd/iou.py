"""Device rotated-box IoU: the tracker's iou_batch_rbox (reference bev/tracker/rbox_tracker.py:87-92,
which calls d3d.box.box2d_iou(.., method="rbox")) as one HIP launch over all N x M pairs."""
import ctypes

import numpy as np
import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64}


_PLANS_MAX = 256
_plans = {}  # validated launches of rbox_iou(a, b, out=...) by (addresses, shapes, strides, dtypes, devices)


def rbox_iou(a, b, out=None):
    """a: (N, >=5), b: (M, >=5) CUDA tensors of [x, y, w, h, yaw, ...] (world convention: length h along the
    heading).  Returns the (N, M) IoU matrix, same dtype (written into `out` when given: a contiguous (N, M) tensor)."""
    key = None
    if out is not None:  # steady state of a camera loop: the same buffers as a validated call -> one lookup and the bound C call
        try:
            key = (a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape, b.shape, out.shape, a.stride(), b.stride(), out.stride(), a.dtype, b.dtype, out.dtype,
                   a.device, b.device, out.device)
            plan = _plans.get(key)
        except (AttributeError, TypeError):
            key, plan = None, None
        if plan is not None:
            fn, args, dev_index = plan
            stream = torch._C._cuda_getCurrentRawStream(dev_index)
            if torch.cuda.current_device() == dev_index:
                st = fn(*args, stream)
            else:
                with torch.cuda.device(dev_index):
                    st = fn(*args, stream)
            if st:
                _lib.check(st)
            return out
    if not (isinstance(a, torch.Tensor) and a.is_cuda and isinstance(b, torch.Tensor) and b.is_cuda):
        raise ValueError("rbox_iou needs CUDA (HIP) tensors")
    if a.dtype != b.dtype or a.dtype not in _DTYPES or a.dim() != 2 or b.dim() != 2 or a.shape[1] < 5 or b.shape[1] < 5:
        raise ValueError("a, b must be (N, >=5) / (M, >=5) tensors of the same float dtype")
    if out is None:
        out = torch.empty((a.shape[0], b.shape[0]), dtype=a.dtype, device=a.device)
    elif (not isinstance(out, torch.Tensor) or tuple(out.shape) != (a.shape[0], b.shape[0]) or out.dtype != a.dtype or out.device != a.device
          or not out.is_contiguous()):
        raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (a.dtype, (a.shape[0], b.shape[0]), a.device))
    a_in, b_in = a, b
    a, b = a.contiguous(), b.contiguous()
    stream = torch.cuda.current_stream(a.device).cuda_stream
    fn = _lib.load().bevwarp_rbox_iou
    args = (a.data_ptr(), a.shape[0], a.shape[1], b.data_ptr(), b.shape[0], b.shape[1], out.data_ptr(), _DTYPES[a.dtype])
    with torch.cuda.device(a.device):
        st = fn(*args, ctypes.c_void_p(stream))
    _lib.check(st)
    if key is not None and a.data_ptr() == a_in.data_ptr() and b.data_ptr() == b_in.data_ptr():  # (no copy was made on the way)
        if len(_plans) >= _PLANS_MAX:
            _plans.clear()
        _plans[key] = (fn, args, a.device.index if a.device.index is not None else torch.cuda.current_device())
    return out


def iou_batch_rbox(bb_test, bb_gt, device="cuda"):
    """Drop-in for the tracker front-end (rbox_tracker.py:87-92): numpy (N, >=5) x (M, >=5) -> numpy (N, M).
    The reference adds pi/2 to both yaws before calling d3d; IoU is invariant to that common rotation
    under this package's rectangle convention, so it is not applied."""
    ta = torch.from_numpy(np.ascontiguousarray(bb_test[:, :5], dtype=np.float64)).to(device)
    tb = torch.from_numpy(np.ascontiguousarray(bb_gt[:, :5], dtype=np.float64)).to(device)
    return rbox_iou(ta, tb).cpu().numpy()


def iou_any(boxes1, boxes2, device="cuda"):
    """`d3d.box.box2d_iou(boxes1, boxes2, method="rbox")` call shape (rbox_tracker.py:92): numpy arrays or torch tensors
    (any device) of [x, y, w, h, yaw] rows in, the (N, M) IoU matrix of the same kind out."""
    if isinstance(boxes1, torch.Tensor):
        b2 = boxes2 if isinstance(boxes2, torch.Tensor) else torch.as_tensor(np.asarray(boxes2))
        dt = boxes1.dtype if boxes1.dtype in _DTYPES else torch.float64
        dev = boxes1.device if boxes1.is_cuda else torch.device(device)
        return rbox_iou(boxes1[:, :5].to(dev, dt), b2[:, :5].to(dev, dt)).to(boxes1.device)
    return iou_batch_rbox(np.asarray(boxes1), np.asarray(boxes2), device=device)
