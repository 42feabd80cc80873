"""torch twin of a subset of bev_amd.rbox (mirrors /root/reference/bev/rbox_torch.py:24-168).

Device-agnostic: every tensor is created on the input's device, so these run on `cuda` tensors
next to the HIP warp.  Differences from the numpy versions that the reference also has:
`rbox_world_bev` measures the similarity's scale from ROW norms (rbox_torch.py:159-160) and does
not special-case empty input.  Deviation: `xywhr2xyxy(external_aa=True)` returns an n x 4 tensor
(the reference stacks `(values, indices)` tuples from `.min(dim=1)`, rbox_torch.py:94-98).
"""
import torch

_MODES = ("bev", "world")


def v2yaw(x, mode):
    assert mode in _MODES
    a, b = (x[:, 0], x[:, 1]) if mode == "bev" else (x[:, 1], x[:, 0])
    return torch.arctan2(a, b)


def yaw2v(x, mode):
    assert mode in _MODES
    s, c = torch.sin(x), torch.cos(x)
    return torch.stack((s, c) if mode == "bev" else (c, s), dim=1)


def yaw2mat(x, mode):
    assert mode in _MODES
    x = x.reshape(-1, 1)
    s, c = torch.sin(x), torch.cos(x)
    cols = [c, s, -s, c] if mode == "bev" else [c, -s, s, c]
    return torch.cat(cols, dim=1).reshape(-1, 2, 2)


_SIGNS = {"bev": (-1, -1, -1, 1, 1, 1, 1, -1.0), "world": (-1, -1, 1, -1, 1, 1, -1, 1.0)}


def xywhr2xyxy(x, mode, external_aa=False):
    assert mode in _MODES
    half_x, half_y = (x[:, 2], x[:, 3]) if mode == "bev" else (x[:, 3], x[:, 2])
    sg = torch.tensor(_SIGNS[mode], dtype=x.dtype, device=x.device)
    local = torch.zeros((x.shape[0], 8), dtype=x.dtype, device=x.device)
    local[:, 0::2] = sg[0::2] * half_x[:, None] / 2
    local[:, 1::2] = sg[1::2] * half_y[:, None] / 2
    pts = torch.matmul(yaw2mat(x[:, 4], mode), local.reshape(-1, 4, 2).transpose(1, 2))
    y = pts.transpose(1, 2).reshape(-1, 8)
    y += x[:, [0, 1, 0, 1, 0, 1, 0, 1]]
    if not external_aa:
        return y
    xs, ys = y[:, 0::2], y[:, 1::2]
    return torch.stack([xs.min(dim=1).values, ys.min(dim=1).values, xs.max(dim=1).values, ys.max(dim=1).values], dim=1)


def xywhr2xyvec(xywhr, mode):
    assert mode in _MODES
    vs = yaw2v(xywhr[:, 4], mode) * xywhr[:, 3:4]
    xs, ys = xywhr[:, 0], xywhr[:, 1]
    return torch.stack([xs, ys, xs + vs[:, 0], ys + vs[:, 1]], dim=1)


def xy82xyvec(xy8):
    vs = xy8[:, 2:4] - xy8[:, :2]
    xs = 0.5 * (xy8[:, 0] + xy8[:, 4])
    ys = 0.5 * (xy8[:, 1] + xy8[:, 5])
    return torch.stack([xs, ys, xs + vs[:, 0], ys + vs[:, 1]], dim=1)


def rbox_world_bev(rbox_src, H, src):
    """n x 5 rboxes between the BEV and the world through a similarity H (rbox_torch.py:123-168)."""
    assert src in _MODES
    target = "world" if src == "bev" else "bev"
    H = H / H[2, 2]
    assert torch.abs(H[2, 0]) + torch.abs(H[2, 1]) < 1e-5

    r_src = rbox_src[:, 4]
    zeros, ones = torch.zeros_like(r_src)[..., None], torch.ones_like(r_src)[..., None]
    v_tgt = H.mm(torch.cat([yaw2v(r_src, src), zeros], dim=1).T).T[:, :2]
    r_tgt = v2yaw(v_tgt, target)
    xy_tgt = H.mm(torch.cat([rbox_src[:, :2], ones], dim=1).T).T[:, :2]

    scale = torch.sqrt(H[0, 0] ** 2 + H[0, 1] ** 2)
    scale_1 = torch.sqrt(H[1, 0] ** 2 + H[1, 1] ** 2)
    assert torch.abs(scale - scale_1) < 1e-5
    return torch.cat([xy_tgt, rbox_src[:, 2:4] * scale, r_tgt[..., None]], dim=1)
