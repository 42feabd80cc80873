"""Parity of the HIP path (through the C ABI of include/bevwarp.h) with the CPU oracle.
Bars (BASELINE.json north_star): uint8 / nearest pixel-for-pixel; float32 bilinear within 1e-5 absolute
(inputs in [0,1)) -- the kernel keeps the oracle's operation order, so float results are asserted
bit-identical as well.  Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as co
from tests import workloads as wl

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5


@pytest.fixture(scope="module")
def W():
    from bev_amd import warp
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return warp


def run_gpu(W, src_np, M, dsize, interp, **kw):
    t = torch.from_numpy(np.ascontiguousarray(src_np)).cuda()
    poison = torch.full((dsize[1], dsize[0]) + tuple(t.shape[2:]), 77, dtype=t.dtype, device="cuda")  # unwritten pixels must not pass as zeros
    if "out" not in kw and t.dim() <= 3:
        kw = dict(kw, out=poison)
    out = W.warp_perspective(t, M, dsize, flags=interp, **kw)
    torch.cuda.synchronize()
    return out.reshape(poison.shape).cpu().numpy() if out is poison else out.cpu().numpy()


def check(got, exp):
    if exp.dtype == np.float32:
        assert np.abs(got - exp).max() <= F32_TOL
    np.testing.assert_array_equal(got, exp)


SHAPES = [
    ("brno", 1280, 720, 512, 512),      # BASELINE configs[0]
    ("brno", 1920, 1080, 320, 640),     # a real BrnoCompSpeed BEV size
    ("keystone", 1920, 1080, 1024, 1024),  # one frame of configs[1]
    ("keystone", 640, 360, 256, 192),
    ("brno", 192, 108, 37, 53),         # ragged destination
    ("keystone", 100, 60, 300, 9),      # height < 16: evaluation blocks are 113 wide
    ("brno", 64, 36, 70, 1),
    ("brno", 853, 481, 200, 120),       # source row stride not a multiple of 4: the two tap rows of a window differ in alignment
]


@pytest.mark.parametrize("kind,sw,sh,dw,dh", SHAPES)
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("interp", [0, 1])
def test_single_frame_parity(W, kind, sw, sh, dw, dh, dtype, interp):
    if kind == "brno":
        M = wl.synth_brno_H(sw, sh, dw, dh) if (sw * 9 == sh * 16) else wl.synth_brno_H(1920, 1080, dw, dh) @ np.diag([1920 / sw, 1080 / sh, 1.0])
    else:
        M = wl.keystone_H(sw, sh, dw, dh)
    src = wl.frame(0, sh, sw, dtype)
    check(run_gpu(W, src, M, (dw, dh), interp), co.warp_perspective(src, M, (dw, dh), interp))


@pytest.mark.parametrize("c", [1, 2, 4])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("interp", [0, 1])
def test_channel_counts(W, c, dtype, interp):
    M = wl.synth_brno_H(640, 360, 160, 200)
    src = wl.frame(3, 360, 640, dtype, c)
    check(run_gpu(W, src, M, (160, 200), interp), co.warp_perspective(src, M, (160, 200), interp))
    g = src[:, :, 0]
    got = run_gpu(W, g, M, (160, 200), interp)
    assert got.shape == (200, 160)
    check(got, co.warp_perspective(g, M, (160, 200), interp))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_batch_per_frame_and_shared_matrix(W, dtype):
    B, sw, sh, dw, dh = 5, 640, 360, 192, 160
    base = wl.keystone_H(sw, sh, dw, dh)
    Ms = np.stack([wl.jitter_H(base, i) for i in range(B)])
    frames = np.stack([wl.frame(i, sh, sw, dtype) for i in range(B)])
    t = torch.from_numpy(frames).cuda()
    got = W.warp_perspective(t, Ms, (dw, dh)).cpu().numpy()
    for i in range(B):
        check(got[i], co.warp_perspective(frames[i], Ms[i], (dw, dh)))
    got = W.warp_perspective(t, base, (dw, dh)).cpu().numpy()
    for i in range(B):
        check(got[i], co.warp_perspective(frames[i], base, (dw, dh)))
    with pytest.raises(ValueError):
        W.warp_perspective(t, Ms[:3], (dw, dh))
    # WARP_INVERSE_MAP and preallocated output
    out = torch.empty((B, dh, dw, 3), dtype=t.dtype, device="cuda")
    r = W.warp_perspective(t, co.invert3x3(base), (dw, dh), flags=W.INTER_LINEAR | W.WARP_INVERSE_MAP, out=out)
    assert r is out
    check(out.cpu().numpy()[2], co.warp_perspective(frames[2], base, (dw, dh)))


def test_row_padding_and_frame_views(W):
    """Strided (row-padded) sources and destinations; a crop view is warped without a copy."""
    sw, sh, dw, dh = 600, 300, 128, 96
    M = wl.synth_brno_H(1920, 1080, dw, dh) @ np.diag([1920 / sw, 1080 / sh, 1.0])
    big = torch.from_numpy(wl.frame(7, sh, sw + 40, np.uint8)).cuda()
    view = big[:, 8:8 + sw]  # rows keep the pitch of the parent: not 16-byte aligned
    assert not view.is_contiguous()
    got = W.warp_perspective(view, M, (dw, dh)).cpu().numpy()
    check(got, co.warp_perspective(np.ascontiguousarray(view.cpu().numpy()), M, (dw, dh)))
    view4 = big[:, 16:16 + 576]
    got = W.warp_perspective(view4, M, (dw, dh)).cpu().numpy()
    check(got, co.warp_perspective(np.ascontiguousarray(view4.cpu().numpy()), M, (dw, dh)))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_border_value_identity_translation(W, dtype):
    src = wl.frame(1, 90, 150, dtype)
    eye = np.eye(3)
    np.testing.assert_array_equal(run_gpu(W, src, eye, (100, 60), 1), src[:60, :100])
    M = np.array([[1, 0, 5.0], [0, 1, -3.0], [0, 0, 1]])
    for interp in (0, 1):
        check(run_gpu(W, src, M, (170, 100), interp, border_value=[7, 200.4, 300][:3]),
              co.warp_perspective(src, M, (170, 100), interp, border_value=[7, 200.4, 300]))
    M = np.array([[1, 0, 0.5], [0, 1, 0.25], [0, 0, 1.0]])
    check(run_gpu(W, src, M, (160, 100), 1, border_value=50), co.warp_perspective(src, M, (160, 100), 1, border_value=50))
    # rotation by 90 degrees and a flip: exact permutations
    h, w = src.shape[:2]
    R = np.array([[0, -1, h - 1.0], [1, 0, 0], [0, 0, 1]])
    np.testing.assert_array_equal(run_gpu(W, src, R, (h, w), 1), np.rot90(src, -1))
    Fm = np.array([[-1, 0, w - 1.0], [0, 1, 0], [0, 0, 1]])
    np.testing.assert_array_equal(run_gpu(W, src, Fm, (w, h), 0), src[:, ::-1])


def test_degenerate_homographies(W):
    src = wl.frame(2, 64, 96, np.uint8)
    # singular forward matrix -> inverse all zeros -> W == 0 everywhere -> every pixel samples (0, 0)
    check(run_gpu(W, src, np.ones((3, 3)), (40, 30), 1), co.warp_perspective(src, np.ones((3, 3)), (40, 30), 1))
    # horizon inside the destination: W changes sign across the tile
    M = np.array([[1.0, 0.2, 3.0], [0.1, 1.0, 2.0], [0.0, 0.02, -0.5]])
    for interp in (0, 1):
        check(run_gpu(W, src, M, (128, 96), interp), co.warp_perspective(src, M, (128, 96), interp))
    # far outside: nothing in bounds
    T = np.array([[1, 0, 1e6], [0, 1, 0], [0, 0, 1.0]])
    assert not run_gpu(W, src, T, (64, 64), 1).any()
    with pytest.raises(ValueError):
        W.warp_perspective(torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda"), np.full((3, 3), np.nan), (8, 8))
    with pytest.raises(ValueError):
        W.warp_perspective(torch.zeros((8, 8, 3), dtype=torch.int16, device="cuda"), np.eye(3), (8, 8))
    with pytest.raises(ValueError):
        W.warp_perspective(torch.zeros((8, 8, 3), dtype=torch.uint8), np.eye(3), (8, 8))


def test_magnification_and_minification_extremes(W):
    """x8 zoom (tiny source region per tile) and /6 shrink (every tap pair in a cache line of its own)."""
    src = wl.frame(4, 720, 1280, np.uint8)
    Z = np.array([[8.0, 0, -300.0], [0, 8.0, -200.0], [0, 0, 1]])
    check(run_gpu(W, src, Z, (512, 256), 1), co.warp_perspective(src, Z, (512, 256), 1))
    S = np.array([[1 / 6.0, 0.01, 1.0], [0.0, 1 / 6.0, 2.0], [0, 1e-5, 1]])
    for dtype in (np.uint8, np.float32):
        s = wl.frame(4, 720, 1280, dtype)
        check(run_gpu(W, s, S, (256, 128), 1), co.warp_perspective(s, S, (256, 128), 1))


def test_numpy_cv2_call_shape(W):
    """bev.warp.warpPerspective(img, H, (u, v)) is the drop-in for the call at vis_homo.py:89."""
    import bev.warp as bw
    img = wl.frame(5, 720, 1280, np.uint8)
    M = wl.synth_brno_H(1280, 720, 512, 512)
    out = bw.warpPerspective(img, M, (512, 512))
    assert out.dtype == np.uint8 and out.shape == (512, 512, 3)
    check(out, co.warp_perspective(img, M, (512, 512)))
    out = bw.warpPerspective(img, M, (512, 512), flags=bw.INTER_NEAREST, borderValue=(1, 2, 3))
    check(out, co.warp_perspective(img, M, (512, 512), 0, border_value=[1, 2, 3]))


def test_footprint_matches_oracle(W):
    M = wl.keystone_H(1920, 1080, 1024, 1024)
    counts, touched = W.footprint((1080, 1920), np.stack([M, wl.jitter_H(M, 1)]), (1024, 1024))
    n0, t0 = co.footprint((1080, 1920), M, (1024, 1024))
    assert int(counts[0]) == n0
    np.testing.assert_array_equal(touched[0].cpu().numpy(), t0)
    n1, _ = co.footprint((1080, 1920), wl.jitter_H(M, 1), (1024, 1024))
    assert int(counts[1]) == n1
    c, _ = W.footprint((360, 640), wl.synth_brno_H(640, 360, 128, 128), (128, 128), flags=W.INTER_NEAREST)
    assert int(c[0]) == co.footprint((360, 640), wl.synth_brno_H(640, 360, 128, 128), (128, 128), 0)[0]


# ---- BASELINE.json full sizes: oracle on sampled frames + size-independent properties -----------------
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_config2_batch32_1080p_to_1024(W, dtype):
    B, sw, sh, dw, dh = 32, 1920, 1080, 1024, 1024
    base = wl.keystone_H(sw, sh, dw, dh)
    Ms = np.stack([wl.jitter_H(base, i) for i in range(B)])
    frames = torch.empty((B, sh, sw, 3), dtype=torch.uint8 if dtype == np.uint8 else torch.float32, device="cuda")
    host = {}
    for i in range(B):
        f = wl.frame(i, sh, sw, dtype)
        if i in (0, 13, 31):
            host[i] = f
        frames[i] = torch.from_numpy(f).cuda()
    out = W.warp_perspective(frames, Ms, (dw, dh))
    for i, f in host.items():
        check(out[i].cpu().numpy(), co.warp_perspective(f, Ms[i], (dw, dh), nthreads=8))
    # property: same frames, same matrix -> identical outputs, regardless of batch position
    frames[5] = frames[0]
    Ms2 = Ms.copy()
    Ms2[5] = Ms[0]
    out2 = W.warp_perspective(frames, Ms2, (dw, dh))
    assert torch.equal(out2[5], out2[0]) and torch.equal(out2[0], out[0])
    # property: identity homography reproduces the top-left crop for every frame
    crop = W.warp_perspective(frames, np.eye(3), (dw, dh))
    assert torch.equal(crop, frames[:, :dh, :dw])
    if dtype == np.float32:
        # property: bilinear sampling is linear in the image
        a, b = frames[:4], frames[4:8]
        lhs = W.warp_perspective(0.25 * a + 0.5 * b, Ms[:4], (dw, dh))
        rhs = 0.25 * W.warp_perspective(a, Ms[:4], (dw, dh)) + 0.5 * W.warp_perspective(b, Ms[:4], (dw, dh))
        assert (lhs - rhs).abs().max().item() < 1e-6


@pytest.mark.parametrize("dtype,interp,kind", [(np.uint8, 0, "keystone"), (np.uint8, 0, "brno"), (np.uint8, 1, "brno"), (np.uint8, 1, "rot25"),
                                               (np.uint8, 0, "rot25"), (np.float32, 0, "keystone"), (np.float32, 1, "brno")])
def test_config2_size_other_paths(W, dtype, interp, kind):
    """configs[1]'s launch size (32 x 1080p -> 1024^2: 24-row tiles, full-height tiles as straight-line code, the tail split) on
    the paths test_config2_batch32_1080p_to_1024 does not take: nearest neighbour (every pass issued up front), the Brno-style and a
    25-degree footprint (patches, edge-cut and outside tiles).  Oracle on three frames of the batch."""
    B, sw, sh, dw, dh = 32, 1920, 1080, 1024, 1024
    base = {"keystone": wl.keystone_H, "brno": wl.synth_brno_H}[kind](sw, sh, dw, dh) if kind != "rot25" else wl.rotated_H(sw, sh, dw, dh, 25.0, 0.6)
    Ms = np.stack([wl.jitter_H(base, i) for i in range(B)])
    host = {i: wl.frame(60 + i, sh, sw, dtype) for i in (0, 17, 31)}
    frames = torch.empty((B, sh, sw, 3), dtype=torch.uint8 if dtype == np.uint8 else torch.float32, device="cuda")
    f0 = torch.from_numpy(host[0]).cuda()
    for i in range(B):
        frames[i] = torch.from_numpy(host[i]).cuda() if i in host else f0.flip(i % 2)
    out = W.warp_perspective(frames, Ms, (dw, dh), flags=interp)
    for i, f in host.items():
        check(out[i].cpu().numpy(), co.warp_perspective(f, Ms[i], (dw, dh), interp, nthreads=8))
    # a frame's result does not depend on its place in the batch (frames 1.. are flips of frame 0: redo one of them alone)
    j = 6
    alone = W.warp_perspective(frames[j:j + 1], Ms[j:j + 1], (dw, dh), flags=interp)
    assert torch.equal(alone[0], out[j])


def test_config4_single_4k_frame(W):
    """One frame of configs[3] (3840x2160 -> 2048x2048, uint8); the 8-GPU sharding is covered in test_shard.py."""
    sw, sh, dw, dh = 3840, 2160, 2048, 2048
    M = wl.keystone_H(sw, sh, dw, dh)
    f = wl.frame(0, sh, sw, np.uint8)
    check(run_gpu(W, f, M, (dw, dh), 1), co.warp_perspective(f, M, (dw, dh), nthreads=8))
    Mb = wl.synth_brno_H(sw, sh, dw, dh)
    check(run_gpu(W, f, Mb, (dw, dh), 1), co.warp_perspective(f, Mb, (dw, dh), nthreads=8))


# ---- every tile shape of the one kernel: ragged last tiles (destination width not a multiple of 256 / 128), fewer than
# ---- 16 rows (evaluation blocks wider than 64 px, lanes straddling them), destinations whose layout rules out the wide
# ---- stores, tiny frames -- all through the row-classified path, all bit-exact with the oracle.
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("interp", [0, 1])
def test_every_tile_shape_matches_oracle(W, dtype, interp):
    cases = [("keystone", 1920, 1080, 1024, 1024), ("brno", 1280, 720, 512, 512), ("brno", 192, 108, 37, 53),
             ("keystone", 856, 480, 300, 200), ("keystone", 100, 60, 300, 9), ("keystone", 640, 360, 1000, 3),
             ("brno", 640, 360, 333, 7), ("keystone", 640, 360, 257, 40), ("brno", 640, 360, 129, 33)]
    for kind, sw, sh, dw, dh in cases:
        M = (wl.synth_brno_H if kind == "brno" else wl.keystone_H)(sw, sh, dw, dh)
        src = wl.frame(2, sh, sw, dtype)
        check(run_gpu(W, src, M, (dw, dh), interp), co.warp_perspective(src, M, (dw, dh), interp))
    # 4-channel and 1-channel pixels, row-padded source view, border value
    M = wl.synth_brno_H(640, 360, 160, 200)
    for c in (1, 4):
        src = wl.frame(3, 360, 640, dtype, c)
        check(run_gpu(W, src, M, (160, 200), interp, border_value=9), co.warp_perspective(src, M, (160, 200), interp, border_value=9))
    big = torch.from_numpy(wl.frame(7, 300, 640, dtype)).cuda()
    view = big[:, 16:16 + 576]
    Mv = wl.keystone_H(576, 300, 128, 96)
    got = W.warp_perspective(view, Mv, (128, 96), flags=interp).cpu().numpy()
    check(got, co.warp_perspective(np.ascontiguousarray(view.cpu().numpy()), Mv, (128, 96), interp))
    # odd destination widths: rows start at byte offsets that rule out the wide stores
    for c in (1, 2, 3):
        src = wl.frame(5, 200, 320, dtype, c)
        Mo = wl.synth_brno_H(320, 200, 301, 41)
        got = W.warp_perspective(torch.from_numpy(src).cuda(), Mo, (301, 41), flags=interp).cpu().numpy()
        check(got, co.warp_perspective(src, Mo, (301, 41), interp))


def test_config4_full_per_gpu_shard_properties(W):
    """configs[3] at the size one GPU holds: 32 x (3840x2160 -> 2048x2048) uint8, per-frame homographies.  Oracle on three
    frames; size-independent properties on all of them (a frame's result does not depend on its batch position, the identity
    homography reproduces the crop, frames of one launch do not leak into each other)."""
    sw, sh, dw, dh, B = 3840, 2160, 2048, 2048, 32
    base = wl.keystone_H(sw, sh, dw, dh)
    Ms = np.stack([wl.jitter_H(base, i) for i in range(B)])
    uniq = [torch.from_numpy(wl.frame(80 + i, sh, sw, np.uint8)).cuda() for i in range(4)]
    frames = torch.empty((B, sh, sw, 3), dtype=torch.uint8, device="cuda")
    for i in range(B):
        frames[i] = uniq[i % 4] if i < 4 else uniq[i % 4].flip(i % 2)
    out = W.warp_perspective(frames, Ms, (dw, dh))
    for i in (0, 3, 31):
        exp = co.warp_perspective(frames[i].cpu().numpy(), Ms[i], (dw, dh), nthreads=8)
        assert np.array_equal(out[i].cpu().numpy(), exp), "frame %d" % i
    # batch-position invariance: frame 5 and its homography moved to slot 29 (and the reverse) give the same pixels
    perm = list(range(B))
    perm[5], perm[29] = 29, 5
    out_p = W.warp_perspective(frames[perm], Ms[perm], (dw, dh))
    assert torch.equal(out_p[29], out[5]) and torch.equal(out_p[5], out[29]) and torch.equal(out_p[11], out[11])
    # a shard is the same as its frames one by one (no leakage between frames of a launch)
    for i in (7, 30):
        assert torch.equal(W.warp_perspective(frames[i], Ms[i], (dw, dh)), out[i])
    # identity homography: the top-left crop of every frame, nearest and bilinear
    for flags in (0, 1):
        crop = W.warp_perspective(frames, np.eye(3), (dw, dh), flags=flags)
        assert torch.equal(crop, frames[:, :dh, :dw])
    del frames, out, out_p, crop
    torch.cuda.empty_cache()


def test_side_by_side_rois_of_one_image(W):
    """warping the left-half ROI of an image into its right-half ROI (equal row strides, disjoint byte columns) is a
    call cv2.warpPerspective accepts; the overlap guard must not refuse it -- and must still refuse ROIs that share a column,
    with a message that names the overlap."""
    h, w = 240, 512
    for dtype in (np.uint8, np.float32):
        img = wl.frame(11, h, w, dtype)
        big = torch.from_numpy(img).cuda()
        left, right = big[:, :w // 2], big[:, w // 2:]
        M = wl.keystone_H(w // 2, h, w // 2, h)
        exp = co.warp_perspective(np.ascontiguousarray(img[:, :w // 2]), M, (w // 2, h))
        out = W.warp_perspective(left, M, (w // 2, h), out=right)
        torch.cuda.synchronize()
        assert out is right
        check(big[:, w // 2:].cpu().numpy(), exp)
        np.testing.assert_array_equal(big[:, :w // 2].cpu().numpy(), img[:, :w // 2])  # the source half is untouched
        with pytest.raises(ValueError, match="overlap"):
            W.warp_perspective(left, M, (w // 2, h), out=big[:, w // 2 - 1:w - 1])
        with pytest.raises(ValueError, match="overlap"):
            W.warp_perspective(big, np.eye(3), (w, h), out=big)


def test_validated_launch_cache_hits_and_misses(W):
    """the plan cache of warp_perspective (bev_amd/warp.py::_plans).  A second identical call takes the cached
    launch and equals the oracle; a matrix tensor of another dtype or on another device -- the allocator can hand a recycled
    address to either -- must still run into the slow path's validation; the cache clears itself at _PLANS_MAX."""
    sw, sh, dw, dh = 320, 180, 128, 96
    M = wl.synth_brno_H(sw, sh, dw, dh)
    f0, f1 = wl.frame(1, sh, sw, np.uint8), wl.frame(2, sh, sw, np.uint8)
    src, out = torch.from_numpy(f0).cuda(), torch.empty((dh, dw, 3), dtype=torch.uint8, device="cuda")
    minv = W.device_inverse(M, src.device)
    W._plans.clear()
    W.warp_perspective(src, None, (dw, dh), out=out, M_inv_device=minv)
    assert len(W._plans) == 1
    src.copy_(torch.from_numpy(f1).cuda())
    W.warp_perspective(src, None, (dw, dh), out=out, M_inv_device=minv)  # the hit: same buffers, new pixels
    torch.cuda.synchronize()
    assert len(W._plans) == 1
    check(out.cpu().numpy(), co.warp_perspective(f1, M, (dw, dh)))
    # same shape and strides, other dtype: 36-byte matrices must not be read as 72-byte ones
    with pytest.raises(ValueError):
        W.warp_perspective(src, None, (dw, dh), out=out, M_inv_device=minv.to(torch.float32))
    with pytest.raises(ValueError):
        W.warp_perspective(src, None, (dw, dh), out=out, M_inv_device=minv.cpu())
    # a key that differs only in the matrix tensor's dtype is a different key even at the same address (simulated: the real
    # thing needs the allocator to recycle the block)
    (key, plan), = list(W._plans.items())
    assert torch.float64 in key and src.device in key
    # the cache empties itself rather than grow without bound
    for i in range(W._PLANS_MAX + 3):
        W._plans[("filler", i)] = plan
    W.warp_perspective(src, None, (dw, dh), out=torch.empty_like(out), M_inv_device=minv)
    assert len(W._plans) <= 4
    W._plans.clear()


@pytest.mark.parametrize("shape,dsize", [((1080, 1920, 3), (852, 480)), ((720, 1280, 3), (640, 360)), ((37, 53, 1), (100, 64)), ((64, 48, 4), (31, 17)),
                                         ((5, 7, 3), (1, 1)), ((1, 1, 3), (9, 4)), ((480, 852, 2), (1920, 1080)), ((33, 65), (17, 9)), ((40, 64, 3), (32, 20))])
def test_resize_matches_the_oracle(shape, dsize):
    """bevwarp_resize == oracle/resize_oracle.c (cv2.resize INTER_LINEAR uint8 as called at vis_homo.py:90; parity unpinned) bit for bit:
    the reference's own 1080p -> 852 x 480, magnification, 1-4 channels, degenerate sizes, the exact 2 x 2 box-mean case."""
    from bev_amd.resize import cv2_resize, resize
    c = shape[2] if len(shape) == 3 else 1
    img = wl.frame(21, shape[0], shape[1], np.uint8, c)
    if len(shape) == 2:
        img = img[:, :, 0]
    exp = co.resize_linear_u8(img, dsize)
    got = resize(torch.from_numpy(img).cuda(), dsize)
    torch.cuda.synchronize()
    assert got.shape == exp.shape
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    np.testing.assert_array_equal(cv2_resize(img, dsize), exp)  # numpy in, numpy out: cv2's call shape


def test_resize_batches_views_and_errors():
    from bev_amd.resize import cv2_resize, resize
    frames = np.stack([wl.frame(30 + i, 90, 160, np.uint8) for i in range(5)])
    t = torch.from_numpy(frames).cuda()
    got = resize(t, (71, 40)).cpu().numpy()
    for i in range(5):
        np.testing.assert_array_equal(got[i], co.resize_linear_u8(frames[i], (71, 40)))
    big = torch.from_numpy(wl.frame(7, 90, 200, np.uint8)).cuda()
    view = big[:, 8:168]  # row-padded view, no copy
    out = torch.full((40, 71, 3), 77, dtype=torch.uint8, device="cuda")
    assert resize(view, (71, 40), out=out) is out
    np.testing.assert_array_equal(out.cpu().numpy(), co.resize_linear_u8(np.ascontiguousarray(view.cpu().numpy()), (71, 40)))
    np.testing.assert_array_equal(cv2_resize(frames[0], None, fx=0.5, fy=0.5), co.resize_linear_u8(frames[0], (80, 45)))  # exact 2 x 2: the box mean
    with pytest.raises(ValueError):
        resize(t.float(), (8, 8))
    with pytest.raises(ValueError):
        resize(t, (8, 8), interpolation=0)
    with pytest.raises(ValueError):
        resize(t, (0, 8))
    with pytest.raises(ValueError, match="overlap"):
        resize(big[:, :100], (100, 90), out=big[:, 50:150])


def test_two_step_small_branch_pixel_for_pixel():
    """vis_homo.py:73-78,90-91 as the reference runs it: resize to 852 x 480, then warp with the scaled calibration's homography; both
    steps on the device, compared with the oracle's two steps."""
    from bev_amd.resize import resize
    from bev_amd import warp
    img = wl.frame(0, 1080, 1920, np.uint8)
    M_small = wl.synth_brno_H(852, 480, 320, 640)
    t = torch.from_numpy(img).cuda()
    got = warp.warp_perspective(resize(t, (852, 480)), M_small, (320, 640)).cpu().numpy()
    exp = co.warp_perspective(co.resize_linear_u8(img, (852, 480)), M_small, (320, 640), 1)
    np.testing.assert_array_equal(got, exp)


def test_resize_random_shapes():
    """40 seeded random (source, destination, channels) triples -- strong minification, magnification, one-pixel sides, odd sizes --
    through bevwarp_resize against the oracle, bit for bit."""
    from bev_amd.resize import resize
    rng = np.random.default_rng(2024)
    for _ in range(40):
        sh, sw = int(rng.integers(1, 200)), int(rng.integers(1, 300))
        dh, dw = int(rng.integers(1, 260)), int(rng.integers(1, 400))
        c = int(rng.integers(1, 5))
        img = rng.integers(0, 256, (sh, sw, c), dtype=np.uint8)
        got = resize(torch.from_numpy(img).cuda(), (dw, dh)).cpu().numpy()
        np.testing.assert_array_equal(got, co.resize_linear_u8(img, (dw, dh)), err_msg="%dx%dx%d -> %dx%d" % (sw, sh, c, dw, dh))


@pytest.mark.gpu
def test_resize_four_pixels_per_lane_kernel_corners():
    """The batch form of bevwarp_resize (four destination pixels per lane, 8-byte tap windows, dword stores): every channel count, right edges of
    0-3 pixels beyond the last full lane, one-tap columns at the end of a row (strong magnification keeps many of them), source rows of
    exactly 8 bytes, a row-padded destination whose rows still start 4-byte aligned (fast kernel) and one whose rows do not (one-pixel
    kernel), batches -- bit for bit against the oracle."""
    from bev_amd.resize import resize
    rng = np.random.default_rng(99)
    cases = []
    for c in (1, 2, 3, 4):
        for dw in (4 * 37, 4 * 37 + 4 // np.gcd(4, c)):  # rows of dw * c bytes, multiples of 4: the fast kernel, full and ragged last lanes
            cases.append((61, max(8 // c, 3) + int(rng.integers(0, 90)), c, 47, dw))
        cases.append((33, (8 + c - 1) // c, c, 20, 64))      # a source row of 8 bytes (or the first count above it): magnification, one-tap columns
        cases.append((50, 97, c, 31, 400))                    # 4 x magnification
    for sh, sw, c, dh, dw in cases:
        img = rng.integers(0, 256, (3, sh, sw, c), dtype=np.uint8)
        got = resize(torch.from_numpy(img).cuda(), (dw, dh)).cpu().numpy()
        for i in range(3):
            np.testing.assert_array_equal(got[i], co.resize_linear_u8(img[i], (dw, dh)).reshape(dh, dw, c), err_msg="%dx%dx%d -> %dx%d" % (sw, sh, c, dw, dh))
    img = rng.integers(0, 256, (120, 213, 3), dtype=np.uint8)
    exp = co.resize_linear_u8(img, (100, 56))
    src_t = torch.from_numpy(img).cuda()
    for pad in (4, 3):  # padded destination rows: 100 * 3 + 12 bytes apart (aligned: fast kernel), 100 * 3 + 9 (not: one-pixel kernel)
        hold = torch.full((56, 100 + pad, 3), 9, dtype=torch.uint8, device="cuda")
        view = hold[:, :100]
        from bev_amd import _lib
        import ctypes
        st = _lib.load().bevwarp_resize(src_t.data_ptr(), view.data_ptr(), 1, 120, 213, 56, 100, 3, 120 * 213 * 3, 213 * 3, view.stride(0) * 56, view.stride(0),
                                        _lib.U8, _lib.INTER_LINEAR, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == 0
        torch.cuda.synchronize()
        np.testing.assert_array_equal(view.cpu().numpy(), exp)
        assert (hold[:, 100:].cpu().numpy() == 9).all()  # the padding is not written


def test_tile_verdict_tables_reproduce_the_plain_call():
    """bevwarp_warp_classes (ABI v7): a table filled once for given matrices and geometry, then read by later launches instead of
    classifying every tile again.  The filled launch writes no pixel; launches that use the table equal bevwarp_warp bit for bit --
    keystone (pair tiles, edge-cut tiles), Brno-style (turned, outside tiles), 8-bit and float, both interpolations; entries that
    were never filled fall back to classification; and the Python entry picks tables up by itself for matrices it owns."""
    import ctypes
    from bev_amd import _lib, warp
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for (B, sw, sh, dw, dh) in ((32, 640, 360, 512, 768), (3, 700, 420, 300, 77)):  # tall-tile launch with a split tail / a small one
        for kind in ("keystone", "brno"):
            base = (wl.keystone_H if kind == "keystone" else wl.synth_brno_H)(sw, sh, dw, dh)
            Ms = np.stack([wl.jitter_H(base, g) for g in range(B)])
            minv = warp.device_inverse(Ms, dev)
            for tdt, ndt, dt, esz in ((torch.uint8, np.uint8, 0, 1), (torch.float32, np.float32, 1, 4)):
                src = torch.stack([torch.from_numpy(wl.frame(70 + g % 3, sh, sw, ndt)) for g in range(B)]).to(dev)
                for interp in (1, 0):
                    n = lib.bevwarp_tile_classes_bytes(B, sh, sw, dh, dw, 3, dt, interp)
                    assert n > 0 and n % 12 == 0
                    table = torch.zeros(n // 4, dtype=torch.int32, device=dev)
                    ref = torch.full((B, dh, dw, 3), 7, dtype=tdt, device=dev)
                    got = torch.full((B, dh, dw, 3), 9, dtype=tdt, device=dev)
                    args = lambda d: (src.data_ptr(), d.data_ptr(), B, sh, sw, dh, dw, 3, src.stride(0) * esz, src.stride(1) * esz, d.stride(0) * esz,
                                      d.stride(1) * esz, minv.data_ptr(), B, dt, interp, None)
                    assert lib.bevwarp_warp(*args(ref), ctypes.c_void_p(stream)) == 0
                    assert lib.bevwarp_warp_classes(*args(got), table.data_ptr(), 1, ctypes.c_void_p(stream)) == 0  # fill: no pixel written
                    torch.cuda.synchronize()
                    assert bool((got == 9).all())
                    t = table.cpu().numpy().view(np.uint32)
                    assert ((t >> 31) == 1).sum() >= B * ((dw + 255) // 256)  # every launched workgroup left its verdict
                    assert lib.bevwarp_warp_classes(*args(got), table.data_ptr(), 0, ctypes.c_void_p(stream)) == 0
                    torch.cuda.synchronize()
                    assert torch.equal(got, ref), (kind, dt, interp)
                    # half the entries wiped: those tiles are classified as usual
                    table[::2] = 0
                    got.fill_(9)
                    assert lib.bevwarp_warp_classes(*args(got), table.data_ptr(), 0, ctypes.c_void_p(stream)) == 0
                    torch.cuda.synchronize()
                    assert torch.equal(got, ref)
    # the Python entry: matrices owned by device_inverse get a table on their first launch and use it from then on
    warp._class_tables.clear()
    sw, sh, dw, dh, B = 640, 360, 512, 96, 2
    Ms = np.stack([wl.jitter_H(wl.keystone_H(sw, sh, dw, dh), g) for g in range(B)])
    frames = np.stack([wl.frame(80 + g, sh, sw, np.uint8) for g in range(B)])
    t = torch.from_numpy(frames).to(dev)
    first = warp.warp_perspective(t, Ms, (dw, dh)).cpu().numpy()
    assert len(warp._class_tables) == 1
    second = warp.warp_perspective(t, Ms, (dw, dh)).cpu().numpy()
    assert len(warp._class_tables) == 1
    for g in range(B):
        exp = co.warp_perspective(frames[g], Ms[g], (dw, dh), 1)
        np.testing.assert_array_equal(first[g], exp)
        np.testing.assert_array_equal(second[g], exp)
    mine = warp.device_inverse(Ms, dev).clone()  # a tensor the caller owns: no table is kept for it
    warp.warp_perspective(t, None, (dw, dh), M_inv_device=mine)
    assert len(warp._class_tables) == 1
