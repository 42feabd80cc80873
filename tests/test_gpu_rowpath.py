"""Parity of warp_rows' tile paths (interior tiles in row segments / patches, edge-cut tiles in blocks / patches, outside
tiles) with the CPU oracle: every block class (FAST / OUT / EDGE / SLOW), every channel count, border values, rounding
ties and their redo passes, W sign changes, degenerate sources, batches, planar stores, turned footprints.
Bit-exact for every dtype (the float kernel keeps the oracle's operation order)."""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as co
from tests import workloads as wl

pytestmark = pytest.mark.gpu


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


def both(W, src, M, dsize, interp, **kw):
    np.testing.assert_array_equal(run_gpu(W, src, M, dsize, interp, **kw), co.warp_perspective(src, M, dsize, interp, **kw))


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("interp", [0, 1])
def test_channels_all_row_classes(W, c, dtype, interp):
    """Brno-style BEV: a third of the rows FAST, a third OUT, the frame's edge crossing the rest."""
    M = wl.synth_brno_H(640, 360, 512, 80)
    src = wl.frame(5, 360, 640, dtype, c)
    both(W, src, M, (512, 80), interp)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("interp", [0, 1])
def test_border_values(W, dtype, interp):
    """Non-zero border: pixels whose four taps are all outside ARE the border value (float bilinear too: the reference's
    "fully outside" path stores cval directly), so OUT rows / tiles are filled for every format; straddling pixels substitute
    the border tap by tap."""
    M = wl.synth_brno_H(640, 360, 512, 64)
    src = wl.frame(6, 360, 640, dtype)
    bv = [7, 200, 33] if dtype == np.uint8 else [0.3, 0.55, 0.7]
    both(W, src, M, (512, 64), interp, border_value=bv)
    both(W, src, M, (512, 64), interp, border_value=bv[0])


def test_float_border_far_outside_is_exact(W):
    """f32, border 0.3, a sub-pixel shift far outside the frame: every pixel == np.float32(0.3) (outside TILES, OUT blocks of
    edge tiles and the guarded sampler all store the value itself), and a frame the border cuts through matches the oracle."""
    rng = np.random.default_rng(3)
    src = rng.random((90, 110, 3), dtype=np.float32)
    bv = [0.3, 0.7, 1.0 / 3.0]
    M = np.array([[1, 0, -400.0 - 13 / 32], [0, 1, -250.0 - 7 / 32], [0, 0, 1.0]])
    out = run_gpu(W, src, M, (300, 70), 1, border_value=bv)
    for k in range(3):
        assert (out[..., k] == np.float32(bv[k])).all()
    for shift in ((-60.0 - 13 / 32, -20.0 - 7 / 32), (95.0 + 5 / 32, 70.0 + 9 / 32), (-100.0 - 1 / 32, 0.25)):
        M = np.array([[1, 0, shift[0]], [0, 1, shift[1]], [0, 0, 1.0]])
        both(W, src, M, (300, 70), 1, border_value=bv)
        both(W, src[:, :, :1].copy(), M, (300, 70), 1, border_value=bv[0])
    # perspective: W changes sign inside the destination (SLOW rows run the guarded sampler for every pixel)
    Mp = np.array([[1.0, 0.02, -20.0], [0.01, 1.0, -8.0], [0.004, 0.0005, -0.5]])
    both(W, src, Mp, (300, 70), 1, border_value=bv)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_rounding_ties_on_fast_rows(W, dtype):
    """Identity, integer and k/64-px shifts, x2 zoom: every coordinate sits on (or next to) a rounding boundary."""
    src = wl.frame(7, 200, 700, dtype)
    for M in (np.eye(3), np.array([[1, 0, -17.0], [0, 1, -9.0], [0, 0, 1]]), np.array([[1, 0, -3.515625], [0, 1, -2.984375], [0, 0, 1]]),
              np.array([[2.0, 0, -40.0], [0, 2.0, -30.0], [0, 0, 1]]), np.array([[0.5, 0, -1.0], [0, 0.5, -1.0], [0, 0, 1]])):
        for interp in (0, 1):
            both(W, src, M, (512, 48), interp)


@pytest.mark.parametrize("interp", [0, 1])
def test_w_sign_change_and_far_coordinates(W, interp):
    """Horizon inside the destination (W changes sign along rows and between rows), coordinates beyond 2^26 px
    next to it, and a destination that sees nothing of the frame."""
    src = wl.frame(8, 120, 160, np.uint8)
    M = np.array([[1.0, 0.2, 3.0], [0.1, 1.0, 2.0], [0.004, 0.02, -1.5]])
    both(W, src, M, (512, 96), interp)
    M2 = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [1.0 / 300.0, 0.0, -1.0]])  # W = 0 along a column
    both(W, src, M2, (512, 32), interp)
    T = np.array([[1, 0, 1e7], [0, 1, 0], [0, 0, 1.0]])
    assert not run_gpu(W, src, T, (512, 32), interp).any()
    both(W, src.astype(np.float32) / 255, M, (256, 64), interp)


@pytest.mark.parametrize("sw,sh", [(1, 1), (2, 2), (3, 2), (5, 1), (2, 40)])
def test_tiny_sources(W, sw, sh):
    """Sources smaller than the unguarded load window: no row may take the FAST class."""
    for dtype in (np.uint8, np.float32):
        src = wl.frame(9, sh, sw, dtype)
        M = np.array([[200.0 / max(sw, 2), 3.0, 20.0], [1.0, 12.0 / max(sh, 2), 4.0], [0, 0, 1.0]])
        for interp in (0, 1):
            both(W, src, M, (256, 32), interp)


def test_padded_rows_batch_and_per_frame_matrices(W):
    """Row-padded source views, several frames, one matrix each, ragged last tile row (dst_h not a multiple of 16)."""
    B, sw, sh, dw, dh = 3, 600, 300, 512, 41
    base = wl.synth_brno_H(1920, 1080, dw, dh) @ np.diag([1920 / sw, 1080 / sh, 1.0])
    Ms = np.stack([wl.jitter_H(base, i, px=7.0) for i in range(B)])
    big = torch.from_numpy(np.stack([wl.frame(10 + i, sh, sw + 24, np.uint8) for i in range(B)])).cuda()
    view = big[:, :, 12:12 + sw]
    got = W.warp_perspective(view, Ms, (dw, dh)).cpu().numpy()
    for i in range(B):
        np.testing.assert_array_equal(got[i], co.warp_perspective(np.ascontiguousarray(view[i].cpu().numpy()), Ms[i], (dw, dh)))


def test_fused_resize_warp_small_branch(W):
    """SURVEY.md 8(f1), vis_homo.py:73-78,90-91: the resize folded into the homography.  Bit-identical to the oracle
    warp through M_small @ S; close to the two-step path (resize, then warp) on a smooth frame."""
    import bev
    from bev_amd.homo import compose_H_bev_img
    calib = bev.Calib(vp1=np.array([1200.0, -300.0]), vp2=np.array([-2500.0, -150.0]), pp=np.array([959.5, 539.5]), height=8, u_size=1920, v_size=1080)
    center = calib.gen_center_in_world()
    bspec = bev.BEVWorldSpec(u_size=512, v_size=256, u_axis="y", v_axis="-x", x_size=64, y_size=64, x_min=center[0] - 20, y_min=center[1] - 32)
    small = calib.scale(align_corners=False, new_u=852, new_v=480)
    M_small = compose_H_bev_img(bspec.gen_H_world_bev(), small.gen_H_world_img())
    yy, xx = np.mgrid[0:1080, 0:1920]
    img = np.stack([128 + 100 * np.sin(xx / 97.0) * np.cos(yy / 71.0), xx * 255.0 / 1919, yy * 255.0 / 1079], axis=2).astype(np.uint8)
    t = torch.from_numpy(img).cuda()
    got = W.warp_perspective_resized(t, M_small, (512, 256), (852, 480)).cpu().numpy()
    S = W.resize_matrix((1920, 1080), (852, 480))
    np.testing.assert_array_equal(got, co.warp_perspective(img, M_small @ S, (512, 256), 1))
    # two-step path: bilinear resize with half-pixel centres (what cv2.resize computes, up to its fixed-point rounding)
    sm = torch.nn.functional.interpolate(t.permute(2, 0, 1)[None].float(), size=(480, 852), mode="bilinear", align_corners=False)
    sm = sm[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).contiguous()
    two = W.warp_perspective(sm, M_small, (512, 256)).cpu().numpy()
    inside = (two.sum(axis=2) > 0) & (got.sum(axis=2) > 0)
    assert np.abs(got.astype(int) - two.astype(int))[inside].mean() < 1.5


@pytest.mark.parametrize("interp", [0, 1])
@pytest.mark.parametrize("c,dw,dh", [(3, 512, 80), (3, 300, 37), (1, 256, 32), (4, 512, 16), (3, 70, 5)])
def test_planar_normalised_output(W, interp, c, dw, dh):
    """uint8 warp written as normalised float32 channel planes in one pass == the uint8 oracle warp,
    converted with float32 multiply-then-add (row-path tiles, ragged tiles and the short-image path)."""
    M = wl.synth_brno_H(640, 360, dw, dh)
    src = wl.frame(11, 360, 640, np.uint8, c)
    mean, std = np.array([0.485, 0.456, 0.406, 0.5])[:c], np.array([0.229, 0.224, 0.225, 0.25])[:c]
    scale, bias = 1.0 / (255.0 * std), -mean / std
    t = torch.from_numpy(src).cuda()
    got = W.warp_to_planar(t, M, (dw, dh), scale=scale, bias=bias, flags=interp, border_value=[9, 60, 200, 17][:c])
    assert got.shape == (c, dh, dw) and got.dtype == torch.float32
    ref = co.warp_perspective(src, M, (dw, dh), interp, border_value=[9, 60, 200, 17][:c]).reshape(dh, dw, c)
    exp = ref.transpose(2, 0, 1).astype(np.float32) * scale.astype(np.float32)[:, None, None] + bias.astype(np.float32)[:, None, None]
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    # batched, default scale 1/255, preallocated output
    frames = torch.from_numpy(np.stack([wl.frame(12 + i, 360, 640, np.uint8, c) for i in range(3)])).cuda()
    out = torch.empty((3, c, dh, dw), dtype=torch.float32, device="cuda")
    r = W.warp_to_planar(frames, M, (dw, dh), flags=interp, out=out)
    assert r is out
    for i in range(3):
        ref = co.warp_perspective(frames[i].cpu().numpy(), M, (dw, dh), interp).reshape(dh, dw, c)
        np.testing.assert_array_equal(out[i].cpu().numpy(), ref.transpose(2, 0, 1).astype(np.float32) * np.float32(1.0 / 255.0) + np.float32(0.0))


def test_graphed_step_replays_bit_exactly(W):
    """tools.graphed_step.GraphedStep: warp + composite + tracker geometry captured once, replayed on new inputs."""
    from bev_amd.compo import composite_reg_img
    from tools.graphed_step import GraphedStep
    M = wl.synth_brno_H(640, 360, 512, 64)
    src = torch.zeros((360, 640, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((64, 512, 3), dtype=torch.uint8, device="cuda")
    minv = W.device_inverse(M, src.device)

    def step():
        W.warp_perspective(src, None, (512, 64), out=out, M_inv_device=minv)
        return composite_reg_img(out, out.flip(0), out.flip(1))

    g = GraphedStep(step)
    for i in range(3):
        f = wl.frame(20 + i, 360, 640, np.uint8)
        src.copy_(torch.from_numpy(f).cuda())
        blended = g.replay()
        torch.cuda.synchronize()
        exp = co.warp_perspective(f, M, (512, 64), 1)
        np.testing.assert_array_equal(out.cpu().numpy(), exp)
        m = exp[:, ::-1].astype(float) / 255
        ref = (exp[::-1].astype(float) * m + exp.astype(float) * (1 - m)).round()
        np.testing.assert_array_equal(blended.cpu().numpy(), np.minimum(ref, 255).astype(np.uint8))


@pytest.mark.parametrize("sw", [637, 638, 639, 640])
def test_row_strides_of_every_alignment(W, sw):
    """8-bit RGB bilinear fetches aligned 12-byte windows: source row strides of every residue mod 4 (the two tap
    rows of a pixel share their alignment only when the stride is a multiple of 4), and an odd frame base."""
    M = wl.synth_brno_H(1920, 1080, 512, 48) @ np.diag([1920 / sw, 1080 / 359, 1.0])
    src = wl.frame(30, 359, sw, np.uint8)
    both(W, src, M, (512, 48), 1)
    buf = torch.zeros(359 * sw * 3 + 8, dtype=torch.uint8, device="cuda")
    for shift in (1, 2, 3):
        view = buf[shift:shift + 359 * sw * 3].view(359, sw, 3)
        view.copy_(torch.from_numpy(src))
        got = W.warp_perspective(view, M, (512, 48)).cpu().numpy()
        np.testing.assert_array_equal(got, co.warp_perspective(src, M, (512, 48), 1))


# ---- launches of hundreds of tiles (per-frame matrices, taller tiles from 2 x the resident workgroups up): batches of small
# ---- frames reach them at sizes the oracle finishes in seconds.
def _batch_case(W, dtype_c, sw, sh, dw, dh, B, kind, interp, pad_to=None, border=None):
    c = dtype_c
    frames = np.stack([wl.frame(20 + i, sh, sw, np.uint8, c) for i in range(B)])
    base = (wl.keystone_H if kind == "keystone" else wl.synth_brno_H)(sw, sh, dw, dh) if kind != "identity" else np.eye(3)
    Ms = np.stack([wl.jitter_H(base, i, px=3.0) for i in range(B)])
    t = torch.from_numpy(frames).cuda()
    if pad_to:  # a row-padded view: row stride pad_to * c bytes
        big = torch.zeros((B, sh, pad_to, c), dtype=torch.uint8, device="cuda")
        big[:, :, :sw] = t
        t = big[:, :, :sw]
    got = W.warp_perspective(t, Ms, (dw, dh), flags=interp, border_value=border).cpu().numpy()
    for i in range(B):
        exp = co.warp_perspective(frames[i], Ms[i], (dw, dh), interp, border_value=0 if border is None else border)
        np.testing.assert_array_equal(got[i], exp, err_msg="frame %d" % i)


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_batched_tiles_all_channel_counts(W, c):
    _batch_case(W, c, 640, 360, 512, 256, 8, "keystone", 1)            # 8 x 2 x 16 = 256 tiles, all inside the frame


def test_batched_tiles_mixed_footprints(W):
    _batch_case(W, 3, 640, 360, 512, 256, 8, "brno", 1)                 # turned footprint: patches, edge-cut and outside tiles
    _batch_case(W, 3, 640, 368, 768, 200, 8, "keystone", 1, border=7)  # ragged last group (200 = 12 x 16 + 8), 3 tiles wide
    _batch_case(W, 3, 1280, 720, 300, 330, 16, "keystone", 1)           # > 1.9 x magnification
    _batch_case(W, 3, 320, 200, 1024, 64, 16, "keystone", 1)            # strong minification


def test_batched_tiles_row_padded_source_and_identity(W):
    _batch_case(W, 3, 636, 360, 512, 256, 8, "keystone", 1, pad_to=640)  # row stride 1920 B, rows 16-byte aligned, width not
    _batch_case(W, 3, 640, 360, 512, 256, 8, "identity", 1)              # integer coordinates: every pixel on a tie boundary
    _batch_case(W, 4, 640, 360, 512, 256, 8, "identity", 1)


def test_batched_tiles_planar_output(W):
    B, sw, sh, dw, dh = 8, 640, 360, 512, 256
    frames = np.stack([wl.frame(40 + i, sh, sw, np.uint8) for i in range(B)])
    Ms = np.stack([wl.jitter_H(wl.keystone_H(sw, sh, dw, dh), i) for i in range(B)])
    scale, bias = [1 / 255.0, 0.5, 2.0], [0.0, -1.0, 3.5]
    got = W.warp_to_planar(torch.from_numpy(frames).cuda(), Ms, (dw, dh), scale=scale, bias=bias).cpu().numpy()
    for i in range(B):
        u8 = co.warp_perspective(frames[i], Ms[i], (dw, dh), 1)
        exp = np.stack([u8[:, :, k].astype(np.float32) * np.float32(scale[k]) + np.float32(bias[k]) for k in range(3)])
        np.testing.assert_array_equal(got[i], exp)


@pytest.mark.parametrize("kind,c,dw,dh", [("keystone", 3, 512, 80), ("brno", 3, 300, 37), ("brno", 1, 256, 32), ("keystone", 4, 130, 16), ("brno", 2, 70, 5)])
def test_planar_output_of_float_sources(W, kind, c, dw, dh):
    """bevwarp_warp_planar for float32 frames: planes of float(warp) * scale + bias, every tile path (interior rows, interior
    blocks, edge blocks), ragged tiles."""
    sw, sh = 640, 360
    M = (wl.keystone_H if kind == "keystone" else wl.synth_brno_H)(sw, sh, dw, dh)
    src = wl.frame(31, sh, sw, np.float32, c)
    scale, bias = np.linspace(0.5, 2.0, c), np.linspace(-1.0, 1.0, c)
    for interp in (0, 1):
        got = W.warp_to_planar(torch.from_numpy(src).cuda(), M, (dw, dh), scale=scale, bias=bias, flags=interp).cpu().numpy()
        ref = co.warp_perspective(src, M, (dw, dh), interp).reshape(dh, dw, c)
        exp = ref.transpose(2, 0, 1) * scale.astype(np.float32)[:, None, None] + bias.astype(np.float32)[:, None, None]
        np.testing.assert_array_equal(got, exp.astype(np.float32))


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("deg,zoom,dw,dh", [(30.0, 2.4, 300, 77), (90.0, 0.8, 256, 48), (-12.0, 1.6, 515, 40), (45.0, 1.2, 256, 24)])
def test_turned_footprints_take_the_patch_layout(W, c, dtype, deg, zoom, dw, dh):
    """Rotated / minified footprints: interior and edge-cut tiles in the patch lane layout (16 x 4 / 32 x 2 pixels per gather
    instruction), ragged tile widths and heights, both interpolations, interleaved and planar stores."""
    sw, sh = 640, 360
    M = wl.rotated_H(sw, sh, dw, dh, deg, zoom)  # the 515 x 40 / 300 x 77 windows reach past the frame: edge and outside tiles too
    src = wl.frame(17, sh, sw, dtype, c)
    for interp in (0, 1):
        both(W, src, M, (dw, dh), interp)
    scale, bias = np.linspace(0.5, 2.0, c), np.linspace(-1.0, 1.0, c)
    got = W.warp_to_planar(torch.from_numpy(src).cuda(), M, (dw, dh), scale=scale, bias=bias).cpu().numpy()
    ref = co.warp_perspective(src, M, (dw, dh), 1).reshape(dh, dw, c).astype(np.float32)
    np.testing.assert_array_equal(got, (ref.transpose(2, 0, 1) * scale.astype(np.float32)[:, None, None] + bias.astype(np.float32)[:, None, None]).astype(np.float32))


@pytest.mark.parametrize("c,dtype", [(3, np.uint8), (1, np.uint8), (3, np.float32), (2, np.float32)])
def test_turned_footprints_through_unaligned_views(W, c, dtype):
    """Patch layout with a source view whose frames / rows are not 4-byte aligned and a destination view that admits no wide
    store (odd base and row stride): the element-store path of every lane."""
    sw, sh, dw, dh = 637, 355, 301, 45
    M = wl.rotated_H(sw, sh, dw, dh, 33.0, 1.7)  # reaches past the frame on two sides
    src = wl.frame(23, sh, sw, dtype, c)
    src_big = torch.zeros((sh, sw + 3, c), dtype=torch.from_numpy(src).dtype, device="cuda")
    src_big[:, 1:1 + sw] = torch.from_numpy(src).cuda()
    for interp in (0, 1):
        out_big = torch.full((dh, dw + 5, c), 77, dtype=src_big.dtype, device="cuda")
        W.warp_perspective(src_big[:, 1:1 + sw], M, (dw, dh), flags=interp, out=out_big[:, 3:3 + dw])
        torch.cuda.synchronize()
        got = out_big.cpu().numpy()
        np.testing.assert_array_equal(got[:, 3:3 + dw], co.warp_perspective(src, M, (dw, dh), interp).reshape(dh, dw, c))
        assert (got[:, :3] == 77).all() and (got[:, 3 + dw:] == 77).all()  # nothing written outside the view


def _random_homography(rng, sw, sh, dw, dh):
    """src -> dst map of a random similarity-plus-perspective window: rotation, 0.4 .. 3 x scale, mild keystone, a shift that
    may push part of the window out of the frame."""
    ang = rng.uniform(-np.pi, np.pi) if rng.random() < 0.7 else rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2])
    zoom = float(np.exp(rng.uniform(np.log(0.4), np.log(3.0))))
    c, s = np.cos(ang) * zoom, np.sin(ang) * zoom
    A = np.array([[c, -s, 0.0], [s, c, 0.0], [rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-4, 2e-4), 1.0]])
    T0 = np.array([[1, 0, -(dw - 1) / 2.0], [0, 1, -(dh - 1) / 2.0], [0, 0, 1.0]])
    T1 = np.array([[1, 0, (sw - 1) / 2.0 + rng.uniform(-0.4, 0.4) * sw], [0, 1, (sh - 1) / 2.0 + rng.uniform(-0.4, 0.4) * sh], [0, 0, 1.0]])
    if rng.random() < 0.25:  # integer-valued similarity: every coordinate on a tie boundary
        A = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]) * [[1], [1], [1]]
        T1 = np.round(T1)
        T0 = np.round(T0)
    return np.linalg.inv(T1 @ A @ T0)


@pytest.mark.parametrize("seed", range(6))
def test_random_homographies_all_formats(W, seed):
    """Seeded sweep: random windows (turned, scaled, keystoned, partly outside the frame, tie-heavy), random sizes, every
    format and both interpolations, single frames and small batches with per-frame matrices."""
    rng = np.random.default_rng(1000 + seed)
    for case in range(10):
        sw, sh = int(rng.integers(40, 700)), int(rng.integers(30, 400))
        dw, dh = int(rng.integers(1, 600)), int(rng.integers(1, 90))
        c = int(rng.integers(1, 5))
        dtype = np.uint8 if rng.random() < 0.6 else np.float32
        interp = int(rng.integers(0, 2))
        B = int(rng.choice([1, 1, 3]))
        Ms = np.stack([_random_homography(rng, sw, sh, dw, dh) for _ in range(B)])
        frames = np.stack([wl.frame(100 * seed + case + i, sh, sw, dtype, c) for i in range(B)])
        border = None if rng.random() < 0.5 else [float(rng.integers(0, 200))] * c
        got = W.warp_perspective(torch.from_numpy(frames).cuda(), Ms, (dw, dh), flags=interp, border_value=border).cpu().numpy()
        for i in range(B):
            exp = co.warp_perspective(frames[i], Ms[i], (dw, dh), interp, border_value=0 if border is None else border)
            np.testing.assert_array_equal(got[i].reshape(exp.shape), exp, err_msg="seed %d case %d frame %d: %dx%d -> %dx%d c=%d %s interp=%d" % (
                seed, case, i, sw, sh, dw, dh, c, np.dtype(dtype).name, interp))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("interp", [0, 1])
def test_row_affine_tiles_and_their_tolerance(W, dtype, interp):
    """Interior tiles whose source row and divide do not depend on the destination column take one reciprocal and one Y
    coordinate per row segment (rows_tiles.inc, `tile_affine`).  Exact members of the class (M3 = M6 = 0: scale + shift,
    keystone), members up to floating-point noise, and matrices that cross the tolerance from below and from above -- all must
    equal the oracle; ties on X, on the uniform Y, and on both."""
    sw, sh, dw, dh = 700, 420, 512, 96
    src = wl.frame(31, sh, sw, dtype)

    def both_inv(M):  # cv2.WARP_INVERSE_MAP: M is the dst -> src map itself, so its entries are exactly what the kernel sees
        np.testing.assert_array_equal(run_gpu(W, src, M, (dw, dh), interp | 16), co.warp_perspective(src, M, (dw, dh), interp, m_is_inverse=True))

    base = np.array([[1.25, 0.0, 20.0], [0.0, 1.5, 30.0], [0.0, 0.004, 1.0]])  # keystone form: Y / W and W depend on y only
    both_inv(base)
    both_inv(np.array([[1.0, 0.0, 7.0], [0.0, 1.0, 5.0], [0.0, 0.0, 1.0]]))          # integer shift: ties everywhere
    both_inv(np.array([[2.0, 0.0, 3.5], [0.0, 1.0, 9.0 + 1 / 64], [0.0, 0.0, 1.0]]))   # X and (uniform) Y on rounding ties
    both_inv(np.array([[1.0, 0.0, 1 / 64], [0.0, 3.0, 1.0 / 3.0], [0.0, 0.0, 2.0]]))
    for eps in (1e-18, 1e-15, 1e-14, 3e-14, 1e-13, 1e-11, 1e-8, 1e-5):  # the tolerance sits near 3e-14 for these sizes
        for (i, j) in ((2, 0), (1, 0)):
            M = base.copy()
            M[i, j] = eps if i == 2 else eps * 300.0
            both_inv(M)
            M[i, j] = -M[i, j]
            both_inv(M)
    # a least-squares keystone (the benchmark's matrix) and its jittered frames, batched
    Ms = np.stack([wl.jitter_H(wl.keystone_H(sw, sh, dw, dh), g) for g in range(4)])
    frames = np.stack([wl.frame(32 + g, sh, sw, dtype) for g in range(4)])
    got = W.warp_perspective(torch.from_numpy(frames).cuda(), Ms, (dw, dh), flags=interp).cpu().numpy()
    for g in range(4):
        np.testing.assert_array_equal(got[g], co.warp_perspective(frames[g], Ms[g], (dw, dh), interp))


@pytest.mark.parametrize("misalign", [0, 1, 2, 3])
def test_pair_tiles_scales_mirrors_and_alignments(W, misalign):
    """Row-affine interior tiles of 8-bit RGB whose left taps advance by 0 .. 2 source pixels per destination pixel fetch the taps of
    two adjacent pixels with one 16-byte load (rows_sample.inc, issue_p / finish_p).  Horizontal scales from strong magnification up
    to and across the 2 - 1/16 limit, mirrored maps (which must not take the path), a divide that changes the scale from the tile's
    top row to its bottom row, ties -- through source views whose frames start 0 .. 3 bytes off a 4-byte boundary (row stride a
    multiple of 4: the RS4 kernel) -- all must equal the oracle."""
    sw, sh, dw, dh = 1020, 300, 512, 96
    src = wl.frame(41 + misalign, sh, sw, np.uint8)
    big = torch.zeros((sh, sw + 4, 3), dtype=torch.uint8, device="cuda")  # row stride 3072: a multiple of 4
    lo = (misalign * 3) % 4  # the view starts `misalign` pixels in: 0 / 3 / 2 / 1 bytes off a 4-byte boundary
    big[:, misalign:misalign + sw] = torch.from_numpy(src).cuda()
    view = big[:, misalign:misalign + sw]
    assert view.data_ptr() % 4 == lo and view.stride(0) % 4 == 0

    def check(M):
        out = torch.full((dh, dw, 3), 77, dtype=torch.uint8, device="cuda")
        W.warp_perspective(view, M, (dw, dh), flags=1 | 16, out=out)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), co.warp_perspective(src, M, (dw, dh), 1, m_is_inverse=True))

    for s in (0.3, 0.75, 1.0, 1.5, 1.875, 1.93, 1.9375, 1.94, 1.97):
        check(np.array([[s, 0.0, 4.25], [0.0, 1.5, 30.0], [0.0, 0.0, 1.0]]))                       # pure scale: the same step on every row
        check(np.array([[s, 0.0, 4.25 + 1 / 64], [0.0, 1.5, 30.0 + 1 / 64], [0.0, 0.0007, 1.0]]))  # keystone: the step shrinks down the tile, ties
    check(np.array([[1.99, 0.0, 1.0], [0.0, 1.5, 30.0], [0.0, -0.0004, 1.0]]))                     # under the limit at the top, over it at the bottom
    check(np.array([[-1.5, 0.0, 1000.0], [0.0, 1.5, 30.0], [0.0, 0.0, 1.0]]))                      # mirrored: left taps run backwards
    check(np.array([[1.0, 0.0, 3.0], [0.0, 1.0, 5.0], [0.0, 0.0, 1.0]]))                           # integer shift: every pixel a tie
    check(np.array([[1.875, 0.0, 0.0], [0.0, 2.0, 8.0], [0.0, 0.0, 1.0]]))                         # reaches the frame's first column


def test_pair_tiles_full_height_batched(W):
    """The same path in the straight-line form of full-height tiles (24 rows: launches of at least two resident rounds), and in the
    half-height workgroups at the end of every XCD's run: 32 frames with per-frame keystones of different horizontal scales."""
    sw, sh, dw, dh, B = 1000, 360, 512, 768, 32
    rng = np.random.default_rng(5)
    frames = np.stack([wl.frame(60 + g % 3, sh, sw, np.uint8) for g in range(B)])
    Ms = []
    for g in range(B):
        s = rng.uniform(0.6, 1.93)
        Ms.append(np.array([[s, 0.0, rng.uniform(2, 6)], [0.0, 0.44, rng.uniform(3, 9)], [0.0, rng.uniform(-1e-4, 1e-4), 1.0]]))
    Ms = np.stack(Ms)
    got = W.warp_perspective(torch.from_numpy(frames).cuda(), Ms, (dw, dh), flags=1 | 16).cpu().numpy()
    for g in range(B):
        np.testing.assert_array_equal(got[g], co.warp_perspective(frames[g], Ms[g], (dw, dh), 1, m_is_inverse=True))
