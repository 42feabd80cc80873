"""Device point projection and rotated IoU against the oracle / the reference's own vectors."""
import numpy as np
import pytest
import torch

from oracle import cpu_oracle as co

pytestmark = pytest.mark.gpu


def test_project_points_reference_vectors(golden):
    from bev_amd.points import project_points
    g = golden["rbox"]
    Hm = np.array(g["H_world_img"])
    for key, out in (("pts2", "pts_world_bev_2"), ("pts3", "pts_world_bev_3")):
        got = project_points(torch.tensor(g[key], dtype=torch.float64, device="cuda"), Hm).cpu().numpy()
        np.testing.assert_allclose(got, np.array(g[out]), rtol=1e-12, atol=0)  # outputs of the reference itself


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", [2, 3])
def test_project_points_config3(golden, dtype, dim):
    """configs[2]: 1e7 image points through H_world_img; oracle = C restatement of pts_world_bev."""
    from bev_amd.points import project_points
    Hm = np.array(golden["calib_from_vps"]["base"]["H_world_img"])
    n = 10_000_000 if dim == 2 else 1_000_003
    rng = np.random.default_rng(7)
    pts = rng.uniform(0, [1920, 1080], (n, 2))
    if dim == 3:
        pts = np.concatenate([pts, rng.uniform(0.5, 2, (n, 1))], axis=1)
    pts = pts.astype(dtype)
    t = torch.from_numpy(pts).cuda()
    got = project_points(t, Hm).cpu().numpy()
    exp = co.project_points(pts, Hm)
    if dtype == np.float64:
        np.testing.assert_allclose(got, exp, rtol=1e-13, atol=0)
    else:
        np.testing.assert_array_equal(got, exp)  # same f64 arithmetic, one final rounding to f32
    # in place, empty, and error behaviour
    r = project_points(t, Hm, out=t)
    assert r is t and np.array_equal(t.cpu().numpy(), got)
    assert project_points(torch.empty((0, dim), dtype=t.dtype, device="cuda"), Hm).shape == (0, dim)
    with pytest.raises(ValueError):
        project_points(torch.zeros((4, 4), device="cuda"), Hm)
    with pytest.raises(ValueError):
        project_points(torch.zeros((4, 2)), Hm)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rbox_iou_config5(dtype):
    """configs[4]'s IoU half: 512 detections x 512 tracks."""
    from bev_amd.iou import iou_batch_rbox, rbox_iou
    rng = np.random.default_rng(11)

    def boxes(n):
        return np.stack([rng.uniform(0, 100, n), rng.uniform(0, 100, n), rng.uniform(1.6, 2.2, n), rng.uniform(3.5, 6, n),
                         rng.uniform(-np.pi, np.pi, n)], axis=1)

    a, b = boxes(512), boxes(512)
    b[:128] = a[:128] + rng.normal(0, 0.3, (128, 5))  # overlapping pairs
    exp = co.rbox_iou(a, b)
    got = rbox_iou(torch.from_numpy(a.astype(dtype)).cuda(), torch.from_numpy(b.astype(dtype)).cuda()).cpu().numpy()
    assert got.shape == (512, 512) and (np.diag(exp)[:128] > 0.2).all()
    if dtype == np.float64:
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-12)
    else:
        np.testing.assert_allclose(got, co.rbox_iou(a.astype(np.float32), b.astype(np.float32)), rtol=0, atol=2e-6)
    # known answers
    sq = torch.tensor([[0, 0, 1, 1, 0.0], [0, 0, 1, 1, np.pi / 4], [5, 5, 1, 1, 0.3]], dtype=torch.float64, device="cuda")
    m = rbox_iou(sq, sq).cpu().numpy()
    inter = 2 * (np.sqrt(2) - 1)
    np.testing.assert_allclose(m, [[1, inter / (2 - inter), 0], [inter / (2 - inter), 1, 0], [0, 0, 1]], atol=1e-12)
    # tracker front-end shape: numpy in, numpy out, extra columns ignored, empty sets
    dets = np.concatenate([a[:7], np.ones((7, 1))], axis=1)
    np.testing.assert_allclose(iou_batch_rbox(dets, b[:9]), exp[:7, :9], atol=1e-12)
    assert rbox_iou(torch.zeros((0, 5), device="cuda"), torch.zeros((3, 5), device="cuda")).shape == (0, 3)


def test_composite_bev_img_matches_reference_arithmetic(golden):
    """bev/tool/compo.py:26-49 (three warps + alpha blend) on the GPU vs oracle warps + the reference's numpy blend."""
    from bev_amd.compo import composite_bev_img
    from bev_amd.homo import homo_from_KRt
    from tests import workloads as wl
    k = golden["homo"]["KRt_in"]
    K = np.array([[800.0, 0, 640.0], [0, 790.0, 360.0], [0, 0, 1.0]])
    RT = np.array(k["T"])
    rng = np.random.default_rng(3)
    bg, fg = wl.frame(0, 720, 1280, np.uint8), wl.frame(1, 720, 1280, np.uint8)
    mask = (rng.random((720, 1280, 1)) > 0.5).astype(np.uint8).repeat(3, 2) * 255
    mask[100:200] = 128
    H_world2bev = np.array([[0.0, 8.0, 160.0], [-8.0, 0.0, 500.0], [0.0, 0.0, 1.0]])
    H_img2world_fix = np.linalg.inv(homo_from_KRt(K, Rt_homo=RT)) @ np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1]])
    got, Hcam = composite_bev_img(torch.from_numpy(bg).cuda(), fg, mask, H_world2bev, H_img2world_fix, K, RT, 320, 640)
    got_np, _ = composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, K, RT, 320, 640)  # numpy in -> numpy out, like the reference
    assert isinstance(got_np, np.ndarray) and np.array_equal(got_np, got.cpu().numpy())
    np.testing.assert_allclose(Hcam, homo_from_KRt(K, Rt_homo=RT))
    bg_b = co.warp_perspective(bg, H_world2bev.dot(H_img2world_fix), (320, 640)).astype(np.float64)
    Hc = H_world2bev.dot(np.linalg.inv(Hcam))
    fg_b = co.warp_perspective(fg, Hc, (320, 640)).astype(np.float64)
    m_b = co.warp_perspective(mask, Hc, (320, 640)).astype(np.float64) / 255
    exp = (fg_b * m_b + bg_b * (1 - m_b)).round()
    exp[exp > 255] = 255
    np.testing.assert_array_equal(got.cpu().numpy(), exp.astype(np.uint8))
    assert got.shape == (640, 320, 3) and got.dtype == torch.uint8
    # the one-launch composite == three device warps + the blend kernel, bit for bit (also where the maps leave the frames)
    from bev_amd.compo import composite_reg_img
    from bev_amd.warp import warp_perspective
    cu = [torch.from_numpy(x).cuda() for x in (bg, fg, mask)]
    three = composite_reg_img(warp_perspective(cu[0], H_world2bev.dot(H_img2world_fix), (320, 640)), warp_perspective(cu[1], Hc, (320, 640)),
                              warp_perspective(cu[2], Hc, (320, 640)))
    assert torch.equal(three, got)
    # a mask / foreground of another size than the background, single-channel images, bw_mode
    fg2, m2 = wl.frame(4, 360, 640, np.uint8), wl.frame(5, 360, 640, np.uint8)
    S = np.diag([0.5, 0.5, 1.0])
    K2 = S @ K
    got2, Hcam2 = composite_bev_img(bg, fg2, m2, H_world2bev, H_img2world_fix, K2, RT, 333, 77)
    Hc2 = H_world2bev.dot(np.linalg.inv(Hcam2))
    exp2 = (co.warp_perspective(fg2, Hc2, (333, 77)).astype(np.float64) * (co.warp_perspective(m2, Hc2, (333, 77)).astype(np.float64) / 255) +
            co.warp_perspective(bg, H_world2bev.dot(H_img2world_fix), (333, 77)).astype(np.float64) *
            (1 - co.warp_perspective(m2, Hc2, (333, 77)).astype(np.float64) / 255)).round()
    np.testing.assert_array_equal(got2, np.minimum(exp2, 255).astype(np.uint8))
    g1, _ = composite_bev_img(bg[:, :, 0], fg[:, :, 1], mask[:, :, 0], H_world2bev, H_img2world_fix, K, RT, 320, 640)
    np.testing.assert_array_equal(g1[:, :, 0], exp.astype(np.uint8)[:, :, 0] * 0 + np.minimum((
        co.warp_perspective(fg[:, :, 1], Hc, (320, 640)).astype(np.float64) * (co.warp_perspective(mask[:, :, 0], Hc, (320, 640)).astype(np.float64) / 255) +
        co.warp_perspective(bg[:, :, 0], H_world2bev.dot(H_img2world_fix), (320, 640)).astype(np.float64) *
        (1 - co.warp_perspective(mask[:, :, 0], Hc, (320, 640)).astype(np.float64) / 255)).round(), 255).astype(np.uint8))


def _tracker_case(n, m, seed=21):
    import bev
    from bev_amd import rbox as host_rbox
    calib = bev.Calib(vp1=np.array([1200.0, -300.0]), vp2=np.array([-2500.0, -150.0]), pp=np.array([959.5, 539.5]), height=8, u_size=1920, v_size=1080)
    center = calib.gen_center_in_world()
    bspec = bev.BEVWorldSpec(u_size=1024, v_size=1024, u_axis="y", v_axis="-x", x_size=64, y_size=64, x_min=center[0] - 20, y_min=center[1] - 32)
    H_world_bev = bspec.gen_H_world_bev()
    H_img_world = np.linalg.inv(calib.gen_H_world_img())
    rng = np.random.default_rng(seed)
    dets_bev = np.column_stack([rng.uniform(0, 1024, (n, 2)), rng.uniform(24, 36, n), rng.uniform(56, 96, n), rng.uniform(-np.pi, np.pi, n)])
    dets_world_host = host_rbox.rbox_world_bev(dets_bev, H_world_bev, "bev") if n else np.zeros((0, 5))
    # trackers: the detections' own world boxes, jittered (so a band of pairs overlaps), plus strays; rows carry the
    # tracker's extra state columns (velocity, id) like the reference's `trackers` array
    k = min(n, (2 * m) // 3)
    near = dets_world_host[:k] + rng.normal(0, [0.4, 0.4, 0.05, 0.1, 0.05], (k, 5))
    far = np.column_stack([rng.uniform(-30, 60, (m - k, 2)), rng.uniform(1.6, 2.2, m - k), rng.uniform(3.5, 6, m - k), rng.uniform(-np.pi, np.pi, m - k)])
    trks = np.column_stack([np.vstack([near, far]), rng.normal(0, 1, (m, 2))])
    return dets_bev, trks, dets_world_host, H_world_bev, H_img_world


@pytest.mark.parametrize("n,m", [(512, 512), (300, 257), (1, 1), (7, 0), (0, 5)])
def test_tracker_step_one_launch_matches_host_functions(n, m):
    """SURVEY.md 8(f4) at configs[4]'s size: detections BEV -> world, IoU against predicted tracker boxes, the gate and the
    image centres from ONE launch == the host functions that the reference's own vectors pin (rbox_world_bev,
    pts_world_bev) and the IoU oracle."""
    from bev_amd import rbox as host_rbox
    from bev_amd.tracker_geom import rbox_world_bev_device, tracker_geometry_step
    dets_bev, trks, dets_world_host, H_world_bev, H_img_world = _tracker_case(n, m)
    out = tracker_geometry_step(dets_bev, trks, H_world_bev, iou_threshold=0.3, H_img_world=H_img_world)
    assert out["dets_world"].shape == (n, 5) and out["iou"].shape == (n, m) and out["candidates"].dtype == torch.bool
    if n == 0:
        return
    np.testing.assert_allclose(out["dets_world"].cpu().numpy(), dets_world_host, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["dets_img"].cpu().numpy(), host_rbox.rbox_world_img(dets_world_host, H_img_world), rtol=1e-10, atol=1e-8)
    if m:
        exp_iou = co.rbox_iou(out["dets_world"].cpu().numpy(), trks[:, :5])
        np.testing.assert_allclose(out["iou"].cpu().numpy(), exp_iou, rtol=0, atol=1e-12)
        if n >= 300:
            assert (exp_iou > 0.3).sum() > 100
        np.testing.assert_array_equal(out["candidates"].cpu().numpy(), out["iou"].cpu().numpy() > 0.3)
    # and back: world -> bev is the inverse map (bevwarp_rbox_transform, both directions)
    back = rbox_world_bev_device(out["dets_world"], np.linalg.inv(H_world_bev), "world").cpu().numpy()
    np.testing.assert_allclose(back[:, :4], dets_bev[:, :4], rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(np.angle(np.exp(1j * (back[:, 4] - dets_bev[:, 4]))), 0, atol=1e-9)
    np.testing.assert_allclose(rbox_world_bev_device(torch.from_numpy(dets_bev).cuda(), H_world_bev, "bev").cpu().numpy(), dets_world_host, rtol=1e-12, atol=1e-12)


def test_tracker_step_float32_and_argument_errors():
    from bev_amd.tracker_geom import rbox_world_bev_device, tracker_geometry_step
    dets_bev, trks, dets_world_host, H_world_bev, H_img_world = _tracker_case(64, 48, seed=5)
    out = tracker_geometry_step(torch.from_numpy(dets_bev).float().cuda(), torch.from_numpy(trks).float().cuda(), H_world_bev, 0.25, H_img_world)
    assert out["iou"].dtype == torch.float32
    np.testing.assert_allclose(out["dets_world"].cpu().numpy(), dets_world_host, rtol=2e-6, atol=2e-5)
    exp_iou = co.rbox_iou(out["dets_world"].double().cpu().numpy(), trks[:, :5].astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(out["iou"].cpu().numpy(), exp_iou, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(out["candidates"].cpu().numpy(), out["iou"].cpu().numpy() > np.float32(0.25))
    # H must be a similarity (the reference asserts it): a projective last row and unequal axis scales are refused
    bad = H_world_bev.copy()
    bad[2, 0] = 1e-3
    with pytest.raises(ValueError):
        tracker_geometry_step(dets_bev, trks, bad)
    with pytest.raises(ValueError):
        rbox_world_bev_device(torch.from_numpy(dets_bev).cuda(), np.diag([1.0, 2.0, 1.0]), "bev")
    with pytest.raises(ValueError):
        tracker_geometry_step(dets_bev[:, :4], trks, H_world_bev)


def test_config5_full_step_eager_and_graphed():
    """BASELINE.json configs[4] as one per-camera step: 1080p -> 1024^2 uint8 BEV warp + the tracker launch on 512 x 512
    boxes, launched eagerly and replayed from a captured graph on new frames / boxes."""
    from bev_amd import warp as W
    from tools.graphed_step import GraphedStep
    from bev_amd.tracker_geom import tracker_geometry_step
    from tests import workloads as wl
    n = m = 512
    dets_bev, trks, dets_world_host, H_world_bev, H_img_world = _tracker_case(n, m, seed=3)
    M = wl.synth_brno_H(1920, 1080, 1024, 1024)
    src = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda")
    bev_img = torch.empty((1024, 1024, 3), dtype=torch.uint8, device="cuda")
    minv = W.device_inverse(M, src.device)
    d_dev, t_dev = torch.zeros((n, 5), dtype=torch.float64, device="cuda"), torch.zeros((m, 7), dtype=torch.float64, device="cuda")
    res = tracker_geometry_step(d_dev, t_dev, H_world_bev, 0.3, H_img_world)  # allocates the outputs once

    def step():
        W.warp_perspective(src, None, (1024, 1024), out=bev_img, M_inv_device=minv)
        return tracker_geometry_step(d_dev, t_dev, H_world_bev, 0.3, H_img_world, out=res)

    g = GraphedStep(step)
    for i, mode in enumerate(["eager", "graph", "graph"]):
        f = wl.frame(30 + i, 1080, 1920, np.uint8)
        db, tk, dw_host, _, _ = _tracker_case(n, m, seed=40 + i)
        src.copy_(torch.from_numpy(f).cuda())
        d_dev.copy_(torch.from_numpy(db).cuda())
        t_dev.copy_(torch.from_numpy(tk).cuda())
        out = step() if mode == "eager" else g.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(bev_img.cpu().numpy(), co.warp_perspective(f, M, (1024, 1024), 1, nthreads=8))
        np.testing.assert_allclose(out["dets_world"].cpu().numpy(), dw_host, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(out["iou"].cpu().numpy(), co.rbox_iou(out["dets_world"].cpu().numpy(), tk[:, :5]), rtol=0, atol=1e-12)


def test_composite_reg_img_matches_numpy_expression():
    """bevwarp_composite == the reference's numpy blend (bev/tool/compo.py:16-23), every byte combination that can
    round differently, odd sizes (scalar tail) and an unaligned view."""
    from bev_amd.compo import composite_reg_img
    rng = np.random.default_rng(5)
    for shape in ((37, 53, 3), (256, 256, 3), (1, 1, 1), (16, 1, 1)):
        bg, fg, m = (rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(3))
        mf = m.astype(float) / 255
        exp = (fg.astype(float) * mf + bg.astype(float) * (1 - mf)).round()
        exp[exp > 255] = 255
        got = composite_reg_img(bg, fg, m)  # numpy in -> numpy out, like the reference
        assert isinstance(got, np.ndarray)
        np.testing.assert_array_equal(got, exp.astype(np.uint8))
        np.testing.assert_array_equal(composite_reg_img(torch.from_numpy(bg).cuda(), fg, m).cpu().numpy(), exp.astype(np.uint8))
    # exhaustive over (fg, mask) with two backgrounds
    fgv, mv = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for b in (0, 255, 77):
        bg = np.full_like(fgv, b)
        mf = mv.astype(float) / 255
        exp = (fgv.astype(float) * mf + bg.astype(float) * (1 - mf)).round()
        exp[exp > 255] = 255
        np.testing.assert_array_equal(composite_reg_img(bg[..., None], fgv[..., None], mv[..., None])[..., 0], exp.astype(np.uint8))


def test_frame_pipeline_matches_resident_path():
    """SURVEY.md 8(f2) ingest: pinned host ring -> H2D -> warp -> D2H on three streams; the results are the resident path's."""
    from bev_amd import warp as W
    from bev_amd.pipeline import FramePipeline
    from tests import workloads as wl
    sw, sh, dw, dh = 640, 360, 256, 192
    M = wl.synth_brno_H(sw, sh, dw, dh)
    frames = [wl.frame(60 + i, sh, sw, np.uint8) for i in range(7)]
    exp = [co.warp_perspective(f, M, (dw, dh), 1) for f in frames]
    for zc in (True, False):  # the kernel storing straight into the pinned host slot / a device frame copied down behind it
        pipe = FramePipeline((sh, sw), 3, M, (dw, dh), depth=3, zero_copy_out=zc)
        got = [r.copy() for r in pipe.run(frames)]  # (results are views of a pinned ring)
        assert len(got) == len(frames)
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)
    pipe = FramePipeline((sh, sw), 3, M, (dw, dh), depth=2, planar=True, scale=0.5, bias=1.0)  # planar frames into host slots
    for g, f in zip(pipe.run(frames[:3]), frames[:3]):
        np.testing.assert_array_equal(g, W.warp_to_planar(torch.from_numpy(f).cuda(), M, (dw, dh), scale=0.5, bias=1.0).cpu().numpy())
    # zero-copy ingest (the decoder writes into the pinned slot), results left on the device, planar egress
    pipe = FramePipeline((sh, sw), 3, M, (dw, dh), depth=2, planar=True, scale=[1 / 255.0, 0.5, 2.0], bias=[0.0, -1.0, 3.5], download=False)
    for i, f in enumerate(frames[:4]):
        pipe.next_input()[...] = f
        pipe.commit()
        r = pipe.result()
        ref = W.warp_to_planar(torch.from_numpy(f).cuda(), M, (dw, dh), scale=[1 / 255.0, 0.5, 2.0], bias=[0.0, -1.0, 3.5])
        assert r.is_cuda and torch.equal(r, ref)
    # float32 frames
    f32 = [wl.frame(70 + i, sh, sw, np.float32) for i in range(3)]
    pipe = FramePipeline((sh, sw), 3, M, (dw, dh), dtype=torch.float32)
    for g, f in zip(pipe.run(f32), f32):
        np.testing.assert_array_equal(g, co.warp_perspective(f, M, (dw, dh), 1))


def test_frame_pipeline_refuses_to_overwrite_an_undelivered_result():
    """A ring of depth d holds d undelivered frames: one more commit would store into a slot whose result nobody has taken."""
    from bev_amd.pipeline import FramePipeline
    from tests import workloads as wl
    sw, sh, dw, dh = 320, 180, 128, 96
    M = wl.synth_brno_H(sw, sh, dw, dh)
    frames = [wl.frame(80 + i, sh, sw, np.uint8) for i in range(4)]
    with FramePipeline((sh, sw), 3, M, (dw, dh), depth=3) as pipe:
        for f in frames[:3]:
            pipe.submit(f)
        with pytest.raises(RuntimeError):
            pipe.submit(frames[3])
        assert pipe.ready() == 3
        got = [pipe.result().copy() for _ in range(3)]  # the three delivered frames are intact
        for g, f in zip(got, frames[:3]):
            np.testing.assert_array_equal(g, co.warp_perspective(f, M, (dw, dh), 1))
        pipe.submit(frames[3])
        np.testing.assert_array_equal(pipe.result(), co.warp_perspective(frames[3], M, (dw, dh), 1))
    pipe.close()  # idempotent


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_warp_composite_every_channel_count(c):
    """bevwarp_warp_composite advertises 1-4 channels and bit-identity with three warps + the blend: guarded taps (a small
    odd-sized foreground / mask, maps that leave the frames) and unguarded ones (interior), every C."""
    from bev_amd.compo import composite_bev_img, composite_reg_img
    from bev_amd.homo import homo_from_KRt
    from bev_amd.warp import warp_perspective
    from tests import workloads as wl
    K = np.array([[400.0, 0, 159.0], [0, 395.0, 88.5], [0, 0, 1.0]])
    cth, sth = np.cos(0.8), np.sin(0.8)
    RT = np.array([[1, 0, 0, 0.0], [0, cth, -sth, 2.0], [0, sth, cth, 14.0], [0, 0, 0, 1.0]])
    H_world2bev = np.array([[0.0, 14.0, 250.0], [-14.0, 0.0, 60.0], [0.0, 0.0, 1.0]])
    H_img2world_fix = np.linalg.inv(homo_from_KRt(K, Rt_homo=RT)) @ np.array([[1, 0, 2.0], [0, 1, -1.0], [0, 0, 1]])
    bg = wl.frame(20, 360, 640, np.uint8, c)
    for (fh, fw, dw, dh) in ((177, 319, 301, 517), (360, 640, 512, 96)):
        fg, mask = wl.frame(21, fh, fw, np.uint8, c), wl.frame(22, fh, fw, np.uint8, c)
        Ks = np.diag([fw / 320.0, fh / 178.0, 1.0]) @ K
        got, Hcam = composite_bev_img(torch.from_numpy(bg).cuda(), fg, mask, H_world2bev, H_img2world_fix, Ks, RT, dw, dh)
        Hb, Hc = H_world2bev.dot(H_img2world_fix), H_world2bev.dot(np.linalg.inv(Hcam))
        cu = [torch.from_numpy(x).cuda() for x in (bg, fg, mask)]
        three = composite_reg_img(warp_perspective(cu[0], Hb, (dw, dh)), warp_perspective(cu[1], Hc, (dw, dh)), warp_perspective(cu[2], Hc, (dw, dh)))
        assert got.shape == (dh, dw, c) and torch.equal(three.reshape(dh, dw, c), got)
        fb, ff, fm = (co.warp_perspective(x, H, (dw, dh)).astype(np.float64).reshape(dh, dw, c) for x, H in ((bg, Hb), (fg, Hc), (mask, Hc)))
        exp = np.minimum((ff * (fm / 255) + fb * (1 - fm / 255)).round(), 255).astype(np.uint8)
        np.testing.assert_array_equal(got.cpu().numpy(), exp)
        assert 0.02 < (fm > 0).mean() < 1.0  # the camera footprint really is cut by the frame's edge and not empty
        if c == 3:  # bw_mode (compo.py:13-14): the foreground goes BGR -> grey -> BGR BEFORE the warp; the kernel converts its taps
            gotg, _ = composite_bev_img(torch.from_numpy(bg).cuda(), fg, mask, H_world2bev, H_img2world_fix, Ks, RT, dw, dh, bw_mode=True)
            g = ((fg[..., 0].astype(np.int64) * 1868 + fg[..., 1].astype(np.int64) * 9617 + fg[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
            fgg = np.ascontiguousarray(np.repeat(g[..., None], 3, 2))
            ffg = co.warp_perspective(fgg, Hc, (dw, dh)).astype(np.float64)
            expg = np.minimum((ffg * (fm / 255) + fb * (1 - fm / 255)).round(), 255).astype(np.uint8)
            np.testing.assert_array_equal(gotg.cpu().numpy(), expg)
            assert not np.array_equal(expg, exp)
    if c != 3:
        with pytest.raises(ValueError):
            composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, Ks, RT, dw, dh, bw_mode=True)


def test_d3d_stand_in_identity_on_the_device():
    """ADVICE r03 (GPU twin of tests/test_host_api.py::test_d3d_stand_in_is_consistent_with_the_rebound_tracker_iou): through
    the HIP kernel, stand_in(a + pi/2, b + pi/2) == iou_batch_rbox(a, b) on non-square boxes."""
    import sys

    from bev_amd import overlay
    from bev_amd.iou import iou_batch_rbox
    saved = {k: sys.modules.pop(k) for k in ("d3d", "d3d.box") if k in sys.modules}
    try:
        if not overlay.ensure_d3d():
            pytest.skip("a real d3d is installed")
        import d3d
        rng = np.random.default_rng(4)
        a = np.column_stack([rng.uniform(0, 12, (40, 2)), rng.uniform(1, 2, 40), rng.uniform(3, 6, 40), rng.uniform(-np.pi, np.pi, 40)])
        b = np.column_stack([rng.uniform(0, 12, (30, 2)), rng.uniform(1, 2, 30), rng.uniform(3, 6, 30), rng.uniform(-np.pi, np.pi, 30)])
        turn = np.array([0, 0, 0, 0, np.pi / 2])
        want = iou_batch_rbox(a, b)
        np.testing.assert_allclose(want, co.rbox_iou(a, b), rtol=0, atol=1e-12)
        np.testing.assert_allclose(d3d.box.box2d_iou(a + turn, b + turn, method="rbox"), want, rtol=0, atol=1e-12)
        ta, tb = torch.from_numpy(a + turn).cuda(), torch.from_numpy(b + turn).cuda()  # tensors in, a tensor on the same device out
        got = d3d.box.box2d_iou(ta, tb, method="rbox")
        assert got.is_cuda
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-12)
    finally:
        overlay.remove_d3d_stand_in()
        sys.modules.update(saved)


def test_composite_maps_survive_cache_churn_under_a_captured_graph():
    """composite_bev_img captured into a hipGraph AFTER an eager call (so the capture finds its maps cached) must pin
    them: churning both caches afterwards must not free the address the graph replays."""
    from bev_amd import compo, warp as W
    from bev_amd.homo import homo_from_KRt
    from tests import workloads as wl
    from tools.graphed_step import GraphedStep
    K = np.array([[400.0, 0, 159.0], [0, 395.0, 88.5], [0, 0, 1.0]])
    cth, sth = np.cos(0.8), np.sin(0.8)
    RT = np.array([[1, 0, 0, 0.0], [0, cth, -sth, 2.0], [0, sth, cth, 14.0], [0, 0, 0, 1.0]])
    H_world2bev = np.array([[0.0, 14.0, 250.0], [-14.0, 0.0, 60.0], [0.0, 0.0, 1.0]])
    H_img2world_fix = np.linalg.inv(homo_from_KRt(K, Rt_homo=RT)) @ np.array([[1, 0, 2.0], [0, 1, -1.0], [0, 0, 1]])
    bg, fg, mask = (torch.from_numpy(wl.frame(90 + i, 180, 320, np.uint8)).cuda() for i in range(3))
    eager, _ = compo.composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, K, RT, 256, 128)  # eager first: the maps are cached, not pinned
    holder = {}

    def step():
        holder["out"], _ = compo.composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, K, RT, 256, 128)
        return holder["out"]

    g = GraphedStep(step)
    # churn: evict everything evictable from both caches and let the allocator reuse what was freed
    for i in range(max(W._MINV_CACHE_MAX, compo._MAPS_MAX) + 8):
        Hs = H_world2bev + np.array([[0, 0, float(i + 1)], [0, 0, 0], [0, 0, 0]])
        compo._composite_maps(Hs, H_img2world_fix, K, RT, bg.device)
    junk = [torch.full((2, 3, 3), float("nan"), dtype=torch.float64, device="cuda") for _ in range(512)]
    torch.cuda.synchronize()
    out = g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    del junk


def test_tracker_step_plan_cache_and_iou_out():
    """The validated-launch cache of tracker_geometry_step: a second call with the same device buffers takes the bound launch and
    sees NEW box values; other matrices (same buffers) are a different key; rbox_iou writes into a caller's tensor and checks it."""
    from bev_amd import tracker_geom as tg
    from bev_amd.iou import rbox_iou
    dets_bev, trks, dets_world_host, H_world_bev, H_img_world = _tracker_case(96, 80, seed=8)
    d, t = torch.from_numpy(dets_bev).cuda(), torch.from_numpy(trks).cuda()
    tg._plans.clear()
    out = tg.tracker_geometry_step(d, t, H_world_bev, 0.3, H_img_world)
    assert not tg._plans
    assert tg.tracker_geometry_step(d, t, H_world_bev, 0.3, H_img_world, out=out) is out and len(tg._plans) == 1
    d2, t2, dw2, _, _ = _tracker_case(96, 80, seed=9)
    d.copy_(torch.from_numpy(d2).cuda())
    t.copy_(torch.from_numpy(t2).cuda())
    assert tg.tracker_geometry_step(d, t, H_world_bev, 0.3, H_img_world, out=out) is out and len(tg._plans) == 1  # the hit
    np.testing.assert_allclose(out["dets_world"].cpu().numpy(), dw2, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["iou"].cpu().numpy(), co.rbox_iou(out["dets_world"].cpu().numpy(), t2[:, :5]), rtol=0, atol=1e-12)
    H2 = H_world_bev.copy()
    H2[0, 2] += 1.0
    tg.tracker_geometry_step(d, t, H2, 0.3, H_img_world, out=out)
    assert len(tg._plans) == 2
    assert abs(out["dets_world"].cpu().numpy()[:, 0] - dw2[:, 0] - 1.0).max() < 1e-9
    with pytest.raises(ValueError):
        tg.tracker_geometry_step(d, t, H_world_bev, 0.3, H_img_world, out={"iou": out["iou"]})
    io = torch.empty((96, 80), dtype=torch.float64, device="cuda")
    assert rbox_iou(out["dets_world"], t, out=io) is io
    np.testing.assert_allclose(io.cpu().numpy(), co.rbox_iou(out["dets_world"].cpu().numpy(), t2[:, :5]), rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        rbox_iou(out["dets_world"], t, out=io[:, :40])
    tg._plans.clear()


def test_many_overlapping_pairs_per_wave():
    """Every box of `b` is a jittered copy of ONE box, so all 64 lanes of a workgroup survive the rejection test and sum a contour."""
    from bev_amd.iou import rbox_iou
    rng = np.random.default_rng(13)
    one = np.array([10.0, 20.0, 2.0, 5.0, 0.4])
    a = one[None, :] + rng.normal(0, [0.3, 0.3, 0.05, 0.1, 0.2], (40, 5))
    b = one[None, :] + rng.normal(0, [0.3, 0.3, 0.05, 0.1, 0.2], (150, 5))
    exp = co.rbox_iou(a, b)
    assert (exp > 0.1).mean() > 0.9
    got = rbox_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-12)


def _random_scale_boxes():
    rng = np.random.default_rng(77)
    n = 200
    a = np.column_stack([rng.uniform(-20, 20, (n, 2)), 10 ** rng.uniform(-2, 1.5, n), 10 ** rng.uniform(-2, 1.5, n), rng.uniform(-7, 7, n)])
    b = np.column_stack([rng.uniform(-20, 20, (n, 2)), 10 ** rng.uniform(-2, 1.5, n), 10 ** rng.uniform(-2, 1.5, n), rng.uniform(-7, 7, n)])
    b[:20] = a[:20]                                   # identical
    b[20:40, :2] = a[20:40, :2]                       # concentric, other size and heading
    b[50:60] = a[50:60] + np.array([1e-9, 0, 0, 0, 0])  # almost identical
    return a, b


def test_rbox_iou_random_scales():
    """Boxes of very different sizes (one inside the other, slivers 1 : 3000), concentric, identical and almost identical boxes: the
    square-root-free rejection test must never drop a pair that intersects.  Two yardsticks: the oracle's float64 clip, which works in WORLD
    coordinates and therefore carries ulp(|centre|) / size of error for centimetre boxes twenty metres from the origin (up to 8e-11 here,
    measured against 50-digit arithmetic in tests/test_oracle_iou.py) -- agreement to 1e-12 where the smaller box side is >= 0.5 m, 2e-10
    everywhere; and 50-digit arithmetic itself (tests/exact_iou.py) on every pair the two disagree on by more than 1e-13 plus a sample: 1e-13 (the
    contour sum's own rounding is eps * |edge of A| * half length of B / union: 1e-14 for two 15 m x 1.4 cm slivers crossing, 2e-15 otherwise).
    (Zero-AREA boxes are left out: the oracle's clip returns the other box's area for a degenerate clip polygon; what d3d does there is
    unknown, and the tracker never produces such boxes.)"""
    from bev_amd.iou import rbox_iou
    a, b = _random_scale_boxes()
    exp = co.rbox_iou(a, b)
    got = rbox_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, exp, rtol=0, atol=2e-10)
    big = (np.minimum(a[:, 2], a[:, 3])[:, None] >= 0.5) & (np.minimum(b[:, 2], b[:, 3])[None, :] >= 0.5)
    np.testing.assert_allclose(got[big], exp[big], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.diag(got)[:20], 1.0, atol=1e-12)
    assert (exp > 0).mean() > 0.02  # the contour sum really ran for many pairs (~1,000 of 40,000)
    pytest.importorskip("mpmath")
    from tests.exact_iou import iou as exact
    pairs = {tuple(p) for p in np.argwhere(np.abs(got - exp) > 1e-13)} | {(i, i) for i in range(60)} | {tuple(p) for p in np.argwhere(exp > 0)[::8]}
    assert len(pairs) > 100
    for i, j in sorted(pairs):
        assert abs(got[i, j] - exact(a[i], b[j])) <= 1e-13, (i, j, a[i], b[j])


def test_rbox_iou_conventions_of_the_oracle_for_odd_inputs():
    """Negative sizes are outside the reference's domain; the kernel keeps what the oracle's clip does with them (a box with ONE negative
    size is a clockwise polygon: as the CLIP polygon it removes everything, as the clipped one it behaves like its mirror image), touching
    boxes give exactly 0, and a NaN box gives 0."""
    from bev_amd.iou import rbox_iou
    rng = np.random.default_rng(5)
    box = lambda n: np.column_stack([rng.uniform(0, 8, (n, 2)), rng.uniform(1.6, 2.2, n), rng.uniform(3.5, 6, n), rng.uniform(-np.pi, np.pi, n)])
    a, b = box(100), box(100)
    for m in (a, b):
        m[:, 2] *= rng.choice([-1, 1], 100)
        m[:, 3] *= rng.choice([-1, 1], 100)
    got = rbox_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, co.rbox_iou(a, b), rtol=0, atol=1e-12)
    a = box(50)
    b = a.copy()
    b[:, 0] += a[:, 3] * np.cos(a[:, 4])  # end to end along the heading
    b[:, 1] += a[:, 3] * np.sin(a[:, 4])
    got = rbox_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert (np.diag(got) == 0).all() and (got >= 0).all()
    a[3, 0] = np.nan
    got = rbox_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert np.isfinite(got).all() and (got[3] == 0).all()


def test_rbox_iou_yaws_of_any_size():
    """The kernel's own sin / cos (three-piece reduction by pi / 2, below 2^20 rad) and the library's (beyond): yaws of many turns, on both
    sides of the switch, against the oracle's libm; and the float32 tracker step with the same yaws."""
    from bev_amd.iou import rbox_iou
    rng = np.random.default_rng(21)
    n = 120
    base = np.column_stack([rng.uniform(0, 12, (n, 2)), rng.uniform(1.6, 2.2, n), rng.uniform(3.5, 6, n), rng.uniform(-np.pi, np.pi, n)])
    for turns in (0.0, 1.0, 37.0, 1e3, 1.6e5, 2e5, 1e7):  # 2^20 rad = 166,886 turns
        a, b = base.copy(), base[::-1].copy()
        a[:, 4] += 2 * np.pi * turns * rng.choice([-1, 1], n)
        b[:, 4] += 2 * np.pi * np.floor(turns * rng.uniform(0.5, 1.0, n))
        exp = co.rbox_iou(a, b)
        assert (exp > 0).mean() > 0.15
        got = rbox_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-12, err_msg="turns %g" % turns)
    # quadrant boundaries: yaws at exact multiples of pi / 4 and one ulp either side
    k = np.arange(-16, 17) * (np.pi / 4)
    yaws = np.concatenate([k, np.nextafter(k, np.inf), np.nextafter(k, -np.inf)])
    a = np.column_stack([np.zeros((len(yaws), 2)), np.full(len(yaws), 2.0), np.full(len(yaws), 5.0), yaws])
    b = np.array([[0.3, -0.2, 1.8, 4.4, 0.1], [0.0, 0.0, 2.0, 5.0, 0.0]])
    got = rbox_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, co.rbox_iou(a, b), rtol=0, atol=1e-12)
