"""Host-side argument validation of bev_amd.warp (no GPU): everything that reaches the kernel as a raw address or a
raw element size is checked first."""
import numpy as np
import pytest
import torch

from bev_amd import warp


def test_out_tensor_must_match_dtype_device_and_size():
    cpu = torch.device("cpu")
    ok = torch.empty((2, 8, 8, 3), dtype=torch.float32)
    warp._check_out(ok, torch.float32, cpu, 2 * 8 * 8 * 3)
    with pytest.raises(ValueError):  # a uint8 `out` for a float32 source would be overrun 4x
        warp._check_out(torch.empty((2, 8, 8, 3), dtype=torch.uint8), torch.float32, cpu, 2 * 8 * 8 * 3)
    with pytest.raises(ValueError):
        warp._check_out(ok, torch.float32, cpu, 2 * 8 * 8 * 3 + 1)
    with pytest.raises(ValueError):  # another device (a host pointer must never reach the kernel)
        warp._check_out(ok, torch.float32, torch.device("cuda", 0), 2 * 8 * 8 * 3)
    with pytest.raises(ValueError):
        warp._check_out(np.zeros(4), torch.float32, cpu, 4)


def test_matrix_tensor_must_be_contiguous_float64_3x3_on_the_device():
    cpu = torch.device("cpu")
    m = torch.eye(3, dtype=torch.float64).repeat(4, 1, 1)
    assert warp._check_minv(m, cpu, 4) == 4
    assert warp._check_minv(m[:1], cpu, 4) == 1
    with pytest.raises(ValueError):
        warp._check_minv(m.float(), cpu, 4)
    with pytest.raises(ValueError):
        warp._check_minv(m.transpose(1, 2), cpu, 4)  # not contiguous
    with pytest.raises(ValueError):
        warp._check_minv(m.reshape(4, 9), cpu, 4)
    with pytest.raises(ValueError):
        warp._check_minv(m[:3], cpu, 4)  # 3 matrices for 4 frames
    with pytest.raises(ValueError):
        warp._check_minv(m, torch.device("cuda", 0), 4)
    with pytest.raises(ValueError):
        warp._check_minv(m.numpy(), cpu, 4)


def test_scalar_border_follows_cv_scalar():
    np.testing.assert_array_equal(warp.scalar_border(7, 3), [7, 0, 0])          # cv::Scalar(7) = (7, 0, 0, 0)
    np.testing.assert_array_equal(warp.scalar_border((1, 2), 3), [1, 2, 0])     # shorter than C: the rest stay 0
    np.testing.assert_array_equal(warp.scalar_border((1, 2, 3, 4), 3), [1, 2, 3])
    np.testing.assert_array_equal(warp.scalar_border(0, 1), [0])
    np.testing.assert_array_equal(warp.scalar_border([5], 4), [5, 0, 0, 0])


def test_matrix_cache_is_a_bounded_lru_that_keeps_recent_entries(monkeypatch):
    monkeypatch.setattr(warp, "_MINV_CACHE_MAX", 4)
    warp._minv_cache.clear()
    mats = [np.eye(3) + np.diag([k, 0, 0]) for k in range(1, 8)]
    first = warp.device_inverse(mats[0], "cpu")
    for m in mats[1:4]:
        warp.device_inverse(m, "cpu")
    assert warp.device_inverse(mats[0], "cpu") is first          # hit: same tensor, now most recently used
    warp.device_inverse(mats[4], "cpu")                          # evicts mats[1], the least recently used
    assert len(warp._minv_cache) == 4
    assert warp.device_inverse(mats[0], "cpu") is first
    again = warp.device_inverse(mats[1], "cpu")
    np.testing.assert_allclose(again.numpy()[0], np.linalg.inv(mats[1]), rtol=1e-15)
    warp._minv_cache.clear()


def test_tracker_out_dict_is_validated_before_its_addresses_reach_the_kernel():
    from bev_amd import tracker_geom as tg
    cpu = torch.device("cpu")

    def mk(n, m, dt=torch.float64, img=True):
        d = {"dets_world": torch.empty((n, 5), dtype=dt), "iou": torch.empty((n, m), dtype=dt), "candidates": torch.empty((n, m), dtype=torch.bool)}
        if img:
            d["dets_img"] = torch.empty((n, 2), dtype=dt)
        return d

    tg._check_out_dict(mk(4, 6), 4, 6, torch.float64, cpu, True)
    tg._check_out_dict(mk(4, 6, img=False), 4, 6, torch.float64, cpu, False)
    for bad in (mk(4, 5), mk(3, 6), mk(4, 6, torch.float32), mk(4, 6, img=False), None, {"dets_world": np.zeros((4, 5))}):
        with pytest.raises(ValueError):  # another n / m, another dtype, no dets_img although H_img_world is given, not a dict of tensors
            tg._check_out_dict(bad, 4, 6, torch.float64, cpu, True)
    d = mk(4, 6)
    d["iou"] = torch.empty((6, 4), dtype=torch.float64).t()  # right shape, not contiguous
    with pytest.raises(ValueError):
        tg._check_out_dict(d, 4, 6, torch.float64, cpu, True)
    d = mk(4, 6)
    d["candidates"] = torch.empty((4, 6), dtype=torch.uint8)
    with pytest.raises(ValueError):
        tg._check_out_dict(d, 4, 6, torch.float64, cpu, True)
    with pytest.raises(ValueError):
        tg._check_out_dict(mk(4, 6), 4, 6, torch.float64, torch.device("cuda", 0), True)


def test_bw_mode_needs_three_channels():
    from bev_amd import compo
    with pytest.raises(ValueError):
        compo.gray_bgr(torch.zeros((4, 4, 1), dtype=torch.uint8))
    g = compo.gray_bgr(torch.tensor([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], dtype=torch.uint8))
    assert g[0, :, 0].tolist() == [29, 150, 76, (1868 * 10 + 9617 * 20 + 4899 * 30 + 8192) >> 14] and (g[..., 0] == g[..., 2]).all()
