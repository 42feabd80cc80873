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
