"""The N > 1 path THROUGH THE HIP LIBRARY on a one-GPU box: two fresh child processes (gloo, both on device 0), each
warping its frame_shard of a 9-frame batch with its own per-frame homographies; the concatenation must equal the oracle frame
by frame (reference independence argument: /root/reference/vis_homo.py:85-91 -- one frame, one H, no carried state)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import cpu_oracle as co
from tests import workloads as wl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_fresh_processes_warp_their_shards_through_the_hip_path(tmp_path):
    world, n_frames = 2, 9
    sw, sh, dw, dh = 640, 360, 384, 256
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BEV_BENCH_SAME_DEVICE="1", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(tmp_path), str(n_frames), str(sw), str(sh), str(dw), str(dh)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    spans = [np.load(tmp_path / ("span%d.npy" % r)).tolist() for r in range(world)]
    assert spans == [[0, 5], [5, 9]]  # contiguous, every frame exactly once
    got = np.concatenate([np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)])
    assert got.shape == (n_frames, dh, dw, 3)
    base = wl.keystone_H(sw, sh, dw, dh)
    for g in range(n_frames):
        np.testing.assert_array_equal(got[g], co.warp_perspective(wl.frame(g, sh, sw, np.uint8), wl.jitter_H(base, g), (dw, dh), 1), err_msg="frame %d" % g)


def test_bench_n2_under_the_launcher_prints_one_json_line():
    """The driver's N > 1 command -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- end to end on a
    one-GPU box: two fresh ranks (BEV_BENCH_SAME_DEVICE=1 puts both on device 0), the gloo control plane, the barrier-bracketed
    timed region, the max over ranks; stdout must be exactly ONE JSON line (no transport chatter in front of it)."""
    env = dict(os.environ, BEV_BENCH_SAME_DEVICE="1", PYTHONPATH=ROOT)
    env.pop("BEV_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-variants", "--no-configs", "--no-probe"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["unit"] == "Mpix/s"
    assert d["kernel_ms_max_over_ranks"] >= d["kernel_ms_min_over_ranks"] > 0
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "roofline" in d and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    # two ranks' frames over the slowest rank's wall time: the aggregate cannot exceed what the kernel times allow
    assert d["value"] <= 2 * 32 * 1024 * 1024 / 1e6 / (d["kernel_ms_min_over_ranks"] * 1e-3) * 1.05
