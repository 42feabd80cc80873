"""Multi-GPU path on CPU: frames shard by rank with no data-path collective; world_size-2 gloo run."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bev_amd import shard


def test_frame_shard_partitions_every_frame_once():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.frame_shard(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            sizes = [b - a for a, b in spans]
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(sizes) - min(sizes) <= 1
    assert shard.frame_shard(256, 8, 3) == (96, 128)  # BASELINE configs[3]: 32 frames per GPU


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = shard.init(backend="gloo")
    assert (r, w) == (rank, world)
    a, b = shard.frame_shard(n_frames, w, r)
    # each rank "processes" its own frames (here: a per-frame checksum); nothing is exchanged
    frames = [np.random.default_rng(1234 + i).integers(0, 256, (4, 6, 3), dtype=np.uint8) for i in range(a, b)]
    np.save(os.path.join(out_dir, "rank%d.npy" % r), np.array([[i, int(f.sum())] for i, f in zip(range(a, b), frames)], dtype=np.int64))
    shard.barrier(device_sync=False)
    slowest = shard.max_over_ranks(1.0 + r)
    total = shard.sum_over_ranks(b - a)
    assert slowest == float(world) and total == float(n_frames)
    dist.destroy_process_group()


def test_two_rank_gloo_run(tmp_path):
    world, n_frames = 2, 9
    mp.spawn(_worker, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    rows = np.concatenate([np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)])
    assert rows[:, 0].tolist() == list(range(n_frames))  # every frame exactly once, in order
    exp = [int(np.random.default_rng(1234 + i).integers(0, 256, (4, 6, 3), dtype=np.uint8).sum()) for i in range(n_frames)]
    assert rows[:, 1].tolist() == exp


def test_control_plane_is_gloo_and_leaves_stdout_clean(tmp_path):
    """A bench line's consumer reads stdout as ONE JSON line: the gloo transport's connection chatter ("[Gloo] Rank 0 is
    connected to ...", printed by its C++ side on file descriptor 1) must land on stderr.  Two fresh processes, as under the
    launcher.  Also: the package's own backend choice is gloo even where GPUs exist -- no RCCL on the control plane."""
    import subprocess
    import sys
    assert shard.CONTROL_BACKEND == "gloo"
    script = tmp_path / "rank.py"
    script.write_text(
        "import json, sys\n"
        "sys.path.insert(0, %r)\n"
        "from bev_amd import shard\n"
        "import torch.distributed as dist\n"
        "r, _, w = shard.init()\n"
        "assert dist.get_backend() == 'gloo'\n"
        "shard.barrier(device_sync=False)\n"
        "m = shard.max_over_ranks(float(r))\n"
        "if r == 0:\n"
        "    print(json.dumps({'max': m, 'world': w}), flush=True)\n"
        "shard.barrier(device_sync=False)\n"
        "dist.destroy_process_group()\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert outs[1][0].strip() == ""
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1 and "[Gloo]" not in outs[0][0]
    import json
    assert json.loads(lines[0]) == {"max": 1.0, "world": 2}
