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
