"""One rank of the N > 1 path, as a fresh process (started by tests/test_gpu_shard.py with RANK / WORLD_SIZE / MASTER_* set):
joins the process group, warps ITS frame_shard of the global batch through the HIP library, saves the result.  Nothing is
exchanged between ranks on the data path; the process group carries the barrier and the max-over-ranks time only -- exactly
what bench.py --gpus N does.  One device stands in for N on a one-GPU box; the control plane is gloo there and on a real node alike."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    out_dir, n_frames = sys.argv[1], int(sys.argv[2])
    sw, sh, dw, dh = (int(v) for v in sys.argv[3:7])
    from bev_amd import shard, warp
    from tests import workloads as wl
    rank, _, world = shard.init(backend=os.environ.get("BEV_BENCH_BACKEND"))  # (None: the package's control backend, gloo)
    dev = torch.device("cuda", 0 if os.environ.get("BEV_BENCH_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    a, b = shard.frame_shard(n_frames, world, rank)
    base = wl.keystone_H(sw, sh, dw, dh)
    Ms = np.stack([wl.jitter_H(base, g) for g in range(a, b)])
    frames = torch.from_numpy(np.stack([wl.frame(g, sh, sw, np.uint8) for g in range(a, b)])).to(dev)
    shard.barrier()
    t0 = time.perf_counter()
    out = warp.warp_perspective(frames, Ms, (dw, dh))
    shard.barrier()
    slowest = shard.max_over_ranks(time.perf_counter() - t0)
    total = shard.sum_over_ranks(b - a)
    assert total == float(n_frames) and slowest > 0
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), out.cpu().numpy())
    np.save(os.path.join(out_dir, "span%d.npy" % rank), np.array([a, b]))
    shard.barrier()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
