"""Frame sharding for multi-GPU runs: one process per GPU, frames split contiguously by rank,
NO data-path collective (every output frame depends on its own source frame and its own 3x3 H only,
reference vis_homo.py:85-91).  torch.distributed is used for rendezvous, barriers and the
max-over-ranks wall time of a timed region -- and that control plane runs on `gloo` with CPU tensors:
nothing of this package crosses xGMI and no RCCL communicator is ever created."""
import contextlib
import os
import sys

import torch

CONTROL_BACKEND = "gloo"  # the only backend this package asks for by itself


def frame_shard(n_frames, world_size, rank):
    """[start, stop) of the frames rank `rank` owns: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(n_frames), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


@contextlib.contextmanager
def _stdout_to_stderr():
    """File descriptor 1 points at stderr inside the block: the gloo transport announces its connections ("[Gloo] Rank 0 is
    connected to ...") on the C++ side's stdout, and a bench line's consumer reads stdout as ONE JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def init(backend=None, device=None):
    """Join the process group if WORLD_SIZE > 1.  The group only carries barriers and scalar reductions of timings, so the
    default backend is gloo on CPU tensors whether or not GPUs are present ("nccl" = RCCL stays selectable for callers that
    want device-side collectives of their own; this package never needs them)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: no hostname resolution on the way
        backend = backend or CONTROL_BACKEND
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        with _stdout_to_stderr():
            dist.init_process_group(backend, **kw)
            dist.barrier()  # (gloo connects lazily: its chatter belongs inside the redirected block)
    return rank, local_rank, world


def _reduce_device():
    import torch.distributed as dist
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def barrier(device_sync=True):
    import torch.distributed as dist
    if device_sync and torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device_sync and torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value, device=None):
    """MAX of a python float over all ranks (the job's wall time is its slowest rank's)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or _reduce_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or _reduce_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
