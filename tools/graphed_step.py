"""HIP-graph replay of a per-camera step.

`GraphedStep` captures any function of static device buffers once (torch.cuda.CUDAGraph, i.e. hipGraph) and replays it
with one launch.  It pays for steps made of MANY small launches.  The per-camera step of this package no longer is one: the
warp is one launch and the tracker geometry is one launch (bevwarp_tracker_step), and two launches issue eagerly in less
time than one graph launch costs on this runtime (BENCH_r02.json configs[4]: eager 32.0 us, replay 34.8 us; round 1's
twelve-launch step was 160 us eager against 78 us replayed).  Use it when a step chains further kernels of the caller's own
(a detector's pre-processing, say); for the two launches alone call them eagerly -- bench.py reports both."""
import torch


class GraphedStep:
    """Capture `fn()` (which must only touch preallocated device tensors and launch on the current stream) once;
    `replay()` re-issues all of its launches as one graph launch.

    Homographies: a captured launch replays the ADDRESS of its matrix tensor.  Pass `M_inv_device` tensors the caller
    keeps alive, or numpy matrices that the warm-up calls have already made resident -- `bev_amd.warp.device_inverse` pins
    every cached entry it hands out during capture and raises on a cache miss instead of uploading inside the capture."""

    def __init__(self, fn, warmup=3):
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # allocations, caches and lazy initialisation happen outside the capture
                fn()
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(self._graph):
            self.result = fn()

    def replay(self):
        self._graph.replay()
        return self.result
