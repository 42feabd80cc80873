"""HIP-graph replay of a per-camera step.

A single warp launch is already cheap to issue (11.5 us per resident 1080p frame back to back, tools/bench_geom.py);
a tracker step is a dozen small launches and is launch-bound: 160 us eagerly, 78 us replayed.  `GraphedStep` captures
any function of static device buffers once (torch.cuda.CUDAGraph, i.e. hipGraph) and replays it with one launch."""
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
