#!/usr/bin/env python3
"""bench.py -- BEV Mpix/s of the batched homography warp (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|u8] [--interp linear|nearest]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one launch of the hot path over one batch of synthetic frames already resident in HBM:
BASELINE.json configs[1] = 32 x (1920x1080x3 -> 1024x1024x3) bilinear warp, "keystone" homography with a
per-frame +-2 px jitter.  The headline line is the float32-pixel variant (north_star:
"float bilinear"); the 8-bit variants of the same workload (the pixel type of the reference's video
frames) are timed in the same run and reported under "variants" (N = 1 only).  Frames shard by rank
with no data-path collective (weak scaling: every rank warps its own 32 frames); torch.distributed is
used for the barrier and the max-over-ranks time only.  Several distinct buffer sets (> 1 GB in total)
are rotated so that a step never finds its frames in the 256 MB Infinity Cache left by the previous one.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / mean launch
duration from HIP events on the launch stream; algorithmic bytes = every destination byte once +
every distinct in-bounds source pixel touched by any tap once (exact footprint, counted on the GPU by
bevwarp_footprint).  `roofline.traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes
(profiles/pmc_traffic.json; FETCH_SIZE x 2 on gfx950 + WRITE_SIZE).  `cpu_baseline` times the CPU oracle
(plain-C restatement of the reference's cv2.warpPerspective path) on this box's host cores on a bounded
sample of the same workload and doubles as the checker of the GPU output.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec; ~5.0-6.0 TB/s streaming copy on this box)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--dtype", choices=["u8", "f32"], default="f32")
    p.add_argument("--interp", choices=["linear", "nearest"], default="linear")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--src", type=int, nargs=2, default=[1920, 1080], metavar=("W", "H"))
    p.add_argument("--dst", type=int, nargs=2, default=[1024, 1024], metavar=("W", "H"))
    p.add_argument("--homography", choices=["keystone", "brno"], default="keystone")
    p.add_argument("--sets", type=int, default=0, help="distinct buffer sets to rotate (0 = enough for > 1 GB)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-variants", action="store_true", help="skip the extra 8-bit measurements (N = 1)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the cpu_baseline sample")
    return p.parse_args()


def cpu_baseline(frames_np, Ms, dsize, interp, gpu_out, budget_s):
    """Time the oracle (kind "port") on a bounded sample; also use it as the checker of the GPU output."""
    from oracle import cpu_oracle as co
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # a one-GPU box owns a 16-core share of the host
    dw, dh = dsize
    mpix = dw * dh / 1e6
    nf = len(frames_np)
    n, t_all, ok = 0, 0.0, True
    t_end = time.perf_counter() + 0.6 * budget_s
    while n < nf or time.perf_counter() < t_end:  # cycle over the sample frames until the budget is spent
        t0 = time.perf_counter()
        exp = co.warp_perspective(frames_np[n % nf], Ms[n % nf], dsize, interp, nthreads=cores)
        t_all += time.perf_counter() - t0
        if n < nf:
            ok = ok and np.array_equal(exp, gpu_out[n])
        n += 1
    m, t_one = 0, 0.0
    t_end = time.perf_counter() + 0.4 * budget_s
    while m < 1 or time.perf_counter() < t_end:
        t0 = time.perf_counter()
        co.warp_perspective(frames_np[m % nf], Ms[m % nf], dsize, interp, nthreads=1)
        t_one += time.perf_counter() - t0
        m += 1
    return {
        "value": round(n * mpix / t_all, 2), "unit": "Mpix/s", "cores": cores, "kind": "port",
        "sample": "%d warps of the step's first %d frames by oracle/liboracle.so with %d OpenMP threads (%.1f s); "
                  "single thread: %d warps (%.1f s)" % (n, nf, cores, t_all, m, t_one),
        "single_thread_value": round(m * mpix / t_one, 2),
        "gpu_output_matches_oracle": bool(ok),
    }


def load_traffic(dtype, interp, args):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), or None when the
    workload is not the one that was profiled (configs[1] at its default shape)."""
    if (args.batch, tuple(args.src), tuple(args.dst), args.homography) != (32, (1920, 1080), (1024, 1024), "keystone"):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get("%s_%s" % (dtype, interp), {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


class Workload:
    """One dtype / interpolation variant of configs[1] resident on one GPU."""

    def __init__(self, args, dtype, interp_name, rank, dev, planar=False):
        from bev_amd import warp
        self.planar = planar  # uint8 in, normalised float32 channel planes out (bevwarp_warp_planar)
        from tests import workloads as wl
        self.warp, self.dtype, self.interp_name, self.args = warp, dtype, interp_name, args
        self.B = B = args.batch
        self.sw, self.sh = sw, sh = args.src
        self.dw, self.dh = dw, dh = args.dst
        C = self.C = 3
        tdtype, ndtype, esz = (torch.uint8, np.uint8, 1) if dtype == "u8" else (torch.float32, np.float32, 4)
        self.esz = esz
        self.interp = warp.INTER_LINEAR if interp_name == "linear" else warp.INTER_NEAREST
        base = (wl.keystone_H if args.homography == "keystone" else wl.synth_brno_H)(sw, sh, dw, dh)
        gidx = [rank * B + i for i in range(B)]  # global frame indices of this rank
        self.Ms = np.stack([wl.jitter_H(base, g) for g in gidx])
        out_esz = 4 if planar else esz
        self.set_bytes = B * (sh * sw * esz + dh * dw * out_esz) * C
        self.nsets = args.sets or max(2, int(np.ceil(1.1e9 / self.set_bytes)))
        self.frames_np = [wl.frame(g, sh, sw, ndtype) for g in gidx[:min(B, 8)]]
        self.srcs, self.dsts = [], []
        for s in range(self.nsets):
            t = torch.empty((B, sh, sw, C), dtype=tdtype, device=dev)
            for i in range(B):
                if s == 0 and i < len(self.frames_np):
                    t[i] = torch.from_numpy(self.frames_np[i]).to(dev)
                elif s == 0:
                    t[i] = torch.from_numpy(wl.frame(gidx[i], sh, sw, ndtype)).to(dev)
                else:  # other sets: same statistics, different bytes (cheap on-device generation)
                    t[i] = self.srcs[0][(i + s) % B].flip(0) if s % 2 else self.srcs[0][(i + s) % B].flip(1)
            self.srcs.append(t)
            self.dsts.append(torch.empty((B, C, dh, dw), dtype=torch.float32, device=dev) if planar else torch.empty((B, dh, dw, C), dtype=tdtype, device=dev))
        self.minv = warp.device_inverse(self.Ms, dev)
        counts, touched = warp.footprint((sh, sw), self.Ms, (dw, dh), flags=self.interp, device=dev)
        self.footprint_px = int(counts.sum().item())
        del touched
        self.algo_bytes = B * dh * dw * C * out_esz + self.footprint_px * C * esz

    def step(self, i):
        k = i % self.nsets
        if self.planar:
            self.warp.warp_to_planar(self.srcs[k], None, (self.dw, self.dh), flags=self.interp, out=self.dsts[k], M_inv_device=self.minv)
            return
        self.warp.warp_perspective(self.srcs[k], None, (self.dw, self.dh), flags=self.interp, out=self.dsts[k], M_inv_device=self.minv)

    def run(self, steps, warmup, barrier):
        for i in range(warmup):
            self.step(i)
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record()
            self.step(i)
            ev[i][1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        launch_ms = np.array([a.elapsed_time(b) for a, b in ev])
        return elapsed, launch_ms

    def roofline(self, launch_ms):
        kernel_s = float(launch_ms.mean()) / 1e3
        achieved = self.algo_bytes / kernel_s / 1e9
        return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None if self.planar else load_traffic(self.dtype, self.interp_name, self.args),
                "kernel": "warp_gather<%s,3,%s>%s" % ("uint8" if self.esz == 1 else "float", self.interp_name, " -> float32 planes" if self.planar else ""),
                "algorithmic_bytes_per_launch": self.algo_bytes, "footprint_px_per_launch": self.footprint_px,
                "kernel_ms_mean": round(kernel_s * 1e3, 4), "kernel_ms_min": round(float(launch_ms.min()), 4),
                "kernel_mpix_per_s": round(self.B * self.dw * self.dh / 1e6 / kernel_s, 1)}


def main():
    args = parse()
    from bev_amd import shard
    rank, local_rank, world = shard.env_rank()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    # Rehearsal knobs for a 1-GPU box (never needed on a real node): BEV_BENCH_SAME_DEVICE=1 puts every rank on
    # device 0 and BEV_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    dev_index = 0 if os.environ.get("BEV_BENCH_SAME_DEVICE") == "1" else local_rank
    backend = os.environ.get("BEV_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    shard.init(backend=backend, device=dev)  # no-op for one process; RCCL only carries the barrier / max below

    main_wl = Workload(args, args.dtype, args.interp, rank, dev)
    elapsed, launch_ms = main_wl.run(args.steps, args.warmup, shard.barrier)
    elapsed = shard.max_over_ranks(elapsed)

    if rank == 0:
        B, dw, dh, sw, sh = main_wl.B, main_wl.dw, main_wl.dh, main_wl.sw, main_wl.sh
        mpix_total = world * B * dw * dh * args.steps / 1e6
        result = {
            "metric": "BEV Mpix/s, 1080p->1024^2 warp; achieved HBM GB/s vs peak",
            "value": round(mpix_total / elapsed, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: batch=%d %dx%dx3 -> %dx%dx3 %s warp per GPU, %s pixels, %s homography (+-2 px per-frame jitter)"
                                   % (B, sw, sh, dw, dh, args.interp, "float32" if args.dtype == "f32" else "uint8", args.homography),
                       "frames_per_gpu": B, "sharding": "frames split by rank, no collectives",
                       "buffer_sets_rotated": main_wl.nsets, "resident_bytes_per_gpu": main_wl.nsets * main_wl.set_bytes},
            "roofline": main_wl.roofline(launch_ms),
        }
        if world == 1 and not args.no_cpu_baseline:
            gpu_out = main_wl.dsts[0][:len(main_wl.frames_np)].cpu().numpy()  # set 0 holds the seeded frames
            result["cpu_baseline"] = cpu_baseline(main_wl.frames_np, main_wl.Ms, (dw, dh), main_wl.interp, gpu_out, args.cpu_seconds)
    del main_wl
    torch.cuda.empty_cache()

    if world == 1 and not args.no_variants:
        variants = []
        for dt, ip, planar in (("u8", "linear", False), ("u8", "nearest", False), ("f32", "linear", False), ("u8", "linear", True)):
            if (dt, ip) == (args.dtype, args.interp) and not planar:
                continue
            w = Workload(args, dt, ip, rank, dev, planar=planar)
            el, lm = w.run(max(20, args.steps // 2), max(5, args.warmup // 2), shard.barrier)
            n = max(20, args.steps // 2)
            variants.append({"dtype": dt if not planar else "u8 -> f32 planar", "interp": ip, "value": round(w.B * w.dw * w.dh * n / 1e6 / el, 1), "unit": "Mpix/s",
                             "ms_per_step": round(el / n * 1e3, 4), "roofline": w.roofline(lm)})
            del w
            torch.cuda.empty_cache()
        result["variants"] = variants

    if rank == 0:
        print(json.dumps(result), flush=True)
    shard.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
