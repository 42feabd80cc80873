#!/usr/bin/env python3
"""bench.py -- BEV Mpix/s of the batched homography warp (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype u8|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one launch of the hot path over one batch of synthetic frames already resident in HBM:
BASELINE.json configs[1] = 32 x (1920x1080x3 -> 1024x1024x3), bilinear, "keystone" homography with a
per-frame +-2 px jitter.  Frames shard by rank with no data-path collective (weak
scaling: every rank warps its own 32 frames); torch.distributed is used for the barrier and the
max-over-ranks time only.  Several distinct buffer sets (> 1 GB in total) are rotated so that a step
never finds its frames in the 256 MB Infinity Cache left by the previous one.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch / mean launch
duration from HIP events on the launch stream; algorithmic bytes = every destination byte once +
every distinct in-bounds source pixel touched by any tap once (exact footprint, counted on the GPU by
bevwarp_footprint).  `cpu_baseline` times the CPU oracle (plain-C restatement of the reference's
cv2.warpPerspective path) on this box's host cores on a bounded sample of the same workload.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--dtype", choices=["u8", "f32"], default="u8")
    p.add_argument("--interp", choices=["linear", "nearest"], default="linear")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--src", type=int, nargs=2, default=[1920, 1080], metavar=("W", "H"))
    p.add_argument("--dst", type=int, nargs=2, default=[1024, 1024], metavar=("W", "H"))
    p.add_argument("--homography", choices=["keystone", "brno"], default="keystone")
    p.add_argument("--sets", type=int, default=0, help="distinct buffer sets to rotate (0 = enough for > 1 GB)")
    p.add_argument("--no-cpu-baseline", action="store_true")
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
    # all cores (OpenMP over 16-row stripes)
    n, t_all, ok = 0, 0.0, True
    t_end = time.perf_counter() + 0.6 * budget_s
    while n < len(frames_np) and (n < 2 or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        exp = co.warp_perspective(frames_np[n], Ms[n], dsize, interp, nthreads=cores)
        t_all += time.perf_counter() - t0
        ok = ok and np.array_equal(exp, gpu_out[n])
        n += 1
    # one core
    m, t_one = 0, 0.0
    t_end = time.perf_counter() + 0.4 * budget_s
    while m < len(frames_np) and (m < 1 or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        co.warp_perspective(frames_np[m], Ms[m], dsize, interp, nthreads=1)
        t_one += time.perf_counter() - t0
        m += 1
    return {
        "value": round(n * mpix / t_all, 2), "unit": "Mpix/s", "cores": cores, "kind": "port",
        "sample": "%d of the step's frames warped by oracle/liboracle.so with %d OpenMP threads (%.1f s); "
                  "single thread: %d frames" % (n, cores, t_all, m),
        "single_thread_value": round(m * mpix / t_one, 2),
        "gpu_output_matches_oracle": bool(ok),
    }


def load_traffic(dtype, interp):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("%s_%s" % (dtype, interp), {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from bev_amd import warp
    from tests import workloads as wl

    B = args.batch
    sw, sh = args.src
    dw, dh = args.dst
    C = 3
    tdtype, ndtype, esz = (torch.uint8, np.uint8, 1) if args.dtype == "u8" else (torch.float32, np.float32, 4)
    interp = warp.INTER_LINEAR if args.interp == "linear" else warp.INTER_NEAREST

    # ---- frames of THIS rank (global frame index = rank * B + i): synthetic, resident before timing
    base = (wl.keystone_H if args.homography == "keystone" else wl.synth_brno_H)(sw, sh, dw, dh)
    gidx = [rank * B + i for i in range(B)]
    Ms = np.stack([wl.jitter_H(base, g) for g in gidx])
    set_bytes = B * (sh * sw + dh * dw) * C * esz
    nsets = args.sets or max(2, int(np.ceil(1.1e9 / set_bytes)))
    srcs, dsts = [], []
    frames_np = [wl.frame(g, sh, sw, ndtype) for g in gidx[:min(B, 8)]]
    for s in range(nsets):
        t = torch.empty((B, sh, sw, C), dtype=tdtype, device=dev)
        for i in range(B):
            if s == 0 and i < len(frames_np):
                t[i] = torch.from_numpy(frames_np[i]).to(dev)
            elif s == 0:
                t[i] = torch.from_numpy(wl.frame(gidx[i], sh, sw, ndtype)).to(dev)
            else:  # other sets: same statistics, different bytes (cheap on-device generation)
                t[i] = srcs[0][(i + s) % B].flip(0) if s % 2 else srcs[0][(i + s) % B].flip(1)
        srcs.append(t)
        dsts.append(torch.empty((B, dh, dw, C), dtype=tdtype, device=dev))
    minv = warp.device_inverse(Ms, dev)

    def step(i):
        k = i % nsets
        warp.warp_perspective(srcs[k], None, (dw, dh), flags=interp, out=dsts[k], M_inv_device=minv)

    # ---- exact footprint of this rank's batch (algorithmic source bytes)
    counts, _ = warp.footprint((sh, sw), Ms, (dw, dh), flags=interp, device=dev)
    footprint_px = int(counts.sum().item())
    del _
    algo_bytes = B * dh * dw * C * esz + footprint_px * C * esz

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step(i)
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    launch_ms = np.array([a.elapsed_time(b) for a, b in ev])
    kernel_s = float(launch_ms.mean()) / 1e3

    result = None
    if rank == 0:
        mpix_total = world * B * dw * dh * args.steps / 1e6
        achieved = algo_bytes / kernel_s / 1e9
        result = {
            "metric": "BEV Mpix/s, 1080p->1024^2 warp; achieved HBM GB/s vs peak",
            "value": round(mpix_total / elapsed, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: batch=%d %dx%dx3 -> %dx%dx3 %s warp per GPU, %s homography (+-2 px per-frame jitter)"
                                   % (B, sw, sh, dw, dh, args.interp, args.homography),
                       "frames_per_gpu": B, "sharding": "frames split by rank, no collectives",
                       "buffer_sets_rotated": nsets, "resident_bytes_per_gpu": nsets * set_bytes},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(args.dtype, args.interp),
                         "kernel": "warp_tiles<%s,3,%s>" % ("uint8" if esz == 1 else "float", args.interp),
                         "algorithmic_bytes_per_launch": algo_bytes, "footprint_px_per_launch": footprint_px,
                         "kernel_ms_mean": round(kernel_s * 1e3, 4), "kernel_ms_min": round(float(launch_ms.min()), 4),
                         "kernel_mpix_per_s": round(B * dw * dh / 1e6 / kernel_s, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            gpu_out = dsts[0][:len(frames_np)].cpu().numpy()  # set 0 holds the seeded frames
            # (set 0 was last written by a step with i % nsets == 0: same inputs, same matrices)
            result["cpu_baseline"] = cpu_baseline(frames_np, Ms, (dw, dh), interp, gpu_out, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
