#!/usr/bin/env python3
"""bench.py -- BEV Mpix/s of the batched homography warp (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|u8] [--interp linear|nearest] [--config 1|3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config 3]

A "step" is one launch of the hot path over one batch of synthetic frames already resident in HBM.  The headline is
BASELINE.json configs[1] = 32 x (1920x1080x3 -> 1024x1024x3) bilinear warp per GPU, "keystone" homography with a per-frame
+-2 px jitter, float32 pixels (north_star: "float bilinear"); `--config 3` switches every rank to its
shard of configs[3] (32 x (3840x2160 -> 2048x2048), 256 frames over 8 GPUs) with the same JSON schema.  Frames shard by
rank with no data-path collective (weak scaling); torch.distributed carries the barrier and the max-over-ranks time only.
Several distinct buffer sets (> 1 GB in total) are rotated so that a step never finds its frames in the 256 MB Infinity
Cache left by the previous one.

ONE JSON line (rank 0).  At N = 1 it also carries, measured in the same run:
  variants   the other pixel types of configs[1] -- uint8 bilinear (the reference's own pixel type, vis_homo.py:86-89) and
             nearest, uint8 -> float32 planes -- and the rotated "brno" footprint for uint8 and float32
  configs    configs[0] (one 720p -> 512^2 uint8 frame: GPU resident / PCIe-inclusive, CPU oracle 1 thread and all cores,
             cv2 when this box happens to have it), configs[2] (1e7 points, f32 / f64), configs[3] (the per-GPU shard,
             uint8 and float32), configs[4] (1080p -> 1024^2 uint8 warp + the 512 x 512 tracker launch), and the
             PCIe-inclusive frame pipeline
  summary    LAST key, <= 1500 characters: {entry: [ms per step (kernel mean), fraction of 8 TB/s, HBM traffic / algorithmic bytes,
             GPU output == oracle]} for every variant and config above, so that the tail of the line alone carries every number
`roofline.measured_ceiling_gbs` = what a pure streaming kernel with the headline's byte mix (its algorithmic source bytes
read with plain 16-byte loads, its destination bytes written with plain 16-byte stores -- the warp's access kinds)
reaches in THIS process on THIS box, on the headline's own buffers, just before the headline runs (tools/streamprobe.hip;
`measured_ceiling_nt_nt_gbs`: non-temporal loads as well, which gathers cannot use); `frac_of_measured` = achieved / that;
`sclk_mhz` = the shader clock the chip held inside the warp kernel (s_memtime / s_memrealtime stamps of the diagnostic
build csrc/variants/clock.so, the method of tools/clock.py) and inside the probe.  u8 variants also carry `roofline_valu`
(vector-ALU instructions per 256-px wave-row, their issue time, VALU busy fraction) from the committed SQ counter pass, and
`bound` says which of the two the counters name.
`roofline.achieved` = algorithmic bytes per launch / mean launch duration from HIP events on the launch stream; algorithmic
bytes = every destination byte once + every distinct in-bounds source pixel touched by any tap once (exact footprint,
counted on the GPU by bevwarp_footprint).  `roofline.traffic` = HBM bytes per launch from the committed rocprofv3 --pmc
passes (profiles/pmc_traffic.json; FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) -- null unless that file was produced from the
kernel source this run was built from.  `cpu_baseline` times the CPU oracle (plain-C restatement of the reference's
cv2.warpPerspective path) on this box's host cores on a bounded sample of the headline workload and doubles as the checker
of the GPU output.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec; ~5.0-6.3 TB/s streaming copy)


def kernel_sources():
    """Every source the warp kernels and their launch geometry are built from (bev_amd/csrc: *.hip, *.h, *.inc), sorted."""
    d = os.path.join(ROOT, "bev_amd", "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h", ".inc")) and not f.startswith("geom_")]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--dtype", choices=["u8", "f32"], default="f32")
    p.add_argument("--interp", choices=["linear", "nearest"], default="linear")
    p.add_argument("--config", type=int, choices=[1, 3], default=1, help="BASELINE.json configs[] index of the per-GPU workload")
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--src", type=int, nargs=2, default=None, metavar=("W", "H"))
    p.add_argument("--dst", type=int, nargs=2, default=None, metavar=("W", "H"))
    p.add_argument("--homography", choices=["keystone", "brno"], default="keystone")
    p.add_argument("--sets", type=int, default=0, help="distinct buffer sets to rotate (0 = enough for > 1 GB)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-probe", action="store_true", help="skip the same-process streaming-ceiling probe and the in-kernel clock reading")
    p.add_argument("--no-variants", action="store_true", help="skip the extra measurements of configs[1] (N = 1)")
    p.add_argument("--no-configs", action="store_true", help="skip the configs block (N = 1)")
    p.add_argument("--cpu-seconds", type=float, default=3.0, help="wall budget of the cpu_baseline sample (the oracle needs < 1 s of 16 cores to time)")
    p.add_argument("--leg-steps", type=int, default=300, help="timed launches of every side measurement (variants, configs[3]); --steps governs the headline only")
    a = p.parse_args()
    shape = {1: ([1920, 1080], [1024, 1024]), 3: ([3840, 2160], [2048, 2048])}[a.config]
    a.src = a.src or shape[0]
    a.dst = a.dst or shape[1]
    return a


def kernel_source_sha():
    h = hashlib.sha256()
    for path in kernel_sources():
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))  # a one-GPU box owns a 16-core share of the host


def cv2_leg(frames_np, Ms, dsize, interp, oracle_out):
    """Opportunistic: when this box already has OpenCV, time cv2.warpPerspective on the same sample and
    report its largest pixel difference from the oracle.  Never a requirement, never installed."""
    try:
        import cv2  # noqa: F401
    except Exception:
        return None
    if not hasattr(cv2, "warpPerspective"):
        return None
    flag = cv2.INTER_LINEAR if interp == 1 else cv2.INTER_NEAREST
    mpix = dsize[0] * dsize[1] / 1e6
    out = {"version": getattr(cv2, "__version__", "?")}
    diff = 0.0
    for f, M, o in zip(frames_np, Ms, oracle_out):
        diff = max(diff, float(np.abs(cv2.warpPerspective(f, M, dsize, flags=flag).astype(np.float64) - o.astype(np.float64)).max()))
    out["max_abs_diff_vs_oracle"] = diff
    for label, nt in (("single_thread_value", 1), ("value", 0)):
        cv2.setNumThreads(nt)
        t0, n = time.perf_counter(), 0
        while n < len(frames_np) or time.perf_counter() - t0 < 1.0:
            cv2.warpPerspective(frames_np[n % len(frames_np)], Ms[n % len(frames_np)], dsize, flags=flag)
            n += 1
        out[label] = round(n * mpix / (time.perf_counter() - t0), 2)
    out["unit"] = "Mpix/s"
    return out


def cpu_baseline(frames_np, Ms, dsize, interp, gpu_out, budget_s):
    """Time the oracle (kind "port") on a bounded sample; also use it as the checker of the GPU output."""
    from oracle import cpu_oracle as co
    cores = host_cores()
    dw, dh = dsize
    mpix = dw * dh / 1e6
    nf = len(frames_np)
    n, t_all, ok, oracle_out = 0, 0.0, True, []
    t_end = time.perf_counter() + 0.6 * budget_s
    while n < nf or time.perf_counter() < t_end:  # cycle over the sample frames until the budget is spent
        t0 = time.perf_counter()
        exp = co.warp_perspective(frames_np[n % nf], Ms[n % nf], dsize, interp, nthreads=cores)
        t_all += time.perf_counter() - t0
        if n < nf:
            ok = ok and np.array_equal(exp, gpu_out[n])
            oracle_out.append(exp)
        n += 1
    m, t_one = 0, 0.0
    t_end = time.perf_counter() + 0.4 * budget_s
    while m < 1 or time.perf_counter() < t_end:
        t0 = time.perf_counter()
        co.warp_perspective(frames_np[m % nf], Ms[m % nf], dsize, interp, nthreads=1)
        t_one += time.perf_counter() - t0
        m += 1
    res = {
        "value": round(n * mpix / t_all, 2), "unit": "Mpix/s", "cores": cores, "kind": "port",
        "sample": "%d warps of the step's first %d frames by oracle/liboracle.so with %d OpenMP threads (%.1f s); "
                  "single thread: %d warps (%.1f s)" % (n, nf, cores, t_all, m, t_one),
        "single_thread_value": round(m * mpix / t_one, 2),
        "gpu_output_matches_oracle": bool(ok),
    }
    cv = cv2_leg(frames_np, Ms, dsize, interp, oracle_out)
    res["cv2_opportunistic"] = cv if cv is not None else "cv2 not importable on this box"
    return res


def load_profile(dtype, interp, homography, args):
    """The committed rocprofv3 --pmc record of this workload (profiles/pmc_traffic.json: HBM bytes per launch, SQ counters), or
    None when the workload is not one that was profiled (configs[1] at its default shape, keystone or brno) or the kernel
    source has changed since."""
    if (args.batch, tuple(args.src), tuple(args.dst)) != (32, (1920, 1080), (1024, 1024)):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("kernel_source_sha") != kernel_source_sha():
        return None
    return d.get("%s_%s%s" % (dtype, interp, "" if homography == "keystone" else "_" + homography))


def load_traffic(dtype, interp, homography, args):
    rec = load_profile(dtype, interp, homography, args)
    return None if rec is None else rec.get("hbm_bytes_per_launch")


_probe_lib = None


def probe_lib():
    """tools/libstreamprobe.so (built by __graft_entry__.build(); measurement aid, not part of libbevwarp.so)."""
    global _probe_lib
    if _probe_lib is None:
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libstreamprobe.so"))
        lib.probe_stream.restype = ctypes.c_int
        lib.probe_stream.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p]
        _probe_lib = lib
    return _probe_lib


def measured_ceiling(w):
    """Same process, same buffers, same byte mix as workload `w`: read its algorithmic source bytes, write its destination
    bytes, nothing else.  Returns GB/s by access kind (median over launches, best grid) and the clock the probe ran at."""
    import ctypes
    try:
        lib = probe_lib()
    except OSError as e:
        return {"error": "tools/libstreamprobe.so not built: %s" % e}
    read_b = min(w.footprint_px * w.C * w.esz, w.srcs[0].numel() * w.esz) // 16 * 16
    write_b = w.dsts[0].numel() * w.dsts[0].element_size() // 16 * 16
    clk = torch.zeros(3, dtype=torch.int64, device=w.srcs[0].device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {"read_bytes": read_b, "write_bytes": write_b}
    for label, ntl, nts in (("plain_loads_plain_stores", 0, 0), ("plain_loads_nt_stores", 0, 1), ("nt_loads_nt_stores", 1, 1)):
        best = None
        for grid in (4096, 8192):
            k = [0]

            def launch():
                i = k[0] % w.nsets
                k[0] += 1
                rc = lib.probe_stream(w.srcs[i].data_ptr(), read_b, w.dsts[i].data_ptr(), write_b, ntl, nts, grid, stream, clk.data_ptr())
                if rc:
                    raise RuntimeError("probe_stream failed: %d" % rc)

            t = float(np.median(event_times(launch, 20, 4)))
            best = t if best is None else min(best, t)
        out[label] = {"us": round(best * 1e6, 1), "gbs": round((read_b + write_b) / best / 1e9, 1)}
    c = clk.cpu().numpy()
    out["sclk_mhz_probe"] = round(100.0 * float(c[0]) / max(float(c[1]), 1.0), 0)
    return out


def kernel_clock(w, launches=300):
    """Shader clock held INSIDE the warp kernel: the diagnostic build of the same sources (csrc/variants/clock.so, -DBEVWARP_CLOCK:
    one s_memtime / s_memrealtime stamp pair per workgroup) launched back to back on the headline's buffers."""
    import ctypes
    path = os.path.join(ROOT, "bev_amd", "csrc", "variants", "clock.so")
    if not os.path.exists(path):
        return None
    from bev_amd import _lib
    lib = ctypes.CDLL(path)
    fn = lib.bevwarp_warp
    fn.restype, fn.argtypes = _lib.SYMBOLS["bevwarp_warp"]
    dbg = lib.bevwarp_debug_clock
    dbg.restype, dbg.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    esz, stream = w.esz, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out4 = (ctypes.c_ulonglong * 16)()  # (16 words: csrc/warp_rows.h kClkWords; [0..2] = shader ticks, 100-MHz ticks, workgroups)

    def run(n):
        for i in range(n):
            s, d = w.srcs[i % w.nsets], w.dsts[i % w.nsets]
            rc = fn(s.data_ptr(), d.data_ptr(), w.B, w.sh, w.sw, w.dh, w.dw, w.C, s.stride(0) * esz, s.stride(1) * esz, d.stride(0) * esz, d.stride(1) * esz,
                    w.minv.data_ptr(), w.minv.shape[0], 0 if esz == 1 else 1, w.interp, None, stream)
            if rc:
                raise RuntimeError("clock build: bevwarp_warp returned %d" % rc)
        torch.cuda.synchronize()

    run(launches)          # reach the sustained state
    assert dbg(out4, 1) == 0
    run(50)
    assert dbg(out4, 0) == 0
    return round(100.0 * out4[0] / max(out4[1], 1), 0) if out4[2] else None


def event_times(fn, n, warm):
    """Per-call durations (s) of fn() from HIP events on the current stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in ev]) * 1e-3


def back_to_back_us(fn, n=500, warm=20):
    """Mean time per call (us) of n calls queued back to back between ONE pair of HIP events: what a camera loop sees.  (Per-call
    event pairs add two hipEventRecord calls per iteration, which on this runtime cost more host time than a 10-us kernel.)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


class Workload:
    """One dtype / interpolation variant of a batched warp resident on one GPU."""

    def __init__(self, args, dtype, interp_name, rank, dev, planar=False, homography=None, device_frames=False, verdicts=False):
        from bev_amd import warp
        from tests import workloads as wl
        self.planar = planar  # float32 channel planes out (bevwarp_warp_planar)
        self.warp, self.dtype, self.interp_name, self.args = warp, dtype, interp_name, args
        self.homography = homography or args.homography
        self.B = B = args.batch
        self.sw, self.sh = sw, sh = args.src
        self.dw, self.dh = dw, dh = args.dst
        C = self.C = 3
        tdtype, ndtype, esz = (torch.uint8, np.uint8, 1) if dtype == "u8" else (torch.float32, np.float32, 4)
        self.esz = esz
        self.interp = warp.INTER_LINEAR if interp_name == "linear" else warp.INTER_NEAREST
        base = (wl.keystone_H if self.homography == "keystone" else wl.synth_brno_H)(sw, sh, dw, dh)
        gidx = [rank * B + i for i in range(B)]  # global frame indices of this rank
        self.Ms = np.stack([wl.jitter_H(base, g) for g in gidx])
        out_esz = 4 if planar else esz
        self.set_bytes = B * (sh * sw * esz + dh * dw * out_esz) * C
        self.nsets = args.sets or max(2, int(np.ceil(1.1e9 / self.set_bytes)))
        self.frames_np = [] if device_frames else [wl.frame(g, sh, sw, ndtype) for g in gidx[:min(B, 8)]]
        self.srcs, self.dsts = [], []
        for s in range(self.nsets):
            if device_frames:  # seeded on the device: same statistics as wl.frame, no host generation (not compared with the oracle)
                gen = torch.Generator(device=dev).manual_seed(1234 + 1000 * rank + s)
                t = (torch.randint(0, 256, (B, sh, sw, C), dtype=torch.uint8, device=dev, generator=gen) if dtype == "u8"
                     else torch.rand((B, sh, sw, C), dtype=torch.float32, device=dev, generator=gen))
            else:
                t = torch.empty((B, sh, sw, C), dtype=tdtype, device=dev)
                for i in range(B):
                    if s == 0 and i < len(self.frames_np):
                        t[i] = torch.from_numpy(self.frames_np[i]).to(dev)
                    elif s == 0:
                        t[i] = t[i % len(self.frames_np)].flip(0) if (i // len(self.frames_np)) % 2 else t[i % len(self.frames_np)].flip(1)
                    else:  # other sets: same statistics, different bytes (cheap on-device generation)
                        t[i] = self.srcs[0][(i + s) % B].flip(0) if s % 2 else self.srcs[0][(i + s) % B].flip(1)
            self.srcs.append(t)
            self.dsts.append(torch.empty((B, C, dh, dw), dtype=torch.float32, device=dev) if planar else torch.empty((B, dh, dw, C), dtype=tdtype, device=dev))
        # The matrix tensor of the timed launches is the CALLER'S (a clone): every launch then classifies its tiles itself (bevwarp_warp).
        # With `verdicts` it is the one device_inverse owns: the Python entry fills a per-tile verdict table on the first (warm-up)
        # launch and the timed launches read it (bevwarp_warp_classes, ABI v7) -- a camera loop's steady state, reported as its own entry.
        self.verdicts = verdicts
        self.minv = warp.device_inverse(self.Ms, dev) if verdicts else warp.device_inverse(self.Ms, dev).clone()
        counts, touched = warp.footprint((sh, sw), self.Ms, (dw, dh), flags=self.interp, device=dev)
        self.footprint_px = int(counts.sum().item())
        # The same footprint at the granularity memory is fetched in: distinct 64-byte sectors / 128-byte lines of the source
        # frames that hold at least one touched pixel.  Taps of a strongly minified far field lie 3-8 pixels apart, so the bytes
        # that MUST cross the HBM interface exceed the touched pixels' own: `traffic` is to be read against this figure too.
        row_bytes = sw * C * esz
        self.sector_bytes = {}
        if row_bytes % 128 == 0:
            bm = touched.view(-1, sw, 1).expand(-1, sw, C * esz).reshape(-1, row_bytes)  # byte mask, one row per source row
            for g in (64, 128):
                self.sector_bytes[g] = int(bm.reshape(-1, row_bytes // g, g).any(dim=2).sum().item()) * g
            del bm
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
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()  # torch.cuda.synchronize + process-group barrier + synchronize: every rank starts its K steps together
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record()
            self.step(i)
            ev[i][1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0  # this rank's K steps; the job's time is the MAX over ranks (taken by the caller)
        barrier()
        launch_ms = np.array([a.elapsed_time(b) for a, b in ev])
        return elapsed, launch_ms

    def roofline(self, launch_ms, ceiling=None, sclk_mhz=None):
        kernel_s = float(launch_ms.mean()) / 1e3
        achieved = self.algo_bytes / kernel_s / 1e9
        rec = None if (self.planar or self.verdicts) else load_profile(self.dtype, self.interp_name, self.homography, self.args)  # (the committed profiles are of the plain launches)
        r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBS, 4),
             "traffic": None if rec is None else rec.get("hbm_bytes_per_launch"),
             "kernel": "warp_rows<%s,3,%s>%s" % ("uint8" if self.esz == 1 else "float", self.interp_name, " -> float32 planes" if self.planar else ""),
             "algorithmic_bytes_per_launch": self.algo_bytes, "footprint_px_per_launch": self.footprint_px,
             "footprint_64B_sector_bytes": self.sector_bytes.get(64), "footprint_128B_line_bytes": self.sector_bytes.get(128),
             "min_hbm_bytes_at_64B_granularity": (self.sector_bytes[64] + self.algo_bytes - self.footprint_px * self.C * self.esz) if self.sector_bytes else None,
             "kernel_ms_mean": round(kernel_s * 1e3, 4), "kernel_ms_min": round(float(launch_ms.min()), 4),
             "kernel_mpix_per_s": round(self.B * self.dw * self.dh / 1e6 / kernel_s, 1)}
        if ceiling is not None and "plain_loads_plain_stores" in ceiling:
            r["measured_ceiling_gbs"] = ceiling["plain_loads_plain_stores"]["gbs"]  # the warp's own access kinds
            r["measured_ceiling_nt_nt_gbs"] = ceiling["nt_loads_nt_stores"]["gbs"]
            r["frac_of_measured"] = round(achieved / ceiling["plain_loads_plain_stores"]["gbs"], 4)
            r["measured_ceiling"] = ceiling
        elif ceiling is not None:
            r["measured_ceiling"] = ceiling
        if sclk_mhz is not None:
            r["sclk_mhz"] = sclk_mhz
        sq = None if rec is None else rec.get("sq")
        if sq and sq.get("SQ_INSTS_VALU"):
            # vector-ALU accounting from the committed SQ pass (per launch): instructions per 256-px (u8) / 128-px (f32) wave-row,
            # the time they take to ISSUE on 1024 SIMDs at 4 cycles each, and the share of the kernel's duration a SIMD's VALU is busy
            clk = (sclk_mhz or rec.get("sclk_mhz") or 2400.0) * 1e6
            wave_rows = self.B * self.dh * ((self.dw + (255 if self.esz == 1 else 127)) // (256 if self.esz == 1 else 128))
            issue_s = sq["SQ_INSTS_VALU"] * 4.0 / 1024.0 / clk
            prof_s = rec.get("kernel_avg_ns_profiled", kernel_s * 1e9) * 1e-9
            busy = sq.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * prof_s * clk)
            r["roofline_valu"] = {"insts_per_launch": sq["SQ_INSTS_VALU"], "insts_per_wave_row": round(sq["SQ_INSTS_VALU"] / wave_rows, 1),
                                  "issue_us": round(issue_s * 1e6, 1), "busy_frac": round(busy, 3), "clock_mhz_used": round(clk / 1e6, 0),
                                  "kernel_us_profiled": round(prof_s * 1e6, 2), "source": rec.get("source")}
            hbm_share = (rec.get("hbm_bytes_per_launch") or self.algo_bytes) / prof_s / 1e9 / (r.get("measured_ceiling_gbs") or 5500.0)
            if busy > hbm_share:  # the VALU is busier than the memory system is full: name it
                r["bound"] = "valu-issue"
            r["hbm_share_of_streaming_ceiling"] = round(hbm_share, 3)
            if max(busy, hbm_share) < 0.9:  # neither saturated: say which is the busier and what else the time goes to
                r["bound_detail"] = ("VALU busy %.2f, HBM traffic at %.2f of the measured streaming rate: neither saturated -- %s" %
                                     (busy, hbm_share, "vector-instruction issue is the busiest unit (pair loads halved the gathers; the rest is per-tile set-up and waits)"
                                      if busy > hbm_share else "the gather path (instructions in flight), not a throughput limit"))
        return r


def oracle_check(w, frames=(0, -1)):
    """The GPU output of buffer set 0 against the CPU oracle on the same bytes, for a couple of frames of the batch (the variants'
    frames are generated on the device, so the source is read back first).  Run AFTER the timed region; checker only."""
    from oracle import cpu_oracle as co
    ok = True
    for f in frames:
        f = f % w.B
        src = w.srcs[0][f].cpu().numpy()
        exp = co.warp_perspective(src, w.Ms[f], (w.dw, w.dh), w.interp, nthreads=host_cores())
        got = w.dsts[0][f].cpu().numpy()
        if w.planar:  # float32 planes: float(v) * (1 / 255) + 0, float32 multiply then add (bev_amd.warp.warp_to_planar's defaults)
            exp = (exp.astype(np.float32) * np.float32(1.0 / 255.0) + np.float32(0.0)).transpose(2, 0, 1)
        ok = ok and bool(np.array_equal(got, exp))
    return ok


def variant_line(w, steps, warmup, barrier, label=None, probe=False, check=True):
    ceiling = measured_ceiling(w) if probe else None
    el, lm = w.run(steps, warmup, barrier)
    sclk = None
    if probe:  # the clock THIS variant's kernel holds (the vector-ALU issue time is priced with it): 8-bit ~2.05 GHz, float ~1.7-2.0
        try:
            sclk = kernel_clock(w, launches=150)
        except Exception as e:  # a diagnostic must not lose the line
            print("kernel_clock failed: %s: %s" % (type(e).__name__, e), file=sys.stderr)
    line = {"dtype": label or w.dtype, "interp": w.interp_name, "homography": w.homography,
            "value": round(w.B * w.dw * w.dh * steps / 1e6 / el, 1), "unit": "Mpix/s", "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
            "roofline": w.roofline(lm, ceiling=ceiling, sclk_mhz=sclk)}
    if check:
        line["matches_oracle"] = oracle_check(w)
    return line


# ---- the other configs of BASELINE.json, bounded (N = 1) -------------------------------------------------------------------
def config0(dev):
    """configs[0]: ONE 1280x720 uint8 frame -> 512x512 BEV, bilinear, synth-brno homography (the canonical CPU row)."""
    from bev_amd import warp
    from bev_amd.pipeline import FramePipeline
    from oracle import cpu_oracle as co
    from tests import workloads as wl
    sw, sh, dw, dh = 1280, 720, 512, 512
    M = wl.synth_brno_H(sw, sh, dw, dh)
    img = wl.frame(0, sh, sw, np.uint8)
    mpix = dw * dh / 1e6
    src, out = torch.from_numpy(img).to(dev), torch.empty((dh, dw, 3), dtype=torch.uint8, device=dev)
    minv = warp.device_inverse(M, dev)
    call = lambda: warp.warp_perspective(src, None, (dw, dh), out=out, M_inv_device=minv)  # noqa: E731
    t = event_times(call, 200, 20)
    b2b = back_to_back_us(call)
    t0 = time.perf_counter()
    for _ in range(2000):
        call()
    host_us = (time.perf_counter() - t0) / 2000 * 1e6  # (the queue back-pressures once it is full: an upper bound of the Python entry's cost)
    torch.cuda.synchronize()
    exp = co.warp_perspective(img, M, (dw, dh), 1, nthreads=1)
    res = {"workload": "one 1280x720x3 uint8 frame -> 512x512 BEV, bilinear, synth-brno homography",
           "gpu_resident": {"us_median": round(b2b, 2), "us_per_call_event_pairs_median": round(float(np.median(t)) * 1e6, 2),
                            "us_min_event_pair": round(float(t.min()) * 1e6, 2), "host_us_per_call": round(host_us, 2),
                            "what": "bev_amd.warp.warp_perspective (the Python entry, validated-launch cache) called back to back; us_median = "
                                    "500 calls between one pair of HIP events / 500",
                            "Mpix_per_s": round(mpix / (b2b * 1e-6), 1), "matches_oracle": bool(np.array_equal(out.cpu().numpy(), exp))}}
    hs = []
    for _ in range(25):
        t0 = time.perf_counter()
        warp.warpPerspective(img, M, (dw, dh))
        hs.append(time.perf_counter() - t0)
    res["gpu_pcie_inclusive_serial"] = {"ms_median": round(float(np.median(hs[5:])) * 1e3, 4), "Mpix_per_s": round(mpix / float(np.median(hs[5:])), 1),
                                        "what": "bev.warp.warpPerspective: pageable numpy frame up, BEV frame down, one call at a time"}
    cores = host_cores()
    for label, nt in (("cpu_oracle_1_thread", 1), ("cpu_oracle_all_cores", cores)):
        ts = []
        for _ in range(55):
            t0 = time.perf_counter()
            co.warp_perspective(img, M, (dw, dh), 1, nthreads=nt)
            ts.append(time.perf_counter() - t0)
        res[label] = {"ms_median": round(float(np.median(ts[5:])) * 1e3, 4), "Mpix_per_s": round(mpix / float(np.median(ts[5:])), 2), "threads": nt, "runs": 50}
    cv = cv2_leg([img], [M], (dw, dh), 1, [exp])
    res["cv2_opportunistic"] = cv if cv is not None else "cv2 not importable on this box"
    return res


def config2(dev):
    """configs[2]: 1e7 (u, v) points through a 3x3 H (bev.rbox.pts_world_bev), float32 and float64."""
    from bev_amd.points import project_points
    H = np.array([[0.02, -0.001, -3.0], [0.0004, 0.05, -20.0], [1e-5, 0.0009, 0.4]])
    N = 10_000_000
    res = {}
    for dt, tdt, esz in (("f32", torch.float32, 4), ("f64", torch.float64, 8)):
        nbuf = max(2, int(np.ceil(600e6 / (N * 2 * esz * 2))))  # rotate past the 256 MB Infinity Cache
        gen = torch.Generator(device=dev).manual_seed(7)
        ins = [(torch.rand((N, 2), dtype=torch.float64, device=dev, generator=gen) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)).to(tdt)
               for _ in range(nbuf)]
        outs = [torch.empty_like(x) for x in ins]
        k = [0]

        def step():
            project_points(ins[k[0] % nbuf], H, out=outs[k[0] % nbuf])
            k[0] += 1

        t = event_times(step, 300, 10)
        nbytes = N * 2 * esz * 2
        # checker: the first and last 100,000 points of buffer 0 against the oracle (float64: 1e-13 relative, the bar of tests/test_gpu_geom.py;
        # float32: the same float64 arithmetic rounded once, bit for bit)
        from oracle import cpu_oracle as co
        sl = np.r_[0:100_000, N - 100_000:N]
        exp, got = co.project_points(ins[0].cpu().numpy()[sl], H), outs[0].cpu().numpy()[sl]
        ok = bool(np.array_equal(got, exp)) if esz == 4 else bool(np.allclose(got, exp, rtol=1e-13, atol=0))
        res[dt] = {"us_mean": round(float(t.mean()) * 1e6, 2), "us_min": round(float(t.min()) * 1e6, 2), "Gpts_per_s": round(N / float(t.mean()) / 1e9, 2),
                   "matches_oracle": ok,
                   "roofline": {"bound": "hbm", "achieved": round(nbytes / float(t.mean()) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(nbytes / float(t.mean()) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                "kernel": "project_points_kernel<%s,2>" % ("float" if esz == 4 else "double"), "algorithmic_bytes_per_launch": nbytes}}
        del ins, outs
        torch.cuda.empty_cache()
    res["workload"] = "1e7 (u, v) points through a 3x3 H, dehomogenised; bytes = N x 2 coordinates x (in + out)"
    return res


def config3(args, dev, barrier):
    """configs[3]: the per-GPU shard, 32 x (3840x2160 -> 2048x2048), uint8 and float32 bilinear (device-generated frames)."""
    a3 = argparse.Namespace(**vars(args))
    a3.src, a3.dst, a3.batch, a3.sets, a3.homography = [3840, 2160], [2048, 2048], 32, 2, "keystone"
    res = {"workload": "32 x (3840x2160x3 -> 2048x2048x3) bilinear, keystone: one GPU's shard of the 256-frame, 8-GPU config"}
    for dt in ("u8", "f32"):
        w = Workload(a3, dt, "linear", 0, dev, device_frames=True)
        res[dt] = variant_line(w, args.leg_steps, 5, barrier, check=True)
        del w
        torch.cuda.empty_cache()
    return res


def config4(dev):
    """configs[4]: one camera frame's step -- 1080p -> 1024^2 uint8 bilinear warp + the tracker launch on 512 x 512 boxes."""
    from bev_amd import warp
    from bev_amd.iou import rbox_iou
    from bev_amd.tracker_geom import tracker_geometry_step
    from tests import workloads as wl
    rng = np.random.default_rng(11)
    M = wl.synth_brno_H(1920, 1080, 1024, 1024)
    frames = [torch.from_numpy(wl.frame(i, 1080, 1920, np.uint8)).to(dev) for i in range(4)]
    outs = [torch.empty((1024, 1024, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
    minv = warp.device_inverse(M, dev)
    H_world_bev = np.array([[0.0, 0.0625, -10.0], [-0.0625, 0.0, 40.0], [0, 0, 1.0]])
    H_img_world = np.linalg.inv(np.array([[0.02, -0.001, -3.0], [0.0004, 0.05, -20.0], [1e-5, 0.0009, 0.4]]))
    dets = torch.from_numpy(np.column_stack([rng.uniform(0, 1024, (512, 2)), rng.uniform(25, 35, 512), rng.uniform(56, 96, 512), rng.uniform(-np.pi, np.pi, 512)])).to(dev)
    trks = torch.from_numpy(np.column_stack([rng.uniform(-10, 54, 512), rng.uniform(-24, 40, 512), rng.uniform(1.6, 2.2, 512), rng.uniform(3.5, 6, 512),
                                             rng.uniform(-np.pi, np.pi, 512)])).to(dev)
    buf = tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world)
    k = [0]

    def step():
        i = k[0] % 4
        warp.warp_perspective(frames[i], None, (1024, 1024), out=outs[i], M_inv_device=minv)
        tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world, out=buf)
        k[0] += 1

    te = event_times(step, 300, 20)
    tt = event_times(lambda: tracker_geometry_step(dets, trks, H_world_bev, 0.3, H_img_world, out=buf), 300, 20)
    iou_out = torch.empty((512, 512), dtype=torch.float64, device=dev)
    dets_world = buf["dets_world"].clone()  # (the IoU matrix alone: the same pairs as the step scores)
    ti = event_times(lambda: rbox_iou(dets_world, trks, out=iou_out), 300, 20)
    from oracle import cpu_oracle as co  # checker, after the timed launches
    ok = bool(np.array_equal(outs[(k[0] - 1) % 4].cpu().numpy(), co.warp_perspective(wl.frame((k[0] - 1) % 4, 1080, 1920, np.uint8), M, (1024, 1024), 1, nthreads=host_cores())))
    ok = ok and bool(np.allclose(buf["iou"].cpu().numpy(), co.rbox_iou(buf["dets_world"].cpu().numpy(), trks.cpu().numpy()[:, :5]), rtol=0, atol=1e-12))
    res = {"matches_oracle": ok, "workload": "one 1920x1080 uint8 frame -> 1024x1024 BEV (bilinear, synth-brno) + bevwarp_tracker_step on 512 detections x 512 tracks (float64)",
           "step_eager_us": round(float(te.mean()) * 1e6, 1), "step_eager_us_min": round(float(te.min()) * 1e6, 1),
           "tracker_launch_us": round(float(tt.mean()) * 1e6, 1), "rbox_iou_512x512_us": round(float(ti.mean()) * 1e6, 1)}
    return res  # (a hipGraph replay of these two launches measures SLOWER than issuing them: tools/bench_geom.py, tools/graphed_step.py)


def composite_config(dev):
    """composite_bev_img (bev/tool/compo.py:26-49) -- one launch against three warps + the blend launch."""
    from bev_amd import warp
    from bev_amd.compo import composite_bev_img, composite_reg_img
    from bev_amd.homo import homo_from_KRt
    from tests import workloads as wl
    K = np.array([[1200.0, 0, 959.5], [0, 1190.0, 539.5], [0, 0, 1.0]])
    c, s_ = np.cos(0.9), np.sin(0.9)
    RT = np.array([[1, 0, 0, 0.0], [0, c, -s_, 2.0], [0, s_, c, 14.0], [0, 0, 0, 1.0]])
    H_world2bev = np.array([[0.0, 24.0, 512.0], [-24.0, 0.0, 900.0], [0.0, 0.0, 1.0]])
    H_img2world_fix = np.linalg.inv(homo_from_KRt(K, Rt_homo=RT)) @ np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1]])
    bg, fg, mask = (torch.from_numpy(wl.frame(i, 1080, 1920, np.uint8)).to(dev) for i in (0, 1, 2))
    Hb_ = H_world2bev.dot(H_img2world_fix)
    one = event_times(lambda: composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, K, RT, 1024, 1024), 100, 10)
    one_b2b = back_to_back_us(lambda: composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, K, RT, 1024, 1024), 200, 10)
    warp_b2b = back_to_back_us(lambda: warp.warp_perspective(bg, Hb_, (1024, 1024)), 200, 10)
    Hb = H_world2bev.dot(H_img2world_fix)
    Hc = H_world2bev.dot(np.linalg.inv(homo_from_KRt(K, Rt_homo=RT)))

    def three():
        return composite_reg_img(warp.warp_perspective(bg, Hb, (1024, 1024)), warp.warp_perspective(fg, Hc, (1024, 1024)), warp.warp_perspective(mask, Hc, (1024, 1024)))

    thr = event_times(three, 100, 10)
    same = bool(torch.equal(three(), composite_bev_img(bg, fg, mask, H_world2bev, H_img2world_fix, K, RT, 1024, 1024)[0]))
    return {"workload": "composite_bev_img: 1080p background + 1080p foreground + mask -> one 1024x1024 uint8 composite (device resident)",
            "one_launch_us": round(float(np.median(one)) * 1e6, 1), "three_warps_plus_blend_us": round(float(np.median(thr)) * 1e6, 1),
            "one_launch_back_to_back_us": round(one_b2b, 1), "one_u8_warp_same_destination_back_to_back_us": round(warp_b2b, 1),
            "kernel": "warp_rows<uint8,3,linear,NSRC=3> (12 waves per workgroup; rocprofv3 kernel time: profiles/r04_geom_rocprofv3_summary.txt)",
            "one_launch_equals_three_warps_plus_blend": same}


def small_branch_config(dev):
    """the reference's "small" branch (vis_homo.py:73-78,90-91) on the headline's batch -- 32 x 1080p uint8 frames resized to 852 x 480
    (cv2.resize, INTER_LINEAR) and the small frames warped to 1024^2 (the reference's two steps, its pixels) against the fused one-pass form
    (the resize folded into the homography: warp_perspective_resized).  HIP-event time per launch, source sets rotated past the Infinity Cache;
    both forms checked against the oracle (first and last frame of a set)."""
    from bev_amd import warp
    from bev_amd.resize import resize
    from tests import workloads as wl
    B, SH, SW, NW, NH, D, nset = 32, 1080, 1920, 852, 480, 1024, 4
    sets = [torch.from_numpy(np.stack([wl.frame(32 * s_ + i, SH, SW, np.uint8) for i in range(B)])).to(dev) for s_ in range(nset)]
    small = [torch.empty((B, NH, NW, 3), dtype=torch.uint8, device=dev) for _ in range(nset)]
    outs = [torch.empty((B, D, D, 3), dtype=torch.uint8, device=dev) for _ in range(nset)]
    M_small = np.stack([wl.jitter_H(wl.keystone_H(NW, NH, D, D), i) for i in range(B)])
    S = warp.resize_matrix((SW, SH), (NW, NH), False)
    M_fused = np.stack([m @ S for m in M_small])
    minv_small, minv_fused = warp.device_inverse(M_small, dev), warp.device_inverse(M_fused, dev)
    k = [0]

    def step(do_resize, do_small, do_fused):
        def fn():
            i = k[0] % nset
            if do_resize:
                resize(sets[i], (NW, NH), out=small[i])
            if do_small:
                warp.warp_perspective(small[i], None, (D, D), out=outs[i], M_inv_device=minv_small)
            if do_fused:
                warp.warp_perspective(sets[i], None, (D, D), out=outs[i], M_inv_device=minv_fused)
            k[0] += 1
        return fn

    t_resize = event_times(step(True, False, False), 100, 10)
    t_two = event_times(step(True, True, False), 100, 10)
    i = (k[0] - 1) % nset
    two_np = outs[i][[0, B - 1]].cpu().numpy()
    t_fused = event_times(step(False, False, True), 100, 10)
    j = (k[0] - 1) % nset
    fused_np = outs[j][[0, B - 1]].cpu().numpy()
    from oracle import cpu_oracle as co  # checker, after the timed launches
    ok_two = ok_fused = True
    for n_, f in enumerate((0, B - 1)):
        src_i, src_j = wl.frame(32 * i + f, SH, SW, np.uint8), wl.frame(32 * j + f, SH, SW, np.uint8)
        ok_two = ok_two and bool(np.array_equal(two_np[n_], co.warp_perspective(co.resize_linear_u8(src_i, (NW, NH)), M_small[f], (D, D), 1, nthreads=host_cores())))
        ok_fused = ok_fused and bool(np.array_equal(fused_np[n_], co.warp_perspective(src_j, M_fused[f], (D, D), 1, nthreads=host_cores())))
    rbytes = B * (SH * SW + NH * NW) * 3
    return {"workload": "32 x 1920x1080 uint8 -> resize 852x480 -> warp 1024x1024 (two steps, the reference's pixels) vs one fused pass",
            "resize_us": round(float(t_resize.mean()) * 1e6, 1), "resize_frac_of_8TBs": round(rbytes / float(t_resize.mean()) / 8e12, 3),
            "two_step_us": round(float(t_two.mean()) * 1e6, 1), "fused_us": round(float(t_fused.mean()) * 1e6, 1),
            "two_step_matches_oracle": ok_two, "fused_matches_oracle": ok_fused, "matches_oracle": ok_two and ok_fused,
            "kernels": "resize_linear_u8_px4_kernel<3> + warp_rows<uint8,3,linear> | warp_rows<uint8,3,linear>"}


def pipeline_config(dev):
    """PCIe-inclusive frames/s of 1080p -> 1024^2 uint8: one call at a time against the three-stream pipeline (never `value`)."""
    from bev_amd import warp
    from bev_amd.pipeline import FramePipeline
    from tests import workloads as wl
    M = wl.keystone_H(1920, 1080, 1024, 1024)
    img = wl.frame(0, 1080, 1920, np.uint8)
    for _ in range(3):
        warp.warpPerspective(img, M, (1024, 1024))
    t0 = time.perf_counter()
    for _ in range(20):
        warp.warpPerspective(img, M, (1024, 1024))
    serial = (time.perf_counter() - t0) / 20
    res = {}
    for label, zc in (("pipelined", True), ("pipelined_copy_down", False)):
        res[label] = _pipeline_rate(FramePipeline((1080, 1920), 3, M, (1024, 1024), depth=3, zero_copy_out=zc), img)
    piped = res["pipelined"]
    return {"workload": "1920x1080x3 uint8 host frames -> 1024x1024 BEV frames on the host (PCIe both ways)", "serial_ms_per_frame": round(serial * 1e3, 4),
            "pipelined_ms_per_frame": round(piped * 1e3, 4), "pipelined_copy_down_ms_per_frame": round(res["pipelined_copy_down"] * 1e3, 4),
            "speedup": round(serial / piped, 2), "pipelined_frames_per_s": round(1 / piped, 1),
            "what": "bev.warp.warpPerspective per frame vs bev_amd.pipeline.FramePipeline (pinned ring, depth 3; H2D and the warp on two streams, the "
                    "kernel storing the BEV frame straight into the pinned host slot; copy_down: a device frame and a D2H copy on a third stream)"}


def _pipeline_rate(pipe, img):
    n = 200
    for i in range(3):  # fill the slots once: afterwards the "decoder" finds its frame already in pinned memory (zero-copy ingest)
        pipe.next_input()[...] = img
        pipe.commit()
    for _ in range(3):
        pipe.result()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for i in range(n):
        pipe.next_input()
        pipe.commit()
        if pipe.ready() >= 3:
            pipe.result()
            done += 1
    while pipe.ready():
        pipe.result()
        done += 1
    return (time.perf_counter() - t0) / n


def summary_of(result):
    """{entry: [ms per step (kernel mean), fraction of 8 TB/s, HBM traffic / algorithmic bytes, GPU output == oracle]}; null where an
    entry has no such figure (latency-bound configs have no roofline fraction; traffic exists for the profiled workloads only)."""
    def r4(v):
        return None if v is None else float("%.4g" % v)

    def of_line(line, ok=None):
        rf = line["roofline"]
        tr = rf.get("traffic")
        return [r4(rf["kernel_ms_mean"]), r4(rf["frac"]), None if not tr else r4(tr / rf["algorithmic_bytes_per_launch"]), line.get("matches_oracle", ok)]

    hom = {"keystone": "key", "brno": "brno"}
    cb = result.get("cpu_baseline") or {}
    head = dict(result, matches_oracle=cb.get("gpu_output_matches_oracle"))
    out = {"%s_%s_%s" % (result["dtype"], "lin" if "linear" in result["config"]["workload"] else "near",
                         "brno" if "brno" in result["config"]["workload"] else "key"): of_line(head)}
    for v in result.get("variants", []):
        name = "%s_%s_%s" % ("u8_planar" if "planar" in v["dtype"] else v["dtype"], "lin" if v["interp"] == "linear" else "near", hom[v["homography"]])
        out[name + ("_tbl" if v.get("verdict_table") else "")] = of_line(v)
        if v.get("interleaved_us"):  # [plain us, table us] of launches interleaved on the same buffers
            out[name + "_tbl_ab"] = [v["interleaved_us"].get("plain"), v["interleaved_us"].get("table")]
    cfg = result.get("configs", {})

    def us(d, key):
        return None if d.get(key) is None else r4(d[key] * 1e-3)

    c0 = cfg.get("configs[0]", {}).get("gpu_resident")
    if c0:
        out["c0_720p"] = [us(c0, "us_median"), None, None, c0.get("matches_oracle")]
    for dt in ("f32", "f64"):
        c2 = cfg.get("configs[2]", {}).get(dt)
        if c2:
            out["c2_pts_" + dt] = [us(c2, "us_mean"), r4(c2["roofline"]["frac"]), None, c2.get("matches_oracle")]
    for dt in ("u8", "f32"):
        c3 = cfg.get("configs[3]", {}).get(dt)
        if c3:
            out["c3_4k_" + dt] = of_line(c3)
    c4 = cfg.get("configs[4]", {})
    if "step_eager_us" in c4:
        out["c4_step"] = [us(c4, "step_eager_us"), None, None, c4.get("matches_oracle")]
        out["c4_tracker"] = [us(c4, "tracker_launch_us"), None, None, None]
        out["c4_iou"] = [us(c4, "rbox_iou_512x512_us"), None, None, None]
    f3 = cfg.get("f3_composite", {})
    if "one_launch_us" in f3:
        out["f3_comp"] = [us(f3, "one_launch_back_to_back_us"), None, None, f3.get("one_launch_equals_three_warps_plus_blend")]
    f1 = cfg.get("f1_small_branch", {})
    if "two_step_us" in f1:
        out["f1_two_step"] = [us(f1, "two_step_us"), None, None, f1.get("two_step_matches_oracle")]
        out["f1_fused"] = [us(f1, "fused_us"), None, None, f1.get("fused_matches_oracle")]
    pc = cfg.get("pcie_pipeline", {})
    if "pipelined_ms_per_frame" in pc:
        out["pcie_frame"] = [r4(pc["pipelined_ms_per_frame"]), None, None, None]
    if cb:
        out["cpu_mpix_s"] = [r4(cb["value"]), cb.get("cores")]
    for k in [k for k, v in cfg.items() if "error" in v]:
        out[k] = "error"
    return out


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
    # Rehearsal knob for a 1-GPU box (never needed on a real node): BEV_BENCH_SAME_DEVICE=1 puts every rank on device 0
    # (tests/test_gpu_shard.py runs the whole N = 2 bench that way).  BEV_BENCH_BACKEND overrides the control-plane backend.
    dev_index = 0 if os.environ.get("BEV_BENCH_SAME_DEVICE") == "1" else local_rank
    backend = os.environ.get("BEV_BENCH_BACKEND", shard.CONTROL_BACKEND)  # gloo: the group carries a barrier and three scalar reductions, on CPU tensors
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    shard.init(backend=backend, device=dev)  # no-op for one process; no RCCL communicator is created (north_star: no collectives)

    big = args.config == 3
    main_wl = Workload(args, args.dtype, args.interp, rank, dev, device_frames=big and args.no_cpu_baseline)
    ceiling = measured_ceiling(main_wl) if (rank == 0 and not args.no_probe) else None  # same process, same buffers, before the headline
    elapsed_local, launch_ms = main_wl.run(args.steps, args.warmup, shard.barrier)
    elapsed = shard.max_over_ranks(elapsed_local)
    kernel_ms_max = shard.max_over_ranks(float(launch_ms.mean()))  # slowest rank's mean launch duration (HIP events)
    kernel_ms_min = -shard.max_over_ranks(-float(launch_ms.mean()))
    sclk = None
    if rank == 0 and not args.no_probe:
        try:
            sclk = kernel_clock(main_wl)
        except Exception as e:  # a diagnostic must not lose the headline
            sclk = None
            print("kernel_clock failed: %s: %s" % (type(e).__name__, e), file=sys.stderr)
    shard.barrier()

    if rank == 0:
        B, dw, dh, sw, sh = main_wl.B, main_wl.dw, main_wl.dh, main_wl.sw, main_wl.sh
        mpix_total = world * B * dw * dh * args.steps / 1e6
        result = {
            "metric": "BEV Mpix/s, %s warp; achieved HBM GB/s vs peak" % ("1080p->1024^2" if not big else "2160p->2048^2"),
            "value": round(mpix_total / elapsed, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[%d]: batch=%d %dx%dx3 -> %dx%dx3 %s warp per GPU, %s pixels, %s homography (+-2 px per-frame jitter)"
                                   % (args.config, B, sw, sh, dw, dh, args.interp, "float32" if args.dtype == "f32" else "uint8", args.homography),
                       "frames_per_gpu": B, "sharding": "frames split by rank, no collectives",
                       "buffer_sets_rotated": main_wl.nsets, "resident_bytes_per_gpu": main_wl.nsets * main_wl.set_bytes,
                       "kernel_source_sha": kernel_source_sha()},
            "roofline": main_wl.roofline(launch_ms, ceiling=ceiling, sclk_mhz=sclk),
            # N > 1: the wall figure above is the slowest rank's K steps; these say whether a sub-linear curve comes from the
            # kernels (a slow device) or from the host side (launch skew)
            "kernel_ms_mean": round(float(launch_ms.mean()), 4), "kernel_ms_max_over_ranks": round(kernel_ms_max, 4),
            "kernel_ms_min_over_ranks": round(kernel_ms_min, 4), "wall_ms_per_step_rank0": round(elapsed_local / args.steps * 1e3, 4),
        }
        if world == 1 and not args.no_cpu_baseline and main_wl.frames_np:
            gpu_out = main_wl.dsts[0][:len(main_wl.frames_np)].cpu().numpy()  # set 0 holds the seeded frames
            result["cpu_baseline"] = cpu_baseline(main_wl.frames_np, main_wl.Ms, (dw, dh), main_wl.interp, gpu_out, args.cpu_seconds)
    del main_wl
    torch.cuda.empty_cache()

    if world == 1 and not args.no_variants:
        variants = []
        n = max(args.leg_steps, args.steps)
        for dt, ip, planar, hom, tbl in (("u8", "linear", False, "keystone", False), ("u8", "nearest", False, "keystone", False), ("f32", "linear", False, "keystone", False),
                                         ("u8", "linear", True, "keystone", False), ("u8", "linear", False, "brno", False), ("f32", "linear", False, "brno", False),
                                         ("u8", "nearest", False, "brno", False), ("u8", "linear", False, "keystone", True)):
            if (dt, ip, hom) == (args.dtype, args.interp, args.homography) and not planar and not tbl:
                continue
            w = Workload(args, dt, ip, rank, dev, planar=planar, homography=hom, device_frames=True, verdicts=tbl)
            variants.append(variant_line(w, n, max(5, args.warmup // 2), shard.barrier, label=dt if not planar else "u8 -> f32 planar",
                                         probe=not args.no_probe and not planar and not tbl))
            if tbl:
                variants[-1]["verdict_table"] = "per-tile verdicts of these matrices filled by one untimed launch, read by the timed ones (bevwarp_warp_classes)"
                # entries of one run are minutes apart on a clock-managed chip: the table's own effect is read from launches INTERLEAVED on
                # this workload's buffers, order drawn at random (plain = a clone of the matrix tensor, which the Python entry keeps no table for)
                plain_m, rng_ab, t_ab = w.minv.clone(), np.random.default_rng(11), {"plain": [], "table": []}
                for r_ab in range(2 * 60 + 6):
                    which = "plain" if (r_ab < 6 and r_ab % 2) or (r_ab >= 6 and rng_ab.random() < 0.5) else "table"
                    k_ab = int(rng_ab.integers(w.nsets))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    w.warp.warp_perspective(w.srcs[k_ab], None, (w.dw, w.dh), flags=w.interp, out=w.dsts[k_ab], M_inv_device=plain_m if which == "plain" else w.minv)
                    e1.record()
                    e1.synchronize()
                    if r_ab >= 6:
                        t_ab[which].append(e0.elapsed_time(e1) * 1e3)
                variants[-1]["interleaved_us"] = {k_: round(float(np.median(v_)), 1) for k_, v_ in t_ab.items()}
            del w
            torch.cuda.empty_cache()
        result["variants"] = variants

    if world == 1 and not args.no_configs:
        cfg = {}
        for name, fn in (("configs[0]", lambda: config0(dev)), ("configs[2]", lambda: config2(dev)), ("configs[3]", lambda: config3(args, dev, shard.barrier)),
                         ("configs[4]", lambda: config4(dev)), ("f3_composite", lambda: composite_config(dev)), ("f1_small_branch", lambda: small_branch_config(dev)),
                         ("pcie_pipeline", lambda: pipeline_config(dev))):
            try:
                cfg[name] = fn()
            except Exception as e:  # a failing side measurement must not lose the headline line
                cfg[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        result["configs"] = cfg

    if rank == 0:
        if world == 1:
            result["summary"] = summary_of(result)  # LAST key: the tail of the line alone carries every number
        print(json.dumps(result), flush=True)
    shard.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
