#!/usr/bin/env python3
"""Ablation builds of the IoU / tracker-step kernels (measurement aid; the product source carries no ablation code).

    python tools/ablate_geom.py <name>=<spec>[+<spec>...] ...   ->  bev_amd/csrc/variants/<name>.so   (time with tools/time_tracker.py)

Specs patch a COPY of bev_amd/csrc/geom_kernels.hip:
    wg256      256 lanes (4 waves) per workgroup instead of 64: a quarter of the workgroups to dispatch
    wg128      128 lanes per workgroup
    noclip     (timing only) the contour sum returns after the corners and reciprocals: what is left is launch + loads + rejection + sincos + stores
    rejectall  (timing only) every pair ends at the rejection test: launch + loads + stores
    empty      (timing only) the scoring lanes store 0 without reading a box: the launch floor of this grid
    outfirst   the tracker step's output workgroups take the first rows of the grid instead of the last
    stamps     (diagnostic) cycle stamps of the contour lanes in rbox_iou_kernel, read by tools/iou_stamps.py
    libsincos  the library's sincos instead of the kernel's own reduced pair
    noout / nohead   (timing only) the output workgroups return at once / the detection's heading costs nothing
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bev_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-undefined-internal".split()


def patch(text, spec):
    def rep(old, new, count=1):
        nonlocal text
        assert text.count(old) >= 1, (spec, old[:70])
        text = text.replace(old, new, count)
    if spec in ("wg256", "wg128"):
        rep("constexpr int kIouThreads = 64;", "constexpr int kIouThreads = %s;" % spec[2:])
    elif spec == "noclip":
        rep("    double acc = 0.0;\n", "    double acc = 0.0;\n    if (hx != 1.2345e300) return (Cx[0] * Cy[1] + Cx[2] * Cy[3] + ix * jy + iy * jx) * 1e-300;\n")
    elif spec == "rejectall":
        rep("&& area > 0)) return 0.0;", "&& area > 0) || area != 1.2345e300) return 0.0;")
    elif spec == "empty":
        rep("    const T* pa = a + (int64_t)i * sa;\n", "    if (na != 123456789) { if (valid) out[(int64_t)i * nb + j] = (T)0; return; }\n    const T* pa = a + (int64_t)i * sa;\n")
        rep("    const T* pd = dets + (int64_t)i * sd;\n", "    if (n != 123456789) { if (valid) { iou[(int64_t)i * m + j] = (T)0; cand[(int64_t)i * m + j] = 0; } return; }\n"
            "    const T* pd = dets + (int64_t)i * sd;\n")
    elif spec == "outfirst":  # the output workgroups take the FIRST rows of the grid instead of the last
        rep("    if ((int)blockIdx.y >= n) {", "    const int out_rows = (n + kIouThreads - 1) / kIouThreads;\n    if ((int)blockIdx.y < out_rows) {")
        rep("        const int i = ((int)blockIdx.y - n) * kIouThreads + tid;", "        const int i = (int)blockIdx.y * kIouThreads + tid;")
        rep("    const int i = blockIdx.y;\n    const int j = blockIdx.x * kIouThreads + tid;", "    const int i = (int)blockIdx.y - out_rows;\n    const int j = blockIdx.x * kIouThreads + tid;")
    elif spec == "noout":  # (timing only) the output workgroups return at once
        rep("        if (blockIdx.x != 0 || i >= n) return;", "        if (blockIdx.x != 0 || i >= n || n != 123456789) return;")
    elif spec == "nohead":  # (timing only) the detection's heading is a constant: no sincos / rsq in front of the contour
        rep("        double sn, cs;\n        sincos(det_yaw, &sn, &cs);\n        const double tx", "        double sn = 0.6, cs = 0.8;\n        const double tx")
    elif spec == "stamps":  # s_memtime / s_memrealtime stamps of the lanes that integrate a contour (rbox_iou_kernel only): tools/iou_stamps.py reads them
        # one record per workgroup (plain stores: atomics on shared words cost more than the kernel); every stamp is pinned between the stages by a data
        # dependence through an empty asm (the value of s_memtime feeds it, the stage's input passes through it), or the scheduler moves the stamps
        rep("constexpr int kIouThreads = 64;", "constexpr int kIouThreads = 64;\n__device__ unsigned long long g_iou_stamps[8192 * 8];\n"
            "#define IOU_NOW() __builtin_amdgcn_s_memtime()")
        rep("__device__ __forceinline__ double pair_iou(bool valid, double acx, double acy, double aw, double ah, double ayaw, HeadingA&& heading_a, const double (&B)[5]) {\n",
            "__device__ __forceinline__ double pair_iou(bool valid, double acx, double acy, double aw, double ah, double ayaw, HeadingA&& heading_a, const double (&B)[5], unsigned long long st_start = 0) {\n"
            "    const unsigned long long st0 = IOU_NOW(), sr0 = __builtin_amdgcn_s_memrealtime();\n    asm volatile(\"\" : \"+v\"(acx) : \"s\"(st0), \"s\"(sr0));\n")
        rep("    double ca, sa, cb, sb;\n    sincos_pair(ayaw, B[4], sa, ca, sb, cb);", "    const unsigned long long st1 = IOU_NOW();\n    asm volatile(\"\" : \"+v\"(ayaw) : \"s\"(st1));\n"
            "    double ca, sa, cb, sb;\n    sincos_pair(ayaw, B[4], sa, ca, sb, cb);\n    asm volatile(\"\" ::\"v\"(sa), \"v\"(ca), \"v\"(sb), \"v\"(cb));\n"
            "    const unsigned long long st2 = IOU_NOW();\n    asm volatile(\"\" : \"+v\"(ca) : \"s\"(st2));")
        rep("    if (inter < 1e-14 * area || B[2] * B[3] < 0) inter = 0.0;", "    asm volatile(\"\" ::\"v\"(inter));\n    const unsigned long long st3 = IOU_NOW(), sr3 = __builtin_amdgcn_s_memrealtime();\n"
            "    if ((unsigned)__builtin_ctzll(__ballot(1)) == (threadIdx.x & 63)) {  // first active lane of the wave\n"
            "        unsigned long long* rec = g_iou_stamps + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8;\n"
            "        rec[0] = 1; rec[1] = st1 - st0; rec[2] = st2 - st1; rec[3] = st3 - st2; rec[4] = st3 - st0; rec[5] = sr3 - sr0; rec[6] = st0 - st_start; rec[7] = __popcll(__ballot(1));\n"
            "    }\n"
            "    if (inter < 1e-14 * area || B[2] * B[3] < 0) inter = 0.0;")
        rep("    asm volatile(\"\" ::\"s\"(a), \"s\"(na), \"s\"(sa), \"s\"(b), \"s\"(nb), \"s\"(sb), \"s\"(out));\n",
            "    const unsigned long long st_start = IOU_NOW();\n    asm volatile(\"\" ::\"s\"(a), \"s\"(na), \"s\"(sa), \"s\"(b), \"s\"(nb), \"s\"(sb), \"s\"(out), \"s\"(st_start));\n")
        rep("    const double v = pair_iou(valid, A[0], A[1], A[2], A[3], A[4], [](double&, double&) {}, B);", "    const double v = pair_iou(valid, A[0], A[1], A[2], A[3], A[4], [](double&, double&) {}, B, st_start);")
        text += ("\nextern \"C\" int bevwarp_debug_iou_stamps(unsigned long long* out, int reset) {\n"
                 "    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bevwarp::g_iou_stamps), 8192 * 64) != hipSuccess) return -1;\n"
                 "    if (reset) { static unsigned long long z[8192 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(bevwarp::g_iou_stamps), z, 8192 * 64) != hipSuccess) return -1; }\n"
                 "    return 0;\n}\n")
    elif spec == "libsincos":  # the library's sincos for both angles, one after the other
        rep("    if (fabs(x0) < 1048576.0 && fabs(x1) < 1048576.0) {", "    if (fabs(x0) < 0.0 && fabs(x1) < 1048576.0) {")
    else:
        raise SystemExit("unknown spec %r" % spec)
    return text


def main():
    os.makedirs(os.path.join(CSRC, "variants"), exist_ok=True)
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    for arg in sys.argv[1:]:
        name, _, specs = arg.partition("=")
        text = open(os.path.join(CSRC, "geom_kernels.hip")).read()
        for s in (specs or name).split("+"):
            text = patch(text, s)
        d = "/tmp/ablate_geom_%s" % name
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "geom_kernels.hip"), "w") as f:
            f.write(text)
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(d, "geom_kernels.hip"),
                               "-o", os.path.join(d, "geom_kernels.o")])
        objs = [os.path.join(CSRC, o) for o in sorted(os.listdir(CSRC)) if o.endswith(".o") and o != "geom_kernels.o"]
        out = os.path.join(CSRC, "variants", name + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out, os.path.join(d, "geom_kernels.o")] + objs)
        print("built", out)


if __name__ == "__main__":
    main()
