#!/usr/bin/env python3
"""Ablation builds of the warp kernel (measurement aid; the product source carries no ablation code).

    python tools/ablate.py <name>=<spec>[+<spec>...] ...   ->  bev_amd/csrc/variants/<name>.so

Each spec patches a COPY of the kernel sources (bev_amd/csrc: warp_rows.h, rows_*.inc, coords.h, sample.h -- whichever file holds the
snippet):
    nostore   store_s keeps its operands alive and returns
    noload    issue_s fabricates taps from the offsets (no vector memory loads)
    noblend   finish_s xors the taps instead of blending
    stsmall   stores go to a few KB per frame (no HBM write traffic)
    ldsmall   taps come from the first 64 KB of the frame (cache hits)
    ldx2 / ldx1   (timing only) the aligned 12-byte tap windows fetched as 8 / 4 bytes: the texture path's cost per returned byte
    notie     no tie-window test in the coordinate chain
    ntload    float taps through non-temporal loads
    noedge    EDGE blocks cost what OUT blocks cost
    ownrow / ownblk   every tile forced to the unturned (row segments; edge tiles: blocks) / turned (patches) lane layout: the slant rule's A/B
    fillall / edgefill / infill   all / edge / interior tiles cost what outside tiles cost
    nostagger / revrows / lpt   dispatch-order experiments: no XCD stagger / a frame's tile rows bottom-up / an XCD walks all its frames tile row by tile row
    waves3 / ahead3 / ahead4   three waves per SIMD (<= 168 VGPRs) / three or four tap sets in flight
    waves5 / ahead1   five waves per SIMD (<= 96 VGPRs) / one tap set in flight in the straight-line bilinear tiles
    stage     interior row-affine tiles take the LDS-staged producer / consumer form (rows_staged.inc; off in the product)
    stagent   (with stage) the staged form's ring fills are non-temporal (aux = 2)
    pwfix / pws5   (with stage) the producer is always wave 3 / rotates with the dispatch order divided by the CUs of an XCD
    ntstore   the wide destination stores are non-temporal
    stsc1 / stsc01   the wide destination stores carry the sc1 / sc0 sc1 cache policy (write-through)
    nopair / allpair   no tile / every row-affine interior tile takes the pair loads (rows_tiles.inc: tile_pair); nosplit: no half-height tail workgroups
    noclass (+ w4x)   timing only, inset footprints: no tile classification (every tile an interior pair tile); w4x: exactly four waves per SIMD
    wg1 / wg2   workgroups of one / two waves instead of four (the host sizes tiles by rows_per_pass(): 6 / 12 rows of 8-bit pixels, 4 / 8 of float)
    pf<N>     tile prefetch: every wave of a full-height unturned interior tile touches its share of the tile's source sectors (N byte loads per lane) before its first taps
    ring16 / ring4   the staged form's ring holds 16 / 4 source rows instead of 8
Values stay live through `asm volatile` so that nothing upstream is dead code (guide, methodology rule 17)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bev_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-undefined-internal".split()


KERNEL_FILES = ("coords.h", "sample.h", "warp_rows.h", "rows_coords.inc", "rows_sample.inc", "rows_store.inc", "rows_tiles.inc", "rows_staged.inc", "rows_run.inc", "warp_kernels.h")
UNITS = ("warp_kernels", "warp_u8_linear", "warp_u8_nearest", "warp_f32_linear", "warp_f32_nearest", "warp_composite")


def apply_patch(files, path):
    """A unified diff against bev_amd/csrc (kept under tools/patches: experiments that are not product code), applied to the copies."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for n, t in files.items():
            open(os.path.join(d, n), "w").write(t)
        subprocess.check_call(["patch", "-s", "-p3", "-d", d, "-i", path])
        for n in files:
            files[n] = open(os.path.join(d, n)).read()


def patch(files, spec):
    """files: {name: text} of the kernel sources; the snippet is replaced (once) in the one file that holds it."""
    def rep(old, new):
        hits = [n for n, t in files.items() if old in t]
        assert len(hits) == 1, (spec, old[:60], hits)
        files[hits[0]] = files[hits[0]].replace(old, new, 1)
    if spec == "nostore":
        rep("    auto store_s = [&](auto own, int xs, int y, const uint4 (&out)[NQ]) __attribute__((always_inline)) {  // xs = first pixel of the segment / block\n",
            "    auto store_s = [&](auto own, int xs, int y, const uint4 (&out)[NQ]) __attribute__((always_inline)) {\n"
            "        asm volatile(\"\" ::\"v\"(out[0].x), \"v\"(out[0].y), \"v\"(out[0].z), \"v\"(out[0].w));\n        if (y != 12345678) return;\n")
    elif spec == "noload":
        rep("#pragma unroll\n        for (int j = 0; j < PPL; j++) {\n            const uint32_t off = S0[j];\n",
            "        for (int j = 0; j < PPL; j++)\n            for (int k = 0; k < WINB / 4; k++) t0[j].w[k] = S0[j] * (k + 3) + (uint32_t)(uintptr_t)b0, "
            "t1[j].w[k] = S0[j] ^ (0x9e3779b9u * (k + 1));\n        if (cls != 12345678) return;\n"
            "#pragma unroll\n        for (int j = 0; j < PPL; j++) {\n            const uint32_t off = S0[j];\n")
    elif spec == "noblend":
        rep("            blend_put(j, w0, w1, fx, fy);\n",
            "            wtr[64 * j + lane] = (w0[0] ^ w1[0] ^ w0[NEED - 1] ^ w1[NEED - 1]) + fx + fy;\n")
    elif spec == "stsmall":
        rep("            uint8_t* d = dframe + (int64_t)(y + st_row) * a.dst_rs + (int64_t)st_x * C;\n",
            "            uint8_t* d = dframe + (int64_t)((y + st_row) & 7) * a.dst_rs + (int64_t)(st_x & 255) * C;\n")
    elif spec in ("ldx2", "ldx1"):  # timing only (wrong pixels): the aligned tap windows fetched as 8 / 4 bytes instead of 12
        n = 8 if spec == "ldx2" else 4
        rep("                const uint32_t offa = off & ~3u;\n                __builtin_memcpy(&t0[j], b0 + offa, WINB);\n                __builtin_memcpy(&t1[j], b1 + offa, WINB);\n",
            "                const uint32_t offa = off & ~3u;\n                __builtin_memcpy(&t0[j], b0 + offa, %d);\n                __builtin_memcpy(&t1[j], b1 + offa, %d);\n" % (n, n))
    elif spec in ("cohload", "cohload2"):  # timing only: the 8 window gathers of a pass replaced by coalesced 16-byte loads of the source
        # row span the pass starts at (1.5 KB of ONE row: what a wave that keeps the other tap row staged would fetch; cohload2: both rows)
        both_rows = "1" if spec == "cohload2" else "0"
        rep("#pragma unroll\n        for (int j = 0; j < PPL; j++) {\n            const uint32_t off = S0[j];\n",
            "        if (f && kAligned) {\n"
            "            const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)S0[0]) & ~15u;\n"
            "            typedef uint32_t q4 __attribute__((ext_vector_type(4)));\n"
            "            q4 r0 = *reinterpret_cast<const q4*>(b0 + start + lane * 16), r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0}, r3 = {0, 0, 0, 0};\n"
            "            if (lane < 32) r1 = *reinterpret_cast<const q4*>(b0 + start + 1024 + lane * 16);\n"
            "            if (" + both_rows + ") { r2 = *reinterpret_cast<const q4*>(b1 + start + lane * 16); if (lane < 32) r3 = *reinterpret_cast<const q4*>(b1 + start + 1024 + lane * 16); }\n"
            "            for (int j = 0; j < PPL; j++)\n"
            "                for (int k = 0; k < WINB / 4; k++) t0[j].w[k] = S0[j] * (k + 3) + r0[k] + r1[(k + j) & 3] + r2[k], t1[j].w[k] = S0[j] ^ (0x9e3779b9u * (k + 1)) ^ r0[3] ^ r3[j & 3];\n"
            "            return;\n        }\n"
            "#pragma unroll\n        for (int j = 0; j < PPL; j++) {\n            const uint32_t off = S0[j];\n")
    elif spec in ("cohf32", "cohf32b"):  # timing only, float RGB: the 16 tap gathers of a pass replaced by coalesced loads of 2.75 KB of ONE source row (b: both rows)
        both_rows = "1" if spec == "cohf32b" else "0"
        rep("#pragma unroll\n        for (int j = 0; j < PPL; j++) {\n            const uint32_t off = S0[j];\n",
            "        if (f && sizeof(T) == 4 && C == 3 && INTERP == kLinear) {\n"
            "            const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)S0[0]) & ~15u;\n"
            "            typedef uint32_t q4 __attribute__((ext_vector_type(4)));\n"
            "            q4 r[6];\n"
            "            for (int i = 0; i < 6; i++) r[i] = q4{0, 0, 0, 0};\n"
            "            r[0] = *reinterpret_cast<const q4*>(b0 + start + lane * 16);\n"
            "            r[1] = *reinterpret_cast<const q4*>(b0 + start + 1024 + lane * 16);\n"
            "            if (lane < 48) r[2] = *reinterpret_cast<const q4*>(b0 + start + 2048 + lane * 16);\n"
            "            if (" + both_rows + ") { r[3] = *reinterpret_cast<const q4*>(b1 + start + lane * 16); r[4] = *reinterpret_cast<const q4*>(b1 + start + 1024 + lane * 16); if (lane < 48) r[5] = *reinterpret_cast<const q4*>(b1 + start + 2048 + lane * 16); }\n"
            "            for (int j = 0; j < PPL; j++)\n"
            "                for (int k = 0; k < LOADB / 4; k++) t0[j].w[k] = r[k % 3][k & 3] ^ r[(k + j) % 3][(k + 1) & 3], t1[j].w[k] = r[3 + k % 3][k & 3] ^ r[0][(k + j) & 3];\n"
            "            return;\n        }\n"
            "#pragma unroll\n        for (int j = 0; j < PPL; j++) {\n            const uint32_t off = S0[j];\n")
    elif spec == "ldsmall":
        rep("            const uint32_t off = S0[j];\n", "            const uint32_t off = S0[j] & 0xffffu;\n")
    elif spec == "ntload":  # float taps through non-temporal loads (streaming probe: nt loads + nt stores is the box's best mix)
        rep("                __builtin_memcpy(&t0[j], b0 + off, LOADB);\n                if (INTERP == kLinear) __builtin_memcpy(&t1[j], b1 + off, LOADB);\n",
            "                for (int k = 0; k < LOADB / 4; k++) {\n"
            "                    t0[j].w[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(b0 + off) + k);\n"
            "                    if (INTERP == kLinear) t1[j].w[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(b1 + off) + k);\n"
            "                }\n")
    elif spec == "noedge":  # EDGE passes cost what OUT passes cost (upper bound of what a cheaper guarded path can gain)
        rep("            else if (cls == kEdge)\n                edge_s(S1, S2);\n", "            else if (cls == kEdge)\n                fill_s();\n")
    elif spec == "notie":
        rep("            tie = min(tie, min(lx[j] & F::kTieMask, ly[j] & F::kTieMask));\n", "")
        rep("            for (int j = 0; j < PPL; j++) tie = min(tie, min(S1[j] & F::kTieMask, S2[j] & F::kTieMask));\n", "            for (int j = 0; j < PPL; j++) tie |= S1[j] >> 31;\n")
    elif spec == "ownrow":  # interior tiles: row segments whatever the slant
        rep("        if (!tile_affine) tile_slanted = fmaxf(edge_slant(0, 1), edge_slant(2, 3)) >", "        if (false) tile_slanted = fmaxf(edge_slant(0, 1), edge_slant(2, 3)) >")
    elif spec == "fillall":  # every tile costs what an outside tile costs: the launch + prologue + store floor
        rep("    if (tile_out) {  // every pixel of the tile is the border value", "    if (true) {")
    elif spec == "edgefill":  # tiles the frame's edge crosses cost what outside tiles cost
        rep("    if (tile_out) {  // every pixel of the tile is the border value", "    if (tile_out || !tile_in) {")
    elif spec == "infill":  # interior tiles cost what outside tiles cost
        rep("    if (tile_out) {  // every pixel of the tile is the border value", "    if (tile_out || tile_in) {")
    elif spec == "ownblk":  # interior tiles: blocks whatever the slant
        rep("        if (!tile_affine) tile_slanted = fmaxf(edge_slant(0, 1), edge_slant(2, 3)) >", "        if (true) tile_slanted = true || fmaxf(edge_slant(0, 1), edge_slant(2, 3)) >")
    elif spec == "ntstore":  # the wide destination stores non-temporal (rounds 2-3 measured them slower on every format)
        rep("    *p = v;\n}", "    __builtin_nontemporal_store(v, p);\n}")
    elif spec in ("pwfix", "pws5"):  # the producer's wave index: always wave 3 / rotating with the dispatch order divided by the CUs of an XCD
        rep("const int p_wave = (int)((blockIdx.x >> 3) & (uint32_t)(kWaves - 1));",
            "const int p_wave = 3;" if spec == "pwfix" else "const int p_wave = (int)((blockIdx.x >> 8) & (uint32_t)(kWaves - 1));")
    elif spec in ("stsc1", "stsc01"):  # the wide destination stores with the sc1 (write-through) / sc0 sc1 cache policy, as inline assembly
        pol = "sc1" if spec == "stsc1" else "sc0 sc1"
        rep("    *p = v;\n}", "    if constexpr (__builtin_vectorelements(V) == 4)\n        asm volatile(\"global_store_dwordx4 %%0, %%1, off %s\" ::\"v\"(p), \"v\"(v) : \"memory\");\n"
            "    else if constexpr (__builtin_vectorelements(V) == 3)  // (a 3-element vector is padded to 16 bytes: count elements, not bytes)\n"
            "        asm volatile(\"global_store_dwordx3 %%0, %%1, off %s\" ::\"v\"(p), \"v\"(v) : \"memory\");\n    else\n        *p = v;\n}" % (pol, pol))
    elif spec == "nostagger":  # every XCD starts at the first item of its run
        rep("    uint32_t in_run = seq + (blockIdx.x & 7u) * (uint32_t)a.stagger;", "    uint32_t in_run = seq;")
    elif spec == "revrows":  # a frame's tile rows from the bottom up
        rep("    const uint32_t ty = fast_div(t, a.tx_magic, (uint32_t)a.tiles_x), tx = t - ty * (uint32_t)a.tiles_x;",
            "    const uint32_t ty_ = fast_div(t, a.tx_magic, (uint32_t)a.tiles_x), tx = t - ty_ * (uint32_t)a.tiles_x, ty = (uint32_t)(a.tiles_per_frame / a.tiles_x) - 1u - ty_;")
    elif spec == "lpt":  # an XCD walks its frames tile row by tile row (row r of all its frames, then row r + 1 ...): the run ends with every frame's last rows
        rep("    const uint32_t frame_idx = fast_div(item, a.tpf_magic, (uint32_t)a.tiles_per_frame);\n    const uint32_t t = item - frame_idx * (uint32_t)a.tiles_per_frame;\n",
            "    uint32_t frame_idx = fast_div(item, a.tpf_magic, (uint32_t)a.tiles_per_frame);\n    uint32_t t = item - frame_idx * (uint32_t)a.tiles_per_frame;\n"
            "    if ((uint32_t)a.chunk % (uint32_t)a.tiles_per_frame == 0u) {\n"
            "        const uint32_t fpx = (uint32_t)a.chunk / (uint32_t)a.tiles_per_frame, per_row = fpx * (uint32_t)a.tiles_x;\n"
            "        const uint32_t r = in_run / per_row, rem = in_run - r * per_row, f = rem / (uint32_t)a.tiles_x;\n"
            "        frame_idx = (blockIdx.x & 7u) * fpx + f;\n        t = r * (uint32_t)a.tiles_x + (rem - f * (uint32_t)a.tiles_x);\n    }\n")
    elif spec in ("wg1", "wg2"):  # workgroups of one / two waves (tiles of 6 / 12 rows of 8-bit pixels): a finished wave's slot is refilled without waiting for three others
        rep("constexpr int kWG = 256;", "constexpr int kWG = %d;" % (64 * int(spec[2:])))
    elif spec.startswith("pf") and spec[2:].isdigit():  # tile prefetch: N byte loads per lane over the tile's source footprint (tools/patches/tile_prefetch.patch)
        apply_patch(files, os.path.join(ROOT, "tools", "patches", "tile_prefetch.patch"))
        rep("constexpr int kPrefetchLoads = 0; ", "constexpr int kPrefetchLoads = %s; " % spec[2:])
    elif spec == "nosplit":  # no half-height workgroups at the end of an XCD's run (the extra workgroups of the grid leave at once)
        rep("    if (seq >= (uint32_t)(a.chunk - a.tail_split)) {", "    if (seq >= (uint32_t)a.chunk) return;\n    if (false) {")
    elif spec == "noclass":  # timing only, all-interior row-affine footprints (abx --homography inset): no tile classification -- every tile is taken for an interior pair tile
        rep("    // -- the passes of this wave over the tile, in order.\n", "    tile_in = true, tile_out = false, tile_slanted = false, tile_affine = true, tile_pair = true;\n    // -- the passes of this wave over the tile, in order.\n")
    elif spec == "w4x":  # exactly four waves per SIMD whatever the register count (with noclass, whose kernels shrink)
        rep("amdgpu_waves_per_eu(NSRC > 1 ? 3 : kWavesPerSimd, 8)", "amdgpu_waves_per_eu(4, 4)")
    elif spec == "nopair":  # no tile takes the pair loads (rows_tiles.inc: tile_pair)
        rep("__ballot(stp >= 0.0 && stp <= 1.9375)", "__ballot(false)")
    elif spec == "allpair":  # every row-affine interior tile takes them (timing / diagnosis only: wrong beyond 2 source pixels per pixel)
        rep("__ballot(stp >= 0.0 && stp <= 1.9375)", "__ballot(true)")
    elif spec == "waves5":  # every warp kernel compiled for five waves per SIMD (<= 96 VGPRs)
        rep("constexpr int kWavesPerSimd = 4;", "constexpr int kWavesPerSimd = 5;")
    elif spec == "waves3":  # three waves per SIMD (<= 168 VGPRs): room for a third tap set
        rep("constexpr int kWavesPerSimd = 4;", "constexpr int kWavesPerSimd = 3;")
        rep("amdgpu_waves_per_eu(NSRC > 1 ? 3 : kWavesPerSimd, 8)", "amdgpu_waves_per_eu(3, 3)")  # (max = 3 too: otherwise the allocator still aims at four waves)
    elif spec in ("ahead3", "ahead4"):  # bilinear straight-line tiles with three / four tap sets in flight
        rep("constexpr int kAhead = INTERP == kNearest ? kFull : 2;", "constexpr int kAhead = INTERP == kNearest ? kFull : %s;" % spec[5:])
    elif spec == "ahead1":  # bilinear straight-line tiles with ONE tap set in flight
        rep("constexpr int kAhead = INTERP == kNearest ? kFull : 2;", "constexpr int kAhead = INTERP == kNearest ? kFull : 1;")
    elif spec == "stage":
        rep("    constexpr bool kStageEnabled = false;", "    constexpr bool kStageEnabled = true;")
    elif spec == "stagent":
        rep("    constexpr int kStageAux = 0; ", "    constexpr int kStageAux = 2; ")
    elif spec in ("ring16", "ring4"):
        raise SystemExit("ring specs are gone: kRing, kFlight are per-format constants of warp_rows.h")
    else:
        raise SystemExit("unknown spec " + spec)
    return files


def main():
    # --clock: the diagnostic build (-DBEVWARP_CLOCK: per-workgroup and per-role stamps, tools/clock.py) of the patched sources; needs
    # `make -C bev_amd/csrc variants/clock.so` first (its bevwarp_api / geom_kernels objects are linked in)
    clock = "--clock" in sys.argv
    if clock:
        sys.argv.remove("--clock")
    base = {n: open(os.path.join(CSRC, n)).read() for n in KERNEL_FILES}
    os.makedirs(os.path.join(CSRC, "variants"), exist_ok=True)
    procs = []
    for arg in sys.argv[1:]:
        name, specs = arg.split("=", 1)
        files = dict(base)
        for spec in [s for s in specs.split("+") if s]:
            files = patch(files, spec)
        tmp = "/tmp/ablate_%s" % name
        os.makedirs(tmp, exist_ok=True)
        for n, t in files.items():
            open(os.path.join(tmp, n), "w").write(t)
        for u in UNITS:  # the translation units themselves are never patched; they include the patched headers from tmp
            open(os.path.join(tmp, u + ".hip"), "w").write(open(os.path.join(CSRC, u + ".hip")).read())
            cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + (["-DBEVWARP_CLOCK"] if clock else []) + ["-I" + os.path.join(ROOT, "include"), "-I" + tmp, "-c", os.path.join(tmp, u + ".hip"), "-o", os.path.join(tmp, u + ".o")]
            procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE)))
    for name, p in procs:
        err = p.communicate()[1].decode()
        if p.returncode:
            raise SystemExit("%s: %s" % (name, err[-2000:]))
    for arg in sys.argv[1:]:
        name = arg.split("=", 1)[0]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(CSRC, "variants", name + ".so"),
                               os.path.join(CSRC, "variants/clock_bevwarp_api.o" if clock else "bevwarp_api.o"),
                               os.path.join(CSRC, "variants/clock_geom_kernels.o" if clock else "geom_kernels.o")] + ["/tmp/ablate_%s/%s.o" % (name, u) for u in UNITS])
        print("built", name)


if __name__ == "__main__":
    main()
