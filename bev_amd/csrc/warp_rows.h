// warp_rows.h -- batched BEV homography warp for MI355X (gfx950, wave64).  See DESIGN.md section 4.
//
// Replaces the per-frame cv2.warpPerspective call of the reference (vis_homo.py:89,91; bev/tool/compo.py:38,46,47).
//
// ONE kernel, warp_rows.  A workgroup of 4 waves owns a tile of TW x tile_h destination pixels (TW = 256 for 8-bit,
// 128 for float pixels); a wave owns whole TW-pixel row segments (rows dealt round-robin to the 4 waves) and pixel j of
// lane l is x0 + 64 j + l, so every load instruction covers 64 consecutive destination pixels.  Each row segment is
// classified from its two end pixels, in scalar registers:
//     FAST  both ends sample inside the frame by a margin, W keeps its sign  -> every pixel does: unguarded tap loads
//           (aligned 12-byte windows + funnel shift for 8-bit RGB), no per-pixel range or sign test at all
//     OUT   both ends beyond the same frame edge                             -> the border value
//     EDGE  the frame's edge crosses the segment                             -> fast coordinates, guarded taps
//     SLOW  W changes sign / is tiny, or coordinates leave the fixed-point range -> exact chain per pixel
// Passes are software-pipelined (full-height tiles: straight-line code with two tap sets in flight); a pass's pixels are
// transposed through a wave-private LDS row and the wave writes all its rows, with contiguous plain stores, after its last
// pass.  No workgroup barrier.
//
// Layout of the sources: coords.h (coordinate arithmetic), sample.h (blending, guarded sampler), this file (the kernel: its
// prologue here, its stages in the fragments rows_coords.inc / rows_sample.inc / rows_store.inc / rows_tiles.inc), and one
// translation unit per pixel type and interpolation (warp_u8_linear.hip, ...) so that the formats compile side by side.
//
// Coordinates are float64.  The reference rounds fX = (X0 + M0 x1) * (32 / W) half-to-even; the fast chain (one
// v_rcp_f64 + Newton step shared by the lane's pixels, FMAs, row terms evaluated once per row) lands within 2^-40
// relative of it and rounds through the float64 mantissa:  t = fX' * 2^27 + (1.5 * 2^52 + 2^26 + 2^8)  leaves
// floor(X / 32) in the HIGH dword (X = the rounded 1/32-px coordinate), X & 31 in bits 27..31 of the low dword and the
// distance to the nearest rounding boundary below.  A pixel whose low bits lie within 2^-19 unit of a boundary -- the
// only place the two chains can disagree -- re-runs the reference chain operation for operation (exact_px).
// 8-bit blending is exact integer arithmetic on v_dot4_u32_u8 / v_dot2_u32_u16; float blending keeps the reference's
// operation order (FMA contraction off).
//
// No MFMA: this is a gather.  The float kernel is bound by HBM; the 8-bit kernels by vector-ALU issue (float64
// coordinate chain + blending) and the texture path's cost per gather instruction.
#pragma once
#include "sample.h"

namespace bevwarp {
namespace {

// ===================================================================================================
// warp_rows<T, C, INTERP, RS4, PLANAR>
//   RS4     8-bit RGB bilinear only: the source row stride is a multiple of 4 bytes (both tap rows of a pixel then share
//           one window alignment and one funnel-shift amount)
//   PLANAR  the destination is C float32 planes, dst[c][y][x] = float(pixel) * pscale[c] + pbias[c] (the layout a detector takes)
// Register budget: 4 waves per SIMD -- what the FAST row loop needs; the rare row classes may spill.
// ===================================================================================================
// Diagnostic build only (-DBEVWARP_CLOCK, tools/clock.py, bench.py's sclk_mhz): wave 0 of every workgroup adds the shader-clock
// ticks (s_memtime) and the 100 MHz reference ticks (s_memrealtime) it lived for; their ratio is the clock the chip held.  The
// kernels of that build are named warp_rows_clockbuild, so that a kernel trace of bench.py does not mix them with the product's.
#ifdef BEVWARP_CLOCK
#define warp_rows warp_rows_clockbuild
// One record of kClkWords counters per workgroup (blockIdx mod kClkRecords), ACCUMULATED with plain read-modify-writes by one lane of the
// workgroup -- no atomics: thousands of workgroups adding to a handful of shared words serialise in the L2's atomic unit and slow the
// very kernel that is being timed several-fold.  Words: [0] shader ticks the workgroup lived, [1] 100-MHz ticks, [2] workgroups;
// staged tiles (rows_staged.inc): [4] producer ticks, [5] rows it found no free slot for at once, [6] source rows, [7] tiles;
// [8] consumer ticks (sum of three), [9] ticks waiting for source rows, [10] of those before the first row, [11] rows.
constexpr int kClkWords = 32, kClkRecords = 8192;  // (a record: words 0 .. 7 as listed, then [8 + 4 c ..] = the four consumer words of consumer c)
static __device__ unsigned long long g_clk[kClkRecords * kClkWords];
__device__ __forceinline__ void clk_add(int word, unsigned long long v) { g_clk[(blockIdx.x & (kClkRecords - 1)) * kClkWords + word] += v; }
#endif
// NSRC = 3 is warp_composite (bev/tool/compo.py:26-49) in one launch: a workgroup of 12 waves, four per source -- waves 0-3
// warp the background, 4-7 the foreground, 8-11 its mask, each group exactly as a workgroup of the plain kernel would, every
// group through its own homography and its own tile classification -- into an LDS copy of the tile instead of memory; after one
// barrier all twelve blend the three LDS tiles and store the composite.  The three warped images never exist in memory and
// every pixel is, by construction, what three bevwarp_warp calls produce.
constexpr int kCompositeRows = 16;  // tallest tile of the composite (its three LDS copies: 48 KB)
template <typename T, int C, int INTERP, bool RS4, bool PLANAR, int NSRC = 1>
__global__ __launch_bounds__(kWG * NSRC) __attribute__((amdgpu_waves_per_eu(NSRC > 1 ? 3 : kWavesPerSimd, 8))) void warp_rows(const WarpArgs a) {
    constexpr int PPL = pixels_per_lane<T>();
    constexpr int TW = 64 * PPL;                                 // tile width
    constexpr int kStrips = PPL;                                 // 64-pixel column strips of a tile (block ownership)
    constexpr int BR = PPL;                                      // rows of a block
    constexpr int PWd = 64 / PPL;                                // lanes per row of a block's patch (BlkSeg)
    constexpr int PBs = (int)sizeof(T) * C;                      // source bytes per pixel
    constexpr int TAPB = INTERP == kLinear ? 2 * PBs : PBs;      // bytes of one row's taps
    constexpr int LOADB = (TAPB + 3) & ~3;                       // loaded per row (whole dwords)
    constexpr int SH = INTERP == kLinear ? kInterBits : 0;
    using F = Fix<INTERP>;
    // 8-bit RGB bilinear: a tap pair (6 bytes at any byte address) is fetched as the ALIGNED 12-byte window around it and
    // funnel-shifted into place.  The texture path turns byte-unaligned 8-byte gathers that miss L1 into data at ~50
    // cycles per wave instruction and 4-byte-aligned 12-byte ones at ~18 (tools/ubench_stream.hip).
    constexpr bool kAligned = sizeof(T) == 1 && C == 3 && INTERP == kLinear;
    constexpr int WINB = kAligned ? 12 : LOADB;  // bytes a FAST row loads per tap row
    constexpr bool kPairable = kAligned && RS4 && NSRC == 1 && PPL == 4;  // (pair tiles: rows_sample.inc issue_p)
    constexpr int kM = kAligned ? 2 : 1;         // FAST: both ends inside by this many pixels (the aligned window starts
                                                 // up to 3 bytes early: never before its row)
    constexpr int TRW = 64 * PPL * (sizeof(T) == 1 ? 1 : C);  // dwords of a wave's transposition row
    static_assert(!RS4 || kAligned, "RS4 only qualifies the aligned-window variant");
    static_assert(NSRC == 1 || (NSRC == 3 && sizeof(T) == 1 && INTERP == kLinear && !PLANAR), "the composite is three 8-bit bilinear warps");
    // Deferred stores (plain kernel): a wave keeps the pixels of ALL its passes over the tile in LDS, one transposition row per
    // pass, and writes them to memory after its last pass.  vmcnt retires in issue order, loads and stores alike, so a store
    // issued in pass n sits in front of the loads of pass n + 1 and their s_waitcnt cannot be satisfied before the store has
    // been acknowledged by the memory system: with either kind of access alone the kernel runs at its ALU time, with both it
    // loses 13 us of 75 (ablations: profiles/r03_tables.txt).  Stored at the end of the tile, nothing waits behind them.
    // (composite: one row, its passes go to the LDS tiles at once.)
    constexpr int kRowsLds = NSRC > 1 ? 1 : (sizeof(T) == 1 ? 6 : 4);  // passes of a wave over the tallest tile (24 / 16 rows)
    // STAGED tiles (rows_staged.inc): one producer wave copies the tile's source rows into a ring of LDS slots with coalesced LDS-DMA
    // loads, three consumer waves read their taps from the ring.  The ring, two transposition rows per consumer and the four
    // hand-off words share the LDS of the deferred-store rows (a tile is processed one way or the other).
    // MEASURED AND NOT ENABLED (round 4, profiles/r04_staged_tiles.txt): bit-exact on the whole GPU suite, and within +-3 % of the gather
    // pipelines on float pixels (equal at 1080p with non-temporal ring fills, -3.4 % on the 4K shard) but 5 % SLOWER on 8-bit pixels,
    // whose consumers carry 15 % more vector instructions (LDS addresses) on SIMDs that are already the limiter.  The fragment stays
    // in the tree as the record of the experiment; `python tools/ablate.py stage=stage` builds a library with it switched on.
    constexpr bool kStageEnabled = false;
    constexpr bool kStageable = kStageEnabled && NSRC == 1 && INTERP == kLinear && C == 3 && (sizeof(T) == 4 || RS4);
    constexpr int kCons = kWaves - 1;                       // consumer waves of a staged tile
    constexpr int kRing = sizeof(T) == 1 ? 16 : 10;         // source rows the ring holds
    constexpr int kFlight = sizeof(T) == 1 ? 12 : 6;        // rows the producer keeps in flight (<= kRing - 2, rows_staged.inc)
    constexpr int kSlot = sizeof(T) == 1 ? 1536 : 3072;     // bytes of one slot: the widest row span a staged tile may have
    constexpr int kStageTr = 2 * TRW * 4;                   // bytes of a consumer's two transposition rows
    constexpr int kStageFlagOff = kRing * kSlot + kCons * kStageTr;
    constexpr int kStageAux = 0;                            // cache policy of the ring fills (0: default, 2: nt)
    constexpr int kTrDwords = kWaves * NSRC * kRowsLds * TRW;
    constexpr int kLdsDwords = kStageable && (kStageFlagOff + 16) / 4 > kTrDwords ? (kStageFlagOff + 16) / 4 : kTrDwords;
    constexpr int kLdsWaves = (kLdsDwords + kRowsLds * TRW - 1) / (kRowsLds * TRW);  // (= kWaves * NSRC unless a staged build needs more: its layout is a byte view)
    __shared__ __attribute__((aligned(16))) uint32_t s_tr[kLdsWaves][kRowsLds][TRW];
    // (composite only) the warped tiles, one packed pixel per dword: [source][row of the tile][pixel]
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[NSRC > 1 ? NSRC * kCompositeRows * TW : 4];
    constexpr int NEED = LOADB / 4;  // dwords of a tap row the blend takes, starting AT the left tap
#ifdef BEVWARP_CLOCK
    struct ClockStamp {
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        __device__ ~ClockStamp() {
            if (threadIdx.x == 0) {
                clk_add(0, __builtin_amdgcn_s_memtime() - t0);
                clk_add(1, __builtin_amdgcn_s_memrealtime() - r0);
                clk_add(2, 1ull);
            }
        }
    } clock_stamp;
#endif
    // block -> (frame, tile): one XCD (blockIdx & 7) works on one contiguous run of items
    // The last `tail_split` tiles an XCD dispatches are cut into an upper and a lower half, one workgroup each: the launch's
    // tail is then made of half-length workgroups.
    uint32_t seq = blockIdx.x >> 3;  // dispatch order within the XCD
    int half = -1;
    if (seq >= (uint32_t)(a.chunk - a.tail_split)) {
        const uint32_t j = seq - (uint32_t)(a.chunk - a.tail_split);
        seq = (uint32_t)(a.chunk - a.tail_split) + (j >> 1);
        half = (int)(j & 1u);
    }
    uint32_t in_run = seq + (blockIdx.x & 7u) * (uint32_t)a.stagger;  // (stagger * 7 < chunk: bevwarp_api.hip)
    if (in_run >= (uint32_t)a.chunk) in_run -= (uint32_t)a.chunk;
    const uint32_t item = (blockIdx.x & 7u) * (uint32_t)a.chunk + in_run;
    if (item >= (uint32_t)a.total_tiles) return;
    const uint32_t cls_idx = item * 3u + (uint32_t)(half + 1);  // this workgroup's entry of the verdict table (full tile, upper half, lower half)
    const uint32_t frame_idx = fast_div(item, a.tpf_magic, (uint32_t)a.tiles_per_frame);
    const uint32_t t = item - frame_idx * (uint32_t)a.tiles_per_frame;
    const uint32_t ty = fast_div(t, a.tx_magic, (uint32_t)a.tiles_x), tx = t - ty * (uint32_t)a.tiles_x;
    const int x0 = (int)tx * TW, y0 = (int)ty * a.tile_h + (half == 1 ? a.tile_h / 2 : 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, in an SGPR
    const int wave = NSRC > 1 ? (wave_all & (kWaves - 1)) : wave_all, sid = NSRC > 1 ? (wave_all >> 2) : 0;  // wave of its group of four / source
    // this wave's source (composite: background, foreground or mask; frames of one launch otherwise)
    const uint8_t* src_base = a.src;
    const double* m_base = a.minv;
    int64_t src_rs = a.src_rs;
    int src_w = a.src_w, src_h = a.src_h;
    if constexpr (NSRC > 1) {
        if (sid > 0) {
            src_base = a.xsrc[sid - 1], m_base = a.xminv[sid - 1], src_rs = a.xsrc_rs[sid - 1];
            src_w = a.xsrc_w[sid - 1], src_h = a.xsrc_h[sid - 1];
        }
    }
    const uint8_t* __restrict__ frame = src_base + (int64_t)frame_idx * a.src_fs;
    uint8_t* __restrict__ dframe = a.dst + (int64_t)frame_idx * a.dst_fs;
    const double* __restrict__ M = m_base + (int64_t)frame_idx * a.m_stride;
    const int y_last = min(y0 + (half >= 0 ? a.tile_h / 2 : a.tile_h), a.dst_h) - 1;
    if (y0 > y_last) return;  // (the lower half of a ragged last tile may be empty)

    SrcView view;
    view.frame = frame;
    view.rs = src_rs;
    view.w = src_w;
    view.h = src_h;
    const bool gray_src = NSRC > 1 && sizeof(T) == 1 && C == 3 && sid == 1 && a.fg_gray != 0;  // (constant false in the plain kernel)
    view.gray = gray_src;
#pragma unroll
    for (int k = 0; k < 4; k++) view.bf[k] = a.bval_f[k];
    view.bu = (uint32_t)a.bval_u8[0] | ((uint32_t)a.bval_u8[1] << 8) | ((uint32_t)a.bval_u8[2] << 16) | ((uint32_t)a.bval_u8[3] << 24);

    // -- limits of unguarded loads
    const int sx_lim = (int)(((int64_t)src_w * PBs - LOADB) / PBs);   // largest sx with sx*PBs + LOADB <= w*PBs
    const int sxw_lim = (int)(((int64_t)src_w * PBs - WINB) / PBs);   // same for the FAST rows' windows
    const int sy_lim = src_h - (INTERP == kLinear ? 2 : 1);
    const bool any_unguarded = (int64_t)src_w * PBs >= LOADB && sy_lim >= 0;
    const uint32_t sx_max = any_unguarded ? (uint32_t)sx_lim : 0u, sy_max = any_unguarded ? (uint32_t)sy_lim : 0u;
    const bool can_fast = (int64_t)src_w * PBs >= 32 && sxw_lim >= 2 * kM && sy_lim >= 2 * kM;

#include "rows_coords.inc"
#include "rows_sample.inc"
#include "rows_store.inc"
#include "rows_tiles.inc"
#include "rows_staged.inc"
#include "rows_run.inc"
    if constexpr (NSRC > 1) {
        // -- composite_reg_img (bev/tool/compo.py:16-23) on the three LDS tiles.  The reference evaluates
        //   round(fg * (m / 255) + bg * (1 - m / 255)) in float64 and clips to 255; with N = fg m + bg (255 - m) that value is N / 255
        // up to 2.3e-13, while N / 255 is never closer than 1 / 510 to a rounding boundary (2 N - 255 is odd), so the result is
        // exactly floor((N + 127) / 255), which never exceeds 255: integer arithmetic, no division ((x * 0x8081) >> 23 == x / 255
        // for x < 2^16).
        __syncthreads();
        const int rows = y_last - y0 + 1;
        for (int u = tid; u < rows * 64; u += kWG * NSRC) {
            const int r = u >> 6, x = x0 + 4 * (u & 63);
            if (x >= a.dst_w) continue;
            const uint4 pb = *reinterpret_cast<const uint4*>(&s_tile[(0 * kCompositeRows + r) * TW + (x - x0)]);
            const uint4 pf = *reinterpret_cast<const uint4*>(&s_tile[(1 * kCompositeRows + r) * TW + (x - x0)]);
            const uint4 pm = *reinterpret_cast<const uint4*>(&s_tile[(2 * kCompositeRows + r) * TW + (x - x0)]);
            const uint32_t b4[4] = {pb.x, pb.y, pb.z, pb.w}, f4[4] = {pf.x, pf.y, pf.z, pf.w}, m4[4] = {pm.x, pm.y, pm.z, pm.w};
            uint32_t p[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                p[i] = 0;
#pragma unroll
                for (int k = 0; k < C; k++) {
                    const uint32_t m = (m4[i] >> (8 * k)) & 0xffu, f = (f4[i] >> (8 * k)) & 0xffu, b = (b4[i] >> (8 * k)) & 0xffu;
                    const uint32_t n = __umul24(f, m) + __umul24(b, 255u - m) + 127u;
                    p[i] |= ((n * 0x8081u) >> 23) << (8 * k);
                }
            }
            uint8_t* d = dframe + (int64_t)(y0 + r) * a.dst_rs + (int64_t)x * C;
            const int lane_px = min(4, a.dst_w - x);
            if (__builtin_expect(a.dst_vec_ok && lane_px == 4, 1)) {
                if constexpr (C == 1) {
                    *reinterpret_cast<uint32_t*>(d) = p[0] | (p[1] << 8) | (p[2] << 16) | (p[3] << 24);
                } else if constexpr (C == 2) {
                    u32x2 o = {p[0] | (p[1] << 16), p[2] | (p[3] << 16)};
                    *reinterpret_cast<u32x2*>(d) = o;
                } else if constexpr (C == 3) {
                    u32x3 o = {p[0] | (p[1] << 24), (p[1] >> 8) | (p[2] << 16), (p[2] >> 16) | (p[3] << 8)};
                    *reinterpret_cast<u32x3*>(d) = o;
                } else {
                    u32x4 o = {p[0], p[1], p[2], p[3]};
                    *reinterpret_cast<u32x4*>(d) = o;
                }
            } else {
                for (int i = 0; i < lane_px; i++)
#pragma unroll
                    for (int k = 0; k < C; k++) d[i * C + k] = (uint8_t)(p[i] >> (8 * k));
            }
        }
    }
}

template <typename T, int C, int INTERP>
void launch_tci(const WarpArgs& a, dim3 grid, hipStream_t stream) {
    constexpr bool kRgb8Lin = sizeof(T) == 1 && C == 3 && INTERP == kLinear;
    if (a.planar) {
        if (kRgb8Lin && a.src_rs % 4 == 0)
            hipLaunchKernelGGL((warp_rows<T, C, INTERP, kRgb8Lin, true>), grid, dim3(kWG), 0, stream, a);
        else
            hipLaunchKernelGGL((warp_rows<T, C, INTERP, false, true>), grid, dim3(kWG), 0, stream, a);
        return;
    }
    if (kRgb8Lin && a.src_rs % 4 == 0)
        hipLaunchKernelGGL((warp_rows<T, C, INTERP, kRgb8Lin, false>), grid, dim3(kWG), 0, stream, a);
    else
        hipLaunchKernelGGL((warp_rows<T, C, INTERP, false, false>), grid, dim3(kWG), 0, stream, a);
}


// every channel count of one pixel type and interpolation: what one translation unit instantiates (warp_u8_linear.hip, ...)
template <typename T, int INTERP>
void launch_channels(const WarpArgs& a, int channels, dim3 grid, hipStream_t stream) {
    switch (channels) {
        case 1: launch_tci<T, 1, INTERP>(a, grid, stream); break;
        case 2: launch_tci<T, 2, INTERP>(a, grid, stream); break;
        case 3: launch_tci<T, 3, INTERP>(a, grid, stream); break;
        default: launch_tci<T, 4, INTERP>(a, grid, stream); break;
    }
}

#ifdef BEVWARP_CLOCK
inline hipError_t read_clock_of_this_unit(unsigned long long* out16, int reset) {  // out16 += this translation unit's counters
    static unsigned long long v[kClkRecords * kClkWords];  // (2 MB: not on the stack; the diagnostic build is single-threaded)
    hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_clk), sizeof(v));
    if (e != hipSuccess) return e;
    for (int r = 0; r < kClkRecords; r++) {
        for (int i = 0; i < 8; i++) out16[i] += v[r * kClkWords + i];
        for (int i = 8; i < 20; i++) out16[8 + (i & 3)] += v[r * kClkWords + i];  // the three consumers' words, folded
    }
    if (reset) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_clk));
        if (e == hipSuccess) e = hipMemset(p, 0, sizeof(v));
    }
    return e;
}
#endif

}  // namespace
}  // namespace bevwarp
