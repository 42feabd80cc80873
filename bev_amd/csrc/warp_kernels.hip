// warp_kernels.hip -- batched BEV homography warp for MI355X (gfx950, wave64).  See DESIGN.md section 4.
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
// Rows are software-pipelined one ahead: issue loads(n+1) -> store(n) -> coordinates(n+2) -> blend(n+1); results are
// transposed through a wave-private LDS row and written with contiguous non-temporal stores.  No workgroup barrier.
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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "warp_kernels.h"

#pragma clang fp contract(off)  // the exact chain rounds after every multiply and add; the fast chain asks for FMAs explicitly

namespace bevwarp {
namespace {

constexpr int kWG = 256;
constexpr int kWaves = kWG / 64;
constexpr int kInterBits = 5;

template <typename T>
constexpr int pixels_per_lane() { return sizeof(T) == 1 ? 4 : 2; }

__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t magic, uint32_t d) {
    // magic = floor(2^32 / d) + 1, exact while n * d < 2^32 (host guarantees); magic == 0 -> plain division
    return magic ? __umulhi(n, magic) : n / d;
}

// ---------------------------------------------------------------------------------------------------
// Exact coordinate chain (float64, no contraction): the reference algorithm operation for operation.
// M = inverse matrix, bx = left edge of the evaluation block the pixel belongs to, x1 = x - bx.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void row_terms(const double* __restrict__ M, int bx, int y, double& X0, double& Y0, double& W0) {
    const double dbx = (double)bx, dy = (double)y;
    X0 = (M[0] * dbx + M[1] * dy) + M[2];
    Y0 = (M[3] * dbx + M[4] * dy) + M[5];
    W0 = (M[6] * dbx + M[7] * dy) + M[8];
}

__device__ __forceinline__ int round_sat(double v) {
    // clamp to the int range then round half to even; a NaN lands on INT_MIN, which like the
    // reference's INT_MAX is outside every admissible source image.
    v = fmin(fmax(v, -2147483648.0), 2147483647.0);
    return (int)rint(v);
}

template <int INTERP>
__device__ __forceinline__ void map_pixel_exact(double Xn, double Yn, double W, int& X, int& Y) {
    W = (W != 0.0) ? ((INTERP == kLinear ? 32.0 : 1.0) / W) : 0.0;  // IEEE division
    X = round_sat(Xn * W);
    Y = round_sat(Yn * W);
}

// ---------------------------------------------------------------------------------------------------
// Fast coordinate chain: fixed point through the float64 mantissa.
//   p  = coordinate in source PIXELS times 2^32 (the numerators carry the 2^32), relative error <= 2^-46
//   t  = p + kMagic,  kMagic = 1.5 * 2^52 + half + win
// t lies in [2^52, 2^53): its mantissa is the integer V = rne(p + half + win) + 2^51, so with U = one output unit
// (2^27 for bilinear = 1/32 px, 2^32 for nearest = 1 px) and half = U / 2:
//   high dword  = 0x43380000 + floor(X / (2^32 / U))      X = the coordinate rounded to output units
//   low dword   = (X mod (2^32 / U)) * U + distance field
// and X equals the reference's rne() unless the distance field lies in [0, 2 win): within win = 2^-19 unit of a rounding
// boundary.  (|ours - reference| <= 2^-45.9 |fX| < 2^-19 for every |fX| < 2^24 the binade admits.)
// ---------------------------------------------------------------------------------------------------
constexpr double kTwo32 = 4294967296.0;
constexpr uint32_t kHiBias = 0x43380000u;   // high dword of 1.5 * 2^52
constexpr uint32_t kHiExp = 0x43300000u;    // exponent field of [2^52, 2^53)
template <int INTERP>
struct Fix {
    static constexpr double kHalf = INTERP == kLinear ? 67108864.0 /* 2^26 */ : 2147483648.0 /* 2^31 */;
    static constexpr double kWin = INTERP == kLinear ? 256.0 /* 2^-19 * 2^27 */ : 8192.0 /* 2^-19 * 2^32 */;
    static constexpr double kMagic = 6755399441055744.0 + kHalf + kWin;
    static constexpr uint32_t kTieMask = INTERP == kLinear ? 0x07fffe00u : 0xffffc000u;  // distance field minus its low 9 / 14 bits
};

// (high, low) dwords of t -> the integer coordinate X of the reference (1/32 px units for bilinear)
template <int INTERP>
__device__ __forceinline__ int fix_to_int(uint32_t hi, uint32_t lo) {
    if (INTERP == kLinear) return (int)(__builtin_amdgcn_alignbit(hi, lo, 27) - 0x67000000u);  // (hi << 5 | lo >> 27) - 32 * kHiBias mod 2^32
    return (int)(hi - kHiBias);
}
// the inverse: an exact coordinate put back into the (high, low) form (distance field cleared)
template <int INTERP>
__device__ __forceinline__ void int_to_fix(int X, uint32_t& hi, uint32_t& lo) {
    if (INTERP == kLinear) {
        hi = kHiBias + (uint32_t)(X >> kInterBits);
        lo = ((uint32_t)X & 31u) << 27 | 0x04000000u;
    } else {
        hi = kHiBias + (uint32_t)X;
        lo = 0x80000000u;
    }
}

__device__ __forceinline__ double rcp_newton(double w) {
    double r = __builtin_amdgcn_rcp(w);  // v_rcp_f64: relative error 2^-24.4 (measured)
    r = __builtin_fma(__builtin_fma(-w, r, 1.0), r, r);  // -> 2^-48.7
    return r;
}

// ---------------------------------------------------------------------------------------------------
// Blending.  u8: 15-bit fixed point of the reference == exact integer form
//   (sum_i p_i * w_i * 32 + 2^14) >> 15  ==  (wy0 * (wx0 p00 + wx1 p01) + wy1 * (wx0 p10 + wx1 p11) + 512) >> 10
// f32: float weights (1-fy)(1-fx).. (exact multiples of 1/1024), 4 products summed left to right.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t blend_u8(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, uint32_t fx, uint32_t fy) {
    const uint32_t wx1 = fx, wx0 = 32u - fx, wy1 = fy, wy0 = 32u - fy;
    const uint32_t h0 = p00 * wx0 + p01 * wx1;
    const uint32_t h1 = p10 * wx0 + p11 * wx1;
    return (h0 * wy0 + h1 * wy1 + 512u) >> 10;
}

__device__ __forceinline__ float blend_f32(float p00, float p01, float p10, float p11, float w00, float w01, float w10, float w11) {
    return ((p00 * w00 + p01 * w01) + p10 * w10) + p11 * w11;
}

__device__ __forceinline__ void weights_f32(int fx, int fy, float& w00, float& w01, float& w10, float& w11) {
    const float s = 1.0f / 32.0f;
    const float tx1 = (float)fx * s, ty1 = (float)fy * s;
    const float tx0 = 1.0f - tx1, ty0 = 1.0f - ty1;
    w00 = ty0 * tx0;
    w01 = ty0 * tx1;
    w10 = ty1 * tx0;
    w11 = ty1 * tx1;
}

// vertical stage of the 8-bit blend: wy0 * top + wy1 * bot + 2^15 as ONE v_dot2_u32_u16 on the packed pair (top and bot
// are horizontal sums <= 8160; the weights are scaled by 64 so that the result byte sits in bits 16..23:
// ((h0 wy0 + h1 wy1) * 64 + 2^15) >> 16 == (S + 512) >> 10)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t vblend_u8(uint32_t top, uint32_t bot, uint32_t wy01) {
    const uint32_t tb = top | (bot << 16);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, tb), __builtin_bit_cast(u16x2, wy01), 32768u, false);
}

// Packed 8-bit blend of up to 4 channels: p?? are pixels with channel k in byte k.  Horizontal sums with v_dot4_u32_u8
// (weights 32 - fx, fx <= 32) on byte-selected tap pairs.
template <int C>
__device__ __forceinline__ uint32_t blend_u8_packed(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, uint32_t fx, uint32_t fy) {
    const uint32_t wlo = fx * 255u + 32u;          // bytes (32 - fx, fx, 0, 0)
    const uint32_t whi = wlo << 16;                // bytes (0, 0, 32 - fx, fx)
    const uint32_t wy01 = fy * 0x3fffc0u + 2048u;  // halves (2048 - 64 fy, 64 fy): the vertical weights as a packed pair
    // (a.k, b.k, a.k', b.k') for channel pairs (0,1) and (2,3)
    const uint32_t t01 = __builtin_amdgcn_perm(p01, p00, 0x05010400u);
    const uint32_t b01 = __builtin_amdgcn_perm(p11, p10, 0x05010400u);
    uint32_t s[4];
    s[0] = vblend_u8(__builtin_amdgcn_udot4(t01, wlo, 0u, false), __builtin_amdgcn_udot4(b01, wlo, 0u, false), wy01);
    if (C > 1) s[1] = vblend_u8(__builtin_amdgcn_udot4(t01, whi, 0u, false), __builtin_amdgcn_udot4(b01, whi, 0u, false), wy01);
    if (C > 2) {
        const uint32_t t23 = __builtin_amdgcn_perm(p01, p00, 0x07030602u);
        const uint32_t b23 = __builtin_amdgcn_perm(p11, p10, 0x07030602u);
        s[2] = vblend_u8(__builtin_amdgcn_udot4(t23, wlo, 0u, false), __builtin_amdgcn_udot4(b23, wlo, 0u, false), wy01);
        if (C > 3) s[3] = vblend_u8(__builtin_amdgcn_udot4(t23, whi, 0u, false), __builtin_amdgcn_udot4(b23, whi, 0u, false), wy01);
    }
    // gather byte 2 of every sum
    uint32_t out = (C > 1) ? __builtin_amdgcn_perm(s[1], s[0], 0x0c0c0602u) : ((s[0] >> 16) & 0xffu);
    if (C == 3) out = __builtin_amdgcn_perm(s[2], out, 0x0c060100u);
    if (C == 4) out = __builtin_amdgcn_perm(__builtin_amdgcn_perm(s[3], s[2], 0x06020c0cu), out, 0x07060100u);
    return out;
}

// 8-bit RGB straight from a tap window: (a1:a0) / (b1:b0) hold bytes 0..7 of the upper / lower source row starting
// at the left tap (left pixel = bytes 0 1 2, right pixel = bytes 3 4 5); the byte selects do the unpacking.
__device__ __forceinline__ uint32_t blend_u8_rgb_window(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t fx, uint32_t fy) {
    const uint32_t wlo = fx * 255u + 32u, whi = wlo << 16;
    const uint32_t wy01 = fy * 0x3fffc0u + 2048u;  // halves (2048 - 64 fy, 64 fy)
    const uint32_t t01 = __builtin_amdgcn_perm(a1, a0, 0x04010300u), u01 = __builtin_amdgcn_perm(b1, b0, 0x04010300u);  // L.c0 R.c0 L.c1 R.c1
    const uint32_t t2 = __builtin_amdgcn_perm(a1, a0, 0x0c0c0502u), u2 = __builtin_amdgcn_perm(b1, b0, 0x0c0c0502u);    // L.c2 R.c2 0 0
    const uint32_t s0 = vblend_u8(__builtin_amdgcn_udot4(t01, wlo, 0u, false), __builtin_amdgcn_udot4(u01, wlo, 0u, false), wy01);
    const uint32_t s1 = vblend_u8(__builtin_amdgcn_udot4(t01, whi, 0u, false), __builtin_amdgcn_udot4(u01, whi, 0u, false), wy01);
    const uint32_t s2 = vblend_u8(__builtin_amdgcn_udot4(t2, wlo, 0u, false), __builtin_amdgcn_udot4(u2, wlo, 0u, false), wy01);
    return __builtin_amdgcn_perm(s2, __builtin_amdgcn_perm(s1, s0, 0x0c0c0602u), 0x0c060100u);
}

// What the guarded sampler needs of the source frame, by value (taking the address of the kernel-argument struct would
// push it to scratch).
struct SrcView {
    const uint8_t* frame;
    int64_t rs;
    int w, h;
    float bf[4];
    uint32_t bu;  // border bytes packed
};

template <typename T>
__device__ __forceinline__ T border_of(const SrcView& a, int k);
template <>
__device__ __forceinline__ uint8_t border_of<uint8_t>(const SrcView& a, int k) { return (uint8_t)(a.bu >> (8 * k)); }
template <>
__device__ __forceinline__ float border_of<float>(const SrcView& a, int k) { return a.bf[k]; }

// A pixel in registers: u8 pixels travel packed in one dword (channel k in byte k, unused bytes 0),
// f32 pixels as C floats.  (A uint8_t[C] array would be demoted to scratch memory.)
template <typename T, int C>
struct Pixel {
    float v[C];
};
template <int C>
struct Pixel<uint8_t, C> {
    uint32_t packed;
};

// One pixel straight from global memory with per-tap bounds checks (EDGE / SLOW rows).  Every tap is loaded from the
// CLAMPED coordinate (always a valid address) and replaced by the border value afterwards when its true coordinate is
// outside: the loads are unconditional, so they all issue before the first wait.
template <typename T, int C, int INTERP>
__device__ __forceinline__ Pixel<T, C> sample_global(const SrcView& a, int X, int Y) {
    const uint8_t* __restrict__ frame = a.frame;
    Pixel<T, C> out;
    if constexpr (sizeof(T) == 1) out.packed = 0;
    if (INTERP == kNearest) {
        const bool in = (unsigned)X < (unsigned)a.w && (unsigned)Y < (unsigned)a.h;
        const int cx = min(max(X, 0), a.w - 1), cy = min(max(Y, 0), a.h - 1);
        const T* p = reinterpret_cast<const T*>(frame + (int64_t)cy * a.rs) + (int64_t)cx * C;
        T t[C];
#pragma unroll
        for (int k = 0; k < C; k++) t[k] = p[k];
#pragma unroll
        for (int k = 0; k < C; k++) {
            const T v = in ? t[k] : border_of<T>(a, k);
            if constexpr (sizeof(T) == 1)
                out.packed |= (uint32_t)v << (8 * k);
            else
                out.v[k] = v;
        }
        return out;
    }
    const int sx = X >> kInterBits, sy = Y >> kInterBits, fx = X & 31, fy = Y & 31;
    const bool xin0 = (unsigned)sx < (unsigned)a.w, xin1 = (unsigned)(sx + 1) < (unsigned)a.w;
    const bool yin0 = (unsigned)sy < (unsigned)a.h, yin1 = (unsigned)(sy + 1) < (unsigned)a.h;
    const int cx0 = min(max(sx, 0), a.w - 1), cx1 = min(max(sx + 1, 0), a.w - 1);
    const int cy0 = min(max(sy, 0), a.h - 1), cy1 = min(max(sy + 1, 0), a.h - 1);
    const T* r0 = reinterpret_cast<const T*>(frame + (int64_t)cy0 * a.rs);
    const T* r1 = reinterpret_cast<const T*>(frame + (int64_t)cy1 * a.rs);
    T t00[C], t01[C], t10[C], t11[C];
#pragma unroll
    for (int k = 0; k < C; k++) {
        t00[k] = r0[(int64_t)cx0 * C + k];
        t01[k] = r0[(int64_t)cx1 * C + k];
        t10[k] = r1[(int64_t)cx0 * C + k];
        t11[k] = r1[(int64_t)cx1 * C + k];
    }
    float w00 = 0, w01 = 0, w10 = 0, w11 = 0;
    if constexpr (sizeof(T) == 4) weights_f32(fx, fy, w00, w01, w10, w11);
#pragma unroll
    for (int k = 0; k < C; k++) {
        const T b = border_of<T>(a, k);
        const T v00 = (xin0 && yin0) ? t00[k] : b;
        const T v01 = (xin1 && yin0) ? t01[k] : b;
        const T v10 = (xin0 && yin1) ? t10[k] : b;
        const T v11 = (xin1 && yin1) ? t11[k] : b;
        if constexpr (sizeof(T) == 1)
            out.packed |= blend_u8(v00, v01, v10, v11, fx, fy) << (8 * k);
        else
            out.v[k] = blend_f32(v00, v01, v10, v11, w00, w01, w10, w11);
    }
    return out;
}

template <int N>
struct Bytes {
    uint32_t w[N / 4];
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ===================================================================================================
// warp_rows<T, C, INTERP, RS4, PLANAR>
//   RS4     8-bit RGB bilinear only: the source row stride is a multiple of 4 bytes (both tap rows of a pixel then share
//           one window alignment and one funnel-shift amount)
//   PLANAR  8-bit sources only: the destination is C float32 planes, dst[c][y][x] = float(pixel) * pscale[c] + pbias[c]
// Register budget: 4 waves per SIMD -- what the FAST row loop needs; the rare row classes may spill.
// ===================================================================================================
#ifndef BEVWARP_U8LIN_WAVES
#define BEVWARP_U8LIN_WAVES 3
#endif
#ifndef BEVWARP_STAGED_WAVES
#define BEVWARP_STAGED_WAVES 4
#endif
// MODE: how an instantiation feeds its taps.  A tile is cut into groups of 4 rows (one per wave); a test of the group's four
// corner pixels says whether the group can be STAGED (every pixel samples inside the frame, W of one sign, the source box
// fits the LDS ring).
//   kGather  every row through the software-pipelined gather loop (taps of the next row in flight in VGPRs: 3 waves per
//            SIMD for 8-bit RGB bilinear).  Used when the source layout rules staging out.
//   kStaged  the staged groups of a tile from LDS (no taps in flight in registers: 4 waves per SIMD, which is what the
//            ALU-bound 8-bit bilinear formats need), then the tile's other rows -- the frame's edge crosses them, or their
//            boxes do not fit -- through the same row classes one row at a time, within the same register budget.
enum { kGather = 0, kStaged = 1 };
// Diagnostic build only (-DBEVWARP_CLOCK, tools/clock.py): wave 0 of every workgroup adds the shader-clock ticks
// (s_memtime) and the 100 MHz reference ticks (s_memrealtime) it lived for; their ratio is the clock the chip held.
#ifdef BEVWARP_CLOCK
__device__ unsigned long long g_clk[16];  // [0..2] life ticks / reference ticks / workgroups, [4..] ticks per phase of the row loop
#define STAMP(i)                                                           \
    do {                                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        phase_[i] += now_ - stamp_;                                        \
        stamp_ = now_;                                                     \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
template <typename T, int C, int INTERP, int MODE>
constexpr int waves_per_simd() {
    return MODE == kStaged ? BEVWARP_STAGED_WAVES : ((sizeof(T) == 1 && C >= 3 && INTERP == kLinear) ? BEVWARP_U8LIN_WAVES : 4);
}
// formats with a staged kernel (bilinear 8-bit pixels: the ones bound by vector-ALU issue)
template <typename T, int C, int INTERP>
constexpr bool has_staged_kernel() { return sizeof(T) == 1 && INTERP == kLinear; }
template <typename T, int C, int INTERP, bool RS4, bool PLANAR, int MODE>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(waves_per_simd<T, C, INTERP, MODE>(), 8))) void warp_rows(const WarpArgs a) {
    constexpr int PPL = pixels_per_lane<T>();
    constexpr int TW = 64 * PPL;
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
    constexpr int kM = kAligned ? 2 : 1;         // FAST: both ends inside by this many pixels (the aligned window starts
                                                 // up to 3 bytes early: never before its row)
    constexpr int TRW = 64 * PPL * (sizeof(T) == 1 ? 1 : C);  // dwords of a wave's transposition row
    static_assert(!PLANAR || sizeof(T) == 1, "planar output is the 8-bit -> float32 egress path");
    static_assert(!RS4 || kAligned, "RS4 only qualifies the aligned-window variant");
    __shared__ __attribute__((aligned(16))) uint32_t s_tr[kWaves][TRW];
    // Staged tiles (8-bit pixels): the source box of every 4-row group of the tile goes to LDS by coalesced 16-byte
    // LDS-DMA row loads, two groups in flight; taps are dword windows read from LDS (+ funnel shift when a pixel is not a
    // whole number of dwords).  kPitch covers a TW-pixel segment magnified up to 1.9 x; a tile whose groups need more
    // than kStageRows source rows or kPitch bytes per row takes the gather path below.
    constexpr bool kStage = MODE == kStaged;
    static_assert(!kStage || has_staged_kernel<T, C, INTERP>(), "no staged kernel for this format");
    constexpr int kStageRows = 10;
    constexpr int kPitch = ((TW * PBs * 19 / 10 + 48) + 255) & ~255;
    constexpr int kBufBytes = kStageRows * kPitch;
    constexpr bool kFunnel = (PBs % 4) != 0;
    constexpr int NEED = LOADB / 4;                 // dwords of a tap row the blend takes, starting AT the left tap
    constexpr int NDW = kFunnel ? NEED + 1 : NEED;  // dwords read per tap row (the aligned window around them)
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kStage ? 2 * kBufBytes : 16];

#ifdef BEVWARP_CLOCK
    struct ClockStamp {
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        __device__ ~ClockStamp() {
            if (threadIdx.x == 0) {
                atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - t0);
                atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - r0);
                atomicAdd(&g_clk[2], 1ull);
            }
        }
    } clock_stamp;
#endif
    // block -> (frame, tile): one XCD (blockIdx & 7) works on one contiguous run of items
    const uint32_t item = (blockIdx.x & 7u) * (uint32_t)a.chunk + (blockIdx.x >> 3);
    if (item >= (uint32_t)a.total_tiles) return;
    const uint32_t frame_idx = fast_div(item, a.tpf_magic, (uint32_t)a.tiles_per_frame);
    const uint32_t t = item - frame_idx * (uint32_t)a.tiles_per_frame;
    const uint32_t ty = fast_div(t, a.tx_magic, (uint32_t)a.tiles_x), tx = t - ty * (uint32_t)a.tiles_x;
    const int x0 = (int)tx * TW, y0 = (int)ty * a.tile_h;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, in an SGPR
    const uint8_t* __restrict__ frame = a.src + (int64_t)frame_idx * a.src_fs;
    uint8_t* __restrict__ dframe = a.dst + (int64_t)frame_idx * a.dst_fs;
    const double* __restrict__ M = a.minv + (int64_t)frame_idx * a.m_stride;
    const int y_last = min(y0 + a.tile_h, a.dst_h) - 1;

    SrcView view;
    view.frame = frame;
    view.rs = a.src_rs;
    view.w = a.src_w;
    view.h = a.src_h;
#pragma unroll
    for (int k = 0; k < 4; k++) view.bf[k] = a.bval_f[k];
    view.bu = (uint32_t)a.bval_u8[0] | ((uint32_t)a.bval_u8[1] << 8) | ((uint32_t)a.bval_u8[2] << 16) | ((uint32_t)a.bval_u8[3] << 24);

    // -- limits of unguarded loads
    const int sx_lim = (int)(((int64_t)a.src_w * PBs - LOADB) / PBs);   // largest sx with sx*PBs + LOADB <= w*PBs
    const int sxw_lim = (int)(((int64_t)a.src_w * PBs - WINB) / PBs);   // same for the FAST rows' windows
    const int sy_lim = a.src_h - (INTERP == kLinear ? 2 : 1);
    const bool any_unguarded = (int64_t)a.src_w * PBs >= LOADB && sy_lim >= 0;
    const uint32_t sx_max = any_unguarded ? (uint32_t)sx_lim : 0u, sy_max = any_unguarded ? (uint32_t)sy_lim : 0u;
    const bool can_fast = (int64_t)a.src_w * PBs >= 32 && sxw_lim >= 2 * kM && sy_lim >= 2 * kM;

    // -- wave-uniform terms of the fast chain (numerators carry the 2^32 of the fixed-point form)
    auto uniform_f64 = [](double v) {  // a wave-uniform double, moved to scalar registers
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    };
    const double m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5], m6 = M[6], m7 = M[7], m8 = M[8];
    const double x0d = (double)x0;
    const double CX = uniform_f64((m0 * x0d + m2) * kTwo32), CY = uniform_f64((m3 * x0d + m5) * kTwo32), CW = uniform_f64(m6 * x0d + m8);
    const double RX = uniform_f64(m1 * kTwo32), RY = uniform_f64(m4 * kTwo32), RW = uniform_f64(m7);             // per destination row
    const double DX = uniform_f64(m0 * (64.0 * kTwo32)), DY = uniform_f64(m3 * (64.0 * kTwo32)), DW = uniform_f64(m6 * 64.0);  // per 64 pixels
    const double ld = (double)lane;
    const double cx0 = (m0 * kTwo32) * ld, cy0 = (m3 * kTwo32) * ld, cw0 = m6 * ld;                              // per lane

    // -- byte offsets of FAST rows straight from the high dwords (24-bit multiplies: the host guarantees row stride < 2^24
    // and frames < 2 GiB; a FAST row has 0 <= sx, sy < 2^15, so the low 24 bits of a high dword are 0x380000 + s)
    const uint32_t rs32 = (uint32_t)a.src_rs;
    const uint32_t fa = kAligned ? (uint32_t)(reinterpret_cast<uintptr_t>(frame) & 3u) : 0u;
    const uint8_t* frame_al = frame - fa;  // 4-byte aligned (frames need not be)
    const uint32_t kOff = fa - 0x380000u * (rs32 + (uint32_t)PBs);
    const uint8_t* dummy = reinterpret_cast<const uint8_t*>(M);  // 72 valid bytes: what rows that are not FAST "load"

    // OUT rows may be filled with the border value when blending four border taps gives it back exactly:
    // always for 8-bit (the fixed-point weights sum to 2^15) and nearest; for float bilinear only for +0
    bool fill_ok = true;
    if (sizeof(T) == 4 && INTERP == kLinear)
        for (int k = 0; k < C; k++) fill_ok = fill_ok && __float_as_uint(a.bval_f[k]) == 0u;

    enum { kFast = 0, kOut = 1, kEdge = 2, kSlow = 3 };
    // the reference's chain for pixel j of this lane (rare: tie windows, SLOW rows); the matrix is re-read here so that the
    // row loop does not carry it in registers
    auto exact_px = [&](int y, int j, int& Xe, int& Ye) __attribute__((always_inline)) {
        const double* Mp = M;
        asm volatile("" : "+s"(Mp));  // (keeps the loads below inside this rare branch)
        double Me[9];
#pragma unroll
        for (int i = 0; i < 9; i++) Me[i] = Mp[i];
        const int x = x0 + 64 * j + lane;
        const int bx = (int)(fast_div((uint32_t)x, a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0);
        double X0, Y0, W0;
        row_terms(Me, bx, y, X0, Y0, W0);
        const double x1 = (double)(x - bx);
        map_pixel_exact<INTERP>(X0 + Me[0] * x1, Y0 + Me[3] * x1, W0 + Me[6] * x1, Xe, Ye);
    };

    // Row state handed from the coordinate stage to the load / blend stages, 3 dwords per pixel:
    //   FAST      S0 = byte offset of the tap window, S1 / S2 = low dwords of tX / tY (fx, fy in bits 27..31)
    //   others    S0 = 0 (the dummy load), S1 / S2 = integer coordinates X, Y
    int out_side = 0;  // set for an OUT row: which frame edge the segment lies beyond, and the sign of W
    int end_sxa = 0, end_sya = 0, end_sxb = 0, end_syb = 0;  // source pixel of the two ends of the row coords_s saw last
    // the fast chain of one row segment: (high, low) dwords of tX / tY for the lane's pixels; returns the tie flag (0 = some
    // coordinate of this lane lies in a tie window) and the high dwords of the lane's first / last W
    auto chain_u = [&](double UX, double UY, double UW, uint32_t (&hx)[PPL], uint32_t (&lx)[PPL], uint32_t (&hy)[PPL], uint32_t (&ly)[PPL],
                       uint32_t& w_first, uint32_t& w_last) __attribute__((always_inline)) -> uint32_t {
        double W[PPL], r[PPL];
        W[0] = UW + cw0;
#pragma unroll
        for (int j = 1; j < PPL; j++) W[j] = W[j - 1] + DW;
        // one reciprocal per lane: 1 / (W0 W1 [W2 W3]), then back-substitution
        if constexpr (PPL == 4) {
            const double p01 = W[0] * W[1], p23 = W[2] * W[3];
            const double inv = rcp_newton(p01 * p23);
            const double i01 = inv * p23, i23 = inv * p01;
            r[0] = i01 * W[1];
            r[1] = i01 * W[0];
            r[2] = i23 * W[3];
            r[3] = i23 * W[2];
        } else {
            const double inv = rcp_newton(W[0] * W[1]);
            r[0] = inv * W[1];
            r[1] = inv * W[0];
        }
        uint32_t tie = 0xffffffffu;
        double Xn = UX + cx0, Yn = UY + cy0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const double tx_ = __builtin_fma(Xn, r[j], F::kMagic), ty_ = __builtin_fma(Yn, r[j], F::kMagic);
            hx[j] = (uint32_t)__double2hiint(tx_), lx[j] = (uint32_t)__double2loint(tx_);
            hy[j] = (uint32_t)__double2hiint(ty_), ly[j] = (uint32_t)__double2loint(ty_);
            tie = min(tie, min(lx[j] & F::kTieMask, ly[j] & F::kTieMask));
            if (j + 1 < PPL) {
                Xn += DX;
                Yn += DY;
            }
        }
        w_first = (uint32_t)__double2hiint(W[0]);
        w_last = (uint32_t)__double2hiint(W[PPL - 1]);
        return tie;
    };
    auto chain = [&](int y, uint32_t (&hx)[PPL], uint32_t (&lx)[PPL], uint32_t (&hy)[PPL], uint32_t (&ly)[PPL], uint32_t& w_first, uint32_t& w_last)
                     __attribute__((always_inline)) -> uint32_t {
        const double dy = (double)y;  // the row terms at the segment's first pixel
        return chain_u(__builtin_fma(RX, dy, CX), __builtin_fma(RY, dy, CY), __builtin_fma(RW, dy, CW), hx, lx, hy, ly, w_first, w_last);
    };
    // rare: the lane's pixels that lie within 2^-19 of a rounding boundary (or are NaN) take the exact chain
    auto fix_ties = [&](int y, uint32_t (&hx)[PPL], uint32_t (&lx)[PPL], uint32_t (&hy)[PPL], uint32_t (&ly)[PPL]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if ((lx[j] & F::kTieMask) == 0 || (ly[j] & F::kTieMask) == 0) {
                int Xe, Ye;
                exact_px(y, j, Xe, Ye);
                int_to_fix<INTERP>(Xe, hx[j], lx[j]);
                int_to_fix<INTERP>(Ye, hy[j], ly[j]);
            }
        }
    };
    auto coords_s = [&](int y, uint32_t (&S0)[PPL], uint32_t (&S1)[PPL], uint32_t (&S2)[PPL]) -> int {
        uint32_t hx[PPL], lx[PPL], hy[PPL], ly[PPL], w_first, w_last;
        const uint32_t tie = chain(y, hx, lx, hy, ly, w_first, w_last);
        // -- classify the segment from its ends (pixel 0 of lane 0, pixel PPL-1 of lane 63), in scalar registers
        auto lane_u32 = [](uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); };
        const uint32_t hxa = lane_u32(hx[0], 0), hya = lane_u32(hy[0], 0), hxb = lane_u32(hx[PPL - 1], 63), hyb = lane_u32(hy[PPL - 1], 63);
        const uint32_t wa = lane_u32(w_first, 0), wb = lane_u32(w_last, 63);
        const uint32_t ea = (wa >> 20) & 0x7ffu, eb = (wb >> 20) & 0x7ffu;  // 2^-199 .. 2^199: the shared reciprocal is safe
        const bool w_ok = ((wa ^ wb) >> 31) == 0 && ea - 824u <= 398u && eb - 824u <= 398u;
        // source pixel of the two ends (a high dword outside the binade gives |s| >= 2^19: outside every limit below)
        const int sxa = (int)(hxa - kHiBias), sya = (int)(hya - kHiBias), sxb = (int)(hxb - kHiBias), syb = (int)(hyb - kHiBias);
        end_sxa = sxa, end_sya = sya, end_sxb = sxb, end_syb = syb;
        const bool in = can_fast && w_ok && (uint32_t)(sxa - kM) <= (uint32_t)(sxw_lim - 2 * kM) && (uint32_t)(sxb - kM) <= (uint32_t)(sxw_lim - 2 * kM) &&
                        (uint32_t)(sya - kM) <= (uint32_t)(sy_lim - 2 * kM) && (uint32_t)(syb - kM) <= (uint32_t)(sy_lim - 2 * kM);
        int cls = kFast;
        if (__builtin_expect(!in, 0)) {  // (the common class costs no further scalar work)
            const bool e_ok = (((hxa ^ kHiExp) | (hya ^ kHiExp) | (hxb ^ kHiExp) | (hyb ^ kHiExp)) >> 20) == 0;  // all four inside the binade
            const bool out = (sxa <= -3 && sxb <= -3) || (sxa > a.src_w && sxb > a.src_w) || (sya <= -3 && syb <= -3) || (sya > a.src_h && syb > a.src_h);
            // kEdge: W of one sign and both ends representable => every pixel between them is (the map is monotone along
            // the segment), taps need guards
            cls = !(e_ok && w_ok) ? kSlow : ((out && fill_ok) ? kOut : kEdge);
            out_side = ((sxa <= -3 && sxb <= -3) ? 1 : (sxa > a.src_w && sxb > a.src_w) ? 2 : (sya <= -3 && syb <= -3) ? 3 : 4) | (int)((wa >> 31) << 3);
        }
        if (tie == 0 && cls != kSlow) fix_ties(y, hx, lx, hy, ly);
        if (__builtin_expect(cls == kFast, 1)) {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                S0[j] = __umul24(hy[j], rs32) + (__umul24(hx[j], (uint32_t)PBs) + kOff);
                S1[j] = lx[j];
                S2[j] = ly[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                S0[j] = 0u;
                S1[j] = (uint32_t)fix_to_int<INTERP>(hx[j], lx[j]);
                S2[j] = (uint32_t)fix_to_int<INTERP>(hy[j], ly[j]);
            }
        }
        return cls;
    };

    // -- issue the row's tap loads (rows that are not FAST load the dummy window: the row loop keeps one shape)
    auto issue_s = [&](int cls, const uint32_t (&S0)[PPL], Bytes<WINB> (&t0)[PPL], Bytes<WINB> (&t1)[PPL]) {
        const bool f = cls == kFast;
        const uint8_t* b0 = f ? (kAligned ? frame_al : frame) : dummy;
        const uint32_t rs_eff = f ? rs32 : 0u;
        const uint8_t* b1 = b0 + rs_eff;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint32_t off = S0[j];
            if constexpr (kAligned && RS4) {  // second tap row: same window alignment, scalar base + row stride
                const uint32_t offa = off & ~3u;
                __builtin_memcpy(&t0[j], b0 + offa, WINB);
                __builtin_memcpy(&t1[j], b1 + offa, WINB);
            } else if constexpr (kAligned) {
                __builtin_memcpy(&t0[j], b0 + (off & ~3u), WINB);
                __builtin_memcpy(&t1[j], b0 + ((off + rs_eff) & ~3u), WINB);
            } else {
                __builtin_memcpy(&t0[j], b0 + off, LOADB);
                if (INTERP == kLinear) __builtin_memcpy(&t1[j], b1 + off, LOADB);
            }
        }
    };

    uint32_t* wtr = &s_tr[wave][0];
    // blend one pixel from its taps -- w0 / w1 = the LOADB bytes of the upper / lower tap row starting AT the left tap -- into
    // the wave's LDS row (pixel 64 j + lane of the segment)
    auto blend_put = [&](int j, const uint32_t (&w0)[NEED], const uint32_t (&w1)[NEED], uint32_t fx, uint32_t fy) __attribute__((always_inline)) {
        if constexpr (sizeof(T) == 1) {
            uint32_t px;
            if constexpr (INTERP == kNearest)
                px = C == 4 ? w0[0] : (w0[0] & ((1u << (8 * (C & 3))) - 1u));
            else if constexpr (C == 3)
                px = blend_u8_rgb_window(w0[0], w0[1], w1[0], w1[1], fx, fy);
            else if constexpr (C == 4)
                px = blend_u8_packed<C>(w0[0], w0[1], w1[0], w1[1], fx, fy);
            else
                px = blend_u8_packed<C>(w0[0], w0[0] >> (8 * C), w1[0], w1[0] >> (8 * C), fx, fy);
            wtr[64 * j + lane] = px;
        } else {
            const float* f0 = reinterpret_cast<const float*>(&w0[0]);
            const float* f1 = reinterpret_cast<const float*>(&w1[0]);
            float* wf = reinterpret_cast<float*>(wtr) + (64 * j + lane) * C;
            if (INTERP == kNearest) {
#pragma unroll
                for (int k = 0; k < C; k++) wf[k] = f0[k];
            } else {
                float w00, w01, w10, w11;
                weights_f32((int)fx, (int)fy, w00, w01, w10, w11);
#pragma unroll
                for (int k = 0; k < C; k++) wf[k] = blend_f32(f0[k], f0[k + C], f1[k], f1[k + C], w00, w01, w10, w11);
            }
        }
    };
    // FAST row of the gather path: taps from the registers the row's loads filled
    auto finish_s = [&](const uint32_t (&S0)[PPL], const uint32_t (&S1)[PPL], const uint32_t (&S2)[PPL], const Bytes<WINB> (&t0)[PPL],
                        const Bytes<WINB> (&t1)[PPL]) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint32_t fx = S1[j] >> 27, fy = S2[j] >> 27;  // (bilinear only)
            uint32_t w0[NEED], w1[NEED];
            if constexpr (kAligned) {
                const uint32_t sh0 = S0[j] << 3, sh1 = RS4 ? sh0 : (S0[j] + rs32) << 3;  // funnel-shift amounts (v_alignbit reads bits 4:0)
#pragma unroll
                for (int k = 0; k < NEED; k++) {
                    w0[k] = __builtin_amdgcn_alignbit(t0[j].w[k + 1], t0[j].w[k], sh0);
                    w1[k] = __builtin_amdgcn_alignbit(t1[j].w[k + 1], t1[j].w[k], sh1);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NEED; k++) {
                    w0[k] = t0[j].w[k];
                    w1[k] = INTERP == kLinear ? t1[j].w[k] : 0u;
                }
            }
            blend_put(j, w0, w1, fx, fy);
        }
    };
    auto put_px = [&](int j, const Pixel<T, C>& v) __attribute__((always_inline)) {
        if constexpr (sizeof(T) == 1) {
            wtr[64 * j + lane] = v.packed;
        } else {
            float* wf = reinterpret_cast<float*>(wtr) + (64 * j + lane) * C;
#pragma unroll
            for (int k = 0; k < C; k++) wf[k] = v.v[k];
        }
    };
    auto border_px = [&]() __attribute__((always_inline)) {
        Pixel<T, C> v;
        if constexpr (sizeof(T) == 1) {
            v.packed = C == 4 ? view.bu : (view.bu & ((1u << (8 * (C & 3))) - 1u));
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) v.v[k] = view.bf[k];
        }
        return v;
    };
    // OUT row: the border value
    auto fill_s = [&]() __attribute__((always_inline)) {
        const Pixel<T, C> v = border_px();
#pragma unroll
        for (int j = 0; j < PPL; j++) put_px(j, v);
    };
    // SLOW row: exact chain and guarded taps for each of the lane's pixels (same ownership, same store order)
    auto slow_s = [&](int y) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            int Xe, Ye;
            exact_px(y, j, Xe, Ye);
            put_px(j, sample_global<T, C, INTERP>(view, Xe, Ye));
        }
    };
    // EDGE row: unguarded loads + the fast blend for the pixels whose taps are inside, the border value for those whose
    // taps are all outside, guarded taps for the few in between
    auto edge_s = [&](const uint32_t (&S1)[PPL], const uint32_t (&S2)[PPL]) __attribute__((always_inline)) {
        Bytes<LOADB> e0[PPL], e1[PPL];
        const uint8_t* frame_r1 = frame + rs32;
        bool inb[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const int sx = (int)S1[j] >> SH, sy = (int)S2[j] >> SH;
            inb[j] = any_unguarded && (uint32_t)sx <= sx_max && (uint32_t)sy <= sy_max;
            if (inb[j]) {
                const uint32_t off = (uint32_t)sy * rs32 + (uint32_t)sx * (uint32_t)PBs;
                __builtin_memcpy(&e0[j], frame + off, LOADB);
                if (INTERP == kLinear) __builtin_memcpy(&e1[j], frame_r1 + off, LOADB);
            } else {
                __builtin_memcpy(&e0[j], dummy, LOADB);
                if (INTERP == kLinear) __builtin_memcpy(&e1[j], dummy, LOADB);
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const int X = (int)S1[j], Y = (int)S2[j];
            const int sx = X >> SH, sy = Y >> SH;
            const uint32_t fx = (uint32_t)X & 31u, fy = (uint32_t)Y & 31u;
            constexpr int kTap = INTERP == kLinear ? 1 : 0;  // taps reach sx + kTap, sy + kTap
            const bool all_out = fill_ok && (sx < -kTap || sx >= a.src_w || sy < -kTap || sy >= a.src_h);
            Pixel<T, C> v;
            if (inb[j]) {
                if constexpr (sizeof(T) == 1) {
                    if (INTERP == kNearest)
                        v.packed = C == 4 ? e0[j].w[0] : (e0[j].w[0] & ((1u << (8 * (C & 3))) - 1u));
                    else if constexpr (C == 3)
                        v.packed = blend_u8_rgb_window(e0[j].w[0], e0[j].w[1], e1[j].w[0], e1[j].w[1], fx, fy);
                    else if constexpr (C == 4)
                        v.packed = blend_u8_packed<C>(e0[j].w[0], e0[j].w[1], e1[j].w[0], e1[j].w[1], fx, fy);
                    else
                        v.packed = blend_u8_packed<C>(e0[j].w[0], e0[j].w[0] >> (8 * C), e1[j].w[0], e1[j].w[0] >> (8 * C), fx, fy);
                } else {
                    const float* f0 = reinterpret_cast<const float*>(&e0[j]);
                    const float* f1 = reinterpret_cast<const float*>(&e1[j]);
                    float w00 = 0, w01 = 0, w10 = 0, w11 = 0;
                    if (INTERP == kLinear) weights_f32((int)fx, (int)fy, w00, w01, w10, w11);
#pragma unroll
                    for (int k = 0; k < C; k++) v.v[k] = INTERP == kNearest ? f0[k] : blend_f32(f0[k], f0[k + C], f1[k], f1[k + C], w00, w01, w10, w11);
                }
            } else if (all_out) {
                v = border_px();
            } else {
                v = sample_global<T, C, INTERP>(view, X, Y);
            }
            put_px(j, v);
        }
    };

    // -- LDS row -> registers in store order (u8: pixels 4 l .. 4 l + 3 of the segment; float: 16-byte unit u * 64 + l)
    constexpr int kVec = sizeof(T) == 1 ? 64 : TRW / 4;  // 16-byte units in the wave's row segment
    constexpr int NQ = (kVec + 63) / 64;
    auto read_back = [&](uint4 (&out)[NQ]) {
        asm volatile("" ::: "memory");  // compiler fence: one wave's LDS operations execute in program order
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int q = u * 64 + lane;
            if (q < kVec) out[u] = reinterpret_cast<const uint4*>(wtr)[q];
        }
        asm volatile("" ::: "memory");  // (the next row's LDS writes cannot pass these reads)
    };
    auto finish_any = [&](int cls, int y, const uint32_t (&S0)[PPL], const uint32_t (&S1)[PPL], const uint32_t (&S2)[PPL], const Bytes<WINB> (&t0)[PPL],
                          const Bytes<WINB> (&t1)[PPL]) {
        if (__builtin_expect(cls == kFast, 1)) {
            finish_s(S0, S1, S2, t0, t1);
        } else {
            if (cls == kOut)
                fill_s();
            else if (cls == kEdge)
                edge_s(S1, S2);
            else
                slow_s(y);
        }
    };

    // -- stores.  The destination is written once and never read back by this kernel: non-temporal stores keep it from
    // displacing source lines in L2 / MALL.  Lanes of a ragged last tile (and every lane when the destination's layout
    // does not admit the wide stores) fall back to element stores.
    const int seg_px = min(TW, a.dst_w - x0);                         // valid pixels of this tile's row segments (> 0)
    const int lane_px = max(0, min(PPL, seg_px - lane * PPL));        // 8-bit: valid pixels of this lane's store unit
    const bool lane_vec = a.dst_vec_ok && lane_px == PPL;
    auto store_s = [&](int y, const uint4 (&out)[NQ]) {
        if constexpr (sizeof(T) == 1) {  // the lane's 4 pixels = 4 C contiguous bytes, one instruction
            const uint32_t p[4] = {out[0].x, out[0].y, out[0].z, out[0].w};
            if constexpr (PLANAR) {  // float planes: one 16-byte store per channel
                uint8_t* dp = dframe + (int64_t)y * a.dst_rs + (int64_t)(x0 + lane * PPL) * 4;
#pragma unroll
                for (int k = 0; k < C; k++) {
                    const float sc = a.pscale[k], bi = a.pbias[k];
                    f32x4 o = {(float)((p[0] >> (8 * k)) & 0xffu) * sc + bi, (float)((p[1] >> (8 * k)) & 0xffu) * sc + bi,
                               (float)((p[2] >> (8 * k)) & 0xffu) * sc + bi, (float)((p[3] >> (8 * k)) & 0xffu) * sc + bi};
                    float* dk = reinterpret_cast<float*>(dp + k * a.dst_ps);
                    if (__builtin_expect(lane_vec, 1)) {
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dk));
                    } else {
                        for (int i = 0; i < lane_px; i++) dk[i] = o[i];
                    }
                }
                return;
            }
            uint8_t* d = dframe + (int64_t)y * a.dst_rs + (int64_t)(x0 + lane * PPL) * C;
            if (__builtin_expect(lane_vec, 1)) {
                if constexpr (C == 1) {
                    __builtin_nontemporal_store(p[0] | (p[1] << 8) | (p[2] << 16) | (p[3] << 24), reinterpret_cast<uint32_t*>(d));
                } else if constexpr (C == 2) {
                    u32x2 o = {p[0] | (p[1] << 16), p[2] | (p[3] << 16)};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(d));
                } else if constexpr (C == 3) {
                    u32x3 o = {p[0] | (p[1] << 24), (p[1] >> 8) | (p[2] << 16), (p[2] >> 16) | (p[3] << 8)};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x3*>(d));
                } else {
                    u32x4 o = {p[0], p[1], p[2], p[3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(d));
                }
            } else {
                for (int i = 0; i < lane_px; i++)
#pragma unroll
                    for (int k = 0; k < C; k++) d[i * C + k] = (uint8_t)(p[i] >> (8 * k));
            }
        } else {
            float* drow = reinterpret_cast<float*>(dframe + (int64_t)y * a.dst_rs) + (int64_t)x0 * C;  // the wave's row segment
            const int nfl = seg_px * C;  // valid floats of the segment
#pragma unroll
            for (int u = 0; u < NQ; u++) {
                const int q = u * 64 + lane;
                if (q >= kVec) continue;
                if (__builtin_expect(a.dst_vec_ok && 4 * q + 4 <= nfl, 1)) {
                    u32x4 o = {out[u].x, out[u].y, out[u].z, out[u].w};
                    __builtin_nontemporal_store(o, &reinterpret_cast<u32x4*>(drow)[q]);
                } else {
                    const uint32_t f[4] = {out[u].x, out[u].y, out[u].z, out[u].w};
                    for (int i = 0; i < 4 && 4 * q + i < nfl; i++) reinterpret_cast<uint32_t*>(drow)[4 * q + i] = f[i];
                }
            }
        }
    };

    // ===== staged tiles ================================================================================
    // A tile is processed in groups of 4 rows (one per wave).  A group all of whose pixels sample inside the frame (its
    // corner pixels do, W of one sign: the image of the group is the convex quadrilateral of its corners) and whose
    // source box fits the LDS ring is STAGED: box(g + 1) streams into LDS while the waves blend group g from LDS.  The
    // other groups of the tile take the row classes directly.  One barrier per group.
    uint32_t ok_groups = 0;  // bit g: group g of this tile is staged
    if constexpr (kStage) {
        const int ng = (y_last - y0) / kWaves + 1;  // groups of this tile (tile_h / 4 <= 16: one lane per group corner)
        if (a.src_stage_ok && can_fast && ng <= 16) {
            // -- corner k of group g in lane 4 g + k: the fast chain for one pixel
            const int cg = lane >> 2, ck = lane & 3;
            const bool lane_on = cg < ng;
            const int cyy = min(y0 + cg * kWaves + ((ck & 2) ? kWaves - 1 : 0), y_last);
            const double cdx = (ck & 1) ? (double)(TW - 1) : 0.0, cdy = (double)cyy;
            const double cW = __builtin_fma(RW, cdy, CW) + m6 * cdx;
            const double cr = rcp_newton(cW);
            const double ctx = __builtin_fma(__builtin_fma(RX, cdy, CX) + (m0 * kTwo32) * cdx, cr, F::kMagic);
            const double cty = __builtin_fma(__builtin_fma(RY, cdy, CY) + (m3 * kTwo32) * cdx, cr, F::kMagic);
            const int csx = (int)((uint32_t)__double2hiint(ctx) - kHiBias), csy = (int)((uint32_t)__double2hiint(cty) - kHiBias);
            const uint32_t cwh = (uint32_t)__double2hiint(cW);
            const bool c_in = (uint32_t)(csx - kM) <= (uint32_t)(sxw_lim - 2 * kM) && (uint32_t)(csy - kM) <= (uint32_t)(sy_lim - 2 * kM) &&
                              ((cwh >> 20) & 0x7ffu) - 824u <= 398u;
            // min / max over the 4 corners of a group (a quad of lanes): two DPP quad permutes each
            auto quad_min = [](int v) {
                v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
                return min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
            };
            const int gx0 = quad_min(csx) - 1, gx1 = -quad_min(-csx) + 1, gy0 = quad_min(csy) - 1, gy1 = -quad_min(-csy) + 1;  // +-1 px: the exact
            // chain may move a coordinate by one unit; taps reach one pixel further right / down
            constexpr int kTapPx = INTERP == kLinear ? 1 : 0;
            const int g_rows = gy1 + kTapPx + 1 - gy0;                    // source rows gy0 .. gy1 + kTapPx
            const int g_b0 = (gx0 * PBs) & ~15;                           // first staged byte of a row (16-byte aligned: rows are)
            const int g_end = (gx1 + kTapPx + 1) * PBs;                   // one byte past the last tap
            const int g_chunks = (g_end - g_b0 + 15) >> 4;                // 16-byte chunks per row
            // the last chunk may read up to 15 bytes past its row: it must stay inside the frame's allocation
            const bool g_tail_ok = (int64_t)(gy1 + kTapPx) * a.src_rs + g_b0 + 16 * (int64_t)g_chunks <= (int64_t)(a.src_h - 1) * a.src_rs + (int64_t)a.src_w * PBs;
            const bool g_fit = gx0 >= 0 && gy0 >= 0 && g_rows <= kStageRows && g_chunks * 16 <= kPitch && g_tail_ok;
            // a group can be staged when its four corners sample inside, its box fits, and W has one sign over it; bit g of
            // ok_groups says so (the four lanes of a quad vote)
            const uint64_t in_mask = __ballot(lane_on && c_in && g_fit), neg_mask = __ballot(lane_on && (cwh >> 31));
            for (int g = 0; g < ng; g++) {
                const uint32_t q_in = (uint32_t)(in_mask >> (4 * g)) & 0xFu, q_neg = (uint32_t)(neg_mask >> (4 * g)) & 0xFu;
                if (q_in == 0xFu && (q_neg == 0u || q_neg == 0xFu)) ok_groups |= 1u << g;
            }
            if (ok_groups != 0) {
                auto group_i = [](int v, int g) { return __builtin_amdgcn_readlane(v, 4 * g); };
                auto staged_group = [&](int g) { return g < ng && ((ok_groups >> g) & 1u) != 0; };
                // -- stage(g): rows of the box are dealt to the waves; lane i moves chunk i (+ 64, ...) of its row
                auto stage = [&](int g) __attribute__((always_inline)) {
                    const int r0 = group_i(gy0, g), nr = group_i(g_rows, g), b0 = group_i(g_b0, g), nch = group_i(g_chunks, g);
                    uint8_t* buf = s_stage + (g & 1) * kBufBytes;
                    for (int rr = wave; rr < nr; rr += kWaves) {
                        const uint8_t* grow = frame + (int64_t)(r0 + rr) * a.src_rs + b0 + lane * 16;
                        uint8_t* lrow = buf + rr * kPitch;
#pragma unroll
                        for (int c0 = 0; c0 < kPitch / 16; c0 += 64) {
                            if (c0 < nch && lane + c0 < nch)
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(grow + c0 * 16),
                                                                 (__attribute__((address_space(3))) void*)(lrow + c0 * 16), 16, 0, 0);
                        }
                    }
                };
                uint4 out[NQ];
                int y_prev = -1;
                if (staged_group(0)) stage(0);
                for (int g = 0; g < ng; g++) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of box g has landed (and row g - 2 is stored)
                    __builtin_amdgcn_s_barrier();                      // ... and everybody else's; every wave is done reading box g - 1
                    if (staged_group(g + 1)) stage(g + 1);
                    if (y_prev >= 0) read_back(out);  // row g - 1 out of the transposition row (written a whole group ago: no LDS latency)
                    const int y = y0 + g * kWaves + wave;
                    const bool on = y <= y_last && staged_group(g);  // (the other groups follow below)
                    uint32_t hx[PPL], lx[PPL], hy[PPL], ly[PPL], wf_, wl_;
                    uint32_t raw0[PPL][NDW], raw1[PPL][NDW], o[PPL];
                    if (on) {
                        if (chain(y, hx, lx, hy, ly, wf_, wl_) == 0) fix_ties(y, hx, lx, hy, ly);
                        // LDS byte offset of a pixel's left tap, from the high dwords: (sy - r0) * kPitch + sx * PBs - b0
                        const uint32_t kL = (uint32_t)(-(group_i(gy0, g) * kPitch) - group_i(g_b0, g)) - 0x380000u * (uint32_t)(kPitch + PBs);
                        const uint8_t* buf = s_stage + (g & 1) * kBufBytes;
#pragma unroll
                        for (int j = 0; j < PPL; j++) {
                            o[j] = __umul24(hy[j], (uint32_t)kPitch) + (__umul24(hx[j], (uint32_t)PBs) + kL);
                            const uint32_t* p0 = reinterpret_cast<const uint32_t*>(buf + (kFunnel ? (o[j] & ~3u) : o[j]));
#pragma unroll
                            for (int k = 0; k < NDW; k++) {
                                raw0[j][k] = p0[k];
                                if (INTERP == kLinear) raw1[j][k] = p0[k + kPitch / 4];
                            }
                        }
                    }
                    if (y_prev >= 0) store_s(y_prev, out);  // behind the DMA and the tap reads: a whole group of arithmetic to complete
                    y_prev = -1;
                    if (on) {
#pragma unroll
                        for (int j = 0; j < PPL; j++) {
                            uint32_t w0[NEED], w1[NEED];
#pragma unroll
                            for (int k = 0; k < NEED; k++) {
                                if constexpr (kFunnel) {
                                    w0[k] = __builtin_amdgcn_alignbit(raw0[j][k + 1], raw0[j][k], o[j] << 3);
                                    w1[k] = INTERP == kLinear ? __builtin_amdgcn_alignbit(raw1[j][k + 1], raw1[j][k], o[j] << 3) : 0u;
                                } else {
                                    w0[k] = raw0[j][k];
                                    w1[k] = INTERP == kLinear ? raw1[j][k] : 0u;
                                }
                            }
                            blend_put(j, w0, w1, lx[j] >> 27, ly[j] >> 27);
                        }
                        y_prev = y;
                    }
                }
                if (y_prev >= 0) read_back(out);
                if (y_prev >= 0) store_s(y_prev, out);
            }
        }
    }

    // -- kStaged: the rows of the groups that were not staged, one at a time (load -> wait -> blend -> store).  The waves
    // of the other workgroups on the CU cover the exposed latency; what matters is that this path fits the staged loop's
    // register budget.
    if constexpr (kStage) {
#ifdef BEVWARP_EXP_NODIRECT
        return;
#endif
        uint32_t A0[PPL], A1[PPL], A2[PPL];
        Bytes<WINB> u0[PPL], u1[PPL];
        uint4 out[NQ];
        int y = y0 + wave;
        if (y > y_last) return;
        if (ok_groups == 0) {  // (the probe of the gather loop below: a wave whose rows all lie beyond one frame edge)
            if (coords_s(y, A0, A1, A2) == kOut && y + kWaves <= y_last) {
                const int side0 = out_side;
                const int y_probe = y + ((y_last - y) / kWaves) * kWaves;
                if (coords_s(y_probe, A0, A1, A2) == kOut && out_side == side0) {
                    fill_s();
                    read_back(out);
                    for (; y <= y_last; y += kWaves) store_s(y, out);
                    return;
                }
            }
        }
        for (int g = 0; y <= y_last; g++, y += kWaves) {
            if ((ok_groups >> g) & 1u) continue;
            const int cls = coords_s(y, A0, A1, A2);
            issue_s(cls, A0, u0, u1);
            finish_any(cls, y, A0, A1, A2, u0, u1);
            read_back(out);
            store_s(y, out);
        }
        return;
    }
    // -- the row loop.  Rows of a tile are dealt to its waves round-robin: neighbouring rows share source lines and run
    // at the same time.
    constexpr int RSTEP = kWaves;
    int yf = y0 + wave;
    const int y_end = y_last;
    if (yf > y_end) return;
    uint32_t A0[PPL], A1[PPL], A2[PPL], B0[PPL], B1[PPL], B2[PPL];  // row state: current / next
    Bytes<WINB> u0[PPL], u1[PPL];
    uint4 out[NQ];
    // -- interior tiles.  When the tile's four corner pixels sample inside the frame by the FAST margin with W of one sign,
    // the tile maps into the convex quadrilateral of their images: every row is FAST and the loop needs no row classes --
    // no end-pixel read-out, no scalar decisions, row terms advanced by one addition each.
    bool tile_in;
    {
        const int ck = lane & 3;
        const double cdx = (ck & 1) ? (double)(TW - 1) : 0.0, cdy = (double)((ck & 2) ? y_last : y0);
        const double cW = __builtin_fma(RW, cdy, CW) + m6 * cdx;
        const double cr = rcp_newton(cW);
        const double ctx = __builtin_fma(__builtin_fma(RX, cdy, CX) + (m0 * kTwo32) * cdx, cr, F::kMagic);
        const double cty = __builtin_fma(__builtin_fma(RY, cdy, CY) + (m3 * kTwo32) * cdx, cr, F::kMagic);
        const int csx = (int)((uint32_t)__double2hiint(ctx) - kHiBias), csy = (int)((uint32_t)__double2hiint(cty) - kHiBias);
        const uint32_t cwh = (uint32_t)__double2hiint(cW);
        // (one pixel more than the rows' own margin: the exact chain may move a coordinate by a unit)
        const bool c_in = (uint32_t)(csx - kM - 1) <= (uint32_t)(sxw_lim - 2 * kM - 2) && (uint32_t)(csy - kM - 1) <= (uint32_t)(sy_lim - 2 * kM - 2) &&
                          ((cwh >> 20) & 0x7ffu) - 824u <= 398u;
        const uint32_t in4 = (uint32_t)__ballot(c_in) & 0xFu, neg4 = (uint32_t)__ballot((cwh >> 31) != 0) & 0xFu;
        tile_in = can_fast && sxw_lim >= 2 * kM + 2 && sy_lim >= 2 * kM + 2 && in4 == 0xFu && (neg4 == 0u || neg4 == 0xFu);
    }
    if (tile_in) {
        double UX = __builtin_fma(RX, (double)yf, CX), UY = __builtin_fma(RY, (double)yf, CY), UW = __builtin_fma(RW, (double)yf, CW);
        const double SX = uniform_f64(RX * (double)RSTEP), SY = uniform_f64(RY * (double)RSTEP), SW = uniform_f64(RW * (double)RSTEP);
        // coordinates of the NEXT row of this wave (each call advances the row terms)
        int y_next = yf;
        auto coords_f = [&](uint32_t (&S0)[PPL], uint32_t (&S1)[PPL], uint32_t (&S2)[PPL]) __attribute__((always_inline)) {
            uint32_t hx[PPL], lx[PPL], hy[PPL], ly[PPL], wf_, wl_;
            if (chain_u(UX, UY, UW, hx, lx, hy, ly, wf_, wl_) == 0) fix_ties(y_next, hx, lx, hy, ly);
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                S0[j] = __umul24(hy[j], rs32) + (__umul24(hx[j], (uint32_t)PBs) + kOff);
                S1[j] = lx[j];
                S2[j] = ly[j];
            }
            UX += SX;
            UY += SY;
            UW += SW;
            y_next += RSTEP;
        };
        // one step: row yf's pixels leave the LDS row, the next row's loads are issued from state C, the row after that
        // gets its coordinates into state N, row yf is stored, the next row is blended
        bool more = yf + RSTEP <= y_end;
        auto step = [&](uint32_t (&C0)[PPL], uint32_t (&C1)[PPL], uint32_t (&C2)[PPL], uint32_t (&N0)[PPL], uint32_t (&N1)[PPL], uint32_t (&N2)[PPL])
                        __attribute__((always_inline)) {
            read_back(out);
            issue_s(kFast, C0, u0, u1);
            const int y_done = yf;
            yf += RSTEP;
            more = yf + RSTEP <= y_end;
            if (more) coords_f(N0, N1, N2);
            store_s(y_done, out);
            finish_s(C0, C1, C2, u0, u1);
        };
        coords_f(A0, A1, A2);
        issue_s(kFast, A0, u0, u1);
        if (more) coords_f(B0, B1, B2);
        finish_s(A0, A1, A2, u0, u1);
        while (more) {  // (two steps per trip: the row states swap roles instead of being copied)
            step(B0, B1, B2, A0, A1, A2);
            if (!more) break;
            step(A0, A1, A2, B0, B1, B2);
        }
        read_back(out);
        store_s(yf, out);
        return;
    }
#ifdef BEVWARP_DIRECT_EDGE
    {
        int cls = coords_s(yf, A0, A1, A2);
        if (cls == kOut && yf + RSTEP <= y_end) {
            const int side0 = out_side;
            const int y_probe = yf + ((y_end - yf) / RSTEP) * RSTEP;
            if (coords_s(y_probe, B0, B1, B2) == kOut && out_side == side0) {
                fill_s();
                read_back(out);
                for (int y = yf; y <= y_end; y += RSTEP) store_s(y, out);
                return;
            }
        }
        for (;;) {
            issue_s(cls, A0, u0, u1);
            finish_any(cls, yf, A0, A1, A2, u0, u1);
            read_back(out);
            store_s(yf, out);
            yf += RSTEP;
            if (yf > y_end) return;
            cls = coords_s(yf, A0, A1, A2);
        }
    }
#endif
    int cls_c = coords_s(yf, A0, A1, A2), cls_n = kSlow;
    if (cls_c == kOut && yf + RSTEP <= y_end) {
        // The wave's first row lies beyond a frame edge: probe its last row.  When that one lies beyond the same edge
        // with W of the same sign, the rows between them map into the convex hull of the two segments and see nothing
        // of the frame either: fill them without computing another coordinate.  (Footprints like the Brno BEV have a
        // third of their rows outside; a wave whose first row is inside never pays for this.)
        const int side0 = out_side;
        const int y_probe = yf + ((y_end - yf) / RSTEP) * RSTEP;
        if (coords_s(y_probe, B0, B1, B2) == kOut && out_side == side0) {
            fill_s();
            read_back(out);
            for (int y = yf; y <= y_end; y += RSTEP) store_s(y, out);
            return;
        }
    }
    issue_s(cls_c, A0, u0, u1);
    // Order inside an iteration: the previous row's pixels out of the LDS row (written an iteration ago: no LDS latency
    // on the path), the next row's loads, the next-but-one row's coordinates, THEN the previous row's store, then the
    // blend.  vmcnt retires in issue order, so a store issued before a row's loads would have to reach L2 before that
    // row's taps can be used; issued after them it has a whole iteration to complete.
    bool more = yf + RSTEP <= y_end;
    if (more) cls_n = coords_s(yf + RSTEP, B0, B1, B2);
    finish_any(cls_c, yf, A0, A1, A2, u0, u1);
#ifdef BEVWARP_CLOCK
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime(), phase_[4] = {0, 0, 0, 0};
#endif
    while (more) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            A0[j] = B0[j];
            A1[j] = B1[j];
            A2[j] = B2[j];
        }
        cls_c = cls_n;
        read_back(out);              // row yf
        issue_s(cls_c, A0, u0, u1);  // row yf + RSTEP
        STAMP(0);
        const int y_done = yf;
        yf += RSTEP;
        more = yf + RSTEP <= y_end;
        if (more) cls_n = coords_s(yf + RSTEP, B0, B1, B2);  // overlaps with the loads in flight
        STAMP(1);
        store_s(y_done, out);
        STAMP(2);
        finish_any(cls_c, yf, A0, A1, A2, u0, u1);
        STAMP(3);
    }
    read_back(out);
    store_s(yf, out);
#ifdef BEVWARP_CLOCK
    if (threadIdx.x == 0)
        for (int i = 0; i < 4; i++) atomicAdd(&g_clk[4 + i], phase_[i]);
#endif
}

// Footprint: mark every in-bounds source pixel any tap would read (measurement aid; exact chain).
template <int INTERP>
__global__ void footprint_kernel(unsigned char* __restrict__ touched, int batch, int src_h, int src_w, int dst_h, int dst_w,
                                 const double* __restrict__ minv, int m_stride, int bw0) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= dst_w) return;
    const double* M = minv + (int64_t)b * m_stride;
    double Mr[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Mr[i] = M[i];
    const int bx = (x / bw0) * bw0;
    double X0, Y0, W0;
    row_terms(Mr, bx, y, X0, Y0, W0);
    const double x1 = (double)(x - bx);
    int X, Y;
    map_pixel_exact<INTERP>(X0 + Mr[0] * x1, Y0 + Mr[3] * x1, W0 + Mr[6] * x1, X, Y);
    const int sx = INTERP == kLinear ? (X >> kInterBits) : X, sy = INTERP == kLinear ? (Y >> kInterBits) : Y;
    unsigned char* tb = touched + (int64_t)b * src_h * src_w;
    const int ntap = INTERP == kLinear ? 2 : 1;
    for (int dy = 0; dy < ntap; dy++)
        for (int dx = 0; dx < ntap; dx++) {
            const int px = sx + dx, py = sy + dy;
            if ((unsigned)px < (unsigned)src_w && (unsigned)py < (unsigned)src_h) tb[(int64_t)py * src_w + px] = 1;
        }
}

template <typename T, int C, int INTERP, int MODE>
void launch_mode(const WarpArgs& a, dim3 grid, hipStream_t stream) {
    constexpr bool kRgb8Lin = sizeof(T) == 1 && C == 3 && INTERP == kLinear;
    if constexpr (sizeof(T) == 1) {
        if (a.planar) {
            if (kRgb8Lin && a.src_rs % 4 == 0)
                hipLaunchKernelGGL((warp_rows<T, C, INTERP, kRgb8Lin, true, MODE>), grid, dim3(kWG), 0, stream, a);
            else
                hipLaunchKernelGGL((warp_rows<T, C, INTERP, false, true, MODE>), grid, dim3(kWG), 0, stream, a);
            return;
        }
    }
    if (kRgb8Lin && a.src_rs % 4 == 0)
        hipLaunchKernelGGL((warp_rows<T, C, INTERP, kRgb8Lin, false, MODE>), grid, dim3(kWG), 0, stream, a);
    else
        hipLaunchKernelGGL((warp_rows<T, C, INTERP, false, false, MODE>), grid, dim3(kWG), 0, stream, a);
}

template <typename T, int C, int INTERP>
void launch_tci(const WarpArgs& a, dim3 grid, hipStream_t stream) {
    if constexpr (has_staged_kernel<T, C, INTERP>()) {
#ifndef BEVWARP_FORCE_GATHER
        if (a.src_stage_ok) {
            launch_mode<T, C, INTERP, kStaged>(a, grid, stream);
            return;
        }
#endif
    }
    launch_mode<T, C, INTERP, kGather>(a, grid, stream);
}

template <typename T>
void launch_t(const WarpArgs& a, int channels, int interp, dim3 grid, hipStream_t stream) {
#define BEVWARP_CASE(C)                                       \
    case C:                                                   \
        if (interp == kNearest)                               \
            launch_tci<T, C, kNearest>(a, grid, stream);      \
        else                                                  \
            launch_tci<T, C, kLinear>(a, grid, stream);       \
        break;
    switch (channels) {
        BEVWARP_CASE(1)
        BEVWARP_CASE(2)
        BEVWARP_CASE(3)
        default:
            BEVWARP_CASE(4)
    }
#undef BEVWARP_CASE
}

}  // namespace

#ifdef BEVWARP_CLOCK
hipError_t debug_read_clock(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_clk), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z));
    }
    return e;
}
#endif

int tile_width(int dtype) { return 64 * (dtype == 0 ? pixels_per_lane<uint8_t>() : pixels_per_lane<float>()); }
int rows_per_pass() { return kWaves; }

hipError_t launch_warp(const WarpArgs& a, int dtype, int channels, int interp, hipStream_t stream) {
    (void)hipGetLastError();  // a stale error left by the host framework is not this call's
    const dim3 grid((unsigned)(8 * a.chunk));
    if (dtype == 0)
        launch_t<uint8_t>(a, channels, interp, grid, stream);
    else
        launch_t<float>(a, channels, interp, grid, stream);
    return hipGetLastError();
}

hipError_t launch_footprint(unsigned char* touched, int batch, int src_h, int src_w, int dst_h, int dst_w, const double* minv,
                            int m_stride, int bw0, int interp, hipStream_t stream) {
    (void)hipGetLastError();
    const dim3 block(256), grid((dst_w + 255) / 256, dst_h, batch);
    if (interp == kNearest)
        hipLaunchKernelGGL(footprint_kernel<kNearest>, grid, block, 0, stream, touched, batch, src_h, src_w, dst_h, dst_w, minv, m_stride, bw0);
    else
        hipLaunchKernelGGL(footprint_kernel<kLinear>, grid, block, 0, stream, touched, batch, src_h, src_w, dst_h, dst_w, minv, m_stride, bw0);
    return hipGetLastError();
}

}  // namespace bevwarp
