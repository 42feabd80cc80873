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
// Ownership of the 64 x PPL pixels a wave computes per pass (a compile-time tag of the code that depends on it):
//   RowSeg  one row segment of 64 PPL pixels: pixel j of lane l is (x0 + 64 j + l, y).  A pass reads two source rows of an
//           axis-aligned map: the interior loop.
//   BlkSeg  a block of 64 x PPL pixels: pixel j of lane l is (xb + l, y + j), xb = the wave's 64-pixel column strip of the
//           tile.
//   PatSeg  the same blocks and passes, other lanes: they form a PATCH of (64 / PPL) x PPL pixels and pixel j of lane l is
//           (xb + (64 / PPL) j + l % (64 / PPL), y + l / (64 / PPL)): one gather instruction covers 16 x 4 (8-bit) /
//           32 x 2 (float) destination pixels instead of 64 x 1, which halves and better the source rows -- cache lines --
//           it runs through when the footprint is turned (25 degrees: -22 %, 45 degrees: -31 %; unturned: +8 %).  Tiles that the frame's edge crosses are cut into these: the edge then runs through a quarter as many
//           passes, and only those pay for guarded taps.
struct RowSeg {
    static constexpr bool blk = false, pat = false;
};
struct BlkSeg {
    static constexpr bool blk = true, pat = false;
};
struct PatSeg {
    static constexpr bool blk = true, pat = true;
};
// where the exact chain is instantiated: inside a row loop (its matrix loads must stay in the rare branch) or after one
struct InLoop {
    static constexpr bool value = true;
};
struct InTail {
    static constexpr bool value = false;
};

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

// the wide destination store of a pass (experiment switch: non-temporal)
#ifndef BEVWARP_NT_STORES
#define BEVWARP_NT_STORES 0
#endif
template <typename V>
__device__ __forceinline__ void wide_store(V* p, const V& v) {
    if (BEVWARP_NT_STORES)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
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
// at the left tap (left pixel = bytes 0 1 2, right pixel = bytes 3 4 5).  VERTICAL FIRST: the byte selects unpack channel c
// of both rows into (L.c, R.c) u16 pairs, v_pk_mul_lo_u16 + v_pk_mad_u16 blend the two rows of both taps at once
// (wy0 t + wy1 b <= 8160), and one v_dot2_u32_u16 per channel does the horizontal sum with the rounding constant, the
// weights scaled by 64 so that the result byte is bits 16..23: 6 perm + 3 mul + 3 mad + 3 dot2 + 2 perm = 17, against 19
// for horizontal-first (4 perm + 6 dot4 + 3 lshl_or + 3 dot2 + 2 perm + one more weight).
__device__ __forceinline__ uint32_t blend_u8_rgb_window(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t fx, uint32_t fy) {
    const u16x2 wx = __builtin_bit_cast(u16x2, fx * 0x3fffc0u + 2048u);  // halves (2048 - 64 fx, 64 fx)
    const unsigned short wy1 = (unsigned short)fy, wy0 = (unsigned short)(32u - fy);
    const u16x2 wy0p = {wy0, wy0}, wy1p = {wy1, wy1};
    uint32_t s[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const uint32_t sel = 0x0c030c00u + 0x00010001u * c;  // (byte c, 0, byte c + 3, 0)
        const u16x2 t = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(a1, a0, sel));
        const u16x2 b = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b1, b0, sel));
        s[c] = __builtin_amdgcn_udot2(t * wy0p + b * wy1p, wx, 32768u, false);
    }
    return __builtin_amdgcn_perm(s[2], __builtin_amdgcn_perm(s[1], s[0], 0x0c0c0602u), 0x0c060100u);
}

// bw_mode of the composite (bev/tool/compo.py:13-14): the foreground is cv2.cvtColor(BGR2GRAY -> GRAY2BGR)'d BEFORE it is warped, so
// the TAPS are converted -- OpenCV's 14-bit fixed point (1868 B + 9617 G + 4899 R + 8192) >> 14 on a pixel packed B, G, R in bytes
// 0, 1, 2 (restated from OpenCV's colour conversion; parity unpinned) -- then one channel is blended and replicated.
__device__ __forceinline__ uint32_t gray_of(uint32_t p) {
    return ((p & 0xffu) * 1868u + ((p >> 8) & 0xffu) * 9617u + ((p >> 16) & 0xffu) * 4899u + 8192u) >> 14;
}
__device__ __forceinline__ uint32_t blend_u8_gray_window(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t fx, uint32_t fy) {
    const uint32_t g = blend_u8(gray_of(a0), gray_of(__builtin_amdgcn_alignbit(a1, a0, 24)), gray_of(b0), gray_of(__builtin_amdgcn_alignbit(b1, b0, 24)), fx, fy);
    return g * 0x010101u;
}

// What the guarded sampler needs of the source frame, by value (taking the address of the kernel-argument struct would
// push it to scratch).
struct SrcView {
    const uint8_t* frame;
    int64_t rs;
    int w, h;
    float bf[4];
    uint32_t bu;  // border bytes packed
    bool gray;    // (composite, bw_mode) 8-bit BGR taps are converted to grey before they are blended
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
    if constexpr (sizeof(T) == 1 && C == 3) {
        if (a.gray) {
            auto tap = [&](const T (&t)[C], bool in) {
                return in ? ((uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16)) : (a.bu & 0xffffffu);
            };
            out.packed = blend_u8(gray_of(tap(t00, xin0 && yin0)), gray_of(tap(t01, xin1 && yin0)), gray_of(tap(t10, xin0 && yin1)), gray_of(tap(t11, xin1 && yin1)), fx, fy) * 0x010101u;
            return out;
        }
    }
    float w00 = 0, w01 = 0, w10 = 0, w11 = 0;
    if constexpr (sizeof(T) == 4) weights_f32(fx, fy, w00, w01, w10, w11);
    // all four taps outside: the reference stores the border value itself (for 8-bit pixels the blend gives it back anyway)
    const bool all_out = sx >= a.w || sx + 1 < 0 || sy >= a.h || sy + 1 < 0;
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
            out.v[k] = all_out ? b : blend_f32(v00, v01, v10, v11, w00, w01, w10, w11);
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
//   PLANAR  the destination is C float32 planes, dst[c][y][x] = float(pixel) * pscale[c] + pbias[c] (the layout a detector takes)
// Register budget: 4 waves per SIMD -- what the FAST row loop needs; the rare row classes may spill.
// ===================================================================================================
#ifndef BEVWARP_U8LIN_WAVES
#define BEVWARP_U8LIN_WAVES 4
#endif
// Diagnostic build only (-DBEVWARP_CLOCK, tools/clock.py): wave 0 of every workgroup adds the shader-clock ticks
// (s_memtime) and the 100 MHz reference ticks (s_memrealtime) it lived for; their ratio is the clock the chip held.
#ifdef BEVWARP_CLOCK
__device__ unsigned long long g_clk[4];
#endif
#ifndef BEVWARP_DEEP_ALL
#define BEVWARP_DEEP_ALL 0  // (experiment: several passes in flight for every format)
#endif
#ifndef BEVWARP_DEPTH
#define BEVWARP_DEPTH 6
#endif
#ifndef BEVWARP_STRAIGHT
#define BEVWARP_STRAIGHT 1
#endif
#ifndef BEVWARP_AHEAD
#define BEVWARP_AHEAD 2  // bilinear: passes of taps in flight in the straight-line form
#endif
// waves per SIMD a format's kernel is compiled for: what its interior loop needs without spilling
constexpr int waves_per_simd_of(bool is_u8, int channels, int interp) {
    return interp == kLinear && (is_u8 || channels == 4) ? BEVWARP_U8LIN_WAVES : 4;
}
template <typename T, int C, int INTERP>
constexpr int waves_per_simd() { return waves_per_simd_of(sizeof(T) == 1, C, INTERP); }
// NSRC = 3 is warp_composite (bev/tool/compo.py:26-49) in one launch: a workgroup of 12 waves, four per source -- waves 0-3
// warp the background, 4-7 the foreground, 8-11 its mask, each group exactly as a workgroup of the plain kernel would, every
// group through its own homography and its own tile classification -- into an LDS copy of the tile instead of memory; after one
// barrier all twelve blend the three LDS tiles and store the composite.  The three warped images never exist in memory and
// every pixel is, by construction, what three bevwarp_warp calls produce.
constexpr int kCompositeRows = 16;  // tallest tile of the composite (its three LDS copies: 48 KB)
template <typename T, int C, int INTERP, bool RS4, bool PLANAR, int NSRC = 1>
__global__ __launch_bounds__(kWG * NSRC) __attribute__((amdgpu_waves_per_eu(NSRC > 1 ? 3 : waves_per_simd<T, C, INTERP>(), 8))) void warp_rows(const WarpArgs a) {
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
    __shared__ __attribute__((aligned(16))) uint32_t s_tr[kWaves * NSRC][kRowsLds][TRW];
    // (composite only) the warped tiles, one packed pixel per dword: [source][row of the tile][pixel]
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[NSRC > 1 ? NSRC * kCompositeRows * TW : 4];
    constexpr int NEED = LOADB / 4;  // dwords of a tap row the blend takes, starting AT the left tap
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

    // -- wave-uniform terms of the fast chain (numerators carry the 2^32 of the fixed-point form)
    auto uniform_f64 = [](double v) {  // a wave-uniform double, moved to scalar registers
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    };
    const double m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5], m6 = M[6], m7 = M[7], m8 = M[8];
    const double x0d = (double)x0;
    const double CX = uniform_f64((m0 * x0d + m2) * kTwo32), CY = uniform_f64((m3 * x0d + m5) * kTwo32), CW = uniform_f64(m6 * x0d + m8);
    const double RX = uniform_f64(m1 * kTwo32), RY = uniform_f64(m4 * kTwo32), RW = uniform_f64(m7);             // per destination row
    const double DX = uniform_f64(m0 * (64.0 * kTwo32)), DY = uniform_f64(m3 * (64.0 * kTwo32)), DW = uniform_f64(m6 * 64.0);  // per 64 pixels
    // block ownership: xb = first pixel of the strip the coordinate stage works in, CXb.. = the row terms there (set_strip)
    int xb = x0, strip_c = 0;
    double CXb = CX, CYb = CY, CWb = CW;
    auto set_strip = [&](int strip) __attribute__((always_inline)) {
        if (strip == strip_c) return;
        strip_c = strip;
        xb = x0 + 64 * strip;
        const double ds = (double)strip;
        CXb = uniform_f64(__builtin_fma(ds, DX, CX)), CYb = uniform_f64(__builtin_fma(ds, DY, CY)), CWb = uniform_f64(__builtin_fma(ds, DW, CW));
    };
    // per lane: the row terms' offset at the lane's first pixel -- `lane` pixels along the row (row segments), or
    // (lane % PWd, lane / PWd) inside the block's patch.  A tile is processed in one ownership: set below, once it is known.
    double cx0, cy0, cw0;

    // -- byte offsets of FAST rows straight from the high dwords (24-bit multiplies: the host guarantees row stride < 2^24
    // and frames < 2 GiB; a FAST row has 0 <= sx, sy < 2^15, so the low 24 bits of a high dword are 0x380000 + s)
    const uint32_t rs32 = (uint32_t)src_rs;
    const uint32_t fa = kAligned ? (uint32_t)(reinterpret_cast<uintptr_t>(frame) & 3u) : 0u;
    const uint8_t* frame_al = frame - fa;  // 4-byte aligned (frames need not be)
    const uint32_t kOff = fa - 0x380000u * (rs32 + (uint32_t)PBs);
    const uint8_t* dummy = reinterpret_cast<const uint8_t*>(M);  // 72 valid bytes: what rows that are not FAST "load"

    // OUT rows / tiles are filled with the border value: a pixel whose four taps all lie outside the frame IS the border
    // value in the reference (remapBilinear's "fully outside" path stores it directly -- float pixels too, no 4-term blend)

    enum { kFast = 0, kOut = 1, kEdge = 2, kSlow = 3 };
    // the reference's chain for pixel j of this lane (rare: tie windows, SLOW rows); the matrix is re-read here so that the
    // row loop does not carry it in registers
    auto exact_px = [&](auto own, auto in_loop, int xs, int y, int j, int& Xe, int& Ye) __attribute__((always_inline)) {  // xs: the block's first pixel
        constexpr bool kBlk = decltype(own)::blk;
        const double* Mp = M;
        if constexpr (decltype(in_loop)::value) asm volatile("" : "+s"(Mp));  // (keeps the loads below inside this rare branch of a row loop)
        double Me[9];
#pragma unroll
        for (int i = 0; i < 9; i++) Me[i] = Mp[i];
        constexpr bool kPat = decltype(own)::pat;
        const int x = kPat ? xs + PWd * j + (lane & (PWd - 1)) : kBlk ? xs + lane : x0 + 64 * j + lane;
        if (kBlk) y += kPat ? lane / PWd : j;
        const int bx = (int)(fast_div((uint32_t)x, a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0);
        double X0, Y0, W0;
        row_terms(Me, bx, y, X0, Y0, W0);
        const double x1 = (double)(x - bx);
        map_pixel_exact<INTERP>(X0 + Me[0] * x1, Y0 + Me[3] * x1, W0 + Me[6] * x1, Xe, Ye);
    };

    // Row state handed from the coordinate stage to the load / blend stages, 3 dwords per pixel:
    //   FAST      S0 = byte offset of the tap window, S1 / S2 = low dwords of tX / tY (fx, fy in bits 27..31)
    //   others    S0 = 0 (the dummy load), S1 / S2 = integer coordinates X, Y
    // the fast chain of one row segment: (high, low) dwords of tX / tY for the lane's pixels; returns the tie flag (0 = some
    // coordinate of this lane lies in a tie window) and the high dwords of the lane's first / last W
    auto chain_u = [&](auto own, double UX, double UY, double UW, uint32_t (&hx)[PPL], uint32_t (&lx)[PPL], uint32_t (&hy)[PPL], uint32_t (&ly)[PPL],
                       uint32_t& w_first, uint32_t& w_last) __attribute__((always_inline)) -> uint32_t {
        constexpr bool kBlk = decltype(own)::blk;
        // from pixel j to pixel j + 1 of a lane: 64 pixels along the row (row segments), one row down (blocks), 64 / PPL pixels (patches)
        constexpr bool kPat = decltype(own)::pat;
        // (patches: a quarter / half of the per-64-pixel terms -- exact, and three multiplies per pass are cheaper than six more
        // scalar registers held through the kernel)
        const double dX = kPat ? DX * (1.0 / PPL) : kBlk ? RX : DX, dY = kPat ? DY * (1.0 / PPL) : kBlk ? RY : DY, dW = kPat ? DW * (1.0 / PPL) : kBlk ? RW : DW;
        double W[PPL], r[PPL];
        W[0] = UW + cw0;
#pragma unroll
        for (int j = 1; j < PPL; j++) W[j] = W[j - 1] + dW;
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
                Xn += dX;
                Yn += dY;
            }
        }
        w_first = (uint32_t)__double2hiint(W[0]);
        w_last = (uint32_t)__double2hiint(W[PPL - 1]);
        return tie;
    };
    auto chain = [&](auto own, int y, uint32_t (&hx)[PPL], uint32_t (&lx)[PPL], uint32_t (&hy)[PPL], uint32_t (&ly)[PPL], uint32_t& w_first,
                     uint32_t& w_last) __attribute__((always_inline)) -> uint32_t {
        constexpr bool kBlk = decltype(own)::blk;
        const double dy = (double)y;  // the row terms at the first pixel of the segment / block
        return chain_u(own, __builtin_fma(RX, dy, kBlk ? CXb : CX), __builtin_fma(RY, dy, kBlk ? CYb : CY), __builtin_fma(RW, dy, kBlk ? CWb : CW), hx,
                       lx, hy, ly, w_first, w_last);
    };
    // rare: the lane's pixels that lie within 2^-19 of a rounding boundary (or are NaN) take the exact chain
    auto fix_ties = [&](auto own, int xs, int y, uint32_t (&hx)[PPL], uint32_t (&lx)[PPL], uint32_t (&hy)[PPL], uint32_t (&ly)[PPL])
                        __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if ((lx[j] & F::kTieMask) == 0 || (ly[j] & F::kTieMask) == 0) {
                int Xe, Ye;
                exact_px(own, InLoop{}, xs, y, j, Xe, Ye);
                int_to_fix<INTERP>(Xe, hx[j], lx[j]);
                int_to_fix<INTERP>(Ye, hy[j], ly[j]);
            }
        }
    };
    // (block ownership: y = the block's first row)
    auto coords_s = [&](auto own, int strip, int y, uint32_t (&S0)[PPL], uint32_t (&S1)[PPL], uint32_t (&S2)[PPL]) __attribute__((always_inline)) -> int {
        constexpr bool kPat = decltype(own)::pat;
        // lanes / pixels of the block's top-right and bottom-left corners (top-left: pixel 0 of lane 0, bottom-right: pixel PPL-1 of lane 63)
        constexpr int kTRl = kPat ? PWd - 1 : 63, kTRj = kPat ? PPL - 1 : 0, kBLl = kPat ? 64 - PWd : 0, kBLj = kPat ? 0 : PPL - 1;
        set_strip(strip);
        uint32_t hx[PPL], lx[PPL], hy[PPL], ly[PPL], w_first, w_last;
        const uint32_t tie = chain(own, y, hx, lx, hy, ly, w_first, w_last);
        // -- classify the block from its four corner pixels, in scalar registers
        auto lane_u32 = [](uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); };
        const uint32_t hxa = lane_u32(hx[0], 0), hya = lane_u32(hy[0], 0), hxb = lane_u32(hx[PPL - 1], 63), hyb = lane_u32(hy[PPL - 1], 63);
        const uint32_t wa = lane_u32(w_first, 0), wb = lane_u32(w_last, 63);
        const uint32_t ea = (wa >> 20) & 0x7ffu, eb = (wb >> 20) & 0x7ffu;  // 2^-199 .. 2^199: the shared reciprocal is safe
        bool w_ok = ((wa ^ wb) >> 31) == 0 && ea - 824u <= 398u && eb - 824u <= 398u;
        // a block has two more corners (W is linear: one sign at the four corners = one sign inside)
        const uint32_t hxc = lane_u32(hx[kTRj], kTRl), hyc = lane_u32(hy[kTRj], kTRl);
        const uint32_t hxd = lane_u32(hx[kBLj], kBLl), hyd = lane_u32(hy[kBLj], kBLl);
        w_ok = w_ok && ((wa ^ lane_u32(kTRj ? w_last : w_first, kTRl)) >> 31) == 0 && ((wa ^ lane_u32(kBLj ? w_last : w_first, kBLl)) >> 31) == 0;
        // source pixel of the two ends (a high dword outside the binade gives |s| >= 2^19: outside every limit below)
        const int sxa = (int)(hxa - kHiBias), sya = (int)(hya - kHiBias), sxb = (int)(hxb - kHiBias), syb = (int)(hyb - kHiBias);
        const int sxc = (int)(hxc - kHiBias), syc = (int)(hyc - kHiBias), sxd = (int)(hxd - kHiBias), syd = (int)(hyd - kHiBias);
        const bool in = can_fast && w_ok && (uint32_t)(sxa - kM) <= (uint32_t)(sxw_lim - 2 * kM) && (uint32_t)(sxb - kM) <= (uint32_t)(sxw_lim - 2 * kM) &&
                        (uint32_t)(sya - kM) <= (uint32_t)(sy_lim - 2 * kM) && (uint32_t)(syb - kM) <= (uint32_t)(sy_lim - 2 * kM) &&
                        (uint32_t)(sxc - kM) <= (uint32_t)(sxw_lim - 2 * kM) && (uint32_t)(sxd - kM) <= (uint32_t)(sxw_lim - 2 * kM) &&
                        (uint32_t)(syc - kM) <= (uint32_t)(sy_lim - 2 * kM) && (uint32_t)(syd - kM) <= (uint32_t)(sy_lim - 2 * kM);
        int cls = kFast;
        if (__builtin_expect(!in, 0)) {  // (the common class costs no further scalar work)
            const bool e_ok = (((hxa ^ kHiExp) | (hya ^ kHiExp) | (hxb ^ kHiExp) | (hyb ^ kHiExp) | (hxc ^ kHiExp) | (hyc ^ kHiExp) | (hxd ^ kHiExp) |
                                (hyd ^ kHiExp)) >> 20) == 0;  // every corner inside the binade
            const int sx_hi = max(max(sxa, sxb), max(sxc, sxd)), sx_lo = min(min(sxa, sxb), min(sxc, sxd));
            const int sy_hi = max(max(sya, syb), max(syc, syd)), sy_lo = min(min(sya, syb), min(syc, syd));
            const bool out = sx_hi <= -3 || sx_lo > src_w || sy_hi <= -3 || sy_lo > src_h;
            // kEdge: W of one sign and both ends representable => every pixel between them is (the map is monotone along
            // the segment), taps need guards
            cls = !(e_ok && w_ok) ? kSlow : (out ? kOut : kEdge);
        }
        if (tie == 0 && cls != kSlow) fix_ties(own, xb, y, hx, lx, hy, ly);
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
    auto issue_s = [&](int cls, const uint32_t (&S0)[PPL], Bytes<WINB> (&t0)[PPL], Bytes<WINB> (&t1)[PPL]) __attribute__((always_inline)) {
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

    uint32_t* const wtr0 = &s_tr[wave_all][0][0];
    uint32_t* wtr = wtr0;  // the LDS row of the pass being blended / read back
    constexpr bool kDefer = NSRC == 1;
    auto lds_row = [&](int k) __attribute__((always_inline)) { wtr = kDefer ? wtr0 + k * TRW : wtr0; };
    // blend one pixel from its taps -- w0 / w1 = the LOADB bytes of the upper / lower tap row starting AT the left tap -- into
    // the wave's LDS row (pixel 64 j + lane of the segment)
    auto blend_put = [&](int j, const uint32_t (&w0)[NEED], const uint32_t (&w1)[NEED], uint32_t fx, uint32_t fy) __attribute__((always_inline)) {
        if constexpr (sizeof(T) == 1) {
            uint32_t px;
            if constexpr (INTERP == kNearest)
                px = C == 4 ? w0[0] : (w0[0] & ((1u << (8 * (C & 3))) - 1u));
            else if constexpr (C == 3)
                px = gray_src ? blend_u8_gray_window(w0[0], w0[1], w1[0], w1[1], fx, fy) : blend_u8_rgb_window(w0[0], w0[1], w1[0], w1[1], fx, fy);
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
                        const Bytes<WINB> (&t1)[PPL]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint32_t fx = S1[j] >> 27, fy = S2[j] >> 27;  // (bilinear only)
            uint32_t w0[NEED], w1[NEED];
            if constexpr (kAligned) {
                // funnel shift by the window's byte phase: v_alignbyte_b32 shifts by 8 * S2[1:0] on gfx950 (tools/probe_alignbyte.hip:
                // the ISA manuals disagree on [1:0] vs [4:0]), so the byte offset itself is the shift operand
                const uint32_t sh0 = S0[j], sh1 = RS4 ? sh0 : S0[j] + rs32;
#pragma unroll
                for (int k = 0; k < NEED; k++) {
                    w0[k] = __builtin_amdgcn_alignbyte(t0[j].w[k + 1], t0[j].w[k], sh0);
                    w1[k] = __builtin_amdgcn_alignbyte(t1[j].w[k + 1], t1[j].w[k], sh1);
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
    auto slow_s = [&](auto own, auto in_loop, int xs, int y) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            int Xe, Ye;
            exact_px(own, in_loop, xs, y, j, Xe, Ye);
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
            const bool all_out = sx < -kTap || sx >= src_w || sy < -kTap || sy >= src_h;
            Pixel<T, C> v;
            if (inb[j]) {
                if constexpr (sizeof(T) == 1) {
                    if (INTERP == kNearest)
                        v.packed = C == 4 ? e0[j].w[0] : (e0[j].w[0] & ((1u << (8 * (C & 3))) - 1u));
                    else if constexpr (C == 3)
                        v.packed = gray_src ? blend_u8_gray_window(e0[j].w[0], e0[j].w[1], e1[j].w[0], e1[j].w[1], fx, fy)
                                            : blend_u8_rgb_window(e0[j].w[0], e0[j].w[1], e1[j].w[0], e1[j].w[1], fx, fy);
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
    static_assert(NQ * 4 >= PPL * C || sizeof(T) == 1, "the planar float path keeps a lane's PPL pixels in the same registers");
    auto read_back = [&](uint4 (&out)[NQ]) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");  // compiler fence: one wave's LDS operations execute in program order
        if constexpr (PLANAR && sizeof(T) == 4) {  // float planes: the lane's own PPL pixels, channel by channel (pixel 64 j + lane)
            uint32_t* o = reinterpret_cast<uint32_t*>(&out[0]);
#pragma unroll
            for (int j = 0; j < PPL; j++)
#pragma unroll
                for (int k = 0; k < C; k++) o[j * C + k] = wtr[(64 * j + lane) * C + k];
            asm volatile("" ::: "memory");
            return;
        }
#pragma unroll
        for (int u = 0; u < NQ; u++) {
            const int q = u * 64 + lane;
            if (q < kVec) out[u] = reinterpret_cast<const uint4*>(wtr)[q];
        }
        asm volatile("" ::: "memory");  // (the next row's LDS writes cannot pass these reads)
    };
    auto finish_any = [&](auto own, int cls, int xs, int y, const uint32_t (&S0)[PPL], const uint32_t (&S1)[PPL], const uint32_t (&S2)[PPL], const Bytes<WINB> (&t0)[PPL],
                          const Bytes<WINB> (&t1)[PPL]) __attribute__((always_inline)) {
        if (__builtin_expect(cls == kFast, 1)) {
            finish_s(S0, S1, S2, t0, t1);
        } else {
            if (cls == kOut)
                fill_s();
            else if (cls == kEdge)
                edge_s(S1, S2);
            else
                slow_s(own, InLoop{}, xs, y);
        }
    };

    // -- stores.  Plain (default-policy) wide stores: the destination is written once and never read back by this kernel, but
    // non-temporal stores -- round 2's choice -- measure SLOWER on every format here (A/B of four builds interleaved on one
    // box, profiles/r03_store_ab.txt: float32 keystone 192.8 -> 183.7 us, 8-bit bilinear 73.7 -> 72.2, nearest 58.3 -> 55.6).
    // Lanes of a ragged last tile (and every lane when the destination's layout does not admit the wide stores) fall back to
    // element stores.  (Also measured and not kept: the wide store as ONE unconditional buffer_store whose masked lanes get an
    // out-of-range offset, so that the compiler's vmcnt accounting sees it -- nearest -4 %, bilinear and float +4 %.)
    auto store_s = [&](auto own, int xs, int y, const uint4 (&out)[NQ]) __attribute__((always_inline)) {  // xs = first pixel of the segment / block
        constexpr bool kBlk = decltype(own)::blk;
        const int seg_px = min(kBlk ? 64 : TW, a.dst_w - xs);  // valid pixels of a row of the segment / block
        // 8-bit: a lane stores the 4 consecutive pixels of its 16-byte unit of the LDS row: pixels 4 l .. of the wave's row; row
        // l / 16 of its block, pixels 4 (l % 16) ..; or (patches: LDS index 64 j + 16 ly + lx) row (l / 4) % 4, pixels 16 (l / 16) + 4 (l % 4) ..
        constexpr bool kPat = decltype(own)::pat;
        const int st_row = kPat ? (lane >> 2) & 3 : kBlk ? lane >> 4 : 0;
        const int st_x = kPat ? xs + 16 * (lane >> 4) + 4 * (lane & 3) : kBlk ? xs + (lane & 15) * PPL : xs + lane * PPL;
        const int lane_px = max(0, min(PPL, a.dst_w - st_x));  // valid pixels of this lane's store unit
        const bool lane_vec = a.dst_vec_ok && lane_px == PPL;
        if constexpr (sizeof(T) == 1) {  // the lane's 4 pixels = 4 C contiguous bytes, one instruction
            const uint32_t p[4] = {out[0].x, out[0].y, out[0].z, out[0].w};
            if constexpr (PLANAR) {  // float planes: one 16-byte store per channel
                if (kBlk && y + st_row > y_last) return;
                uint8_t* dp = dframe + (int64_t)(y + st_row) * a.dst_rs + (int64_t)st_x * 4;
#pragma unroll
                for (int k = 0; k < C; k++) {
                    const float sc = a.pscale[k], bi = a.pbias[k];
                    f32x4 o = {(float)((p[0] >> (8 * k)) & 0xffu) * sc + bi, (float)((p[1] >> (8 * k)) & 0xffu) * sc + bi,
                               (float)((p[2] >> (8 * k)) & 0xffu) * sc + bi, (float)((p[3] >> (8 * k)) & 0xffu) * sc + bi};
                    float* dk = reinterpret_cast<float*>(dp + k * a.dst_ps);
                    if (__builtin_expect(lane_vec, 1)) {
                        wide_store(reinterpret_cast<f32x4*>(dk), o);
                    } else {
                        for (int i = 0; i < lane_px; i++) dk[i] = o[i];
                    }
                }
                return;
            }
            if (kBlk && y + st_row > y_last) return;
            if constexpr (NSRC > 1) {  // composite: the lane's four packed pixels go to this source's LDS copy of the tile
                *reinterpret_cast<uint4*>(&s_tile[(sid * kCompositeRows + (y + st_row - y0)) * TW + (st_x - x0)]) = out[0];
                return;
            }
            uint8_t* d = dframe + (int64_t)(y + st_row) * a.dst_rs + (int64_t)st_x * C;
            if (__builtin_expect(lane_vec, 1)) {
                if constexpr (C == 1) {
                    *reinterpret_cast<uint32_t*>(d) = p[0] | (p[1] << 8) | (p[2] << 16) | (p[3] << 24);
                } else if constexpr (C == 2) {
                    u32x2 o = {p[0] | (p[1] << 16), p[2] | (p[3] << 16)};
                    *reinterpret_cast<u32x2*>(d) = o;
                } else if constexpr (C == 3) {
                    u32x3 o = {p[0] | (p[1] << 24), (p[1] >> 8) | (p[2] << 16), (p[2] >> 16) | (p[3] << 8)};
                    wide_store(reinterpret_cast<u32x3*>(d), o);
                } else {
                    u32x4 o = {p[0], p[1], p[2], p[3]};
                    wide_store(reinterpret_cast<u32x4*>(d), o);
                }
            } else {
                for (int i = 0; i < lane_px; i++)
#pragma unroll
                    for (int k = 0; k < C; k++) d[i * C + k] = (uint8_t)(p[i] >> (8 * k));
            }
        } else if constexpr (PLANAR) {  // float source -> float planes: 64 consecutive floats of a plane row per instruction
            const float* o = reinterpret_cast<const float*>(&out[0]);
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const int px = kPat ? xs + PWd * j + (lane & (PWd - 1)) : kBlk ? xs + lane : xs + 64 * j + lane, py = kPat ? y + lane / PWd : kBlk ? y + j : y;
                if (px >= a.dst_w || py > y_last) continue;
#pragma unroll
                for (int k = 0; k < C; k++) {
                    float* dk = reinterpret_cast<float*>(dframe + k * a.dst_ps + (int64_t)py * a.dst_rs) + px;
                    *dk = o[j * C + k] * a.pscale[k] + a.pbias[k];
                }
            }
        } else {
            const int nfl = seg_px * C;  // valid floats of a row of the segment / block
            // blocks: the LDS row holds the block's rows one after the other (runs of 64 pixels); patches: pixels in the order
            // 64 j + PWd ly + lx, i.e. runs of PWd pixels: run n is row n % BR of the block, pixels PWd (n / BR) ..
            constexpr int kRunUnits = (kPat ? PWd : 64) * C / 4;
            static_assert(sizeof(T) == 1 || (PWd * C) % 4 == 0, "a run of the patch is a whole number of 16-byte units");
#pragma unroll
            for (int u = 0; u < NQ; u++) {
                const int q = u * 64 + lane;
                if (q >= kVec) continue;
                const int run = kBlk ? q / kRunUnits : 0;
                const int r = kPat ? run % BR : run, qr = kBlk ? (kPat ? run / BR : 0) * kRunUnits + (q - run * kRunUnits) : q;  // row of the block, unit within the row
                if (kBlk && y + r > y_last) continue;
                float* drow = reinterpret_cast<float*>(dframe + (int64_t)(y + r) * a.dst_rs) + (int64_t)xs * C;
                if (__builtin_expect(a.dst_vec_ok && 4 * qr + 4 <= nfl, 1)) {
                    u32x4 o = {out[u].x, out[u].y, out[u].z, out[u].w};
                    wide_store(&reinterpret_cast<u32x4*>(drow)[qr], o);
                } else {
                    const uint32_t f[4] = {out[u].x, out[u].y, out[u].z, out[u].w};
                    for (int i = 0; i < 4 && 4 * qr + i < nfl; i++) reinterpret_cast<uint32_t*>(drow)[4 * qr + i] = f[i];
                }
            }
        }
    };

    // -- the row loop.  Rows of a tile are dealt to its waves round-robin: neighbouring rows share source lines and run
    // at the same time.
    uint32_t A0[PPL], A1[PPL], A2[PPL], B0[PPL], B1[PPL], B2[PPL];  // row state: current / next
    Bytes<WINB> u0[PPL], u1[PPL];
    uint4 out[NQ];
    // -- interior tiles.  When the tile's four corner pixels sample inside the frame by the FAST margin with W of one sign,
    // the tile maps into the convex quadrilateral of their images: every row is FAST and the loop needs no row classes --
    // no end-pixel read-out, no scalar decisions, row terms advanced by one addition each.
    // -- the tile's four corner pixels decide how it is processed
    bool tile_in, tile_slanted, tile_out, tile_affine;
    {
        const int ck = lane & 3;
        const double cdx = (ck & 1) ? (double)(TW - 1) : 0.0, cdy = (double)((ck & 2) ? y_last : y0);
        const double cW = __builtin_fma(RW, cdy, CW) + m6 * cdx;
        const double cr = rcp_newton(cW);
        const double ctx = __builtin_fma(__builtin_fma(RX, cdy, CX) + (m0 * kTwo32) * cdx, cr, F::kMagic);
        const double cty = __builtin_fma(__builtin_fma(RY, cdy, CY) + (m3 * kTwo32) * cdx, cr, F::kMagic);
        const uint32_t chx = (uint32_t)__double2hiint(ctx), chy = (uint32_t)__double2hiint(cty);
        const int csx = (int)(chx - kHiBias), csy = (int)(chy - kHiBias);
        const uint32_t cwh = (uint32_t)__double2hiint(cW);
        const bool c_w = ((cwh >> 20) & 0x7ffu) - 824u <= 398u;
        // (one pixel more than the rows' own margin: the exact chain may move a coordinate by a unit)
        const bool c_in = (uint32_t)(csx - kM - 1) <= (uint32_t)(sxw_lim - 2 * kM - 2) && (uint32_t)(csy - kM - 1) <= (uint32_t)(sy_lim - 2 * kM - 2) && c_w;
        const uint32_t in4 = (uint32_t)__ballot(c_in) & 0xFu, neg4 = (uint32_t)__ballot((cwh >> 31) != 0) & 0xFu;
        const bool one_sign = neg4 == 0u || neg4 == 0xFu;
        // interior: every pixel of the tile samples inside the frame (the tile maps into the convex quadrilateral of its
        // corners' images) -- no row classes at all
        tile_in = can_fast && sxw_lim >= 2 * kM + 2 && sy_lim >= 2 * kM + 2 && in4 == 0xFu && one_sign;
        // outside: all four corners beyond the same frame edge (coordinates representable, W sane) -- the border value
        const bool c_rep = c_w && (((chx ^ kHiExp) | (chy ^ kHiExp)) >> 20) == 0;
        auto all4 = [](bool v) { return ((uint32_t)__ballot(v) & 0xFu) == 0xFu; };
        tile_out = one_sign && all4(c_rep) && (all4(csx <= -3) || all4(csx > src_w) || all4(csy <= -3) || all4(csy > src_h));
        // Turned footprints.  A row gather's 64 lanes lie on a source line that crosses dy source rows per 64 destination
        // pixels -- a cache line each once dy passes the lines the run would touch anyway -- and pixels sqrt(dx^2 + dy^2) / 64 apart;
        // a 16 x 4 patch of the same 64 pixels crosses a quarter of them.  Measured crossover (A/B over angles and
        // minifications): 8-bit patches win when dy (dx^2 + dy^2) / 64^2 > ~15 (5 degrees at 1.4 x minification, 13
        // degrees at 1 x, 2.5 degrees at 2 x).  Both from the tile's top and bottom edges (corner images).
        auto edge_slant = [&](int l0, int l1) {
            const float dy = (float)abs(__builtin_amdgcn_readlane(csy, l1) - __builtin_amdgcn_readlane(csy, l0));
            const float dx = (float)abs(__builtin_amdgcn_readlane(csx, l1) - __builtin_amdgcn_readlane(csx, l0));
            return dy * (dx * dx + dy * dy);  // (over the tile's width = kStrips x 64 pixels: kStrips^3 times the per-64-pixel figure)
        };
        // The crossover scales with the bytes of a pixel (the lines a 64-pixel run touches anyway): 45 / bytes -- 15 for 8-bit
        // RGB, 11 for RGBA, 45 for grey, < 4 for float RGB (whose 32 x 2 patches cost an unturned footprint nothing).
        // (Single-channel float is the exception: its 32 x 2 patches lose to row segments up to ~20 degrees: 45.)
        constexpr float kSlantThr = sizeof(T) == 4 && C == 1 ? 45.0f : 45.0f / (float)PBs;
        tile_slanted = fmaxf(edge_slant(0, 1), edge_slant(2, 3)) > kSlantThr * 4096.0f * (float)(kStrips * kStrips * kStrips);
        // Row-affine tiles.  Taking W and Y at a segment's first pixel for all TW of them leaves out m6 x in W and m3 x in Y: in
        // the fixed-point units of the chain (source pixels x 2^32) at most
        //     E = TW (|m3| 2^32 r + |m6| r^2 max(|X|, |Y|))        r = 1 / min |W|, X / Y the numerators, all over the tile's corners
        // (W of one sign is linear, so its extremes are at the corners; the numerators are affine too).  The tie window leaves
        // room: the chain itself is within 2^-21.9 output units of the reference for every coordinate the binade admits, the
        // window is 2^-19, so E <= 2^-21 units keeps every pixel outside a window on the reference's side of its rounding boundary.
        // Exactly zero for M3 = M6 = 0; ~1e-5 for a keystone whose matrix came out of a least-squares fit.
        {
            const double aw = fabs(cW), ax = fabs(__builtin_fma(RX, cdy, CX) + (m0 * kTwo32) * cdx), ay = fabs(__builtin_fma(RY, cdy, CY) + (m3 * kTwo32) * cdx);
            auto max4 = [](double v) {
                v = fmax(v, __shfl_xor(v, 1));
                return fmax(v, __shfl_xor(v, 2));
            };
            const double w_min = -max4(-aw), n_max = max4(fmax(ax, ay));
            const double rr = 1.0 / w_min;
            const double E = (double)TW * (fabs(m3) * kTwo32 * rr + fabs(m6) * rr * rr * n_max);
            constexpr double kUnit = INTERP == kLinear ? 134217728.0 /* 2^27 */ : kTwo32;
            tile_affine = __builtin_amdgcn_readfirstlane((int)(E <= kUnit * (1.0 / 2097152.0) /* 2^-21 */)) != 0;
        }
    }
    // -- the passes of this wave over the tile, in order.
    //   row segments: rows y0 + w + 4 i (neighbouring rows share source lines and run at the same time)
    //   blocks:       strip after strip; inside a strip the four waves take neighbouring blocks (rotated by the strip index so
    //                 that a ragged tile height does not always short-change the same wave): together they walk a compact
    //                 64 x 4 BR patch of the destination at any time, which is what keeps a rotated footprint in L1
    struct Pass {
        int strip, y;
    };
    const int n_strips = (min(TW, a.dst_w - x0) + 63) >> 6;
    auto first_pass = [&](auto own, Pass& p) __attribute__((always_inline)) -> bool {
        constexpr bool kBlk = decltype(own)::blk;
        p.strip = 0;
        p.y = kBlk ? y0 + BR * wave : y0 + wave;
        if (!kBlk) return p.y <= y_last;
        while (p.y > y_last) {
            if (++p.strip >= n_strips) return false;
            p.y = y0 + BR * ((wave + p.strip) & (kWaves - 1));
        }
        return true;
    };
    auto next_pass = [&](auto own, Pass& p) __attribute__((always_inline)) -> bool {
        constexpr bool kBlk = decltype(own)::blk;
        p.y += kBlk ? BR * kWaves : kWaves;
        if (!kBlk) return p.y <= y_last;
        while (p.y > y_last) {
            if (++p.strip >= n_strips) return false;
            p.y = y0 + BR * ((wave + p.strip) & (kWaves - 1));
        }
        return true;
    };
    auto pass_x = [&](const Pass& p) { return x0 + 64 * p.strip; };

    {
        const bool patches = tile_slanted;  // (slanted tiles: interior and edge-cut alike)
        const double lx = (double)(patches ? lane & (PWd - 1) : lane), ly = patches ? (double)(lane / PWd) : 0.0;
        cx0 = __builtin_fma(RX, ly, (m0 * kTwo32) * lx), cy0 = __builtin_fma(RY, ly, (m3 * kTwo32) * lx), cw0 = __builtin_fma(RW, ly, m6 * lx);
    }
    auto run_tile = [&]() __attribute__((always_inline)) {
    if (tile_out) {  // every pixel of the tile is the border value
        Pass p;
        if (!first_pass(BlkSeg{}, p)) return;
        fill_s();
        read_back(out);
        do store_s(BlkSeg{}, pass_x(p), p.y, out);
        while (next_pass(BlkSeg{}, p));
        return;
    }
    // -- interior tiles: no row classes; software-pipelined (the next pass's loads in flight, the pass after that getting its
    // coordinates, while the previous one is stored and the current one blended).
    // Pixels in a tie window are not fixed where they are found: the pass is blended from its fast coordinates (a unit off
    // at worst -- still inside the frame, the tile test keeps a pixel of margin for exactly this), a flag is shifted into a
    // scalar mask, and flagged passes (rare) are redone whole by the exact chain after the loop.  With the exact chain out
    // of the loop its state fits 128 VGPRs: four waves per SIMD instead of three (-12 %: DESIGN.md section 6.2).
    // ROW-AFFINE tiles (`row_affine` tag; row segments only).  When the source row and the perspective divide do not depend on the
    // destination column -- inverse matrix with M3 = M6 = 0: the rectification / inverse-perspective form, every keystone, every
    // scale + shift; judged per tile with the tolerance derived at `tile_affine` -- a row segment needs ONE reciprocal and ONE Y
    // coordinate per pass, the same in every lane: X = (UX + x M0) r is one add and one FMA per pixel, the tap row's byte offset is
    // a scalar, fy and the Y tie flag are wave-uniform.  18 float64 instructions per 256-pixel row instead of 38, 4 offset
    // instructions instead of 12.
    auto interior = [&](auto own, auto row_affine) __attribute__((always_inline)) {
        constexpr bool kBlk = decltype(own)::blk;
        constexpr bool kAff = decltype(row_affine)::value;
        static_assert(!kAff || !kBlk, "row-affine passes are row segments");
        Pass p_cur, p_nxt;  // p_cur: the pass whose pixels sit in the LDS row; p_nxt: the pass whose coordinates are in the C state
        if (!first_pass(own, p_nxt)) return;
        // row terms of the pass the coordinate stage is at; a row segment's advance by one addition per pass
        double UX = 0, UY = 0, UW = 0;
        const double SX = uniform_f64(RX * (double)kWaves), SY = uniform_f64(RY * (double)kWaves), SW = uniform_f64(RW * (double)kWaves);
        if (!kBlk) UX = __builtin_fma(RX, (double)p_nxt.y, CX), UY = __builtin_fma(RY, (double)p_nxt.y, CY), UW = __builtin_fma(RW, (double)p_nxt.y, CW);
        auto coords_f = [&](const Pass& p, uint32_t (&S0)[PPL], uint32_t (&S1)[PPL], uint32_t (&S2)[PPL]) __attribute__((always_inline)) {
            uint32_t hx[PPL], lx[PPL], hy[PPL], ly[PPL], wf_, wl_;
            if constexpr (kAff) {
                // W, 1 / W and Y at the segment's first pixel stand for the whole segment (tile_affine bounds what that ignores)
                const double r = rcp_newton(UW);
                const double ty_ = __builtin_fma(UY, r, F::kMagic);
                const uint32_t hyu = (uint32_t)__builtin_amdgcn_readfirstlane(__double2hiint(ty_)), lyu = (uint32_t)__builtin_amdgcn_readfirstlane(__double2loint(ty_));
                const uint32_t row_off = (hyu & 0xffffffu) * rs32 + kOff;  // scalar; the low 24 bits like v_mul_u32_u24 (kOff carries their bias)
                double Xn = UX + cx0;
#pragma unroll
                for (int j = 0; j < PPL; j++) {
                    const double tx_ = __builtin_fma(Xn, r, F::kMagic);
                    S0[j] = __umul24((uint32_t)__double2hiint(tx_), (uint32_t)PBs) + row_off;
                    S1[j] = (uint32_t)__double2loint(tx_);
                    S2[j] = lyu;
                    if (j + 1 < PPL) Xn += DX;
                }
                UX += SX;
                UY += SY;
                UW += SW;
                return;
            }
            if constexpr (kBlk) {
                set_strip(p.strip);
                chain(own, p.y, hx, lx, hy, ly, wf_, wl_);
            } else {
                chain_u(own, UX, UY, UW, hx, lx, hy, ly, wf_, wl_);
                UX += SX;
                UY += SY;
                UW += SW;
            }
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                S0[j] = __umul24(hy[j], rs32) + (__umul24(hx[j], (uint32_t)PBs) + kOff);
                S1[j] = lx[j];
                S2[j] = ly[j];
            }
        };
        uint64_t tie_passes = 0;  // bit k: the pass blended k passes before the last one has a pixel in a tie window
        int n_blended = 0;        // (at most 64 passes per wave and tile: the host keeps tiles <= 64 rows)
        auto note_ties = [&](const uint32_t (&S1)[PPL], const uint32_t (&S2)[PPL]) __attribute__((always_inline)) {
            uint32_t tie = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < PPL; j++) tie = min(tie, min(S1[j] & F::kTieMask, S2[j] & F::kTieMask));
            tie_passes = (tie_passes << 1) | (uint64_t)(__ballot(tie == 0) != 0ull);
            n_blended++;
        };
        bool more;
        int k_blend = 0;  // index of the pass being blended (its LDS row when stores are deferred)
        auto step = [&](uint32_t (&C0)[PPL], uint32_t (&C1)[PPL], uint32_t (&C2)[PPL], uint32_t (&N0)[PPL], uint32_t (&N1)[PPL], uint32_t (&N2)[PPL])
                        __attribute__((always_inline)) {
            if constexpr (!kDefer) read_back(out);  // p_cur's pixels (written a step ago: no LDS latency on the path)
            issue_s(kFast, C0, u0, u1);
            const Pass p_st = p_cur;
            p_cur = p_nxt;
            more = next_pass(own, p_nxt);
            if (more) coords_f(p_nxt, N0, N1, N2);
            if constexpr (!kDefer) store_s(own, pass_x(p_st), p_st.y, out);  // behind the loads: vmcnt retires in issue order
            lds_row(++k_blend);
            finish_s(C0, C1, C2, u0, u1);
            note_ties(C1, C2);
        };
        // Nearest neighbour: the taps of up to kDepth passes in flight -- a ring of pass slots; the loads of a pass are issued as soon
        // as the slot's previous pass has been put away, and with kDepth = 6 every pass of a wave over a 24-row tile has its loads out
        // before the first is consumed (a tap is one dword per pixel: 4 registers per slot).  A nearest pass has next to no arithmetic
        // to hide its loads behind, so the depth is what it runs at -- A/B on one box, 48 rounds, buffer sets drawn at random
        // (profiles/r03_late_ab.txt): depth 1 / 2 / 4 / 6 = 58.2 / 56.8 / 54.9 / 51.1 us per 32 frames (-12 %), a 25-degree
        // footprint 66.9 -> 63.2.  (Behind the ring's branches the compiler cannot count the loads issued after the ones a pass
        // consumes and waits for vmcnt(0): the ring only pays once everything is issued up front.  Full-height tiles take the
        // straight-line form below, whose counts are exact; the ring is what ragged tiles run.)
        constexpr bool kDeep = kDefer && (INTERP == kNearest || BEVWARP_DEEP_ALL);
        // A wave with a full set of rows (kRowsLds passes: every wave of a full-height tile) runs them as STRAIGHT-LINE code.  Only
        // there does the compiler know how many loads were issued after the ones a pass is about to consume -- behind a loop's or the
        // ring's branches it has to assume none and waits for vmcnt(0), i.e. for every load in flight, the newest included.  Nearest:
        // all six passes' loads up front.  Bilinear: TWO tap sets, the loads of pass n + 2 issued right after pass n has been
        // blended and consumed after pass n + 1 has, under s_waitcnt vmcnt(13..8) -- a whole step in flight instead of the
        // coordinate stage of one: 8-bit row-affine tiles 80.1 -> 77.1 us per 32 frames (-3.7 %; A/B, profiles/r03_late_ab.txt),
        // float unchanged (191.5: it runs at the streaming rate either way).  The same two sets behind a loop's branches measured
        // nothing at all (80.1 -> 80.8): the waits, not the loads, were what did not overlap.  A third set does not fit 128
        // registers (the scheduler sinks its loads back to where two sets put them); the general chain spills with two.
        bool straight = false;
        if constexpr (kDefer && !kBlk && BEVWARP_STRAIGHT && (INTERP == kNearest || kAff || sizeof(T) == 4)) {  // (8-bit bilinear, general chain: two tap sets spill)
            constexpr int kFull = kRowsLds;
            constexpr int kAhead = INTERP == kNearest ? kFull : BEVWARP_AHEAD;  // passes of taps in flight
            if ((y_last - p_nxt.y) / kWaves + 1 == kFull) {
                straight = true;
                uint32_t R0[kAhead][PPL], R1[kAhead][PPL], R2[kAhead][PPL];
                Bytes<WINB> r0[kAhead][PPL], r1[kAhead][PPL];
#pragma unroll
                for (int k = 0; k < kAhead; k++) {
                    coords_f(p_nxt, R0[k], R1[k], R2[k]);
                    note_ties(R1[k], R2[k]);
                    issue_s(kFast, R0[k], r0[k], r1[k]);
                }
#pragma unroll
                for (int k = 0; k < kFull; k++) {
                    constexpr int kA = kAhead;
                    const int d = k % kA;
                    lds_row(k);
                    finish_s(R0[d], R1[d], R2[d], r0[d], r1[d]);
                    if (k + kA < kFull) {
                        coords_f(p_nxt, R0[d], R1[d], R2[d]);
                        note_ties(R1[d], R2[d]);
                        issue_s(kFast, R0[d], r0[d], r1[d]);
                    }
                }
            }
        }
        if (straight) {
        } else if constexpr (kDeep) {
            constexpr int kDepth = BEVWARP_DEPTH;  // passes in flight
            uint32_t R0[kDepth][PPL], R1[kDepth][PPL], R2[kDepth][PPL];  // ring of pass states (nearest: dead once the loads are out)
            Bytes<WINB> r0[kDepth][PPL], r1[kDepth][PPL];
            bool live[kDepth];
            bool any = true;  // passes are issued in order: the first slot that finds none ends the tile
#pragma unroll
            for (int d = 0; d < kDepth; d++) {
                live[d] = any && (d == 0 || next_pass(own, p_nxt));
                any = live[d];
                if (live[d]) {
                    coords_f(p_nxt, R0[d], R1[d], R2[d]);
                    note_ties(R1[d], R2[d]);
                    issue_s(kFast, R0[d], r0[d], r1[d]);
                }
            }
            int kb = 0;
            for (bool done = false; !done;) {
#pragma unroll
                for (int d = 0; d < kDepth; d++) {
                    if (done || !live[d]) {
                        done = true;
                        continue;
                    }
                    lds_row(kb++);
                    finish_s(R0[d], R1[d], R2[d], r0[d], r1[d]);
                    any = any && next_pass(own, p_nxt);
                    live[d] = any;
                    if (any) {
                        coords_f(p_nxt, R0[d], R1[d], R2[d]);
                        note_ties(R1[d], R2[d]);
                        issue_s(kFast, R0[d], r0[d], r1[d]);
                    }
                }
            }
        } else {
        coords_f(p_nxt, A0, A1, A2);
        issue_s(kFast, A0, u0, u1);
        p_cur = p_nxt;
        more = next_pass(own, p_nxt);
        if (more) coords_f(p_nxt, B0, B1, B2);
        lds_row(0);
        finish_s(A0, A1, A2, u0, u1);
        note_ties(A1, A2);
        while (more) {  // (two steps per trip: the states swap roles instead of being copied)
            step(B0, B1, B2, A0, A1, A2);
            if (!more) break;
            step(A0, A1, A2, B0, B1, B2);
        }
        if constexpr (!kDefer) {
            read_back(out);
            store_s(own, pass_x(p_cur), p_cur.y, out);
        }
        }
        if (__builtin_expect(tie_passes != 0, 0)) {  // redo the flagged passes: every pixel by the exact chain and the generic sampler
            Pass p;
            first_pass(own, p);
            int k = n_blended - 1, kr = 0;
            do {
                if ((tie_passes >> k) & 1ull) {
                    lds_row(kr);
                    slow_s(own, InTail{}, pass_x(p), p.y);
                    if constexpr (!kDefer) {
                        read_back(out);
                        store_s(own, pass_x(p), p.y, out);
                    }
                }
                k--, kr++;
            } while (next_pass(own, p));
        }
        if constexpr (kDefer) {  // every pass of this wave, LDS -> memory: the stores trail the tile's last load
            Pass p;
            first_pass(own, p);
            int kr = 0;
            do {
                lds_row(kr++);
                read_back(out);
                store_s(own, pass_x(p), p.y, out);
            } while (next_pass(own, p));
        }
    };
    // (the branch hints keep the common path -- interior tile, row segments -- the fall-through: with the patch code in the
    // kernel its layout otherwise costs unturned footprints 3 %)
    if (__builtin_expect(tile_in, 1)) {
        if (__builtin_expect(!tile_slanted, 1)) {
            if (tile_affine)
                interior(RowSeg{}, std::true_type{});
            else
                interior(RowSeg{}, std::false_type{});
        } else {
            interior(PatSeg{}, std::false_type{});
        }
        return;
    }
    // -- the frame's edge crosses the tile (or W changes sign in it): blocks with a class each, the same pipeline
    auto edge_tile = [&](auto own) __attribute__((always_inline)) {
        Pass p_cur, p_nxt;
        if (!first_pass(own, p_nxt)) return;
        int k_blend = 0;
        int cls_c = coords_s(own, p_nxt.strip, p_nxt.y, A0, A1, A2), cls_n = kSlow;
        issue_s(cls_c, A0, u0, u1);
        p_cur = p_nxt;
        bool more = next_pass(own, p_nxt);
        if (more) cls_n = coords_s(own, p_nxt.strip, p_nxt.y, B0, B1, B2);
        lds_row(0);
        finish_any(own, cls_c, pass_x(p_cur), p_cur.y, A0, A1, A2, u0, u1);
        while (more) {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                A0[j] = B0[j];
                A1[j] = B1[j];
                A2[j] = B2[j];
            }
            cls_c = cls_n;
            if constexpr (!kDefer) read_back(out);
            issue_s(cls_c, A0, u0, u1);
            const Pass p_st = p_cur;
            p_cur = p_nxt;
            more = next_pass(own, p_nxt);
            if (more) cls_n = coords_s(own, p_nxt.strip, p_nxt.y, B0, B1, B2);  // overlaps with the loads in flight
            if constexpr (!kDefer) store_s(own, pass_x(p_st), p_st.y, out);
            lds_row(++k_blend);
            finish_any(own, cls_c, pass_x(p_cur), p_cur.y, A0, A1, A2, u0, u1);
        }
        if constexpr (!kDefer) {
            read_back(out);
            store_s(own, pass_x(p_cur), p_cur.y, out);
        } else {
            Pass p;
            first_pass(own, p);
            int kr = 0;
            do {
                lds_row(kr++);
                read_back(out);
                store_s(own, pass_x(p), p.y, out);
            } while (next_pass(own, p));
        }
    };
    if (tile_slanted)
        edge_tile(PatSeg{});
    else
        edge_tile(BlkSeg{});
    };  // run_tile
    run_tile();
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
hipError_t debug_read_clock(unsigned long long* out4, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_clk), sizeof(unsigned long long) * 4);
    if (e == hipSuccess && reset) {
        unsigned long long z[4] = {0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z));
    }
    return e;
}
#endif

int tile_width(int dtype) { return 64 * (dtype == 0 ? pixels_per_lane<uint8_t>() : pixels_per_lane<float>()); }
int rows_per_pass() { return kWaves; }  // (a multiple of every format's block height)

// Workgroups of one launch that are resident at the same time: CUs x (waves per SIMD the kernel is compiled for) -- a
// workgroup is 4 waves, one per SIMD.  The launch geometry is sized against this (bevwarp_api.hip).
int resident_workgroups(int dtype, int channels, int interp) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus * waves_per_simd_of(dtype == 0, channels, interp);
}

hipError_t launch_warp(const WarpArgs& a, int dtype, int channels, int interp, hipStream_t stream) {
    (void)hipGetLastError();  // a stale error left by the host framework is not this call's
    const dim3 grid((unsigned)(8 * (a.chunk + a.tail_split)));
    if (dtype == 0)
        launch_t<uint8_t>(a, channels, interp, grid, stream);
    else
        launch_t<float>(a, channels, interp, grid, stream);
    return hipGetLastError();
}

hipError_t launch_warp_composite(const WarpArgs& a, int channels, hipStream_t stream) {
    (void)hipGetLastError();
    const dim3 grid((unsigned)(8 * a.chunk)), block(kWG * 3);
    switch (channels) {
        case 1: hipLaunchKernelGGL((warp_rows<uint8_t, 1, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((warp_rows<uint8_t, 2, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL((warp_rows<uint8_t, 3, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((warp_rows<uint8_t, 4, kLinear, false, false, 3>), grid, block, 0, stream, a); break;
    }
    return hipGetLastError();
}
int composite_max_rows() { return kCompositeRows; }

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
