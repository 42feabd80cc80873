// sample.h -- blending and the guarded sampler of the warp kernel (device code, gfx950).  See DESIGN.md section 4.3.
#pragma once
#include "coords.h"

namespace bevwarp {
namespace {

// the wide destination store of a pass: a PLAIN store.  Non-temporal stores measured slower on every format in rounds 2 and 3
// (profiles/r03_store_ab.txt, r03_late_ab.txt).
template <typename V>
__device__ __forceinline__ void wide_store(V* p, const V& v) {
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

}  // namespace
}  // namespace bevwarp
