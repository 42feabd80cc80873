// warp_kernels.hip -- batched BEV homography warp for MI355X (gfx950, wave64).  See DESIGN.md section 4.
//
// Replaces the per-frame cv2.warpPerspective call of the reference (vis_homo.py:89,91;
// bev/tool/compo.py:38,46,47).  Three kernels share the coordinate chain, the blending arithmetic and the
// guarded border sampling and differ in how source taps reach registers:
//
//   warp_gather (default)  one wave = one TW-pixel row segment (TW = 256 for 8-bit, 128 for float pixels); pixel j
//       of lane l is x0 + 64 j + l.  Every row is classified from its two end pixels (FAST / OUT / EDGE / SLOW);
//       FAST rows load their taps straight from global memory (aligned 12-byte windows + funnel shift for 8-bit
//       RGB), blend, transpose through a wave-private LDS row and store contiguously, software-pipelined one row
//       ahead.  No workgroup barrier.
//   warp_tiles  (BEVWARP_MODE=1)  the workgroup stages the source bounding box of a 64 x 32 tile into LDS with
//       coalesced row loads (8-bit RGB widened to 4 B / pixel; other formats by LDS-DMA), tabulates the row terms
//       in LDS and samples from LDS; 16-row bands when the box exceeds the LDS budget.
//   warp_wave   (BEVWARP_MODE=3)  every wave stages the box of its own 64 x 4 block into a private LDS slot.
//
// Coordinates are float64.  The fast path replaces the IEEE division by rcp + Newton (one reciprocal shared by the
// lane's pixels) and rounds through the float64 mantissa; it equals the reference's rounding chain unless the
// coordinate lies within 2^-19 of a rounding boundary -- those pixels (and anything non-finite or far outside) re-run
// the exact chain, operation for operation.  8-bit blending is exact integer arithmetic on v_dot4_u32_u8; float
// blending keeps the reference's operation order (FMA contraction off).
//
// No MFMA: this is a gather.  The float kernel is bound by HBM; the 8-bit kernels by the texture path's cost per
// gather instruction and by vector-ALU issue (float64 coordinate chain + blending), DESIGN.md section 6.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <type_traits>

#include "warp_kernels.h"

#pragma clang fp contract(off)  // every multiply and add of the coordinate chain rounds separately

namespace bevwarp {
namespace {

constexpr int kWG = 256;
constexpr int kLX = 16;                          // lanes along x inside a wave
constexpr int kLY = 4;                           // lanes along y inside a wave
constexpr int kBandRows = kLY * (kWG / 64);      // 16 rows per workgroup pass
constexpr int kInterBits = 5;
#ifndef BEVWARP_GATHER_LX
#define BEVWARP_GATHER_LX 64
#endif
constexpr int kGatherLX = BEVWARP_GATHER_LX;     // warp_gather: lanes of a wave along x (64 = one row per wave)
constexpr int kRowTabBytes = kMaxTileH * 3 * 8;  // per-row X0, Y0, W0 at the head of dynamic LDS

template <typename T>
constexpr int pixels_per_lane() { return sizeof(T) == 1 ? 4 : 2; }

// LDS bytes per pixel: u8x3 is widened to 4, everything else is stored as is.
template <typename T, int C>
constexpr int lds_pixel_bytes() { return (sizeof(T) == 1 && C == 3) ? 4 : (int)sizeof(T) * C; }
// Formats with a staged fast path: every tap must be a whole number of aligned dwords.
template <typename T, int C>
constexpr bool has_staged_path() { return (sizeof(T) == 1 && (C == 3 || C == 4)) || sizeof(T) == 4; }

struct U3 {
    uint32_t x, y, z;
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
// Fast coordinate path.  r ~= 1/W to 2^-48; p = Xn * r; t = p * 2^s + (1.5 * 2^52 + 2^19) puts
// V = round(fX * 2^20 + 2^19) into the mantissa (fX = coordinate in output units, 1/32 px for
// bilinear).  X = V >> 20 equals rne(fX_exact) whenever the low 20 bits of V are not within 2 of a
// wrap (|fX' - fX| <= 2^22 * 2^-47 << 2^-20 for |fX| < 2^22).  Returns false when the exact chain
// must decide (tie window, |fX| >= 2^22, NaN / Inf, W == 0).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_newton(double w) {
    double r = __builtin_amdgcn_rcp(w);  // v_rcp_f64: relative error 2^-24.4 (measured)
    r = __builtin_fma(__builtin_fma(-w, r, 1.0), r, r);  // -> 2^-48.7
    return r;
}

template <int INTERP>
__device__ __forceinline__ bool round_fast(double p, int& X) {
    constexpr double kScale = INTERP == kLinear ? 33554432.0 /* 32 * 2^20 */ : 1048576.0 /* 2^20 */;
    constexpr double kMagic = 6755399441055744.0 + 524288.0;  // 1.5 * 2^52 + 2^19
    const double t = __builtin_fma(p, kScale, kMagic);
    const uint32_t lo = (uint32_t)__double2loint(t), hi = (uint32_t)__double2hiint(t);
    // bits 20..51 of the mantissa field hold (V >> 20) + 2^31 (the 2^51 of the magic): flip the top bit
    X = (int)(__builtin_amdgcn_alignbit(hi, lo, 20) ^ 0x80000000u);
    const bool tie_window = ((lo + 2u) & 0xfffffu) < 4u;
    // |fX| < 2^22  <=>  |V| < 2^42 + ..: the high dword stays within 0x400 of the magic's
    const bool in_range = (hi + 0x400u - 0x43380000u) < 0x800u;
    return in_range && !tie_window;
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

// Packed 8-bit blend of up to 4 channels: p?? are pixels with channel k in byte k.  Horizontal sums
// with v_dot4_u32_u8 (weights 32-fx, fx <= 32), vertical with 24-bit mads scaled by 64 so that the
// result byte sits in bits 16..23:  ((h0*wy0 + h1*wy1) * 64 + 2^15) >> 16 == (S + 512) >> 10.
// vertical stage of the 8-bit blend: wy0 * top + wy1 * bot + 2^15 as ONE v_dot2_u32_u16 on the packed pair (top and bot
// are horizontal sums <= 8160, wy0 + wy1 = 2048)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t vblend_u8(uint32_t top, uint32_t bot, uint32_t wy01) {
    const uint32_t tb = top | (bot << 16);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, tb), __builtin_bit_cast(u16x2, wy01), 32768u, false);
}

template <int C>
__device__ __forceinline__ uint32_t blend_u8_packed(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, uint32_t fx, uint32_t fy) {
    const uint32_t wlo = fx * 255u + 32u;         // bytes (32 - fx, fx, 0, 0)
    const uint32_t whi = wlo << 16;               // bytes (0, 0, 32 - fx, fx)
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

// What the border / fallback sampler needs of the source frame, by value (taking the address of the
// kernel-argument struct would push it to scratch).
struct SrcView {
    const uint8_t* frame;
    int64_t rs;
    int w, h;
    float bf[4];
    uint32_t bu;  // border bytes packed
};
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

// One pixel straight from global memory with per-tap bounds checks (border, fallback tiles).  Every tap is loaded
// from the CLAMPED coordinate (always a valid address) and replaced by the border value afterwards when its true
// coordinate is outside: the loads are unconditional, so they all issue before the first wait (conditional loads
// compile to one memory round trip each).
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

// One pixel from the staged LDS region.  `px` = LDS pixel index of tap (sx, sy), pitch in pixels.
template <typename T, int C, int INTERP>
__device__ __forceinline__ Pixel<T, C> sample_lds(const uint8_t* __restrict__ lds, uint32_t px, uint32_t pitch_px, int fx, int fy) {
    constexpr int PB = lds_pixel_bytes<T, C>();
    Pixel<T, C> out;
    if constexpr (sizeof(T) == 1) {
        const uint32_t* l = reinterpret_cast<const uint32_t*>(lds);
        if (INTERP == kNearest) {
            out.packed = l[px];
            return out;
        }
        out.packed = blend_u8_packed<C>(l[px], l[px + 1], l[px + pitch_px], l[px + pitch_px + 1], (uint32_t)fx, (uint32_t)fy);
    } else {
        const float* l0 = reinterpret_cast<const float*>(lds + (size_t)px * PB);
        if (INTERP == kNearest) {
#pragma unroll
            for (int k = 0; k < C; k++) out.v[k] = l0[k];
            return out;
        }
        const float* l1 = l0 + (size_t)pitch_px * C;
        float t0[2 * C], t1[2 * C];
#pragma unroll
        for (int k = 0; k < 2 * C; k++) {
            t0[k] = l0[k];
            t1[k] = l1[k];
        }
        float w00, w01, w10, w11;
        weights_f32(fx, fy, w00, w01, w10, w11);
#pragma unroll
        for (int k = 0; k < C; k++) out.v[k] = blend_f32(t0[k], t0[k + C], t1[k], t1[k + C], w00, w01, w10, w11);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------
// Region staging.  rows x npx pixels starting at (ax0, ry0); ax0 and npx are multiples of 4.
// A row is covered by the smallest power-of-two group of lanes (16 / 32 / 64) that holds its
// load units, so a 256-thread pass stages 16 / 8 / 4 rows with coalesced contiguous loads.
// ---------------------------------------------------------------------------------------------------
template <typename T, int C>
__device__ __forceinline__ void stage_region(const WarpArgs& a, const uint8_t* __restrict__ frame, uint8_t* __restrict__ lds, int ax0,
                                             int ry0, int rows, int npx, int tid) {
    constexpr bool kWiden = sizeof(T) == 1 && C == 3;
    constexpr int PB = lds_pixel_bytes<T, C>();
    const int upr = kWiden ? (npx >> 2) : ((npx * PB) >> 4);  // load units per row (12 B -> 16 B, or 16 B)
    const int lg = upr <= 16 ? 4 : (upr <= 32 ? 5 : 6);
    const int lanes = 1 << lg, rstep = kWG >> lg;
    const int q0 = tid & (lanes - 1);
    const uint8_t* base = frame + (int64_t)ry0 * a.src_rs + (int64_t)ax0 * (kWiden ? 3 : PB);
    uint4* l = reinterpret_cast<uint4*>(lds);
    if constexpr (!kWiden) {
        // Natural layout: LDS-DMA (global_load_lds_dwordx4).  No VGPR staging and no ds_write: a wave-instruction
        // moves 64 consecutive 16-byte chunks (its lanes' global addresses are free) to 1 KiB of LDS at a
        // wave-uniform base, so the dense row-major image is filled in chunk order and every load of the
        // region is in flight at once.
        const int total = rows * upr;
        int r = tid / upr, q = tid - r * upr;  // chunk -> (row, column) once; then steps of 256 chunks
        const int dr = kWG / upr, dq = kWG - dr * upr;
        for (int c0 = 0; c0 < total; c0 += kWG) {
            if (c0 + tid < total) {
                const uint8_t* g = base + (int64_t)r * a.src_rs + q * 16;
                uint8_t* lw = lds + (size_t)(c0 + (tid & ~63)) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)lw, 16, 0, 0);
            }
            q += dq;
            r += dr;
            if (q >= upr) {
                q -= upr;
                r++;
            }
        }
        return;
    }
    using Unit = typename std::conditional<kWiden, U3, uint4>::type;
    constexpr int UB = kWiden ? 12 : 16;
    // Memory-level parallelism: a thread first ISSUES up to kBatch loads (its rows of kBatch consecutive
    // passes), then widens / writes them to LDS -- one exposed HBM latency per batch instead of per load.
    // Branch-free inside a batch (rows past the end are clamped to the last row: a redundant, identical
    // load + store) so that the compiler emits the loads back to back instead of load / wait / write chains.
    constexpr int kBatch = 8;
    const int last = rows - 1;
    for (int q = q0; q < upr; q += lanes) {
        for (int r0 = tid >> lg; r0 < rows; r0 += rstep * kBatch) {
            Unit v[kBatch];
            int rr[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; i++) {
                rr[i] = min(r0 + i * rstep, last);
                v[i] = *reinterpret_cast<const Unit*>(base + (int64_t)rr[i] * a.src_rs + q * UB);
            }
#pragma unroll
            for (int i = 0; i < kBatch; i++) {
                if constexpr (kWiden) {
                    uint4 o;
                    o.x = v[i].x & 0x00ffffffu;
                    o.y = __builtin_amdgcn_alignbyte(v[i].y, v[i].x, 3) & 0x00ffffffu;
                    o.z = __builtin_amdgcn_alignbyte(v[i].z, v[i].y, 2) & 0x00ffffffu;
                    o.w = v[i].z >> 8;
                    l[rr[i] * upr + q] = o;
                } else {
                    l[rr[i] * upr + q] = v[i];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Output of one lane: PPL pixels, contiguous in the row (12 B for u8x3, 24 B for f32x3).
// ---------------------------------------------------------------------------------------------------
template <typename T, int C, int PPL>
__device__ __forceinline__ void store_pixels(const WarpArgs& a, uint8_t* __restrict__ drow, int x, int nvalid, const Pixel<T, C>* v) {
    if constexpr (sizeof(T) == 1) {
        if (a.planar) {  // bevwarp_warp_planar, tiles outside the row path: scalar float stores into the channel planes
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                if (j >= nvalid) break;
#pragma unroll
                for (int k = 0; k < C; k++)
                    reinterpret_cast<float*>(drow + k * a.dst_ps)[x + j] = (float)((v[j].packed >> (8 * k)) & 0xffu) * a.pscale[k] + a.pbias[k];
            }
            return;
        }
    }
    T* d = reinterpret_cast<T*>(drow) + (int64_t)x * C;
    if (nvalid == PPL && a.dst_vec_ok) {
        if constexpr (sizeof(T) == 1 && C == 3) {
            U3 o;
            o.x = v[0].packed | (v[1].packed << 24);
            o.y = (v[1].packed >> 8) | (v[2].packed << 16);
            o.z = (v[2].packed >> 16) | (v[3].packed << 8);
            *reinterpret_cast<U3*>(d) = o;
            return;
        }
        if constexpr (sizeof(T) == 1 && C == 4) {
            *reinterpret_cast<uint4*>(d) = make_uint4(v[0].packed, v[1].packed, v[2].packed, v[3].packed);
            return;
        }
        if constexpr (sizeof(T) == 4) {  // 2 pixels x C floats = C units of 8 B
            float f[PPL * C];
#pragma unroll
            for (int j = 0; j < PPL; j++)
#pragma unroll
                for (int k = 0; k < C; k++) f[j * C + k] = v[j].v[k];
            if constexpr ((PPL * C) % 4 == 0) {
#pragma unroll
                for (int k = 0; k < PPL * C / 4; k++) reinterpret_cast<float4*>(d)[k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
            } else {
#pragma unroll
                for (int k = 0; k < PPL * C / 2; k++) reinterpret_cast<float2*>(d)[k] = make_float2(f[2 * k], f[2 * k + 1]);
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        if (j >= nvalid) break;
#pragma unroll
        for (int k = 0; k < C; k++) {
            if constexpr (sizeof(T) == 1)
                d[j * C + k] = (T)((v[j].packed >> (8 * k)) & 0xffu);
            else
                d[j * C + k] = v[j].v[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One lane's PPL pixels of one row when the whole band samples strictly inside its staged region
// ("interior": no border taps, W of one sign and sane magnitude, one evaluation block, full lanes).
// Straight-line code: all LDS reads of the lane's pixels are issued before the blends.
//   rt      LDS row terms X0, Y0, W0 of this row
//   raw*    coordinates as they come out of the mantissa: X + 2^31 (top bit flipped); the shifts
//           below keep working on that biased form and `idx_bias` absorbs the offsets modulo 2^32.
// ---------------------------------------------------------------------------------------------------
template <typename T, int C, int INTERP, int PPL>
__device__ __forceinline__ void row_fast(const double* __restrict__ rt, const double (&mx)[PPL], const double (&my)[PPL],
                                         const double (&mw)[PPL], const uint8_t* __restrict__ lds_px, uint32_t npx, uint32_t idx_bias,
                                         uint8_t* __restrict__ dptr) {
    constexpr double kScale = INTERP == kLinear ? 33554432.0 /* 32 * 2^20 */ : 1048576.0 /* 2^20 */;
    constexpr double kMagic = 6755399441055744.0 + 524288.0;  // 1.5 * 2^52 + 2^19
    const double X0 = rt[0], Y0 = rt[1], W0 = rt[2];
    double W[PPL], r[PPL];
#pragma unroll
    for (int j = 0; j < PPL; j++) W[j] = W0 + mw[j];
    // one reciprocal per lane: 1 / (W0 W1 [W2 W3]), scaled by 2^s (exact), then back-substitution
    if constexpr (PPL == 4) {
        const double p01 = W[0] * W[1], p23 = W[2] * W[3];
        const double inv = rcp_newton(p01 * p23) * kScale;
        const double i01 = inv * p23, i23 = inv * p01;
        r[0] = i01 * W[1];
        r[1] = i01 * W[0];
        r[2] = i23 * W[3];
        r[3] = i23 * W[2];
    } else {
        const double inv = rcp_newton(W[0] * W[1]) * kScale;
        r[0] = inv * W[1];
        r[1] = inv * W[0];
    }
    uint32_t rawX[PPL], rawY[PPL];
    uint32_t tie = 0xffffffffu;  // min over the lane's coordinates of the distance-to-tie field
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        const double tx = (X0 + mx[j]) * r[j] + kMagic, ty = (Y0 + my[j]) * r[j] + kMagic;
        const uint32_t lox = (uint32_t)__double2loint(tx), loy = (uint32_t)__double2loint(ty);
        rawX[j] = __builtin_amdgcn_alignbit((uint32_t)__double2hiint(tx), lox, 20);
        rawY[j] = __builtin_amdgcn_alignbit((uint32_t)__double2hiint(ty), loy, 20);
        tie = min(tie, min((lox + 2u) & 0xffffcu, (loy + 2u) & 0xffffcu));
    }
    if (tie == 0) {  // rare: within 2^-19 of a rounding tie (or NaN / Inf, whose low dword is 0): exact chain
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const double tx = (X0 + mx[j]) * r[j] + kMagic, ty = (Y0 + my[j]) * r[j] + kMagic;
            if ((((uint32_t)__double2loint(tx) + 2u) & 0xffffcu) == 0 || (((uint32_t)__double2loint(ty) + 2u) & 0xffffcu) == 0) {
                int X, Y;
                map_pixel_exact<INTERP>(X0 + mx[j], Y0 + my[j], W[j], X, Y);
                rawX[j] = (uint32_t)X ^ 0x80000000u;
                rawY[j] = (uint32_t)Y ^ 0x80000000u;
            }
        }
    }
    constexpr int SH = INTERP == kLinear ? kInterBits : 0;
    constexpr int PB = lds_pixel_bytes<T, C>();
    Pixel<T, C> v[PPL];
    if constexpr (sizeof(T) == 1) {
        const uint32_t* l = reinterpret_cast<const uint32_t*>(lds_px);
        uint32_t p00[PPL], p01[PPL], p10[PPL], p11[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint32_t idx = (rawY[j] >> SH) * npx + (rawX[j] >> SH) + idx_bias;
            p00[j] = l[idx];
            if (INTERP == kLinear) {
                p01[j] = l[idx + 1];
                p10[j] = l[idx + npx];
                p11[j] = l[idx + npx + 1];
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; j++)
            v[j].packed = INTERP == kLinear ? blend_u8_packed<C>(p00[j], p01[j], p10[j], p11[j], rawX[j] & 31u, rawY[j] & 31u) : p00[j];
    } else {
        float t0[PPL][2 * C], t1[PPL][2 * C];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint32_t idx = (rawY[j] >> SH) * npx + (rawX[j] >> SH) + idx_bias;
            const float* l0 = reinterpret_cast<const float*>(lds_px + (size_t)idx * PB);
            const float* l1 = l0 + (size_t)npx * C;
#pragma unroll
            for (int k = 0; k < (INTERP == kLinear ? 2 * C : C); k++) {
                t0[j][k] = l0[k];
                if (INTERP == kLinear) t1[j][k] = l1[k];
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (INTERP == kLinear) {
                float w00, w01, w10, w11;
                weights_f32((int)(rawX[j] & 31u), (int)(rawY[j] & 31u), w00, w01, w10, w11);
#pragma unroll
                for (int k = 0; k < C; k++) v[j].v[k] = blend_f32(t0[j][k], t0[j][k + C], t1[j][k], t1[j][k + C], w00, w01, w10, w11);
            } else {
#pragma unroll
                for (int k = 0; k < C; k++) v[j].v[k] = t0[j][k];
            }
        }
    }
    // full-lane vector store (the caller checked alignment and that the lane's pixels are inside the row)
    if constexpr (sizeof(T) == 1 && C == 3) {
        U3 o;
        o.x = v[0].packed | (v[1].packed << 24);
        o.y = (v[1].packed >> 8) | (v[2].packed << 16);
        o.z = (v[2].packed >> 16) | (v[3].packed << 8);
        *reinterpret_cast<U3*>(dptr) = o;
    } else if constexpr (sizeof(T) == 1) {
        *reinterpret_cast<uint4*>(dptr) = make_uint4(v[0].packed, v[1].packed, v[2].packed, v[3].packed);
    } else {
        float f[PPL * C];
#pragma unroll
        for (int j = 0; j < PPL; j++)
#pragma unroll
            for (int k = 0; k < C; k++) f[j * C + k] = v[j].v[k];
        if constexpr ((PPL * C) % 4 == 0) {
#pragma unroll
            for (int k = 0; k < PPL * C / 4; k++) reinterpret_cast<float4*>(dptr)[k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < PPL * C / 2; k++) reinterpret_cast<float2*>(dptr)[k] = make_float2(f[2 * k], f[2 * k + 1]);
        }
    }
}

// Source region of a band of tile rows, shared through LDS.
struct Region {
    int rx0, rx1, ry0, ry1;  // inclusive, clipped to the image; rx1 < rx0 = empty
    int ax0, npx, rows;      // staged extent (x aligned to 4 pixels)
    bool ok;                 // corner box is trustworthy (W keeps its sign)
    bool interior;           // not clipped by the image: every tap of every pixel lies inside
};

// Diagnostic build only (-DBEVWARP_TIMING): wave 0 of every workgroup adds the shader-clock ticks it spent
// in each phase to g_phase[]; never compiled into the shipped library (tools/phases.py reads it).
#ifdef BEVWARP_TIMING
__device__ unsigned long long g_phase[16];
#define STAMP(i)                                                                      \
    do {                                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                 \
        if (threadIdx.x == 0) atomicAdd(&g_phase[i], now_ - stamp_);                  \
        stamp_ = now_;                                                                \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

#ifndef BEVWARP_WAVES_PER_EU
#define BEVWARP_WAVES_PER_EU 4
#endif
template <typename T, int C, int INTERP>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(BEVWARP_WAVES_PER_EU, 8))) void warp_tiles(const WarpArgs a) {
    constexpr int PPL = pixels_per_lane<T>();
    constexpr int TW = kLX * PPL;
    constexpr bool kStagedFmt = has_staged_path<T, C>();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_corner[4][4];  // X, Y (whole pixels) and sign of W per corner

    double* rowtab = reinterpret_cast<double*>(smem);
    uint8_t* lds_px = smem + kRowTabBytes;

    // ---- which tile: XCD-aware order.  Workgroups are dealt round-robin over the 8 XCDs, so ids
    // b and b + 8 share an L2; give each XCD one contiguous run of (frame, tile) items in raster order.
    const uint32_t item = (blockIdx.x & 7u) * (uint32_t)a.chunk + (blockIdx.x >> 3);
    if (item >= (uint32_t)a.total_tiles) return;
    const uint32_t frame_idx = fast_div(item, a.tpf_magic, (uint32_t)a.tiles_per_frame);
    const uint32_t t = item - frame_idx * (uint32_t)a.tiles_per_frame;
    const uint32_t ty = fast_div(t, a.tx_magic, (uint32_t)a.tiles_x), tx = t - ty * (uint32_t)a.tiles_x;
    const int x0 = (int)tx * TW, y0 = (int)ty * a.tile_h;
    const int tid = threadIdx.x;

    const uint8_t* __restrict__ frame = a.src + (int64_t)frame_idx * a.src_fs;
    uint8_t* __restrict__ dframe = a.dst + (int64_t)frame_idx * a.dst_fs;
    const double* __restrict__ M = a.minv + (int64_t)frame_idx * a.m_stride;
    double Mr[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Mr[i] = M[i];

    const int x_last = min(x0 + TW, a.dst_w) - 1, y_tile_last = min(y0 + a.tile_h, a.dst_h) - 1;
#ifdef BEVWARP_TIMING
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) atomicAdd(&g_phase[15], 1ull);
#endif

    // ---- per-lane constants: the lane's PPL pixels sit at x1 = x - bx inside their evaluation block
    const int lane = tid & 63, wave = tid >> 6;
    const int lxi = lane & (kLX - 1), lyi = lane >> 4;
    const int xg = x0 + lxi * PPL;
    const int tile_bx = (int)(fast_div((uint32_t)x0, a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0);
    // every pixel of the tile shares one evaluation block (always true for dst_h >= 16: bw0 = 64 or dst_w)
    const bool one_bx = (int)(fast_div((uint32_t)(x0 + TW - 1), a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0) == tile_bx;
    int bxj[PPL];
    double mx[PPL], my[PPL], mw[PPL];
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        const int x = xg + j;
        bxj[j] = one_bx ? tile_bx : (int)(fast_div((uint32_t)x, a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0);
        const double x1 = (double)(x - bxj[j]);
        mx[j] = Mr[0] * x1;
        my[j] = Mr[3] * x1;
        mw[j] = Mr[6] * x1;
    }
    const int nvalid_x = max(0, min(PPL, a.dst_w - xg));
    SrcView view;
    view.frame = frame;
    view.rs = a.src_rs;
    view.w = a.src_w;
    view.h = a.src_h;
#pragma unroll
    for (int k = 0; k < 4; k++) view.bf[k] = a.bval_f[k];
    view.bu = (uint32_t)a.bval_u8[0] | ((uint32_t)a.bval_u8[1] << 8) | ((uint32_t)a.bval_u8[2] << 16) | ((uint32_t)a.bval_u8[3] << 24);

    STAMP(0);  // matrix load + per-lane constants
    // ---- bands: the whole tile if its source region fits the LDS budget, else 16 rows at a time
    const bool can_stage = kStagedFmt && a.src_vec_ok;
    int band_h = a.tile_h;
    bool first = true;
    for (int y_lo = y0; y_lo <= y_tile_last;) {
        const int y_hi = min(y_lo + band_h, y_tile_last + 1) - 1;
        // -- corners of the band (approximate chain is enough: the box gets a 2-pixel margin) and row table
        if (!first) __syncthreads();  // the previous band's LDS reads are done
        first = false;
        if (can_stage && tid < 4) {
            const int cx = (tid & 1) ? x_last : x0, cy = (tid & 2) ? y_hi : y_lo;
            const int bx = (int)(fast_div((uint32_t)cx, a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0);
            double X0, Y0, W0;
            row_terms(Mr, bx, cy, X0, Y0, W0);
            const double x1 = (double)(cx - bx);
            const double W = W0 + Mr[6] * x1;
            const double r = rcp_newton(W);
            const double px = (X0 + Mr[0] * x1) * r, py = (Y0 + Mr[3] * x1) * r;
            // whole-pixel coordinate, saturated well inside int range; NaN -> flagged through sign 0
            const bool fin = fabs(px) < 1e9 && fabs(py) < 1e9;
            s_corner[tid][0] = fin ? (int)floor(px) : 0;
            s_corner[tid][1] = fin ? (int)floor(py) : 0;
            // W must keep its sign and a sane magnitude on the band (it is linear, so the corners bound it):
            // the shared reciprocal multiplies up to four of them.
            s_corner[tid][2] = (!fin || !(fabs(W) > 1e-60 && fabs(W) < 1e60)) ? 0 : (W > 0 ? 1 : -1);
        }
        if (one_bx && tid >= 64 && tid < 64 + (y_hi - y_lo + 1)) {
            double X0, Y0, W0;
            row_terms(Mr, tile_bx, y_lo + tid - 64, X0, Y0, W0);
            rowtab[(tid - 64) * 3 + 0] = X0;
            rowtab[(tid - 64) * 3 + 1] = Y0;
            rowtab[(tid - 64) * 3 + 2] = W0;
        }
        __syncthreads();
        STAMP(1);  // corners + row table + barrier

        Region R;
        R.rx0 = 0, R.rx1 = -1, R.ry0 = 0, R.ry1 = -1, R.ax0 = 0, R.npx = 0, R.rows = 0, R.ok = false, R.interior = false;
        bool staged = false;
        if (can_stage) {
            int c[4][3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) c[i][k] = __builtin_amdgcn_readfirstlane(s_corner[i][k]);
            const int sg = c[0][2] + c[1][2] + c[2][2] + c[3][2];
            if (sg == 4 || sg == -4) {
                // +-2 px: rounding inside the band and the approximate corner chain; +1: right / lower tap
                const int lx = min(min(c[0][0], c[1][0]), min(c[2][0], c[3][0])) - 2;
                const int hx = max(max(c[0][0], c[1][0]), max(c[2][0], c[3][0])) + 3;
                const int ly = min(min(c[0][1], c[1][1]), min(c[2][1], c[3][1])) - 2;
                const int hy = max(max(c[0][1], c[1][1]), max(c[2][1], c[3][1])) + 3;
                R.ok = true;
                R.rx0 = max(lx, 0);
                R.rx1 = min(hx, a.src_w - 1);
                R.ry0 = max(ly, 0);
                R.ry1 = min(hy, a.src_h - 1);
                R.interior = lx >= 0 && hx <= a.src_w - 1 && ly >= 0 && hy <= a.src_h - 1;
                if (R.rx0 <= R.rx1 && R.ry0 <= R.ry1) {
                    R.ax0 = R.rx0 & ~3;
                    R.npx = (R.rx1 | 3) - R.ax0 + 1;
                    R.rows = R.ry1 - R.ry0 + 1;
                    staged = (int64_t)R.rows * R.npx * lds_pixel_bytes<T, C>() <= (int64_t)a.lds_bytes;
                }
            }
            const bool outside = R.ok && (R.rx0 > R.rx1 || R.ry0 > R.ry1);  // nothing of the band samples the image
            if (!staged && !outside && band_h > kBandRows) {
                band_h = kBandRows;  // box too big for LDS (or not trustworthy): retry this tile in 16-row bands
                continue;
            }
        }
        STAMP(2);  // region
        if (staged) {
            if constexpr (kStagedFmt) stage_region<T, C>(a, frame, lds_px, R.ax0, R.ry0, R.rows, R.npx, tid);
            STAMP(3);  // staging loads + LDS writes of this wave
            __syncthreads();
            STAMP(4);  // staging barrier
        }
        const bool interior = staged && R.interior;
        const uint32_t fast_w = (uint32_t)max(R.rx1 - R.rx0 + (INTERP == kLinear ? 0 : 1), 0);  // sx - rx0 < fast_w <=> taps inside
        const uint32_t fast_h = (uint32_t)max(R.ry1 - R.ry0 + (INTERP == kLinear ? 0 : 1), 0);
        const int px_bias = -R.ry0 * R.npx - R.ax0;  // LDS pixel index = sy * npx + sx + px_bias

        // -- fast rows: every tap of the band lies inside the staged region, full lanes, one evaluation block
        if constexpr (kStagedFmt) {
            if (interior && one_bx && x0 + TW <= a.dst_w && a.dst_vec_ok) {
                constexpr uint32_t kRawBias = INTERP == kLinear ? (1u << 26) : (1u << 31);  // raw coordinates carry + 2^31
                const uint32_t idx_bias = (uint32_t)px_bias - kRawBias * (uint32_t)R.npx - kRawBias;
                for (int y = y_lo + wave * kLY + lyi; y <= y_hi; y += kBandRows)
                    row_fast<T, C, INTERP, PPL>(rowtab + (y - y_lo) * 3, mx, my, mw, lds_px, (uint32_t)R.npx, idx_bias,
                                                dframe + (int64_t)y * a.dst_rs + (int64_t)xg * C * sizeof(T));
                y_lo = y_hi + 1;
                STAMP(5);  // fast rows
                continue;
            }
        }
        // -- general rows (image border, ragged tiles, unstaged bands)
        const bool approx_ok = R.ok;  // the shared-reciprocal shortcut needs W of one sign and sane magnitude
        for (int y = y_lo + wave * kLY + lyi; y <= y_hi; y += kBandRows) {
            if (nvalid_x == 0) break;
            double X0, Y0, W0;
            if (one_bx) {
                X0 = rowtab[(y - y_lo) * 3 + 0];
                Y0 = rowtab[(y - y_lo) * 3 + 1];
                W0 = rowtab[(y - y_lo) * 3 + 2];
            } else {
                row_terms(Mr, bxj[0], y, X0, Y0, W0);
            }
            double Wj[PPL], Xn[PPL], Yn[PPL];
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                if (!one_bx && j > 0 && bxj[j] != bxj[j - 1]) row_terms(Mr, bxj[j], y, X0, Y0, W0);
                Wj[j] = W0 + mw[j];
                Xn[j] = X0 + mx[j];
                Yn[j] = Y0 + my[j];
            }
            // one reciprocal for the lane's pixels: 1 / (W0 W1 [W2 W3]) and back-substitution
            double rj[PPL];
            if constexpr (PPL == 4) {
                const double p01 = Wj[0] * Wj[1], p23 = Wj[2] * Wj[3];
                const double inv = rcp_newton(p01 * p23);
                const double i01 = inv * p23, i23 = inv * p01;
                rj[0] = i01 * Wj[1];
                rj[1] = i01 * Wj[0];
                rj[2] = i23 * Wj[3];
                rj[3] = i23 * Wj[2];
            } else {
                const double inv = rcp_newton(Wj[0] * Wj[1]);
                rj[0] = inv * Wj[1];
                rj[1] = inv * Wj[0];
            }
            int X[PPL], Y[PPL];
            uint32_t redo = 0;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const bool okx = round_fast<INTERP>(Xn[j] * rj[j], X[j]);
                const bool oky = round_fast<INTERP>(Yn[j] * rj[j], Y[j]);
                if (!(approx_ok && okx && oky)) redo |= 1u << j;
            }
            if (redo) {  // rare: rounding ties, W == 0, non-finite or far-away coordinates -> the exact chain decides
#pragma unroll
                for (int j = 0; j < PPL; j++)
                    if (redo & (1u << j)) map_pixel_exact<INTERP>(Xn[j], Yn[j], Wj[j], X[j], Y[j]);
            }
            Pixel<T, C> v[PPL];
            uint32_t slow = 0;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const int sx = INTERP == kLinear ? (X[j] >> kInterBits) : X[j], sy = INTERP == kLinear ? (Y[j] >> kInterBits) : Y[j];
                bool fast = interior;
                if (!interior) fast = staged && (uint32_t)(sx - R.rx0) < fast_w && (uint32_t)(sy - R.ry0) < fast_h;
                if (fast) {
                    if constexpr (kStagedFmt)
                        v[j] = sample_lds<T, C, INTERP>(lds_px, (uint32_t)(sy * R.npx + sx + px_bias), (uint32_t)R.npx, X[j] & 31, Y[j] & 31);
                } else {
                    slow |= 1u << j;
                }
            }
            if (slow) {  // image border, unstaged bands
#pragma unroll
                for (int j = 0; j < PPL; j++)
                    if (slow & (1u << j)) v[j] = sample_global<T, C, INTERP>(view, X[j], Y[j]);
            }
            store_pixels<T, C, PPL>(a, dframe + (int64_t)y * a.dst_rs, xg, nvalid_x, v);
        }
        y_lo = y_hi + 1;
        STAMP(6);  // general rows
    }
}

#ifdef BEVWARP_TIMING
}  // namespace
hipError_t debug_read_phases(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
    }
    return e;
}
namespace {
#endif

// ===================================================================================================
// warp_gather: the same warp WITHOUT LDS staging.  Every lane loads its taps straight from global
// memory (two unaligned 8-byte loads per 8-bit RGB pixel: the 6 bytes of a tap pair + 2 spare) and
// relies on the vector L1 / per-XCD L2 for the reuse between neighbouring pixels.  No barriers, no
// LDS: occupancy is bounded by registers only and a wave keeps all the loads of its PPL pixels in
// flight.  Requires every TW-wide tile to lie in one evaluation block (host checks).
// ===================================================================================================
#ifndef BEVWARP_F32_WAVES
#define BEVWARP_F32_WAVES 4
#endif
template <int N>
struct Bytes {
    uint32_t w[N / 4];
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (register budget: 4 waves per SIMD -- what the FAST row loop needs without spilling; the rare row
// classes may spill)
// RS4: the source row stride is a multiple of 4 bytes (both tap rows of a pixel then share one window alignment)
template <typename T, int C, int INTERP, bool RS4 = false>
__global__ __launch_bounds__(kWG) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 1 ? (C == 3 && INTERP == kLinear ? 3 : 4) : BEVWARP_F32_WAVES))) void warp_gather(const WarpArgs a) {
    constexpr int PPL = pixels_per_lane<T>();
    constexpr int GX = kGatherLX, GY = 64 / kGatherLX, GROWS = GY * (kWG / 64);  // lanes along x / y, rows per pass
    constexpr int TW = GX * PPL;
    constexpr int PBs = (int)sizeof(T) * C;                      // source bytes per pixel
    constexpr int TAPB = INTERP == kLinear ? 2 * PBs : PBs;      // bytes of one row's taps
    constexpr int LOADB = (TAPB + 3) & ~3;                       // loaded per row (whole dwords)
    constexpr double kScale = INTERP == kLinear ? 33554432.0 : 1048576.0;
    constexpr double kMagic = 6755399441055744.0 + 524288.0;
    constexpr int SH = INTERP == kLinear ? kInterBits : 0;
    // wave-private LDS row used by the interior path to transpose results into store order (u8: one dword per
    // pixel, float: C floats per pixel)
    __shared__ __attribute__((aligned(16))) uint32_t s_tr[kWG / 64][64 * PPL * (sizeof(T) == 1 ? 1 : C)];

    const uint32_t item = (blockIdx.x & 7u) * (uint32_t)a.chunk + (blockIdx.x >> 3);
    if (item >= (uint32_t)a.total_tiles) return;
    const uint32_t frame_idx = fast_div(item, a.tpf_magic, (uint32_t)a.tiles_per_frame);
    const uint32_t t = item - frame_idx * (uint32_t)a.tiles_per_frame;
    const uint32_t ty = fast_div(t, a.tx_magic, (uint32_t)a.tiles_x), tx = t - ty * (uint32_t)a.tiles_x;
    const int x0 = (int)tx * TW, y0 = (int)ty * a.tile_h;
    const int tid = threadIdx.x;
    const uint8_t* __restrict__ frame = a.src + (int64_t)frame_idx * a.src_fs;
    uint8_t* __restrict__ dframe = a.dst + (int64_t)frame_idx * a.dst_fs;
    const double* __restrict__ M = a.minv + (int64_t)frame_idx * a.m_stride;
    double Mr[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Mr[i] = M[i];

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, in an SGPR
    const int lxi = lane % GX, lyi = lane / GX;
    const int xg = x0 + lxi * PPL;
    // evaluation block of THIS lane's pixels (a lane never straddles two: host checks bw0 % PPL == 0 or one block per row)
    const int tile_bx = (int)(fast_div((uint32_t)min(xg, a.dst_w - 1), a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0);
    double mx[PPL], my[PPL], mw[PPL];
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        const double x1 = (double)(xg + j - tile_bx);
        mx[j] = Mr[0] * x1;
        my[j] = Mr[3] * x1;
        mw[j] = Mr[6] * x1;
    }
    // row-independent products of the row terms (X0 = (M0*bx + M1*y) + M2)
    const double bxd = (double)tile_bx;
    const double bX = Mr[0] * bxd, bY = Mr[3] * bxd, bW = Mr[6] * bxd;
    const int nvalid_x = max(0, min(PPL, a.dst_w - xg));
    const int y_last = min(y0 + a.tile_h, a.dst_h) - 1;
    SrcView view;
    view.frame = frame;
    view.rs = a.src_rs;
    view.w = a.src_w;
    view.h = a.src_h;
#pragma unroll
    for (int k = 0; k < 4; k++) view.bf[k] = a.bval_f[k];
    view.bu = (uint32_t)a.bval_u8[0] | ((uint32_t)a.bval_u8[1] << 8) | ((uint32_t)a.bval_u8[2] << 16) | ((uint32_t)a.bval_u8[3] << 24);
    // an unguarded row load reads LOADB bytes from the tap: it must stay inside the row
    const int sx_lim = (int)(((int64_t)a.src_w * PBs - LOADB) / PBs);  // largest sx with sx*PBs + LOADB <= w*PBs
    const int sy_lim = a.src_h - (INTERP == kLinear ? 2 : 1);
    const bool any_fast = (int64_t)a.src_w * PBs >= LOADB && sy_lim >= 0;
    const uint32_t sx_max = any_fast ? (uint32_t)sx_lim : 0u, sy_max = any_fast ? (uint32_t)sy_lim : 0u;

    // -- one row of the lane: exact fixed-point coordinates; returns whether unguarded loads are allowed
    auto coords = [&](int y, int (&X)[PPL], int (&Y)[PPL]) -> bool {
        const double dy = (double)y;
        const double X0 = (bX + Mr[1] * dy) + Mr[2], Y0 = (bY + Mr[4] * dy) + Mr[5], W0 = (bW + Mr[7] * dy) + Mr[8];
        double W[PPL], r[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) W[j] = W0 + mw[j];
        if constexpr (PPL == 4) {
            const double p01 = W[0] * W[1], p23 = W[2] * W[3];
            const double inv = rcp_newton(p01 * p23) * kScale;
            const double i01 = inv * p23, i23 = inv * p01;
            r[0] = i01 * W[1];
            r[1] = i01 * W[0];
            r[2] = i23 * W[3];
            r[3] = i23 * W[2];
        } else {
            const double inv = rcp_newton(W[0] * W[1]) * kScale;
            r[0] = inv * W[1];
            r[1] = inv * W[0];
        }
        // W is linear along the row: its end values bound the lane's pixels.  The shared reciprocal needs one
        // sign and a sane magnitude (no overflow / denormals in the product), else the exact chain runs.
        const double wa = W[0], wb = W[PPL - 1];
        const bool w_ok = (wa > 0) == (wb > 0) && fabs(wa) > 1e-60 && fabs(wa) < 1e60 && fabs(wb) > 1e-60 && fabs(wb) < 1e60;
        uint32_t tie = 0xffffffffu, expo = 0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const double tx_ = (X0 + mx[j]) * r[j] + kMagic, ty_ = (Y0 + my[j]) * r[j] + kMagic;
            const uint32_t lox = (uint32_t)__double2loint(tx_), loy = (uint32_t)__double2loint(ty_);
            const uint32_t hix = (uint32_t)__double2hiint(tx_), hiy = (uint32_t)__double2hiint(ty_);
            X[j] = (int)(__builtin_amdgcn_alignbit(hix, lox, 20) ^ 0x80000000u);
            Y[j] = (int)(__builtin_amdgcn_alignbit(hiy, loy, 20) ^ 0x80000000u);
            tie = min(tie, min((lox + 2u) & 0xffffcu, (loy + 2u) & 0xffffcu));
            // the mantissa trick holds while the sum keeps the magic's exponent (|coordinate| < 2^31)
            expo |= (hix ^ 0x43300000u) | (hiy ^ 0x43300000u);
        }
        if (tie == 0 || (expo >> 20) != 0 || !w_ok) {  // rare: tie window, non-finite / huge coordinates, W near 0
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const double tx_ = (X0 + mx[j]) * r[j] + kMagic, ty_ = (Y0 + my[j]) * r[j] + kMagic;
                const uint32_t lox = (uint32_t)__double2loint(tx_), loy = (uint32_t)__double2loint(ty_);
                const uint32_t hix = (uint32_t)__double2hiint(tx_), hiy = (uint32_t)__double2hiint(ty_);
                // |coordinate| >= 2^22 (1/32 px units) is far outside any admissible image but no longer exact: redo too
                const bool far = (uint32_t)(X[j] + (1 << 22)) >= (1u << 23) || (uint32_t)(Y[j] + (1 << 22)) >= (1u << 23);
                if (!w_ok || far || ((lox + 2u) & 0xffffcu) == 0 || ((loy + 2u) & 0xffffcu) == 0 || ((hix ^ 0x43300000u) >> 20) != 0 ||
                    ((hiy ^ 0x43300000u) >> 20) != 0)
                    map_pixel_exact<INTERP>(X0 + mx[j], Y0 + my[j], W[j], X[j], Y[j]);
            }
        }
        // taps fully inside the frame (and the dword-rounded loads inside their row)?
        bool inb = any_fast && nvalid_x == PPL;
#pragma unroll
        for (int j = 0; j < PPL; j++) inb = inb && (uint32_t)(X[j] >> SH) <= sx_max && (uint32_t)(Y[j] >> SH) <= sy_max;
        return inb;
    };
    // -- issue the row's tap loads (32-bit offsets: a frame is < 2 GiB)
    const uint32_t rs32 = (uint32_t)a.src_rs;
    auto issue = [&](const int (&X)[PPL], const int (&Y)[PPL], Bytes<LOADB> (&t0)[PPL], Bytes<LOADB> (&t1)[PPL]) {
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const uint32_t off = (uint32_t)(Y[j] >> SH) * rs32 + (uint32_t)(X[j] >> SH) * (uint32_t)PBs;
#if defined(BEVWARP_ABLATE) && (BEVWARP_ABLATE & 1)  // diagnostic builds only: no tap loads
            for (int k = 0; k < LOADB / 4; k++) t0[j].w[k] = off + k, t1[j].w[k] = off ^ k;
#else
            __builtin_memcpy(&t0[j], frame + off, LOADB);
            if (INTERP == kLinear) __builtin_memcpy(&t1[j], frame + (off + rs32), LOADB);
#endif
        }
    };
    // -- blend (or border-sample) and store the row
    auto finish = [&](int y, bool fast, const int (&X)[PPL], const int (&Y)[PPL], const Bytes<LOADB> (&t0)[PPL], const Bytes<LOADB> (&t1)[PPL]) {
        Pixel<T, C> v[PPL];
        if (__builtin_expect(fast, 1)) {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const uint32_t fx = (uint32_t)X[j] & 31u, fy = (uint32_t)Y[j] & 31u;
                if constexpr (sizeof(T) == 1) {
                    if (INTERP == kNearest) {
                        v[j].packed = C == 4 ? t0[j].w[0] : (t0[j].w[0] & ((1u << (8 * (C & 3))) - 1u));
                    } else {
                        // left tap = bytes 0..C-1, right tap = bytes C..2C-1 of the row's load
                        uint32_t l0, r0, l1, r1;
                        if constexpr (C == 4) {
                            l0 = t0[j].w[0], r0 = t0[j].w[1], l1 = t1[j].w[0], r1 = t1[j].w[1];
                        } else if constexpr (C == 3) {
                            l0 = t0[j].w[0], r0 = __builtin_amdgcn_alignbyte(t0[j].w[1], t0[j].w[0], 3);
                            l1 = t1[j].w[0], r1 = __builtin_amdgcn_alignbyte(t1[j].w[1], t1[j].w[0], 3);
                        } else {
                            l0 = t0[j].w[0], r0 = t0[j].w[0] >> (8 * C), l1 = t1[j].w[0], r1 = t1[j].w[0] >> (8 * C);
                        }
#if defined(BEVWARP_ABLATE) && (BEVWARP_ABLATE & 2)  // diagnostic builds only: no blend arithmetic
                        v[j].packed = (l0 ^ r0 ^ l1 ^ r1) + fx + fy;
#else
                        v[j].packed = blend_u8_packed<C>(l0, r0, l1, r1, fx, fy);  // bytes >= C of the taps are never selected
#endif
                    }
                } else {
                    const float* f0 = reinterpret_cast<const float*>(&t0[j]);
                    const float* f1 = reinterpret_cast<const float*>(&t1[j]);
                    if (INTERP == kNearest) {
#pragma unroll
                        for (int k = 0; k < C; k++) v[j].v[k] = f0[k];
                    } else {
                        float w00, w01, w10, w11;
                        weights_f32((int)fx, (int)fy, w00, w01, w10, w11);
#pragma unroll
                        for (int k = 0; k < C; k++) v[j].v[k] = blend_f32(f0[k], f0[k + C], f1[k], f1[k + C], w00, w01, w10, w11);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < PPL; j++) v[j] = sample_global<T, C, INTERP>(view, X[j], Y[j]);
        }
#if defined(BEVWARP_ABLATE) && (BEVWARP_ABLATE & 4)  // diagnostic builds only: keep the values alive, store one row in 64
        if ((y & 63) == 0)
#endif
        store_pixels<T, C, PPL>(a, dframe + (int64_t)y * a.dst_rs, xg, nvalid_x, v);
    };

    // -- row-classified path: tiles that are a whole number of evaluation blocks wide.  Every row segment (TW pixels
    // of one row) is classified on its own from its two end pixels: a projective map with W of one sign sends the
    // segment to a straight segment of the source plane, so when both ends sample inside the frame (by a pixel of
    // margin) every pixel does (FAST: no bounds / range / sign tests, unguarded loads), and when both ends lie beyond
    // the same edge no pixel touches the frame (OUT: the border value).  When the frame's edge crosses the segment
    // (EDGE) the coordinates of the fast chain still hold and only the pixels next to the edge take guarded taps;
    // when W changes sign or is tiny, or coordinates pass 2^26 px (SLOW), the row takes the exact per-pixel chain.
    if (GY == 1 && a.bw0 == 64 && any_fast && (int64_t)a.src_w * PBs >= 32 && x0 + TW <= a.dst_w && a.dst_vec_ok) {
        // lane-INTERLEAVED ownership: pixel j of lane l is x0 + 64 j + l, so one load instruction covers 64 consecutive
        // destination pixels whose taps sit in a handful of cache lines.  Pixel j lies in evaluation block j of the
        // tile and x1 = l for every j.  Results are transposed to consecutive-per-lane order through a wave-private
        // LDS row before the (contiguous) store.
        enum { kFast = 0, kOut = 1, kEdge = 2, kSlow = 3 };
        // 8-bit RGB bilinear: a tap pair (6 bytes at any byte address) is fetched as the ALIGNED 12-byte window around
        // it and funnel-shifted into place.  The texture path turns byte-unaligned 8-byte gathers that miss L1 into
        // data at ~50 cycles per wave instruction and 4-byte-aligned 12-byte ones at ~18 (tools/ubench_stream.hip).
        constexpr bool kAligned = sizeof(T) == 1 && C == 3 && INTERP == kLinear;
        constexpr int WINB = kAligned ? 12 : LOADB;  // bytes a FAST row loads per tap row
        constexpr int kInMargin = kAligned ? 2 : 1;   // FAST: both ends inside by this many pixels (the aligned window
                                                      // starts up to 3 bytes early: never before its row)
        const int sxw_lim = (int)(((int64_t)a.src_w * PBs - WINB) / PBs);  // largest sx with sx*PBs + WINB <= w*PBs
        const uint32_t fa = kAligned ? (uint32_t)(reinterpret_cast<uintptr_t>(frame) & 3u) : 0u;
        const uint8_t* frame_al = frame - fa;  // 4-byte aligned (frames need not be)
        const uint8_t* frame_al_r1 = frame_al + rs32;
        // Row terms.  The reference's chain is X0 = (M0*bx + M1*y) + M2, X = X0 + M0*(x - bx) per evaluation block;
        // the fast chain below only has to land within 2^-20 of a coordinate unit of it (anything closer than
        // 2^-19 to a rounding boundary is redone exactly), which leaves ~12 bits of slack over float64 rounding.
        // So a row evaluates the chain once, for the segment's first pixel (UX, UY, UW: wave-uniform), and adds
        // per-lane constants M0*(64 j + lane): 22 operations per row instead of 40.  exact_px() restates the
        // reference's order of operations for the rare exact redo.
        const double x1d = (double)lane;
        double cxj[PPL], cyj[PPL], cwj[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const double dj = (double)(64 * j + lane);
            cxj[j] = Mr[0] * dj;
            cyj[j] = Mr[3] * dj;
            cwj[j] = Mr[6] * dj;
        }
        auto uniform_f64 = [](double v) {  // a wave-uniform double, moved to scalar registers
            return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
        };
        const double bX0 = uniform_f64(Mr[0] * (double)x0), bY0 = uniform_f64(Mr[3] * (double)x0), bW0 = uniform_f64(Mr[6] * (double)x0);
        auto exact_px = [&](int y, int j, double& Xn, double& Yn, double& Wn) __attribute__((always_inline)) {
            double X0, Y0, W0;
            row_terms(Mr, x0 + 64 * j, y, X0, Y0, W0);
            Xn = X0 + Mr[0] * x1d;
            Yn = Y0 + Mr[3] * x1d;
            Wn = W0 + Mr[6] * x1d;
        };
        // OUT rows may be filled with the border value when blending four border taps gives it back exactly:
        // always for 8-bit (the fixed-point weights sum to 2^15) and nearest; for float bilinear only for +0
        bool fill_ok = true;
        if (sizeof(T) == 4 && INTERP == kLinear)
            for (int k = 0; k < C; k++) fill_ok = fill_ok && __float_as_uint(a.bval_f[k]) == 0u;
        int out_side = 0;  // set by coords_s for an OUT row: which frame edge the segment lies beyond, and the sign of W
        auto coords_s = [&](int y, uint32_t (&RX)[PPL], uint32_t (&RY)[PPL]) -> int {
            double W[PPL], Xn[PPL], Yn[PPL], r[PPL];
            const double dy = (double)y;
            const double UX = (bX0 + Mr[1] * dy) + Mr[2], UY = (bY0 + Mr[4] * dy) + Mr[5], UW = (bW0 + Mr[7] * dy) + Mr[8];
#if defined(BEVWARP_ABLATE) && (BEVWARP_ABLATE & 8)  // diagnostic builds only: identity map instead of the projective chain
            for (int j = 0; j < PPL; j++) {
                RX[j] = 0x80000000u + ((uint32_t)(x0 + 64 * j + lane) << 5) + 7u + (uint32_t)(UX > 1e300);
                RY[j] = 0x80000000u + ((uint32_t)y << 5) + 9u;
            }
            return kFast;
#endif
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                W[j] = UW + cwj[j];
                Xn[j] = UX + cxj[j];
                Yn[j] = UY + cyj[j];
            }
            if constexpr (PPL == 4) {
                const double p01 = W[0] * W[1], p23 = W[2] * W[3];
                const double inv = rcp_newton(p01 * p23) * kScale;
                const double i01 = inv * p23, i23 = inv * p01;
                r[0] = i01 * W[1];
                r[1] = i01 * W[0];
                r[2] = i23 * W[3];
                r[3] = i23 * W[2];
            } else {
                const double inv = rcp_newton(W[0] * W[1]) * kScale;
                r[0] = inv * W[1];
                r[1] = inv * W[0];
            }
            uint32_t tie = 0xffffffffu, hxa = 0, hya = 0, hxb = 0, hyb = 0;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                // kMagic + 2: the low word then reads (fraction + 2) and one mask tests the window [-2, 2) around a
                // rounding boundary; outside that window the extra 2 does not change the integer part
                const double tx_ = Xn[j] * r[j] + (kMagic + 2.0), ty_ = Yn[j] * r[j] + (kMagic + 2.0);
                const uint32_t lox = (uint32_t)__double2loint(tx_), loy = (uint32_t)__double2loint(ty_);
                const uint32_t hix = (uint32_t)__double2hiint(tx_), hiy = (uint32_t)__double2hiint(ty_);
                RX[j] = __builtin_amdgcn_alignbit(hix, lox, 20);
                RY[j] = __builtin_amdgcn_alignbit(hiy, loy, 20);
                tie = min(tie, min(lox & 0xffffcu, loy & 0xffffcu));
                if (j == 0) hxa = hix, hya = hiy;
                if (j == PPL - 1) hxb = hix, hyb = hiy;
            }
            // -- classify the segment from its ends (pixel 0 of lane 0, pixel PPL-1 of lane 63), in scalar registers.
            // The mantissa trick is valid while the sum keeps the magic's exponent (|coordinate| < 2^31 units).
            auto lane_u32 = [](uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); };  // (the builtin returns int)
            const uint32_t kExp = 0x43300000u;
            const uint32_t e_bad = ((lane_u32(hxa, 0) ^ kExp) | (lane_u32(hya, 0) ^ kExp) | (lane_u32(hxb, 63) ^ kExp) | (lane_u32(hyb, 63) ^ kExp)) >> 20;
            const uint32_t wa = lane_u32((uint32_t)__double2hiint(W[0]), 0), wb = lane_u32((uint32_t)__double2hiint(W[PPL - 1]), 63);
            const uint32_t ea = (wa >> 20) & 0x7ffu, eb = (wb >> 20) & 0x7ffu;  // 2^-199 .. 2^199: the shared reciprocal is safe
            const bool w_ok = ((wa ^ wb) >> 31) == 0 && ea - 824u <= 398u && eb - 824u <= 398u;
            constexpr uint32_t kCoordBias = INTERP == kLinear ? (1u << 26) : (1u << 31);
            const int sxa = (int)((lane_u32(RX[0], 0) >> SH) - kCoordBias), sya = (int)((lane_u32(RY[0], 0) >> SH) - kCoordBias);
            const int sxb = (int)((lane_u32(RX[PPL - 1], 63) >> SH) - kCoordBias), syb = (int)((lane_u32(RY[PPL - 1], 63) >> SH) - kCoordBias);
            constexpr int kM = kInMargin;
            const bool in = (uint32_t)(sxa - kM) <= (uint32_t)(sxw_lim - 2 * kM) && (uint32_t)(sxb - kM) <= (uint32_t)(sxw_lim - 2 * kM) &&
                            (uint32_t)(sya - kM) <= (uint32_t)(sy_lim - 2 * kM) && (uint32_t)(syb - kM) <= (uint32_t)(sy_lim - 2 * kM) &&
                            sxw_lim >= 2 * kM && sy_lim >= 2 * kM;
            int cls = kFast;
            if (__builtin_expect(!(e_bad == 0 && w_ok && in), 0)) {  // (the common class costs no further scalar work)
                const bool out = (sxa <= -3 && sxb <= -3) || (sxa > a.src_w && sxb > a.src_w) || (sya <= -3 && syb <= -3) || (sya > a.src_h && syb > a.src_h);
                // kEdge: the coordinates are good, taps need guards
                cls = !(e_bad == 0 && w_ok) ? kSlow : ((out && fill_ok) ? kOut : kEdge);
                out_side = ((sxa <= -3 && sxb <= -3) ? 1 : (sxa > a.src_w && sxb > a.src_w) ? 2 : (sya <= -3 && syb <= -3) ? 3 : 4) | (int)((wa >> 31) << 3);
            }
            if (tie == 0) {  // rare: within 2^-19 of a rounding tie -> the exact chain decides
#pragma unroll
                for (int j = 0; j < PPL; j++) {
                    const double tx_ = Xn[j] * r[j] + (kMagic + 2.0), ty_ = Yn[j] * r[j] + (kMagic + 2.0);
                    if (((uint32_t)__double2loint(tx_) & 0xffffcu) == 0 || ((uint32_t)__double2loint(ty_) & 0xffffcu) == 0) {
                        double Xq, Yq, Wq;
                        exact_px(y, j, Xq, Yq, Wq);
                        int Xe, Ye;
                        map_pixel_exact<INTERP>(Xq, Yq, Wq, Xe, Ye);
                        RX[j] = (uint32_t)Xe ^ 0x80000000u;
                        RY[j] = (uint32_t)Ye ^ 0x80000000u;
                    }
                }
            }
            return cls;
        };
        const uint8_t* frame_r1 = frame + rs32;
        // (rows that are not FAST load from offset 0: the row loop keeps one shape for every class)
        auto issue_s = [&](int cls, const uint32_t (&RX)[PPL], const uint32_t (&RY)[PPL], Bytes<WINB> (&t0)[PPL], Bytes<WINB> (&t1)[PPL],
                           uint32_t (&S0)[PPL], uint32_t (&S1)[PPL]) {
            // FAST rows sample inside the frame: 0 <= sx, sy < 2^15, so 16-bit fields drop the 2^31 bias and 24-bit
            // multiplies build the byte offset (the host guarantees row stride < 2^24 and frames < 2 GiB)
            const uint32_t rs_eff = cls == kFast ? rs32 : 0u, pb_eff = cls == kFast ? (uint32_t)PBs : 0u;
            const uint32_t o_base = kAligned ? (cls == kFast ? fa : 4u) : 0u;  // (4: a dummy window inside the frame)
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const uint32_t sx = __builtin_amdgcn_ubfe(RX[j], SH, 16), sy = __builtin_amdgcn_ubfe(RY[j], SH, 16);
                const uint32_t off = __umul24(sy, rs_eff) + (__umul24(sx, pb_eff) + o_base);
#if defined(BEVWARP_ABLATE) && (BEVWARP_ABLATE & 1)  // diagnostic builds only: no tap loads
                for (int k = 0; k < WINB / 4; k++) t0[j].w[k] = off + k, t1[j].w[k] = off ^ k;
                S0[j] = S1[j] = off;
#else
                if constexpr (kAligned && RS4) {  // second tap row: same window alignment, scalar base + row stride
                    const uint32_t offa = off & ~3u;
                    __builtin_memcpy(&t0[j], frame_al + offa, WINB);
                    __builtin_memcpy(&t1[j], frame_al_r1 + offa, WINB);
                    S0[j] = S1[j] = off << 3;  // funnel-shift amount (v_alignbit reads bits 4:0)
                } else if constexpr (kAligned) {
                    const uint32_t off1 = off + rs_eff;
                    __builtin_memcpy(&t0[j], frame_al + (off & ~3u), WINB);
                    __builtin_memcpy(&t1[j], frame_al + (off1 & ~3u), WINB);
                    S0[j] = off << 3, S1[j] = off1 << 3;  // funnel-shift amounts (v_alignbit reads bits 4:0)
                } else {
                    __builtin_memcpy(&t0[j], frame + off, LOADB);
                    if (INTERP == kLinear) __builtin_memcpy(&t1[j], frame_r1 + off, LOADB);  // second tap row: same offset, base + row stride
                }
#endif
            }
        };
        uint32_t* wtr = &s_tr[wave][0];
        // a finished row waits in registers (store order) until the NEXT row's loads have been issued: vmcnt
        // retires in issue order, so a store issued before those loads would have to complete before their data
        // can be used; issued after them it has a whole iteration to complete
        constexpr int kVec = sizeof(T) == 1 ? 64 : 64 * PPL * C / 4;  // 16-byte units in the wave's row segment
        constexpr int NQ = (kVec + 63) / 64;
        // LDS row -> registers in store order (u8: pixels 4l .. 4l+3 of the segment; float: 16-byte unit u*64 + l)
        auto read_back = [&](uint4 (&out)[NQ]) {
            asm volatile("" ::: "memory");  // compiler fence: one wave's LDS operations execute in program order
#pragma unroll
            for (int u = 0; u < NQ; u++) {
                const int q = u * 64 + lane;
                if (q < kVec) out[u] = reinterpret_cast<const uint4*>(wtr)[q];
            }
            asm volatile("" ::: "memory");  // (the next row's LDS writes cannot pass these reads)
        };
        auto finish_s = [&](const uint32_t (&RX)[PPL], const uint32_t (&RY)[PPL], const Bytes<WINB> (&t0)[PPL],
                            const Bytes<WINB> (&t1)[PPL], const uint32_t (&S0)[PPL], const uint32_t (&S1)[PPL]) {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const uint32_t fx = RX[j] & 31u, fy = RY[j] & 31u;
                if constexpr (sizeof(T) == 1) {
                    uint32_t px;
                    if (INTERP == kNearest) {
                        px = C == 4 ? t0[j].w[0] : (t0[j].w[0] & ((1u << (8 * (C & 3))) - 1u));
#if defined(BEVWARP_ABLATE) && (BEVWARP_ABLATE & 2)  // diagnostic builds only: no blend arithmetic
                    } else if constexpr (C == 3) {
                        px = (t0[j].w[0] ^ t0[j].w[1] ^ t1[j].w[0] ^ t1[j].w[1]) + fx + fy;
#endif
                    } else if constexpr (kAligned) {
                        const uint32_t a0 = __builtin_amdgcn_alignbit(t0[j].w[1], t0[j].w[0], S0[j]), a1 = __builtin_amdgcn_alignbit(t0[j].w[2], t0[j].w[1], S0[j]);
                        const uint32_t b0 = __builtin_amdgcn_alignbit(t1[j].w[1], t1[j].w[0], S1[j]), b1 = __builtin_amdgcn_alignbit(t1[j].w[2], t1[j].w[1], S1[j]);
                        px = blend_u8_rgb_window(a0, a1, b0, b1, fx, fy);
                    } else if constexpr (C == 3) {
                        px = blend_u8_rgb_window(t0[j].w[0], t0[j].w[1], t1[j].w[0], t1[j].w[1], fx, fy);
                    } else if constexpr (C == 4) {
                        px = blend_u8_packed<C>(t0[j].w[0], t0[j].w[1], t1[j].w[0], t1[j].w[1], fx, fy);
                    } else {
                        px = blend_u8_packed<C>(t0[j].w[0], t0[j].w[0] >> (8 * C), t1[j].w[0], t1[j].w[0] >> (8 * C), fx, fy);
                    }
                    wtr[64 * j + lane] = px;
                } else {
                    const float* f0 = reinterpret_cast<const float*>(&t0[j]);
                    const float* f1 = reinterpret_cast<const float*>(&t1[j]);
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
            }
        };
        // the other row classes fill the LDS row their own way
        // OUT row: the border value
        auto fill_s = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                if constexpr (sizeof(T) == 1) {
                    wtr[64 * j + lane] = C == 4 ? view.bu : (view.bu & ((1u << (8 * (C & 3))) - 1u));
                } else {
                    float* wf = reinterpret_cast<float*>(wtr) + (64 * j + lane) * C;
#pragma unroll
                    for (int k = 0; k < C; k++) wf[k] = view.bf[k];
                }
            }
        };
        // SLOW row: exact chain and guarded taps for each of the lane's pixels (same ownership, same store order)
        auto slow_s = [&](int y) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                double Xj, Yj, Wj;
                exact_px(y, j, Xj, Yj, Wj);
                int Xe, Ye;
                map_pixel_exact<INTERP>(Xj, Yj, Wj, Xe, Ye);
                const Pixel<T, C> v = sample_global<T, C, INTERP>(view, Xe, Ye);
                if constexpr (sizeof(T) == 1) {
                    wtr[64 * j + lane] = v.packed;
                } else {
                    float* wf = reinterpret_cast<float*>(wtr) + (64 * j + lane) * C;
#pragma unroll
                    for (int k = 0; k < C; k++) wf[k] = v.v[k];
                }
            }
        };
        // EDGE row: unguarded window loads + the fast blend for the pixels whose taps are inside, the border value for
        // those whose taps are all outside, guarded taps for the few in between
        auto edge_s = [&](const uint32_t (&RX)[PPL], const uint32_t (&RY)[PPL], Bytes<WINB> (&t0)[PPL], Bytes<WINB> (&t1)[PPL]) __attribute__((always_inline)) {
            int X[PPL], Y[PPL];  // (t0, t1: the pipeline's tap registers, idle for a row that is not FAST)
            bool inb[PPL];
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                X[j] = (int)(RX[j] ^ 0x80000000u), Y[j] = (int)(RY[j] ^ 0x80000000u);
                const int sx = X[j] >> SH, sy = Y[j] >> SH;
                inb[j] = (uint32_t)sx <= sx_max && (uint32_t)sy <= sy_max;
                const uint32_t off = inb[j] ? (uint32_t)sy * rs32 + (uint32_t)sx * (uint32_t)PBs : 0u;  // (0: any in-bounds address)
                __builtin_memcpy(&t0[j], frame + off, LOADB);
                if (INTERP == kLinear) __builtin_memcpy(&t1[j], frame_r1 + off, LOADB);
            }
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const int sx = X[j] >> SH, sy = Y[j] >> SH;
                const uint32_t fx = (uint32_t)X[j] & 31u, fy = (uint32_t)Y[j] & 31u;
                constexpr int kTap = INTERP == kLinear ? 1 : 0;  // taps reach sx + kTap, sy + kTap
                const bool all_out = fill_ok && (sx < -kTap || sx >= a.src_w || sy < -kTap || sy >= a.src_h);
                Pixel<T, C> v;
                if (inb[j]) {
                    if constexpr (sizeof(T) == 1) {
                        if (INTERP == kNearest)
                            v.packed = C == 4 ? t0[j].w[0] : (t0[j].w[0] & ((1u << (8 * (C & 3))) - 1u));
                        else if constexpr (C == 3)
                            v.packed = blend_u8_rgb_window(t0[j].w[0], t0[j].w[1], t1[j].w[0], t1[j].w[1], fx, fy);
                        else if constexpr (C == 4)
                            v.packed = blend_u8_packed<C>(t0[j].w[0], t0[j].w[1], t1[j].w[0], t1[j].w[1], fx, fy);
                        else
                            v.packed = blend_u8_packed<C>(t0[j].w[0], t0[j].w[0] >> (8 * C), t1[j].w[0], t1[j].w[0] >> (8 * C), fx, fy);
                    } else {
                        const float* f0 = reinterpret_cast<const float*>(&t0[j]);
                        const float* f1 = reinterpret_cast<const float*>(&t1[j]);
                        float w00 = 0, w01 = 0, w10 = 0, w11 = 0;
                        if (INTERP == kLinear) weights_f32((int)fx, (int)fy, w00, w01, w10, w11);
#pragma unroll
                        for (int k = 0; k < C; k++) v.v[k] = INTERP == kNearest ? f0[k] : blend_f32(f0[k], f0[k + C], f1[k], f1[k + C], w00, w01, w10, w11);
                    }
                } else if (all_out) {
                    if constexpr (sizeof(T) == 1) {
                        v.packed = C == 4 ? view.bu : (view.bu & ((1u << (8 * (C & 3))) - 1u));
                    } else {
#pragma unroll
                        for (int k = 0; k < C; k++) v.v[k] = view.bf[k];
                    }
                } else {
                    v = sample_global<T, C, INTERP>(view, X[j], Y[j]);
                }
                if constexpr (sizeof(T) == 1) {
                    wtr[64 * j + lane] = v.packed;
                } else {
                    float* wf = reinterpret_cast<float*>(wtr) + (64 * j + lane) * C;
#pragma unroll
                    for (int k = 0; k < C; k++) wf[k] = v.v[k];
                }
            }
        };
        auto finish_any = [&](int cls, int y, const uint32_t (&RX)[PPL], const uint32_t (&RY)[PPL], Bytes<WINB> (&t0)[PPL],
                              Bytes<WINB> (&t1)[PPL], const uint32_t (&S0)[PPL], const uint32_t (&S1)[PPL], uint4 (&out)[NQ]) {
            if (__builtin_expect(cls == kFast, 1)) {
                finish_s(RX, RY, t0, t1, S0, S1);
            } else {
                if (cls == kOut)
                    fill_s();
                else if (cls == kEdge)
                    edge_s(RX, RY, t0, t1);
                else
                    slow_s(y);
            }
            read_back(out);
        };
        // the destination is written once and never read back by this kernel: non-temporal stores keep it from
        // displacing source lines in L2 / MALL (f32: -8 % kernel time)
        auto store_s = [&](int y, const uint4 (&out)[NQ]) {
            uint8_t* drow = dframe + (int64_t)y * a.dst_rs + (int64_t)x0 * C * sizeof(T);  // the wave's row segment
#if defined(BEVWARP_ABLATE) && (BEVWARP_ABLATE & 4)  // diagnostic builds only: keep the values alive, store one row in 64
            if ((y & 63) != 0 && out[0].x != 0x12345678u) return;
#endif
            if constexpr (sizeof(T) == 1) {  // the lane's 4 pixels = 4 C contiguous bytes, one instruction
                const uint32_t p0 = out[0].x, p1 = out[0].y, p2 = out[0].z, p3 = out[0].w;
                if (a.planar) {  // float planes: one 16-byte store per channel (bevwarp_warp_planar)
                    uint8_t* dp = dframe + (int64_t)y * a.dst_rs + (int64_t)(x0 + lane * PPL) * 4;
#pragma unroll
                    for (int k = 0; k < C; k++) {
                        const float sc = a.pscale[k], bi = a.pbias[k];
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        f32x4 o = {(float)((p0 >> (8 * k)) & 0xffu) * sc + bi, (float)((p1 >> (8 * k)) & 0xffu) * sc + bi,
                                   (float)((p2 >> (8 * k)) & 0xffu) * sc + bi, (float)((p3 >> (8 * k)) & 0xffu) * sc + bi};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dp + k * a.dst_ps));
                    }
                    return;
                }
                uint8_t* d = drow + lane * (PPL * C);
                if constexpr (C == 1) {
                    __builtin_nontemporal_store(p0 | (p1 << 8) | (p2 << 16) | (p3 << 24), reinterpret_cast<uint32_t*>(d));
                } else if constexpr (C == 2) {
                    u32x2 o = {p0 | (p1 << 16), p2 | (p3 << 16)};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(d));
                } else if constexpr (C == 3) {
                    u32x3 o = {p0 | (p1 << 24), (p1 >> 8) | (p2 << 16), (p2 >> 16) | (p3 << 8)};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x3*>(d));
                } else {
                    u32x4 o = {p0, p1, p2, p3};
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(d));
                }
            } else {
#pragma unroll
                for (int u = 0; u < NQ; u++) {
                    const int q = u * 64 + lane;
                    u32x4 o = {out[u].x, out[u].y, out[u].z, out[u].w};
                    if (q < kVec) __builtin_nontemporal_store(o, &reinterpret_cast<u32x4*>(drow)[q]);
                }
            }
        };
        // rows of a tile are dealt to its waves round-robin: neighbouring rows share source lines and run at the same time
#if defined(BEVWARP_ROWS_CONTIG)  // experiment: each wave takes a contiguous quarter of the tile's rows
        constexpr int RSTEP = 1;
        const int rpw = a.tile_h / (kWG / 64);
        int yf = y0 + wave * rpw;
        const int y_end = min(yf + rpw - 1, y_last);
#else
        constexpr int RSTEP = GROWS;
        int yf = y0 + wave;
        const int y_end = y_last;
#endif
        if (yf > y_end) return;
        uint32_t RXc[PPL], RYc[PPL], RXn[PPL], RYn[PPL];
        Bytes<WINB> u0[PPL], u1[PPL];
        uint32_t S0[PPL], S1[PPL];
        uint4 out[NQ];
        int cls_c = coords_s(yf, RXc, RYc), cls_n = kSlow;
        if (cls_c == kOut && yf + RSTEP <= y_end) {
            // The wave's first row lies beyond a frame edge: probe its last row.  When that one lies beyond the same
            // edge with W of the same sign, the rows between them map into the convex hull of the two segments and
            // see nothing of the frame either: fill them without computing a coordinate.  (Footprints like the
            // Brno BEV have a third of their rows outside; a wave whose first row is inside never pays for this.)
            const int side0 = out_side;
            const int y_probe = yf + ((y_end - yf) / RSTEP) * RSTEP;
            if (coords_s(y_probe, RXn, RYn) == kOut && out_side == side0) {
                fill_s();
                read_back(out);
                for (int y = yf; y <= y_end; y += RSTEP) store_s(y, out);
                return;
            }
        }
        issue_s(cls_c, RXc, RYc, u0, u1, S0, S1);
        // Order inside an iteration: next row's loads, THEN the finished row's store, then the arithmetic.  vmcnt
        // retires in issue order, so a store issued before a row's loads would have to reach L2 before that row's
        // taps can be used; issued after them it has a whole iteration to complete.
        bool more = yf + RSTEP <= y_end;
        if (more) cls_n = coords_s(yf + RSTEP, RXn, RYn);
        finish_any(cls_c, yf, RXc, RYc, u0, u1, S0, S1, out);
        while (more) {
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                RXc[j] = RXn[j];
                RYc[j] = RYn[j];
            }
            cls_c = cls_n;
            issue_s(cls_c, RXc, RYc, u0, u1, S0, S1);  // row yf + GROWS
            store_s(yf, out);                  // row yf
            yf += RSTEP;
            more = yf + RSTEP <= y_end;
            if (more) cls_n = coords_s(yf + RSTEP, RXn, RYn);  // overlaps with the loads in flight
            finish_any(cls_c, yf, RXc, RYc, u0, u1, S0, S1, out);
        }
        store_s(yf, out);
        return;
    }

    // -- general rows (image border, ragged tiles), software-pipelined the same way
    int y = y0 + wave * GY + lyi;
    if (y > y_last || nvalid_x == 0) return;
    int Xc[PPL], Yc[PPL];
    Bytes<LOADB> t0[PPL], t1[PPL];
    bool fc = coords(y, Xc, Yc);
    if (fc) issue(Xc, Yc, t0, t1);
    for (;;) {
        const int yn = y + GROWS;
        const bool has_next = yn <= y_last;
        int Xn[PPL], Yn[PPL];
        bool fn = false;
        if (has_next) fn = coords(yn, Xn, Yn);
        finish(y, fc, Xc, Yc, t0, t1);
        if (!has_next) break;
        if (fn) issue(Xn, Yn, t0, t1);
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            Xc[j] = Xn[j];
            Yc[j] = Yn[j];
        }
        fc = fn;
        y = yn;
    }
}

// ===================================================================================================
// warp_wave: wave-private LDS tiles filled by LDS-DMA.  Each WAVE (64 lanes = 16 x 4 block of lanes,
// PPL pixels per lane) computes its block's exact fixed-point coordinates, takes the source bounding
// box from its four corner lanes (v_readlane), copies that box from global memory to its own LDS slot
// with global_load_lds_dwordx4 (coalesced 16-byte chunks, no VGPR staging, no widening) and samples
// from LDS.  No workgroup barrier anywhere: the only synchronisation is the wave's own vmcnt.
// Compared with warp_gather the texture-address unit sees a handful of coalesced 1 KiB instructions
// per block instead of 64-address gathers; compared with warp_tiles there is no barrier, no corner
// approximation (the box comes from the pixels' real coordinates) and occupancy is not tied to a
// whole-workgroup tile.  Blocks whose box is clipped by the image, exceeds the slot, or whose layout is
// not 16-byte aligned fall back to the per-lane gather of the same iteration.
// ===================================================================================================
template <typename T, int C, int INTERP>
__global__ __launch_bounds__(kWG) void warp_wave(const WarpArgs a) {
    constexpr int PPL = pixels_per_lane<T>();
    constexpr int TW = kLX * PPL;
    constexpr int PBs = (int)sizeof(T) * C;
    constexpr int TAPB = INTERP == kLinear ? 2 * PBs : PBs;
    constexpr int LOADB = (TAPB + 3) & ~3;
    constexpr bool kFunnel = (PBs % 4) != 0;  // taps are not dword aligned in LDS: read one more dword and shift
    constexpr double kScale = INTERP == kLinear ? 33554432.0 : 1048576.0;
    constexpr double kMagic = 6755399441055744.0 + 524288.0;
    constexpr int SH = INTERP == kLinear ? kInterBits : 0;
    constexpr int TAPS = INTERP == kLinear ? 1 : 0;  // extra tap to the right / below
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t item = (blockIdx.x & 7u) * (uint32_t)a.chunk + (blockIdx.x >> 3);
    if (item >= (uint32_t)a.total_tiles) return;
    const uint32_t frame_idx = fast_div(item, a.tpf_magic, (uint32_t)a.tiles_per_frame);
    const uint32_t t = item - frame_idx * (uint32_t)a.tiles_per_frame;
    const uint32_t ty = fast_div(t, a.tx_magic, (uint32_t)a.tiles_x), tx = t - ty * (uint32_t)a.tiles_x;
    const int x0 = (int)tx * TW, y0 = (int)ty * a.tile_h;
    const int tid = threadIdx.x;
    const uint8_t* __restrict__ frame = a.src + (int64_t)frame_idx * a.src_fs;
    uint8_t* __restrict__ dframe = a.dst + (int64_t)frame_idx * a.dst_fs;
    const double* __restrict__ M = a.minv + (int64_t)frame_idx * a.m_stride;
    double Mr[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Mr[i] = M[i];

    const int lane = tid & 63, wave = tid >> 6;
    const int lxi = lane & (kLX - 1), lyi = lane >> 4;
    const int xg = x0 + lxi * PPL;
    const int tile_bx = (int)(fast_div((uint32_t)x0, a.bw0_magic, (uint32_t)a.bw0) * (uint32_t)a.bw0);
    double mx[PPL], my[PPL], mw[PPL];
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        const double x1 = (double)(xg + j - tile_bx);
        mx[j] = Mr[0] * x1;
        my[j] = Mr[3] * x1;
        mw[j] = Mr[6] * x1;
    }
    const double bxd = (double)tile_bx;
    const double bX = Mr[0] * bxd, bY = Mr[3] * bxd, bW = Mr[6] * bxd;
    const int nvalid_x = max(0, min(PPL, a.dst_w - xg));
    const int y_last = min(y0 + a.tile_h, a.dst_h) - 1;
    const bool full_x = x0 + TW <= a.dst_w;  // uniform: every lane of the tile owns PPL real pixels
    SrcView view;
    view.frame = frame;
    view.rs = a.src_rs;
    view.w = a.src_w;
    view.h = a.src_h;
#pragma unroll
    for (int k = 0; k < 4; k++) view.bf[k] = a.bval_f[k];
    view.bu = (uint32_t)a.bval_u8[0] | ((uint32_t)a.bval_u8[1] << 8) | ((uint32_t)a.bval_u8[2] << 16) | ((uint32_t)a.bval_u8[3] << 24);
    const uint32_t rs32 = (uint32_t)a.src_rs;
    uint8_t* slot = smem + (size_t)wave * a.lds_bytes;
    const int slot_chunks = (a.lds_bytes >> 4) - 1;                  // one chunk of slack for the funnel read
    const int row_chunk_bytes = (int)(((int64_t)a.src_w * PBs) & ~15);  // bytes of a row made of whole chunks

    for (int yb = y0 + wave * kLY; yb <= y_last; yb += kBandRows) {  // wave-uniform
        const bool row_valid = yb + lyi <= y_last;
        const int y = min(yb + lyi, y_last);  // lanes below the tile recompute its last row (never stored)
        // ---- coordinates (same chain as warp_gather)
        const double dy = (double)y;
        const double X0 = (bX + Mr[1] * dy) + Mr[2], Y0 = (bY + Mr[4] * dy) + Mr[5], W0 = (bW + Mr[7] * dy) + Mr[8];
        double W[PPL], r[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) W[j] = W0 + mw[j];
        if constexpr (PPL == 4) {
            const double p01 = W[0] * W[1], p23 = W[2] * W[3];
            const double inv = rcp_newton(p01 * p23) * kScale;
            const double i01 = inv * p23, i23 = inv * p01;
            r[0] = i01 * W[1];
            r[1] = i01 * W[0];
            r[2] = i23 * W[3];
            r[3] = i23 * W[2];
        } else {
            const double inv = rcp_newton(W[0] * W[1]) * kScale;
            r[0] = inv * W[1];
            r[1] = inv * W[0];
        }
        const double wa = W[0], wb = W[PPL - 1];
        const bool w_ok = (wa > 0) == (wb > 0) && fabs(wa) > 1e-60 && fabs(wa) < 1e60 && fabs(wb) > 1e-60 && fabs(wb) < 1e60;
        int X[PPL], Y[PPL];
        uint32_t tie = 0xffffffffu, expo = 0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            const double tx_ = (X0 + mx[j]) * r[j] + kMagic, ty_ = (Y0 + my[j]) * r[j] + kMagic;
            const uint32_t lox = (uint32_t)__double2loint(tx_), loy = (uint32_t)__double2loint(ty_);
            const uint32_t hix = (uint32_t)__double2hiint(tx_), hiy = (uint32_t)__double2hiint(ty_);
            X[j] = (int)(__builtin_amdgcn_alignbit(hix, lox, 20) ^ 0x80000000u);
            Y[j] = (int)(__builtin_amdgcn_alignbit(hiy, loy, 20) ^ 0x80000000u);
            tie = min(tie, min((lox + 2u) & 0xffffcu, (loy + 2u) & 0xffffcu));
            expo |= (hix ^ 0x43300000u) | (hiy ^ 0x43300000u);
        }
        if (tie == 0 || (expo >> 20) != 0 || !w_ok) {  // rare: the exact chain decides
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const double tx_ = (X0 + mx[j]) * r[j] + kMagic, ty_ = (Y0 + my[j]) * r[j] + kMagic;
                const uint32_t lox = (uint32_t)__double2loint(tx_), loy = (uint32_t)__double2loint(ty_);
                const uint32_t hix = (uint32_t)__double2hiint(tx_), hiy = (uint32_t)__double2hiint(ty_);
                const bool far = (uint32_t)(X[j] + (1 << 22)) >= (1u << 23) || (uint32_t)(Y[j] + (1 << 22)) >= (1u << 23);
                if (!w_ok || far || ((lox + 2u) & 0xffffcu) == 0 || ((loy + 2u) & 0xffffcu) == 0 || ((hix ^ 0x43300000u) >> 20) != 0 ||
                    ((hiy ^ 0x43300000u) >> 20) != 0)
                    map_pixel_exact<INTERP>(X0 + mx[j], Y0 + my[j], W[j], X[j], Y[j]);
            }
        }
        // ---- source box of the wave's block from its four corner pixels (+-1 px for rounding inside the block);
        // the per-pixel test below makes correctness independent of this estimate.
        const int cx[4] = {__builtin_amdgcn_readlane(X[0] >> SH, 0), __builtin_amdgcn_readlane(X[PPL - 1] >> SH, 15),
                           __builtin_amdgcn_readlane(X[0] >> SH, 48), __builtin_amdgcn_readlane(X[PPL - 1] >> SH, 63)};
        const int cy[4] = {__builtin_amdgcn_readlane(Y[0] >> SH, 0), __builtin_amdgcn_readlane(Y[PPL - 1] >> SH, 15),
                           __builtin_amdgcn_readlane(Y[0] >> SH, 48), __builtin_amdgcn_readlane(Y[PPL - 1] >> SH, 63)};
        const int rx0 = min(min(cx[0], cx[1]), min(cx[2], cx[3])) - 1, rx1 = max(max(cx[0], cx[1]), max(cx[2], cx[3])) + 1 + TAPS;
        const int ry0 = min(min(cy[0], cy[1]), min(cy[2], cy[3])) - 1, ry1 = max(max(cy[0], cy[1]), max(cy[2], cy[3])) + 1 + TAPS;
        const int xb0 = (rx0 * PBs) & ~15, xb1 = ((rx1 + 1) * PBs + 15) & ~15;
        const int cpr = (xb1 - xb0) >> 4, rows = ry1 - ry0 + 1;
        const int total = rows * cpr;
        const bool staged = a.src_vec_ok && full_x && rx0 >= 0 && ry0 >= 0 && ry1 < a.src_h && xb1 <= row_chunk_bytes && rx1 - rx0 < 4096 &&
                            rows < 1024 && total <= slot_chunks;
        Pixel<T, C> v[PPL];
        if (staged) {
            // ---- LDS-DMA: chunk c = (row r, column q) -> slot + 16 c; 64 chunks (1 KiB) per wave-instruction
            const uint32_t mdiv = 0xffffffffu / (uint32_t)cpr + 1u;  // c / cpr == umulhi(c, mdiv) (c, cpr < 2^16)
            const uint32_t gbase = (uint32_t)ry0 * rs32 + (uint32_t)xb0;
            for (int c0 = 0; c0 < total; c0 += 64) {
                const uint32_t c = (uint32_t)(c0 + lane);
                if (c < (uint32_t)total) {
                    const uint32_t rr = cpr == 1 ? c : __umulhi(c, mdiv), q = c - rr * (uint32_t)cpr;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(frame + (gbase + rr * rs32 + q * 16u)),
                                                     (__attribute__((address_space(3))) void*)(slot + (size_t)c0 * 16), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA has landed; only this wave reads the slot
            const uint32_t pitch = (uint32_t)cpr * 16u;
            const uint32_t lim_x = (uint32_t)(rx1 - rx0 - TAPS), lim_y = (uint32_t)(ry1 - ry0 - TAPS);
            uint32_t slow = 0;
            Bytes<LOADB> t0[PPL], t1[PPL];
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const int sx = X[j] >> SH, sy = Y[j] >> SH;
                if ((uint32_t)(sx - rx0) <= lim_x && (uint32_t)(sy - ry0) <= lim_y) {
                    const uint32_t A = (uint32_t)(sy - ry0) * pitch + (uint32_t)(sx * PBs - xb0);
                    if constexpr (kFunnel) {
                        const uint32_t* l0 = reinterpret_cast<const uint32_t*>(slot + (A & ~3u));
                        const uint32_t* l1 = reinterpret_cast<const uint32_t*>(slot + (A & ~3u) + pitch);
                        const uint32_t sh8 = (A & 3u) * 8u;
                        uint32_t w0[LOADB / 4 + 1], w1[LOADB / 4 + 1];
#pragma unroll
                        for (int k = 0; k <= LOADB / 4; k++) {
                            w0[k] = l0[k];
                            if (INTERP == kLinear) w1[k] = l1[k];
                        }
#pragma unroll
                        for (int k = 0; k < LOADB / 4; k++) {
                            t0[j].w[k] = __builtin_amdgcn_alignbit(w0[k + 1], w0[k], sh8);
                            if (INTERP == kLinear) t1[j].w[k] = __builtin_amdgcn_alignbit(w1[k + 1], w1[k], sh8);
                        }
                    } else {
                        const uint32_t* l0 = reinterpret_cast<const uint32_t*>(slot + A);
                        const uint32_t* l1 = reinterpret_cast<const uint32_t*>(slot + A + pitch);
#pragma unroll
                        for (int k = 0; k < LOADB / 4; k++) {
                            t0[j].w[k] = l0[k];
                            if (INTERP == kLinear) t1[j].w[k] = l1[k];
                        }
                    }
                } else {
                    slow |= 1u << j;
                }
            }
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const uint32_t fx = (uint32_t)X[j] & 31u, fy = (uint32_t)Y[j] & 31u;
                if constexpr (sizeof(T) == 1) {
                    if (INTERP == kNearest) {
                        v[j].packed = C == 4 ? t0[j].w[0] : (t0[j].w[0] & ((1u << (8 * (C & 3))) - 1u));
                    } else {
                        uint32_t l0, r0, l1, r1;
                        if constexpr (C == 4) {
                            l0 = t0[j].w[0], r0 = t0[j].w[1], l1 = t1[j].w[0], r1 = t1[j].w[1];
                        } else if constexpr (C == 3) {
                            l0 = t0[j].w[0], r0 = __builtin_amdgcn_alignbyte(t0[j].w[1], t0[j].w[0], 3);
                            l1 = t1[j].w[0], r1 = __builtin_amdgcn_alignbyte(t1[j].w[1], t1[j].w[0], 3);
                        } else {
                            l0 = t0[j].w[0], r0 = t0[j].w[0] >> (8 * C), l1 = t1[j].w[0], r1 = t1[j].w[0] >> (8 * C);
                        }
                        v[j].packed = blend_u8_packed<C>(l0, r0, l1, r1, fx, fy);
                    }
                } else {
                    const float* f0 = reinterpret_cast<const float*>(&t0[j]);
                    const float* f1 = reinterpret_cast<const float*>(&t1[j]);
                    if (INTERP == kNearest) {
#pragma unroll
                        for (int k = 0; k < C; k++) v[j].v[k] = f0[k];
                    } else {
                        float w00, w01, w10, w11;
                        weights_f32((int)fx, (int)fy, w00, w01, w10, w11);
#pragma unroll
                        for (int k = 0; k < C; k++) v[j].v[k] = blend_f32(f0[k], f0[k + C], f1[k], f1[k + C], w00, w01, w10, w11);
                    }
                }
            }
            if (slow) {  // never expected (the box bounds the block); kept so that correctness does not rest on it
#pragma unroll
                for (int j = 0; j < PPL; j++)
                    if (slow & (1u << j)) v[j] = sample_global<T, C, INTERP>(view, X[j], Y[j]);
            }
        } else {
            // ---- image border, ragged tiles, oversized or unaligned boxes: per-lane global sampling
#pragma unroll
            for (int j = 0; j < PPL; j++) v[j] = sample_global<T, C, INTERP>(view, X[j], Y[j]);
        }
        if (row_valid && nvalid_x > 0) store_pixels<T, C, PPL>(a, dframe + (int64_t)y * a.dst_rs, xg, nvalid_x, v);
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

// experiments only: BEVWARP_GATHER_PAD_LDS=<bytes> of unused dynamic LDS per workgroup lowers warp_gather's occupancy
inline size_t gather_pad_lds() {
    const char* e = getenv("BEVWARP_GATHER_PAD_LDS");
    return e ? (size_t)atoi(e) : 0;
}

template <typename T, int C>
hipError_t launch_tc(const WarpArgs& a, int interp, dim3 grid, size_t lds, hipStream_t stream) {
    if (a.gather == 2) {
        const size_t wl = (size_t)a.lds_bytes * (kWG / 64);
        if (interp == kNearest)
            hipLaunchKernelGGL((warp_wave<T, C, kNearest>), grid, dim3(kWG), wl, stream, a);
        else
            hipLaunchKernelGGL((warp_wave<T, C, kLinear>), grid, dim3(kWG), wl, stream, a);
        return hipGetLastError();
    }
    if (a.gather) {
        if (interp == kNearest)
            hipLaunchKernelGGL((warp_gather<T, C, kNearest>), grid, dim3(kWG), gather_pad_lds(), stream, a);
        else
            if (sizeof(T) == 1 && C == 3 && a.src_rs % 4 == 0)
                hipLaunchKernelGGL((warp_gather<T, C, kLinear, sizeof(T) == 1 && C == 3>), grid, dim3(kWG), gather_pad_lds(), stream, a);
            else
                hipLaunchKernelGGL((warp_gather<T, C, kLinear>), grid, dim3(kWG), gather_pad_lds(), stream, a);
        return hipGetLastError();
    }
    if (interp == kNearest)
        hipLaunchKernelGGL((warp_tiles<T, C, kNearest>), grid, dim3(kWG), lds, stream, a);
    else
        hipLaunchKernelGGL((warp_tiles<T, C, kLinear>), grid, dim3(kWG), lds, stream, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_t(const WarpArgs& a, int channels, int interp, dim3 grid, size_t lds, hipStream_t stream) {
    switch (channels) {
        case 1: return launch_tc<T, 1>(a, interp, grid, lds, stream);
        case 2: return launch_tc<T, 2>(a, interp, grid, lds, stream);
        case 3: return launch_tc<T, 3>(a, interp, grid, lds, stream);
        default: return launch_tc<T, 4>(a, interp, grid, lds, stream);
    }
}

}  // namespace

int tile_width(int dtype, int kernel) {
    const int ppl = dtype == 0 ? pixels_per_lane<uint8_t>() : pixels_per_lane<float>();
    return (kernel == 1 ? kGatherLX : kLX) * ppl;
}
int band_rows(int kernel) { return kernel == 1 ? (64 / kGatherLX) * (kWG / 64) : kBandRows; }
int pixels_per_lane_of(int dtype) { return dtype == 0 ? pixels_per_lane<uint8_t>() : pixels_per_lane<float>(); }

hipError_t launch_warp(const WarpArgs& a, int dtype, int channels, int interp, hipStream_t stream) {
    const dim3 grid((unsigned)(8 * a.chunk));
    const size_t lds = (size_t)a.lds_bytes + kRowTabBytes;
    return dtype == 0 ? launch_t<uint8_t>(a, channels, interp, grid, lds, stream) : launch_t<float>(a, channels, interp, grid, lds, stream);
}

hipError_t launch_footprint(unsigned char* touched, int batch, int src_h, int src_w, int dst_h, int dst_w, const double* minv,
                            int m_stride, int bw0, int interp, hipStream_t stream) {
    const dim3 block(256), grid((dst_w + 255) / 256, dst_h, batch);
    if (interp == kNearest)
        hipLaunchKernelGGL(footprint_kernel<kNearest>, grid, block, 0, stream, touched, batch, src_h, src_w, dst_h, dst_w, minv, m_stride, bw0);
    else
        hipLaunchKernelGGL(footprint_kernel<kLinear>, grid, block, 0, stream, touched, batch, src_h, src_w, dst_h, dst_w, minv, m_stride, bw0);
    return hipGetLastError();
}

}  // namespace bevwarp
