// coords.h -- coordinate arithmetic of the warp kernel (device code, gfx950).  See DESIGN.md section 4.1.
//
// The reference (cv2.warpPerspective at vis_homo.py:89,91; bev/tool/compo.py:38,46,47) rounds fX = (X0 + M0 x1) * (32 / W) half-to-even
// in float64.  Two chains compute it here: the EXACT one, operation for operation (IEEE division, no contraction), and the FAST one
// (one v_rcp_f64 + Newton step, FMAs, fixed point through the float64 mantissa), which proves per pixel that it rounded the same
// way and hands the pixel to the exact chain when it cannot.
#pragma once
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
// every format's kernel is compiled for four waves per SIMD (<= 128 VGPRs): what its interior loop needs without spilling;
// the host sizes launches against it (resident_workgroups)
constexpr int kWavesPerSimd = 4;

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

}  // namespace
}  // namespace bevwarp
