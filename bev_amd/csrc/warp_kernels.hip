// warp_kernels.hip -- batched BEV homography warp for MI355X (gfx950, wave64).  See DESIGN.md.
//
// Replaces the per-frame cv2.warpPerspective call of the reference (vis_homo.py:89,91;
// bev/tool/compo.py:38,46,47).  One workgroup (256 threads = 4 waves) produces one 64 x TH tile of
// one BEV frame:
//   1. four lanes map the tile's corner pixels; their source bounding box (+ tap margin) is the
//      tile's source region,
//   2. the region is staged into LDS with coalesced row loads (u8x3 is widened to 4 B / pixel so a
//      tap is one aligned dword; other formats keep their natural layout, 16 B per load),
//   3. every lane owns 4 consecutive BEV pixels of one row: it evaluates the inverse homography in
//      float64 with the operation order of the reference algorithm (bit-exact coordinates), samples
//      from LDS and writes 12 / 48 contiguous bytes.
// Pixels whose taps leave the staged region (image border, degenerate tiles) take a per-lane path
// that reads global memory with per-tap bounds checks; tiles whose region does not fit the LDS
// budget, or whose layout is not load-aligned, use that path for every pixel.
//
// No MFMA: this is a gather, bounded by HBM bandwidth and by the float64 coordinate chain.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "warp_kernels.h"

#pragma clang fp contract(off)  // every multiply and add of the coordinate chain rounds separately

namespace bevwarp {
namespace {

constexpr int kWG = 256;
constexpr int kLX = 16;                  // lanes along x inside a wave
constexpr int kLY = 4;                   // lanes along y inside a wave
constexpr int kPPL = 4;                  // consecutive pixels per lane
constexpr int kTileW = kLX * kPPL;       // 64 = block width of the reference algorithm for h >= 16
constexpr int kRowsPerPass = kLY * (kWG / 64);  // 16 rows per workgroup pass
constexpr int kInterBits = 5;

// LDS bytes per pixel: u8x3 is widened to 4, everything else is stored as is.
template <typename T, int C>
constexpr int lds_pixel_bytes() { return (sizeof(T) == 1 && C == 3) ? 4 : (int)sizeof(T) * C; }
// Formats with a staged fast path: every tap must be a whole number of aligned dwords.
template <typename T, int C>
constexpr bool has_staged_path() { return (sizeof(T) == 1 && (C == 3 || C == 4)) || sizeof(T) == 4; }

struct U3 {
    uint32_t x, y, z;
};

// ---------------------------------------------------------------------------------------------------
// Coordinate chain (float64, no contraction).  M = inverse matrix, bx = left edge of the 64-wide
// evaluation block the pixel belongs to, x1 = x - bx.
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
__device__ __forceinline__ void map_pixel(double X0, double Y0, double W0, double mx, double my, double mw, int& X, int& Y,
                                          double* w_out = nullptr) {
    double W = W0 + mw;
    if (w_out) *w_out = W;
    W = (W != 0.0) ? ((INTERP == kLinear ? 32.0 : 1.0) / W) : 0.0;
    X = round_sat((X0 + mx) * W);
    Y = round_sat((Y0 + my) * W);
}

// ---------------------------------------------------------------------------------------------------
// Blending.  u8: 15-bit fixed point of the reference == exact integer form below
//   (sum_i p_i * w_i * 32 + 2^14) >> 15  ==  (wy0 * (wx0 p00 + wx1 p01) + wy1 * (wx0 p10 + wx1 p11) + 512) >> 10
// f32: float weights (1-fy)(1-fx).. (exact multiples of 1/1024), 4 products summed left to right.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t blend_u8(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, uint32_t fx, uint32_t fy) {
    const uint32_t wx1 = fx, wx0 = 32u - fx, wy1 = fy, wy0 = 32u - fy;
    const uint32_t h0 = p00 * wx0 + p01 * wx1;
    const uint32_t h1 = p10 * wx0 + p11 * wx1;
    return (h0 * wy0 + h1 * wy1 + 512u) >> 10;
}

__device__ __forceinline__ float blend_f32(float p00, float p01, float p10, float p11, int fx, int fy) {
    const float s = 1.0f / 32.0f;
    const float tx1 = (float)fx * s, ty1 = (float)fy * s;
    const float tx0 = 1.0f - tx1, ty0 = 1.0f - ty1;
    return ((p00 * (ty0 * tx0) + p01 * (ty0 * tx1)) + p10 * (ty1 * tx0)) + p11 * (ty1 * tx1);
}

template <typename T>
__device__ __forceinline__ T border_of(const WarpArgs& a, int k);
template <>
__device__ __forceinline__ uint8_t border_of<uint8_t>(const WarpArgs& a, int k) { return a.bval_u8[k]; }
template <>
__device__ __forceinline__ float border_of<float>(const WarpArgs& a, int k) { return a.bval_f[k]; }

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

// One pixel straight from global memory with per-tap bounds checks (border, fallback tiles).
template <typename T, int C, int INTERP>
__device__ Pixel<T, C> sample_global(const WarpArgs& a, const uint8_t* __restrict__ frame, int X, int Y) {
    Pixel<T, C> out;
    if constexpr (sizeof(T) == 1) out.packed = 0;
    if (INTERP == kNearest) {
        const bool in = (unsigned)X < (unsigned)a.src_w && (unsigned)Y < (unsigned)a.src_h;
        const T* p = reinterpret_cast<const T*>(frame + (int64_t)Y * a.src_rs) + (int64_t)X * C;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const T v = in ? p[k] : border_of<T>(a, k);
            if constexpr (sizeof(T) == 1)
                out.packed |= (uint32_t)v << (8 * k);
            else
                out.v[k] = v;
        }
        return out;
    }
    const int sx = X >> kInterBits, sy = Y >> kInterBits, fx = X & 31, fy = Y & 31;
    const bool xin0 = (unsigned)sx < (unsigned)a.src_w, xin1 = (unsigned)(sx + 1) < (unsigned)a.src_w;
    const bool yin0 = (unsigned)sy < (unsigned)a.src_h, yin1 = (unsigned)(sy + 1) < (unsigned)a.src_h;
    const T* r0 = reinterpret_cast<const T*>(frame + (int64_t)sy * a.src_rs) + (int64_t)sx * C;
    const T* r1 = reinterpret_cast<const T*>(frame + (int64_t)(sy + 1) * a.src_rs) + (int64_t)sx * C;
#pragma unroll
    for (int k = 0; k < C; k++) {
        const T b = border_of<T>(a, k);
        const T v00 = (xin0 && yin0) ? r0[k] : b;
        const T v01 = (xin1 && yin0) ? r0[k + C] : b;
        const T v10 = (xin0 && yin1) ? r1[k] : b;
        const T v11 = (xin1 && yin1) ? r1[k + C] : b;
        if constexpr (sizeof(T) == 1)
            out.packed |= blend_u8(v00, v01, v10, v11, fx, fy) << (8 * k);
        else
            out.v[k] = blend_f32(v00, v01, v10, v11, fx, fy);
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
        const uint32_t p00 = l[px], p01 = l[px + 1], p10 = l[px + pitch_px], p11 = l[px + pitch_px + 1];
        out.packed = 0;
#pragma unroll
        for (int k = 0; k < C; k++)
            out.packed |= blend_u8((p00 >> (8 * k)) & 0xffu, (p01 >> (8 * k)) & 0xffu, (p10 >> (8 * k)) & 0xffu,
                                   (p11 >> (8 * k)) & 0xffu, fx, fy) << (8 * k);
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
#pragma unroll
        for (int k = 0; k < C; k++) out.v[k] = blend_f32(t0[k], t0[k + C], t1[k], t1[k + C], fx, fy);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------
// Region staging.  rows x npx pixels starting at (ax0, ry0); ax0 and npx are multiples of 4.
// ---------------------------------------------------------------------------------------------------
template <typename T, int C>
__device__ __forceinline__ void stage_region(const WarpArgs& a, const uint8_t* __restrict__ frame, uint8_t* __restrict__ lds, int ax0,
                                             int ry0, int rows, int npx, int tid) {
    if constexpr (sizeof(T) == 1 && C == 3) {
        // 4 pixels = 12 source bytes (3 dwords, 4-byte aligned) -> 4 LDS dwords B|G|R|0
        const int gpr = npx >> 2;  // groups per row
        const int total = rows * gpr;
        const int step_r = kWG / gpr, step_q = kWG % gpr;
        int r = tid / gpr, q = tid % gpr;
        const uint8_t* base = frame + (int64_t)ry0 * a.src_rs + (int64_t)ax0 * 3;
        uint4* l = reinterpret_cast<uint4*>(lds);
        for (int g = tid; g < total; g += kWG) {
            const U3 v = *reinterpret_cast<const U3*>(base + (int64_t)r * a.src_rs + q * 12);
            uint4 o;
            o.x = v.x & 0x00ffffffu;
            o.y = __builtin_amdgcn_alignbyte(v.y, v.x, 3) & 0x00ffffffu;
            o.z = __builtin_amdgcn_alignbyte(v.z, v.y, 2) & 0x00ffffffu;
            o.w = v.z >> 8;
            l[r * gpr + q] = o;
            r += step_r;
            q += step_q;
            if (q >= gpr) {
                q -= gpr;
                r++;
            }
        }
    } else {
        constexpr int PB = lds_pixel_bytes<T, C>();
        const int cpr = (npx * PB) >> 4;  // 16-byte chunks per row
        const int total = rows * cpr;
        const int step_r = kWG / cpr, step_q = kWG % cpr;
        int r = tid / cpr, q = tid % cpr;
        const uint8_t* base = frame + (int64_t)ry0 * a.src_rs + (int64_t)ax0 * PB;
        uint4* l = reinterpret_cast<uint4*>(lds);
        for (int g = tid; g < total; g += kWG) {
            l[r * cpr + q] = *reinterpret_cast<const uint4*>(base + (int64_t)r * a.src_rs + q * 16);
            r += step_r;
            q += step_q;
            if (q >= cpr) {
                q -= cpr;
                r++;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Output of one lane: kPPL pixels, contiguous in the row (12 B for u8x3, 48 B for f32x3).
// ---------------------------------------------------------------------------------------------------
template <typename T, int C>
__device__ __forceinline__ void store_pixels(const WarpArgs& a, uint8_t* __restrict__ drow, int x, int nvalid, const Pixel<T, C>* v) {
    T* d = reinterpret_cast<T*>(drow) + (int64_t)x * C;
    if (nvalid == kPPL && a.dst_vec_ok) {
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
        if constexpr (sizeof(T) == 4) {  // 4 pixels x C floats = C chunks of 16 B
            float f[kPPL * C];
#pragma unroll
            for (int j = 0; j < kPPL; j++)
#pragma unroll
                for (int k = 0; k < C; k++) f[j * C + k] = v[j].v[k];
#pragma unroll
            for (int k = 0; k < C; k++) reinterpret_cast<float4*>(d)[k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < kPPL; j++) {
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

template <typename T, int C, int INTERP>
__global__ __launch_bounds__(kWG) void warp_tiles(const WarpArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_corner[4][3];  // X, Y (fixed point) and sign of W per corner

    // ---- which tile: XCD-aware order.  Workgroups are dealt round-robin over the 8 XCDs, so ids
    // b and b + 8 share an L2; give each XCD one contiguous run of (frame, tile) items in raster order.
    const int64_t item = (int64_t)(blockIdx.x & 7) * a.chunk + (blockIdx.x >> 3);
    if (item >= a.total_tiles) return;
    const int frame_idx = (int)(item / a.tiles_per_frame);
    const int t = (int)(item - (int64_t)frame_idx * a.tiles_per_frame);
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int x0 = tx * kTileW, y0 = ty * a.tile_h;
    const int tid = threadIdx.x;

    const uint8_t* __restrict__ frame = a.src + (int64_t)frame_idx * a.src_fs;
    uint8_t* __restrict__ dframe = a.dst + (int64_t)frame_idx * a.dst_fs;
    const double* __restrict__ M = a.minv + (int64_t)frame_idx * a.m_stride;
    double Mr[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Mr[i] = M[i];

    const int x_last = min(x0 + kTileW, a.dst_w) - 1, y_last = min(y0 + a.tile_h, a.dst_h) - 1;

    // ---- source region of the tile from its four corner pixels
    constexpr bool kStagedFmt = has_staged_path<T, C>();
    bool staged = false;
    int ax0 = 0, ry0 = 0, rows = 0, npx = 0, rx0 = 0, rx1 = -1, ry1 = -1;
    if (kStagedFmt && a.src_vec_ok) {
        if (tid < 4) {
            const int cx = (tid & 1) ? x_last : x0, cy = (tid & 2) ? y_last : y0;
            const int bx = (cx / a.bw0) * a.bw0;
            double X0, Y0, W0, W;
            int X, Y;
            row_terms(Mr, bx, cy, X0, Y0, W0);
            const double x1 = (double)(cx - bx);
            map_pixel<INTERP>(X0, Y0, W0, Mr[0] * x1, Mr[3] * x1, Mr[6] * x1, X, Y, &W);
            s_corner[tid][0] = INTERP == kLinear ? (X >> kInterBits) : X;
            s_corner[tid][1] = INTERP == kLinear ? (Y >> kInterBits) : Y;
            // |W| must stay clear of 0 on the whole tile for the corner box to bound it
            s_corner[tid][2] = (W > 1e-300) ? 1 : ((W < -1e-300) ? -1 : 0);
        }
        __syncthreads();
        const int sg = s_corner[0][2] + s_corner[1][2] + s_corner[2][2] + s_corner[3][2];
        const int mnx = min(min(s_corner[0][0], s_corner[1][0]), min(s_corner[2][0], s_corner[3][0]));
        const int mxx = max(max(s_corner[0][0], s_corner[1][0]), max(s_corner[2][0], s_corner[3][0]));
        const int mny = min(min(s_corner[0][1], s_corner[1][1]), min(s_corner[2][1], s_corner[3][1]));
        const int mxy = max(max(s_corner[0][1], s_corner[1][1]), max(s_corner[2][1], s_corner[3][1]));
        if (sg == 4 || sg == -4) {
            // +-1 px for rounding inside the tile, +1 for the right / lower tap; clipped to the image.
            // (64-bit so that saturated coordinates cannot wrap.)
            const int64_t lx = (int64_t)mnx - 1, hx = (int64_t)mxx + 2, ly = (int64_t)mny - 1, hy = (int64_t)mxy + 2;
            rx0 = (int)max<int64_t>(lx, 0);
            rx1 = (int)min<int64_t>(hx, a.src_w - 1);
            ry0 = (int)max<int64_t>(ly, 0);
            ry1 = (int)min<int64_t>(hy, a.src_h - 1);
            if (rx0 <= rx1 && ry0 <= ry1) {
                ax0 = rx0 & ~3;
                npx = (rx1 | 3) - ax0 + 1;
                rows = ry1 - ry0 + 1;
                staged = (int64_t)rows * npx * lds_pixel_bytes<T, C>() <= (int64_t)a.lds_bytes;
            }
        }
    }
    if (staged) {
        if constexpr (kStagedFmt) stage_region<T, C>(a, frame, smem, ax0, ry0, rows, npx, tid);
        __syncthreads();
    }

    // ---- per-lane constants
    const int lane = tid & 63, wave = tid >> 6;
    const int lxi = lane & (kLX - 1), lyi = lane >> 4;
    const int xg = x0 + lxi * kPPL;  // first pixel of this lane's group
    int bxj[kPPL];
    double mx[kPPL], my[kPPL], mw[kPPL];
#pragma unroll
    for (int j = 0; j < kPPL; j++) {
        const int x = xg + j;
        bxj[j] = (x / a.bw0) * a.bw0;
        const double x1 = (double)(x - bxj[j]);
        mx[j] = Mr[0] * x1;
        my[j] = Mr[3] * x1;
        mw[j] = Mr[6] * x1;
    }
    const bool one_block = bxj[0] == bxj[kPPL - 1];
    const int nvalid_x = max(0, min(kPPL, a.dst_w - xg));
    const uint32_t fast_w = (uint32_t)max(rx1 - rx0 + (INTERP == kLinear ? 0 : 1), 0);  // sx - rx0 < fast_w  <=> sx(+1) in region
    const uint32_t fast_h = (uint32_t)max(ry1 - ry0 + (INTERP == kLinear ? 0 : 1), 0);

    for (int pass = 0; pass * kRowsPerPass < a.tile_h; pass++) {
        const int y = y0 + pass * kRowsPerPass + wave * kLY + lyi;
        if (y > y_last || nvalid_x == 0) continue;
        double X0, Y0, W0;
        row_terms(Mr, bxj[0], y, X0, Y0, W0);
        Pixel<T, C> v[kPPL];
#pragma unroll
        for (int j = 0; j < kPPL; j++) {
            if (!one_block && j > 0) row_terms(Mr, bxj[j], y, X0, Y0, W0);
            int X, Y;
            map_pixel<INTERP>(X0, Y0, W0, mx[j], my[j], mw[j], X, Y);
            const int sx = INTERP == kLinear ? (X >> kInterBits) : X, sy = INTERP == kLinear ? (Y >> kInterBits) : Y;
            const uint32_t ox = (uint32_t)(sx - rx0), oy = (uint32_t)(sy - ry0);
            bool fast = false;
            if constexpr (kStagedFmt) fast = staged && ox < fast_w && oy < fast_h;
            if (fast) {
                const uint32_t px = (uint32_t)(sy - ry0) * (uint32_t)npx + (uint32_t)(sx - ax0);
                if constexpr (kStagedFmt) v[j] = sample_lds<T, C, INTERP>(smem, px, (uint32_t)npx, X & 31, Y & 31);
            } else {
                v[j] = sample_global<T, C, INTERP>(a, frame, X, Y);
            }
        }
        store_pixels<T, C>(a, dframe + (int64_t)y * a.dst_rs, xg, nvalid_x, v);
    }
}

// Footprint: mark every in-bounds source pixel any tap would read (measurement aid).
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
    map_pixel<INTERP>(X0, Y0, W0, Mr[0] * x1, Mr[3] * x1, Mr[6] * x1, X, Y);
    const int sx = INTERP == kLinear ? (X >> kInterBits) : X, sy = INTERP == kLinear ? (Y >> kInterBits) : Y;
    unsigned char* tb = touched + (int64_t)b * src_h * src_w;
    const int ntap = INTERP == kLinear ? 2 : 1;
    for (int dy = 0; dy < ntap; dy++)
        for (int dx = 0; dx < ntap; dx++) {
            const int px = sx + dx, py = sy + dy;
            if ((unsigned)px < (unsigned)src_w && (unsigned)py < (unsigned)src_h) tb[(int64_t)py * src_w + px] = 1;
        }
}

template <typename T, int C>
hipError_t launch_tc(const WarpArgs& a, int interp, dim3 grid, size_t lds, hipStream_t stream) {
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

int tile_width() { return kTileW; }
int rows_per_pass() { return kRowsPerPass; }

hipError_t launch_warp(const WarpArgs& a, int dtype, int channels, int interp, hipStream_t stream) {
    const dim3 grid((unsigned)(8 * a.chunk));
    const size_t lds = (size_t)a.lds_bytes;
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
