// warp_kernels.hip -- host side of the warp kernels: launch geometry helpers, dispatch to the per-format translation units
// (warp_u8_linear.hip, warp_u8_nearest.hip, warp_f32_linear.hip, warp_f32_nearest.hip, warp_composite.hip), and the footprint
// kernel (measurement aid).  The kernel itself is warp_rows.h.
#include "coords.h"

namespace bevwarp {

// one per translation unit
void launch_u8_linear(const WarpArgs& a, int channels, dim3 grid, hipStream_t stream);
void launch_u8_nearest(const WarpArgs& a, int channels, dim3 grid, hipStream_t stream);
void launch_f32_linear(const WarpArgs& a, int channels, dim3 grid, hipStream_t stream);
void launch_f32_nearest(const WarpArgs& a, int channels, dim3 grid, hipStream_t stream);

namespace {

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

}  // namespace

#ifdef BEVWARP_CLOCK
hipError_t launch_u8_linear_clock(unsigned long long*, int), launch_u8_nearest_clock(unsigned long long*, int), launch_f32_linear_clock(unsigned long long*, int),
    launch_f32_nearest_clock(unsigned long long*, int), launch_composite_clock(unsigned long long*, int);
hipError_t debug_read_clock(unsigned long long* out4, int reset) {  // the sum over the translation units
    for (int i = 0; i < 16; i++) out4[i] = 0;  // (16 words: warp_rows.h, kClkWords)
    for (auto fn : {launch_u8_linear_clock, launch_u8_nearest_clock, launch_f32_linear_clock, launch_f32_nearest_clock, launch_composite_clock}) {
        const hipError_t e = fn(out4, reset);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
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
    (void)dtype, (void)channels, (void)interp;  // (every format is compiled for the same occupancy)
    return cus * kWavesPerSimd * (4 / kWaves);
}

hipError_t launch_warp(const WarpArgs& a, int dtype, int channels, int interp, hipStream_t stream) {
    (void)hipGetLastError();  // a stale error left by the host framework is not this call's
    const dim3 grid((unsigned)(8 * (a.chunk + a.tail_split)));
    if (dtype == 0)
        (interp == kNearest ? launch_u8_nearest : launch_u8_linear)(a, channels, grid, stream);
    else
        (interp == kNearest ? launch_f32_nearest : launch_f32_linear)(a, channels, grid, stream);
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
