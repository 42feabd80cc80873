// geom_kernels.hip -- point projection and rotated-box IoU for MI355X (gfx950).
//
// project_points: the device twin of pts_world_bev (reference bev/rbox.py:136-151): one pass over
// memory instead of numpy's concat / dot / transpose / divide temporaries.  HBM-bound (32 B per
// f64 2-D point); each lane handles two points = one 16-byte load and one 16-byte store (f32),
// or one point = 16 B (f64).
// rbox_iou: N x M IoU of rotated rectangles (reference bev/tracker/rbox_tracker.py:87-92 -> d3d),
// one lane per pair: circumscribed-circle rejection first, convex clipping for the pairs that may touch.
// Latency/ALU bound (output ~1 MB).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "warp_kernels.h"

namespace bevwarp {
namespace {

struct H9 {
    double h[9];
};

template <typename T, int DIM>
// (`out` may be `in`: every lane reads its own point before it writes it, and neither pointer is __restrict__)
__global__ __launch_bounds__(256) void project_points_kernel(const T* in, T* out, int64_t n, const H9 H) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double x, y, w = 1.0;
        if constexpr (DIM == 2 && sizeof(T) == 8) {
            const double2 p = reinterpret_cast<const double2*>(in)[i];
            x = p.x;
            y = p.y;
        } else if constexpr (DIM == 2) {
            const float2 p = reinterpret_cast<const float2*>(in)[i];
            x = p.x;
            y = p.y;
        } else {
            x = (double)in[i * 3];
            y = (double)in[i * 3 + 1];
            w = (double)in[i * 3 + 2];
        }
        const double X = H.h[0] * x + H.h[1] * y + H.h[2] * w;
        const double Y = H.h[3] * x + H.h[4] * y + H.h[5] * w;
        const double Z = H.h[6] * x + H.h[7] * y + H.h[8] * w;
        const double ox = X / Z, oy = Y / Z;
        if constexpr (DIM == 2 && sizeof(T) == 8) {
            reinterpret_cast<double2*>(out)[i] = make_double2(ox, oy);
        } else if constexpr (DIM == 2) {
            reinterpret_cast<float2*>(out)[i] = make_float2((float)ox, (float)oy);
        } else {
            out[i * 3] = (T)ox;
            out[i * 3 + 1] = (T)oy;
            out[i * 3 + 2] = (T)(Z / Z);
        }
    }
}

// ---- rotated-rectangle IoU ------------------------------------------------------------------------
struct Quad {
    double x[4], y[4];
};

__device__ __forceinline__ Quad corners_of(double cx, double cy, double w, double h, double yaw) {
    // "world" convention of the reference (rbox.py:87-95): length h along +x, width w along y at yaw 0
    const double hx = 0.5 * h, hy = 0.5 * w, c = cos(yaw), s = sin(yaw);
    const double lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};  // counter-clockwise
    Quad q;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        q.x[i] = c * lx[i] - s * ly[i] + cx;
        q.y[i] = s * lx[i] + c * ly[i] + cy;
    }
    return q;
}

// Sutherland-Hodgman: clip polygon (<= 8 vertices) against the 4 half-planes of quad B.  The two vertex lists
// ping-pong in LDS (vertex-major, one column per thread: dynamic indexing without scratch memory); the signed
// distance of a vertex is computed once and carried to the next edge test.  Same expressions, same order as
// oracle/warp_oracle.c.
constexpr int kIouThreads = 256;
__device__ __forceinline__ double intersection_area(const Quad& A, const Quad& B, double* __restrict__ lds, int tid) {
    auto P = [&](int buf, int xy, int k) -> double& { return lds[((buf * 2 + xy) * 8 + k) * kIouThreads + tid]; };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        P(0, 0, i) = A.x[i];
        P(0, 1, i) = A.y[i];
    }
    int cur = 0, n = 4;
    for (int e = 0; e < 4 && n > 0; e++) {
        const int e1 = (e + 1) & 3;
        const double bx = B.x[e], by = B.y[e];
        const double ex = B.x[e1] - bx, ey = B.y[e1] - by;
        int m = 0;
        double xi = P(cur, 0, 0), yi = P(cur, 1, 0);
        double di = ex * (yi - by) - ey * (xi - bx);
        for (int i = 0; i < n; i++) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double xj = P(cur, 0, j), yj = P(cur, 1, j);
            const double dj = ex * (yj - by) - ey * (xj - bx);
            if (di >= 0) {
                P(cur ^ 1, 0, m) = xi;
                P(cur ^ 1, 1, m) = yi;
                m++;
            }
            if ((di >= 0) != (dj >= 0)) {
                const double t = di / (di - dj);
                P(cur ^ 1, 0, m) = xi + t * (xj - xi);
                P(cur ^ 1, 1, m) = yi + t * (yj - yi);
                m++;
            }
            xi = xj, yi = yj, di = dj;
        }
        n = m;
        cur ^= 1;
    }
    if (n < 3) return 0.0;
    double a = 0.0;
    double xi = P(cur, 0, 0), yi = P(cur, 1, 0);
    for (int i = 0; i < n; i++) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        const double xj = P(cur, 0, j), yj = P(cur, 1, j);
        a += xi * yj - xj * yi;
        xi = xj, yi = yj;
    }
    return fabs(0.5 * a);
}

template <typename T>
__global__ __launch_bounds__(kIouThreads) void rbox_iou_kernel(const T* __restrict__ a, int na, int sa, const T* __restrict__ b, int nb, int sb,
                                                       T* __restrict__ out) {
    __shared__ double s_poly[2 * 2 * 8 * kIouThreads];  // 64 KiB: two vertex lists per thread
    const int j = blockIdx.x * blockDim.x + threadIdx.x;  // column (box of b) -> coalesced stores
    const int i = blockIdx.y;
    if (j >= nb) return;
    const T* pa = a + (int64_t)i * sa;
    const T* pb = b + (int64_t)j * sb;
    {   // boxes whose circumscribed circles are apart cannot intersect: the clip below would return exactly 0 for
        // them.  In a tracker step almost every pair ends here, before any sin / cos or clipping (the 1e-4 margin
        // keeps near-touching pairs on the exact path; NaNs fall through to it as well).
        const double dx = (double)pa[0] - (double)pb[0], dy = (double)pa[1] - (double)pb[1];
        const double ra2 = (double)pa[2] * (double)pa[2] + (double)pa[3] * (double)pa[3];
        const double rb2 = (double)pb[2] * (double)pb[2] + (double)pb[3] * (double)pb[3];
        const double rs = 0.5 * (sqrt(ra2) + sqrt(rb2));
        if (dx * dx + dy * dy > rs * rs * 1.0001 && fabs((double)pa[2] * (double)pa[3]) + fabs((double)pb[2] * (double)pb[3]) > 0) {
            out[(int64_t)i * nb + j] = (T)0.0;
            return;
        }
    }
    const Quad A = corners_of((double)pa[0], (double)pa[1], (double)pa[2], (double)pa[3], (double)pa[4]);
    const Quad B = corners_of((double)pb[0], (double)pb[1], (double)pb[2], (double)pb[3], (double)pb[4]);
    const double inter = intersection_area(A, B, s_poly, (int)threadIdx.x);
    const double uni = fabs((double)pa[2] * (double)pa[3]) + fabs((double)pb[2] * (double)pb[3]) - inter;
    out[(int64_t)i * nb + j] = (T)(uni > 0 ? inter / uni : 0.0);
}

// ---- alpha composite (reference bev/tool/compo.py:16-23) ------------------------------------------------
// out = uint8(min(round_half_even(fg * (mask / 255) + bg * (1 - mask / 255)), 255)), float64 like numpy; 16 bytes per lane.
__device__ __forceinline__ uint32_t composite_byte(uint32_t bg, uint32_t fg, uint32_t m) {
    const double a = (double)m / 255.0;
    const double v = rint((double)fg * a + (double)bg * (1.0 - a));
    return (uint32_t)(v > 255.0 ? 255.0 : v);
}
// (`out` may be `bg` or `fg`: a lane reads its 16 bytes of each input before it writes them; no pointer is __restrict__)
__global__ __launch_bounds__(256) void composite_kernel(const uint8_t* bg, const uint8_t* fg, const uint8_t* mask, uint8_t* out, int64_t n, int vec_ok) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= n) return;
    if (vec_ok && i0 + 16 <= n) {
        const uint4 B = *reinterpret_cast<const uint4*>(bg + i0), F = *reinterpret_cast<const uint4*>(fg + i0), M = *reinterpret_cast<const uint4*>(mask + i0);
        const uint32_t b[4] = {B.x, B.y, B.z, B.w}, f[4] = {F.x, F.y, F.z, F.w}, m[4] = {M.x, M.y, M.z, M.w};
        uint32_t o[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            o[w] = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) o[w] |= composite_byte((b[w] >> (8 * k)) & 255u, (f[w] >> (8 * k)) & 255u, (m[w] >> (8 * k)) & 255u) << (8 * k);
        }
        *reinterpret_cast<uint4*>(out + i0) = make_uint4(o[0], o[1], o[2], o[3]);
        return;
    }
    for (int64_t i = i0; i < n && i < i0 + 16; i++) out[i] = (uint8_t)composite_byte(bg[i], fg[i], mask[i]);
}

}  // namespace

hipError_t launch_composite(const uint8_t* bg, const uint8_t* fg, const uint8_t* mask, uint8_t* out, int64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();  // a stale error left by the host framework is not this call's
    const int vec_ok = (((uintptr_t)bg | (uintptr_t)fg | (uintptr_t)mask | (uintptr_t)out) & 15) == 0;
    const int64_t lanes = (n + 15) / 16;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, bg, fg, mask, out, n, vec_ok);
    return hipGetLastError();
}

hipError_t launch_project_points(const void* in, void* out, int64_t n, int dim, const double* H, int dtype, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    H9 h;
    for (int i = 0; i < 9; i++) h.h[i] = H[i];
    const int block = 256;
    const int64_t want = (n + block - 1) / block;
    const int grid = (int)(want < 256 * 8 ? want : 256 * 8);  // grid-stride above 8 blocks per CU
    if (dtype == 2) {
        if (dim == 2)
            hipLaunchKernelGGL((project_points_kernel<double, 2>), dim3(grid), dim3(block), 0, stream, (const double*)in, (double*)out, n, h);
        else
            hipLaunchKernelGGL((project_points_kernel<double, 3>), dim3(grid), dim3(block), 0, stream, (const double*)in, (double*)out, n, h);
    } else {
        if (dim == 2)
            hipLaunchKernelGGL((project_points_kernel<float, 2>), dim3(grid), dim3(block), 0, stream, (const float*)in, (float*)out, n, h);
        else
            hipLaunchKernelGGL((project_points_kernel<float, 3>), dim3(grid), dim3(block), 0, stream, (const float*)in, (float*)out, n, h);
    }
    return hipGetLastError();
}

hipError_t launch_rbox_iou(const void* a, int na, int a_stride, const void* b, int nb, int b_stride, void* out, int dtype,
                           hipStream_t stream) {
    if (na == 0 || nb == 0) return hipSuccess;
    (void)hipGetLastError();
    const dim3 block(256), grid((nb + 255) / 256, na);
    if (dtype == 2)
        hipLaunchKernelGGL(rbox_iou_kernel<double>, grid, block, 0, stream, (const double*)a, na, a_stride, (const double*)b, nb, b_stride, (double*)out);
    else
        hipLaunchKernelGGL(rbox_iou_kernel<float>, grid, block, 0, stream, (const float*)a, na, a_stride, (const float*)b, nb, b_stride, (float*)out);
    return hipGetLastError();
}

}  // namespace bevwarp
