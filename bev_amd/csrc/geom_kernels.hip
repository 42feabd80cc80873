// geom_kernels.hip -- point projection and rotated-box IoU for MI355X (gfx950).
//
// project_points: the device twin of pts_world_bev (reference bev/rbox.py:136-151): one pass over
// memory instead of numpy's concat / dot / transpose / divide temporaries.  HBM-bound (32 B per
// f64 2-D point); each lane handles two points = one 16-byte load and one 16-byte store (f32),
// or one point = 16 B (f64).
// rbox_iou: N x M IoU of rotated rectangles (reference bev/tracker/rbox_tracker.py:87-92 -> d3d),
// one lane per pair, convex clipping fully in registers.  Latency/ALU bound (output ~1 MB).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "warp_kernels.h"

namespace bevwarp {
namespace {

struct H9 {
    double h[9];
};

template <typename T, int DIM>
__global__ __launch_bounds__(256) void project_points_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n, const H9 H) {
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

// Sutherland-Hodgman: clip polygon (<= 8 vertices) against the 4 half-planes of quad B.
__device__ __forceinline__ double intersection_area(const Quad& A, const Quad& B) {
    double px[8], py[8], qx[8], qy[8];
    int n = 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        px[i] = A.x[i];
        py[i] = A.y[i];
    }
    for (int e = 0; e < 4 && n > 0; e++) {
        const int e1 = (e + 1) & 3;
        const double ex = B.x[e1] - B.x[e], ey = B.y[e1] - B.y[e];
        int m = 0;
        for (int i = 0; i < n; i++) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double di = ex * (py[i] - B.y[e]) - ey * (px[i] - B.x[e]);
            const double dj = ex * (py[j] - B.y[e]) - ey * (px[j] - B.x[e]);
            if (di >= 0) {
                qx[m] = px[i];
                qy[m] = py[i];
                m++;
            }
            if ((di >= 0) != (dj >= 0)) {
                const double t = di / (di - dj);
                qx[m] = px[i] + t * (px[j] - px[i]);
                qy[m] = py[i] + t * (py[j] - py[i]);
                m++;
            }
        }
        n = m;
        for (int i = 0; i < n; i++) {
            px[i] = qx[i];
            py[i] = qy[i];
        }
    }
    if (n < 3) return 0.0;
    double a = 0.0;
    for (int i = 0; i < n; i++) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a += px[i] * py[j] - px[j] * py[i];
    }
    return fabs(0.5 * a);
}

template <typename T>
__global__ __launch_bounds__(256) void rbox_iou_kernel(const T* __restrict__ a, int na, int sa, const T* __restrict__ b, int nb, int sb,
                                                       T* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;  // column (box of b) -> coalesced stores
    const int i = blockIdx.y;
    if (j >= nb) return;
    const T* pa = a + (int64_t)i * sa;
    const T* pb = b + (int64_t)j * sb;
    const Quad A = corners_of((double)pa[0], (double)pa[1], (double)pa[2], (double)pa[3], (double)pa[4]);
    const Quad B = corners_of((double)pb[0], (double)pb[1], (double)pb[2], (double)pb[3], (double)pb[4]);
    const double inter = intersection_area(A, B);
    const double uni = fabs((double)pa[2] * (double)pa[3]) + fabs((double)pb[2] * (double)pb[3]) - inter;
    out[(int64_t)i * nb + j] = (T)(uni > 0 ? inter / uni : 0.0);
}

}  // namespace

hipError_t launch_project_points(const void* in, void* out, int64_t n, int dim, const double* H, int dtype, hipStream_t stream) {
    if (n == 0) return hipSuccess;
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
    const dim3 block(256), grid((nb + 255) / 256, na);
    if (dtype == 2)
        hipLaunchKernelGGL(rbox_iou_kernel<double>, grid, block, 0, stream, (const double*)a, na, a_stride, (const double*)b, nb, b_stride, (double*)out);
    else
        hipLaunchKernelGGL(rbox_iou_kernel<float>, grid, block, 0, stream, (const float*)a, na, a_stride, (const float*)b, nb, b_stride, (float*)out);
    return hipGetLastError();
}

}  // namespace bevwarp
