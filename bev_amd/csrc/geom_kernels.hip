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
#include <math.h>
#include <stdint.h>

#include "warp_kernels.h"

namespace bevwarp {
namespace {

struct H9 {
    double h[9];
};

// 1 / z to the last bit or two of float64: v_rcp_f64 (2^-24) and two Newton steps.  The two IEEE divisions of a point
// (~62 issue cycles each) were what bound the float32 case; the results stay within 2 ulp of the quotients (the parity
// bar of this path is rtol 1e-12 against the reference's own vectors).
__device__ __forceinline__ double rcp_full(double z) {
    double r = __builtin_amdgcn_rcp(z);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void project_one(const H9& H, double x, double y, double w, double& ox, double& oy) {
    const double X = H.h[0] * x + H.h[1] * y + H.h[2] * w;
    const double Y = H.h[3] * x + H.h[4] * y + H.h[5] * w;
    const double Z = H.h[6] * x + H.h[7] * y + H.h[8] * w;
    // (Z == 0 or non-finite: the quotients themselves, so that inf / nan come out as numpy's do)
    const bool tame = fabs(Z) > 1e-300 && fabs(Z) < 1e300;
    const double r = rcp_full(Z);
    ox = tame ? X * r : X / Z;
    oy = tame ? Y * r : Y / Z;
}

// non-temporal loads + plain stores: the fastest of the four pairs (A/B: tools/ab_points.py, profiles/r03_points_ab.txt)
template <typename V>
__device__ __forceinline__ V pp_load(const V* p) {
    return __builtin_nontemporal_load(p);
}
template <typename V>
__device__ __forceinline__ void pp_store(const V& v, V* p) {
    *p = v;
}
typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef double pf64x2 __attribute__((ext_vector_type(2)));
// (`out` may be `in`: every lane reads its own points before it writes them, and neither pointer is __restrict__)
// 16 bytes per lane per step (two float32 points / one float64 point), streamed once: non-temporal both ways.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void project_points_kernel(const T* in, T* out, int64_t n, const H9 H, int vec16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (DIM == 2 && sizeof(T) == 4) {
        if (!vec16) {  // buffers that are only 8-byte aligned (a view starting at an odd point): one point per step
            for (int64_t i = t0; i < n; i += stride) {
                const float2 p = reinterpret_cast<const float2*>(in)[i];
                double ax, ay;
                project_one(H, (double)p.x, (double)p.y, 1.0, ax, ay);
                reinterpret_cast<float2*>(out)[i] = make_float2((float)ax, (float)ay);
            }
            return;
        }
        const int64_t pairs = n >> 1;
        for (int64_t i = t0; i < pairs; i += stride) {
            const pf32x4 p = pp_load(reinterpret_cast<const pf32x4*>(in) + i);
            double ax, ay, bx, by;
            project_one(H, (double)p.x, (double)p.y, 1.0, ax, ay);
            project_one(H, (double)p.z, (double)p.w, 1.0, bx, by);
            const pf32x4 o = {(float)ax, (float)ay, (float)bx, (float)by};
            pp_store(o, reinterpret_cast<pf32x4*>(out) + i);
        }
        if ((n & 1) && t0 == 0) {  // the odd last point
            double ax, ay;
            project_one(H, (double)in[2 * (n - 1)], (double)in[2 * (n - 1) + 1], 1.0, ax, ay);
            out[2 * (n - 1)] = (float)ax;
            out[2 * (n - 1) + 1] = (float)ay;
        }
    } else if constexpr (DIM == 2) {
        for (int64_t i = t0; i < n; i += stride) {
            const pf64x2 p = pp_load(reinterpret_cast<const pf64x2*>(in) + i);
            double ox, oy;
            project_one(H, p.x, p.y, 1.0, ox, oy);
            const pf64x2 o = {ox, oy};
            pp_store(o, reinterpret_cast<pf64x2*>(out) + i);
        }
    } else {
        for (int64_t i = t0; i < n; i += stride) {
            const double x = (double)in[i * 3], y = (double)in[i * 3 + 1], w = (double)in[i * 3 + 2];
            double ox, oy;
            project_one(H, x, y, w, ox, oy);
            const double Z = H.h[6] * x + H.h[7] * y + H.h[8] * w;
            out[i * 3] = (T)ox;
            out[i * 3 + 1] = (T)oy;
            out[i * 3 + 2] = (T)(Z / Z);
        }
    }
}

// ---- rotated-rectangle IoU ------------------------------------------------------------------------
// Area of (box A) n (box B) without clipping a vertex list.  In B's own frame B is the axis-aligned box |x| <= hx, |y| <= hy, and the map
// C(x, y) = (clamp(x, -hx, hx), clamp(y, -hy, hy)) is the identity inside it and flattens everything outside onto its border (Jacobian 1
// inside, 0 outside), so the area enclosed by the IMAGE C(dA) of A's outline is exactly area(A n B) = the contour integral of x~ dy~ along
// it.  Along an edge p + t d, t in [0, 1], y~ moves (at the rate dy) only while y is inside the slab |y| < hy, i.e. for t in [b0, b1], and
// x~ = clamp(px + t dx) is piecewise linear with corners where the edge meets x = +-hx, at t = a0 <= a1.  So the edge contributes
//     dy * integral over [b0, b1] of clamp(px + t dx, -hx, hx) dt
// = dy times three trapezoids over the nodes b0 <= a0' <= a1' <= b1 (the a's clamped into [b0, b1], the b's into [0, 1]).  Four independent
// edges of ~40 float64 instructions, straight-line: no vertex lists in LDS, no branches, no loop-carried chain.  The Sutherland-Hodgman
// walk this replaces (rounds 1-4; it is what oracle/warp_oracle.c still does) was a serial chain of ~600 float64 instructions that lasted
// 4 of the kernel's 8 us although only a few lanes of a wave ever ran it -- one wave issues one float64 instruction per 4-8 cycles whatever
// the number of active lanes, so the kernel's duration is the instruction count of this path.  (A first form, the symmetric shoelace sum over
// the six image points of every edge with all four line parameters sorted, was 65 instructions per edge: 0.3 us more.  The edge loop rolled
// into one copy of the code -- in case the cold instruction cache were the limit -- measured 1 us SLOWER: it is issue, not fetch.)
// Robust by construction: a parameter that is wrong (dx ~ 0: the reciprocal overflows to inf / NaN, the clamps turn it into 0 or 1) only
// puts a node ON the edge where no corner is, or misplaces a true corner by no more than the edge's distance from that line, i.e. by
// rounding error; touching and coincident edges need no tie rules.  Against 50-digit arithmetic: <= 2e-15 on the IoU (1e-14 for crossing
// 1000 : 1 slivers: the terms are of the order |edge of A| * hx; tests: 1e-13); the
// oracle's float64 clip, which works in world coordinates, is itself only good to ulp(|centre|) / size (tests/test_oracle_iou.py).
constexpr int kIouThreads = 64;
__device__ __forceinline__ double rcp_full_of(double z) {
    double r = __builtin_amdgcn_rcp(z);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double clamp_sym(double v, double h) { return fmin(fmax(v, -h), h); }

// sin and cos of one angle below 2^20 in magnitude: x = n pi/2 + r with pi/2 in three float64 pieces (Cody-Waite, one fma each: |r| <= pi/4
// to an absolute 2e-16), the two polynomial kernels of fdlibm (k_sin.c / k_cos.c coefficients), quadrant by n mod 4.  Absolute error
// <= 3e-16 -- what a box corner needs; the library's sincos also keeps the RELATIVE error of a result near 0, and pays a double-double
// reduction for it.  No branch inside, so two calls interleave (the library's pair ran one after the other: 1.6 us of the kernel's 6.5).
__device__ __forceinline__ void sincos_reduced(double x, double& sn, double& cs) {
    const double n = __builtin_rint(x * 0.6366197723675814);
    double r = __builtin_fma(n, -1.5707963267948966, x);
    r = __builtin_fma(n, -6.123233995736766e-17, r);
    r = __builtin_fma(n, 1.4973849048591698e-33, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double sr = __builtin_fma(r * z, ps, r);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)n;
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}
// (sin, cos) of two angles at once; the library only for an angle of 2^20 and beyond, an infinity or a NaN
__device__ __forceinline__ void sincos_pair(double x0, double x1, double& s0, double& c0, double& s1, double& c1) {
    if (fabs(x0) < 1048576.0 && fabs(x1) < 1048576.0) {
        sincos_reduced(x0, s0, c0);
        sincos_reduced(x1, s1, c1);
    } else {
        sincos(x0, &s0, &c0);
        sincos(x1, &s1, &c1);
    }
}
// A: centre (lx, ly) in B's frame, half-length vector (ux, uy), half-width vector (vx, vy); B: half sizes hx (length, along x), hy.
__device__ __forceinline__ double intersection_area(double lx, double ly, double ux, double uy, double vx, double vy, double hx, double hy) {
    // corners, counter-clockwise from (-u - v); edges 0 / 2 run along +-2u, edges 1 / 3 along +-2v: four reciprocals serve the sixteen
    // line parameters t = (+-h - p) / d = -p / d -+ h / |d|  (ordered as written)
    const double cx[4] = {lx - ux - vx, lx + ux - vx, lx + ux + vx, lx - ux + vx}, cy[4] = {ly - uy - vy, ly + uy - vy, ly + uy + vy, ly - uy + vy};
    const double dux = 2.0 * ux, duy = 2.0 * uy, dvx = 2.0 * vx, dvy = 2.0 * vy;
    const double iux = rcp_full_of(dux), iuy = rcp_full_of(duy), ivx = rcp_full_of(dvx), ivy = rcp_full_of(dvy);
    const double wux = hx * fabs(iux), wuy = hy * fabs(iuy), wvx = hx * fabs(ivx), wvy = hy * fabs(ivy);
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double sgn = k < 2 ? 1.0 : -1.0;
        const double dx = sgn * ((k & 1) ? dvx : dux), dy = sgn * ((k & 1) ? dvy : duy);
        const double ix = sgn * ((k & 1) ? ivx : iux), iy = sgn * ((k & 1) ? ivy : iuy);
        const double wx = (k & 1) ? wvx : wux, wy = (k & 1) ? wvy : wuy;
        const double px = cx[k], py = cy[k];
        const double mx = -px * ix, my = -py * iy;
        // (fmin / fmax return the other operand for a NaN: every parameter passes a clamp against finite bounds)
        const double b0 = fmin(fmax(my - wy, 0.0), 1.0), b1 = fmin(fmax(my + wy, 0.0), 1.0);
        const double a0 = fmin(fmax(mx - wx, b0), b1), a1 = fmin(fmax(mx + wx, b0), b1);
        const double x0 = clamp_sym(__builtin_fma(b0, dx, px), hx), x1 = clamp_sym(__builtin_fma(a0, dx, px), hx);
        const double x2 = clamp_sym(__builtin_fma(a1, dx, px), hx), x3 = clamp_sym(__builtin_fma(b1, dx, px), hx);
        const double sum = __builtin_fma(x2 + x3, b1 - a1, __builtin_fma(x1 + x2, a1 - a0, (x0 + x1) * (a0 - b0)));
        acc = __builtin_fma(dy, sum, acc);  // (twice the edge's integral: the trapezoids' 1 / 2 is applied once, below)
    }
    return fabs(0.5 * acc);
}

// IoU of box A with box B, centres / sizes in float64.  First a rejection test that needs no square root and no sin / cos: the
// circumscribed circles have diameters sqrt(w^2 + h^2), and ((da + db) / 2)^2 <= (da^2 + db^2) / 2, so centres further apart than that
// bound belong to boxes that cannot intersect -- exactly 0 is returned for them.  In a tracker step almost every pair ends there (the 1e-4
// margin keeps near-touching pairs on the exact path; NaNs fall through to it as well).  The lanes that remain turn A into B's frame
// (`heading_a(c, s)` turns the cosine and sine of `ayaw` into those of A's heading only now: the tracker step derives them from the
// detection's direction vector) and sum the contour.  Conventions of the oracle's clip that are kept: length h along the heading, width w across ("world",
// rbox.py:87-95); an intersection below 1e-14 of the boxes' area is the rounding noise of twenty signed terms of that order and reads as 0
// (disjoint boxes give exactly 0, as they do there); a B with ONE negative size is a clockwise clip polygon there, which clips everything
// away: 0.
template <typename HeadingA>
__device__ __forceinline__ double pair_iou(bool valid, double acx, double acy, double aw, double ah, double ayaw, HeadingA&& heading_a, const double (&B)[5]) {
    const double dx = acx - B[0], dy = acy - B[1];
    const double da2 = aw * aw + ah * ah, db2 = B[2] * B[2] + B[3] * B[3];
    const double area = fabs(aw * ah) + fabs(B[2] * B[3]);
    if (!valid || (dx * dx + dy * dy > 0.5 * (da2 + db2) * 1.0001 && area > 0)) return 0.0;
    double ca, sa, cb, sb;
    sincos_pair(ayaw, B[4], sa, ca, sb, cb);
    heading_a(ca, sa);  // (in: cosine and sine of `ayaw`; out: of A's heading in B's world)
    const double cr = ca * cb + sa * sb, sr = sa * cb - ca * sb;  // A's heading relative to B's
    const double hxa = 0.5 * ah, hya = 0.5 * aw;
    double inter = intersection_area(cb * dx + sb * dy, cb * dy - sb * dx, cr * hxa, sr * hxa, -sr * hya, cr * hya, 0.5 * fabs(B[3]), 0.5 * fabs(B[2]));
    if (inter < 1e-14 * area || B[2] * B[3] < 0) inter = 0.0;
    const double uni = area - inter;
    return uni > 0 ? inter / uni : 0.0;
}

template <typename T>
__global__ __launch_bounds__(kIouThreads) void rbox_iou_kernel(const T* __restrict__ a, int na, int sa, const T* __restrict__ b, int nb, int sb,
                                                       T* __restrict__ out) {
    // (every scalar argument and all five numbers of box A -- the same in every lane -- are fetched up front, in one scalar-memory round trip
    // each: left to the compiler they arrive one by one, each in front of its first use, A's yaw behind the rejection branch)
    asm volatile("" ::"s"(a), "s"(na), "s"(sa), "s"(b), "s"(nb), "s"(sb), "s"(out));
    const int j = blockIdx.x * kIouThreads + threadIdx.x;  // column (box of b) -> coalesced stores
    const int i = blockIdx.y;
    const bool valid = j < nb;
    const T* pa = a + (int64_t)i * sa;
    const T* pb = b + (int64_t)(valid ? j : 0) * sb;
    const T a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], a4 = pa[4];
    const double B[5] = {(double)pb[0], (double)pb[1], (double)pb[2], (double)pb[3], (double)pb[4]};
    asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4));
    const double A[5] = {(double)a0, (double)a1, (double)a2, (double)a3, (double)a4};
    const double v = pair_iou(valid, A[0], A[1], A[2], A[3], A[4], [](double&, double&) {}, B);
    if (valid) out[(int64_t)i * nb + j] = (T)v;
}

// ---- rotated boxes through a similarity H (reference bev/rbox.py:173-219) ------------------------------------------
// H is normalised (h[8] = 1) and affine (h[6], h[7] ~ 0: the host checks); scale = sqrt(h00^2 + h10^2) (dist_world_bev).
struct SimH {
    double h[9];
    double scale;
};
template <typename T>
__device__ __forceinline__ void box_through(const T* __restrict__ p, const SimH& H, int src_is_bev, double (&o)[5]) {
    const double x = (double)p[0], y = (double)p[1], r = (double)p[4];
    // yaw2v (rbox.py:28-36): a BEV yaw is measured from the v axis (sin, cos), a world yaw from the x axis (cos, sin);
    // angle_world_bev pushes that direction through H and reads it back in the TARGET's convention (v2yaw, :20-27)
    double sn, cs;
    sincos(r, &sn, &cs);
    const double vx = src_is_bev ? sn : cs, vy = src_is_bev ? cs : sn;
    const double tx = H.h[0] * vx + H.h[1] * vy, ty = H.h[3] * vx + H.h[4] * vy;
    o[4] = src_is_bev ? atan2(ty, tx) : atan2(tx, ty);
    const double X = H.h[0] * x + H.h[1] * y + H.h[2], Y = H.h[3] * x + H.h[4] * y + H.h[5], W = H.h[6] * x + H.h[7] * y + H.h[8];
    o[0] = X / W;  // pts_world_bev (rbox.py:136-151)
    o[1] = Y / W;
    o[2] = (double)p[2] * H.scale;  // dist_world_bev (:153-160)
    o[3] = (double)p[3] * H.scale;
}

template <typename T>
__global__ __launch_bounds__(256) void rbox_transform_kernel(const T* __restrict__ in, int n, int stride, const SimH H, int src_is_bev, T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double o[5];
    box_through<T>(in + (int64_t)i * stride, H, src_is_bev, o);
#pragma unroll
    for (int k = 0; k < 5; k++) out[(int64_t)i * 5 + k] = (T)o[k];
}

// ---- one tracker step in one launch (reference bev/tool/rbox_tracking_BrnoCompSpeed.py:88-109 with the association
// front-end of bev/tracker/rbox_tracker.py:383-405): SCORING workgroup (jb, i), i < n, scores detection i against 64 predicted tracker
// boxes and writes the IoU row segment and the `iou > threshold` gate; OUTPUT workgroups (rows n .. of the grid) write the world boxes and
// their image-plane centres (rbox_world_img, rbox.py:221-226), 64 detections each.  What a pair needs first is only the detection's world
// CENTRE and SIZE (two divisions): the yaw -- a sincos and an atan2, the long dependent chain round 3 ran in thread 0 of every workgroup
// before anybody could start -- is needed by the few pairs that survive the rejection test (as a direction vector) and by the outputs.
template <typename T>
__global__ __launch_bounds__(kIouThreads) void tracker_step_kernel(const T* __restrict__ dets, int n, int sd, const T* __restrict__ trks, int m, int st,
                                                                   const SimH Hwb, const H9 Him, int has_img, double thr, T* __restrict__ dets_world,
                                                                   T* __restrict__ iou, uint8_t* __restrict__ cand, T* __restrict__ dets_img) {
    const int tid = (int)threadIdx.x;
    // Every scalar argument is wanted NOW: left alone, the compiler fetches each kernel argument right in front of its first use, behind
    // the branches -- five scalar-memory round trips one after the other at the head of the scoring path.
    asm volatile("" ::"s"(dets), "s"(trks), "s"(n), "s"(sd), "s"(m), "s"(st), "s"(has_img), "s"(thr), "s"(dets_world), "s"(iou), "s"(cand), "s"(dets_img));
    if ((int)blockIdx.y >= n) {
        // OUTPUT workgroups (rows n .. of the grid, column 0 only): lane l writes the world box and the image-plane centre of detection
        // 64 (blockIdx.y - n) + l.  The yaw's sincos + atan2 chain runs here, 64 detections at a time, beside the scoring workgroups --
        // round 3 ran it in lane 0 of every scoring workgroup, in front of the pairs.
        const int i = ((int)blockIdx.y - n) * kIouThreads + tid;
        if (blockIdx.x != 0 || i >= n) return;
        double o[5];
        box_through<T>(dets + (int64_t)i * sd, Hwb, 1, o);
#pragma unroll
        for (int k = 0; k < 5; k++) dets_world[(int64_t)i * 5 + k] = (T)o[k];
        if (has_img) {
            const double Xi = Him.h[0] * o[0] + Him.h[1] * o[1] + Him.h[2], Yi = Him.h[3] * o[0] + Him.h[4] * o[1] + Him.h[5];
            const double Wi = Him.h[6] * o[0] + Him.h[7] * o[1] + Him.h[8];
            dets_img[(int64_t)i * 2] = (T)(Xi / Wi);
            dets_img[(int64_t)i * 2 + 1] = (T)(Yi / Wi);
        }
        return;
    }
    if (m <= 0) return;
    const int i = blockIdx.y;
    const int j = blockIdx.x * kIouThreads + tid;
    const bool valid = j < m;
    const T* pd = dets + (int64_t)i * sd;
    const T* pb = trks + (int64_t)(valid ? j : 0) * st;
    // the detection (the same five scalars in every lane: one scalar-memory fetch, yaw included -- read where the heading needs it, the
    // yaw's fetch sat behind the rejection branch: a second round trip, 1.5 of the kernel's 8 us) and the lane's tracker box, side by side
    const T d0 = pd[0], d1 = pd[1], d2 = pd[2], d3 = pd[3], d4 = pd[4];
    const double B[5] = {(double)pb[0], (double)pb[1], (double)pb[2], (double)pb[3], (double)pb[4]};
    asm volatile("" ::"s"(d0), "s"(d1), "s"(d2), "s"(d3), "s"(d4));
    // centre and size of the detection in the world, as box_through computes them (every lane: the same scalars, no LDS round trip);
    // rounded to the storage type like the dets_world output -- the box the tracker would see
    const double x = (double)d0, y = (double)d1, det_yaw = (double)d4;
    const double X = Hwb.h[0] * x + Hwb.h[1] * y + Hwb.h[2], Y = Hwb.h[3] * x + Hwb.h[4] * y + Hwb.h[5], W = Hwb.h[6] * x + Hwb.h[7] * y + Hwb.h[8];
    const double acx = (double)(T)(X / W), acy = (double)(T)(Y / W), aw = (double)(T)((double)d2 * Hwb.scale), ah = (double)(T)((double)d3 * Hwb.scale);
    const double v = pair_iou(valid, acx, acy, aw, ah, det_yaw, [&](double& c, double& s_) {
        // The world heading of a BEV detection is the direction H gives its (sin yaw, cos yaw) vector (angle_world_bev, rbox.py:162-171):
        // its cosine and sine are that vector normalised -- no atan2 followed by a sincos of the result.  (For float32 boxes this is the
        // heading BEFORE the yaw is rounded to float32 for dets_world: 6e-8 rad, inside the 2e-6 the float32 IoU is compared at.)
        const double sn = s_, cs = c;
        const double tx = Hwb.h[0] * sn + Hwb.h[1] * cs, ty = Hwb.h[3] * sn + Hwb.h[4] * cs;
        const double q = tx * tx + ty * ty;  // 1 / sqrt(q): the hardware estimate and two Newton steps (<= 2 ulp; a square root and a quotient cost four times that)
        double inv = __builtin_amdgcn_rsq(q);
        inv = __builtin_fma(0.5 * inv, __builtin_fma(-q * inv, inv, 1.0), inv);
        inv = __builtin_fma(0.5 * inv, __builtin_fma(-q * inv, inv, 1.0), inv);
        c = tx * inv, s_ = ty * inv;
    }, B);
    if (valid) {
        iou[(int64_t)i * m + j] = (T)v;
        cand[(int64_t)i * m + j] = (uint8_t)((double)(T)v > thr);
    }
}

// ---- alpha composite (reference bev/tool/compo.py:16-23) ------------------------------------------------
// out = uint8(min(round_half_even(fg * (mask / 255) + bg * (1 - mask / 255)), 255)) as numpy evaluates it in float64, computed
// exactly in integers: with N = fg m + bg (255 - m) the float64 value is N / 255 up to 2.3e-13 and N / 255 is never closer than
// 1 / 510 to a rounding boundary (2 N - 255 is odd), so the result is floor((N + 127) / 255) <= 255; (x * 0x8081) >> 23 == x / 255
// for x < 2^16.  (Checked against the float64 expression for all 2^24 byte triples: tests/test_host_api.py.)  16 bytes per lane.
__device__ __forceinline__ uint32_t composite_byte(uint32_t bg, uint32_t fg, uint32_t m) {
    return ((__umul24(fg, m) + __umul24(bg, 255u - m) + 127u) * 0x8081u) >> 23;
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

// ---- cv2.resize(img, (w, h)), INTER_LINEAR, 8-bit: the resize the reference's "small" branch applies before it warps
// (vis_homo.py:90).  OpenCV's classic bilinear path (resize.cpp, restated from memory -- parity unpinned, oracle/resize_oracle.c):
// sampling position (d + 0.5) * scale - 0.5 evaluated in float64 and rounded to float32, fraction in float32, 11-bit coefficients
// (saturate_cast<short>((1 - f) * 2048), (f * 2048)), the horizontal sums as integers, the vertical pass
// ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2; columns whose right tap would lie beyond the last source column read ONE
// tap with weight 2048, rows are clipped to the image and keep their coefficients.  One lane per destination pixel.
__device__ __forceinline__ void resize_axis(int d, double scale, int src_n, bool clamp, int& s, int& c0, int& c1, bool& one_tap) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    one_tap = false;
    if (clamp) {
        if (s < 0) s = 0, f = 0.f;
        if (s + 1 >= src_n) {
            one_tap = true;  // (the first clamped column and every one after it: sx is monotone in dx)
            if (s >= src_n - 1) s = src_n - 1, f = 0.f;
        }
    }
    auto sat = [](float v) {
        const float r = rintf(v);
        return (int)fminf(fmaxf(r, -32768.f), 32767.f);
    };
    c0 = sat((1.f - f) * 2048.f);
    c1 = sat(f * 2048.f);
}
template <int C>
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int src_h, int src_w, int dst_h, int dst_w,
                                                               int64_t src_fs, int64_t src_rs, int64_t dst_fs, int64_t dst_rs, double scale_x, double scale_y,
                                                               int box2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dst_w) return;
    const uint8_t* frame = src + (int64_t)blockIdx.z * src_fs;
    uint8_t* d = dst + (int64_t)blockIdx.z * dst_fs + (int64_t)y * dst_rs + (int64_t)x * C;
    if (box2) {  // scale exactly 2 x 2: INTER_LINEAR is INTER_AREA's box mean there
        const uint8_t* r0 = frame + (int64_t)(2 * y) * src_rs + (int64_t)(2 * x) * C;
        const uint8_t* r1 = r0 + src_rs;
#pragma unroll
        for (int k = 0; k < C; k++) d[k] = (uint8_t)(((int)r0[k] + (int)r0[C + k] + (int)r1[k] + (int)r1[C + k] + 2) >> 2);
        return;
    }
    int sx, a0, a1, sy, b0, b1;
    bool one, unused;
    resize_axis(x, scale_x, src_w, true, sx, a0, a1, one);
    resize_axis(y, scale_y, src_h, false, sy, b0, b1, unused);
    const int y0 = min(max(sy, 0), src_h - 1), y1 = min(max(sy + 1, 0), src_h - 1);
    const uint8_t* p0 = frame + (int64_t)y0 * src_rs + (int64_t)sx * C;
    const uint8_t* p1 = frame + (int64_t)y1 * src_rs + (int64_t)sx * C;
    const int right = one ? 0 : C;  // (never read beyond the row: a one-tap column takes its own pixel twice, the second with weight 0)
#pragma unroll
    for (int k = 0; k < C; k++) {
        const int h0 = one ? (int)p0[k] * 2048 : (int)p0[k] * a0 + (int)p0[right + k] * a1;
        const int h1 = one ? (int)p1[k] * 2048 : (int)p1[k] * a0 + (int)p1[right + k] * a1;
        d[k] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}


// The same arithmetic, four destination pixels per lane -- the form a batch of frames runs (32 x 1080p -> 852 x 480 is 199 MB in, 39 MB out:
// HBM-bound, where the one-pixel kernel above issued twelve byte loads and three byte stores per pixel).  Per pixel and source row ONE
// unaligned 8-byte load holds both taps (2 C <= 8 bytes from sx * C; global memory takes unaligned vector loads on gfx950, a window that
// straddles a 64-byte line costs a second request), the lane's 4 C result bytes leave as C aligned dwords.  The host takes this kernel
// only when every row of both images starts 4-byte aligned and a source row holds >= 8 bytes; a window that would end beyond its row (the
// last columns: one-tap pixels) is fetched byte by byte, as are the <= 3 pixels of a ragged right edge.
template <int C>
__global__ __launch_bounds__(256) void resize_linear_u8_px4_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int src_h, int src_w, int dst_h, int dst_w,
                                                                   int64_t src_fs, int64_t src_rs, int64_t dst_fs, int64_t dst_rs, double scale_x, double scale_y) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= dst_w) return;
    const uint8_t* frame = src + (int64_t)blockIdx.z * src_fs;
    uint8_t* d = dst + (int64_t)blockIdx.z * dst_fs + (int64_t)y * dst_rs + (int64_t)x0 * C;
    int sy, b0, b1;
    bool unused;
    resize_axis(y, scale_y, src_h, false, sy, b0, b1, unused);
    const uint8_t* r0 = frame + (int64_t)min(max(sy, 0), src_h - 1) * src_rs;
    const uint8_t* r1 = frame + (int64_t)min(max(sy + 1, 0), src_h - 1) * src_rs;
    const int row_bytes = src_w * C;
    uint32_t word[C] = {};
#pragma unroll
    for (int p = 0; p < 4; p++) {
        if (x0 + p < dst_w) {
            int sx, a0, a1;
            bool one;
            resize_axis(x0 + p, scale_x, src_w, true, sx, a0, a1, one);
            const int off = sx * C;
            uint32_t t0[2 * C], t1[2 * C];  // both taps of both rows, channel by channel
            if (off + 8 <= row_bytes) {
                uint64_t w0, w1;
                __builtin_memcpy(&w0, r0 + off, 8);
                __builtin_memcpy(&w1, r1 + off, 8);
#pragma unroll
                for (int k = 0; k < 2 * C; k++) t0[k] = (uint32_t)(w0 >> (8 * k)) & 255u, t1[k] = (uint32_t)(w1 >> (8 * k)) & 255u;
            } else {
                const int right = one ? 0 : C;  // (never read beyond the row: a one-tap column takes its own pixel twice)
#pragma unroll
                for (int k = 0; k < C; k++) t0[k] = r0[off + k], t0[C + k] = r0[off + right + k], t1[k] = r1[off + k], t1[C + k] = r1[off + right + k];
            }
#pragma unroll
            for (int k = 0; k < C; k++) {
                const int h0 = one ? (int)t0[k] * 2048 : (int)t0[k] * a0 + (int)t0[C + k] * a1;
                const int h1 = one ? (int)t1[k] * 2048 : (int)t1[k] * a0 + (int)t1[C + k] * a1;
                const uint32_t v = (uint32_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 255u;
                const int byte = p * C + k;  // position among the lane's 4 C result bytes
                word[byte >> 2] |= v << (8 * (byte & 3));
            }
        }
    }
    if (x0 + 4 <= dst_w) {
        uint32_t* dw = reinterpret_cast<uint32_t*>(d);
#pragma unroll
        for (int k = 0; k < C; k++) dw[k] = word[k];
    } else {
        for (int i = 0; i < (dst_w - x0) * C; i++) d[i] = (uint8_t)(word[i >> 2] >> (8 * (i & 3)));
    }
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
    const int vec16 = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    const int block = 256;
    const int64_t want = (n + block - 1) / block;
    // float32 (two points per lane and step): grid-stride above 8 workgroups per CU; float64 (one point per lane): a workgroup per 256
    // points and no loop at all -- A/B of the grid caps 8 / 16 / 32 per CU / none, round 4 (profiles/r04_points_ab.txt): float32 within
    // 1 % of each other (none: -5 %), float64 5.43 / 5.62 / 5.71 / 5.89 TB/s
    const int64_t cap = dtype == 2 ? (int64_t)1 << 30 : 256 * 8;
    const int grid = (int)(want < cap ? want : cap);
    if (dtype == 2) {
        if (dim == 2)
            hipLaunchKernelGGL((project_points_kernel<double, 2>), dim3(grid), dim3(block), 0, stream, (const double*)in, (double*)out, n, h, vec16);
        else
            hipLaunchKernelGGL((project_points_kernel<double, 3>), dim3(grid), dim3(block), 0, stream, (const double*)in, (double*)out, n, h, vec16);
    } else {
        if (dim == 2)
            hipLaunchKernelGGL((project_points_kernel<float, 2>), dim3(grid), dim3(block), 0, stream, (const float*)in, (float*)out, n, h, vec16);
        else
            hipLaunchKernelGGL((project_points_kernel<float, 3>), dim3(grid), dim3(block), 0, stream, (const float*)in, (float*)out, n, h, vec16);
    }
    return hipGetLastError();
}

hipError_t launch_rbox_iou(const void* a, int na, int a_stride, const void* b, int nb, int b_stride, void* out, int dtype,
                           hipStream_t stream) {
    if (na == 0 || nb == 0) return hipSuccess;
    (void)hipGetLastError();
    const dim3 block(kIouThreads), grid((nb + kIouThreads - 1) / kIouThreads, na);
    if (dtype == 2)
        hipLaunchKernelGGL(rbox_iou_kernel<double>, grid, block, 0, stream, (const double*)a, na, a_stride, (const double*)b, nb, b_stride, (double*)out);
    else
        hipLaunchKernelGGL(rbox_iou_kernel<float>, grid, block, 0, stream, (const float*)a, na, a_stride, (const float*)b, nb, b_stride, (float*)out);
    return hipGetLastError();
}

hipError_t launch_resize_linear_u8(const uint8_t* src, uint8_t* dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels, int64_t src_fs,
                                   int64_t src_rs, int64_t dst_fs, int64_t dst_rs, hipStream_t stream) {
    (void)hipGetLastError();
    // cv::resize: scale = 1 / ((double)dst / src) -- two roundings, not src / dst
    const double scale_x = 1.0 / ((double)dst_w / src_w), scale_y = 1.0 / ((double)dst_h / src_h);
    const int box2 = fabs(scale_x - 2.0) < 2.220446049250313e-16 && fabs(scale_y - 2.0) < 2.220446049250313e-16;
    // four pixels per lane wherever the dword stores and the 8-byte tap windows are legal (see the kernel); the 2 x 2 box mean and odd layouts
    // stay with the one-pixel kernel
    const bool px4 = !box2 && (int64_t)src_w * channels >= 8 && ((uintptr_t)dst & 3) == 0 && (dst_rs & 3) == 0 && (dst_fs & 3) == 0;
    if (px4) {
        const dim3 block4(256), grid4((dst_w + 1023) / 1024, dst_h, batch);
#define BEVWARP_RESIZE4_CASE(C)                                                                                                                            \
    case C:                                                                                                                                               \
        hipLaunchKernelGGL(resize_linear_u8_px4_kernel<C>, grid4, block4, 0, stream, src, dst, src_h, src_w, dst_h, dst_w, src_fs, src_rs, dst_fs, dst_rs, \
                           scale_x, scale_y);                                                                                                             \
        break;
        switch (channels) {
            BEVWARP_RESIZE4_CASE(1)
            BEVWARP_RESIZE4_CASE(2)
            BEVWARP_RESIZE4_CASE(3)
            default:
                BEVWARP_RESIZE4_CASE(4)
        }
#undef BEVWARP_RESIZE4_CASE
        return hipGetLastError();
    }
    const dim3 block(256), grid((dst_w + 255) / 256, dst_h, batch);
#define BEVWARP_RESIZE_CASE(C)                                                                                                                         \
    case C:                                                                                                                                            \
        hipLaunchKernelGGL(resize_linear_u8_kernel<C>, grid, block, 0, stream, src, dst, src_h, src_w, dst_h, dst_w, src_fs, src_rs, dst_fs, dst_rs, \
                           scale_x, scale_y, box2);                                                                                                    \
        break;
    switch (channels) {
        BEVWARP_RESIZE_CASE(1)
        BEVWARP_RESIZE_CASE(2)
        BEVWARP_RESIZE_CASE(3)
        default:
            BEVWARP_RESIZE_CASE(4)
    }
#undef BEVWARP_RESIZE_CASE
    return hipGetLastError();
}

namespace {
SimH make_sim(const double* H, double scale) {
    SimH s;
    for (int i = 0; i < 9; i++) s.h[i] = H[i];
    s.scale = scale;
    return s;
}
}  // namespace

hipError_t launch_rbox_transform(const void* in, int n, int stride, const double* H, double scale, int src_is_bev, void* out, int dtype, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    const SimH s = make_sim(H, scale);
    const dim3 block(256), grid((n + 255) / 256);
    if (dtype == 2)
        hipLaunchKernelGGL(rbox_transform_kernel<double>, grid, block, 0, stream, (const double*)in, n, stride, s, src_is_bev, (double*)out);
    else
        hipLaunchKernelGGL(rbox_transform_kernel<float>, grid, block, 0, stream, (const float*)in, n, stride, s, src_is_bev, (float*)out);
    return hipGetLastError();
}

hipError_t launch_tracker_step(const void* dets, int n, int det_stride, const void* trks, int m, int trk_stride, const double* H_world_bev, double scale,
                               const double* H_img_world, double iou_threshold, void* dets_world, void* iou, unsigned char* cand, void* dets_img,
                               int dtype, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    const SimH s = make_sim(H_world_bev, scale);
    H9 him;
    for (int i = 0; i < 9; i++) him.h[i] = H_img_world ? H_img_world[i] : 0.0;
    // rows 0 .. n - 1: scoring workgroups (detection x 64 tracks); rows n ..: output workgroups, 64 detections each (column 0 works)
    const dim3 block(kIouThreads), grid(m > 0 ? (m + kIouThreads - 1) / kIouThreads : 1, n + (n + kIouThreads - 1) / kIouThreads);
    if (dtype == 2)
        hipLaunchKernelGGL(tracker_step_kernel<double>, grid, block, 0, stream, (const double*)dets, n, det_stride, (const double*)trks, m, trk_stride, s, him,
                           H_img_world != nullptr, iou_threshold, (double*)dets_world, (double*)iou, cand, (double*)dets_img);
    else
        hipLaunchKernelGGL(tracker_step_kernel<float>, grid, block, 0, stream, (const float*)dets, n, det_stride, (const float*)trks, m, trk_stride, s, him,
                           H_img_world != nullptr, iou_threshold, (float*)dets_world, (float*)iou, cand, (float*)dets_img);
    return hipGetLastError();
}

}  // namespace bevwarp
