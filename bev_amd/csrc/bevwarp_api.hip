// bevwarp_api.hip -- the extern "C" surface declared in include/bevwarp.h: argument validation,
// launch geometry, error mapping.  No allocation, no synchronisation, no CPU fallback.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bevwarp.h"
#include "warp_kernels.h"

#pragma clang fp contract(off)

namespace {

thread_local char g_hip_error[256] = "";

int hip_fail(hipError_t e) {
    snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s", hipGetErrorName(e), hipGetErrorString(e));
    return BEVWARP_ERR_HIP;
}

bool finite9(const double* m, int n) {
    for (int i = 0; i < 9 * n; i++)
        if (!isfinite(m[i])) return false;
    return true;
}

// Evaluation block width of the reference algorithm (OpenCV WarpPerspectiveInvoker, BLOCK_SZ = 32):
// bh0 = min(16, h); bw0 = min(1024 / bh0, w).  Values depend on bw0 only.
int block_width(int dst_w, int dst_h) {
    const int bh0 = dst_h < 16 ? dst_h : 16;
    const int bw0 = 1024 / bh0;
    return bw0 < dst_w ? bw0 : dst_w;
}

// Does some byte of a strided source coincide with some byte of a strided destination?  The kernel reads taps of a frame while
// other workgroups store: an in-place call would corrupt silently, so overlap is refused.  Bounding byte ranges first; when those
// intersect but both sides walk their rows with ONE common stride S (equal row strides; frame strides multiples of S, or one
// frame), every row of either side starts at a fixed residue mod S, and two regions of one allocation that lie side by side (the
// left-half ROI of an image warped into its right half, say) are disjoint exactly when their residue intervals are.
bool regions_overlap(uintptr_t s0, uint64_t s_row_bytes, int s_rows, int64_t s_rs, int64_t s_fs, uintptr_t d0, uint64_t d_row_bytes, int d_rows, int64_t d_rs,
                     int64_t d_fs, int batch) {
    const uintptr_t s1 = s0 + (uint64_t)(batch - 1) * s_fs + (uint64_t)(s_rows - 1) * s_rs + s_row_bytes;
    const uintptr_t d1 = d0 + (uint64_t)(batch - 1) * d_fs + (uint64_t)(d_rows - 1) * d_rs + d_row_bytes;
    if (!(s0 < d1 && d0 < s1)) return false;
    if (s_rs == d_rs && s_rs > 0 && (batch == 1 || (s_fs % s_rs == 0 && d_fs % s_rs == 0)) && s_row_bytes + d_row_bytes <= (uint64_t)s_rs) {
        const uint64_t S = (uint64_t)s_rs, a = s0 % S, b = d0 % S;
        if ((b + S - a) % S >= s_row_bytes && (a + S - b) % S >= d_row_bytes) return false;  // column-disjoint: no row of one meets a row of the other
    }
    return true;
}

// division by invariants as a multiply-high; exact while n_max * d < 2^32, else the kernel divides
uint32_t div_magic(uint64_t n_max, uint32_t d) { return (n_max * d < (1ull << 32) && d > 1) ? (uint32_t)((1ull << 32) / d) + 1u : 0u; }

}  // namespace

#ifdef BEVWARP_CLOCK
namespace bevwarp { hipError_t debug_read_clock(unsigned long long* out4, int reset); }
extern "C" int bevwarp_debug_clock(unsigned long long* out4, int reset) { return (int)bevwarp::debug_read_clock(out4, reset); }
#endif

extern "C" {

int bevwarp_version(void) { return BEVWARP_ABI_VERSION; }

const char* bevwarp_strerror(int status) {
    switch (status) {
        case BEVWARP_OK: return "ok";
        case BEVWARP_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or misaligned stride)";
        case BEVWARP_ERR_UNSUPPORTED: return "unsupported dtype / channel count / interpolation";
        case BEVWARP_ERR_TOO_LARGE: return "source image side exceeds 32767 px, a row 16 MiB or a frame 2 GiB";
        case BEVWARP_ERR_NOT_FINITE: return "homography contains NaN or Inf";
        case BEVWARP_ERR_HIP: return "HIP runtime error (see bevwarp_last_hip_error)";
        case BEVWARP_ERR_OVERLAP: return "source and destination overlap in memory (an in-place warp would read taps that other workgroups have already overwritten)";
        default: return "unknown status";
    }
}

const char* bevwarp_last_hip_error(void) { return g_hip_error; }

int bevwarp_invert_homography(const double* S, double* D, int n) {
    if (!S || !D || n < 0) return BEVWARP_ERR_BAD_ARG;
    if (!finite9(S, n)) return BEVWARP_ERR_NOT_FINITE;
    for (int k = 0; k < n; k++, S += 9, D += 9) {
        // cv::invert, 3x3 double: cofactors times 1/det, in this evaluation order
        double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
        if (d == 0.0) {
            memset(D, 0, 9 * sizeof(double));
            continue;
        }
        d = 1.0 / d;
        double t[9];
        t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
        t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
        t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
        t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
        t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
        t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
        t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
        t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
        t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
        memcpy(D, t, sizeof(t));
    }
    return BEVWARP_OK;
}

}  // extern "C"

namespace {
struct PlanarOut {  // bevwarp_warp_planar: float32 channel planes instead of interleaved pixels of the source type
    int64_t plane_stride;
    const double *scale, *bias;
};

int warp_impl(const void* src, void* dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels,
              int64_t src_frame_stride, int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_row_stride, const double* M_inv,
              int m_count, int dtype, int interp, const double* border_value, void* stream, const PlanarOut* po, void* classes = nullptr,
              int classes_mode = 0, int64_t* classes_bytes = nullptr) {
    using namespace bevwarp;
    if (!classes_bytes && (!src || !dst || !M_inv)) return BEVWARP_ERR_BAD_ARG;
    if (batch < 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return BEVWARP_ERR_BAD_ARG;
    if (dtype != BEVWARP_U8 && dtype != BEVWARP_F32) return BEVWARP_ERR_UNSUPPORTED;
    if (interp != BEVWARP_NEAREST && interp != BEVWARP_LINEAR) return BEVWARP_ERR_UNSUPPORTED;
    if (channels < 1 || channels > 4) return BEVWARP_ERR_UNSUPPORTED;
    if (m_count != 1 && m_count != batch) return BEVWARP_ERR_BAD_ARG;
    const int esz = dtype == BEVWARP_U8 ? 1 : 4;
    const int64_t pix = (int64_t)channels * esz;
    if (src_row_stride < src_w * pix) return BEVWARP_ERR_BAD_ARG;
    if (batch > 1 && src_frame_stride < src_h * src_row_stride) return BEVWARP_ERR_BAD_ARG;
    if ((src_row_stride % esz) || (src_frame_stride % esz) || ((uintptr_t)src % esz)) return BEVWARP_ERR_BAD_ARG;
    if (po) {  // destination: `channels` float32 planes per frame
        if (dst_row_stride < (int64_t)dst_w * 4 || po->plane_stride < dst_h * dst_row_stride) return BEVWARP_ERR_BAD_ARG;
        if (batch > 1 && dst_frame_stride < channels * po->plane_stride) return BEVWARP_ERR_BAD_ARG;
        if ((dst_row_stride % 4) || (po->plane_stride % 4) || (dst_frame_stride % 4) || ((uintptr_t)dst % 4)) return BEVWARP_ERR_BAD_ARG;
    } else {
        if (dst_row_stride < dst_w * pix) return BEVWARP_ERR_BAD_ARG;
        if (batch > 1 && dst_frame_stride < dst_h * dst_row_stride) return BEVWARP_ERR_BAD_ARG;
        if ((dst_row_stride % esz) || (dst_frame_stride % esz) || ((uintptr_t)dst % esz)) return BEVWARP_ERR_BAD_ARG;
    }
    if (src_w > 32767 || src_h > 32767) return BEVWARP_ERR_TOO_LARGE;
    if ((int64_t)src_h * src_row_stride >= ((int64_t)1 << 31) || src_row_stride >= (1 << 24)) return BEVWARP_ERR_TOO_LARGE;  // (kernels use 24-bit multiplies)
    if (batch == 0) return BEVWARP_OK;
    if (po) {  // (float planes: bounding ranges only -- a frame's planes need not share the rows' stride)
        const uintptr_t s0 = (uintptr_t)src, s1 = s0 + (uint64_t)(batch - 1) * src_frame_stride + (uint64_t)(src_h - 1) * src_row_stride + (uint64_t)src_w * pix;
        const uint64_t dst_frame_bytes = (uint64_t)(channels - 1) * po->plane_stride + (uint64_t)(dst_h - 1) * dst_row_stride + (uint64_t)dst_w * 4;
        const uintptr_t d0 = (uintptr_t)dst, d1 = d0 + (uint64_t)(batch - 1) * dst_frame_stride + dst_frame_bytes;
        if (s0 < d1 && d0 < s1) return BEVWARP_ERR_OVERLAP;
    } else if (regions_overlap((uintptr_t)src, (uint64_t)src_w * pix, src_h, src_row_stride, src_frame_stride, (uintptr_t)dst, (uint64_t)dst_w * pix, dst_h,
                               dst_row_stride, dst_frame_stride, batch)) {
        return BEVWARP_ERR_OVERLAP;
    }

    WarpArgs a;
    memset(&a, 0, sizeof(a));
    a.src = (const uint8_t*)src;
    a.dst = (uint8_t*)dst;
    a.minv = M_inv;
    a.src_fs = src_frame_stride;
    a.src_rs = src_row_stride;
    a.dst_fs = dst_frame_stride;
    a.dst_rs = dst_row_stride;
    a.batch = batch;
    a.src_h = src_h;
    a.src_w = src_w;
    a.dst_h = dst_h;
    a.dst_w = dst_w;
    a.m_stride = m_count == 1 ? 0 : 9;
    a.bw0 = block_width(dst_w, dst_h);
    // tile = tile_width x tile_h destination pixels per workgroup.  16 rows (four per wave) is what launches that fill the
    // chip and the HBM-bound float formats take; the ALU-bound 8-bit formats amortise the per-tile set-up over 24 rows once the launch
    // fills the chip more than twice (taller tiles gain on footprints that lie inside the frame and lose on those the
    // frame's edge cuts up, whose tiles differ widely in cost: A/B in DESIGN.md section 6).
    // Launches that do not fill the chip -- a camera's single frame, the reference's own call shape -- take lower tiles, down
    // to one pass per wave, until there is a workgroup for every resident slot: the frame's latency is then one tile's, spread
    // over all CUs (720p -> 512^2: 18.7 -> 12.3 us, 1080p -> 1024^2: 15.2 -> 11.5 us; from four frames up 16 rows win).
    const int tw = tile_width(dtype);
    const int64_t per_row_of_tiles = (int64_t)batch * ((dst_w + tw - 1) / tw);
    const int64_t resident = resident_workgroups(dtype, channels, interp);
    a.tile_h = rows_per_pass() * 4;
    while (a.tile_h > rows_per_pass() && per_row_of_tiles * ((dst_h + a.tile_h - 1) / a.tile_h) < resident) a.tile_h /= 2;
    const int tall = rows_per_pass() * 6;  // (24 rows)
    if (dtype == BEVWARP_U8 && per_row_of_tiles * ((dst_h + tall - 1) / tall) >= 2 * resident) a.tile_h = tall;
    a.tiles_x = (dst_w + tw - 1) / tw;
    const int tiles_y = (dst_h + a.tile_h - 1) / a.tile_h;
    a.tiles_per_frame = a.tiles_x * tiles_y;
    a.total_tiles = (int64_t)batch * a.tiles_per_frame;
    const int64_t chunk = (a.total_tiles + 7) / 8;
    if (chunk * 8 > 0x7fffffffLL || dst_w > (1 << 20) || dst_h > (1 << 20)) return BEVWARP_ERR_TOO_LARGE;
    a.chunk = (int)chunk;
    // Frames of one launch usually share a footprint: left alone, all eight XCDs would be in the same part of a frame --
    // outside tiles (store-bound) or interior tiles (latency-bound) -- at the same time.  XCD k starts k/8 of a frame in.
    a.stagger = chunk >= a.tiles_per_frame ? a.tiles_per_frame / 8 : 0;
    // one resident round of half-height workgroups at the end of launches of at least two rounds (tile_h / 2 stays a multiple
    // of 4); measured neutral to -2.5 % on footprints whose tiles cost alike, -8..-14 % on a perspective BEV from 12 frames up
    const int64_t round_per_xcd = resident / 8;
    a.tail_split = (a.tile_h % (2 * rows_per_pass()) == 0 && chunk >= 2 * round_per_xcd) ? (int)round_per_xcd : 0;
    a.tpf_magic = div_magic((uint64_t)chunk * 8, (uint32_t)a.tiles_per_frame);
    a.tx_magic = div_magic((uint64_t)a.tiles_per_frame, (uint32_t)a.tiles_x);
    a.bw0_magic = div_magic((uint64_t)dst_w + tw, (uint32_t)a.bw0);
    // wide stores: one lane writes its 4 consecutive 8-bit pixels (4 C bytes; 12-byte stores need 4-byte alignment) or 16
    // bytes of float data
    const int dst_align = (dtype == BEVWARP_U8 && !po) ? (channels == 4 ? 16 : (channels == 2 ? 8 : 4)) : 16;
    a.dst_vec_ok = ((uintptr_t)dst % dst_align == 0) && (dst_row_stride % dst_align == 0) && (dst_frame_stride % dst_align == 0);
    if (po) {
        a.planar = 1;
        a.dst_ps = po->plane_stride;
        a.dst_vec_ok = a.dst_vec_ok && (po->plane_stride % 16 == 0);
        for (int k = 0; k < 4; k++) {
            const double sc = (po->scale && k < channels) ? po->scale[k] : 1.0, bi = (po->bias && k < channels) ? po->bias[k] : 0.0;
            if (!isfinite(sc) || !isfinite(bi)) return BEVWARP_ERR_NOT_FINITE;
            a.pscale[k] = (float)sc;
            a.pbias[k] = (float)bi;
        }
    }
    for (int k = 0; k < 4; k++) {
        const double b = (border_value && k < channels) ? border_value[k] : 0.0;
        if (!isfinite(b)) return BEVWARP_ERR_NOT_FINITE;
        a.bval_f[k] = (float)b;
        const double r = nearbyint(b);  // saturate_cast<uchar>: round half to even, clamp
        a.bval_u8[k] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    if (classes_bytes) {  // (bevwarp_tile_classes_bytes: the table of this launch geometry -- full tile, upper half, lower half per tile)
        *classes_bytes = 3 * a.total_tiles * (int64_t)sizeof(uint32_t);
        return BEVWARP_OK;
    }
    if (classes) {
        if ((uintptr_t)classes % 4) return BEVWARP_ERR_BAD_ARG;
        if (classes_mode == BEVWARP_CLASSES_FILL)
            a.classify_out = (uint32_t*)classes;
        else
            a.tile_class = (const uint32_t*)classes;
    }
    const hipError_t e = launch_warp(a, dtype, channels, interp, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}
}  // namespace

extern "C" {

int bevwarp_warp(const void* src, void* dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels,
                 int64_t src_frame_stride, int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_row_stride, const double* M_inv,
                 int m_count, int dtype, int interp, const double* border_value, void* stream) {
    return warp_impl(src, dst, batch, src_h, src_w, dst_h, dst_w, channels, src_frame_stride, src_row_stride, dst_frame_stride, dst_row_stride,
                     M_inv, m_count, dtype, interp, border_value, stream, nullptr);
}

int bevwarp_warp_classes(const void* src, void* dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels, int64_t src_frame_stride,
                         int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_row_stride, const double* M_inv, int m_count, int dtype, int interp,
                         const double* border_value, void* classes, int mode, void* stream) {
    if (!classes || (mode != BEVWARP_CLASSES_USE && mode != BEVWARP_CLASSES_FILL)) return BEVWARP_ERR_BAD_ARG;
    return warp_impl(src, dst, batch, src_h, src_w, dst_h, dst_w, channels, src_frame_stride, src_row_stride, dst_frame_stride, dst_row_stride,
                     M_inv, m_count, dtype, interp, border_value, stream, nullptr, classes, mode);
}

int64_t bevwarp_tile_classes_bytes(int batch, int src_h, int src_w, int dst_h, int dst_w, int channels, int dtype, int interp) {
    if (batch <= 0) return 0;
    int64_t n = 0;
    const int64_t esz = dtype == BEVWARP_U8 ? 1 : 4, srs = (int64_t)src_w * channels * esz, drs = (int64_t)dst_w * channels * esz;
    // (the geometry depends on the sizes and the format only; the pointers are placeholders that pass the argument checks)
    const int st = warp_impl((const void*)(uintptr_t)0x1000, (void*)(uintptr_t)0x700000000000ull, batch, src_h, src_w, dst_h, dst_w, channels, src_h * srs, srs, dst_h * drs, drs, (const double*)16, 1, dtype,
                             interp, nullptr, nullptr, nullptr, nullptr, 0, &n);
    return st == BEVWARP_OK ? n : (int64_t)st;
}

int bevwarp_warp_planar(const void* src, void* dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels,
                        int64_t src_frame_stride, int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_plane_stride,
                        int64_t dst_row_stride, const double* M_inv, int m_count, int dtype, int interp, const double* border_value,
                        const double* scale, const double* bias, void* stream) {
    const PlanarOut po = {dst_plane_stride, scale, bias};
    return warp_impl(src, dst, batch, src_h, src_w, dst_h, dst_w, channels, src_frame_stride, src_row_stride, dst_frame_stride, dst_row_stride,
                     M_inv, m_count, dtype, interp, border_value, stream, &po);
}

int bevwarp_composite(const void* bg, const void* fg, const void* mask, void* out, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!bg || !fg || !mask || !out))) return BEVWARP_ERR_BAD_ARG;
    const hipError_t e = bevwarp::launch_composite((const uint8_t*)bg, (const uint8_t*)fg, (const uint8_t*)mask, (uint8_t*)out, n, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

int bevwarp_warp_composite(const void* bg, int bg_h, int bg_w, int64_t bg_row_stride, const void* fg, const void* mask, int fg_h, int fg_w,
                           int64_t fg_row_stride, int64_t mask_row_stride, void* dst, int dst_h, int dst_w, int64_t dst_row_stride, int channels,
                           const double* M_inv_bg, const double* M_inv_cam, int fg_gray, void* stream) {
    using namespace bevwarp;
    if (!bg || !fg || !mask || !dst || !M_inv_bg || !M_inv_cam) return BEVWARP_ERR_BAD_ARG;
    if (bg_h <= 0 || bg_w <= 0 || fg_h <= 0 || fg_w <= 0 || dst_h <= 0 || dst_w <= 0) return BEVWARP_ERR_BAD_ARG;
    if (channels < 1 || channels > 4) return BEVWARP_ERR_UNSUPPORTED;
    if (fg_gray && channels != 3) return BEVWARP_ERR_UNSUPPORTED;  // BGR2GRAY needs three channels
    if (bg_row_stride < (int64_t)bg_w * channels || fg_row_stride < (int64_t)fg_w * channels || mask_row_stride < (int64_t)fg_w * channels ||
        dst_row_stride < (int64_t)dst_w * channels)
        return BEVWARP_ERR_BAD_ARG;
    if (bg_w > 32767 || bg_h > 32767 || fg_w > 32767 || fg_h > 32767 || dst_w > (1 << 20) || dst_h > (1 << 20)) return BEVWARP_ERR_TOO_LARGE;
    const int64_t rs_max = bg_row_stride > fg_row_stride ? (bg_row_stride > mask_row_stride ? bg_row_stride : mask_row_stride)
                                                         : (fg_row_stride > mask_row_stride ? fg_row_stride : mask_row_stride);
    if (rs_max >= (1 << 24) || (int64_t)bg_h * bg_row_stride >= ((int64_t)1 << 31) || (int64_t)fg_h * fg_row_stride >= ((int64_t)1 << 31) ||
        (int64_t)fg_h * mask_row_stride >= ((int64_t)1 << 31))
        return BEVWARP_ERR_TOO_LARGE;  // (the kernel's 24-bit multiplies)
    {  // the destination must not overlap a source (as for bevwarp_warp)
        const void* sp[3] = {bg, fg, mask};
        const int sh_[3] = {bg_h, fg_h, fg_h}, sw_[3] = {bg_w, fg_w, fg_w};
        const int64_t srs_[3] = {bg_row_stride, fg_row_stride, mask_row_stride};
        for (int i = 0; i < 3; i++)
            if (regions_overlap((uintptr_t)sp[i], (uint64_t)sw_[i] * channels, sh_[i], srs_[i], 0, (uintptr_t)dst, (uint64_t)dst_w * channels, dst_h, dst_row_stride, 0, 1))
                return BEVWARP_ERR_OVERLAP;
    }
    WarpArgs a;
    memset(&a, 0, sizeof(a));
    a.src = (const uint8_t*)bg, a.dst = (uint8_t*)dst, a.minv = M_inv_bg;
    a.src_rs = bg_row_stride, a.dst_rs = dst_row_stride;
    a.batch = 1, a.src_h = bg_h, a.src_w = bg_w, a.dst_h = dst_h, a.dst_w = dst_w;
    a.m_stride = 0;
    a.bw0 = block_width(dst_w, dst_h);
    a.xsrc[0] = (const uint8_t*)fg, a.xsrc[1] = (const uint8_t*)mask;
    a.xminv[0] = a.xminv[1] = M_inv_cam;
    a.xsrc_rs[0] = fg_row_stride, a.xsrc_rs[1] = mask_row_stride;
    a.xsrc_h[0] = a.xsrc_h[1] = fg_h, a.xsrc_w[0] = a.xsrc_w[1] = fg_w;
    a.fg_gray = fg_gray != 0;
    // one 12-wave workgroup per tile, one workgroup per CU: the tallest tile (<= the LDS copies' 16 rows) that still gives every
    // CU a workgroup
    const int tw = tile_width(BEVWARP_U8);
    a.tiles_x = (dst_w + tw - 1) / tw;
    const int64_t cus = resident_workgroups(BEVWARP_U8, channels, BEVWARP_LINEAR) / 4;
    a.tile_h = composite_max_rows();
    while (a.tile_h > rows_per_pass() && (int64_t)a.tiles_x * ((dst_h + a.tile_h - 1) / a.tile_h) < cus) a.tile_h /= 2;
    a.tiles_per_frame = a.tiles_x * ((dst_h + a.tile_h - 1) / a.tile_h);
    a.total_tiles = a.tiles_per_frame;
    a.chunk = (int)((a.total_tiles + 7) / 8);
    a.stagger = 0, a.tail_split = 0;
    a.tpf_magic = div_magic((uint64_t)a.chunk * 8, (uint32_t)a.tiles_per_frame);
    a.tx_magic = div_magic((uint64_t)a.tiles_per_frame, (uint32_t)a.tiles_x);
    a.bw0_magic = div_magic((uint64_t)dst_w + tw, (uint32_t)a.bw0);
    const int dst_align = channels == 4 ? 16 : (channels == 2 ? 8 : 4);
    a.dst_vec_ok = ((uintptr_t)dst % dst_align == 0) && (dst_row_stride % dst_align == 0);
    const hipError_t e = launch_warp_composite(a, channels, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

int bevwarp_resize(const void* src, void* dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels, int64_t src_frame_stride,
                   int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_row_stride, int dtype, int interp, void* stream) {
    if (!src || !dst || batch < 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return BEVWARP_ERR_BAD_ARG;
    if (dtype != BEVWARP_U8 || interp != BEVWARP_LINEAR || channels < 1 || channels > 4) return BEVWARP_ERR_UNSUPPORTED;
    if (src_row_stride < (int64_t)src_w * channels || dst_row_stride < (int64_t)dst_w * channels) return BEVWARP_ERR_BAD_ARG;
    if (batch > 1 && (src_frame_stride < src_h * src_row_stride || dst_frame_stride < dst_h * dst_row_stride)) return BEVWARP_ERR_BAD_ARG;
    if (src_w > (1 << 24) || src_h > (1 << 24) || dst_w > (1 << 24) || dst_h > 65535 || batch > 65535) return BEVWARP_ERR_TOO_LARGE;
    if (batch == 0) return BEVWARP_OK;
    if (regions_overlap((uintptr_t)src, (uint64_t)src_w * channels, src_h, src_row_stride, src_frame_stride, (uintptr_t)dst, (uint64_t)dst_w * channels, dst_h,
                        dst_row_stride, dst_frame_stride, batch))
        return BEVWARP_ERR_OVERLAP;
    const hipError_t e = bevwarp::launch_resize_linear_u8((const uint8_t*)src, (uint8_t*)dst, batch, src_h, src_w, dst_h, dst_w, channels, src_frame_stride,
                                                          src_row_stride, dst_frame_stride, dst_row_stride, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

int bevwarp_footprint(unsigned char* touched, int batch, int src_h, int src_w, int dst_h, int dst_w, const double* M_inv, int m_count,
                      int interp, void* stream) {
    if (!touched || !M_inv || batch < 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return BEVWARP_ERR_BAD_ARG;
    if (interp != BEVWARP_NEAREST && interp != BEVWARP_LINEAR) return BEVWARP_ERR_UNSUPPORTED;
    if (m_count != 1 && m_count != batch) return BEVWARP_ERR_BAD_ARG;
    if (src_w > 32767 || src_h > 32767 || dst_h > 65535 || batch > 65535) return BEVWARP_ERR_TOO_LARGE;
    if (batch == 0) return BEVWARP_OK;
    const hipError_t e = bevwarp::launch_footprint(touched, batch, src_h, src_w, dst_h, dst_w, M_inv, m_count == 1 ? 0 : 9,
                                                   block_width(dst_w, dst_h), interp, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

int bevwarp_project_points(const void* in, void* out, int64_t n, int dim, const double* H, int dtype, void* stream) {
    if (!H || n < 0 || (n > 0 && (!in || !out))) return BEVWARP_ERR_BAD_ARG;
    if (dim != 2 && dim != 3) return BEVWARP_ERR_BAD_ARG;
    if (dtype != BEVWARP_F32 && dtype != BEVWARP_F64) return BEVWARP_ERR_UNSUPPORTED;
    if (!finite9(H, 1)) return BEVWARP_ERR_NOT_FINITE;
    const int esz = dtype == BEVWARP_F32 ? 4 : 8;
    const int need = dim == 2 ? 2 * esz : esz;  // 2-D points move as one 8 / 16 byte unit
    if (((uintptr_t)in % need) || ((uintptr_t)out % need)) return BEVWARP_ERR_BAD_ARG;
    const hipError_t e = bevwarp::launch_project_points(in, out, n, dim, H, dtype, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

// H / H[2][2] when H is a similarity of the plane in the sense of bev/rbox.py:173-219 (last row ~ (0, 0, 1), equal scale on
// both axes -- the reference asserts both); scale = sqrt(h00^2 + h10^2)
static int normalise_similarity(const double* H, double* Hn, double* scale) {
    if (!H || !finite9(H, 1) || H[8] == 0.0) return BEVWARP_ERR_NOT_FINITE;
    for (int i = 0; i < 9; i++) Hn[i] = H[i] / H[8];
    if (fabs(Hn[6]) + fabs(Hn[7]) >= 1e-5) return BEVWARP_ERR_BAD_ARG;
    const double s0 = sqrt(Hn[0] * Hn[0] + Hn[3] * Hn[3]), s1 = sqrt(Hn[1] * Hn[1] + Hn[4] * Hn[4]);
    if (!(fabs(s0 - s1) < 1e-5)) return BEVWARP_ERR_BAD_ARG;
    *scale = s0;
    return BEVWARP_OK;
}

int bevwarp_rbox_transform(const void* boxes, int n, int stride, const double* H, int src_is_bev, void* out, int dtype, void* stream) {
    if (n < 0 || stride < 5 || (n > 0 && (!boxes || !out))) return BEVWARP_ERR_BAD_ARG;
    if (dtype != BEVWARP_F32 && dtype != BEVWARP_F64) return BEVWARP_ERR_UNSUPPORTED;
    double Hn[9], scale;
    const int st = normalise_similarity(H, Hn, &scale);
    if (st != BEVWARP_OK) return st;
    const hipError_t e = bevwarp::launch_rbox_transform(boxes, n, stride, Hn, scale, src_is_bev != 0, out, dtype, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

int bevwarp_tracker_step(const void* dets_bev, int n, int det_stride, const void* trks_world, int m, int trk_stride, const double* H_world_bev,
                         const double* H_img_world, double iou_threshold, void* dets_world, void* iou, unsigned char* candidates, void* dets_img,
                         int dtype, void* stream) {
    if (n < 0 || m < 0 || det_stride < 5 || trk_stride < 5) return BEVWARP_ERR_BAD_ARG;
    if (n > 0 && (!dets_bev || !dets_world)) return BEVWARP_ERR_BAD_ARG;
    if (n > 0 && m > 0 && (!trks_world || !iou || !candidates)) return BEVWARP_ERR_BAD_ARG;
    if (H_img_world && n > 0 && !dets_img) return BEVWARP_ERR_BAD_ARG;
    if (dtype != BEVWARP_F32 && dtype != BEVWARP_F64) return BEVWARP_ERR_UNSUPPORTED;
    if (n > 64000) return BEVWARP_ERR_TOO_LARGE;  // (grid rows: n scoring + n / 64 output workgroups <= 65535)
    if (!(iou_threshold == iou_threshold)) return BEVWARP_ERR_NOT_FINITE;
    double Hn[9], scale;
    const int st = normalise_similarity(H_world_bev, Hn, &scale);
    if (st != BEVWARP_OK) return st;
    if (H_img_world && !finite9(H_img_world, 1)) return BEVWARP_ERR_NOT_FINITE;
    const hipError_t e = bevwarp::launch_tracker_step(dets_bev, n, det_stride, trks_world, m, trk_stride, Hn, scale, H_img_world, iou_threshold, dets_world, iou,
                                                      candidates, dets_img, dtype, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

int bevwarp_rbox_iou(const void* a, int na, int a_stride, const void* b, int nb, int b_stride, void* out, int dtype, void* stream) {
    if (na < 0 || nb < 0 || a_stride < 5 || b_stride < 5) return BEVWARP_ERR_BAD_ARG;
    if (na > 0 && nb > 0 && (!a || !b || !out)) return BEVWARP_ERR_BAD_ARG;
    if (dtype != BEVWARP_F32 && dtype != BEVWARP_F64) return BEVWARP_ERR_UNSUPPORTED;
    if (na > 65535) return BEVWARP_ERR_TOO_LARGE;
    const hipError_t e = bevwarp::launch_rbox_iou(a, na, a_stride, b, nb, b_stride, out, dtype, (hipStream_t)stream);
    return e == hipSuccess ? BEVWARP_OK : hip_fail(e);
}

}  // extern "C"
