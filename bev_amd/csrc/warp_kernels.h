// Internal interface between the C ABI (bevwarp_api.hip) and the kernels.  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bevwarp {

constexpr int kNearest = 0;
constexpr int kLinear = 1;

struct WarpArgs {
    const uint8_t* src;
    uint8_t* dst;
    const double* minv;          // device, inverse matrices
    int64_t src_fs, src_rs;      // frame / row strides in bytes
    int64_t dst_fs, dst_rs;
    int64_t total_tiles;         // batch * tiles_per_frame (< 2^31)
    uint32_t tpf_magic, tx_magic, bw0_magic;  // floor(2^32 / d) + 1 for d = tiles_per_frame, tiles_x, bw0; 0 = divide
    int batch, src_h, src_w, dst_h, dst_w;
    int m_stride;                // 9 (one matrix per frame) or 0 (shared)
    int bw0;                     // evaluation block width of the reference algorithm
    int tiles_x, tiles_per_frame;
    int tile_h;                  // rows per workgroup (a multiple of the 4 rows one pass of its waves covers)
    int chunk;                   // items per XCD: grid = 8 * chunk
    int tail_split;              // the last tail_split tiles of every XCD's dispatch order run as two half-height workgroups
    int stagger;                 // XCD k walks its run starting k * stagger items in (0: all start at the run's first item)
    int dst_vec_ok;              // destination layout admits the wide stores
    float bval_f[4];
    uint8_t bval_u8[4];
    // planar float output of 8-bit warps (bevwarp_warp_planar): dst[c][y][x] = float(pixel) * pscale[c] + pbias[c]
    int planar;
    int64_t dst_ps;              // bytes between channel planes
    float pscale[4], pbias[4];
    // composite (bevwarp_warp_composite): sources 1 and 2 (foreground, mask) beside src / minv / src_rs / src_h / src_w (background)
    const uint8_t* xsrc[2];
    const double* xminv[2];
    int64_t xsrc_rs[2];
    int xsrc_h[2], xsrc_w[2];
    int fg_gray;                 // composite, bw_mode: source 1 (the foreground) is converted to grey tap by tap
    // per-tile verdicts (bevwarp_warp_classes): entry 3 * item + (half + 1) = 0x80000000 | in | out << 1 | slanted << 2 | affine << 3 | pair << 4
    const uint32_t* tile_class;  // read instead of classifying (entries without the top bit are classified as usual); NULL: classify
    uint32_t* classify_out;      // fill mode: every workgroup writes its tile's verdict and leaves; NULL: warp
};

int tile_width(int dtype);   // destination pixels per row segment of one wave: 256 (8-bit) / 128 (float)
int rows_per_pass();         // rows one pass of a workgroup's waves covers
int resident_workgroups(int dtype, int channels, int interp);  // workgroups of this format's kernel the device holds at once
hipError_t launch_warp(const WarpArgs& a, int dtype, int channels, int interp, hipStream_t stream);
hipError_t launch_warp_composite(const WarpArgs& a, int channels, hipStream_t stream);  // warp_rows<..., NSRC = 3>
int composite_max_rows();    // tallest tile the composite's LDS copies hold
hipError_t launch_footprint(unsigned char* touched, int batch, int src_h, int src_w, int dst_h, int dst_w, const double* minv,
                            int m_stride, int bw0, int interp, hipStream_t stream);
hipError_t launch_project_points(const void* in, void* out, int64_t n, int dim, const double* H, int dtype, hipStream_t stream);
hipError_t launch_composite(const uint8_t* bg, const uint8_t* fg, const uint8_t* mask, uint8_t* out, int64_t n, hipStream_t stream);
hipError_t launch_rbox_iou(const void* a, int na, int a_stride, const void* b, int nb, int b_stride, void* out, int dtype,
                           hipStream_t stream);

hipError_t launch_resize_linear_u8(const uint8_t* src, uint8_t* dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels, int64_t src_fs,
                                   int64_t src_rs, int64_t dst_fs, int64_t dst_rs, hipStream_t stream);
hipError_t launch_rbox_transform(const void* in, int n, int stride, const double* H, double scale, int src_is_bev, void* out, int dtype, hipStream_t stream);
hipError_t launch_tracker_step(const void* dets, int n, int det_stride, const void* trks, int m, int trk_stride, const double* H_world_bev, double scale,
                               const double* H_img_world, double iou_threshold, void* dets_world, void* iou, unsigned char* cand, void* dets_img,
                               int dtype, hipStream_t stream);

}  // namespace bevwarp
