/*
 * bevwarp.h -- C ABI of libbevwarp.so, the MI355X (gfx950) BEV homography-warp path.
 *
 * The reference (minghanz/bev) has no FFI / plugin interface for this path: its pixel work is one
 * third-party call.  Every entry point below names the reference interface it stands in for
 * (paths relative to the reference repository root):
 *
 *   bevwarp_warp            cv2.warpPerspective(img, H_bev_img, (u_size, v_size))
 *                             vis_homo.py:89, vis_homo.py:91, bev/tool/compo.py:38,46,47
 *   bevwarp_warp_classes, bevwarp_tile_classes_bytes
 *                           the same call inside a camera loop (one H_bev_img, every frame of the video): vis_homo.py:85-91
 *   bevwarp_invert_homography  the cv::invert(M) step inside that call (M is the forward src->dst map)
 *   bevwarp_warp_planar     the same warp (8-bit or float32 frames), written as normalised float32 channel planes in the same
 *                             pass (the layout step between vis_homo.py:89 and a detector's input;
 *                             the reference leaves it to its callers)
 *   bevwarp_composite       composite_reg_img(bg, fg, fg_mask), bev/tool/compo.py:5-24 (the blend after the three warps
 *                             of composite_bev_img, :26-49)
 *   bevwarp_warp_composite  composite_bev_img(bg, fg, fg_mask, ...), bev/tool/compo.py:26-49: the three warps and the blend
 *                             in one launch, no warped image in memory
 *   bevwarp_footprint       -- measurement aid (SURVEY.md 8(d) "footprint_px"), no reference twin
 *   bevwarp_project_points  pts_world_bev(pts_src, H), bev/rbox.py:136-151; rbox_world_img, :221-226;
 *                             Calib.gen_center_in_world, bev/calib.py:135-138
 *   bevwarp_rbox_iou        iou_batch_rbox -> d3d.box.box2d_iou(.., method="rbox"),
 *                             bev/tracker/rbox_tracker.py:87-92 (call site :393-394)
 *   bevwarp_rbox_transform  rbox_world_bev(rbox_src, H, src), bev/rbox.py:173-219 (yaw conventions :20-36)
 *   bevwarp_tracker_step    one frame of bev/tool/rbox_tracking_BrnoCompSpeed.py:88-109 up to the assignment: detections
 *                             BEV -> world (rbox_world_bev), IoU against the trackers' predicted boxes (iou_batch_rbox),
 *                             the `iou > iou_threshold` gate of associate_detections_to_trackers
 *                             (bev/tracker/rbox_tracker.py:383-405) and the image-plane centres (rbox_world_img,
 *                             bev/rbox.py:221-226) -- ONE launch; the Hungarian assignment and the Kalman filters stay
 *                             on the host
 *
 * Conventions
 *   - Plain C: pointers, sizes, enums.  No torch / HIP types in signatures (`stream` is a hipStream_t
 *     passed as void*; NULL = the default stream).
 *   - All data pointers are DEVICE pointers (HIP) unless marked HOST.  The library never allocates,
 *     frees or retains caller memory; every call is asynchronous and ordered on `stream`.
 *   - Images are interleaved HWC, strides in BYTES.  Homographies are 3x3 row-major float64.
 *   - Return value: BEVWARP_OK (0) or a negative bevwarp_status; bevwarp_strerror() describes it.
 *     Nothing throws across the ABI.  There is NO CPU fallback: without a HIP device calls fail.
 *   - Thread-safe and re-entrant: no global mutable state.
 */
#ifndef BEVWARP_H
#define BEVWARP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEVWARP_ABI_VERSION 7

typedef enum bevwarp_status {
    BEVWARP_OK = 0,
    BEVWARP_ERR_BAD_ARG = -1,      /* NULL pointer, non-positive size, misaligned stride                 */
    BEVWARP_ERR_UNSUPPORTED = -2,  /* dtype / channel count / interpolation outside the supported set    */
    BEVWARP_ERR_TOO_LARGE = -3,    /* source side > 32767 px (fixed-point map range), a source row >= 16 MiB or frame >= 2 GiB */
    BEVWARP_ERR_NOT_FINITE = -4,   /* homography contains NaN / Inf                                      */
    BEVWARP_ERR_HIP = -5,          /* a HIP runtime call failed; see bevwarp_last_hip_error()            */
    BEVWARP_ERR_OVERLAP = -6       /* source and destination share bytes (ABI v5; v4 reported BAD_ARG)   */
} bevwarp_status;

typedef enum bevwarp_dtype { BEVWARP_U8 = 0, BEVWARP_F32 = 1, BEVWARP_F64 = 2 } bevwarp_dtype;

/* Same numeric values as cv2.INTER_NEAREST / cv2.INTER_LINEAR. */
typedef enum bevwarp_interp { BEVWARP_NEAREST = 0, BEVWARP_LINEAR = 1 } bevwarp_interp;

int bevwarp_version(void);
const char *bevwarp_strerror(int status);
/* Text of the last HIP error seen by the calling thread ("" if none). */
const char *bevwarp_last_hip_error(void);

/* HOST helper.  M_inv[i] = inverse of M_fwd[i] (n matrices of 9 doubles each) with the closed-form
 * cofactor / (1/det) evaluation order OpenCV uses for 3x3 doubles; a singular matrix inverts to
 * all zeros.  Callers upload M_inv to the device and hand it to bevwarp_warp. */
int bevwarp_invert_homography(const double *M_fwd /*HOST*/, double *M_inv /*HOST*/, int n);

/*
 * dst[b] = warpPerspective(src[b], M[b], (dst_w, dst_h)) for b in [0, batch), BORDER_CONSTANT.
 *
 *   src, dst       device; `channels` interleaved values of `dtype` (BEVWARP_U8 | BEVWARP_F32) per pixel.
 *                  src and dst must not overlap (an in-place warp would read taps other workgroups have already
 *                  overwritten): the call returns BEVWARP_ERR_OVERLAP when the byte ranges [src, last byte of
 *                  frame batch-1] and [dst, last byte of frame batch-1] intersect -- unless both walk their rows
 *                  with one common stride (equal row strides, frame strides multiples of it) and their rows
 *                  occupy disjoint byte columns of that stride: two ROIs of one image that lie side by side are
 *                  accepted, as cv2.warpPerspective accepts them.
 *                  "each tap outside replaced by the border value": a pixel whose four taps are ALL outside
 *                  is the border value itself (float32 too), as in OpenCV's remapBilinear.
 *   *_frame_stride bytes between consecutive frames; *_row_stride bytes between rows (>= row bytes).
 *   M_inv          device; INVERSE (dst px -> src px) matrices, float64, row-major;
 *                  m_count == batch (one per frame) or 1 (shared by all frames).
 *   interp         BEVWARP_NEAREST: (X, Y) = round-half-even((x', y') / w'); copy or border.
 *                  BEVWARP_LINEAR : coordinates quantised to 1/32 px, 4 taps, each tap outside the
 *                  source replaced by the border value; u8 in 15-bit fixed point, f32 in float.
 *   border_value   HOST, `channels` doubles, or NULL for 0.
 *   channels       1..4.   src_w, src_h <= 32767.
 */
int bevwarp_warp(const void *src, void *dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels,
                 int64_t src_frame_stride, int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_row_stride,
                 const double *M_inv, int m_count, int dtype, int interp, const double *border_value /*HOST*/,
                 void *stream);

/*
 * bevwarp_warp with the per-tile verdicts of an earlier launch (ABI v7).  Which way a tile is processed -- inside the frame, outside,
 * cut by its edge; turned, rectification-form, pair loads -- follows from the matrices, the sizes and the format alone, and deriving it
 * is a tenth of the 8-bit kernel's instructions.  A caller that warps many batches through the SAME matrices and geometry (a camera
 * loop: cv2.warpPerspective per frame with one H_bev_img, vis_homo.py:85-91) fills a table once and hands it to every later call:
 *
 *   classes   device, bevwarp_tile_classes_bytes(...) bytes, 4-byte aligned.
 *   mode      BEVWARP_CLASSES_FILL: write the verdicts, no pixel (src / dst are not accessed, but take the arguments of the warps
 *                                   the table is meant for: the table is only valid for that batch, those sizes, strides' alignment,
 *                                   format, interpolation and -- above all -- those M_inv CONTENTS);
 *             BEVWARP_CLASSES_USE : warp, reading the verdicts.  Entries that were never filled are classified as usual.
 * A table that does not belong to the matrices it is used with makes the kernel trust a wrong verdict (a tile taken for interior is
 * sampled without guards): like a wrong pointer, that is the caller's to get right.  bevwarp_warp never needs a table.
 */
#define BEVWARP_CLASSES_USE 0
#define BEVWARP_CLASSES_FILL 1
int64_t bevwarp_tile_classes_bytes(int batch, int src_h, int src_w, int dst_h, int dst_w, int channels, int dtype, int interp);
int bevwarp_warp_classes(const void *src, void *dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels,
                         int64_t src_frame_stride, int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_row_stride,
                         const double *M_inv, int m_count, int dtype, int interp, const double *border_value /*HOST*/,
                         void *classes, int mode, void *stream);

/*
 * float32 PLANAR destination, one pass:
 *   dst[b][c][y][x] = (float)warp(src[b])[y][x][c] * scale[c] + bias[c]      (float32 multiply, then add)
 * where warp is exactly what bevwarp_warp computes for `dtype` (BEVWARP_U8 | BEVWARP_F32: same interpolation, rounding
 * and border).
 *   dst             device float32; dst_plane_stride bytes between channel planes, dst_row_stride between rows,
 *                   dst_frame_stride between frames (all multiples of 4; multiples of 16 enable the wide stores).
 *   scale, bias     HOST, `channels` doubles each (converted to float32); NULL = 1 and 0.
 * Overlap of src and dst: BEVWARP_ERR_OVERLAP when the bounding byte ranges intersect.
 */
int bevwarp_warp_planar(const void *src, void *dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels,
                        int64_t src_frame_stride, int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_plane_stride,
                        int64_t dst_row_stride, const double *M_inv, int m_count, int dtype, int interp,
                        const double *border_value /*HOST*/, const double *scale /*HOST*/, const double *bias /*HOST*/,
                        void *stream);

/*
 * out[i] = uint8(min(round_half_even(fg[i] * (mask[i] / 255) + bg[i] * (1 - mask[i] / 255)), 255)) for i in [0, n), computed in
 * float64 like the reference's numpy expression.  Device uint8 arrays of n bytes each (images of equal shape, any
 * channel count, flattened); `out` may alias `bg` or `fg`.
 */
int bevwarp_composite(const void *bg, const void *fg, const void *mask, void *out, int64_t n, void *stream);

/*
 * dst = composite_reg_img(warp(bg, M_bg), warp(fg, M_cam), warp(mask, M_cam)) -- bev/tool/compo.py:26-49 -- where the three
 * warps are bevwarp_warp's BEVWARP_U8 / BEVWARP_LINEAR / zero-border warp to (dst_w, dst_h) and the blend is
 * bevwarp_composite's, computed per pixel in one launch: the warped images are never written.  Bit-identical to the
 * three-warp sequence.
 *   bg (bg_h x bg_w), fg and mask (fg_h x fg_w each), dst: device uint8, `channels` (1..4) interleaved, strides in bytes.
 *   M_inv_bg, M_inv_cam: device, 9 float64 each, INVERSE maps (dst px -> bg px / camera px).
 *   fg_gray: non-zero = the reference's bw_mode (compo.py:13-14): the foreground is converted BGR -> grey -> BGR before it is
 *            warped, i.e. tap by tap ((1868 B + 9617 G + 4899 R + 8192) >> 14, OpenCV's 8-bit form); channels must be 3.
 * Overlap of dst with a source: BEVWARP_ERR_OVERLAP (same rule as bevwarp_warp).
 */
int bevwarp_warp_composite(const void *bg, int bg_h, int bg_w, int64_t bg_row_stride, const void *fg, const void *mask,
                           int fg_h, int fg_w, int64_t fg_row_stride, int64_t mask_row_stride, void *dst, int dst_h,
                           int dst_w, int64_t dst_row_stride, int channels, const double *M_inv_bg,
                           const double *M_inv_cam, int fg_gray, void *stream);

/*
 * dst[b] = cv2.resize(src[b], (dst_w, dst_h)) with the default INTER_LINEAR, uint8, `channels` (1..4) interleaved -- the resize of the
 * reference's "small" branch (img_small = cv2.resize(img, (new_u, new_v)), vis_homo.py:90) in front of its warp (:91).  OpenCV's classic
 * bilinear path: sampling at (d + 0.5) * scale - 0.5, 11-bit coefficients, replicated edge, an exact 2 x 2 decimation = the box mean
 * (restated from memory of resize.cpp; parity unpinned -- oracle/resize_oracle.c).  dtype must be BEVWARP_U8, interp BEVWARP_LINEAR.
 * Device pointers, strides in bytes; src and dst must not overlap (BEVWARP_ERR_OVERLAP).
 */
int bevwarp_resize(const void *src, void *dst, int batch, int src_h, int src_w, int dst_h, int dst_w, int channels,
                   int64_t src_frame_stride, int64_t src_row_stride, int64_t dst_frame_stride, int64_t dst_row_stride, int dtype,
                   int interp, void *stream);

/*
 * Marks every in-bounds source pixel that any tap of any destination pixel of the same warp would
 * read: touched[b][y][x] = 1 (device bytes, batch x src_h x src_w, caller zero-initialises).
 * sum(touched) is the exact "footprint_px" term of the algorithmic-bytes figure.
 */
int bevwarp_footprint(unsigned char *touched, int batch, int src_h, int src_w, int dst_h, int dst_w,
                      const double *M_inv, int m_count, int interp, void *stream);

/*
 * out[i] = dehomogenise(H @ (in[i], 1))            for dim == 2   (n x 2 in, n x 2 out)
 * out[i] = (H @ in[i]) / (H @ in[i])[2]            for dim == 3   (n x 3 in, n x 3 out)
 *   in, out  device, contiguous, dtype BEVWARP_F32 | BEVWARP_F64 (arithmetic is float64 either way).
 *   H        HOST, 9 doubles (copied at call time).  in == out is allowed.
 */
int bevwarp_project_points(const void *in, void *out, int64_t n, int dim, const double *H /*HOST*/, int dtype,
                           void *stream);

/*
 * out[i][j] = IoU of rotated rectangles a[i], b[j].  Rows are [x, y, w, h, yaw, ...] with
 * `a_stride` / `b_stride` values per row (>= 5); at yaw 0 the length h lies along +x and the width w
 * along y (bev/rbox.py:87-95, the "world" convention).  out is na x nb, same dtype.
 *   dtype BEVWARP_F32 | BEVWARP_F64 (arithmetic is float64).
 */
int bevwarp_rbox_iou(const void *a, int na, int a_stride, const void *b, int nb, int b_stride, void *out, int dtype,
                     void *stream);

/*
 * out[i] = rbox_world_bev(boxes[i], H, src): rows [x, y, w, h, yaw, ...] (`stride` values per row, >= 5) through the
 * similarity H (HOST, 9 doubles; normalised by H[8]; BEVWARP_ERR_BAD_ARG when its last row is not (0, 0, 1) to 1e-5 or its
 * axes scale differently, the two conditions the reference asserts).  src_is_bev != 0: rows are BEV boxes (yaw from the
 * v axis), out rows are world boxes (yaw from the x axis); 0: the other way round.  out is n x 5, same dtype
 * (BEVWARP_F32 | BEVWARP_F64, arithmetic float64).
 */
int bevwarp_rbox_transform(const void *boxes, int n, int stride, const double *H /*HOST*/, int src_is_bev, void *out,
                           int dtype, void *stream);

/*
 * One tracker step, one launch:
 *   dets_world[i]     = rbox_world_bev(dets_bev[i], H_world_bev, "bev")                         n x 5
 *   iou[i][j]         = IoU(dets_world[i], trks_world[j])         (as bevwarp_rbox_iou)         n x m
 *   candidates[i][j]  = iou[i][j] > iou_threshold                 (uint8 0 / 1)                 n x m
 *   dets_img[i]       = dehomogenise(H_img_world @ (dets_world[i].xy, 1))   when H_img_world    n x 2
 * dets_bev rows have det_stride >= 5 values, trks_world rows trk_stride >= 5 (a tracker's state row may carry more).
 * H_world_bev as in bevwarp_rbox_transform; H_img_world (HOST, 9 doubles) may be NULL (then dets_img is not touched).
 * m == 0 is allowed (only dets_world / dets_img are produced).  n <= 65535.
 */
int bevwarp_tracker_step(const void *dets_bev, int n, int det_stride, const void *trks_world, int m, int trk_stride,
                         const double *H_world_bev /*HOST*/, const double *H_img_world /*HOST, may be NULL*/,
                         double iou_threshold, void *dets_world, void *iou, unsigned char *candidates, void *dets_img,
                         int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVWARP_H */
