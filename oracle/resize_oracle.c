/*
 * oracle/resize_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT the product path.
 *
 * CPU restatement (plain C99, scalar) of the resize the reference's "small" branch applies to a frame before it warps it:
 *
 *     img_small = cv2.resize(img, (new_u, new_v))            /root/reference/vis_homo.py:90   (default: INTER_LINEAR)
 *
 * PARITY STATUS: **parity unpinned**.  The arithmetic lives in OpenCV (un-vendored, un-pinned, absent from this image); the
 * reference holds no fixture for it.  What follows restates, from memory of OpenCV 3.x-4.x modules/imgproc/src/resize.cpp, the
 * classic (non-IPP, non-OpenCL) bilinear path for 8-bit images:
 *
 *   cv::resize:        scale_x = 1. / ((double)dst_w / src_w), scale_y likewise (NOT src_w / dst_w: two roundings);
 *                      INTER_LINEAR with an exact 2 x 2 integer scale is switched to INTER_AREA (the 2 x 2 box mean, rounded
 *                      (sum + 2) >> 2), because the two coincide up to rounding
 *   resizeGeneric_:    per destination column  fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = floor(fx);  fx -= sx;
 *                      sx < 0 -> sx = 0, fx = 0;   sx >= src_w - 1 -> sx = src_w - 1, fx = 0 (the column reads ONE tap);
 *                      coefficients (1 - fx, fx) in float, times INTER_RESIZE_COEF_SCALE = 2048, saturate_cast<short> (round half
 *                      to even);  rows alike, except that the row indices sy, sy + 1 are CLIPPED to [0, src_h - 1] and the
 *                      coefficients are left as computed
 *   HResizeLinear:     h = S[sx] * a0 + S[sx + 1] * a1                       (int; columns at or beyond the first clamped one: S[sx] * 2048)
 *   VResizeLinear<uchar>:  dst = (uchar)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2)
 *
 * Pinned by analytic known answers (identity, constant images, integer magnification at known phases, the 2 x 2 case, edge
 * replication) and by an independently written numpy twin (oracle/resize_numpy.py): tests/test_oracle_resize.py.
 *
 * Build: see oracle/Makefile (-ffp-contract=off: (dx + 0.5) * scale - 0.5 must round after the multiply).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define COEF_BITS 11
#define COEF_SCALE (1 << COEF_BITS)

static short sat_short_f(float v)
{
    const long r = lrintf(v); /* round half to even (default rounding mode) */
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

/* offsets and coefficients of one axis: ofs[d] = first tap, coef[2 d .. 2 d + 1]; returns dmax = first destination index whose
 * taps were clamped at the far end (reads one tap) */
static int axis_tables(int src_n, int dst_n, int *ofs, short *coef, int clamp_far)
{
    const double inv_scale = (double)dst_n / src_n, scale = 1.0 / inv_scale;
    int dmax = dst_n;
    for (int d = 0; d < dst_n; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (clamp_far) { /* columns: clamp the index AND zero the fraction; rows keep their fraction and clip the row index later */
            if (s < 0) {
                s = 0;
                f = 0.f;
            }
            if (s + 1 >= src_n) {
                if (d < dmax) dmax = d;
                if (s >= src_n - 1) {
                    s = src_n - 1;
                    f = 0.f;
                }
            }
        }
        ofs[d] = s;
        coef[2 * d] = sat_short_f((1.f - f) * COEF_SCALE);
        coef[2 * d + 1] = sat_short_f(f * COEF_SCALE);
    }
    return dmax;
}

static int clip_row(int y, int n) { return y < 0 ? 0 : (y < n ? y : n - 1); }

int oracle_resize_linear_u8(const unsigned char *src, int src_h, int src_w, int64_t src_row_stride, unsigned char *dst, int dst_h, int dst_w,
                            int64_t dst_row_stride, int channels)
{
    if (!src || !dst || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || channels < 1 || channels > 4) return -1;
    const int cn = channels;
    {   /* the exact 2 x 2 decimation is INTER_AREA's box mean */
        const double sx = 1.0 / ((double)dst_w / src_w), sy = 1.0 / ((double)dst_h / src_h);
        if (fabs(sx - 2.0) < 2.220446049250313e-16 && fabs(sy - 2.0) < 2.220446049250313e-16) {
            for (int y = 0; y < dst_h; y++) {
                const unsigned char *r0 = src + (int64_t)(2 * y) * src_row_stride, *r1 = r0 + src_row_stride;
                unsigned char *d = dst + (int64_t)y * dst_row_stride;
                for (int x = 0; x < dst_w; x++)
                    for (int k = 0; k < cn; k++)
                        d[x * cn + k] = (unsigned char)((r0[2 * x * cn + k] + r0[(2 * x + 1) * cn + k] + r1[2 * x * cn + k] + r1[(2 * x + 1) * cn + k] + 2) >> 2);
            }
            return 0;
        }
    }
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dst_w), *yofs = (int *)malloc(sizeof(int) * (size_t)dst_h);
    short *alpha = (short *)malloc(sizeof(short) * 2 * (size_t)dst_w), *beta = (short *)malloc(sizeof(short) * 2 * (size_t)dst_h);
    int *h0 = (int *)malloc(sizeof(int) * (size_t)dst_w * cn), *h1 = (int *)malloc(sizeof(int) * (size_t)dst_w * cn);
    if (!xofs || !yofs || !alpha || !beta || !h0 || !h1) return -2;
    const int xmax = axis_tables(src_w, dst_w, xofs, alpha, 1);
    axis_tables(src_h, dst_h, yofs, beta, 0);
    for (int y = 0; y < dst_h; y++) {
        const unsigned char *rows[2] = {src + (int64_t)clip_row(yofs[y], src_h) * src_row_stride, src + (int64_t)clip_row(yofs[y] + 1, src_h) * src_row_stride};
        int *h[2] = {h0, h1};
        for (int r = 0; r < 2; r++)
            for (int x = 0; x < dst_w; x++)
                for (int k = 0; k < cn; k++) {
                    const unsigned char *S = rows[r] + xofs[x] * cn + k;
                    h[r][x * cn + k] = x < xmax ? S[0] * alpha[2 * x] + S[cn] * alpha[2 * x + 1] : S[0] * COEF_SCALE;
                }
        const int b0 = beta[2 * y], b1 = beta[2 * y + 1];
        unsigned char *d = dst + (int64_t)y * dst_row_stride;
        for (int i = 0; i < dst_w * cn; i++) d[i] = (unsigned char)((((b0 * (h0[i] >> 4)) >> 16) + ((b1 * (h1[i] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs), free(yofs), free(alpha), free(beta), free(h0), free(h1);
    return 0;
}
