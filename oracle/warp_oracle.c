/*
 * oracle/warp_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT the product path.
 *
 * CPU restatement (plain C99, scalar) of the pixel path the reference reaches
 * through third-party OpenCV:
 *
 *     bev = cv2.warpPerspective(img, H_bev_img, (bspec.u_size, bspec.v_size))
 *         /root/reference/vis_homo.py:89, :91
 *         /root/reference/bev/tool/compo.py:38, :46, :47
 *
 * plus the point projection of /root/reference/bev/rbox.py:136-151
 * (pts_world_bev) and the rotated-box IoU the tracker obtains from d3d at
 * /root/reference/bev/tracker/rbox_tracker.py:87-92.
 *
 * PARITY STATUS: **parity unpinned** for the warp and the IoU.  The arithmetic
 * lives in un-vendored, un-pinned third-party packages (opencv-python; d3d)
 * that are absent from /root/reference and from this image, and the reference
 * holds no test, fixture or golden image for them.  The warp below restates
 * the published algorithm of OpenCV 3.x-4.x `modules/imgproc/src/imgwarp.cpp`
 * (cv::warpPerspective -> cv::invert 3x3 -> WarpPerspectiveInvoker -> cv::remap
 * with CV_16SC2 + CV_16UC1 fixed-point maps; INTER_BITS = 5,
 * INTER_REMAP_COEF_BITS = 15), the classic scalar code path -- NOT the IPP or
 * OpenCL dispatches and not the OpenCV 5 rewrite, which may differ in last
 * bits.  "Bit-exact" anywhere in this repository means: equal to THIS
 * algorithm, not to whatever cv2 build a user has installed.  Every rule here
 * -- including remapBilinear's three BORDER_CONSTANT paths (fully inside,
 * fully outside = cval stored directly, straddling = per-tap substitution) --
 * is restated from memory of that source file; none could be checked against
 * OpenCV in this image.  It is pinned
 * by the analytic known-answer tests in tests/test_oracle_warp.py and by an
 * independent numpy twin (oracle/warp_numpy.py).  pts_world_bev IS pinned:
 * tests/golden/reference_vectors.json holds outputs of the reference itself.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  bev_amd/ never does.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: the coordinate chain must round after every
 * multiply and add, exactly as a non-FMA x86-64 build of OpenCV does.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define INTER_BITS 5
#define INTER_TAB_SIZE (1 << INTER_BITS)
#define INTER_REMAP_COEF_BITS 15
#define INTER_REMAP_COEF_SCALE (1 << INTER_REMAP_COEF_BITS)

enum { ORACLE_U8 = 0, ORACLE_F32 = 1 };
enum { ORACLE_NEAREST = 0, ORACLE_LINEAR = 1 };

int oracle_version(void) { return 1; }

/* cv::invert, 3x3 CV_64F, DECOMP_LU -> closed-form cofactor path
 * (core/src/lapack.cpp, `n == 3` branch): det3 then t[i] = cofactor * (1/det). */
int oracle_invert3x3(const double *S, double *D)
{
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.0) {
        memset(D, 0, 9 * sizeof(double)); /* cv::invert zero-fills dst when singular */
        return 0;
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
    return 1;
}

/* cv::saturate_cast<int>(double) == cvRound == lrint under round-to-nearest-even,
 * applied after the explicit clamp to [INT_MIN, INT_MAX] the invoker performs with
 * std::max / std::min ((b < a) ? b : a semantics, so a NaN collapses to INT_MAX). */
static inline int round_clamped(double v)
{
    double hi = (double)INT_MAX, lo = (double)INT_MIN;
    double m = (v < hi) ? v : hi;  /* std::min(hi, v) */
    double c = (lo < m) ? m : lo;  /* std::max(lo, m) */
    return (int)nearbyint(c);      /* default rounding mode: ties to even */
}

static inline short sat_short(int v)
{
    return (short)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v));
}

static inline unsigned char sat_u8_from_double(double v)
{
    int iv = (int)nearbyint(v);
    return (unsigned char)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

/* Block shape of WarpPerspectiveInvoker: BLOCK_SZ = 32 ->
 * bh0 = min(16, h); bw0 = min(1024 / bh0, w); bh0 = min(1024 / bw0, h).
 * Only bw0 influences values (X0/Y0/W0 are evaluated at the block's left edge). */
int oracle_block_width(int dst_w, int dst_h)
{
    const int BLOCK_SZ = 32;
    int bh0 = BLOCK_SZ / 2 < dst_h ? BLOCK_SZ / 2 : dst_h;
    int bw0 = BLOCK_SZ * BLOCK_SZ / bh0 < dst_w ? BLOCK_SZ * BLOCK_SZ / bh0 : dst_w;
    return bw0;
}

/* Fixed-point source coordinate of dst pixel (x, y): the invoker's inner loop.
 * M is the INVERSE (dst -> src) matrix.  nearest: sx, sy are whole pixels, alpha 0.
 * linear: sx, sy = X >> 5 and alpha = (Y & 31) * 32 + (X & 31). */
static inline void map_pixel(const double *M, int bw0, int x, int y, int interp, short *sx, short *sy, int *alpha)
{
    int bx = (x / bw0) * bw0;
    int x1 = x - bx;
    double X0 = M[0] * bx + M[1] * y + M[2];
    double Y0 = M[3] * bx + M[4] * y + M[5];
    double W0 = M[6] * bx + M[7] * y + M[8];
    double W = W0 + M[6] * x1;
    if (interp == ORACLE_NEAREST) {
        W = W ? 1. / W : 0;
        int X = round_clamped((X0 + M[0] * x1) * W);
        int Y = round_clamped((Y0 + M[3] * x1) * W);
        *sx = sat_short(X);
        *sy = sat_short(Y);
        *alpha = 0;
    } else {
        W = W ? INTER_TAB_SIZE / W : 0;
        int X = round_clamped((X0 + M[0] * x1) * W);
        int Y = round_clamped((Y0 + M[3] * x1) * W);
        *sx = sat_short(X >> INTER_BITS);
        *sy = sat_short(Y >> INTER_BITS);
        *alpha = (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1));
    }
}

/* BilinearTab_f / BilinearTab_i as initInterTab2D(INTER_LINEAR) leaves them.
 * 1-D taps are {1.f - i/32.f, i/32.f}; 2-D float weights are their float products
 * (exact: multiples of 1/1024).  Integer weights = saturate_cast<short>(v * 32768);
 * every entry sums to 32768 except (0,0) where 32768 saturates to 32767 and the
 * fix-up adds the missing 1 to tap 3, giving {32767, 0, 0, 1}. */
static float tab_f[INTER_TAB_SIZE * INTER_TAB_SIZE][4];
static short tab_i[INTER_TAB_SIZE * INTER_TAB_SIZE][4];
static int tab_ready = 0;

static void init_tabs(void)
{
    if (tab_ready) return;
    const float scale = 1.f / INTER_TAB_SIZE;
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++) {
            float ty[2] = {1.f - i * scale, i * scale};
            float tx[2] = {1.f - j * scale, j * scale};
            int isum = 0;
            for (int k1 = 0; k1 < 2; k1++)
                for (int k2 = 0; k2 < 2; k2++) {
                    float v = ty[k1] * tx[k2];
                    tab_f[i * INTER_TAB_SIZE + j][k1 * 2 + k2] = v;
                    int iv = (int)lrintf(v * INTER_REMAP_COEF_SCALE);
                    short s = sat_short(iv);
                    tab_i[i * INTER_TAB_SIZE + j][k1 * 2 + k2] = s;
                    isum += s;
                }
            if (isum != INTER_REMAP_COEF_SCALE) /* only (0,0): 32767 -> tap 3 takes the +1 */
                tab_i[i * INTER_TAB_SIZE + j][3] = (short)(tab_i[i * INTER_TAB_SIZE + j][3] - (isum - INTER_REMAP_COEF_SCALE));
        }
    tab_ready = 1;
}

void oracle_bilinear_tab_i(short *out /* 1024*4 */)
{
    init_tabs();
    memcpy(out, tab_i, sizeof(tab_i));
}

static void warp_rows(const unsigned char *src, int sh, int sw, int64_t sstep, unsigned char *dst, int dw,
                      int64_t dstep, int cn, const double *M, int bw0, int dtype, int interp, const double *bval,
                      int y_begin, int y_end, unsigned char *touched)
{
    unsigned char cval_u8[4] = {0, 0, 0, 0};
    float cval_f[4] = {0, 0, 0, 0};
    for (int k = 0; k < cn; k++) {
        double b = bval ? bval[k] : 0.0;
        cval_u8[k] = sat_u8_from_double(b);
        cval_f[k] = (float)b;
    }
    const int esz = dtype == ORACLE_U8 ? 1 : 4;
    for (int y = y_begin; y < y_end; y++) {
        unsigned char *drow = dst ? dst + (int64_t)y * dstep : NULL;
        for (int x = 0; x < dw; x++) {
            short sx, sy;
            int alpha;
            map_pixel(M, bw0, x, y, interp, &sx, &sy, &alpha);
            if (interp == ORACLE_NEAREST) {
                /* remapNearest, BORDER_CONSTANT */
                int inside = (unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh;
                if (touched && inside) touched[(int64_t)sy * sw + sx] = 1;
                if (!drow) continue;
                unsigned char *D = drow + (int64_t)x * cn * esz;
                if (inside)
                    memcpy(D, src + (int64_t)sy * sstep + (int64_t)sx * cn * esz, (size_t)cn * esz);
                else if (dtype == ORACLE_U8)
                    memcpy(D, cval_u8, (size_t)cn);
                else
                    memcpy(D, cval_f, (size_t)cn * 4);
                continue;
            }
            /* remapBilinear, BORDER_CONSTANT, three code paths of the original:
             *   fully inside   -> the plain 4-tap sum;
             *   fully outside  -> `sx >= ssize.width || sx + 1 < 0 || sy >= ssize.height || sy + 1 < 0`
             *                     stores cval[k] DIRECTLY (no blend: for float32 a 4-term sum of
             *                     cval * w would differ from cval by an ulp for most border values);
             *   straddling     -> each of the 4 taps individually replaced by the border value.
             * The first and third coincide; the second is handled here.  (Restated from memory of
             * imgwarp.cpp like the rest of this file: parity unpinned.) */
            if (drow && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
                unsigned char *D = drow + (int64_t)x * cn * esz;
                if (dtype == ORACLE_U8)
                    memcpy(D, cval_u8, (size_t)cn);
                else
                    memcpy(D, cval_f, (size_t)cn * 4);
                continue;
            }
            int in00 = sx >= 0 && sy >= 0 && sx < sw && sy < sh;
            int in01 = sx + 1 >= 0 && sy >= 0 && sx + 1 < sw && sy < sh;
            int in10 = sx >= 0 && sy + 1 >= 0 && sx < sw && sy + 1 < sh;
            int in11 = sx + 1 >= 0 && sy + 1 >= 0 && sx + 1 < sw && sy + 1 < sh;
            if (touched) {
                if (in00) touched[(int64_t)sy * sw + sx] = 1;
                if (in01) touched[(int64_t)sy * sw + sx + 1] = 1;
                if (in10) touched[(int64_t)(sy + 1) * sw + sx] = 1;
                if (in11) touched[(int64_t)(sy + 1) * sw + sx + 1] = 1;
            }
            if (!drow) continue;
            const unsigned char *S = src + (int64_t)sy * sstep + (int64_t)sx * cn * esz;
            if (dtype == ORACLE_U8) {
                const short *w = tab_i[alpha];
                unsigned char *D = drow + (int64_t)x * cn;
                for (int k = 0; k < cn; k++) {
                    int v0 = in00 ? S[k] : cval_u8[k];
                    int v1 = in01 ? S[k + cn] : cval_u8[k];
                    int v2 = in10 ? S[sstep + k] : cval_u8[k];
                    int v3 = in11 ? S[sstep + cn + k] : cval_u8[k];
                    /* FixedPtCast<int, uchar, 15> */
                    int r = (v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3] + (1 << (INTER_REMAP_COEF_BITS - 1))) >>
                            INTER_REMAP_COEF_BITS;
                    D[k] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
                }
            } else {
                const float *w = tab_f[alpha];
                const float *Sf = (const float *)S;
                const int64_t fstep = sstep / 4;
                float *D = (float *)(drow + (int64_t)x * cn * 4);
                for (int k = 0; k < cn; k++) {
                    float v0 = in00 ? Sf[k] : cval_f[k];
                    float v1 = in01 ? Sf[k + cn] : cval_f[k];
                    float v2 = in10 ? Sf[fstep + k] : cval_f[k];
                    float v3 = in11 ? Sf[fstep + cn + k] : cval_f[k];
                    D[k] = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
                }
            }
        }
    }
}

/*
 * dst = warpPerspective(src, M, (dst_w, dst_h)), BORDER_CONSTANT.
 *   M            3x3 row-major doubles; forward (src->dst) unless m_is_inverse.
 *   dtype        ORACLE_U8 / ORACLE_F32 (pixels), cn channels interleaved (HWC).
 *   strides      in BYTES.
 *   border_value cn doubles or NULL (= 0).
 *   nthreads     OpenMP threads over row stripes (values do not depend on it).
 *   touched      optional sh*sw byte map: set to 1 for every in-bounds source pixel
 *                referenced by any tap (the "footprint" of SURVEY.md §8(d)); dst may
 *                be NULL when only the footprint is wanted.
 * Returns 0, or -1 on bad arguments.
 */
int oracle_warp_perspective(const void *src, int src_h, int src_w, int64_t src_row_stride, void *dst, int dst_h,
                            int dst_w, int64_t dst_row_stride, int channels, const double *M, int m_is_inverse,
                            int dtype, int interp, const double *border_value, int nthreads, unsigned char *touched)
{
    if (!src || !M || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) return -1;
    if (channels < 1 || channels > 4) return -1;
    if (dtype != ORACLE_U8 && dtype != ORACLE_F32) return -1;
    if (interp != ORACLE_NEAREST && interp != ORACLE_LINEAR) return -1;
    init_tabs();
    double Mi[9];
    if (m_is_inverse)
        memcpy(Mi, M, sizeof(Mi));
    else
        oracle_invert3x3(M, Mi);
    const int bw0 = oracle_block_width(dst_w, dst_h);
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
    if (nthreads > 1 && !touched) {
        const int stripe = 16;
        const int nstripes = (dst_h + stripe - 1) / stripe;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (int s = 0; s < nstripes; s++) {
            int y0 = s * stripe, y1 = y0 + stripe < dst_h ? y0 + stripe : dst_h;
            warp_rows((const unsigned char *)src, src_h, src_w, src_row_stride, (unsigned char *)dst, dst_w,
                      dst_row_stride, channels, Mi, bw0, dtype, interp, border_value, y0, y1, NULL);
        }
        return 0;
    }
#endif
    warp_rows((const unsigned char *)src, src_h, src_w, src_row_stride, (unsigned char *)dst, dst_w, dst_row_stride,
              channels, Mi, bw0, dtype, interp, border_value, 0, dst_h, touched);
    return 0;
}

/* Fixed-point maps alone (for tests): sxy = dst_h*dst_w*2 shorts, alpha = dst_h*dst_w ints. */
int oracle_warp_maps(int dst_h, int dst_w, const double *M_inv, int interp, short *sxy, int *alpha)
{
    const int bw0 = oracle_block_width(dst_w, dst_h);
    for (int y = 0; y < dst_h; y++)
        for (int x = 0; x < dst_w; x++) {
            short sx, sy;
            int a;
            map_pixel(M_inv, bw0, x, y, interp, &sx, &sy, &a);
            sxy[((int64_t)y * dst_w + x) * 2] = sx;
            sxy[((int64_t)y * dst_w + x) * 2 + 1] = sy;
            if (alpha) alpha[(int64_t)y * dst_w + x] = a;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * pts_world_bev (/root/reference/bev/rbox.py:136-151): N x 2 -> append 1 -> H . p ->
 * divide by the third row -> N x 2; N x 3 in keeps the homogeneous column (== 1 out).
 * numpy evaluates H.dot(pts.T) through BLAS; the summation order of the three terms
 * is not observable beyond 1 ulp, tests use a relative tolerance of 1e-12 (f64).
 * ---------------------------------------------------------------------------------- */
int oracle_project_points_f64(const double *in, double *out, int64_t n, int dim, const double *H)
{
    if (dim != 2 && dim != 3) return -1;
    for (int64_t i = 0; i < n; i++) {
        double x = in[i * dim], y = in[i * dim + 1], w = dim == 3 ? in[i * dim + 2] : 1.0;
        double X = H[0] * x + H[1] * y + H[2] * w;
        double Y = H[3] * x + H[4] * y + H[5] * w;
        double Z = H[6] * x + H[7] * y + H[8] * w;
        out[i * dim] = X / Z;
        out[i * dim + 1] = Y / Z;
        if (dim == 3) out[i * dim + 2] = Z / Z;
    }
    return 0;
}

int oracle_project_points_f32(const float *in, float *out, int64_t n, int dim, const double *H)
{
    if (dim != 2 && dim != 3) return -1;
    for (int64_t i = 0; i < n; i++) {
        double x = in[i * dim], y = in[i * dim + 1], w = dim == 3 ? in[i * dim + 2] : 1.0;
        double X = H[0] * x + H[1] * y + H[2] * w;
        double Y = H[3] * x + H[4] * y + H[5] * w;
        double Z = H[6] * x + H[7] * y + H[8] * w;
        out[i * dim] = (float)(X / Z);
        out[i * dim + 1] = (float)(Y / Z);
        if (dim == 3) out[i * dim + 2] = (float)(Z / Z);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Rotated-rectangle IoU.  The reference calls d3d.box.box2d_iou(a, b, method="rbox")
 * after adding pi/2 to both yaws (/root/reference/bev/tracker/rbox_tracker.py:87-92);
 * d3d is absent, so this follows the reference's OWN rectangle convention instead:
 * corners = xywhr2xyxy(box, "world") (/root/reference/bev/rbox.py:87-95,101-106): at
 * yaw 0 the length h lies along +x and the width w along y.  IoU is invariant to the
 * common +pi/2 the tracker applies under that reading.  **parity unpinned.**
 * Intersection by Sutherland-Hodgman clipping of quad A against the 4 half-planes of
 * quad B (both counter-clockwise), area by the shoelace formula.
 * ---------------------------------------------------------------------------------- */
static void rbox_corners(const double *b, double q[4][2])
{
    const double hx = b[3] * 0.5, hy = b[2] * 0.5, c = cos(b[4]), s = sin(b[4]);
    const double lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy}; /* CCW */
    for (int i = 0; i < 4; i++) {
        q[i][0] = c * lx[i] - s * ly[i] + b[0];
        q[i][1] = s * lx[i] + c * ly[i] + b[1];
    }
}

static double poly_area(double p[][2], int n)
{
    double a = 0;
    for (int i = 0; i < n; i++) {
        int j = (i + 1) % n;
        a += p[i][0] * p[j][1] - p[j][0] * p[i][1];
    }
    return 0.5 * a;
}

double oracle_rbox_iou_pair(const double *a, const double *b)
{
    double qa[4][2], qb[4][2];
    rbox_corners(a, qa);
    rbox_corners(b, qb);
    double poly[16][2], tmp[16][2];
    int n = 4;
    memcpy(poly, qa, sizeof(qa));
    for (int e = 0; e < 4 && n > 0; e++) {
        const double ex = qb[(e + 1) % 4][0] - qb[e][0], ey = qb[(e + 1) % 4][1] - qb[e][1];
        int m = 0;
        for (int i = 0; i < n; i++) {
            const int j = (i + 1) % n;
            const double di = ex * (poly[i][1] - qb[e][1]) - ey * (poly[i][0] - qb[e][0]);
            const double dj = ex * (poly[j][1] - qb[e][1]) - ey * (poly[j][0] - qb[e][0]);
            if (di >= 0) {
                tmp[m][0] = poly[i][0];
                tmp[m][1] = poly[i][1];
                m++;
            }
            if ((di >= 0) != (dj >= 0)) {
                const double t = di / (di - dj);
                tmp[m][0] = poly[i][0] + t * (poly[j][0] - poly[i][0]);
                tmp[m][1] = poly[i][1] + t * (poly[j][1] - poly[i][1]);
                m++;
            }
        }
        n = m;
        memcpy(poly, tmp, sizeof(double) * 2 * (size_t)n);
    }
    double inter = n >= 3 ? fabs(poly_area(poly, n)) : 0.0;
    const double ua = fabs(a[2] * a[3]) + fabs(b[2] * b[3]) - inter;
    return ua > 0 ? inter / ua : 0.0;
}

/* a: na x stride_a doubles (first 5 = x,y,w,h,yaw), b likewise; out: na x nb */
int oracle_rbox_iou(const double *a, int na, int stride_a, const double *b, int nb, int stride_b, double *out)
{
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) out[(int64_t)i * nb + j] = oracle_rbox_iou_pair(a + (int64_t)i * stride_a, b + (int64_t)j * stride_b);
    return 0;
}
