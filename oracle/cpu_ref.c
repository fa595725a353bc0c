/*
 * oracle/cpu_ref.c — CPU restatement of the reference's per-frame pixel chain.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under image_processing_amd/ (the product) may
 * include, link, dlopen or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the reported baseline.
 *
 * PARITY UNPINNED.  The reference (/root/reference) holds no pixel arithmetic of its
 * own: every stage on the hot path is a call into OpenCV 3.x, which is an un-vendored,
 * un-pinned dependency (image_processing/CMakeLists.txt:26 `find_package(OpenCV REQUIRED)`,
 * inferred 3.3 <= v <= 3.4.x) that is absent from this image, and the
 * reference has no tests, fixtures or golden images.  This file therefore restates
 * OpenCV's *published* 8-bit fixed-point algorithms (modules/imgproc color_hsv.cpp,
 * smooth / fixedpoint filters, morph.cpp, moments.cpp, demosaicing.cpp and
 * modules/core arithm.cpp::inRange) as specified in SURVEY.md Appendix A, and anchors
 * the call sequence and parameters on the reference's own call sites:
 *     objectPolygonPath.cpp:182-190   (GaussianBlur 3x3, cvtColor BGR2HSV, GaussianBlur 5x5,
 *                                      inRange, morphOps, GRAY2BGR + bitwise_and)
 *     objectPolygonPath.cpp:414-433   (morphOps: E3x4, D8x2, E3x2, D8x2, E3x4)
 *     objectPolygonPath.cpp:31-32     (RECT 3x3 / RECT 8x8 structuring elements)
 *     pose_estimate.cpp:203-206       (cvtColor BGR2HSV, inRange on the ROI crop)
 *     pose_estimate.cpp:405-415       (morphOps: E3x2, D8x2, E3x2)
 *     objectPolygonPath.cpp:449-452   (moments(...).m00 area gate)
 * It is checked against the hand-derived known-answer vectors of SURVEY.md §A.8
 * (tests/golden/kat_vectors.json) and against an independent numpy/scipy restatement
 * (tests/np_ref.py); neither is an output of a real OpenCV.
 *
 * Every function is a scalar, definition-first loop: obviously correct beats fast.
 * The one "fast" function (orc_chain_*) exists so that bench.py can time a CPU port
 * of the whole chain; it calls the same per-stage functions.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef uint8_t u8;

/* ------------------------------------------------------------------------------------
 * A.1  cvtColor(BGR2HSV), 8-bit, hue range 180 (OpenCV RGB2HSV_b).
 * Call sites: objectPolygonPath.cpp:183, pose_estimate.cpp:203.
 * bidx = 0 for BGR input, 2 for RGB input (cv_bridge toCvCopy(...,BGR8) on an rgb8
 * message swaps channels first: pose_estimate.cpp:180; folding the swap here is
 * arithmetically the same thing).
 * ---------------------------------------------------------------------------------- */
static int g_sdiv[256], g_hdiv180[256];
static int g_tab_ready = 0;

static void orc_init_tables(void)
{
    if (g_tab_ready) return;
    g_sdiv[0] = g_hdiv180[0] = 0;
    for (int i = 1; i < 256; i++) {
        /* saturate_cast<int>(double) == cvRound == round-half-to-even (lrint) */
        g_sdiv[i]    = (int)lrint((255 << 12) / (1.0 * i));
        g_hdiv180[i] = (int)lrint((180 << 12) / (6.0 * i));
    }
    g_tab_ready = 1;
}

ORC_API void orc_hsv_tables(int *sdiv, int *hdiv180)
{
    orc_init_tables();
    memcpy(sdiv, g_sdiv, sizeof g_sdiv);
    memcpy(hdiv180, g_hdiv180, sizeof g_hdiv180);
}

static inline void hsv_px(const u8 *s, u8 *d, int bidx)
{
    int b = s[bidx], g = s[1], r = s[bidx ^ 2];
    int v = b, vmin = b;
    if (g > v) v = g;
    if (r > v) v = r;
    if (g < vmin) vmin = g;
    if (r < vmin) vmin = r;
    int diff = v - vmin;
    int vr = (v == r) ? -1 : 0;
    int vg = (v == g) ? -1 : 0;
    int sat = (diff * g_sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * g_hdiv180[diff] + (1 << 11)) >> 12; /* arithmetic shift: floor for negatives */
    if (h < 0) h += 180;
    d[0] = (u8)(h < 0 ? 0 : (h > 255 ? 255 : h));
    d[1] = (u8)sat;
    d[2] = (u8)v;
}

ORC_API void orc_bgr2hsv(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int bidx)
{
    orc_init_tables();
    for (int y = 0; y < h; y++) {
        const u8 *s = src + (size_t)y * sstride;
        u8 *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) hsv_px(s + 3 * x, d + 3 * x, bidx);
    }
}

/* ------------------------------------------------------------------------------------
 * A.2  BORDER_REFLECT_101 (OpenCV borderInterpolate): -k -> k, (n-1)+k -> (n-1)-k.
 * ---------------------------------------------------------------------------------- */
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

/* ------------------------------------------------------------------------------------
 * A.3  GaussianBlur(src, dst, Size(k,k), 0, 0) on CV_8U, any channel count.
 * Call sites: objectPolygonPath.cpp:182 (3x3 on BGR), :184 (5x5 on HSV, hue blurred
 * linearly), pose_estimate.cpp:204 (5x5, commented out).
 * Taps are 8.8 fixed point; k in {1,3,5,7} with sigma<=0 come from OpenCV's fixed small
 * table; other odd k use sigma = 0.3*((k-1)*0.5-1)+0.8 and 4.x's error-diffusion
 * conversion (getGaussianKernelFixedPoint_ED).  dst = (sum_j ky[j]*(sum_i kx[i]*p) + 2^15) >> 16.
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_gaussian_taps(int k, int *taps)
{
    if (k < 1 || (k & 1) == 0 || k > 31) return -1;
    static const int t1[] = {256};
    static const int t3[] = {64, 128, 64};
    static const int t5[] = {16, 64, 96, 64, 16};
    static const int t7[] = {8, 28, 56, 72, 56, 28, 8};
    const int *fixed = k == 1 ? t1 : k == 3 ? t3 : k == 5 ? t5 : k == 7 ? t7 : NULL;
    if (fixed) {
        memcpy(taps, fixed, (size_t)k * sizeof(int));
        return 0;
    }
    long double sigma = 0.3L * ((k - 1) * 0.5L - 1.0L) + 0.8L;
    long double scale2x = -0.5L / (sigma * sigma);
    long double g[31], sum = 0;
    for (int i = 0; i < k; i++) {
        long double x = i - (k - 1) * 0.5L;
        g[i] = expl(scale2x * x * x);
        sum += g[i];
    }
    for (int i = 0; i < k; i++) g[i] /= sum;
    int n2 = k / 2;
    long double err = 0;
    long long acc = 0;
    for (int i = 0; i < n2; i++) {
        long double adj = g[i] * 256.0L + err;
        long long v0 = llrintl(adj); /* cvRound: half-to-even */
        err = adj - (long double)v0;
        taps[i] = taps[k - 1 - i] = (int)v0;
        acc += v0;
    }
    taps[n2] = (int)(256 - 2 * acc);
    return 0;
}

ORC_API int orc_gaussian_blur(const u8 *src, size_t sstride, u8 *dst, size_t dstride,
                              int w, int h, int cn, int k)
{
    int taps[31];
    if (orc_gaussian_taps(k, taps)) return -1;
    int r = k / 2;
    /* row pass into a full 8.8 plane (uint16 is enough: sum(taps)=256 -> max 65280) */
    size_t rowlen = (size_t)w * cn;
    uint16_t *tmp = (uint16_t *)malloc(rowlen * (size_t)h * sizeof(uint16_t));
    if (!tmp) return -2;
    for (int y = 0; y < h; y++) {
        const u8 *s = src + (size_t)y * sstride;
        uint16_t *t = tmp + (size_t)y * rowlen;
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                unsigned acc = 0;
                for (int i = 0; i < k; i++) acc += (unsigned)taps[i] * s[reflect101(x + i - r, w) * cn + c];
                t[x * cn + c] = (uint16_t)acc;
            }
    }
    for (int y = 0; y < h; y++) {
        u8 *d = dst + (size_t)y * dstride;
        for (size_t xc = 0; xc < rowlen; xc++) {
            uint32_t acc = 0;
            for (int j = 0; j < k; j++)
                acc += (uint32_t)taps[j] * tmp[(size_t)reflect101(y + j - r, h) * rowlen + xc];
            d[xc] = (u8)((acc + (1u << 15)) >> 16);
        }
    }
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * A.4  inRange(src_u8c3, Scalar lo, Scalar hi, dst_u8c1).
 * Call sites: objectPolygonPath.cpp:186, pose_estimate.cpp:205; bounds arrive as the raw
 * dynamic_reconfigure ints 0..256 (cfg/hsvThresh.cfg:8-13).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_in_range_bounds(const int32_t lo[3], const int32_t hi[3], int L[3], int U[3])
{
    for (int c = 0; c < 3; c++) {
        int l = lo[c], u = hi[c];
        if (l > u || l > 255 || u < 0) {
            l = 1;
            u = 0; /* never matches */
        } else {
            if (l < 0) l = 0;
            if (u > 255) u = 255;
        }
        L[c] = l;
        U[c] = u;
    }
}

ORC_API void orc_in_range(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h,
                          const int32_t lo[3], const int32_t hi[3])
{
    int L[3], U[3];
    orc_in_range_bounds(lo, hi, L, U);
    for (int y = 0; y < h; y++) {
        const u8 *s = src + (size_t)y * sstride;
        u8 *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int ok = 1;
            for (int c = 0; c < 3; c++) ok &= (s[3 * x + c] >= L[c]) & (s[3 * x + c] <= U[c]);
            d[x] = ok ? 255 : 0;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A.6  getStructuringElement.  shape: 0 RECT, 1 CROSS, 2 ELLIPSE (OpenCV enum values).
 * Call sites: objectPolygonPath.cpp:31-32, pose_estimate.cpp:40-41 (RECT 3x3, RECT 8x8).
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_structuring_element(int shape, int kw, int kh, u8 *elem)
{
    if (kw < 1 || kh < 1) return -1;
    int ax = kw / 2, ay = kh / 2;
    int r = kh / 2, c = kw / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < kh; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0 || (shape == 1 && i == ay)) {
            j2 = kw;
        } else if (shape == 1) {
            j1 = ax;
            j2 = j1 + 1;
        } else if (shape == 2) {
            int dy = i - r;
            if (abs(dy) <= r) {
                int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < kw ? c + dx + 1 : kw;
            }
        } else
            return -1;
        for (int j = 0; j < kw; j++) elem[i * kw + j] = (j >= j1 && j < j2) ? 1 : 0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * A.5  erode / dilate, CV_8U single channel, one iteration, BORDER_CONSTANT with
 * morphologyDefaultBorderValue(): out-of-image samples are ignored.
 * op: 0 erode (min), 1 dilate (max).  anchor (-1,-1) -> (kw/2, kh/2).
 * Call sites: objectPolygonPath.cpp:416-432, pose_estimate.cpp:407-414.
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_morph(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int op,
                      const u8 *elem, int kw, int kh, int ax, int ay)
{
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    if (ax >= kw || ay >= kh) return -1;
    for (int y = 0; y < h; y++) {
        u8 *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int acc = op ? 0 : 255;
            for (int j = 0; j < kh; j++) {
                int yy = y + j - ay;
                if (yy < 0 || yy >= h) continue;
                const u8 *s = src + (size_t)yy * sstride;
                for (int i = 0; i < kw; i++) {
                    if (!elem[j * kw + i]) continue;
                    int xx = x + i - ax;
                    if (xx < 0 || xx >= w) continue;
                    int v = s[xx];
                    if (op ? (v > acc) : (v < acc)) acc = v;
                }
            }
            d[x] = (u8)acc;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * a-7  Raster moments of a mask == cv::moments(mask, binaryImage=true) m00/m10/m01, plus
 * the second-order sums used for the orientation row.
 * m[0]=m00 m[1]=m10 m[2]=m01 m[3]=m20 m[4]=m11 m[5]=m02   (all exact integers)
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_moments(const u8 *mask, size_t stride, int w, int h, uint64_t m[6])
{
    memset(m, 0, 6 * sizeof(uint64_t));
    for (int y = 0; y < h; y++) {
        const u8 *s = mask + (size_t)y * stride;
        for (int x = 0; x < w; x++)
            if (s[x]) {
                m[0] += 1;
                m[1] += (uint64_t)x;
                m[2] += (uint64_t)y;
                m[3] += (uint64_t)x * x;
                m[4] += (uint64_t)x * y;
                m[5] += (uint64_t)y * y;
            }
    }
}

/* ------------------------------------------------------------------------------------
 * A.7  bayer_rggb8 -> BGR8 == cv::COLOR_BayerBG2BGR (cv_bridge's mapping), bilinear.
 * Interior pixels (1..H-2, 1..W-2) interpolate; border columns copy their inner
 * neighbour, then border rows copy their inner neighbour row.  H<=2 or W<=2 -> zeros
 * (OpenCV writes zeros when there is no interior).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_bayer_rggb2bgr(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h)
{
    for (int y = 0; y < h; y++) memset(dst + (size_t)y * dstride, 0, (size_t)w * 3);
    if (w <= 2 || h <= 2) return;
#define P(yy, xx) ((int)src[(size_t)(yy) * sstride + (xx)])
    for (int y = 1; y < h - 1; y++) {
        u8 *d = dst + (size_t)y * dstride;
        for (int x = 1; x < w - 1; x++) {
            int ey = y & 1, ex = x & 1;
            int R, G, B;
            int c = P(y, x);
            int cross = (P(y - 1, x) + P(y + 1, x) + P(y, x - 1) + P(y, x + 1) + 2) >> 2;
            int diag = (P(y - 1, x - 1) + P(y - 1, x + 1) + P(y + 1, x - 1) + P(y + 1, x + 1) + 2) >> 2;
            int hor = (P(y, x - 1) + P(y, x + 1) + 1) >> 1;
            int ver = (P(y - 1, x) + P(y + 1, x) + 1) >> 1;
            if (!ey && !ex) { /* R site */
                R = c, G = cross, B = diag;
            } else if (ey && ex) { /* B site */
                B = c, G = cross, R = diag;
            } else if (!ey && ex) { /* G on an R row: R left/right, B above/below */
                G = c, R = hor, B = ver;
            } else { /* G on a B row */
                G = c, B = hor, R = ver;
            }
            d[3 * x + 0] = (u8)B;
            d[3 * x + 1] = (u8)G;
            d[3 * x + 2] = (u8)R;
        }
        memcpy(d, d + 3, 3);
        memcpy(d + 3 * (size_t)(w - 1), d + 3 * (size_t)(w - 2), 3);
    }
#undef P
    memcpy(dst, dst + dstride, (size_t)w * 3);
    memcpy(dst + (size_t)(h - 1) * dstride, dst + (size_t)(h - 2) * dstride, (size_t)w * 3);
}

/* ------------------------------------------------------------------------------------
 * GRAY2BGR + bitwise_and (objectPolygonPath.cpp:189-190): frame & {mask,mask,mask}.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_mask_and(const u8 *bgr, size_t sstride, const u8 *mask, size_t mstride, u8 *dst,
                          size_t dstride, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++)
                dst[(size_t)y * dstride + 3 * x + c] =
                    bgr[(size_t)y * sstride + 3 * x + c] & mask[(size_t)y * mstride + x];
}

/* ------------------------------------------------------------------------------------
 * Whole chain for one frame, in the reference's call order (objectPolygonPath.cpp:182-188):
 *   [demosaic] -> [GaussianBlur pre_k] -> BGR2HSV -> [GaussianBlur post_k] -> inRange xN
 *   -> morph ops -> moments.
 * Stage outputs that the caller passes non-NULL are filled (tightly packed).
 * morph ops: n_morph entries of {op, shape, kw, kh, ax, ay}.
 * Used by tests (stage-by-stage parity against the HIP path) and by bench.py's
 * cpu_baseline leg ("port").
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int encoding;   /* 0 BGR8, 1 RGB8, 2 BAYER_RGGB8 */
    int pre_blur_k; /* 0/1 = none */
    int post_blur_k;
    int n_morph;
    int morph[32][6];
    int n_ranges;
    int32_t lo[8][3];
    int32_t hi[8][3];
} orc_chain_cfg;

ORC_API int orc_chain(const orc_chain_cfg *cfg, const u8 *frame, size_t stride, int w, int h,
                      u8 *out_bgr, u8 *out_hsv, u8 *out_blur, u8 *out_thresh /* [n_ranges][h][w] */,
                      u8 *out_mask /* [n_ranges][h][w] */, uint64_t *out_moments /* [n_ranges][6] */)
{
    size_t px = (size_t)w * h;
    u8 *a = (u8 *)malloc(px * 3), *b = (u8 *)malloc(px * 3);
    u8 *m0 = (u8 *)malloc(px), *m1 = (u8 *)malloc(px);
    int rc = 0;
    if (!a || !b || !m0 || !m1) {
        rc = -2;
        goto done;
    }
    int bidx = 0;
    if (cfg->encoding == 2) {
        orc_bayer_rggb2bgr(frame, stride, a, (size_t)w * 3, w, h);
    } else {
        for (int y = 0; y < h; y++) memcpy(a + (size_t)y * w * 3, frame + (size_t)y * stride, (size_t)w * 3);
        if (cfg->encoding == 1) bidx = 2;
    }
    if (cfg->pre_blur_k > 1) {
        if ((rc = orc_gaussian_blur(a, (size_t)w * 3, b, (size_t)w * 3, w, h, 3, cfg->pre_blur_k))) goto done;
        u8 *t = a; a = b; b = t;
    }
    if (out_bgr) memcpy(out_bgr, a, px * 3);
    orc_bgr2hsv(a, (size_t)w * 3, b, (size_t)w * 3, w, h, bidx);
    if (out_hsv) memcpy(out_hsv, b, px * 3);
    if (cfg->post_blur_k > 1) {
        if ((rc = orc_gaussian_blur(b, (size_t)w * 3, a, (size_t)w * 3, w, h, 3, cfg->post_blur_k))) goto done;
        u8 *t = a; a = b; b = t;
    }
    if (out_blur) memcpy(out_blur, b, px * 3);
    for (int r = 0; r < cfg->n_ranges; r++) {
        orc_in_range(b, (size_t)w * 3, m0, (size_t)w, w, h, cfg->lo[r], cfg->hi[r]);
        if (out_thresh) memcpy(out_thresh + (size_t)r * px, m0, px);
        for (int i = 0; i < cfg->n_morph; i++) {
            const int *mo = cfg->morph[i];
            u8 elem[31 * 31];
            if (mo[2] > 31 || mo[3] > 31) { rc = -1; goto done; }
            if ((rc = orc_structuring_element(mo[1], mo[2], mo[3], elem))) goto done;
            if ((rc = orc_morph(m0, (size_t)w, m1, (size_t)w, w, h, mo[0], elem, mo[2], mo[3], mo[4], mo[5]))) goto done;
            u8 *t = m0; m0 = m1; m1 = t;
        }
        if (out_mask) memcpy(out_mask + (size_t)r * px, m0, px);
        if (out_moments) orc_moments(m0, (size_t)w, w, h, out_moments + 6 * (size_t)r);
    }
done:
    free(a); free(b); free(m0); free(m1);
    return rc;
}
