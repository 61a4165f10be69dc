/* svo_oracle.c -- CPU restatement of the stereo-VO hot path.  TEST INFRASTRUCTURE (see svo_oracle.h).
 *
 * Citations: H  = libstereo-odometry/include/libstereo-odometry.h
 *            P  = libstereo-odometry/src/process_new_image_pair.cpp
 *            S2 = .../stage2_detect.cpp   S3 = .../stage3_match_left_right.cpp
 *            S4 = .../stage4_match_consecutive.cpp   S5 = .../stage5_optimization.cpp
 *            C  = .../common.cpp          SAD = .../compute_SAD8.cpp
 * all under /root/reference.  Where the reference calls OpenCV / MRPT / Eigen (absent: SURVEY.md 8c) the
 * published algorithm is restated and FROZEN here; those places are marked [frozen].
 *
 * Compile with -ffp-contract=off: the float formulas below are written so that the HIP kernels can match
 * them bit for bit (one IEEE operation per written operator, no fused multiply-add).
 */
#include "svo_oracle.h"
#include "../include/svo_orb_tables.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define EDGE_THRESHOLD 31          /* S2:486 */
#define HARRIS_BLOCK 7
#define RANSAC_MAX_HYP 1000        /* [frozen]  cv::findFundamentalMat's maxIters default of OpenCV >= 3.4 (2.4 sized its schedule from the confidence alone) */
#define MAXOCT SVO_MAX_OCTAVES

/* ------------------------------------------------------------------------------------------------ */
/* small helpers                                                                                    */
/* ------------------------------------------------------------------------------------------------ */
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }
static void* xcalloc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s ? s : 1); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }

/* total order on floats via their bit pattern (no NaNs on this path) */
static uint32_t ord32(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

static int cmp_u64_desc(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? 1 : (x > y ? -1 : 0); }
static int cmp_u64_asc(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : (x > y ? 1 : 0); }
static int cmp_u32_desc(const void* a, const void* b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? 1 : (x > y ? -1 : 0); }

int svo_oracle_version(void) { return SVO_ORACLE_VERSION; }

void svo_oracle_params_defaults(svo_params* p)
{
    memset(p, 0, sizeof(*p));
    p->nOctaves = 3;                         /* S1:27-30 (forced to 1 for dmORB, S1:80) */
    p->detect_method = SVO_DM_ORB;           /* north-star selector (reference default dmFASTER, S2:45) */
    p->non_maximal_suppression = 1;          /* S2:49 */
    p->nmsMethod = SVO_NMS_STANDARD;         /* S2:50 */
    p->min_distance = 3;                     /* S2:51 */
    p->orb_nfeats = 500;                     /* S2:52 */
    p->orb_nlevels = 8;                      /* S2:53 */
    p->minimum_ORB_response = 0.0;           /* S2:54 */
    p->fast_min_th = 5; p->fast_max_th = 30; /* S2:55 */
    p->initial_FAST_threshold = 20;          /* S2:56 */
    p->match_method = SVO_SM_DESC_BF;        /* north-star selector (reference default smSAD, S3:47) */
    p->orb_max_distance = 40;                /* S3:50 */
    p->orb_min_th = 30; p->orb_max_th = 100; /* S3:51 */
    p->enable_robust_1to1_match = 0;         /* S3:52 */
    p->max_y_diff = 0;                       /* S3:54 */
    p->ifm_method = SVO_IFM_DESC_BF;         /* H:610 */
    p->ifm_win_w = 16; p->ifm_win_h = 16;    /* no reference default (C:84); SURVEY appendix A #13 */
    p->filter_fund_matrix = 0;
    p->use_robust_kernel = 1;                /* C:70 */
    p->kernel_param = 3.0;                   /* C:71 */
    p->max_iters = 100;                      /* C:72 */
    p->initial_max_iters = 10;               /* C:73 */
    p->min_mod_out_vector = 1e-3;            /* C:74 */
    p->max_incr_cost = 3;                    /* C:76 */
    p->residual_threshold = 10.0;            /* C:77 */
    p->bad_tracking_th = 5;                  /* C:78 */
    p->use_previous_pose_as_initial = 1;     /* C:79 */
    p->use_custom_initial_pose = 0;          /* C:80 */
    p->vo_use_matches_ids = 0;               /* P:35 */
}

/* ------------------------------------------------------------------------------------------------ */
/* compute_SAD8_default  (SAD:71-98) -- known-answer only                                           */
/* ------------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------------------------
 * Stage 1: grey conversion + rectification (stage1_rectify.cpp:47-85).
 *   S1:50-51  CImage(obs->imageLeft, FAST_REF_OR_CONVERT_TO_GRAY)  -> cvCvtColor(BGR2GRAY)
 *   S1:66-72  m_stereo_rectifier.rectify(...)  -> mrpt::vision::CStereoRectifyMap -> cv::remap(INTER_LINEAR,
 *             BORDER_CONSTANT 0) with the maps of cv::initUndistortRectifyMap
 * Neither MRPT nor OpenCV is available (parity unpinned, see the header of this file); the published fixed-point
 * rules are frozen here:
 *   grey  = (4899 R + 9617 G + 1868 B + 8192) >> 14                      (8-bit cvtColor, 14-bit coefficients)
 *   remap : coordinates rounded to 1/32 pixel (cvRound = round-half-even of 32 x), sx = ix >> 5, fx = ix & 31;
 *           weights (32-fx)(32-fy)*32 ... fx*fy*32 (their sum is exactly 32768, which is what cv::remap's
 *           INTER_REMAP_COEF_SCALE table holds); out = (sum p*w + 16384) >> 15; taps outside the image read 0.
 * Deviation: the reference hands the UNCONVERTED images to the rectifier (S1:70-72 passes obs->imageLeft, not the
 * grey copy); for grey input that is the same thing, for colour input this restatement converts first.
 * ------------------------------------------------------------------------------------------------------------ */
static inline int grey_of(const uint8_t* p, int channels)
{
    if (channels == 1) return p[0];
    return (4899 * (int)p[2] + 9617 * (int)p[1] + 1868 * (int)p[0] + 8192) >> 14;
}

int svo_oracle_map_fixed(float mx, float my, int w, int h, int* sx, int* sy, int* fx, int* fy)
{
    *sx = *sy = -2; *fx = *fy = 0;
    if (!(mx > -4.0f && mx < (float)(w + 4) && my > -4.0f && my < (float)(h + 4))) return 0;      /* also NaN */
    const long ix = lrintf(mx * 32.0f), iy = lrintf(my * 32.0f);         /* cvRound: round half to even */
    const int x = (int)(ix >> 5), y = (int)(iy >> 5);
    if (x < -1 || x >= w || y < -1 || y >= h) return 0;
    *sx = x; *sy = y; *fx = (int)(ix & 31); *fy = (int)(iy & 31);
    return 1;
}

void svo_oracle_prepare(const uint8_t* src, int w, int h, long stride, int channels,
                        const float* map_x, const float* map_y, uint8_t* dst, long dst_stride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (!map_x || !map_y) { dst[(long)y * dst_stride + x] = (uint8_t)grey_of(src + (long)y * stride + (long)x * channels, channels); continue; }
            int sx, sy, fx, fy, v = 0;
            if (svo_oracle_map_fixed(map_x[(long)y * w + x], map_y[(long)y * w + x], w, h, &sx, &sy, &fx, &fy)) {
                int p[2][2];
                for (int dy = 0; dy < 2; dy++)
                    for (int dx = 0; dx < 2; dx++) {
                        const int xx = sx + dx, yy = sy + dy;
                        p[dy][dx] = (xx >= 0 && xx < w && yy >= 0 && yy < h) ? grey_of(src + (long)yy * stride + (long)xx * channels, channels) : 0;
                    }
                const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
                v = (p[0][0] * w00 + p[0][1] * w01 + p[1][0] * w10 + p[1][1] * w11 + 16384) >> 15;
            }
            dst[(long)y * dst_stride + x] = (uint8_t)v;
        }
}

uint32_t svo_oracle_sad8(const uint8_t* l, const uint8_t* r, size_t stride, int lx, int ly, int rx, int ry)
{
    const uint8_t* pl = l + stride * (size_t)(ly - 3) + (lx - 3);   /* window [x-3,x+4] x [y-3,y+4] */
    const uint8_t* pr = r + stride * (size_t)(ry - 3) + (rx - 3);
    uint32_t sum = 0;
    for (int y = 0; y < 8; y++) {
        for (int x = 0; x < 8; x++) { int d = (int)pl[x] - (int)pr[x]; sum += (uint32_t)(d > 0 ? d : -d); }
        pl += stride; pr += stride;
    }
    return sum;
}

/* ------------------------------------------------------------------------------------------------ */
/* pyramids                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
/* [frozen] ORB level geometry: scale_l = (float)1.2^l, size = round(W/scale) (S2:484-485 -> cv::ORB). */
int svo_oracle_pyramid_sizes(int w, int h, int nlevels, int* lw, int* lh, float* scale)
{
    for (int l = 0; l < nlevels; l++) {
        float sf = (float)pow(1.2, (double)l);
        scale[l] = sf;
        lw[l] = (int)lrintf((float)w / sf);
        lh[l] = (int)lrintf((float)h / sf);
    }
    return nlevels;
}

/* [frozen, oracle v7] cv::resize(..., INTER_LINEAR) on 8-bit images as OpenCV 2.4 / 3.x compute it (modules/imgproc/src/resize.cpp;
 * cv::ORB builds every pyramid level from the one before with it: S2:482-493 -> ORB_Impl::detectAndCompute).  Written from memory of that
 * source -- none of OpenCV is in this image:
 *   coefficients   scale = 1. / ((double)dst / src);  fx = (float)((d + 0.5) * scale - 0.5);  sx = cvFloor(fx);  fx -= sx;
 *                  sx < 0 -> (0, fx = 0);  sx >= src - 1 -> (src - 1, fx = 0);
 *                  the two taps' weights are rounded SEPARATELY to shorts: saturate_cast<short>((1.f - fx) * 2048), (fx * 2048)
 *                  (cvRound = round half to even; they add up to 2048 except where a float rounding of 1.f - fx crosses a tie)
 *   horizontal     HResizeLinear<uchar, int, short, 2048>:  D[x] = S[sx] * a0 + S[sx + 1] * a1          (int, 19 bits)
 *   vertical       the uchar specialisation of VResizeLinear (the SIMD body and its scalar tail are the same expression):
 *                      dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
 *                  i.e. the row sums lose their 4 low bits, each product is cut to quarter grey levels BEFORE the two are added, and
 *                  only then comes a rounding -- not the single (v + 2^21) >> 22 the generic FixedPtCast would give.  Oracle versions
 *                  up to 6 used that single rounding with exact-rational weights: one grey level away on ~10 % of the pixels
 *                  (VERDICT r05 "missing" #3). */
static long cv_round_f(float v) { return lrintf(v); }       /* cvRound: SSE cvtss2si / lrint, round half to even (default FP environment) */
void svo_oracle_resize_table(int src, int dst, int* idx, int* w01)
{
    const double inv_scale = (double)dst / src, scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; d++) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)floor((double)fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= src - 1) { fx = 0.f; sx = src - 1; }
        long a0 = cv_round_f((1.f - fx) * 2048.f), a1 = cv_round_f(fx * 2048.f);
        if (a0 > 32767) a0 = 32767; if (a1 > 32767) a1 = 32767;            /* saturate_cast<short> (never reached: both are in [0, 2048]) */
        idx[d] = sx; w01[d] = (int)a0 | ((int)a1 << 16);
    }
}

void svo_oracle_resize(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh)
{
    int* xi = (int*)xmalloc(sizeof(int) * (size_t)dw), *xf = (int*)xmalloc(sizeof(int) * (size_t)dw);
    int* yi = (int*)xmalloc(sizeof(int) * (size_t)dh), *yf = (int*)xmalloc(sizeof(int) * (size_t)dh);
    svo_oracle_resize_table(sw, dw, xi, xf);
    svo_oracle_resize_table(sh, dh, yi, yf);
    for (int y = 0; y < dh; y++) {
        int y0 = yi[y], y1 = y0 + 1 < sh ? y0 + 1 : sh - 1, b0 = yf[y] & 0xFFFF, b1 = yf[y] >> 16;
        const uint8_t* r0 = src + (size_t)y0 * sstride, *r1 = src + (size_t)y1 * sstride;
        for (int x = 0; x < dw; x++) {
            int x0 = xi[x], x1 = x0 + 1 < sw ? x0 + 1 : sw - 1, a0 = xf[x] & 0xFFFF, a1 = xf[x] >> 16;
            const int S0 = r0[x0] * a0 + r0[x1] * a1, S1 = r1[x0] * a0 + r1[x1] * a1;
            const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)y * dw + x] = (uint8_t)(v > 255 ? 255 : v);            /* uchar(...) of a value that cannot exceed 255 when a0 + a1 = b0 + b1 = 2048; clamped for the one-off weights */
        }
    }
    free(xi); free(xf); free(yi); free(yf);
}

/* [frozen] mrpt CImagePyramid::buildPyramidFast smooth halving (S1:82-83): 2x2 box average, round half up */
void svo_oracle_half_smooth(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst)
{
    int dw = sw / 2, dh = sh / 2;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const uint8_t* p = src + (size_t)(2 * y) * sstride + 2 * x;
            dst[(size_t)y * dw + x] = (uint8_t)((p[0] + p[1] + p[sstride] + p[sstride + 1] + 2) >> 2);
        }
}

/* ------------------------------------------------------------------------------------------------ */
/* FAST-9/16 [frozen]  (Rosten & Drummond; stands in for the FAST inside cv::ORB, S2:482-493, and     */
/* cv::FastFeatureDetector, S2:510-511)                                                              */
/* ------------------------------------------------------------------------------------------------ */
static const int fast_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int fast_dy[16] = { -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3 };

/* score = largest t for which the pixel is still a FAST-9 corner at threshold t
 *       = max over the 16 arcs of 9 contiguous circle pixels of min one-sided difference, minus 1.
 * returns 0 when score < th (not a corner at the requested threshold) */
static int fast_score_px(const uint8_t* p, int stride, int th)
{
    int c = p[0], d[16];
    unsigned bright = 0, dark = 0;
    for (int i = 0; i < 16; i++) {
        d[i] = (int)p[fast_dy[i] * stride + fast_dx[i]] - c;
        if (d[i] > th) bright |= 1u << i;
        if (-d[i] > th) dark |= 1u << i;
    }
    /* 9 contiguous set bits in a 16-bit ring */
    unsigned hit = 0;
    for (int pass = 0; pass < 2; pass++) {
        unsigned m = pass ? dark : bright, r = m | (m << 16);
        unsigned x = r & (r >> 1); x &= x >> 2; x &= x >> 4; x &= r >> 8;
        if (x & 0xFFFFu) hit = 1;
    }
    if (!hit) return 0;
    int best = 0;
    for (int s = 0; s < 16; s++) {
        int mb = 255, md = 255;
        for (int k = 0; k < 9; k++) { int v = d[(s + k) & 15]; if (v < mb) mb = v; if (-v < md) md = -v; }
        if (mb > best) best = mb;
        if (md > best) best = md;
    }
    return best - 1;
}

static void fast_score_region(const uint8_t* img, int w, int h, int stride, int th, int x0, int y0, int x1, int y1, uint8_t* score /* w*h, zeroed */)
{
    if (x0 < 3) x0 = 3; if (y0 < 3) y0 = 3; if (x1 > w - 3) x1 = w - 3; if (y1 > h - 3) y1 = h - 3;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++)
            score[(size_t)y * w + x] = (uint8_t)fast_score_px(img + (size_t)y * stride + x, stride, th);
}

void svo_oracle_fast_score_map(const uint8_t* img, int w, int h, int stride, int th, uint8_t* score)
{
    memset(score, 0, (size_t)w * h);
    fast_score_region(img, w, h, stride, th, 3, 3, w - 3, h - 3, score);
}

/* 3x3 non-max suppression (strictly greater than all 8 neighbours), then border filter
 * x,y in [EDGE, dim-EDGE). Appends key32 = score<<24 | (0xFFFFFF - pos), pos = y*w+x, row-major. */
static int fast_nms_candidates(const uint8_t* score, int w, int h, uint32_t* keys, int cap)
{
    int n = 0;
    for (int y = EDGE_THRESHOLD; y < h - EDGE_THRESHOLD; y++)
        for (int x = EDGE_THRESHOLD; x < w - EDGE_THRESHOLD; x++) {
            const uint8_t* s = score + (size_t)y * w + x;
            int v = s[0];
            if (!v) continue;
            if (v > s[-1] && v > s[1] && v > s[-w - 1] && v > s[-w] && v > s[-w + 1] && v > s[w - 1] && v > s[w] && v > s[w + 1]) {
                if (n < cap) keys[n] = ((uint32_t)v << 24) | (0xFFFFFFu - (uint32_t)(y * w + x));
                n++;
            }
        }
    return n < cap ? n : cap;
}

/* ------------------------------------------------------------------------------------------------ */
/* Harris, orientation, steered BRIEF  [frozen]  (Rublee et al. 2011; cv::ORB stand-in)             */
/* ------------------------------------------------------------------------------------------------ */
static float harris_response(const uint8_t* img, int stride, int x, int y)
{
    int a = 0, b = 0, c = 0;
    const int r = HARRIS_BLOCK / 2;
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) {
            const uint8_t* p = img + (size_t)(y + dy) * stride + (x + dx);
            int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-stride + 1] - (int)p[-stride - 1]) + ((int)p[stride + 1] - (int)p[stride - 1]);
            int Iy = ((int)p[stride] - (int)p[-stride]) * 2 + ((int)p[stride - 1] - (int)p[-stride - 1]) + ((int)p[stride + 1] - (int)p[-stride + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    /* cv::ORB's HarrisResponses, operator for operator: scale = 1.f / ((1 << 2) * blockSize * 255.f), scale_sq_sq = scale * scale * scale * scale
     * (left to right), response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq */
    const float s = 1.0f / (4.0f * 7.0f * 255.0f);
    const float s4 = s * s * s * s;
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float tr = fa + fb;
    return (fa * fb - fc * fc - 0.04f * tr * tr) * s4;
}

float svo_oracle_harris(const uint8_t* img, int stride, int x, int y) { return harris_response(img, stride, x, y); }

/* polynomial atan2 in degrees, 0..360 (the classic 7th-order minimax fit used by fast image-processing
 * libraries; ~0.3 deg max error).  One IEEE float operation per operator. */
static float atan2_deg(float y, float x)
{
    const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.220446e-16f);
        c2 = c * c;
        a = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

/* [frozen] sine and cosine of a float angle in [0, 2 pi] in single precision, one IEEE operation per operator (+ - * only, so that
 * the device computes the same bits): Cody-Waite reduction by pi/2 in three parts, then the Cephes single-precision minimax
 * polynomials on |r| <= pi/4.  Stands in for `(float)cos(angle), (float)sin(angle)` of cv::ORB's computeOrbDescriptor: within one
 * unit in the last place of the correctly rounded value (tests/test_oracle_units.py), i.e. a sample coordinate moves by < 2e-6 px. */
void svo_oracle_sincosf(float x, float* sn, float* cs)
{
    const int k = (int)(x * 0.63661977f + 0.5f);                     /* nearest multiple of pi/2: 0..4 */
    const float fk = (float)k;
    const float r = ((x - fk * 1.5703125f) - fk * 4.837512969970703125e-4f) - fk * 7.54978995489188216e-8f;
    const float z = r * r;
    const float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    switch (k & 3) {
    case 0: *sn = sp; *cs = cp; break;
    case 1: *sn = cp; *cs = -sp; break;
    case 2: *sn = -sp; *cs = -cp; break;
    default: *sn = -cp; *cs = sp; break;
    }
}

/* cv::ORB per keypoint (OpenCV >= 2.4 orb.cpp): IC_Angle -- integer moments over the radius-15 disc, angle = fastAtan2(m01, m10) --
 * and computeOrbDescriptor on the level blurred by GaussianBlur(7x7, sigma 2): the 256 pairs of bit_pattern_31_ rotated by the
 * CONTINUOUS angle (a = cos, b = sin of angle * (float)(CV_PI / 180); x = px * a - py * b, y = px * b + py * a in float, cvRound =
 * round half to even), bit i = blurred(p0) < blurred(p1), LSB first.  Only the (2 * 18 + 1)^2 blurred pixels a rotated pair can
 * reach are computed, from the (2 * 21 + 1)^2 raw window; keypoints keep 31 px from every border, so no border rule is involved.
 * Blur: 8-bit fixed point of cv::sepFilter2D's CV_8U path -- taps cvRound(256 g), both passes in integers, one rounding
 * (sum + 2^15) >> 16, saturated. */
#define DESC_R SVO_BRIEF_REACH
#define DESC_W (2 * DESC_R + 1)
/* computeOrbDescriptor's sampling: `center` points at the keypoint in a blurred image of row pitch `stride` */
static void steered_brief(const uint8_t* center, int stride, float angle_deg, uint8_t* desc)
{
    float a, b;
    svo_oracle_sincosf(angle_deg * 0.017453292f, &b, &a);          /* angle *= (float)(CV_PI / 180.f); a = cos, b = sin */
    memset(desc, 0, SVO_DESC_BYTES);
    for (int i = 0; i < SVO_BRIEF_NPAIRS; i++) {
        const int8_t* pr = svo_brief_pat[i];
        const float x0 = (float)pr[0] * a - (float)pr[1] * b, y0 = (float)pr[0] * b + (float)pr[1] * a;
        const float x1 = (float)pr[2] * a - (float)pr[3] * b, y1 = (float)pr[2] * b + (float)pr[3] * a;
        const int t0 = center[(int)lrintf(y0) * stride + (int)lrintf(x0)], t1 = center[(int)lrintf(y1) * stride + (int)lrintf(x1)];
        if (t0 < t1) desc[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
}
static void orb_angle_desc(const uint8_t* img, int stride, int x, int y, float* angle_out, uint8_t* desc)
{
    const uint8_t* c = img + (size_t)y * stride + x;
    /* intensity centroid over the radius-15 disc */
    int m10 = 0, m01 = 0;
    for (int v = -SVO_HALF_PATCH; v <= SVO_HALF_PATCH; v++) {
        int um = svo_umax[v < 0 ? -v : v];
        for (int u = -um; u <= um; u++) { int I = c[v * stride + u]; m10 += u * I; m01 += v * I; }
    }
    float angle = atan2_deg((float)m01, (float)m10);
    *angle_out = angle;
    if (!desc) return;                                   /* the orientation alone: reads the radius-15 disc only */
    int tmp[DESC_W + 6][DESC_W];
    uint8_t bl[DESC_W][DESC_W];
    for (int r = 0; r < DESC_W + 6; r++)
        for (int q = 0; q < DESC_W; q++) {
            const uint8_t* p = c + (r - DESC_R - 3) * stride + (q - DESC_R);
            int s = 0;
            for (int k = 0; k < 7; k++) s += svo_gauss7[k] * p[k - 3];
            tmp[r][q] = s;
        }
    for (int r = 0; r < DESC_W; r++)
        for (int q = 0; q < DESC_W; q++) {
            int s = 0;
            for (int k = 0; k < 7; k++) s += svo_gauss7[k] * tmp[r + k][q];
            s = (s + 32768) >> 16;
            bl[r][q] = (uint8_t)(s > 255 ? 255 : s);
        }
    steered_brief(&bl[DESC_R][DESC_R], DESC_W, angle, desc);
}

/* the steering alone, on an image that is ALREADY blurred (third-party cross-check against scikit-image's _orb_loop, which applies
 * the same table with the same rotation to whatever image it is given); (x, y) must keep 18 pixels from every border */
void svo_oracle_steered_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32)
{
    steered_brief(blurred + (size_t)y * stride + x, stride, angle_deg, desc32);
}

/* orientation (degrees, [0, 360)) and steered BRIEF-256 of ONE position: the per-keypoint step of stage 2 on its own, for the
 * third-party cross-check (tests/test_oracle_thirdparty.py).  (x, y) must keep 21 pixels from every border; with desc32 == NULL only
 * the orientation is computed and 15 pixels are enough (the sanitizer build found the descriptor's window read for a corner 16 pixels
 * from the border when the descriptor was computed and thrown away). */
float svo_oracle_orb_angle(const uint8_t* img, int stride, int x, int y, uint8_t* desc32)
{
    float a = 0.f;
    orb_angle_desc(img, stride, x, y, &a, desc32);
    return a;
}

/* ------------------------------------------------------------------------------------------------ */
/* cv::ORB::detectAndCompute stand-in  [frozen]   (call site S2:482-493)                            */
/* ------------------------------------------------------------------------------------------------ */
static void orb_level_quota(int nfeatures, int nlevels, int* q)
{
    float factor = (float)(1.0 / 1.2);
    float nd = (float)nfeatures * (1.0f - factor) / (1.0f - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { q[l] = (int)lrintf(nd); sum += q[l]; nd *= factor; }
    q[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

void svo_oracle_level_quota(int nfeatures, int nlevels, int* q) { orb_level_quota(nfeatures, nlevels, q); }

int svo_oracle_orb_detect(const uint8_t* img, int w, int h, int stride, int nfeatures, int nlevels,
                          int fast_th, svo_keypoint* kps, uint8_t* desc, int cap)
{
    if (nlevels < 1) nlevels = 1; if (nlevels > 16) nlevels = 16;
    int lw[16], lh[16], quota[16]; float sc[16];
    svo_oracle_pyramid_sizes(w, h, nlevels, lw, lh, sc);
    orb_level_quota(nfeatures, nlevels, quota);
    uint8_t* bufs[16]; const uint8_t* lv[16]; int ls[16];
    lv[0] = img; ls[0] = stride; bufs[0] = NULL;
    for (int l = 1; l < nlevels; l++) {
        bufs[l] = (uint8_t*)xmalloc((size_t)lw[l] * lh[l]);
        svo_oracle_resize(lv[l - 1], lw[l - 1], lh[l - 1], ls[l - 1], bufs[l], lw[l], lh[l]);
        lv[l] = bufs[l]; ls[l] = lw[l];
    }
    int n = 0;
    for (int l = 0; l < nlevels; l++) {
        const int W = lw[l], H = lh[l];
        if (W <= 2 * EDGE_THRESHOLD || H <= 2 * EDGE_THRESHOLD || quota[l] <= 0) continue;
        uint8_t* score = (uint8_t*)xcalloc((size_t)W * H, 1);
        fast_score_region(lv[l], W, H, ls[l], fast_th, EDGE_THRESHOLD - 1, EDGE_THRESHOLD - 1, W - EDGE_THRESHOLD + 1, H - EDGE_THRESHOLD + 1, score);
        int ccap = (W - 2 * EDGE_THRESHOLD) * (H - 2 * EDGE_THRESHOLD);
        uint32_t* keys = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)ccap);
        int nc = fast_nms_candidates(score, W, H, keys, ccap);
        free(score);
        /* KeyPointsFilter::retainBest(2 * quota) by FAST score: the best 2*quota AND everything that ties with the last of them
         * ("the boundary response ... in the case of FAST may be ambiguous": integer scores tie routinely) */
        qsort(keys, (size_t)nc, sizeof(uint32_t), cmp_u32_desc);
        if (nc > 2 * quota[l]) {
            int cut = 2 * quota[l];
            const uint32_t boundary = keys[cut - 1] >> 24;
            while (cut < nc && (keys[cut] >> 24) == boundary) cut++;
            nc = cut;
        }
        /* Harris response, keep the best quota by (response desc, position asc) */
        uint64_t* hk = (uint64_t*)xmalloc(sizeof(uint64_t) * (size_t)(nc ? nc : 1));
        for (int i = 0; i < nc; i++) {
            uint32_t pos = 0xFFFFFFu - (keys[i] & 0xFFFFFFu);
            int x = (int)(pos % (uint32_t)W), y = (int)(pos / (uint32_t)W);
            float r = harris_response(lv[l], ls[l], x, y);
            hk[i] = ((uint64_t)ord32(r) << 32) | (uint64_t)(0xFFFFFFFFu - pos);
        }
        qsort(hk, (size_t)nc, sizeof(uint64_t), cmp_u64_desc);
        if (nc > quota[l]) nc = quota[l];
        for (int i = 0; i < nc && n < cap; i++) {
            uint32_t pos = 0xFFFFFFFFu - (uint32_t)(hk[i] & 0xFFFFFFFFu);
            int x = (int)(pos % (uint32_t)W), y = (int)(pos / (uint32_t)W);
            svo_keypoint* k = &kps[n];
            float ang;
            orb_angle_desc(lv[l], ls[l], x, y, &ang, desc + (size_t)n * SVO_DESC_BYTES);
            k->x = (float)x * sc[l]; k->y = (float)y * sc[l];
            k->size = 31.0f * sc[l];
            k->angle = ang;
            k->response = harris_response(lv[l], ls[l], x, y);
            k->octave = l; k->class_id = -1;
            n++;
        }
        free(hk); free(keys);
    }
    for (int l = 1; l < nlevels; l++) free(bufs[l]);
    return n;
}

/* cv::FastFeatureDetector::detect (threshold, NMS on, TYPE_9_16) + cv::ORB::create()->compute  [frozen]
 * (call site S2:510-512): keypoints in row-major order, size 7, response = FAST score; compute() drops
 * those closer than 31 px to the border and adds orientation + descriptor at level 0. */
int svo_oracle_fast_orb_detect(const uint8_t* img, int w, int h, int stride, int fast_th,
                               svo_keypoint* kps, uint8_t* desc, int cap)
{
    if (w <= 2 * EDGE_THRESHOLD || h <= 2 * EDGE_THRESHOLD) return 0;
    uint8_t* score = (uint8_t*)xcalloc((size_t)w * h, 1);
    fast_score_region(img, w, h, stride, fast_th, EDGE_THRESHOLD - 1, EDGE_THRESHOLD - 1, w - EDGE_THRESHOLD + 1, h - EDGE_THRESHOLD + 1, score);
    int n = 0;
    for (int y = EDGE_THRESHOLD; y < h - EDGE_THRESHOLD; y++)
        for (int x = EDGE_THRESHOLD; x < w - EDGE_THRESHOLD; x++) {
            const uint8_t* s = score + (size_t)y * w + x;
            int v = s[0];
            if (!v) continue;
            if (!(v > s[-1] && v > s[1] && v > s[-w - 1] && v > s[-w] && v > s[-w + 1] && v > s[w - 1] && v > s[w] && v > s[w + 1])) continue;
            if (n >= cap) continue;
            svo_keypoint* k = &kps[n];
            float ang;
            orb_angle_desc(img, stride, x, y, &ang, desc + (size_t)n * SVO_DESC_BYTES);
            k->x = (float)x; k->y = (float)y; k->size = 7.0f; k->angle = ang; k->response = (float)v;
            k->octave = 0; k->class_id = -1;
            n++;
        }
    free(score);
    return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* m_non_max_sup  (S2:225-283 mask overload, S2:296-370 copying overload)                            */
/* ------------------------------------------------------------------------------------------------ */
/* DEVIATION: the reference's unstable std::sort on response (S2:242, 324) is replaced
 * by the total order (response desc, input index asc). */
static int nms_walk(const svo_keypoint* kps, int n, int min_distance, int img_w, int img_h, int num_out_points,
                    int32_t* out_order, uint8_t* survivors)
{
    if (n <= 0) return 0;
    uint64_t* keys = (uint64_t*)xmalloc(sizeof(uint64_t) * (size_t)n);
    for (int i = 0; i < n; i++) keys[i] = ((uint64_t)ord32(kps[i].response) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
    qsort(keys, (size_t)n, sizeof(uint64_t), cmp_u64_desc);                       /* S2:322-324 */
    const unsigned cell = (unsigned)((double)min_distance / 2.0);                   /* S2:331 */
    const float inv = 1.0f / (float)cell;                                           /* S2:332 */
    const unsigned glx = (unsigned)(1 + (float)(size_t)img_w * inv);                /* S2:334 */
    const unsigned gly = (unsigned)(1 + (float)(size_t)img_h * inv);                /* S2:335 */
    uint8_t* occ = (uint8_t*)xcalloc((size_t)glx * gly, 1);                         /* S2:337-338 */
    int k = 0, c = 0;
    while (c < num_out_points && k < n) {                                           /* S2:342 */
        int idx = (int)(0xFFFFFFFFu - (uint32_t)(keys[k++] & 0xFFFFFFFFu));
        const svo_keypoint* kp = &kps[idx];
        size_t sx = (size_t)(kp->x * inv), sy = (size_t)(kp->y * inv);              /* S2:348-349 */
        if (sx >= glx || sy >= gly) continue;   /* out of the grid: undefined in the reference; skipped */
        if (occ[sx * gly + sy]) continue;                                           /* S2:351 */
        occ[sx * gly + sy] = 1;                                                     /* S2:355-359 */
        if (sx > 0) occ[(sx - 1) * gly + sy] = 1;
        if (sy > 0) occ[sx * gly + sy - 1] = 1;
        if (sx < glx - 1) occ[(sx + 1) * gly + sy] = 1;
        if (sy < gly - 1) occ[sx * gly + sy + 1] = 1;
        if (out_order) out_order[c] = idx;                                          /* S2:362-363 */
        if (survivors) survivors[idx] = 1;                                          /* S2:279 */
        ++c;
    }
    free(occ); free(keys);
    return c;
}

int svo_oracle_nms_copy(const svo_keypoint* kps, int n, int min_distance, int img_w, int img_h, int num_out_points, int32_t* out_order)
{
    return nms_walk(kps, n, min_distance, img_w, img_h, num_out_points, out_order, NULL);
}

void svo_oracle_nms_mask(const svo_keypoint* kps, int n, int min_distance, int img_w, int img_h, int num_out_points, uint8_t* survivors)
{
    memset(survivors, 0, (size_t)(n > 0 ? n : 0));                                  /* S2:238 */
    nms_walk(kps, n, min_distance, img_w, img_h, num_out_points, NULL, survivors);
}

/* ------------------------------------------------------------------------------------------------ */
/* m_adaptive_non_max_sup  (S2:141-215): suppression radius of every keypoint = squared distance to  */
/* the nearest keypoint that is "robustly stronger" (response > it / 0.9), then keep by radius.      */
/* ------------------------------------------------------------------------------------------------ */
/* Kept quirks: the strongest keypoint bounds every radius whatever its response ratio (S2:176) and is skipped by
 * the inner loop (k2 > 0, S2:179); the squared distance is evaluated in float (cv::Point2f arithmetic, S2:176/185)
 * and compared as double; only the first min(num_out_points, N) entries of the radius order are looked at and those
 * with radius <= min_radius_th^2 (default 0: exact duplicates of a stronger point) are dropped (S2:207-214).
 * DEVIATION (appendix A #7): both std::sort calls are unstable; total orders (response desc, index asc) and
 * (radius desc, index asc) are used instead. */
int svo_oracle_anms_copy(const svo_keypoint* kps, int n, int num_out_points, double min_radius_th, int32_t* out_order)
{
    const int actual = num_out_points < n ? num_out_points : n;                      /* S2:151 */
    if (actual <= 0) return 0;                                                      /* S2:152 */
    const double CROB = 0.9;                                                        /* S2:154 */
    uint64_t* keys = (uint64_t*)xmalloc(sizeof(uint64_t) * (size_t)n);
    for (int i = 0; i < n; i++) keys[i] = ((uint64_t)ord32(kps[i].response) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
    qsort(keys, (size_t)n, sizeof(uint64_t), cmp_u64_desc);                         /* S2:160 */
    int32_t* sorted = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < n; i++) sorted[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFu));
    float* radius = (float*)xmalloc(sizeof(float) * (size_t)n);
    const svo_keypoint* s0 = &kps[sorted[0]];
    radius[sorted[0]] = INFINITY;                                                   /* S2:167 */
    for (int k1 = 1; k1 < n; k1++) {                                                /* S2:171-192 */
        const svo_keypoint* a = &kps[sorted[k1]];
        float dx = a->x - s0->x, dy = a->y - s0->y;
        float min_ri = fabsf(dx * dx + dy * dy);                                    /* S2:176 */
        for (int k2 = k1 - 1; k2 > 0; --k2) {                                       /* S2:179 */
            const svo_keypoint* b = &kps[sorted[k2]];
            if ((double)a->response < CROB * (double)b->response) {                 /* S2:183 */
                dx = a->x - b->x; dy = a->y - b->y;
                const float ri = fabsf(dx * dx + dy * dy);                          /* S2:185 */
                if (ri < min_ri) min_ri = ri;
            }
        }
        radius[sorted[k1]] = min_ri;
    }
    for (int i = 0; i < n; i++) keys[i] = ((uint64_t)ord32(radius[i]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
    qsort(keys, (size_t)n, sizeof(uint64_t), cmp_u64_desc);                         /* S2:196 */
    const double th2 = min_radius_th * min_radius_th;                               /* S2:194 */
    int nk = 0;
    for (int i = 0; i < actual; i++) {                                              /* S2:207-214 */
        const int idx = (int)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFu));
        if ((double)radius[idx] > th2) out_order[nk++] = idx;
    }
    free(keys); free(sorted); free(radius);
    return nk;
}

/* ------------------------------------------------------------------------------------------------ */
/* m_update_indexes(order=true)  (S2:65-130)                                                         */
/* ------------------------------------------------------------------------------------------------ */
/* DEVIATION (appendix A #7): unstable sort on pt.y (S2:89) -> total order (pt.y asc, input index asc). */
void svo_oracle_row_sort_index(const svo_keypoint* kps, int n, int img_h, int32_t* order, int64_t* idx)
{
    uint64_t* keys = (uint64_t*)xmalloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) keys[i] = ((uint64_t)ord32(kps[i].y) << 32) | (uint32_t)i;
    qsort(keys, (size_t)n, sizeof(uint64_t), cmp_u64_asc);
    for (int i = 0; i < n; i++) order[i] = (int32_t)(keys[i] & 0xFFFFFFFFu);
    free(keys);
    for (int r = 0; r < img_h; r++) idx[r] = 0;                                     /* S2:74 fresh vector */
    int64_t from = 0; size_t feats_till_now = 0, current_row = 0;
    for (int i = 0; i < n; i++) {                                                   /* S2:107-129 */
        const svo_keypoint* f = &kps[order[i]];
        if (i == 0) { current_row = (size_t)f->y; from = (int64_t)current_row; continue; }   /* S2:110-117 (zeros already there) */
        if (f->y == (float)(int)current_row) { ++feats_till_now; continue; }         /* S2:119-123 */
        current_row = (size_t)f->y;                                                 /* S2:124 */
        int64_t to = (int64_t)current_row;
        ++feats_till_now;
        for (int64_t r = from; r < to && r < img_h; r++) idx[r] = (int64_t)feats_till_now;   /* S2:126-127 */
        from = to;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Hamming brute force  [frozen]  (cv::BFMatcher(NORM_HAMMING,false).match; S3:88-94, S4:141-142)    */
/* ------------------------------------------------------------------------------------------------ */
static int hamming256(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int k = 0; k < 4; k++) { uint64_t x, y; memcpy(&x, a + 8 * k, 8); memcpy(&y, b + 8 * k, 8); d += __builtin_popcountll(x ^ y); }
    return d;
}

/* one best train index per query row, FIRST minimum on ties; idx = -1 when there is no train row */
void svo_oracle_hamming_bf(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist)
{
    for (int i = 0; i < nq; i++) {
        int best = 0x7FFFFFFF, bj = -1;
        for (int j = 0; j < nt; j++) {
            int d = hamming256(q + (size_t)i * SVO_DESC_BYTES, t + (size_t)j * SVO_DESC_BYTES);
            if (d < best) { best = d; bj = j; }
        }
        idx[i] = bj; dist[i] = bj < 0 ? 0 : best;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* stage 3: left-right matching  (S3:62-484)                                                        */
/* ------------------------------------------------------------------------------------------------ */
#define INVALID_IDX (-1)

static int match_lr_bf(const svo_params* p, int orb_th, const svo_keypoint* kl, const uint8_t* dl, int nl,
                       const svo_keypoint* kr, const uint8_t* dr, int nr, int img_w, svo_dmatch* out, int cap)
{
    if (nl <= 0 || nr <= 0) return 0;
    int32_t* bi = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)nl), *bd = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)nl);
    svo_oracle_hamming_bf(dl, nl, dr, nr, bi, bd);                                  /* S3:91-94 */
    uint8_t* keep = (uint8_t*)xmalloc((size_t)nl);
    memset(keep, 1, (size_t)nl);
    if (p->enable_robust_1to1_match) {                                              /* S3:124-148 */
        double* cd = (double*)xmalloc(sizeof(double) * (size_t)nr);
        int32_t* cq = (int32_t*)xcalloc((size_t)nr, sizeof(int32_t));
        for (int j = 0; j < nr; j++) cd[j] = -1.0;
        for (int k = 0; k < nl; k++) {
            int idR = bi[k];
            if (cd[idR] < 0 || cd[idR] > (double)(float)bd[k]) { cd[idR] = (double)(float)bd[k]; cq[idR] = k; }
        }
        for (int k = 0; k < nl; k++) if (k != cq[bi[k]]) keep[k] = 0;
        free(cd); free(cq);
    }
    const double min_disp = 1, max_disp = (double)img_w;                            /* S3:155-156 */
    int m = 0;
    for (int k = 0; k < nl; k++) {                                                  /* S3:159-175 */
        if (!keep[k]) continue;
        const int diff = (int)(kl[k].y - kr[bi[k]].y);                              /* S3:162 float diff -> int */
        const int disp = (int)(kl[k].x - kr[bi[k]].x);                              /* S3:163 */
        const float distance = (float)bd[k];
        if ((double)abs(diff) > p->max_y_diff || distance > (float)orb_th || (double)disp < min_disp || (double)disp > max_disp) continue;
        if (m < cap) { out[m].queryIdx = k; out[m].trainIdx = bi[k]; out[m].imgIdx = 0; out[m].distance = distance; }
        m++;
    }
    free(bi); free(bd); free(keep);
    return m < cap ? m : cap;
}

/* smDescRbR (S3:185-419) with its quirks kept */
static int match_lr_rbr(const svo_params* p, const svo_keypoint* kl, const uint8_t* dl, int nl, const int64_t* idxL,
                        const svo_keypoint* kr, const uint8_t* dr, int nr, const int64_t* idxR,
                        int img_w, int img_h, int detect_method, svo_dmatch* out, int cap)
{
    double minimum_response = 0;                                                    /* S3:189-193 */
    if (detect_method == SVO_DM_ORB) minimum_response = p->minimum_ORB_response;
    const size_t max_distance = (size_t)p->orb_max_distance;                        /* S3:205 */
    const double max_ratio = 1;                                                     /* S3:196 */
    int* left_matches = (int*)xmalloc(sizeof(int) * (size_t)(nl > 0 ? nl : 1));
    int* ra_first = (int*)xmalloc(sizeof(int) * (size_t)(nr > 0 ? nr : 1));
    uint32_t* ra_second = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(nr > 0 ? nr : 1));
    for (int i = 0; i < nl; i++) left_matches[i] = INVALID_IDX;                     /* S3:228 */
    for (int j = 0; j < nr; j++) { ra_first[j] = INVALID_IDX; ra_second[j] = 0xFFFFFFFFu; }   /* S3:229 */
    const int max_disparity = (int)((double)img_w * 0.7);                           /* S3:247 */
    const int d_round = (int)round(p->max_y_diff);
    for (size_t y = 0; y + 1 < (size_t)img_h; y++) {                                /* S3:250 */
        const size_t L0 = (size_t)idxL[y], L1 = (size_t)idxL[y + 1];                /* S3:253 */
        const int mrr = (int)y - d_round;
        const size_t min_row_right = (size_t)(mrr > 0 ? mrr : 0);                   /* S3:254 */
        size_t max_row_right = y + (size_t)d_round;                                 /* S3:255 */
        if (max_row_right > (size_t)(img_h - 1)) max_row_right = (size_t)(img_h - 1);
        const size_t R0 = (size_t)idxR[min_row_right], R1 = (size_t)idxR[max_row_right];   /* S3:256 */
        const size_t nFL = L1 - L0, nFR = R1 - R0;                                  /* S3:259-260 (unsigned wrap kept) */
        if (nFL == 0 || nFR == 0) continue;                                         /* S3:262 */
        if (L1 < L0 || R1 < R0) continue;   /* wrapped ranges: the reference's for-loops simply do not execute */
        for (size_t iL = L0; iL < L1; iL++) {                                       /* S3:265 */
            const svo_keypoint* fL = &kl[iL];
            uint32_t min_1 = 0xFFFFFFFFu, min_2 = 0xFFFFFFFFu; int min_idx = INVALID_IDX;   /* S3:270-272 */
            for (size_t iR = R0; iR < R1; iR++) {                                   /* S3:274 */
                const svo_keypoint* fR = &kr[iR];
                if ((double)fL->response < minimum_response || (double)fR->response < minimum_response) continue;   /* S3:279 */
                const int disparity = (int)(fL->x - fR->x);                         /* S3:283 */
                if (disparity < 1 || disparity > max_disparity) continue;           /* S3:284 */
                uint8_t d = 0;                                                      /* S3:321-329: uint8_t accumulator wraps */
                for (int k = 0; k < SVO_DESC_BYTES; k++) {
                    uint8_t x_or = dl[iL * SVO_DESC_BYTES + k] ^ dr[iR * SVO_DESC_BYTES + k];
                    uint8_t count; for (count = 0; x_or; count++) x_or &= (uint8_t)(x_or - 1);
                    d = (uint8_t)(d + count);
                }
                const size_t dist = (size_t)d;                                      /* S3:331 */
                if (dist > max_distance) continue;                                  /* S3:334 */
                if (dist < min_1) { min_2 = min_1; min_1 = (uint32_t)dist; min_idx = (int)iR; }   /* S3:338-343 */
                else if (dist < min_2) min_2 = (uint32_t)dist;
                (void)max_ratio;                                                    /* S3:347-349: no effect */
            }
            if (min_idx != INVALID_IDX) {                                           /* S3:357 */
                if (p->enable_robust_1to1_match) {                                  /* S3:359-377 */
                    if (ra_first[min_idx] == INVALID_IDX) { left_matches[iL] = min_idx; ra_first[min_idx] = (int)iL; ra_second[min_idx] = min_1; }
                    else if (min_1 < ra_second[min_idx]) { left_matches[ra_first[min_idx]] = INVALID_IDX; left_matches[iL] = min_idx; ra_first[min_idx] = (int)iL; ra_second[min_idx] = min_1; }
                } else if (ra_first[min_idx] == INVALID_IDX) {                      /* S3:381-386 */
                    left_matches[iL] = min_idx; ra_first[min_idx] = (int)iL; ra_second[min_idx] = min_1;
                }
            }
        }
    }
    int m = 0;
    for (int i = 0; i < nl; i++)                                                    /* S3:398-409 */
        if (left_matches[i] != INVALID_IDX) {
            int fr = left_matches[i];
            /* DMatch(i,fr,d): three-argument constructor = (queryIdx, trainIdx, distance); imgIdx = -1 */
            if (m < cap) { out[m].queryIdx = i; out[m].trainIdx = fr; out[m].imgIdx = -1; out[m].distance = (float)ra_second[fr]; }
            m++;
        }
    free(left_matches); free(ra_first); free(ra_second);
    return m < cap ? m : cap;
}

/* matches_lr_row_index (S3:425-445).  DEVIATION (appendix A #11): entry [imgH] = number of pairings,
 * not number of left keypoints (the reference value makes S4:530,557-561 read out of bounds). */
static void matches_row_index(const svo_dmatch* m, int nm, const svo_keypoint* kl, int img_h, int64_t* ri)
{
    int idx = 0;
    for (int y = 0; y < img_h; y++) {
        ri[y] = idx;
        while (idx < nm && kl[m[idx].queryIdx].y <= (float)y) idx++;                /* S3:441 */
    }
    ri[img_h] = nm;
}

int svo_oracle_match_lr(const svo_params* p, int orb_th, const svo_keypoint* kl, const uint8_t* dl, int nl,
                        const int64_t* idxl, const svo_keypoint* kr, const uint8_t* dr, int nr,
                        const int64_t* idxr, int img_w, int img_h, svo_dmatch* out, int cap, int64_t* row_index)
{
    int m;
    if (p->match_method == SVO_SM_DESC_BF) m = match_lr_bf(p, orb_th, kl, dl, nl, kr, dr, nr, img_w, out, cap);
    else if (p->match_method == SVO_SM_DESC_RBR) m = match_lr_rbr(p, kl, dl, nl, idxl, kr, dr, nr, idxr, img_w, img_h, p->detect_method, out, cap);
    else return -2;
    if (row_index) matches_row_index(out, m, kl, img_h, row_index);
    return m;
}

/* ------------------------------------------------------------------------------------------------ */
/* fundamental-matrix RANSAC  [frozen]  (cv::findFundamentalMat(FM_RANSAC,1.0,0.99); S4:202,237,684,696) */
/* ------------------------------------------------------------------------------------------------ */
/* cv::RNG [frozen, v5; recalled: OpenCV's source is not in /root/reference].  core.hpp / operations.hpp: a multiply-with-carry generator,
 *   state = (uint64)(unsigned)state * 4164903690U + (unsigned)(state >> 32);  next() = (unsigned)state;
 *   uniform(int a, int b) = a == b ? a : (int)(next() % (b - a) + a);
 * and RANSACPointSetRegistrator::run seeds a NEW one per call: `RNG rng((uint64)-1);` (modules/calib3d/src/ptsetreg.cpp), so every
 * cv::findFundamentalMat call of the reference (S4:202, 237, 684, 696) draws the same raw sequence. */
typedef struct { uint64_t state; } cv_rng;
static uint32_t cv_rng_next(cv_rng* r) { r->state = (uint64_t)(uint32_t)r->state * 4164903690U + (uint32_t)(r->state >> 32); return (uint32_t)r->state; }
void svo_oracle_cv_rng_raw(uint32_t* out, int count) { cv_rng r; r.state = 0xFFFFFFFFFFFFFFFFULL; for (int i = 0; i < count; i++) out[i] = cv_rng_next(&r); }

/* haveCollinearPoints (modules/calib3d/src/precomp.hpp), as FMEstimatorCallback::checkSubset calls it with count = 7 for either image:
 * is the LAST selected point on a line through two earlier ones (or do two of the three coincide)?  The points are Point2f: the
 * differences are taken IN FLOAT and only then widened (`double dx1 = ptr[j].x - ptr[i].x;`) -- [oracle v7; versions 5-6 widened each
 * coordinate first, which can flip a decision at the FLT_EPSILON-relative threshold for sub-pixel coordinates, ADVICE r05]. */
static int have_collinear(const float* pts, const int* idx, int count)
{
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = (double)(pts[2 * idx[j]] - pts[2 * idx[i]]), dy1 = (double)(pts[2 * idx[j] + 1] - pts[2 * idx[i] + 1]);
        for (int k = 0; k < j; k++) {
            const double dx2 = (double)(pts[2 * idx[k]] - pts[2 * idx[i]]), dy2 = (double)(pts[2 * idx[k] + 1] - pts[2 * idx[i] + 1]);
            if (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;    /* FLT_EPSILON */
        }
    }
    return 0;
}

/* RANSACPointSetRegistrator::getSubset (checkPartialSubsets = false): seven indices by rng.uniform(0, count), a draw that repeats an
 * earlier one of the same attempt is drawn again; a complete attempt that fails checkSubset (collinearity in either image) costs one of
 * the maxAttempts (10000, as run() passes) and is drawn afresh.  *attempts receives the attempts this call used (statistics only). */
static int ransac_get_subset(cv_rng* rng, const float* p1, const float* p2, int n, int* idx, int max_attempts, int* attempts)
{
    int iters = 0;
    for (; iters < max_attempts; iters++) {
        for (int i = 0; i < 7; i++) {
            int v, dup;
            do { v = (int)(cv_rng_next(rng) % (uint32_t)n); dup = 0; for (int k = 0; k < i; k++) if (idx[k] == v) dup = 1; } while (dup);
            idx[i] = v;
        }
        if (have_collinear(p1, idx, 7) || have_collinear(p2, idx, 7)) continue;
        break;
    }
    if (attempts) *attempts = iters + (iters < max_attempts);
    return iters < max_attempts;
}

/* test hook: the first `count` samples findFundamentalMat's RANSAC would draw for these points (7 indices each); returns how many exist */
int svo_oracle_ransac_samples(const float* p1, const float* p2, int n, int count, int* idx7)
{
    cv_rng rng; rng.state = 0xFFFFFFFFFFFFFFFFULL;
    int k = 0;
    if (n < 8) return 0;
    for (; k < count; k++) if (!ransac_get_subset(&rng, p1, p2, n, idx7 + 7 * k, 10000, NULL)) break;
    return k;
}

/* The 7-point algorithm (cv::findFundamentalMat's run7Point, what its RANSAC solves per sample: S4:202, 237): the seven epipolar
 * constraints leave a two-dimensional null space {f1, f2}; det(lambda (f1 - f2) + f2) = 0 is a cubic in lambda with one or three
 * real roots, each a rank-2 fundamental matrix.  OpenCV takes the null space from an SVD of the raw-pixel system and the roots
 * from Cardano's formulas (acos / cos / cbrt); the MODELS do not depend on either choice -- the set {F : F in the null space,
 * det F = 0} is what it is, in whatever basis and coordinates -- so they are computed here in a form the device can repeat bit for
 * bit: Hartley-normalised coordinates (conditioning only; the solutions map back exactly), Gauss-Jordan with full pivoting for the
 * null space, and for the cubic Newton's iteration from a guaranteed upper bound of its largest-magnitude root (convex side:
 * monotone, quadratic), deflation, and two polishing steps on the other two roots.  +, -, *, /, sqrt only, one IEEE operation per
 * operator.  Returns the number of models (1 or 3; 1..3 when the cubic's leading coefficient is exactly zero, see below), written to
 * Fm[3][9] in the order: root of largest magnitude (of the depressed cubic), then (-t1 + sqrt D) / 2, (-t1 - sqrt D) / 2.  A sample
 * whose linear system is rank-deficient yields NaN entries: no inliers, on both sides alike. */
static double cbrt_rough(double x)          /* x >= 0: within ~4 % of the cube root (exponent / 3 on the bit pattern: integer arithmetic) */
{
    union { double d; uint64_t u; } v; v.d = x;
    v.u = ((v.u >> 32) / 3u + 715094163u) << 32;
    return v.d;
}

static int seven_point(const float* p1, const float* p2, const int* s, double* Fm)
{
    double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
    for (int i = 0; i < 7; i++) { c1x += (double)p1[2 * s[i]]; c1y += (double)p1[2 * s[i] + 1]; c2x += (double)p2[2 * s[i]]; c2y += (double)p2[2 * s[i] + 1]; }
    c1x = c1x / 7.0; c1y = c1y / 7.0; c2x = c2x / 7.0; c2y = c2y / 7.0;
    double d1 = 0, d2 = 0;
    for (int i = 0; i < 7; i++) {
        double ax = (double)p1[2 * s[i]] - c1x, ay = (double)p1[2 * s[i] + 1] - c1y;
        double bx = (double)p2[2 * s[i]] - c2x, by = (double)p2[2 * s[i] + 1] - c2y;
        d1 += sqrt(ax * ax + ay * ay); d2 += sqrt(bx * bx + by * by);
    }
    const double s1 = 9.8994949366116654 / d1;    /* sqrt(2) / (d1 / 7) */
    const double s2 = 9.8994949366116654 / d2;
    double A[7][9];
    for (int i = 0; i < 7; i++) {
        double x1 = ((double)p1[2 * s[i]] - c1x) * s1, y1 = ((double)p1[2 * s[i] + 1] - c1y) * s1;
        double x2 = ((double)p2[2 * s[i]] - c2x) * s2, y2 = ((double)p2[2 * s[i] + 1] - c2y) * s2;
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2; A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2; A[i][6] = x1; A[i][7] = y1; A[i][8] = 1.0;
    }
    int perm[9]; for (int j = 0; j < 9; j++) perm[j] = j;
    for (int k = 0; k < 7; k++) {
        double best = -1.0; int pi = k, pj = k;
        for (int i = k; i < 7; i++) for (int j = k; j < 9; j++) { double v = fabs(A[i][j]); if (v > best) { best = v; pi = i; pj = j; } }
        if (pi != k) for (int j = 0; j < 9; j++) { double t = A[k][j]; A[k][j] = A[pi][j]; A[pi][j] = t; }
        if (pj != k) { for (int i = 0; i < 7; i++) { double t = A[i][k]; A[i][k] = A[i][pj]; A[i][pj] = t; } int t = perm[k]; perm[k] = perm[pj]; perm[pj] = t; }
        const double piv = A[k][k];
        for (int j = k; j < 9; j++) A[k][j] = A[k][j] / piv;
        for (int i = 0; i < 7; i++) { if (i == k) continue; const double f = A[i][k]; for (int j = k; j < 9; j++) A[i][j] = A[i][j] - f * A[k][j]; }
    }
    /* null space: the two free unknowns sit at positions 7 and 8; g = f1 - f2 (run7Point's "f1[i] -= f2[i]") */
    double g[9], f2[9];
    g[perm[7]] = 1.0; g[perm[8]] = -1.0; f2[perm[7]] = 0.0; f2[perm[8]] = 1.0;
    for (int i = 0; i < 7; i++) { f2[perm[i]] = -A[i][8]; g[perm[i]] = A[i][8] - A[i][7]; }
    /* det(lambda g + f2) = a3 lambda^3 + a2 lambda^2 + a1 lambda + a0, by cofactors along the first row */
    const double g00 = g[4] * g[8] - g[5] * g[7], g01 = g[3] * g[8] - g[5] * g[6], g02 = g[3] * g[7] - g[4] * g[6];
    const double h00 = f2[4] * f2[8] - f2[5] * f2[7], h01 = f2[3] * f2[8] - f2[5] * f2[6], h02 = f2[3] * f2[7] - f2[4] * f2[6];
    /* mixed minors: d/d(lambda) of the 2x2 minors at lambda = 0 */
    const double m00 = (g[4] * f2[8] + f2[4] * g[8]) - (g[5] * f2[7] + f2[5] * g[7]);
    const double m01 = (g[3] * f2[8] + f2[3] * g[8]) - (g[5] * f2[6] + f2[5] * g[6]);
    const double m02 = (g[3] * f2[7] + f2[3] * g[7]) - (g[4] * f2[6] + f2[4] * g[6]);
    const double a3 = (g[0] * g00 - g[1] * g01) + g[2] * g02;
    const double a0 = (f2[0] * h00 - f2[1] * h01) + f2[2] * h02;
    const double a2 = ((f2[0] * g00 - f2[1] * g01) + f2[2] * g02) + ((g[0] * m00 - g[1] * m01) + g[2] * m02);
    const double a1 = ((g[0] * h00 - g[1] * h01) + g[2] * h02) + ((f2[0] * m00 - f2[1] * m01) + f2[2] * m02);
    const double t1x = -(s1 * c1x), t1y = -(s1 * c1y), t2x = -(s2 * c2x), t2y = -(s2 * c2y);
    if (a3 == 0.0) {
        /* [oracle v7] The cubic has lost its leading term: cv::solveCubic's `a0 == 0` branch (run7Point hands it det(lambda f1' + f2);
         * it tests for EXACT zero) solves what is left as a quadratic or a linear equation.  det g = 0 also means that g itself --
         * lambda -> infinity in this parametrisation -- is a singular matrix of the null space; in another basis of the same null
         * space (OpenCV's comes from an SVD) that solution sits at a finite lambda, so it belongs to the model set and comes FIRST
         * here, followed by the finite roots in solveCubic's order.  Up to version 6 this case divided by zero (NaN models, no
         * inliers).  Measure-zero on real data; reachable with integer coordinates. */
        double lam[2]; int nq = 0;
        if (a2 == 0.0) { if (a1 != 0.0) { lam[0] = -a0 / a1; nq = 1; } }
        else {
            double d = a1 * a1 - (4.0 * a2) * a0;
            if (d >= 0.0) {
                const int two = d > 0.0;
                d = sqrt(d);
                const double q1 = (-a1 + d) * 0.5, q2 = (a1 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { lam[0] = q1 / a2; lam[1] = a0 / q1; } else { lam[0] = q2 / a2; lam[1] = a0 / q2; }
                nq = two ? 2 : 1;
            }
        }
        for (int k = 0; k <= nq; k++) {
            double f[9];
            for (int i = 0; i < 9; i++) f[i] = k == 0 ? g[i] : g[i] * lam[k - 1] + f2[i];
            double M[3][3];
            for (int r = 0; r < 3; r++) {
                M[r][0] = f[3 * r] * s1; M[r][1] = f[3 * r + 1] * s1;
                M[r][2] = (f[3 * r] * t1x + f[3 * r + 1] * t1y) + f[3 * r + 2];
            }
            double* F = Fm + 9 * k;
            for (int c = 0; c < 3; c++) {
                F[c] = s2 * M[0][c]; F[3 + c] = s2 * M[1][c];
                F[6 + c] = (t2x * M[0][c] + t2y * M[1][c]) + M[2][c];
            }
        }
        return nq + 1;
    }
    /* monic, then depressed: lambda = t - A / 3,  t^3 + p t + q = 0 */
    const double Am = a2 / a3, Bm = a1 / a3, Cm = a0 / a3;
    const double sh = Am / 3.0;
    const double p = Bm - Am * sh;
    const double q = ((2.0 * sh) * sh) * sh - sh * Bm + Cm;
    /* its root of largest magnitude has the sign of -q: u = |t| is the largest root of u^3 + p u - Q, Q = |q| */
    const double Q = fabs(q), pn = p < 0.0 ? -p : 0.0;
    double u = cbrt_rough(2.0 * Q);
    const double ub = sqrt(2.0 * pn);
    if (ub > u) u = ub;
    u = u * 1.1;                                   /* >= the root: u^3 = Q + |p| u <= 2 max(Q, |p| u) */
    for (int it = 0; it < 12; it++) {
        const double f = (u * u + p) * u - Q, d = (3.0 * u) * u + p;
        if (!(d > 0.0)) break;
        u = u - f / d;
    }
    const double t1 = q > 0.0 ? -u : u;
    const double disc = (-3.0 * t1) * t1 - 4.0 * p;
    double t[3]; int n = 1;
    t[0] = t1;
    if (disc >= 0.0) {
        const double sq = sqrt(disc);
        t[1] = (sq - t1) * 0.5; t[2] = (-sq - t1) * 0.5;
        for (int k = 1; k < 3; k++)
            for (int it = 0; it < 2; it++) {
                const double f = (t[k] * t[k] + p) * t[k] + q, d = (3.0 * t[k]) * t[k] + p;
                if (d != 0.0) t[k] = t[k] - f / d;
            }
        n = 3;
    }
    for (int k = 0; k < n; k++) {
        const double lam = t[k] - sh;
        double f[9];
        for (int i = 0; i < 9; i++) f[i] = g[i] * lam + f2[i];
        /* F = T2^T * f * T1 */
        double M[3][3];
        for (int r = 0; r < 3; r++) {
            M[r][0] = f[3 * r] * s1; M[r][1] = f[3 * r + 1] * s1;
            M[r][2] = (f[3 * r] * t1x + f[3 * r + 1] * t1y) + f[3 * r + 2];
        }
        double* F = Fm + 9 * k;
        for (int c = 0; c < 3; c++) {
            F[c] = s2 * M[0][c]; F[3 + c] = s2 * M[1][c];
            F[6 + c] = (t2x * M[0][c] + t2y * M[1][c]) + M[2][c];
        }
    }
    return n;
}

/* test hook: the models of the 7-point algorithm for correspondences 0..6 of the given lists */
int svo_oracle_seven_point(const float* p1, const float* p2, double* F27)
{
    const int s[7] = { 0, 1, 2, 3, 4, 5, 6 };
    return seven_point(p1, p2, s, F27);
}

/* symmetric point-to-epipolar-line error, max of the two squared distances */
static int fm_inlier(const double* F, float fx1, float fy1, float fx2, float fy2)
{
    const double x1 = (double)fx1, y1 = (double)fy1, x2 = (double)fx2, y2 = (double)fy2;
    double a = (F[0] * x1 + F[1] * y1) + F[2], b = (F[3] * x1 + F[4] * y1) + F[5], c = (F[6] * x1 + F[7] * y1) + F[8];
    const double sB = 1.0 / (a * a + b * b), dB = (x2 * a + y2 * b) + c;
    a = (F[0] * x2 + F[3] * y2) + F[6]; b = (F[1] * x2 + F[4] * y2) + F[7]; c = (F[2] * x2 + F[5] * y2) + F[8];
    const double sA = 1.0 / (a * a + b * b), dA = (x1 * a + y1 * b) + c;
    const double eA = (dA * dA) * sA, eB = (dB * dB) * sB;
    const double e = eA < eB ? eB : eA;                 /* std::max(d1*d1*s1, d2*d2*s2): a NaN first operand stays */
    return (float)e <= 1.0f;   /* findInliers: err is stored as float, threshold 1.0 px squared as float */
}

/* the same error as FMEstimatorCallback::computeError stores it: a float */
static float fm_error(const double* F, float fx1, float fy1, float fx2, float fy2)
{
    const double x1 = (double)fx1, y1 = (double)fy1, x2 = (double)fx2, y2 = (double)fy2;
    double a = (F[0] * x1 + F[1] * y1) + F[2], b = (F[3] * x1 + F[4] * y1) + F[5], c = (F[6] * x1 + F[7] * y1) + F[8];
    const double sB = 1.0 / (a * a + b * b), dB = (x2 * a + y2 * b) + c;
    a = (F[0] * x2 + F[3] * y2) + F[6]; b = (F[1] * x2 + F[4] * y2) + F[7]; c = (F[2] * x2 + F[5] * y2) + F[8];
    const double sA = 1.0 / (a * a + b * b), dA = (x1 * a + y1 * b) + c;
    const double eA = (dA * dA) * sA, eB = (dB * dB) * sB;
    return (float)(eA < eB ? eB : eA);
}

/* natural logarithm from IEEE +, -, *, / only, one operation per operator (the HIP kernels repeat it verbatim, so
 * both sides get the same bits; libm's and the device library's log may differ in the last place).  x normal, > 0.
 * ln x = e ln 2 + 2 atanh(s), s = (m - 1) / (m + 1), m in [sqrt(1/2), sqrt(2)]: |s| <= 0.1716, series to s^17. */
static double svo_ln(double x)
{
    union { double d; uint64_t u; } v; v.d = x;
    int e = (int)((v.u >> 52) & 0x7FF) - 1023;
    v.u = (v.u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 0.058823529411764705;                 /* 1/17 */
    p = p * s2 + 0.066666666666666666;               /* 1/15 */
    p = p * s2 + 0.076923076923076927;               /* 1/13 */
    p = p * s2 + 0.090909090909090912;               /* 1/11 */
    p = p * s2 + 0.1111111111111111;                 /* 1/9 */
    p = p * s2 + 0.14285714285714285;                /* 1/7 */
    p = p * s2 + 0.2;
    p = p * s2 + 0.33333333333333331;
    p = p * s2 + 1.0;
    return (double)e * 0.69314718055994529 + 2.0 * (s * p);
}

/* cv::RANSACUpdateNumIters(p = 0.99, ep = 1 - cnt / n, modelPoints = 7, maxIters) as OpenCV's ptsetreg.cpp writes it
 * [frozen; recalled, the source is not in /root/reference]: 0 when every point is an inlier, maxIters when the
 * estimate is no smaller, else cvRound(log(1 - p) / log(1 - (1 - ep)^modelPoints)). */
static int ransac_update_niters(int cnt, int n, int max_iters)
{
    const double w = (double)cnt / (double)n, w2 = w * w, w4 = w2 * w2, w7 = (w4 * w2) * w;
    const double denom = 1.0 - w7;
    if (denom < 2.2250738585072014e-308) return 0;
    const double num = -4.6051701859880909;          /* log(1 - 0.99) */
    const double d = svo_ln(denom);
    if (d >= 0.0 || -num >= (double)max_iters * (-d)) return max_iters;
    return (int)rint(num / d);
}

int svo_oracle_ransac_niters(int cnt, int n, int max_iters) { return ransac_update_niters(cnt, n, max_iters); }

/* cv::findFundamentalMat(FM_RANSAC, 1.0, 0.99) (RANSACPointSetRegistrator::run with modelPoints = 7): per iteration one minimal
 * sample -- drawn by OpenCV's own generator and rejection rules since v5 (cv::RNG seeded (uint64)-1 per call, getSubset, checkSubset) --,
 * every model the 7-point solver returns for it is scored in turn, a model with more inliers than any before (and than
 * modelPoints - 1) becomes the result and shrinks the iteration budget; the budget is tested once per iteration, so all models
 * of a sample are scored.  best_hyp = the SAMPLE the winning model came from, n_hyp_used = samples visited.
 * Exactly seven points: findFundamentalMat runs the 7-point kernel directly and sets the whole mask (no sampling; whichever model it
 * returns, seven inliers are below the eight the reference asks for at S4:205, 240).
 * Eight to fourteen points (since v6): findFundamentalMat hands the set to its LMedS registrator instead (`npoints >= 15` is the RANSAC's
 * condition, in OpenCV 2.4's cvFindFundamentalMat and in 3.x / 4.x's findFundamentalMat alike): lmeds_fundamental below. */
/* LMeDSPointSetRegistrator::run (modelPoints = 7, confidence 0.99, maxIters 1000): niters = RANSACUpdateNumIters(0.99, outlierRatio 0.45, 7, 1000)
 * = cvRound(ln 0.01 / ln(1 - 0.55^7)) = 300 samples, never shortened; the SAME generator and getSubset as the RANSAC (seed (uint64)-1, a sample
 * that cannot be found ends the loop, at iteration 0 the whole call); every model of a sample: the n errors as floats, their median = the
 * element nth_element leaves at n / 2 -- OpenCV sorts the float bit patterns AS INTS, which is the float order for the non-negative errors
 * here and puts x86's negative default NaN (0 * inf on a degenerate line) first: a NaN is given that pattern on every platform --; a model
 * whose median is below the smallest so far (strictly, starting from DBL_MAX: an infinite or NaN median never wins) becomes the result.
 * Then sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(minMedian), at least 0.001, and the mask is err <= (float)(sigma^2).
 * best_hyp = the sample of the winning model, n_hyp_used = samples visited.  The count returned is the mask's (S4:204 counts the mask;
 * findFundamentalMat itself returns no matrix below 7 inliers, which S4 never looks at). */
#define LMEDS_MAX_N 14
static int lmeds_niters(void)
{
    const double num = log(1.0 - 0.99), denom = log(1.0 - pow(1.0 - 0.45, 7.0));
    const int k = (denom >= 0.0 || -num >= 1000.0 * (-denom)) ? 1000 : (int)rint(num / denom);
    return k > 3 ? k : 3;
}
static int32_t float_bits_x86(float e)
{
    union { float f; int32_t i; } v; v.f = e;
    return e != e ? (int32_t)0xFFC00000u : v.i;
}
static int lmeds_fundamental(const float* p1, const float* p2, int n, uint8_t* mask, double* F9, int* best_hyp, int* n_hyp_used)
{
    const int niters = lmeds_niters();
    double minMedian = 1.7976931348623157e308, Fbest[9] = { 0 };
    int best_k = -1, k;
    cv_rng rng; rng.state = 0xFFFFFFFFFFFFFFFFULL;
    for (k = 0; k < niters; k++) {
        int s[7]; double Fm[27];
        if (!ransac_get_subset(&rng, p1, p2, n, s, 1000, NULL)) break;     /* [oracle v7] LMeDSPointSetRegistrator::run calls getSubset with its DEFAULT maxAttempts = 1000 (only the RANSAC's run passes 10000) */
        const int nm = seven_point(p1, p2, s, Fm);
        for (int j = 0; j < nm; j++) {
            const double* F = Fm + 9 * j;
            float err[LMEDS_MAX_N]; int32_t key[LMEDS_MAX_N];
            for (int i = 0; i < n; i++) { err[i] = fm_error(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]); key[i] = float_bits_x86(err[i]); }
            /* nth_element(int*, +n/2, +n): the element of rank n / 2 in the int order (ties share a value: which of them is immaterial) */
            float med = 0.f;
            for (int i = 0; i < n; i++) {
                int below = 0, equal = 0;
                for (int q = 0; q < n; q++) { below += key[q] < key[i]; equal += key[q] == key[i]; }
                if (below <= n / 2 && n / 2 < below + equal) { med = err[i]; break; }
            }
            const double median = (double)med;
            if (median < minMedian) { minMedian = median; best_k = k; memcpy(Fbest, F, 9 * sizeof(double)); }
        }
    }
    if (n_hyp_used) *n_hyp_used = k;
    if (getenv("SVO_ORACLE_TRACE")) fprintf(stderr, "lmeds n=%d visited=%d best_k=%d minMedian=%.9g\n", n, k, best_k, minMedian);
    if (best_k < 0) return 0;
    double sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (double)(n - 7)) * sqrt(minMedian);
    if (!(sigma > 0.001)) sigma = 0.001;                       /* MAX(sigma, 0.001) */
    const float t = (float)(sigma * sigma);
    int cnt = 0;
    for (int i = 0; i < n; i++) { mask[i] = (uint8_t)(fm_error(Fbest, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= t); cnt += mask[i]; }
    if (F9) memcpy(F9, Fbest, sizeof(Fbest));
    if (best_hyp) *best_hyp = best_k;
    return cnt;
}

int svo_oracle_ransac_fundamental(const float* p1, const float* p2, int n, uint8_t* mask, double* F9, int* best_hyp, int* n_hyp_used)
{
    for (int i = 0; i < n; i++) mask[i] = 0;
    if (best_hyp) *best_hyp = -1;
    if (n_hyp_used) *n_hyp_used = 0;
    if (n < 7) return 0;
    if (n == 7) {
        int s[7] = { 0, 1, 2, 3, 4, 5, 6 }; double Fm[27];
        const int nm = seven_point(p1, p2, s, Fm);
        for (int i = 0; i < n; i++) mask[i] = 1;
        if (F9 && nm > 0) memcpy(F9, Fm, 9 * sizeof(double));
        if (best_hyp) *best_hyp = 0;
        return 7;
    }
    if (n <= LMEDS_MAX_N) return lmeds_fundamental(p1, p2, n, mask, F9, best_hyp, n_hyp_used);
    int niters = RANSAC_MAX_HYP, best_cnt = 0, best_k = -1;
    double Fbest[9] = { 0 };
    cv_rng rng; rng.state = 0xFFFFFFFFFFFFFFFFULL;                 /* RNG rng((uint64)-1) */
    int k;
    for (k = 0; k < niters; k++) {
        int s[7]; double Fm[27];
        if (!ransac_get_subset(&rng, p1, p2, n, s, 10000, NULL)) break;        /* `if (!found) { if (iter == 0) return false; break; }` */
        const int nm = seven_point(p1, p2, s, Fm);
        for (int j = 0; j < nm; j++) {
            const double* F = Fm + 9 * j;
            int cnt = 0;
            for (int i = 0; i < n; i++) cnt += fm_inlier(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
            if (cnt > (best_cnt > 6 ? best_cnt : 6)) {
                best_cnt = cnt; best_k = k; memcpy(Fbest, F, 9 * sizeof(double));
                niters = ransac_update_niters(cnt, n, niters);      /* confidence 0.99 */
            }
        }
    }
    if (n_hyp_used) *n_hyp_used = k;
    if (getenv("SVO_ORACLE_TRACE")) fprintf(stderr, "ransac n=%d visited=%d best_k=%d best_cnt=%d budget=%d\n", n, k, best_k, best_cnt, niters);
    if (best_k < 0) return 0;
    int cnt = 0;
    for (int i = 0; i < n; i++) { mask[i] = (uint8_t)fm_inlier(Fbest, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]); cnt += mask[i]; }
    if (F9) memcpy(F9, Fbest, sizeof(Fbest));
    if (best_hyp) *best_hyp = best_k;
    return cnt;
}

/* ------------------------------------------------------------------------------------------------ */
/* stage 4: inter-frame tracking  (S4:71-801)                                                       */
/* ------------------------------------------------------------------------------------------------ */
/* svo_result.track_stats of the call under way (SVO_TS_*): filled by track_bf / track_win, summed over the octaves by
 * svo_oracle_process.  Thread-local: bench.py replays several oracle instances on several threads. */
static __thread int g_ts[8];
static int track_bf(int orb_th,
                    const svo_keypoint* pkl, const uint8_t* pdl, const svo_keypoint* pkr, const uint8_t* pdr, const svo_dmatch* pm, int npm,
                    const svo_keypoint* ckl, const uint8_t* cdl, const svo_keypoint* ckr, const uint8_t* cdr, const svo_dmatch* cm, int ncm,
                    svo_index_pair* out, int cap)
{
    if (npm <= 0 || ncm <= 0) return 0;
    /* S4:105-131 gather the descriptors of the matched features */
    uint8_t* preL = (uint8_t*)xmalloc((size_t)npm * 32), *preR = (uint8_t*)xmalloc((size_t)npm * 32);
    uint8_t* curL = (uint8_t*)xmalloc((size_t)ncm * 32), *curR = (uint8_t*)xmalloc((size_t)ncm * 32);
    for (int k = 0; k < npm; k++) { memcpy(preL + (size_t)k * 32, pdl + (size_t)pm[k].queryIdx * 32, 32); memcpy(preR + (size_t)k * 32, pdr + (size_t)pm[k].trainIdx * 32, 32); }
    for (int k = 0; k < ncm; k++) { memcpy(curL + (size_t)k * 32, cdl + (size_t)cm[k].queryIdx * 32, 32); memcpy(curR + (size_t)k * 32, cdr + (size_t)cm[k].trainIdx * 32, 32); }
    int32_t* tL = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)npm), *dL = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)npm);
    int32_t* tR = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)npm), *dR = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)npm);
    svo_oracle_hamming_bf(preL, npm, curL, ncm, tL, dL);                            /* S4:141 */
    svo_oracle_hamming_bf(preR, npm, curR, ncm, tR, dR);                            /* S4:142 */
    /* S4:145-160 joint sequential distance + collision filter (only KEPT entries mark) */
    uint8_t* ltm = (uint8_t*)xcalloc((size_t)ncm, 1), *rtm = (uint8_t*)xcalloc((size_t)ncm, 1);
    int* kq = (int*)xmalloc(sizeof(int) * (size_t)npm);
    int nk = 0;
    for (int k = 0; k < npm; k++) {
        if (!((float)dL[k] > (float)orb_th || (float)dR[k] > (float)orb_th)) g_ts[SVO_TS_THRESHOLD]++;
        if ((float)dL[k] > (float)orb_th || (float)dR[k] > (float)orb_th || ltm[tL[k]] || rtm[tR[k]]) continue;
        ltm[tL[k]] = rtm[tR[k]] = 1;
        kq[nk++] = k;
    }
    g_ts[SVO_TS_COLLISION] += nk;
    /* S4:171-241 fundamental matrix on left-left, then right-right pixel pairs */
    float* p1 = (float*)xmalloc(sizeof(float) * 2 * (size_t)(nk ? nk : 1)), *p2 = (float*)xmalloc(sizeof(float) * 2 * (size_t)(nk ? nk : 1));
    uint8_t* inL = (uint8_t*)xmalloc((size_t)(nk ? nk : 1)), *inR = (uint8_t*)xmalloc((size_t)(nk ? nk : 1));
    for (int i = 0; i < nk; i++) {
        int k = kq[i];
        const svo_keypoint* a = &pkl[pm[k].queryIdx], *b = &ckl[cm[tL[k]].queryIdx];   /* S4:183-189 */
        p1[2 * i] = a->x; p1[2 * i + 1] = a->y; p2[2 * i] = b->x; p2[2 * i + 1] = b->y;
    }
    int hypL = 0, hypR = 0;
    const int numInL = svo_oracle_ransac_fundamental(p1, p2, nk, inL, NULL, NULL, &hypL);  /* S4:202-205 */
    for (int i = 0; i < nk; i++) {
        int k = kq[i];
        const svo_keypoint* a = &pkr[pm[k].trainIdx], *b = &ckr[cm[tR[k]].trainIdx];   /* S4:218-224 */
        p1[2 * i] = a->x; p1[2 * i + 1] = a->y; p2[2 * i] = b->x; p2[2 * i + 1] = b->y;
    }
    const int numInR = svo_oracle_ransac_fundamental(p1, p2, nk, inR, NULL, NULL, &hypR);  /* S4:237-240 */
    const int goodFL = numInL >= 8, goodFR = numInR >= 8;
    g_ts[SVO_TS_INLIERS_L] += numInL; g_ts[SVO_TS_INLIERS_R] += numInR; g_ts[SVO_TS_HYP_L] += hypL; g_ts[SVO_TS_HYP_R] += hypR;
    int t = 0;
    for (int i = 0; i < nk; i++) {
        int k = kq[i];
        if (goodFL && goodFR && (inL[i] == 0 || inR[i] == 0)) continue;             /* S4:243-255 */
        g_ts[SVO_TS_BOTH_MASKS]++;
        if (tL[k] != tR[k]) continue;                                               /* S4:282 consistency */
        if (t < cap) { out[t].first = k; out[t].second = tL[k]; }                   /* S4:285 */
        t++;
    }
    g_ts[SVO_TS_TRACKED] += t < cap ? t : cap;
    free(preL); free(preR); free(curL); free(curR); free(tL); free(dL); free(tR); free(dR);
    free(ltm); free(rtm); free(kq); free(p1); free(p2); free(inL); free(inR);
    return t < cap ? t : cap;
}

/* ifmDescWin (S4:435-738) with its quirks kept */
static int track_win(const svo_params* p,
                     const svo_keypoint* pkl, const uint8_t* pdl, const svo_keypoint* pkr, const svo_dmatch* pm, int npm, const int64_t* pri,
                     const svo_keypoint* ckl, const uint8_t* cdl, const svo_keypoint* ckr, const svo_dmatch* cm, int ncm, const int64_t* cri,
                     int img_w, int img_h, svo_index_pair* out, int cap)
{
    const int WIN_W = p->ifm_win_w, WIN_H = p->ifm_win_h;                           /* S4:442-443 */
    const int awx = img_w - 1, awy = img_h - 1;                                     /* S4:489-490 */
    int* cmf = (int*)xmalloc(sizeof(int) * (size_t)(ncm > 0 ? ncm : 1));
    uint32_t* cms = (uint32_t*)xmalloc(sizeof(uint32_t) * (size_t)(ncm > 0 ? ncm : 1));
    for (int i = 0; i < ncm; i++) { cmf[i] = INVALID_IDX; cms[i] = 0xFFFFFFFFu; }   /* S4:509 */
    for (int y = 0; y < img_h - 1; y++) {                                           /* S4:514 */
        const int64_t p0 = pri[y], p1 = pri[y + 1];                                 /* S4:517-518 */
        if (p1 - p0 <= 0) continue;
        const int wy_min = (y - WIN_W) > 0 ? (y - WIN_W) : 0;                       /* S4:525 */
        const int wy_max = awy < (y + WIN_W) ? awy : (y + WIN_W);                   /* S4:526 */
        const int64_t c0 = cri[wy_min], c1 = cri[wy_max + 1];                       /* S4:529-530 */
        if (c1 - c0 <= 0) continue;
        for (int64_t pi = p0; pi < p1; pi++) {
            const svo_keypoint* pl = &pkl[pm[pi].queryIdx], *pr = &pkr[pm[pi].trainIdx];
            int64_t best_c = -1; uint8_t best_orb = 255;                            /* S4:543-545 */
            const int a = (int)(pl->x - (float)WIN_H), b = (int)(pl->x + (float)WIN_H);
            const int c = (int)(pr->x - (float)WIN_H), d = (int)(pr->x + (float)WIN_H);
            const int wxl0 = a > 0 ? a : 0, wxl1 = awx < b ? awx : b;               /* S4:552-553 */
            const int wxr0 = c > 0 ? c : 0, wxr1 = awx < d ? awx : d;               /* S4:554-555 */
            for (int64_t ci = c0; ci < c1; ci++) {                                  /* S4:557 */
                const svo_keypoint* fl = &ckl[cm[ci].queryIdx], *fr = &ckr[cm[ci].trainIdx];
                if (fl->x < (float)wxl0 || fl->x > (float)wxl1 || fr->x < (float)wxr0 || fr->x > (float)wxr1) continue;   /* S4:567 */
                uint8_t orb_l = 0;                                                  /* S4:596-609: left only, uint8_t wrap */
                for (int k = 0; k < SVO_DESC_BYTES; k++) {
                    uint8_t x_or = pdl[(size_t)pm[pi].queryIdx * 32 + k] ^ cdl[(size_t)cm[ci].queryIdx * 32 + k];
                    uint8_t count; for (count = 0; x_or; count++) x_or &= (uint8_t)(x_or - 1);
                    orb_l = (uint8_t)(orb_l + count);
                }
                if ((uint32_t)orb_l < (uint32_t)best_orb) { best_orb = orb_l; best_c = ci; }   /* S4:614-618 */
            }
            if (best_c >= 0) {                                                      /* S4:622-636 */
                const uint32_t bp = (uint32_t)best_orb;
                if (cmf[best_c] == INVALID_IDX) { cmf[best_c] = (int)pi; cms[best_c] = bp; }
                if (cmf[best_c] != INVALID_IDX && bp < cms[best_c]) { cmf[best_c] = (int)pi; cms[best_c] = bp; }
            }
        }
    }
    /* S4:640-679 survivors in ascending current index */
    int np = 0;
    for (int i = 0; i < ncm; i++) if (cmf[i] != INVALID_IDX) np++;
    float* l1 = (float*)xmalloc(sizeof(float) * 2 * (size_t)(np ? np : 1)), *l2 = (float*)xmalloc(sizeof(float) * 2 * (size_t)(np ? np : 1));
    float* r1 = (float*)xmalloc(sizeof(float) * 2 * (size_t)(np ? np : 1)), *r2 = (float*)xmalloc(sizeof(float) * 2 * (size_t)(np ? np : 1));
    svo_index_pair* pot = (svo_index_pair*)xmalloc(sizeof(svo_index_pair) * (size_t)(np ? np : 1));
    int j = 0;
    for (int i = 0; i < ncm; i++) if (cmf[i] != INVALID_IDX) {
        int pi = cmf[i];
        l1[2 * j] = pkl[pm[pi].queryIdx].x; l1[2 * j + 1] = pkl[pm[pi].queryIdx].y;
        r1[2 * j] = pkr[pm[pi].trainIdx].x; r1[2 * j + 1] = pkr[pm[pi].trainIdx].y;
        l2[2 * j] = ckl[cm[i].queryIdx].x; l2[2 * j + 1] = ckl[cm[i].queryIdx].y;
        r2[2 * j] = ckr[cm[i].trainIdx].x; r2[2 * j + 1] = ckr[cm[i].trainIdx].y;
        pot[j].first = pi; pot[j].second = i; j++;
    }
    uint8_t* inL = (uint8_t*)xmalloc((size_t)(np ? np : 1)), *inR = (uint8_t*)xmalloc((size_t)(np ? np : 1));
    /* (the HIP path evaluates both models unconditionally; the counters follow that: the right RANSAC's figures are reported
     * even when the left one found no model and the reference would not have run it -- the tracked pairs are the same) */
    int hypL = 0, hypR = 0;
    const int numInL = svo_oracle_ransac_fundamental(l1, l2, np, inL, NULL, NULL, &hypL);   /* S4:684-687 */
    const int numInR = svo_oracle_ransac_fundamental(r1, r2, np, inR, NULL, NULL, &hypR);   /* S4:696-699 */
    const int use_f = numInL >= 8 && numInR >= 8;
    g_ts[SVO_TS_THRESHOLD] += np; g_ts[SVO_TS_COLLISION] += np;
    g_ts[SVO_TS_INLIERS_L] += numInL; g_ts[SVO_TS_INLIERS_R] += numInR; g_ts[SVO_TS_HYP_L] += hypL; g_ts[SVO_TS_HYP_R] += hypR;
    int t = 0;
    for (int i = 0; i < np; i++) {                                                  /* S4:708-714 */
        if (use_f && (!inL[i] || !inR[i])) continue;
        if (t < cap) out[t] = pot[i];
        t++;
    }
    g_ts[SVO_TS_BOTH_MASKS] += t; g_ts[SVO_TS_TRACKED] += t < cap ? t : cap;
    free(cmf); free(cms); free(l1); free(l2); free(r1); free(r2); free(pot); free(inL); free(inR);
    return t < cap ? t : cap;
}

int svo_oracle_track(const svo_params* p, int orb_th,
                     const svo_keypoint* pkl, const uint8_t* pdl, const svo_keypoint* pkr, const uint8_t* pdr,
                     const svo_dmatch* pm, int npm, const int64_t* pri,
                     const svo_keypoint* ckl, const uint8_t* cdl, const svo_keypoint* ckr, const uint8_t* cdr,
                     const svo_dmatch* cm, int ncm, const int64_t* cri,
                     int img_w, int img_h, svo_index_pair* out, int cap)
{
    if (p->ifm_method == SVO_IFM_DESC_BF) return track_bf(orb_th, pkl, pdl, pkr, pdr, pm, npm, ckl, cdl, ckr, cdr, cm, ncm, out, cap);
    if (p->ifm_method == SVO_IFM_DESC_WIN) return track_win(p, pkl, pdl, pkr, pm, npm, pri, ckl, cdl, ckr, cm, ncm, cri, img_w, img_h, out, cap);
    return -2;   /* S4:740 THROW_EXCEPTION("Undefined inter-frame matching method") */
}
/* the same call with its counters (svo_result.track_stats' SVO_TS_* of this one octave) handed back: stats8 = 8 ints */
int svo_oracle_track_stats(const svo_params* p, int orb_th,
                           const svo_keypoint* pkl, const uint8_t* pdl, const svo_keypoint* pkr, const uint8_t* pdr,
                           const svo_dmatch* pm, int npm, const int64_t* pri,
                           const svo_keypoint* ckl, const uint8_t* cdl, const svo_keypoint* ckr, const uint8_t* cdr,
                           const svo_dmatch* cm, int ncm, const int64_t* cri,
                           int img_w, int img_h, svo_index_pair* out, int cap, int* stats8)
{
    int keep[8];
    memcpy(keep, g_ts, sizeof(keep));
    memset(g_ts, 0, sizeof(g_ts));
    const int t = svo_oracle_track(p, orb_th, pkl, pdl, pkr, pdr, pm, npm, pri, ckl, cdl, ckr, cdr, cm, ncm, cri, img_w, img_h, out, cap);
    if (stats8) memcpy(stats8, g_ts, sizeof(g_ts));
    memcpy(g_ts, keep, sizeof(keep));
    return t;
}

/* ------------------------------------------------------------------------------------------------ */
/* stage 5: projection + Jacobian  (S5:35-257)                                                      */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { double r[9]; double dr[3][9]; int small; } rot_t;

static void rodrigues_with_derivs(const double* dp, rot_t* R)
{
    const double w1 = dp[0], w2 = dp[1], w3 = dp[2];
    const double w12 = w1 * w1, w22 = w2 * w2, w32 = w3 * w3;
    const double tt = sqrt(w1 * w1 + w2 * w2 + w3 * w3);                            /* S5:55 */
    const double tt2 = tt * tt, tt3 = tt2 * tt, tt4 = tt3 * tt;
    const double sin_tt = sin(tt), cos_tt = cos(tt);
    double* r = R->r; double (*d)[9] = R->dr;
    memset(R, 0, sizeof(*R));
    if (tt < 1e-5) {                                                                /* S5:65-97 */
        R->small = 1;
        r[0] = 1; r[1] = -w3; r[2] = w2; r[3] = w3; r[4] = 1; r[5] = -w1; r[6] = -w2; r[7] = w1; r[8] = 1;
        d[2][1] = -1; d[1][2] = 1; d[2][3] = 1; d[0][5] = -1; d[1][6] = -1; d[0][7] = 1;
        return;
    }
    const double u = (cos_tt - 1) / tt2;                                            /* S5:102-110 */
    const double dudw1 = ((-sin_tt * w1 / tt) * tt2 - (cos_tt - 1) * 2 * w1) / tt4;
    const double dudw2 = ((-sin_tt * w2 / tt) * tt2 - (cos_tt - 1) * 2 * w2) / tt4;
    const double dudw3 = ((-sin_tt * w3 / tt) * tt2 - (cos_tt - 1) * 2 * w3) / tt4;
    const double v = sin_tt / tt;
    const double dvdw1 = w1 * (tt * cos_tt - sin_tt) / tt3;
    const double dvdw2 = w2 * (tt * cos_tt - sin_tt) / tt3;
    const double dvdw3 = w3 * (tt * cos_tt - sin_tt) / tt3;
    r[0] = (w22 + w32) * u + 1; r[1] = -w3 * v - w1 * w2 * u; r[2] = w2 * v - w1 * w3 * u;      /* S5:113-123 */
    r[3] = w3 * v - w1 * w2 * u; r[4] = (w12 + w32) * u + 1; r[5] = -w1 * v - w2 * w3 * u;
    r[6] = -w2 * v - w1 * w3 * u; r[7] = w1 * v - w2 * w3 * u; r[8] = (w12 + w22) * u + 1;
    /* first row S5:126-136 */
    d[0][0] = (w22 + w32) * dudw1; d[1][0] = 2 * w2 * u + (w22 + w32) * dudw2; d[2][0] = 2 * w3 * u + (w22 + w32) * dudw3;
    d[0][1] = -w3 * dvdw1 - (w2 * u + w1 * w2 * dudw1); d[1][1] = -w3 * dvdw2 - (w1 * u + w1 * w2 * dudw2); d[2][1] = -(v + w3 * dvdw3) - w1 * w2 * dudw3;
    d[0][2] = w2 * dvdw1 - (w3 * u + w1 * w3 * dudw1); d[1][2] = (v + w2 * dvdw2) - w1 * w3 * dudw2; d[2][2] = w2 * dvdw3 - (w1 * u + w1 * w3 * dudw3);
    /* second row S5:139-149 */
    d[0][3] = w3 * dvdw1 - (w2 * u + w1 * w2 * dudw1); d[1][3] = w3 * dvdw2 - (w1 * u + w1 * w2 * dudw2); d[2][3] = (v + w3 * dvdw3) - w1 * w2 * dudw3;
    d[0][4] = 2 * w1 * u + (w12 + w32) * dudw1; d[1][4] = (w12 + w32) * dudw2; d[2][4] = 2 * w3 * u + (w12 + w32) * dudw3;
    d[0][5] = -(v + w1 * dvdw1) - w2 * w3 * dudw1; d[1][5] = -w1 * dvdw2 - (w3 * u + w2 * w3 * dudw2); d[2][5] = -w1 * dvdw3 - (w2 * u + w2 * w3 * dudw3);
    /* third row S5:152-162 */
    d[0][6] = -w2 * dvdw1 - (w3 * u + w1 * w3 * dudw1); d[1][6] = -(v + w2 * dvdw2) - w1 * w3 * dudw2; d[2][6] = -w2 * dvdw3 - (w1 * u + w1 * w3 * dudw3);
    d[0][7] = (v + w1 * dvdw1) - w2 * w3 * dudw1; d[1][7] = w1 * dvdw2 - (w3 * u + w2 * w3 * dudw2); d[2][7] = w1 * dvdw3 - (w2 * u + w2 * w3 * dudw3);
    d[0][8] = 2 * w1 * u + (w12 + w22) * dudw1; d[1][8] = 2 * w2 * u + (w12 + w22) * dudw2; d[2][8] = (w22 + w32) * dudw3;   /* S5:162 as written */
}

static void project_one(const rot_t* R, const double* dp, const svo_stereo_camera* cam, const double* X, float* pix, double* J)
{
    const double* r = R->r;
    const double X1p = X[0], Y1p = X[1], Z1p = X[2];
    const double X1c = r[0] * X1p + r[1] * Y1p + r[2] * Z1p + dp[3];                /* S5:180-182 */
    const double Y1c = r[3] * X1p + r[4] * Y1p + r[5] * Z1p + dp[4];
    const double Z1c = r[6] * X1p + r[7] * Y1p + r[8] * Z1p + dp[5];
    const double X2c = X1c - cam->baseline;                                         /* S5:185 */
    pix[0] = (float)(cam->l_fx * X1c / Z1c + cam->l_cx);                            /* S5:189-193: stored as float */
    pix[1] = (float)(cam->l_fy * Y1c / Z1c + cam->l_cy);
    pix[2] = (float)(cam->r_fx * X2c / Z1c + cam->r_cx);
    pix[3] = (float)(cam->r_fy * Y1c / Z1c + cam->r_cy);
    for (int j = 0; j < 6; j++) {                                                   /* S5:201-255 */
        double X1cd, Y1cd, Z1cd;
        if (j < 3) {
            if (R->small) {
                if (j == 0) { X1cd = 0; Y1cd = -Z1p; Z1cd = Y1p; }
                else if (j == 1) { X1cd = Z1p; Y1cd = 0; Z1cd = -X1p; }
                else { X1cd = -Y1p; Y1cd = X1p; Z1cd = 0; }
            } else {
                const double* d = R->dr[j];
                X1cd = d[0] * X1p + d[1] * Y1p + d[2] * Z1p;
                Y1cd = d[3] * X1p + d[4] * Y1p + d[5] * Z1p;
                Z1cd = d[6] * X1p + d[7] * Y1p + d[8] * Z1p;
            }
        } else { X1cd = j == 3; Y1cd = j == 4; Z1cd = j == 5; }
        J[0 * 6 + j] = cam->l_fx * (X1cd * Z1c - X1c * Z1cd) / (Z1c * Z1c);         /* S5:251-254 */
        J[1 * 6 + j] = cam->l_fy * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
        J[2 * 6 + j] = cam->r_fx * (X1cd * Z1c - X2c * Z1cd) / (Z1c * Z1c);
        J[3 * 6 + j] = cam->r_fy * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
    }
}

void svo_oracle_project(const double* lmks3, int n, const svo_stereo_camera* cam, const double* delta6, float* pix, double* jac)
{
    rot_t R; rodrigues_with_derivs(delta6, &R);
    for (int i = 0; i < n; i++) project_one(&R, delta6, cam, lmks3 + 3 * i, pix + 4 * i, jac + 24 * i);
}

/* ------------------------------------------------------------------------------------------------ */
/* 6x6 solve  [frozen]  (Eigen::JacobiSVD(H).solve(g) + condition number; S5:375-388)               */
/* ------------------------------------------------------------------------------------------------ */
/* H is symmetric positive semi-definite, so its singular values are |eigenvalues|: cyclic Jacobi
 * eigen-decomposition, pseudo-inverse with Eigen's default rank threshold (6*eps*sigma_max). */
static int solve_sym6(const double* Hin, const double* g, double* x)
{
    double A[36], V[36];
    memcpy(A, Hin, sizeof(A));
    for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, dg = 0;
        for (int p = 0; p < 6; p++) { dg += A[p * 7] * A[p * 7]; for (int q = p + 1; q < 6; q++) off += A[p * 6 + q] * A[p * 6 + q]; }
        if (!(off > 1e-32 * dg)) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[p * 6 + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; k++) { const double akp = A[k * 6 + p], akq = A[k * 6 + q]; A[k * 6 + p] = c * akp - s * akq; A[k * 6 + q] = s * akp + c * akq; }
                for (int k = 0; k < 6; k++) { const double apk = A[p * 6 + k], aqk = A[q * 6 + k]; A[p * 6 + k] = c * apk - s * aqk; A[q * 6 + k] = s * apk + c * aqk; }
                for (int k = 0; k < 6; k++) { const double vkp = V[k * 6 + p], vkq = V[k * 6 + q]; V[k * 6 + p] = c * vkp - s * vkq; V[k * 6 + q] = s * vkp + c * vkq; }
            }
    }
    double smax = 0, smin = DBL_MAX; int has_nan = 0;
    for (int i = 0; i < 6; i++) { const double s = fabs(A[i * 7]); if (isnan(s)) has_nan = 1; if (s > smax) smax = s; if (s < smin) smin = s; }
    const double cond = smax / smin;                                                /* S5:379 */
    if (has_nan || isnan(cond)) return 0;                                           /* S5:380-386 */
    const double thr = 6.0 * DBL_EPSILON * smax;
    for (int i = 0; i < 6; i++) x[i] = 0;
    for (int i = 0; i < 6; i++) {
        const double lam = A[i * 7];
        if (!(fabs(lam) > thr)) continue;
        double dot = 0;
        for (int k = 0; k < 6; k++) dot += V[k * 6 + i] * g[k];
        const double coef = dot / lam;
        for (int k = 0; k < 6; k++) x[k] += coef * V[k * 6 + i];
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------ */
/* pose conversion  [frozen]  (CPose3D(CPose3DRotVec(delta).getInverse()); S5:717-718)              */
/* ------------------------------------------------------------------------------------------------ */
void svo_oracle_delta_to_pose(const double* dp, double* pose)
{
    const double w1 = dp[0], w2 = dp[1], w3 = dp[2];
    const double th = sqrt(w1 * w1 + w2 * w2 + w3 * w3);
    double R[9];
    if (th < 1e-10) {
        R[0] = 1; R[1] = -w3; R[2] = w2; R[3] = w3; R[4] = 1; R[5] = -w1; R[6] = -w2; R[7] = w1; R[8] = 1;
    } else {
        const double a = sin(th) / th, b = (1.0 - cos(th)) / (th * th);
        R[0] = 1 - b * (w2 * w2 + w3 * w3); R[1] = -a * w3 + b * w1 * w2; R[2] = a * w2 + b * w1 * w3;
        R[3] = a * w3 + b * w1 * w2; R[4] = 1 - b * (w1 * w1 + w3 * w3); R[5] = -a * w1 + b * w2 * w3;
        R[6] = -a * w2 + b * w1 * w3; R[7] = a * w1 + b * w2 * w3; R[8] = 1 - b * (w1 * w1 + w2 * w2);
    }
    /* inverse: Ri = R^T, ti = -R^T t */
    double Ri[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };
    pose[0] = -(Ri[0] * dp[3] + Ri[1] * dp[4] + Ri[2] * dp[5]);
    pose[1] = -(Ri[3] * dp[3] + Ri[4] * dp[4] + Ri[5] * dp[5]);
    pose[2] = -(Ri[6] * dp[3] + Ri[7] * dp[4] + Ri[8] * dp[5]);
    /* yaw-pitch-roll of R = Rz(yaw) Ry(pitch) Rx(roll) */
    const double pitch = atan2(-Ri[6], hypot(Ri[0], Ri[3]));
    double yaw, roll;
    if (fabs(Ri[7]) + fabs(Ri[8]) < 10 * DBL_EPSILON) {   /* gimbal lock */
        roll = 0.0;
        yaw = pitch > 0 ? atan2(Ri[5], Ri[2]) : atan2(-Ri[5], -Ri[2]);
    } else { roll = atan2(Ri[7], Ri[8]); yaw = atan2(Ri[3], Ri[0]); }
    pose[3] = yaw; pose[4] = pitch; pose[5] = roll;
}

/* ------------------------------------------------------------------------------------------------ */
/* getProjectedCoords  (H:175-182, C:415-466)                                                        */
/* ------------------------------------------------------------------------------------------------ */
/* delta = [rotation vector, translation] of the INVERSE of the pose (x y z yaw pitch roll): CPose3D::inverse() then
 * CPose3DRotVec(CPose3D) (C:456-461).  [frozen]: MRPT is absent; the log map goes through the unit quaternion
 * (Shepperd's branch selection), theta = 2 atan2(|v|, q0), which is well conditioned at 0 and at pi. */
void svo_oracle_pose_to_delta(const double* pose, double* dp)
{
    const double cy = cos(pose[3]), sy = sin(pose[3]), cp = cos(pose[4]), sp = sin(pose[4]), cr = cos(pose[5]), sr = sin(pose[5]);
    /* R = Rz(yaw) Ry(pitch) Rx(roll) */
    const double R[9] = { cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                          sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                          -sp, cp * sr, cp * cr };
    const double Ri[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };     /* inverse rotation */
    dp[3] = -(Ri[0] * pose[0] + Ri[1] * pose[1] + Ri[2] * pose[2]);
    dp[4] = -(Ri[3] * pose[0] + Ri[4] * pose[1] + Ri[5] * pose[2]);
    dp[5] = -(Ri[6] * pose[0] + Ri[7] * pose[1] + Ri[8] * pose[2]);
    double q0, q1, q2, q3;
    const double tr = Ri[0] + Ri[4] + Ri[8];
    if (tr > 0.0) { const double s = sqrt(tr + 1.0) * 2.0; q0 = 0.25 * s; q1 = (Ri[7] - Ri[5]) / s; q2 = (Ri[2] - Ri[6]) / s; q3 = (Ri[3] - Ri[1]) / s; }
    else if (Ri[0] > Ri[4] && Ri[0] > Ri[8]) { const double s = sqrt(1.0 + Ri[0] - Ri[4] - Ri[8]) * 2.0; q0 = (Ri[7] - Ri[5]) / s; q1 = 0.25 * s; q2 = (Ri[1] + Ri[3]) / s; q3 = (Ri[2] + Ri[6]) / s; }
    else if (Ri[4] > Ri[8]) { const double s = sqrt(1.0 + Ri[4] - Ri[0] - Ri[8]) * 2.0; q0 = (Ri[2] - Ri[6]) / s; q1 = (Ri[1] + Ri[3]) / s; q2 = 0.25 * s; q3 = (Ri[5] + Ri[7]) / s; }
    else { const double s = sqrt(1.0 + Ri[8] - Ri[0] - Ri[4]) * 2.0; q0 = (Ri[3] - Ri[1]) / s; q1 = (Ri[2] + Ri[6]) / s; q2 = (Ri[5] + Ri[7]) / s; q3 = 0.25 * s; }
    if (q0 < 0.0) { q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3; }
    const double vn = sqrt(q1 * q1 + q2 * q2 + q3 * q3);
    if (vn < 1e-12) { dp[0] = 2.0 * q1; dp[1] = 2.0 * q2; dp[2] = 2.0 * q3; }
    else { const double k = 2.0 * atan2(vn, q0) / vn; dp[0] = k * q1; dp[1] = k * q2; dp[2] = k * q3; }
}

/* tracked_first[m] != -1 marks pairing m as tracked elsewhere (skipped, C:430-431); the others are triangulated from
 * the previous frame (C:436-455, the formula of S5:529-544) and projected after the change in pose (C:464).
 * pix: 4 floats per emitted point (uL vL uR vR, the pair<TPixelCoordf,TPixelCoordf> of H:181).  Returns their number. */
int svo_oracle_projected_coords(const svo_dmatch* pre_matches, int n_pre, const svo_keypoint* pre_left, const svo_keypoint* pre_right,
                                const int32_t* tracked_first, const svo_stereo_camera* cam, const double* change_pose6, float* pix)
{
    double dp[6]; svo_oracle_pose_to_delta(change_pose6, dp);
    rot_t R; rodrigues_with_derivs(dp, &R);
    int nb = 0; double jac[24];
    for (int m = 0; m < n_pre; m++) {
        if (tracked_first[m] != -1) continue;
        const double ul = pre_left[pre_matches[m].queryIdx].x, vl = pre_left[pre_matches[m].queryIdx].y, ur = pre_right[pre_matches[m].trainIdx].x;
        const double cul = cam->l_cx, cvl = cam->l_cy, fl = cam->l_fx, cur = cam->r_cx, fr = cam->r_fx;
        const double disparity = fl * (cur - ur) + fr * (ul - cul);
        const double b_d = cam->baseline / disparity;
        const double X[3] = { b_d * fr * (ul - cul), b_d * fr * (vl - cvl), b_d * fl * fr };
        project_one(&R, dp, cam, X, pix + 4 * nb, jac);
        nb++;
    }
    return nb;
}

/* ------------------------------------------------------------------------------------------------ */
/* estimator state                                                                                   */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { svo_keypoint* kps; uint8_t* desc; int n; int64_t* row_index; int rows; } feat_set;
typedef struct { svo_dmatch* m; int n; int64_t* row_index; int rows; int64_t* ids; int n_ids; } pairing_t;
typedef struct { int present; int n_oct; int w[MAXOCT], h[MAXOCT]; feat_set f[2][MAXOCT]; pairing_t pr[MAXOCT]; } pair_data;

struct svo_oracle {
    svo_params p;
    int m_current_fast_th, m_current_orb_th;        /* C:35-36 */
    int m_error;                                    /* C:38 */
    int m_reset;                                    /* C:32 */
    int64_t m_last_match_ID, m_last_kf_max_id;      /* C:33 (m_last_kf_max_id uninitialised in the reference; 0 here) */
    int m_num_tracked_last_frame, m_num_tracked_last_kf;
    double m_last_computed_pose[6];                 /* C:49 */
    unsigned m_it_counter;
    pair_data* cur, *prev;
    svo_index_pair* tracked[MAXOCT]; int n_tracked[MAXOCT];
    double* residual; int n_residual;
    int32_t* outliers; int n_outliers;
};

static void free_pair(pair_data* d)
{
    if (!d) return;
    for (int o = 0; o < MAXOCT; o++) {
        for (int s = 0; s < 2; s++) { free(d->f[s][o].kps); free(d->f[s][o].desc); free(d->f[s][o].row_index); }
        free(d->pr[o].m); free(d->pr[o].row_index); free(d->pr[o].ids);
    }
    free(d);
}

svo_oracle* svo_oracle_create(void)
{
    svo_oracle* o = (svo_oracle*)xcalloc(1, sizeof(*o));
    svo_oracle_params_defaults(&o->p);
    o->m_current_fast_th = 20; o->m_current_orb_th = 60;    /* C:35-36 */
    o->m_error = SVO_VOEC_NONE;
    return o;
}

void svo_oracle_destroy(svo_oracle* o)
{
    if (!o) return;
    if (o->prev != o->cur) free_pair(o->prev);
    free_pair(o->cur);
    for (int i = 0; i < MAXOCT; i++) free(o->tracked[i]);
    free(o->residual); free(o->outliers); free(o);
}

void svo_oracle_set_params(svo_oracle* o, const svo_params* p)
{
    o->p = *p;
    o->m_current_fast_th = p->initial_FAST_threshold;       /* H:532, H:661 */
    o->m_current_orb_th = (int)p->orb_max_distance;         /* H:539, H:662 */
}
void svo_oracle_get_params(const svo_oracle* o, svo_params* p) { *p = o->p; }
void svo_oracle_set_fast_threshold(svo_oracle* o, int v) { int lo = o->p.fast_min_th, hi = o->p.fast_max_th; int m = v > lo ? v : lo; o->m_current_fast_th = hi < m ? hi : m; }
void svo_oracle_set_orb_threshold(svo_oracle* o, int v) { int lo = o->p.orb_min_th, hi = o->p.orb_max_th; int m = v > lo ? v : lo; o->m_current_orb_th = hi < m ? hi : m; }
int svo_oracle_get_fast_threshold(const svo_oracle* o) { return o->m_current_fast_th; }
int svo_oracle_get_orb_threshold(const svo_oracle* o) { return o->m_current_orb_th; }
void svo_oracle_reset_ids(svo_oracle* o) { o->m_reset = 1; }
void svo_oracle_set_this_frame_as_kf(svo_oracle* o)
{
    if (!o->cur || o->cur->pr[0].n_ids <= 0) return;
    int64_t mx = o->cur->pr[0].ids[0];
    for (int i = 1; i < o->cur->pr[0].n_ids; i++) if (o->cur->pr[0].ids[i] > mx) mx = o->cur->pr[0].ids[i];
    o->m_last_kf_max_id = mx;
}

/* ------------------------------------------------------------------------------------------------ */
/* stage 2 driver  (S2:385-671)                                                                      */
/* ------------------------------------------------------------------------------------------------ */
static int stage2_detect(svo_oracle* o, pair_data* d, int side, const uint8_t* const* oct_img, const int* oct_stride)
{
    const svo_params* p = &o->p;
    const int nOct = d->n_oct;
    size_t kps_to_detect[MAXOCT];
    kps_to_detect[0] = (size_t)((double)(size_t)p->orb_nfeats * (double)(2 * nOct) / (pow(2, nOct) - 1));   /* S2:405 */
    for (int oc = 1; oc < nOct; oc++) kps_to_detect[oc] = (size_t)round((double)kps_to_detect[0] / pow(2, oc));   /* S2:407 */
    for (int oc = 0; oc < nOct; oc++) {
        const int W = d->w[oc], H = d->h[oc];
        int cap; svo_keypoint* kv; uint8_t* dv; int n;
        if (p->detect_method == SVO_DM_ORB) {                                       /* S2:458-497 */
            const size_t nfe = p->non_maximal_suppression ? (size_t)(1.5 * (double)(size_t)p->orb_nfeats) : (size_t)p->orb_nfeats;   /* S2:461-464 */
            cap = (int)nfe + 16;
            kv = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)cap); dv = (uint8_t*)xmalloc((size_t)cap * 32);
            n = svo_oracle_orb_detect(oct_img[oc], W, H, oct_stride[oc], (int)nfe, p->orb_nlevels, o->m_current_fast_th, kv, dv, cap);
        } else if (p->detect_method == SVO_DM_FAST_ORB) {                           /* S2:502-515 */
            cap = W * H / 9 + 16;
            kv = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)cap); dv = (uint8_t*)xmalloc((size_t)cap * 32);
            n = svo_oracle_fast_orb_detect(oct_img[oc], W, H, oct_stride[oc], o->m_current_fast_th, kv, dv, cap);
        } else return -2;                                                           /* S2:578 (KLT/FASTER out of scope) */
        feat_set* fs = &d->f[side][oc];
        int32_t* order = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        int nk = n;
        if (p->non_maximal_suppression) {                                           /* S2:583-610 */
            if (p->nmsMethod == SVO_NMS_STANDARD) nk = svo_oracle_nms_copy(kv, n, p->min_distance, W, H, (int)kps_to_detect[oc], order);    /* S2:585-597 */
            else if (p->nmsMethod == SVO_NMS_ADAPTIVE) nk = svo_oracle_anms_copy(kv, n, (int)kps_to_detect[oc], 0.0, order);            /* S2:599-606 */
            else { free(order); free(kv); free(dv); return -2; }                                                                      /* S2:608 */
        } else for (int i = 0; i < n; i++) order[i] = i;                            /* S2:613-614 */
        svo_keypoint* k2 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(nk > 0 ? nk : 1));
        uint8_t* d2 = (uint8_t*)xmalloc((size_t)(nk > 0 ? nk : 1) * 32);
        for (int i = 0; i < nk; i++) { k2[i] = kv[order[i]]; memcpy(d2 + (size_t)i * 32, dv + (size_t)order[i] * 32, 32); }
        free(kv); free(dv);
        /* S2:618 m_update_indexes(order = true) */
        int32_t* ro = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)(nk > 0 ? nk : 1));
        fs->row_index = (int64_t*)xmalloc(sizeof(int64_t) * (size_t)H); fs->rows = H;
        svo_oracle_row_sort_index(k2, nk, H, ro, fs->row_index);
        fs->kps = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(nk > 0 ? nk : 1));
        fs->desc = (uint8_t*)xmalloc((size_t)(nk > 0 ? nk : 1) * 32);
        for (int i = 0; i < nk; i++) { fs->kps[i] = k2[ro[i]]; memcpy(fs->desc + (size_t)i * 32, d2 + (size_t)ro[i] * 32, 32); }
        fs->n = nk;
        free(k2); free(d2); free(ro); free(order);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* stage 5  (S5:392-736)                                                                             */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { const svo_keypoint* l1, *r1, *l2, *r2; } lists_t;

/* m_evalRGN (S5:275-390) */
static int eval_rgn(const svo_params* P, const svo_keypoint* l2, const svo_keypoint* r2, int nL, const uint8_t* mask,
                    const double* lmks, int n_non_masked, const double* deltaPose, const svo_stereo_camera* cam,
                    double* out_newPose, double* out_gradient, double* out_residual, int* residual_init, double* out_cost, int* out_error_code)
{
    (void)n_non_masked;
    if (!*residual_init) { for (int i = 0; i < nL; i++) out_residual[i] = DBL_MAX; *residual_init = 1; }   /* S5:296 */
    *out_cost = 0; *out_error_code = SVO_VOEC_NONE;
    double g[6] = { 0 }, H[36] = { 0 };
    rot_t R; rodrigues_with_derivs(deltaPose, &R);                                  /* S5:313 */
    const double b2 = P->use_robust_kernel ? P->kernel_param * P->kernel_param : 0; /* S5:316 */
    const double b2_1 = P->use_robust_kernel ? 1. / b2 : 0;                         /* S5:317 */
    for (int m = 0, i = 0; m < nL; ++m) {                                           /* S5:318 */
        if (!mask[m]) continue;
        float pix[4]; double J[24];
        project_one(&R, deltaPose, cam, lmks + 3 * i, pix, J);
        int good = 1;
        for (int k = 0; k < 24; k++) if (isnan(J[k]) || isinf(J[k])) good = 0;      /* S5:322, H:919-928 */
        if (!good) { ++i; continue; }
        const double rlx = (double)(l2[m].x - pix[0]);                              /* S5:335-338: float subtraction */
        const double rly = (double)(l2[m].y - pix[1]);
        const double rrx = (double)(r2[m].x - pix[2]);
        const double rry = (double)(r2[m].y - pix[3]);
        const double ri[4] = { rlx, rly, rrx, rry };
        const double s = rlx * rlx + rly * rly + rrx * rrx + rry * rry;             /* S5:344 */
        out_residual[m] = s;                                                        /* S5:345 */
        double rho_p = 1, fi;
        if (P->use_robust_kernel) { const double nn = sqrt(1 + (s * b2_1)); rho_p = 1 / nn; fi = b2 * (nn - 1); }   /* S5:351-356 */
        else fi = 0.5 * s;                                                          /* S5:359 */
        *out_cost += fi;
        for (int a = 0; a < 6; a++) {                                               /* S5:364-369: H NOT weighted by rho_p */
            double jr = 0;
            for (int k = 0; k < 4; k++) jr += J[k * 6 + a] * ri[k];
            g[a] += rho_p * jr;
            for (int b = 0; b < 6; b++) { double h = 0; for (int k = 0; k < 4; k++) h += J[k * 6 + a] * J[k * 6 + b]; H[a * 6 + b] += h; }
        }
        ++i;
    }
    memcpy(out_gradient, g, sizeof(g));
    if (!solve_sym6(H, g, out_newPose)) { *out_error_code = SVO_VOEC_BAD_COND_NUMBER; return 0; }   /* S5:375-388 */
    return 1;
}

static void triangulate(const svo_keypoint* l1, const svo_keypoint* r1, int nL, const uint8_t* surv, const svo_stereo_camera* cam, double* lmks)
{
    const double cul = cam->l_cx, cvl = cam->l_cy, fl = cam->l_fx, cur = cam->r_cx, fr = cam->r_fx, baseline = cam->baseline;   /* S5:510-515 */
    for (int m = 0, i = 0; m < nL; ++m) {                                           /* S5:530-544 */
        if (!surv[m]) continue;
        const double ul = l1[m].x, vl = l1[m].y, ur = r1[m].x;
        const double b_d = baseline / (fl * (cur - ur) + fr * (ul - cul));
        lmks[3 * i] = b_d * fr * (ul - cul); lmks[3 * i + 1] = b_d * fr * (vl - cvl); lmks[3 * i + 2] = b_d * fl * fr;
        ++i;
    }
}

/* stage5_optimize on gathered lists.  cur_match_idx[i] = tracked_pairs[..].second of point i */
static void stage5_core(svo_oracle* o, const svo_keypoint* l1, const svo_keypoint* r1, const svo_keypoint* l2, const svo_keypoint* r2,
                        const int32_t* cur_match_idx, int nL, int img_w, int img_h, const svo_stereo_camera* cam,
                        const double* initial_estimation, svo_result* result, double** res_out, int* nres_out, int32_t** outl_out, int* noutl_out)
{
    const svo_params* P = &o->p;
    uint8_t* survivors = (uint8_t*)xcalloc((size_t)(nL > 0 ? nL : 1), 1);
    svo_oracle_nms_mask(l1, nL, P->min_distance, img_w, img_h, nL, survivors);      /* S5:465-474 */
    double* out_residual = (double*)xmalloc(sizeof(double) * (size_t)(nL > 0 ? nL : 1));
    int residual_init = 0, n_res = 0;
    int32_t* outliers = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)(nL > 0 ? nL : 1));
    int n_outliers = 0;
    double pCost = 0, cCost = 0; int done = 0, abort_ = 0;
    double deltaPose[6] = { 0, 0, 0, 0, 0, 0 }, out_newPose[6] = { 0 }, out_grad[6];
    if (P->use_custom_initial_pose) { if (initial_estimation) memcpy(deltaPose, initial_estimation, sizeof(deltaPose)); }   /* S5:504-505 */
    else if (P->use_previous_pose_as_initial) memcpy(deltaPose, o->m_last_computed_pose, sizeof(deltaPose));   /* S5:506-507 */
    double* lmks = (double*)xmalloc(sizeof(double) * 3 * (size_t)(nL > 0 ? nL : 1));
    int n_non_masked = 0;
    for (int i = 0; i < nL; i++) n_non_masked += survivors[i];                      /* S5:520 */
    if (n_non_masked < 8) { result->valid = 0; goto finish; }                        /* S5:521-526 */
    triangulate(l1, r1, nL, survivors, cam, lmks);                                  /* S5:529-544 */
    {
        unsigned timesInc = 0; int out_error_code;
        result->num_it = 0;
        while (result->num_it < P->initial_max_iters && !done && !abort_) {         /* S5:549 */
            pCost = cCost;
            int cond = eval_rgn(P, l2, r2, nL, survivors, lmks, n_non_masked, deltaPose, cam, out_newPose, out_grad, out_residual, &residual_init, &cCost, &result->error_code);
            n_res = nL;
            if (!cond) { o->m_error = result->error_code; result->valid = 0; n_res = 0; goto finish; }   /* S5:569-573 (residual not swapped out) */
            for (int k = 0; k < 6; k++) deltaPose[k] += out_newPose[k];             /* S5:576-577 */
            if (result->num_it > 0) {                                               /* S5:580-596 */
                double m = 0; for (int c = 0; c < 6; c++) m += out_newPose[c] * out_newPose[c];
                done = sqrt(m) < P->min_mod_out_vector;
                if (pCost < cCost) { if (++timesInc > (unsigned)P->max_incr_cost) { result->error_code = SVO_VOEC_INCR_FUNC_COST_STG1; abort_ = 1; } }
            }
            result->num_it++;
        }
        for (int i = 0; i < n_res; ++i) {                                           /* S5:601-611 */
            if (out_residual[i] > P->residual_threshold) survivors[i] = 0;
            else outliers[n_outliers++] = cur_match_idx[i];                         /* "outliers" holds inliers */
        }
        n_non_masked = 0; for (int i = 0; i < nL; i++) n_non_masked += survivors[i];   /* S5:615 */
        if (n_non_masked < 8) { result->valid = 0; goto finish; }                    /* S5:616-621 */
        triangulate(l1, r1, nL, survivors, cam, lmks);                              /* S5:623-638 */
        done = 0; abort_ = 0;                                                       /* S5:645 */
        result->num_it_final = 0;
        while (result->num_it_final < P->max_iters && !done && !abort_) {           /* S5:650 */
            pCost = cCost;
            int cond = eval_rgn(P, l2, r2, nL, survivors, lmks, n_non_masked, deltaPose, cam, out_newPose, out_grad, out_residual, &residual_init, &cCost, &out_error_code);
            n_res = nL;
            if (!cond) { o->m_error = SVO_VOEC_BAD_COND_NUMBER; result->valid = 0; goto finish; }   /* S5:670-675; m_error set at S5:384 */
            for (int k = 0; k < 6; k++) deltaPose[k] += out_newPose[k];
            if (result->num_it_final > 0) {                                         /* S5:682-698 */
                double m = 0; for (int c = 0; c < 6; c++) m += out_newPose[c] * out_newPose[c];
                done = sqrt(m) < P->min_mod_out_vector;
                if (pCost < cCost) { if (++timesInc > (unsigned)P->max_incr_cost) { abort_ = 1; result->error_code = SVO_VOEC_INCR_FUNC_COST_STG2; } }
            }
            result->num_it_final++;
        }
        memcpy(result->delta, deltaPose, sizeof(deltaPose));
        svo_oracle_delta_to_pose(deltaPose, result->outPose);                       /* S5:717-718 */
        if (!P->use_custom_initial_pose && P->use_previous_pose_as_initial) memcpy(o->m_last_computed_pose, deltaPose, sizeof(deltaPose));   /* S5:720-721 */
        result->tracked_feats_from_last_frame = o->m_num_tracked_last_frame;        /* S5:724-725 */
        result->tracked_feats_from_last_KF = o->m_num_tracked_last_kf;
        result->valid = !abort_;                                                    /* S5:727 */
    }
finish:
    free(lmks); free(survivors);
    *res_out = out_residual; *nres_out = n_res; *outl_out = outliers; *noutl_out = n_outliers;
    result->n_residual = n_res; result->n_outliers = n_outliers;
}

static void store_stage5_outputs(svo_oracle* o, double* res, int nres, int32_t* outl, int noutl)
{
    free(o->residual); free(o->outliers);
    o->residual = res; o->n_residual = nres; o->outliers = outl; o->n_outliers = noutl;
}

int svo_oracle_change_in_pose(svo_oracle* o, const svo_index_pair* tracked, int n_tracked,
                              const svo_dmatch* pre_m, const svo_dmatch* cur_m,
                              const svo_keypoint* pre_l, const svo_keypoint* pre_r,
                              const svo_keypoint* cur_l, const svo_keypoint* cur_r,
                              const svo_stereo_camera* cam, const double* init6,
                              svo_result* res, double* residual, int32_t* outliers)
{
    /* C:355-413: single octave, image size from the camera record (C:402-403) */
    const int n = n_tracked;
    svo_keypoint* l1 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(n ? n : 1)), *r1 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(n ? n : 1));
    svo_keypoint* l2 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(n ? n : 1)), *r2 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(n ? n : 1));
    int32_t* cmi = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; i++) {                                                   /* S5:419-461, nOctaves == 1 */
        l1[i] = pre_l[pre_m[tracked[i].first].queryIdx]; r1[i] = pre_r[pre_m[tracked[i].first].trainIdx];
        l2[i] = cur_l[cur_m[tracked[i].second].queryIdx]; r2[i] = cur_r[cur_m[tracked[i].second].trainIdx];
        cmi[i] = tracked[i].second;
    }
    memset(res, 0, sizeof(*res));
    double* ro; int nro; int32_t* oo; int noo;
    stage5_core(o, l1, r1, l2, r2, cmi, n, cam->ncols, cam->nrows, cam, init6, res, &ro, &nro, &oo, &noo);
    if (residual) memcpy(residual, ro, sizeof(double) * (size_t)nro);
    if (outliers) memcpy(outliers, oo, sizeof(int32_t) * (size_t)noo);
    free(ro); free(oo); free(l1); free(r1); free(l2); free(r2); free(cmi);
    return res->valid;
}

/* ------------------------------------------------------------------------------------------------ */
/* processNewImagePair  (P:41-385)                                                                  */
/* ------------------------------------------------------------------------------------------------ */
int svo_oracle_process(svo_oracle* o, const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                       const svo_stereo_camera* cam, int repeat, svo_result* result)
{
    const svo_params* p = &o->p;
    memset(result, 0, sizeof(*result));
    result->error_code = SVO_VOEC_NONE;                                             /* P:50 */
    if (p->detect_method < 0 || p->detect_method > 3) return -2;                    /* P:54-61 */
    if (p->match_method < 0 || p->match_method > 2) return -2;                      /* P:62-68 */
    if (p->ifm_method < 0 || p->ifm_method > 3) return -2;                          /* P:69-76 */
    if (!left || !right) return -3;                                                 /* P:81 */
    /* P:86-89 shift unless repeating or recovering from a tracking / conditioning failure */
    if (!repeat && o->m_error != SVO_VOEC_BAD_TRACKING && o->m_error != SVO_VOEC_BAD_COND_NUMBER) {
        if (o->prev && o->prev != o->cur) free_pair(o->prev);
        o->prev = o->cur; o->cur = NULL;
    }
    o->m_error = SVO_VOEC_NONE;                                                     /* P:95 */
    if (o->cur && o->cur != o->prev) free_pair(o->cur);
    pair_data* cur = (pair_data*)xcalloc(1, sizeof(pair_data));                     /* P:100 */
    o->cur = cur; cur->present = 1;
    /* stage 1 substitute: inputs are rectified gray; build the octave pyramid (S1:80-83) */
    const int nOct = p->detect_method == SVO_DM_ORB ? 1 : (p->nOctaves < 1 ? 1 : (p->nOctaves > MAXOCT ? MAXOCT : p->nOctaves));
    cur->n_oct = nOct;
    const uint8_t* oimg[2][MAXOCT]; int ostride[2][MAXOCT]; uint8_t* obuf[2][MAXOCT];
    for (int s = 0; s < 2; s++) {
        oimg[s][0] = s ? right : left; ostride[s][0] = stride; obuf[s][0] = NULL;
        cur->w[0] = w; cur->h[0] = h;
        for (int oc = 1; oc < nOct; oc++) {
            cur->w[oc] = cur->w[oc - 1] / 2; cur->h[oc] = cur->h[oc - 1] / 2;
            obuf[s][oc] = (uint8_t*)xmalloc((size_t)cur->w[oc] * cur->h[oc]);
            svo_oracle_half_smooth(oimg[s][oc - 1], cur->w[oc - 1], cur->h[oc - 1], ostride[s][oc - 1], obuf[s][oc]);
            oimg[s][oc] = obuf[s][oc]; ostride[s][oc] = cur->w[oc];
        }
    }
    int rc = 0;
    rc = stage2_detect(o, cur, 0, oimg[0], ostride[0]);                             /* P:166 */
    if (!rc) rc = stage2_detect(o, cur, 1, oimg[1], ostride[1]);                    /* P:167 */
    for (int s = 0; s < 2; s++) for (int oc = 1; oc < nOct; oc++) free(obuf[s][oc]);
    if (rc) return rc;
    result->n_octaves = nOct;
    for (int oc = 0; oc < nOct; oc++) { result->detected_left[oc] = cur->f[0][oc].n; result->detected_right[oc] = cur->f[1][oc].n; }   /* P:171-176 */
    /* P:254-267 reset of the match IDs */
    if (o->m_reset) {
        o->m_last_match_ID = 0;
        if (o->prev) for (int oc = 0; oc < nOct; oc++) for (int m = 0; m < o->prev->pr[oc].n_ids; m++) o->prev->pr[oc].ids[m] = o->m_last_match_ID++;
        o->m_reset = 0;
        o->m_last_kf_max_id = o->m_last_match_ID - 1;
    }
    /* stage 3 (P:269) */
    const int use_ids = p->vo_use_matches_ids && !(o->prev && o->prev->present);    /* S3:67 */
    for (int oc = 0; oc < nOct; oc++) {
        feat_set* fl = &cur->f[0][oc], *fr = &cur->f[1][oc];
        pairing_t* pr = &cur->pr[oc];
        pr->m = (svo_dmatch*)xmalloc(sizeof(svo_dmatch) * (size_t)(fl->n > 0 ? fl->n : 1));
        pr->row_index = (int64_t*)xmalloc(sizeof(int64_t) * (size_t)(cur->h[oc] + 1)); pr->rows = cur->h[oc] + 1;
        int m = svo_oracle_match_lr(p, o->m_current_orb_th, fl->kps, fl->desc, fl->n, fl->row_index, fr->kps, fr->desc, fr->n, fr->row_index,
                                    cur->w[oc], cur->h[oc], pr->m, fl->n, pr->row_index);
        if (m < 0) return m;
        pr->n = m;
        pr->ids = (int64_t*)xcalloc((size_t)(m > 0 ? m : 1), sizeof(int64_t)); pr->n_ids = 0;
        if (use_ids) { for (int i = 0; i < m; i++) pr->ids[i] = o->m_last_match_ID++; pr->n_ids = m; }   /* S3:172-173 */
        result->stereo_matches[oc] = m;                                             /* P:274-276 */
    }
    for (int oc = 0; oc < MAXOCT; oc++) { free(o->tracked[oc]); o->tracked[oc] = NULL; o->n_tracked[oc] = 0; }
    store_stage5_outputs(o, NULL, 0, NULL, 0);
    if (o->prev && o->prev->present) {                                              /* P:305 */
        pair_data* prev = o->prev;
        o->m_num_tracked_last_frame = 0; o->m_num_tracked_last_kf = 0;              /* S4:743 */
        memset(g_ts, 0, sizeof(g_ts));
        for (int oc = 0; oc < nOct; oc++) {                                         /* P:314 stage4_track */
            pairing_t* pp = &prev->pr[oc], *cp = &cur->pr[oc];
            int cap = pp->n > 0 ? pp->n : 1;
            if (cp->n > cap) cap = cp->n;
            o->tracked[oc] = (svo_index_pair*)xmalloc(sizeof(svo_index_pair) * (size_t)cap);
            int t = svo_oracle_track(p, o->m_current_orb_th,
                                     prev->f[0][oc].kps, prev->f[0][oc].desc, prev->f[1][oc].kps, prev->f[1][oc].desc, pp->m, pp->n, pp->row_index,
                                     cur->f[0][oc].kps, cur->f[0][oc].desc, cur->f[1][oc].kps, cur->f[1][oc].desc, cp->m, cp->n, cp->row_index,
                                     cur->w[oc], cur->h[oc], o->tracked[oc], cap);
            if (t < 0) return t;
            o->n_tracked[oc] = t;
            if (p->vo_use_matches_ids) {                                            /* S4:268-305 / 716-733 */
                uint8_t* ct = (uint8_t*)xcalloc((size_t)(cp->n > 0 ? cp->n : 1), 1);
                cp->n_ids = cp->n;
                for (int k = 0; k < t; k++) {
                    int pi = o->tracked[oc][k].first, ci = o->tracked[oc][k].second;
                    cp->ids[ci] = pi < pp->n_ids ? pp->ids[pi] : 0; ct[ci] = 1;
                }
                for (int k = 0; k < cp->n; k++) if (!ct[k]) cp->ids[k] = o->m_last_match_ID++;
                free(ct);
            }
            o->m_num_tracked_last_frame += t;                                       /* S4:746 */
            for (int k = 0; k < cp->n_ids; k++) if (cp->ids[k] <= o->m_last_kf_max_id) o->m_num_tracked_last_kf++;   /* S4:747-751 */
        }
        for (int k = 0; k < 8; k++) result->track_stats[k] = g_ts[k];
        if (o->m_num_tracked_last_frame < p->bad_tracking_th) {                     /* P:326-330 */
            o->m_error = result->error_code = SVO_VOEC_BAD_TRACKING;
        }
        if (o->m_error != SVO_VOEC_BAD_TRACKING) {                                  /* P:332-341 stage5_optimize */
            int nT = o->m_num_tracked_last_frame;
            svo_keypoint* l1 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(nT ? nT : 1)), *r1 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(nT ? nT : 1));
            svo_keypoint* l2 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(nT ? nT : 1)), *r2 = (svo_keypoint*)xmalloc(sizeof(svo_keypoint) * (size_t)(nT ? nT : 1));
            int32_t* cmi = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)(nT ? nT : 1));
            int pc = 0;
            for (int oc = 0; oc < nOct; oc++) {                                     /* S5:419-461 */
                const float scale_norm = (float)(size_t)(nOct > 1 ? pow(2, oc) : 1);
                for (int i = 0; i < o->n_tracked[oc]; i++, pc++) {
                    const svo_dmatch* pm = &prev->pr[oc].m[o->tracked[oc][i].first], *cm = &cur->pr[oc].m[o->tracked[oc][i].second];
                    l1[pc] = prev->f[0][oc].kps[pm->queryIdx]; r1[pc] = prev->f[1][oc].kps[pm->trainIdx];
                    l2[pc] = cur->f[0][oc].kps[cm->queryIdx]; r2[pc] = cur->f[1][oc].kps[cm->trainIdx];
                    if (nOct > 1) { l1[pc].x *= scale_norm; l1[pc].y *= scale_norm; r1[pc].x *= scale_norm; r1[pc].y *= scale_norm;
                                    l2[pc].x *= scale_norm; l2[pc].y *= scale_norm; r2[pc].x *= scale_norm; r2[pc].y *= scale_norm; }
                    cmi[pc] = o->tracked[oc][i].second;
                }
            }
            double* ro; int nro; int32_t* oo; int noo;
            stage5_core(o, l1, r1, l2, r2, cmi, nT, prev->w[0], prev->h[0], cam, NULL, result, &ro, &nro, &oo, &noo);   /* S5:465-468 */
            store_stage5_outputs(o, ro, nro, oo, noo);
            free(l1); free(r1); free(l2); free(r2); free(cmi);
        }
    } else {
        result->error_code = SVO_VOEC_FIRST_ITERATION; result->valid = 0;           /* P:348-352 */
    }
    if (!repeat) ++o->m_it_counter;                                                 /* P:380-381 */
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* getters                                                                                            */
/* ------------------------------------------------------------------------------------------------ */
static const pair_data* pick(const svo_oracle* o, int which) { return which ? o->prev : o->cur; }

int svo_oracle_get_keypoints(const svo_oracle* o, int which, int side, int octave, svo_keypoint* kps, uint8_t* desc, int cap)
{
    const pair_data* d = pick(o, which); if (!d || octave >= d->n_oct) return 0;
    const feat_set* f = &d->f[side][octave]; int n = f->n < cap ? f->n : cap;
    if (kps) memcpy(kps, f->kps, sizeof(svo_keypoint) * (size_t)n);
    if (desc) memcpy(desc, f->desc, (size_t)n * 32);
    return f->n;
}
int svo_oracle_get_row_index(const svo_oracle* o, int which, int side, int octave, int64_t* idx, int cap)
{
    const pair_data* d = pick(o, which); if (!d || octave >= d->n_oct) return 0;
    const feat_set* f = &d->f[side][octave]; int n = f->rows < cap ? f->rows : cap;
    if (idx) memcpy(idx, f->row_index, sizeof(int64_t) * (size_t)n);
    return f->rows;
}
int svo_oracle_get_matches(const svo_oracle* o, int which, int octave, svo_dmatch* m, int cap)
{
    const pair_data* d = pick(o, which); if (!d || octave >= d->n_oct) return 0;
    int n = d->pr[octave].n < cap ? d->pr[octave].n : cap;
    if (m) memcpy(m, d->pr[octave].m, sizeof(svo_dmatch) * (size_t)n);
    return d->pr[octave].n;
}
int svo_oracle_get_matches_row_index(const svo_oracle* o, int which, int octave, int64_t* idx, int cap)
{
    const pair_data* d = pick(o, which); if (!d || octave >= d->n_oct) return 0;
    int n = d->pr[octave].rows < cap ? d->pr[octave].rows : cap;
    if (idx) memcpy(idx, d->pr[octave].row_index, sizeof(int64_t) * (size_t)n);
    return d->pr[octave].rows;
}
int svo_oracle_get_match_ids(const svo_oracle* o, int which, int octave, int64_t* ids, int cap)
{
    const pair_data* d = pick(o, which); if (!d || octave >= d->n_oct) return 0;
    int n = d->pr[octave].n_ids < cap ? d->pr[octave].n_ids : cap;
    if (ids) memcpy(ids, d->pr[octave].ids, sizeof(int64_t) * (size_t)n);
    return d->pr[octave].n_ids;
}
int svo_oracle_get_tracked(const svo_oracle* o, int octave, svo_index_pair* t, int cap)
{
    if (octave >= MAXOCT) return 0;
    int n = o->n_tracked[octave] < cap ? o->n_tracked[octave] : cap;
    if (t && n > 0) memcpy(t, o->tracked[octave], sizeof(svo_index_pair) * (size_t)n);
    return o->n_tracked[octave];
}
int svo_oracle_get_residuals(const svo_oracle* o, double* r, int cap)
{
    int n = o->n_residual < cap ? o->n_residual : cap;
    if (r && n > 0) memcpy(r, o->residual, sizeof(double) * (size_t)n);
    return o->n_residual;
}
int svo_oracle_get_outliers(const svo_oracle* o, int32_t* idx, int cap)
{
    int n = o->n_outliers < cap ? o->n_outliers : cap;
    if (idx && n > 0) memcpy(idx, o->outliers, sizeof(int32_t) * (size_t)n);
    return o->n_outliers;
}
