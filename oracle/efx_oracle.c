/*
 * efx_oracle.c -- CPU ORACLE (test infrastructure only; see efx_oracle.h for the parity status:
 * "parity unpinned").  Plain C99, single-threaded like the reference CPU module
 * (serial loops at bad.cpp:341 and hash_sift.cpp:343).
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fno-fast-math  (no FMA contraction: DESIGN.md S8).
 * All citations are relative to /root/reference/.
 */
#include "efx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * OpenCV scalar helpers the reference relies on (semantics of cvRound/cvFloor/saturate_cast):
 * cvRound = round half to even (SSE cvtss2si / lrint), cvFloor = floor.
 * ---------------------------------------------------------------------------------------------- */
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

/* Optional OpenMP over rows / candidates / keypoints for the multi-core CPU baseline: every loop that is
 * parallelised writes disjoint outputs, so the results do not depend on the thread count.  1 = as the reference. */
static int g_threads = 1;
void efxo_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int efxo_get_threads(void) { return g_threads; }
#define EFXO_PAR _Pragma("omp parallel for schedule(dynamic, 16) num_threads(g_threads)")
static int cv_floor_f(float v) { return (int)floorf(v); }
static uint8_t sat_u8_f(float v)
{
    int iv = cv_round_f(v);
    return (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

/* ================================================================================================
 * BAD  (modules/efficient_features/src/bad.cpp)
 * ============================================================================================== */

void efxo_integral(const uint8_t* img, int rows, int cols, int stride, int32_t* out)
{
    /* cv::integral(CV_8U -> CV_32S): out[(y+1)][(x+1)] = sum_{v<=y,u<=x} img[v][u]  (bad.cpp:286).
     * CV_32S wraps for frames above ~8.4 Mpx of bright pixels (8K); sums are taken mod 2^32 (unsigned
     * arithmetic, defined behaviour) so that box differences stay exact, as they do in OpenCV. */
    const int ow = cols + 1;
    uint32_t* o32 = (uint32_t*)out;
    memset(out, 0, sizeof(int32_t) * (size_t)ow);
    for (int y = 0; y < rows; y++) {
        uint32_t* o = o32 + (size_t)(y + 1) * ow;
        const uint32_t* up = o32 + (size_t)y * ow;
        const uint8_t* p = img + (size_t)y * stride;
        uint32_t run = 0;
        o[0] = 0;
        for (int x = 0; x < cols; x++) {
            run += p[x];
            o[x + 1] = up[x + 1] + run;
        }
    }
}

typedef struct { int x1, x2, y1, y2, r; } box_t; /* BoxPairParams, bad.cpp:39-42 */

#define BAD_ROUNDNUM(x) ((int)((x) + 0.5f))      /* bad.cpp:28 */
#define BAD_DEG2RAD 0.017453292519943295         /* bad.cpp:29 (double) */
#define BAD_EXTRA_MARGIN 1.75f                   /* bad.cpp:30 */

/* isKeypointInTheBorder, bad.cpp:86-103 (patch 32x32) */
static int bad_in_border(float x, float y, float size, int img_w, int img_h, float scale_factor)
{
    const float s = scale_factor * size / (float)(32 + 32);
    const float bw = (float)32 * s * BAD_EXTRA_MARGIN;
    const float bh = (float)32 * s * BAD_EXTRA_MARGIN;
    if (x < bw || x + bw >= (float)img_w) return 1;
    if (y < bh || y + bh >= (float)img_h) return 1;
    return 0;
}

/* rectifyBoxes, bad.cpp:115-157 */
static void bad_rectify(const int32_t* boxes, int nbits, box_t* out, float x, float y, float size, float angle,
                        float scale_factor)
{
    float m00, m01, m02, m10, m11, m12;
    const float s = scale_factor * size / (0.5f * (float)(32 + 32));
    if (angle == -1) {
        m00 = s;
        m01 = 0.0f;
        m02 = -0.5f * s * (float)32 + x;
        m10 = 0.0f;
        m11 = s;
        m12 = -s * 0.5f * (float)32 + y;
    } else {
        /* float angle * double constant -> double cos/sin -> float (bad.cpp:138-139) */
        const float cosine = (angle >= 0) ? (float)cos(angle * BAD_DEG2RAD) : 1.f;
        const float sine = (angle >= 0) ? (float)sin(angle * BAD_DEG2RAD) : 0.f;
        m00 = s * cosine;
        m01 = -s * sine;
        m02 = (-s * cosine + s * sine) * (float)32 * 0.5f + x;
        m10 = s * sine;
        m11 = s * cosine;
        m12 = (-s * sine - s * cosine) * (float)32 * 0.5f + y;
    }
    for (int i = 0; i < nbits; i++) {
        const float bx1 = (float)boxes[5 * i + 0], bx2 = (float)boxes[5 * i + 1];
        const float by1 = (float)boxes[5 * i + 2], by2 = (float)boxes[5 * i + 3];
        out[i].x1 = BAD_ROUNDNUM(m00 * bx1 + m01 * by1 + m02);
        out[i].y1 = BAD_ROUNDNUM(m10 * bx1 + m11 * by1 + m12);
        out[i].x2 = BAD_ROUNDNUM(m00 * bx2 + m01 * by2 + m02);
        out[i].y2 = BAD_ROUNDNUM(m10 * bx2 + m11 * by2 + m12);
        out[i].r = BAD_ROUNDNUM(s * (float)boxes[5 * i + 4]);
    }
}

static void clamp_box(int cx, int cy, int r, int fw, int fh, int* x1, int* y1, int* x2, int* y2)
{
    /* bad.cpp:180-200 (fw/fh are the integral image's cols/rows) */
    int a = cx - r;
    if (a < 0) a = 0; else if (a >= fw - 1) a = fw - 2;
    int b = cy - r;
    if (b < 0) b = 0; else if (b >= fh - 1) b = fh - 2;
    int c = cx + r + 1;
    if (c <= 0) c = 1; else if (c >= fw) c = fw - 1;
    int d = cy + r + 1;
    if (d <= 0) d = 1; else if (d >= fh) d = fh - 1;
    *x1 = a; *y1 = b; *x2 = c; *y2 = d;
}

/* computeBadResponse, bad.cpp:166-251 */
static float bad_response_clamped(const box_t* bp, const int32_t* I, int fw, int fh)
{
    int x1, y1, x2, y2;
    clamp_box(bp->x1, bp->y1, bp->r, fw, fh, &x1, &y1, &x2, &y2);
    const uint32_t* U = (const uint32_t*)I;     /* box sums mod 2^32: exact whenever the true sum fits */
    uint32_t A = U[(size_t)y1 * fw + x1], B = U[(size_t)y1 * fw + x2];
    uint32_t C = U[(size_t)y2 * fw + x1], D = U[(size_t)y2 * fw + x2];
    const float sum1 = (float)(int32_t)(A + D - B - C);
    const int area1 = (y2 - y1) * (x2 - x1);
    const float avg1 = sum1 / (float)area1;

    clamp_box(bp->x2, bp->y2, bp->r, fw, fh, &x1, &y1, &x2, &y2);
    A = U[(size_t)y1 * fw + x1]; B = U[(size_t)y1 * fw + x2];
    C = U[(size_t)y2 * fw + x1]; D = U[(size_t)y2 * fw + x2];
    const float sum2 = (float)(int32_t)(A + D - B - C);
    const int area2 = (y2 - y1) * (x2 - x1);
    const float avg2 = sum2 / (float)area2;
    return avg1 - avg2;
}

/* Index guard for the unclamped integer path: the reference indexes the integral image unchecked
 * (bad.cpp:371-390).  For keypoints that pass the border test with size >= ~11 the taps are always
 * inside; for degenerate sizes the reference would read out of bounds (UB).  Oracle and HIP kernel both
 * clamp the tap coordinates into the integral image so the result is defined. */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void efxo_bad_compute(const uint8_t* img, int rows, int cols, int stride,
                      const float* kps, int n, float scale_factor,
                      const int32_t* boxes, const float* thresholds, int nbits,
                      uint8_t* desc)
{
    if (n <= 0) return;
    const int fw = cols + 1, fh = rows + 1;
    int32_t* I = (int32_t*)malloc(sizeof(int32_t) * (size_t)fw * fh);
    efxo_integral(img, rows, cols, stride, I);
    const int nbytes = nbits / 8;
    if (nbits > 512) { free(I); return; }

    EFXO_PAR
    for (int k = 0; k < n; k++) {                                   /* bad.cpp:341 */
        box_t bp[512];
        const float x = kps[4 * k + 0], y = kps[4 * k + 1], size = kps[4 * k + 2], angle = kps[4 * k + 3];
        uint8_t* d = desc + (size_t)k * nbytes;
        uint8_t byte = 0;
        bad_rectify(boxes, nbits, bp, x, y, size, angle, scale_factor);
        if (bad_in_border(x, y, size, cols, rows, scale_factor)) {  /* bad.cpp:345-361 */
            for (int b = 0; b < nbits; b++) {
                const int bit = 7 - (b % 8);
                const float resp = bad_response_clamped(&bp[b], I, fw, fh);
                byte |= (uint8_t)((resp <= thresholds[b]) << bit);
                if (bit == 0) { *d++ = byte; byte = 0; }
            }
        } else {                                                    /* bad.cpp:362-403 */
            for (int b = 0; b < nbits; b++) {
                const int bit = 7 - (b % 8);
                const int r = bp[b].r;
                const int ax1 = clampi(bp[b].x1 - r, 0, fw - 1), ay1 = clampi(bp[b].y1 - r, 0, fh - 1);
                const int ax2 = clampi(bp[b].x1 + r + 1, 0, fw - 1), ay2 = clampi(bp[b].y1 + r + 1, 0, fh - 1);
                const int bx1 = clampi(bp[b].x2 - r, 0, fw - 1), by1 = clampi(bp[b].y2 - r, 0, fh - 1);
                const int bx2 = clampi(bp[b].x2 + r + 1, 0, fw - 1), by2 = clampi(bp[b].y2 + r + 1, 0, fh - 1);
                const int side = 1 + (r << 1);
                const uint32_t* U = (const uint32_t*)I;
                const int area_resp = (int32_t)(U[(size_t)ay1 * fw + ax1] + U[(size_t)ay2 * fw + ax2]
                                     - U[(size_t)ay1 * fw + ax2] - U[(size_t)ay2 * fw + ax1]
                                     - U[(size_t)by1 * fw + bx1] - U[(size_t)by2 * fw + bx2]
                                     + U[(size_t)by1 * fw + bx2] + U[(size_t)by2 * fw + bx1]);
                byte |= (uint8_t)(((float)area_resp <= (thresholds[b] * (float)(side * side))) << bit);
                if (bit == 0) { *d++ = byte; byte = 0; }
            }
        }
    }
    free(I);
}

/* ================================================================================================
 * HashSIFT  (modules/efficient_features/src/hash_sift.cpp)
 * ============================================================================================== */

#define HS_R_BINS 4
#define HS_C_BINS 4
#define HS_ORI_BINS 8
#define HS_MAG_TH 0.2f
#define HS_INT_FACTOR 512.f
#define HS_SCL_FCTR 3.f

static const float HS_PI_1 = (float)3.1415926535897932384626433832795;   /* CV_PI, hash_sift.cpp:29 */
static const float HS_PI_2 = (float)6.283185307179586476925286766559;    /* CV_2PI, hash_sift.cpp:30 */

void efxo_hashsift_patch(const uint8_t* img, int rows, int cols, int stride,
                         const float* kp4, float crop_scale, uint8_t* patch)
{
    /* rectifyPatch, hash_sift.cpp:111-138 */
    const int w = 32, h = 32;
    const float px = kp4[0], py = kp4[1], size = kp4[2], angle = kp4[3];
    const float s = crop_scale * size / (0.5f * (float)(w + h));
    const float theta = HS_PI_1 * angle / 180;
    const float cost = s * (angle >= 0 ? cosf(theta) : 1.f);
    const float sint = s * (angle >= 0 ? sinf(theta) : 0.f);
    const float M00 = +cost, M01 = -sint, M02 = (-cost + sint) * (float)w / 2.f + px;
    const float M10 = +sint, M11 = +cost, M12 = (-sint - cost) * (float)h / 2.f + py;

    /* warpAffineLinear, hash_sift.cpp:68-109 */
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const float u = M00 * (float)x + M01 * (float)y + M02;
            const float v = M10 * (float)x + M11 * (float)y + M12;
            uint8_t val = 0;
            const int ui = cv_floor_f(u);
            const int vi = cv_floor_f(v);
            if (ui >= 0 && ui + 1 < cols && vi >= 0 && vi + 1 < rows) {
                const uint8_t* p = img + (size_t)vi * stride + ui;
                const float du = u - (float)ui;
                const float dv = v - (float)vi;
                const float t0 = (1 - du) * (float)p[0] + du * (float)p[1];
                const float t1 = (1 - du) * (float)p[stride] + du * (float)p[stride + 1];
                const float t2 = (1 - dv) * t0 + dv * t1;
                int iv = (int)(t2 + 0.5f);
                if (iv > 255) iv = 255;
                val = (uint8_t)iv;
            }
            patch[y * w + x] = val;
        }
    }
}

static float hs_squared(float x) { return x * x; }
static float hs_normsq(float x, float y) { return hs_squared(x) + hs_squared(y); }

static void hs_normalize(float* d, int n)
{
    /* normalize, hash_sift.cpp:150-160 */
    float sum = 0;
    for (int i = 0; i < n; i++) sum += hs_squared(d[i]);
    float norm = sqrtf(sum);
    if (norm < FLT_EPSILON) norm = FLT_EPSILON;
    const float scale = 1.f / norm;
    for (int i = 0; i < n; i++) d[i] *= scale;
}

/* computePatchSIFT, hash_sift.cpp:200-331 (STEP1_PYRAMID is false: no patch blur) */
/* fixed_point != 0: the histogram is summed as the HIP kernel does it (hashsift_kernels.hip): every vote is converted
 * to 15.17 fixed point (rounded half up) and added as an integer (order independent), the bin is converted back to
 * float at the end; the two 128-term sums of squares of the normalisation are added in the device's tree order.
 * This is a CPU MODEL OF THE DEVICE ARITHMETIC used to (a) check the kernel bit for bit and (b) quantify how far that
 * arithmetic is from the reference's sequentially rounded float sums (fixed_point == 0). */
/* the device's order of the 128-term sum of squares: element i with i + 64, then the butterfly 32, 16, .. 1 (float add
 * is commutative, so every lane of the butterfly holds the same value); otherwise hs_normalize */
static void hs_normalize_tree128(float* d)
{
    float t[64];
    for (int i = 0; i < 64; i++) t[i] = d[i] * d[i] + d[i + 64] * d[i + 64];
    for (int off = 32; off >= 1; off >>= 1) {
        float u[64];
        for (int i = 0; i < 64; i++) u[i] = t[i] + t[i ^ off];
        memcpy(t, u, sizeof(t));
    }
    float norm = sqrtf(t[0]);
    if (norm < 1.1920929e-07f) norm = 1.1920929e-07f;
    const float scale = 1.f / norm;
    for (int i = 0; i < 128; i++) d[i] = d[i] * scale;
}

/* fixed_point: bit 0 = fixed-point histogram, bit 1 = tree-ordered norms (the device model is 3; 1 and 2 isolate the two
 * differences from the reference arithmetic: efxo_hashsift_responses_model) */
static void hs_patch_sift(const uint8_t* patch, float* desc /*128*/, float kp_scale, int fixed_point)
{
    const int tree_norm = (fixed_point >> 1) & 1;
    fixed_point &= 1;
    const int h = 32, w = 32, dh = h - 2, dw = w - 2;
    const float kp_radius = kp_scale * (float)h * 0.5f;
    const float kernel_sigma = 0.5f * (float)HS_C_BINS * HS_SCL_FCTR * kp_radius;
    const float dist_scale = -1.f / ((float)2 * kernel_sigma * kernel_sigma);
    const float cx = 0.5f * (float)dw;
    const float cy = 0.5f * (float)dh;

    float hist[HS_R_BINS + 2][HS_C_BINS + 2][HS_ORI_BINS + 2];
    uint64_t h64[HS_R_BINS + 2][HS_C_BINS + 2][HS_ORI_BINS + 2];
    memset(hist, 0, sizeof(hist));
    memset(h64, 0, sizeof(h64));
    /* fixed_point: the device's histogram arithmetic -- 15.17 fixed point, every vote rounded half up (floor(v * 2^17 + 1/2),
     * the sum formed exactly), integer sums (order independent), cuda-efficient-features_amd/csrc/hashsift_kernels.hip */
#define HS_VOTE(R, Cc, O, V) do { if (fixed_point) h64[R][Cc][O] += (uint64_t)(uint32_t)floor((double)((V) * 131072.f) + 0.5); \
        else hist[R][Cc][O] += (V); } while (0)

    /* HistBin, hash_sift.cpp:162-184 */
    const float cellh = HS_SCL_FCTR * (kp_scale * (float)h * 0.5f);
    const float cellw = HS_SCL_FCTR * (kp_scale * (float)w * 0.5f);
    const float scaleR = 1.f / cellh, scaleC = 1.f / cellw, scaleO = (float)HS_ORI_BINS / HS_PI_2;
    const float halfh = 0.5f * (float)h, halfw = 0.5f * (float)w;
    const float rbin0 = (float)(HS_R_BINS / 2) - 0.5f, cbin0 = (float)(HS_C_BINS / 2) - 0.5f;

    for (int y = 0; y < dh; y++) {
        const uint8_t* pT = patch + (y + 0) * w + 1;
        const uint8_t* pC = patch + (y + 1) * w + 1;
        const uint8_t* pB = patch + (y + 2) * w + 1;
        const float rbin = scaleR * ((float)(y + 1) - halfh) + rbin0;
        const int ri = cv_floor_f(rbin);
        const float rf = rbin - (float)ri;
        for (int x = 0; x < dw; x++) {
            const float mag_scale = expf(dist_scale * hs_normsq((float)x - cx, (float)y - cy));
            const float dx = (float)(pC[x + 1] - pC[x - 1]);
            const float dy = (float)(pT[x] - pB[x]);
            const float mag = mag_scale * sqrtf(hs_normsq(dx, dy));
            const float ori = atan2f(dy, dx);

            const float cbin = scaleC * ((float)(x + 1) - halfw) + cbin0;
            const int ci = cv_floor_f(cbin);
            const float cf = cbin - (float)ci;

            const float obin = scaleO * ori;
            int oi = cv_floor_f(obin);
            const float of = obin - (float)oi;
            if (oi < 0) oi += HS_ORI_BINS;
            if (oi >= HS_ORI_BINS) oi -= HS_ORI_BINS;

            /* distribute(value, weight): v1 = weight*value, v0 = value - v1  (hash_sift.cpp:193-198) */
            const float v1 = rf * mag, v0 = mag - v1;
            const float v01 = cf * v0, v00 = v0 - v01;
            const float v11 = cf * v1, v10 = v1 - v11;
            const float v001 = of * v00, v000 = v00 - v001;
            const float v011 = of * v01, v010 = v01 - v011;
            const float v101 = of * v10, v100 = v10 - v101;
            const float v111 = of * v11, v110 = v11 - v111;

            HS_VOTE(ri + 1, ci + 1, oi + 0, v000);
            HS_VOTE(ri + 1, ci + 1, oi + 1, v001);
            HS_VOTE(ri + 1, ci + 2, oi + 0, v010);
            HS_VOTE(ri + 1, ci + 2, oi + 1, v011);
            HS_VOTE(ri + 2, ci + 1, oi + 0, v100);
            HS_VOTE(ri + 2, ci + 1, oi + 1, v101);
            HS_VOTE(ri + 2, ci + 2, oi + 0, v110);
            HS_VOTE(ri + 2, ci + 2, oi + 1, v111);
        }
    }
#undef HS_VOTE
    if (fixed_point)
        for (int r = 0; r < HS_R_BINS + 2; r++)
            for (int c = 0; c < HS_C_BINS + 2; c++)
                for (int o = 0; o < HS_ORI_BINS + 2; o++) hist[r][c][o] = (float)((double)(uint32_t)h64[r][c][o] * (1.0 / 131072.0));
    /* circular orientation fold + copy, hash_sift.cpp:293-308 */
    for (int r = 0; r < HS_R_BINS; r++)
        for (int c = 0; c < HS_C_BINS; c++) {
            float* ph = hist[r + 1][c + 1];
            ph[0] += ph[HS_ORI_BINS + 0];
            ph[1] += ph[HS_ORI_BINS + 1];
            for (int k = 0; k < HS_ORI_BINS; k++) desc[(r * HS_R_BINS + c) * HS_ORI_BINS + k] = ph[k];
        }
    if (tree_norm) hs_normalize_tree128(desc); else hs_normalize(desc, 128);          /* step 7 */
    for (int i = 0; i < 128; i++) desc[i] = desc[i] < HS_MAG_TH ? desc[i] : HS_MAG_TH;   /* step 8 */
    if (tree_norm) hs_normalize_tree128(desc); else hs_normalize(desc, 128);
    for (int k = 0; k < 128; k++) desc[k] = (float)sat_u8_f(HS_INT_FACTOR * desc[k]);   /* step 9 */
}

void efxo_hashsift_responses(const uint8_t* img, int rows, int cols, int stride,
                             const float* kps, int n, float crop_scale, float* responses)
{
    /* computePatchSIFTs, hash_sift.cpp:333-351; keypointScale = 1/6 */
    const float kp_scale = 1.f / 6;
    EFXO_PAR
    for (int i = 0; i < n; i++) {
        uint8_t patch[32 * 32];
        float* r = responses + (size_t)i * 129;
        r[0] = 1;
        efxo_hashsift_patch(img, rows, cols, stride, kps + 4 * i, crop_scale, patch);
        hs_patch_sift(patch, r + 1, kp_scale, 0);
    }
}

/* the same with the device's fixed-point histogram sums (see hs_patch_sift) */
void efxo_hashsift_responses_fixedpoint(const uint8_t* img, int rows, int cols, int stride,
                                        const float* kps, int n, float crop_scale, float* responses)
{
    const float kp_scale = 1.f / 6;
    EFXO_PAR
    for (int i = 0; i < n; i++) {
        uint8_t patch[32 * 32];
        float* r = responses + (size_t)i * 129;
        r[0] = 1;
        efxo_hashsift_patch(img, rows, cols, stride, kps + 4 * i, crop_scale, patch);
        hs_patch_sift(patch, r + 1, kp_scale, 3);
    }
}

/* mode: bit 0 = fixed-point histogram, bit 1 = tree-ordered norms (0 = the reference arithmetic, 3 = the device model) */
void efxo_hashsift_responses_model(const uint8_t* img, int rows, int cols, int stride,
                                   const float* kps, int n, float crop_scale, int mode, float* responses)
{
    const float kp_scale = 1.f / 6;
    EFXO_PAR
    for (int i = 0; i < n; i++) {
        uint8_t patch[32 * 32];
        float* r = responses + (size_t)i * 129;
        r[0] = 1;
        efxo_hashsift_patch(img, rows, cols, stride, kps + 4 * i, crop_scale, patch);
        hs_patch_sift(patch, r + 1, kp_scale, mode & 3);
    }
}

void efxo_hashsift_project(const float* responses, int n, const float* W, int nbits, float* T, uint8_t* desc)
{
    /* matmulAndSign, hash_sift.cpp:353-378.  cv::gemm (third party) accumulates CV_32F products in
     * double (GEMMSingleMul<float,double>) -- restated as a k-ordered double sum rounded to float. */
    const int nbytes = nbits / 8;
    EFXO_PAR
    for (int i = 0; i < n; i++) {
        const float* r = responses + (size_t)i * 129;
        uint8_t* d = desc + (size_t)i * nbytes;
        for (int b = 0; b < nbytes; b++) {
            uint8_t byte = 0;
            for (int j = 0; j < 8; j++) {
                const float* w = W + (size_t)(b * 8 + j) * 129;
                double acc = 0;
                for (int k = 0; k < 129; k++) acc += (double)r[k] * (double)w[k];
                const float t = (float)acc;
                if (T) T[(size_t)i * nbits + b * 8 + j] = t;
                byte |= (uint8_t)((t > 0) << (7 - j));
            }
            d[b] = byte;
        }
    }
}

void efxo_hashsift_compute(const uint8_t* img, int rows, int cols, int stride,
                           const float* kps, int n, float crop_scale,
                           const float* W, int nbits, uint8_t* desc)
{
    if (n <= 0) return;
    float* resp = (float*)malloc(sizeof(float) * (size_t)n * 129);
    efxo_hashsift_responses(img, rows, cols, stride, kps, n, crop_scale, resp);
    efxo_hashsift_project(resp, n, W, nbits, NULL, desc);
    free(resp);
}

/* ================================================================================================
 * Detector  (modules/cuda_efficient_features/src/cuda_efficient_features.{cpp,cu}, cuda_fast.cu)
 * ============================================================================================== */

void efxo_pyramid_geometry(int rows, int cols, float scale_factor, int nlevels,
                           int* lrows, int* lcols, float* scales)
{
    /* calcImagePyramid, cuda_efficient_features.cpp:144-155: float scale chain, cvRound(float) */
    float scale = 1.f;
    lrows[0] = rows; lcols[0] = cols; scales[0] = scale;
    for (int s = 1; s < nlevels; s++) {
        scale *= scale_factor;
        const float inv = 1.f / scale;
        lrows[s] = cv_round_f(inv * (float)rows);
        lcols[s] = cv_round_f(inv * (float)cols);
        scales[s] = scale;
    }
}

void efxo_level_quotas(int total, float scale_factor, int nlevels, int* q)
{
    /* calcNumFeaturesPerLevel, cuda_efficient_features.cpp:159-174.
     * `1 / scaleFactor` is a FLOAT division widened to double. */
    const double factor = (double)(1 / scale_factor);
    double nf = total * (1 - factor) / (1 - pow(factor, nlevels));
    int sum = 0;
    for (int s = 0; s < nlevels - 1; s++) {
        q[s] = cv_round_d(nf);
        sum += q[s];
        nf *= factor;
    }
    q[nlevels - 1] = total - sum > 0 ? total - sum : 0;
}

/* Spec S5 weight of the upper (i2 = i1 + 1) / lower (i1) neighbour of destination index o, s = (float)o * f rounded */
#ifndef EFX_S5_FUSED_WEIGHTS
#define EFX_S5_FUSED_WEIGHTS 0
#endif
#if EFX_S5_FUSED_WEIGHTS
#define EFXO_S5_W_HI(o, f, s, i2) fmaf(-(float)(o), (f), (float)(i2))
#define EFXO_S5_W_LO(o, f, s, i1) fmaf((float)(o), (f), -(float)(i1))
#else
#define EFXO_S5_W_HI(o, f, s, i2) ((float)(i2) - (s))
#define EFXO_S5_W_LO(o, f, s, i1) ((s) - (float)(i1))
#endif
int efxo_s5_fused_weights(void) { return EFX_S5_FUSED_WEIGHTS; }

void efxo_resize_linear(const uint8_t* src, int srows, int scols, int sstride,
                        uint8_t* dst, int drows, int dcols, int dstride)
{
    /* Spec S5: cv::cuda::resize(INTER_LINEAR) as in opencv_contrib cudawarping resize_linear:
     * src = dst * (1/f) with f = dsize/ssize (double) cast to float, no half-pixel offset, floor,
     * +1 neighbour clamped to the last row/col, four float weights (each a rounded product), accumulated by
     * fused multiply-adds in (00,01,10,11) order, round-to-nearest-even saturate. */
    const float fx = (float)(1.0 / ((double)dcols / (double)scols));
    const float fy = (float)(1.0 / ((double)drows / (double)srows));
    EFXO_PAR
    for (int dy = 0; dy < drows; dy++) {
        const float sy = (float)dy * fy;
        int y1 = cv_floor_f(sy);
        if (y1 > srows - 1) y1 = srows - 1;
        const int y2 = y1 + 1;
        const int y2r = y2 < srows - 1 ? y2 : srows - 1;
        for (int dx = 0; dx < dcols; dx++) {
            const float sx = (float)dx * fx;
            int x1 = cv_floor_f(sx);
            if (x1 > scols - 1) x1 = scols - 1;
            const int x2 = x1 + 1;
            const int x2r = x2 < scols - 1 ? x2 : scols - 1;
            /* `out = out + src_reg * (wx * wy)` four times (opencv_contrib cudawarping resize_linear / LinearFilter): under
             * nvcc's default -fmad each statement is ONE fused multiply-add of the pixel and the ROUNDED weight product into
             * the accumulator -- the same contraction spec S6 models for the Gaussian (the reference binary is one nvcc build:
             * both third-party kernels are contracted or neither is).  The first statement adds to 0.f and is the
             * rounded product either way. */
            /* The four weights `(x2 - src_x)`, `(src_x - x1)`, ... are kept as SEPARATE subtractions of the rounded product
             * src_x = dst_x * fx: nvcc's contraction could also turn them into fma(-dst_x, fx, x2) / fma(dst_x, fx, -x1),
             * whose result differs in the last bit whenever dst_x * fx is inexact.  Nothing in this image can decide it (no CUDA
             * build, no dump of cv::cuda::resize); the other reading is one switch away, in the oracle and in the HIP library
             * alike: build both with -DEFX_S5_FUSED_WEIGHTS=1 (tests/test_oracle_detector.py::test_s5_weight_switch keeps the
             * two readings alive and distinct).  Default: separate -- the weights are then exactly the CPU cv::resize-style
             * fractions, and every pyramid fixture of this tree was made that way. */
            const float wx0 = EFXO_S5_W_HI(dx, fx, sx, x2), wx1 = EFXO_S5_W_LO(dx, fx, sx, x1);
            const float wy0 = EFXO_S5_W_HI(dy, fy, sy, y2), wy1 = EFXO_S5_W_LO(dy, fy, sy, y1);
            float out = 0.f;
            out = fmaf((float)src[(size_t)y1 * sstride + x1], wx0 * wy0, out);
            out = fmaf((float)src[(size_t)y1 * sstride + x2r], wx1 * wy0, out);
            out = fmaf((float)src[(size_t)y2r * sstride + x1], wx0 * wy1, out);
            out = fmaf((float)src[(size_t)y2r * sstride + x2r], wx1 * wy1, out);
            dst[(size_t)dy * dstride + dx] = sat_u8_f(out);
        }
    }
}

void efxo_gaussian_taps(float taps[7])
{
    /* Spec S6: cv::getGaussianKernel(7, 2): exp(-(i-3)^2 / (2*sigma^2)) normalised, double -> float */
    double e[7], sum = 0;
    for (int i = 0; i < 7; i++) {
        const double x = (double)(i - 3);
        e[i] = exp(-(x * x) / (2.0 * 2.0 * 2.0));
        sum += e[i];
    }
    for (int i = 0; i < 7; i++) taps[i] = (float)(e[i] / sum);
}

static int reflect101(int p, int len)
{
    /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

void efxo_gaussian7(const uint8_t* src, int rows, int cols, int sstride, uint8_t* dst, int dstride)
{
    /* Spec S6: separable; row pass u8 -> float, column pass float -> u8 (round-half-even saturate); each pass
     * accumulates acc = fma(tap_j, v_j, acc) for j = 0..6 from acc = 0 (single rounding per tap: what nvcc's
     * default FMA contraction makes of the `sum = sum + src * kernel[k]` loop of the filter the reference calls).
     * Call site: cuda_efficient_features.cpp:193,305. */
    float taps[7];
    efxo_gaussian_taps(taps);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * cols);
    EFXO_PAR
    for (int y = 0; y < rows; y++) {
        const uint8_t* p = src + (size_t)y * sstride;
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int j = 0; j < 7; j++) acc = fmaf(taps[j], (float)p[reflect101(x + j - 3, cols)], acc);
            tmp[(size_t)y * cols + x] = acc;
        }
    }
    EFXO_PAR
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int j = 0; j < 7; j++) acc = fmaf(taps[j], tmp[(size_t)reflect101(y + j - 3, rows) * cols + x], acc);
            dst[(size_t)y * dstride + x] = sat_u8_f(acc);
        }
    }
    free(tmp);
}

/* Bresenham circle r=3, bit k = position k, same order as cuda_fast.cu:179-207 / :51-156
 * (SURVEY 8a5): starts at (y+3,x) and walks towards +x. */
static const int FAST_DX[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int FAST_DY[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

static int has_arc9(unsigned m)
{
    /* >= 9 circularly contiguous set bits in a 16-bit ring == the c_table lookup of cuda_fast.cu:160-166 */
    unsigned d = m | (m << 16);
    d &= d >> 1;   /* runs of 2 */
    d &= d >> 2;   /* runs of 4 */
    d &= d >> 4;   /* runs of 8 */
    d &= d >> 1;   /* runs of 9 */
    return (d & 0xffffu) != 0;
}

int efxo_has_arc9(unsigned mask16) { return has_arc9(mask16 & 0xffffu); }

int efxo_fast9_at(const uint8_t* img, int stride, int x, int y, int threshold)
{
    /* diffType, cuda_fast.cu:36-40: strict comparisons */
    const int v = img[(size_t)y * stride + x];
    unsigned darker = 0, brighter = 0;
    {   /* early exit of cuda_fast.cu:193-197: a 9-arc contains one pixel of every opposing pair */
        const int e = img[(size_t)y * stride + x + 3] - v, w = img[(size_t)y * stride + x - 3] - v;
        if (e >= -threshold && e <= threshold && w >= -threshold && w <= threshold) return 0;
    }
    for (int k = 0; k < 16; k++) {
        const int c = img[(size_t)(y + FAST_DY[k]) * stride + (x + FAST_DX[k])];
        const int diff = c - v;
        if (diff < -threshold) darker |= 1u << k;
        if (diff > threshold) brighter |= 1u << k;
    }
    return has_arc9(darker) || has_arc9(brighter);
}

int efxo_fast9_detect(const uint8_t* img, int rows, int cols, int stride, int threshold, int border,
                      int16_t* xy, int max_out)
{
    /* calcKeypointsKernel, cuda_fast.cu:168-222: pixels with a 3-px margin AND mask != 0, where the mask
     * is 255 inside [border, cols-border) x [border, rows-border) (cuda_efficient_features.cpp:176-182). */
    int n = 0;
    const int b = border > 3 ? border : 3;
    if (rows <= 2 * b || cols <= 2 * b) return 0;
    uint8_t* flag = (uint8_t*)calloc((size_t)rows * cols, 1);
    EFXO_PAR
    for (int y = b; y < rows - b; y++)
        for (int x = b; x < cols - b; x++) flag[(size_t)y * cols + x] = (uint8_t)efxo_fast9_at(img, stride, x, y, threshold);
    for (int y = b; y < rows - b; y++)
        for (int x = b; x < cols - b; x++)
            if (flag[(size_t)y * cols + x]) {
                if (n < max_out) { xy[2 * n] = (int16_t)x; xy[2 * n + 1] = (int16_t)y; }
                n++;
            }
    free(flag);
    return n;
}

float efxo_harris(const uint8_t* img, int stride, int x0, int y0)
{
    /* calcResponse, cuda_efficient_features.cu:99-139, with spec S4: the 49 products are summed as exact
     * int32 (|Sobel| <= 1020), then ONE fixed float formula (no contraction):
     *   K = SCALE*SCALE, a = float(Sxx)*K, b = float(Syy)*K, c = float(Sxy)*K,
     *   R = (a*b - c*c) - (0.04f*(a+b))*(a+b)                                                   */
    const float SCALE = 1.f / (float)(4 * 7 * 255);
    int sxx = 0, sxy = 0, syy = 0;
    for (int iy = -3; iy <= 3; iy++)
        for (int ix = -3; ix <= 3; ix++) {
            const uint8_t* p = img + (size_t)(y0 + iy) * stride + (x0 + ix);
            const int v00 = p[-stride - 1], v01 = p[-stride], v02 = p[-stride + 1];
            const int v10 = p[-1], v12 = p[1];
            const int v20 = p[stride - 1], v21 = p[stride], v22 = p[stride + 1];
            const int dx = (v02 + 2 * v12 + v22) - (v00 + 2 * v10 + v20);
            const int dy = (v20 + 2 * v21 + v22) - (v00 + 2 * v01 + v02);
            sxx += dx * dx;
            sxy += dx * dy;
            syy += dy * dy;
        }
    const float K = SCALE * SCALE;
    const float a = (float)sxx * K, b = (float)syy * K, c = (float)sxy * K;
    const float det = a * b - c * c;
    const float tr = a + b;
    return det - 0.04f * tr * tr;
}

/* Spec S7: deterministic atan2 shared with the HIP kernel.  Double precision, IEEE +,-,*,/ only,
 * no contraction: octant reduction, then atan(t) for t in [0,1] via
 * atan(t) = pi/4 + atan((t-1)/(t+1)) when t > tan(pi/8), and a 24-term odd Taylor series (|u| <= 0.4143,
 * truncation error < 1e-19).  Result in degrees, float, nominal range [0,360). */
float efxo_atan2_deg(int m01, int m10)
{
    const double PI = 3.14159265358979323846;
    const double y = (double)m01, x = (double)m10;
    const double ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    if (ax == 0 && ay == 0) return 0.f;
    const double mn = ax < ay ? ax : ay, mx = ax < ay ? ay : ax;
    const double t = mn / mx;
    double base = 0.0, u = t;
    if (t > 0.41421356237309503) { base = PI / 4; u = (t - 1.0) / (t + 1.0); }
    const double u2 = u * u;
    double s = 0.0;
    for (int k = 23; k >= 0; k--) {
        const double ck = 1.0 / (double)(2 * k + 1);
        s = (k & 1 ? -ck : ck) + u2 * s;          /* Horner on u^2: sum (-1)^k u^(2k) / (2k+1) */
    }
    double a = base + u * s;
    if (ay > ax) a = PI / 2 - a;
    if (x < 0) a = PI - a;
    if (y < 0) a = -a;
    if (a < 0) a = a + 2 * PI;
    return (float)(a * (180.0 / PI));
}

/* U_MAX of IC_Angle, cuda_efficient_features.cu:143 (pinned against the reference's text by tests/test_reference_table_pins.py) */
static const int IC_U_MAX[17] = { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3, 0 };
void efxo_ic_umax(int* out17) { for (int i = 0; i < 17; i++) out17[i] = IC_U_MAX[i]; }

float efxo_ic_angle(const uint8_t* img, int stride, int x, int y)
{
    /* IC_Angle, cuda_efficient_features.cu:141-172 */
    const int* U_MAX = IC_U_MAX;
    int m01 = 0, m10 = 0;
    const uint8_t* c = img + (size_t)y * stride + x;
    for (int dx = -EFXO_HALF_PATCH; dx <= EFXO_HALF_PATCH; dx++) m10 += dx * c[dx];
    for (int dy = 1; dy <= EFXO_HALF_PATCH; dy++) {
        int ysum = 0;
        const int d = U_MAX[dy];
        for (int dx = -d; dx <= d; dx++) {
            const int vT = c[-dy * stride + dx];
            const int vB = c[dy * stride + dx];
            ysum += (vB - vT);
            m10 += dx * (vB + vT);
        }
        m01 += dy * ysum;
    }
    return efxo_atan2_deg(m01, m10);
}

/* ---- canonical order (spec S1): 64x64 tile row-major, then 16x16 cell row-major inside the tile,
 *      then pixel raster inside the cell ---- */
static uint64_t canon_key(int x, int y, int tiles_x)
{
    const int tx = x / EFXO_TILE, ty = y / EFXO_TILE;
    const int cx = (x % EFXO_TILE) / EFXO_CELL, cy = (y % EFXO_TILE) / EFXO_CELL;
    const int px = x % EFXO_CELL, py = y % EFXO_CELL;
    return ((uint64_t)(ty * tiles_x + tx) << 12) | (uint64_t)((cy * 4 + cx) << 8) | (uint64_t)(py << 4) | (uint64_t)px;
}

typedef struct { uint64_t key; int16_t x, y; float resp; } cand_t;

static int cmp_cand(const void* a, const void* b)
{
    const uint64_t ka = ((const cand_t*)a)->key, kb = ((const cand_t*)b)->key;
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

typedef struct { float resp; int16_t x, y; int idx; } sel_t;
static int cmp_sel(const void* a, const void* b)
{
    /* response descending, then raster (y, x) ascending: spec S3 */
    const sel_t* p = (const sel_t*)a; const sel_t* q = (const sel_t*)b;
    if (p->resp > q->resp) return -1;
    if (p->resp < q->resp) return 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return 0;
}

/* radiusSuppression + IsMaxPoint, cuda_efficient_features.cu:62-97, 281-342.  keep[i] = 1 if survivor. */
static void radius_nms(const cand_t* c, int n, int w, int h, int radius, uint8_t* keep)
{
    const int image_radius = radius * radius;                       /* cvCeil(radius*radius), :291 */
    const int block_radius = (radius + EFXO_CELL - 1) / EFXO_CELL;  /* cvCeil(radius / CELL_SIZE), :292 */
    const int gw = (w + EFXO_CELL - 1) / EFXO_CELL, gh = (h + EFXO_CELL - 1) / EFXO_CELL;
    int* start = (int*)calloc((size_t)gw * gh + 1, sizeof(int));
    int* ids = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) start[(c[i].y / EFXO_CELL) * gw + c[i].x / EFXO_CELL + 1]++;
    for (int i = 0; i < gw * gh; i++) start[i + 1] += start[i];
    int* cur = (int*)malloc(sizeof(int) * ((size_t)gw * gh + 1));
    memcpy(cur, start, sizeof(int) * ((size_t)gw * gh + 1));
    for (int i = 0; i < n; i++) ids[cur[(c[i].y / EFXO_CELL) * gw + c[i].x / EFXO_CELL]++] = i;
    EFXO_PAR
    for (int i = 0; i < n; i++) {
        const int bx1 = c[i].x / EFXO_CELL, by1 = c[i].y / EFXO_CELL;
        const int minx = bx1 - block_radius > 0 ? bx1 - block_radius : 0;
        const int maxx = bx1 + block_radius < gw - 1 ? bx1 + block_radius : gw - 1;
        const int miny = by1 - block_radius > 0 ? by1 - block_radius : 0;
        const int maxy = by1 + block_radius < gh - 1 ? by1 + block_radius : gh - 1;
        int is_max = 1;
        for (int by = miny; by <= maxy && is_max; by++)
            for (int bx = minx; bx <= maxx && is_max; bx++)
                for (int k = start[by * gw + bx]; k < start[by * gw + bx + 1]; k++) {
                    const int j = ids[k];
                    if (j == i) continue;
                    const int dx = c[i].x - c[j].x, dy = c[i].y - c[j].y;
                    if (c[i].resp <= c[j].resp && dx * dx + dy * dy < image_radius) { is_max = 0; break; }
                }
        keep[i] = (uint8_t)is_max;
    }
    free(cur); free(ids); free(start);
}

int efxo_pyramid_level(const uint8_t* img, int rows, int cols, int stride, float scale_factor, int level, uint8_t* dst)
{
    if (level < 0 || level >= EFXO_MAX_LEVELS) return -1;
    int lr[EFXO_MAX_LEVELS], lc[EFXO_MAX_LEVELS]; float sc[EFXO_MAX_LEVELS];
    efxo_pyramid_geometry(rows, cols, scale_factor, level + 1, lr, lc, sc);
    uint8_t* prev = (uint8_t*)malloc((size_t)rows * cols);
    for (int y = 0; y < rows; y++) memcpy(prev + (size_t)y * cols, img + (size_t)y * stride, (size_t)cols);
    for (int s = 1; s <= level; s++) {
        uint8_t* cur = (uint8_t*)malloc((size_t)lr[s] * lc[s]);
        efxo_resize_linear(prev, lr[s - 1], lc[s - 1], lc[s - 1], cur, lr[s], lc[s], lc[s]);
        free(prev);
        prev = cur;
    }
    memcpy(dst, prev, (size_t)lr[level] * lc[level]);
    free(prev);
    return 0;
}

int efxo_detect_and_compute(const uint8_t* img, int rows, int cols, int stride,
                            const efxo_params* p, int desc_type,
                            const void* params_a, const void* params_b,
                            float* kps_out, uint8_t* desc_out, int16_t* lvl_xy_out, int capacity,
                            efxo_stats* stats)
{
    return efxo_detect_and_compute_masked(img, rows, cols, stride, NULL, 0, p, desc_type, params_a, params_b, kps_out, desc_out,
                                          lvl_xy_out, capacity, stats);
}

/* Spec S12 (the reference accepts `mask` and ignores it, cuda_efficient_features.cpp:225-250): a FAST corner at level
 * coordinates (x, y) of level s exists only if the level-0 mask is non-zero at the pixel the keypoint will be reported
 * at, ((short)(scale_s x + 0.5f), (short)(scale_s y + 0.5f)) (scalePoints, cuda_efficient_features.cu:236-248), clamped
 * into the mask.  Masked corners do not take part in the cap, the NMS or the quota. */
static int mask_allows(const uint8_t* mask, int mstride, int rows, int cols, float scale, int x, int y)
{
    if (!mask) return 1;
    int sx = (int16_t)(scale * (float)x + 0.5f), sy = (int16_t)(scale * (float)y + 0.5f);
    if (sx > cols - 1) sx = cols - 1;
    if (sy > rows - 1) sy = rows - 1;
    return mask[(size_t)sy * mstride + sx] != 0;
}

int efxo_detect_and_compute_masked(const uint8_t* img, int rows, int cols, int stride, const uint8_t* mask, int mstride,
                                   const efxo_params* p, int desc_type,
                                   const void* params_a, const void* params_b,
                                   float* kps_out, uint8_t* desc_out, int16_t* lvl_xy_out, int capacity,
                                   efxo_stats* stats)
{
    if (!img || !p || p->nlevels < 1 || p->nlevels > EFXO_MAX_LEVELS || p->first_level < 0 || capacity < 0) return -1;
    if (desc_type < -1 || desc_type > 3) return -1;
    const int nl = p->nlevels;
    int lr[EFXO_MAX_LEVELS], lc[EFXO_MAX_LEVELS], quota[EFXO_MAX_LEVELS];
    float sc[EFXO_MAX_LEVELS];
    efxo_pyramid_geometry(rows, cols, p->scale_factor, nl, lr, lc, sc);
    efxo_level_quotas(p->nfeatures, p->scale_factor, nl, quota);
    if (stats) memset(stats, 0, sizeof(*stats));

    const int nbits = (desc_type == 0 || desc_type == 2) ? 256 : 512;
    const int nbytes = nbits / 8;

    /* pyramid: level s from level s-1 (cuda_efficient_features.cpp:136-157) */
    uint8_t* pyr[EFXO_MAX_LEVELS];
    pyr[0] = (uint8_t*)malloc((size_t)rows * cols);
    for (int y = 0; y < rows; y++) memcpy(pyr[0] + (size_t)y * cols, img + (size_t)y * stride, (size_t)cols);
    for (int s = 1; s < nl; s++) {
        pyr[s] = (uint8_t*)malloc((size_t)(lr[s] > 0 ? lr[s] : 1) * (lc[s] > 0 ? lc[s] : 1));
        if (lr[s] > 0 && lc[s] > 0 && lr[s - 1] > 0 && lc[s - 1] > 0)
            efxo_resize_linear(pyr[s - 1], lr[s - 1], lc[s - 1], lc[s - 1], pyr[s], lr[s], lc[s], lc[s]);
    }

    int total = 0;
    for (int s = p->first_level; s < nl; s++) {                     /* cuda_efficient_features.cpp:244-273 */
        const int w = lc[s], h = lr[s];
        if (w <= 0 || h <= 0) continue;
        const uint8_t* L = pyr[s];
        const int cap = cv_round_d(0.1 * (double)(w * h));          /* :252 */
        const int tiles_x = (w + EFXO_TILE - 1) / EFXO_TILE;

        /* FAST: count first, then collect all (the cap is applied in canonical order, spec S2) */
        int xy_cap = cap > 1024 ? cap : 1024;
        int16_t* xy = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)xy_cap);
        int ncand = efxo_fast9_detect(L, h, w, w, p->fast_threshold, EFXO_HALF_PATCH, xy, xy_cap);
        if (ncand > xy_cap) {   /* more corners than the 10% cap: collect them all, cap in canonical order */
            free(xy);
            xy = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)ncand);
            efxo_fast9_detect(L, h, w, w, p->fast_threshold, EFXO_HALF_PATCH, xy, ncand);
        }
        cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(ncand > 0 ? ncand : 1));
        {
            int kept = 0;
            for (int i = 0; i < ncand; i++) {
                if (!mask_allows(mask, mstride, rows, cols, sc[s], xy[2 * i], xy[2 * i + 1])) continue;       /* spec S12 */
                c[kept].x = xy[2 * i]; c[kept].y = xy[2 * i + 1];
                c[kept].key = canon_key(c[kept].x, c[kept].y, tiles_x);
                kept++;
            }
            ncand = kept;
        }
        qsort(c, (size_t)ncand, sizeof(cand_t), cmp_cand);
        int n = ncand < cap ? ncand : cap;                          /* cuda_fast.cu:245 */
        if (stats) { stats->n_candidates[s] = ncand; stats->n_after_cap[s] = n; }

        EFXO_PAR
        for (int i = 0; i < n; i++) c[i].resp = efxo_harris(L, w, c[i].x, c[i].y);   /* :262 */

        uint8_t* keep = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
        radius_nms(c, n, w, h, p->nonmax_radius, keep);             /* :264 */
        int m = 0;
        for (int i = 0; i < n; i++) if (keep[i]) c[m++] = c[i];     /* canonical order preserved */
        if (stats) stats->n_after_nms[s] = m;

        if (m > quota[s]) {                                         /* limitPoints, .cu:344-358 + spec S3 */
            sel_t* sel = (sel_t*)malloc(sizeof(sel_t) * (size_t)m);
            for (int i = 0; i < m; i++) { sel[i].resp = c[i].resp; sel[i].x = c[i].x; sel[i].y = c[i].y; sel[i].idx = i; }
            qsort(sel, (size_t)m, sizeof(sel_t), cmp_sel);
            memset(keep, 0, (size_t)m);
            for (int i = 0; i < quota[s]; i++) keep[sel[i].idx] = 1;
            int mm = 0;
            for (int i = 0; i < m; i++) if (keep[i]) c[mm++] = c[i];
            m = mm;
            free(sel);
        }
        if (stats) stats->n_kept[s] = m;
        if (total + m > capacity) m = capacity - total;             /* caller-provided capacity */

        /* angles (.cu:376-390), descriptors on the blurred level (.cpp:302-307), scalePoints (.cu:236-248) */
        float* kp4 = (float*)malloc(sizeof(float) * 4 * (size_t)(m > 0 ? m : 1));
        EFXO_PAR
        for (int i = 0; i < m; i++) {
            kp4[4 * i + 0] = (float)c[i].x;
            kp4[4 * i + 1] = (float)c[i].y;
            kp4[4 * i + 2] = (float)EFXO_PATCH_SIZE;                /* convertKeypointsKernel, .cu:260 */
            kp4[4 * i + 3] = efxo_ic_angle(L, w, c[i].x, c[i].y);
        }
        if (desc_type >= 0 && desc_out && m > 0) {
            uint8_t* blur = (uint8_t*)malloc((size_t)w * h);
            efxo_gaussian7(L, h, w, w, blur, w);
            if (desc_type <= 1)
                efxo_bad_compute(blur, h, w, w, kp4, m, 1.f, (const int32_t*)params_a, (const float*)params_b, nbits,
                                 desc_out + (size_t)total * nbytes);
            else
                efxo_hashsift_compute(blur, h, w, w, kp4, m, 1.f, (const float*)params_a, nbits,
                                      desc_out + (size_t)total * nbytes);
            free(blur);
        }
        for (int i = 0; i < m; i++) {
            const int o = total + i;
            const int16_t sx = (int16_t)(sc[s] * (float)c[i].x + 0.5f);
            const int16_t sy = (int16_t)(sc[s] * (float)c[i].y + 0.5f);
            uint32_t packed = (uint32_t)(uint16_t)sx | ((uint32_t)(uint16_t)sy << 16);
            memcpy(&kps_out[0 * (size_t)capacity + o], &packed, 4);
            kps_out[1 * (size_t)capacity + o] = c[i].resp;
            kps_out[2 * (size_t)capacity + o] = kp4[4 * i + 3];
            int32_t oct = s;
            memcpy(&kps_out[3 * (size_t)capacity + o], &oct, 4);
            kps_out[4 * (size_t)capacity + o] = sc[s] * (float)EFXO_PATCH_SIZE;
            if (lvl_xy_out) { lvl_xy_out[o] = c[i].x; lvl_xy_out[(size_t)capacity + o] = c[i].y; }
        }
        total += m;
        free(kp4); free(keep); free(c); free(xy);
    }
    for (int s = 0; s < nl; s++) free(pyr[s]);
    return total;
}

/* Spec S11: cv::cvtColor(COLOR_BGR2GRAY / COLOR_BGRA2GRAY) for 8-bit images as OpenCV >= 3.4 computes it
 * (fixed point, BY15 = 3735, GY15 = 19235, RY15 = 9798, shift 15, + 1 << 14): third-party arithmetic behind
 * bad.cpp:268-281, hash_sift.cpp:51-66 and samples/sample_common.cpp:35-45.  channels = 3 or 4. */
void efxo_bgr2gray(const uint8_t* src, int rows, int cols, int sstride, int channels, uint8_t* dst, int dstride)
{
    for (int y = 0; y < rows; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < cols; x++, s += channels)
            d[x] = (uint8_t)((3735u * s[0] + 19235u * s[1] + 9798u * s[2] + 16384u) >> 15);
    }
}

/* Spec S13 (the reference asserts !useProvidedKeypoints, cuda_efficient_features.cpp:229): detectAndCompute with
 * useProvidedKeypoints = true skips detection and describes the given keypoints exactly as detectAndCompute would
 * have: on the blurred pyramid level `octave`, at level coordinates ((int)(x / scale + 0.5f), (int)(y / scale + 0.5f))
 * -- the inverse of scalePoints (cuda_efficient_features.cu:236-248) for scale >= 1 --, size 31, angle as given
 * (cuda_efficient_features.cpp:302-307).  kps: the 5 x n matrix (rows `capacity` floats apart).  Keypoints whose
 * octave is outside [0, nlevels) get an all-zero descriptor. */
int efxo_compute_provided(const uint8_t* img, int rows, int cols, int stride, const efxo_params* p, int desc_type,
                          const void* params_a, const void* params_b, const float* kps, int capacity, int n, uint8_t* desc_out)
{
    if (!img || !p || p->nlevels < 1 || p->nlevels > EFXO_MAX_LEVELS || desc_type < 0 || desc_type > 3 || n < 0) return -1;
    const int nl = p->nlevels;
    int lr[EFXO_MAX_LEVELS], lc[EFXO_MAX_LEVELS];
    float sc[EFXO_MAX_LEVELS];
    efxo_pyramid_geometry(rows, cols, p->scale_factor, nl, lr, lc, sc);
    const int nbits = (desc_type == 0 || desc_type == 2) ? 256 : 512;
    const int nbytes = nbits / 8;
    memset(desc_out, 0, (size_t)n * nbytes);
    uint8_t* prev = (uint8_t*)malloc((size_t)rows * cols);
    for (int y = 0; y < rows; y++) memcpy(prev + (size_t)y * cols, img + (size_t)y * stride, (size_t)cols);
    for (int s = 0; s < nl; s++) {
        uint8_t* L = prev;
        if (s > 0) {
            L = (uint8_t*)malloc((size_t)(lr[s] > 0 ? lr[s] : 1) * (lc[s] > 0 ? lc[s] : 1));
            if (lr[s] > 0 && lc[s] > 0 && lr[s - 1] > 0 && lc[s - 1] > 0)
                efxo_resize_linear(prev, lr[s - 1], lc[s - 1], lc[s - 1], L, lr[s], lc[s], lc[s]);
            free(prev);
            prev = L;
        }
        const int w = lc[s], h = lr[s];
        if (w <= 0 || h <= 0) continue;
        int m = 0;
        for (int i = 0; i < n; i++) { int32_t o; memcpy(&o, &kps[3 * (size_t)capacity + i], 4); if (o == s) m++; }
        if (!m) continue;
        float* kp4 = (float*)malloc(sizeof(float) * 4 * (size_t)m);
        int* idx = (int*)malloc(sizeof(int) * (size_t)m);
        m = 0;
        for (int i = 0; i < n; i++) {
            int32_t o; memcpy(&o, &kps[3 * (size_t)capacity + i], 4);
            if (o != s) continue;
            uint32_t packed; memcpy(&packed, &kps[i], 4);
            const int16_t x = (int16_t)(packed & 0xffffu), y = (int16_t)(packed >> 16);
            kp4[4 * m + 0] = (float)(int)((float)x / sc[s] + 0.5f);
            kp4[4 * m + 1] = (float)(int)((float)y / sc[s] + 0.5f);
            kp4[4 * m + 2] = (float)EFXO_PATCH_SIZE;
            kp4[4 * m + 3] = kps[2 * (size_t)capacity + i];
            idx[m++] = i;
        }
        uint8_t* blur = (uint8_t*)malloc((size_t)w * h);
        uint8_t* d = (uint8_t*)malloc((size_t)m * nbytes);
        efxo_gaussian7(L, h, w, w, blur, w);
        if (desc_type <= 1)
            efxo_bad_compute(blur, h, w, w, kp4, m, 1.f, (const int32_t*)params_a, (const float*)params_b, nbits, d);
        else
            efxo_hashsift_compute(blur, h, w, w, kp4, m, 1.f, (const float*)params_a, nbits, d);
        for (int j = 0; j < m; j++) memcpy(desc_out + (size_t)idx[j] * nbytes, d + (size_t)j * nbytes, (size_t)nbytes);
        free(blur); free(d); free(kp4); free(idx);
    }
    free(prev);
    return 0;
}

/* ---- HPatches exporter helpers (samples/hpatches_description.cpp) ---- */

/* calcUMax, hpatches_description.cpp:107-126: end of each row of a circular patch.  umax must hold patch_size/2 + 2 ints. */
void efxo_calc_umax(int patch_size, int* umax)
{
    const int half = patch_size / 2;
    int v, v0;
    const int vmax = (int)floor((double)((float)half * sqrtf(2.f) / 2 + 1));
    const int vmin = (int)ceil((double)((float)half * sqrtf(2.f) / 2));
    for (v = 0; v < half + 2; v++) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)half * half - (double)v * v));
    for (v = half, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* cv::fastAtan2 (OpenCV >= 4.6, modules/core/src/mathfuncs_core.simd.hpp, third-party: restated from its published
 * source): 7th-order odd polynomial in min/max, degrees in [0, 360). */
float efxo_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ICAngles, hpatches_description.cpp:128-162: intensity-centroid angle of a circular patch of `patch_size` around
 * (floor(x), floor(y)), cv::fastAtan2 of the integer moments.  Pixels outside the image count as 0 (the sample reads
 * out of bounds there).  kp4: n x {x, y, size, angle}; only the angle is written. */
void efxo_ic_angles(const uint8_t* img, int rows, int cols, int stride, float* kp4, int n, int patch_size)
{
    const int half = patch_size / 2;
    int* umax = (int*)malloc(sizeof(int) * (size_t)(half + 2));
    efxo_calc_umax(patch_size, umax);
    for (int i = 0; i < n; i++) {
        const int cx = (int)floorf(kp4[4 * i]), cy = (int)floorf(kp4[4 * i + 1]);
        int m01 = 0, m10 = 0;
#define PX(xx, yy) (((xx) >= 0 && (xx) < cols && (yy) >= 0 && (yy) < rows) ? (int)img[(size_t)(yy) * stride + (xx)] : 0)
        for (int u = -half; u <= half; ++u) m10 += u * PX(cx + u, cy);
        for (int v = 1; v <= half; ++v) {
            int v_sum = 0;
            const int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                const int val_plus = PX(cx + u, cy + v), val_minus = PX(cx + u, cy - v);
                v_sum += (val_plus - val_minus);
                m10 += u * (val_plus + val_minus);
            }
            m01 += v * v_sum;
        }
#undef PX
        kp4[4 * i + 3] = efxo_fast_atan2((float)m01, (float)m10);
    }
    free(umax);
}
