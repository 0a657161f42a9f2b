// bad_kernel.hip -- BAD (box average difference) descriptor for gfx950.
//
// Arithmetic: the reference CPU descriptor, modules/efficient_features/src/bad.cpp:86-157 (border test,
// rectifyBoxes), :166-251 (clamped float path), :320-405 (integer fast path), bit i -> byte i/8 MSB first.
//
// MI355X design: no global integral image and no global blur pass (the reference does both per level:
// cuda_bad.cu:350-363, cuda_efficient_features.cpp:305).  One workgroup per keypoint stages the S x S
// window that can contain all boxes of that keypoint into LDS (~3 KB of u8 from L2 instead of a
// 4 B/px global integral), optionally applies the 7x7 sigma-2 Gaussian of detectAndCompute in LDS
// (spec S6, identical op order to the oracle), builds a window-local integral image in LDS and evaluates
// all box pairs from it.  Box sums are exact integers, so the local integral gives the same sums as the
// global one; the float expressions are evaluated exactly as in bad.cpp (compile with -ffp-contract=off).

#include "efx_device.h"
#include "blur_window.h"
#include "bad_affine.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }


// computeBadResponse for a keypoint near the frame border (bad.cpp:166-251): boxes clamped to the frame, float means.
// I: window-local integral with IP ints per row (zero beyond the frame), window [wx0, wx0 + S) x [wy0, wy0 + S);
// fw x fh: integral dimensions of the whole frame.
__device__ __forceinline__ bool bad_border_bit(const Affine& A, uint2 bq, const int* I, int IP, int S, int wx0, int wy0, int fw, int fh)
{
    const float x1f = (float)(bq.x & 31u), x2f = (float)((bq.x >> 5) & 31u);
    const float y1f = (float)((bq.x >> 10) & 31u), y2f = (float)((bq.x >> 15) & 31u);
    // transform, bad.cpp:151-155: CV_ROUNDNUM(x) = (int)(x + 0.5f)
    const int cx1 = (int)((A.m00 * x1f + A.m01 * y1f + A.m02) + 0.5f);
    const int cy1 = (int)((A.m10 * x1f + A.m11 * y1f + A.m12) + 0.5f);
    const int cx2 = (int)((A.m00 * x2f + A.m01 * y2f + A.m02) + 0.5f);
    const int cy2 = (int)((A.m10 * x2f + A.m11 * y2f + A.m12) + 0.5f);
    const int r = (int)((A.s * (float)(bq.x >> 20)) + 0.5f);
    const float thr = __uint_as_float(bq.y);
    int ax1 = cx1 - r; if (ax1 < 0) ax1 = 0; else if (ax1 >= fw - 1) ax1 = fw - 2;
    int ay1 = cy1 - r; if (ay1 < 0) ay1 = 0; else if (ay1 >= fh - 1) ay1 = fh - 2;
    int ax2 = cx1 + r + 1; if (ax2 <= 0) ax2 = 1; else if (ax2 >= fw) ax2 = fw - 1;
    int ay2 = cy1 + r + 1; if (ay2 <= 0) ay2 = 1; else if (ay2 >= fh) ay2 = fh - 1;
    int lx1 = clampi(ax1 - wx0, 0, S), ly1 = clampi(ay1 - wy0, 0, S);
    int lx2 = clampi(ax2 - wx0, 0, S), ly2 = clampi(ay2 - wy0, 0, S);
    const float sum1 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
    const int area1 = (ay2 - ay1) * (ax2 - ax1);
    const float avg1 = sum1 / (float)area1;

    int bx1 = cx2 - r; if (bx1 < 0) bx1 = 0; else if (bx1 >= fw - 1) bx1 = fw - 2;
    int by1 = cy2 - r; if (by1 < 0) by1 = 0; else if (by1 >= fh - 1) by1 = fh - 2;
    int bx2 = cx2 + r + 1; if (bx2 <= 0) bx2 = 1; else if (bx2 >= fw) bx2 = fw - 1;
    int by2 = cy2 + r + 1; if (by2 <= 0) by2 = 1; else if (by2 >= fh) by2 = fh - 1;
    lx1 = clampi(bx1 - wx0, 0, S); ly1 = clampi(by1 - wy0, 0, S);
    lx2 = clampi(bx2 - wx0, 0, S); ly2 = clampi(by2 - wy0, 0, S);
    const float sum2 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
    const int area2 = (by2 - by1) * (bx2 - bx1);
    const float avg2 = sum2 / (float)area2;
    return (avg1 - avg2) <= thr;
}

// rectifyBoxes etc. for keypoint lists that do not come from the detector (bad_affine.h)
__global__ __launch_bounds__(256) void bad_affine_kernel(const float4* __restrict__ kp4, const uint8_t* __restrict__ kps5, size_t kps5_pitch, const int* __restrict__ kp_level,
                                                         const LevelTable* __restrict__ T, const uint8_t* __restrict__ img0, int pitch0,
                                                         const uint8_t* __restrict__ pyramid, int rows0, int cols0,
                                                         const int* __restrict__ d_count, int n,
                                                         float scale_factor, float reach, int smax, int sfixed, Affine* __restrict__ aff)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int count = d_count ? min(*d_count, n) : n;
    if (i >= count) return;
    int rows = rows0, cols = cols0, l = 0;
    const uint8_t* img = img0; int pitch = pitch0;
    if (kp_level) {
        l = kp_level[i]; rows = T->lv[l].rows; cols = T->lv[l].cols;
        if (l > 0) { img = pyramid + T->lv[l].img_off; pitch = T->lv[l].pitch; }
    }
    aff[i] = efx_bad_affine(efx_load_keypoint(kp4, kps5, kps5_pitch, i), img, pitch, rows, cols, l, scale_factor, reach, smax, sfixed);
}

// LDS plan (dynamic): [ I: (S+1)^2 int32, aliased by raw: (S+6) x RPB u8 | hb: HR x HP float ]   (BlurGeom, blur_window.h)
// SF != 0: every keypoint is known to need exactly an SF x SF window (size 31, scale 1 -> 48: the detector's
// keypoints), so all index arithmetic and loop bounds fold to constants.
template <bool BLUR, int SF>
__global__ __launch_bounds__(256) void bad_kernel(
    const uint8_t* __restrict__ img0, int pitch0, int rows0, int cols0,
    const uint8_t* __restrict__ pyramid, const LevelTable* __restrict__ T,
    const float4* __restrict__ kp4, const int* __restrict__ kp_level, const int* __restrict__ d_count, int n,
    float scale_factor, int smax, const BadParamsDev* __restrict__ P, const Affine* __restrict__ aff, float taps0, float taps1, float taps2, float taps3,
    uint8_t* __restrict__ desc, size_t desc_pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // the grid is sized for the capacity; only the first `count` workgroups have a keypoint.  Neighbouring keypoints go
    // to the same XCD (their windows share L2 lines): chunked over the COUNT, so that a frame with few keypoints still
    // uses all eight XCDs
    const int count = d_count ? min(*d_count, n) : n;
    if ((int)blockIdx.x >= count) return;
    const int kid = xcd_chunked(blockIdx.x, count);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;

    const Affine A = aff[kid];                                   // uniform address: scalar loads
    const uint8_t* img = img0; int pitch = pitch0, rows = rows0, cols = cols0;
    if (kp_level) {
        const int l = A.level;
        if (l > 0) { const LevelDev& L = T->lv[l]; img = pyramid + L.img_off; pitch = L.pitch; rows = L.rows; cols = L.cols; }
        else { rows = T->lv[0].rows; cols = T->lv[0].cols; }
    }
    const int nbits = P->nbits;
    const bool fits = A.S != 0;
    const int S = SF ? SF : (fits ? A.S : smax);
    const int wx0 = A.wx0, wy0 = A.wy0;

    int* I = reinterpret_cast<int*>(smem);                       // (S+1) x (S+1)
    const int IP = S + 1;
    const BlurGeom bg(S);
    uint8_t* raw = smem;                                         // aliases I (dead before I is written)
    size_t ibytes = (size_t)IP * IP * 4;
    if (BLUR && bg.raw_bytes() > ibytes) ibytes = bg.raw_bytes();
    float* hb = reinterpret_cast<float*>(smem + ((ibytes + 15) & ~(size_t)15));

    if (fits) {
        if (BLUR) {
            // 7x7 sigma-2 Gaussian of the window (spec S6) on packed fp32 FMAs; the quantised pixels go straight into
            // the integral's source (zero outside the frame)
            efx_blur_window_lds<256>(img, pitch, rows, cols, wx0, wy0, S, raw, hb, taps0, taps1, taps2, taps3, tid,
                [&](int r0, int i, int c, uint32_t pk) {
                    const int r = r0 + i;
                    const bool rin = (wy0 + r) < rows;
                    I[(r + 1) * IP + (c + 1)] = (rin && (wx0 + c) < cols) ? (int)(pk & 0xffu) : 0;
                    I[(r + 1) * IP + (c + 2)] = (rin && (wx0 + c + 1) < cols) ? (int)(pk >> 8) : 0;
                });
        } else {
            for (int r = wid; r < S; r += 4) {
                const int gy = wy0 + r;
                const uint8_t* src = img + (size_t)(gy < rows ? gy : 0) * pitch;
                for (int c = lane; c < S; c += 64) {
                    const int gx = wx0 + c;
                    I[(r + 1) * IP + (c + 1)] = (gy < rows && gx < cols) ? (int)src[gx] : 0;
                }
            }
        }
        for (int i = tid; i <= S; i += 256) { I[i] = 0; I[i * IP] = 0; }
        __syncthreads();
        // window-local integral: row prefix, then column prefix
        for (int r = tid; r < S; r += 256) {
            int* p = I + (r + 1) * IP + 1;
            int run = 0;
            for (int c = 0; c < S; c++) { run += p[c]; p[c] = run; }
        }
        __syncthreads();
        for (int c = tid; c < S; c += 256) {
            int* p = I + IP + 1 + c;
            int run = 0;
            for (int r = 0; r < S; r++) { run += p[r * IP]; p[r * IP] = run; }
        }
    }
    __syncthreads();

    const bool border = (A.border & 1) != 0;
    const int fw = cols + 1, fh = rows + 1;       // integral image dims of the full frame

    for (int b0 = 0; b0 < nbits; b0 += 256) {
        const int b = b0 + tid;
        bool bit = false;
        if (b < nbits && fits) {
            const uint2 bq = P->box[b];
            const float x1f = (float)(bq.x & 31u), x2f = (float)((bq.x >> 5) & 31u);
            const float y1f = (float)((bq.x >> 10) & 31u), y2f = (float)((bq.x >> 15) & 31u);
            // transform, bad.cpp:151-155: CV_ROUNDNUM(x) = (int)(x + 0.5f)
            const int cx1 = (int)((A.m00 * x1f + A.m01 * y1f + A.m02) + 0.5f);
            const int cy1 = (int)((A.m10 * x1f + A.m11 * y1f + A.m12) + 0.5f);
            const int cx2 = (int)((A.m00 * x2f + A.m01 * y2f + A.m02) + 0.5f);
            const int cy2 = (int)((A.m10 * x2f + A.m11 * y2f + A.m12) + 0.5f);
            const int r = (int)((A.s * (float)(bq.x >> 20)) + 0.5f);
            const float thr = __uint_as_float(bq.y);
            if (border) {
                // computeBadResponse, bad.cpp:166-251: boxes clamped to the frame, float means
                int ax1 = cx1 - r; if (ax1 < 0) ax1 = 0; else if (ax1 >= fw - 1) ax1 = fw - 2;
                int ay1 = cy1 - r; if (ay1 < 0) ay1 = 0; else if (ay1 >= fh - 1) ay1 = fh - 2;
                int ax2 = cx1 + r + 1; if (ax2 <= 0) ax2 = 1; else if (ax2 >= fw) ax2 = fw - 1;
                int ay2 = cy1 + r + 1; if (ay2 <= 0) ay2 = 1; else if (ay2 >= fh) ay2 = fh - 1;
                int lx1 = clampi(ax1 - wx0, 0, S), ly1 = clampi(ay1 - wy0, 0, S);
                int lx2 = clampi(ax2 - wx0, 0, S), ly2 = clampi(ay2 - wy0, 0, S);
                const float sum1 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
                const int area1 = (ay2 - ay1) * (ax2 - ax1);
                const float avg1 = sum1 / (float)area1;

                int bx1 = cx2 - r; if (bx1 < 0) bx1 = 0; else if (bx1 >= fw - 1) bx1 = fw - 2;
                int by1 = cy2 - r; if (by1 < 0) by1 = 0; else if (by1 >= fh - 1) by1 = fh - 2;
                int bx2 = cx2 + r + 1; if (bx2 <= 0) bx2 = 1; else if (bx2 >= fw) bx2 = fw - 1;
                int by2 = cy2 + r + 1; if (by2 <= 0) by2 = 1; else if (by2 >= fh) by2 = fh - 1;
                lx1 = clampi(bx1 - wx0, 0, S); ly1 = clampi(by1 - wy0, 0, S);
                lx2 = clampi(bx2 - wx0, 0, S); ly2 = clampi(by2 - wy0, 0, S);
                const float sum2 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
                const int area2 = (by2 - by1) * (bx2 - bx1);
                const float avg2 = sum2 / (float)area2;
                bit = (avg1 - avg2) <= thr;
            } else {
                // integer fast path, bad.cpp:365-393 (tap coordinates clamped into the frame: spec S9)
                // The window lies inside the frame (or, for frames smaller than the window, the pixels beyond the
                // frame are zero, so the integral is constant there): clamping to the frame and then to the
                // window equals clamping to the window.
                // Detector keypoints (SF != 0: size 31, scale <= 1, not within 27 px of the frame edge on this path): the
                // window is never clamped by the frame and R >= s * reach + 1 puts every tap inside it, so no clamp.
                auto wc = [&](int v) -> int { return SF ? v : clampi(v, 0, S); };
                const int lax1 = wc(cx1 - r - wx0), lay1 = wc(cy1 - r - wy0);
                const int lax2 = wc(cx1 + r + 1 - wx0), lay2 = wc(cy1 + r + 1 - wy0);
                const int lbx1 = wc(cx2 - r - wx0), lby1 = wc(cy2 - r - wy0);
                const int lbx2 = wc(cx2 + r + 1 - wx0), lby2 = wc(cy2 + r + 1 - wy0);
                const int side = 1 + (r << 1);
                const int area_resp = I[lay1 * IP + lax1] + I[lay2 * IP + lax2] - I[lay1 * IP + lax2] - I[lay2 * IP + lax1]
                                    - I[lby1 * IP + lbx1] - I[lby2 * IP + lbx2] + I[lby1 * IP + lbx2] + I[lby2 * IP + lbx1];
                bit = (float)area_resp <= (thr * (float)(side * side));
            }
        }
        // 8 consecutive box pairs -> one byte, MSB first (bad.cpp:349,368)
        const unsigned long long m = __ballot(bit);
        if ((lane & 7) == 0 && b < nbits) {
            const unsigned v = (unsigned)(m >> lane) & 0xffu;
            desc[(size_t)kid * desc_pitch + (b >> 3)] = (uint8_t)(__brev(v) >> 24);
        }
    }
}

// ================================================================================================
// The detector's keypoints (detectAndCompute: size 31, scale 1 -> 48 x 48 window, blurred level): the same arithmetic
// as bad_kernel<true, 48>, with everything that does not depend on the keypoint taken out of the workgroup and the LDS
// traffic trimmed (the kernel is bound by the LDS pipe in its integral and box phases, by VALU issue in the blur:
// profiles/r02_bad_phases.txt):
//   * the box pairs come from BadParamsDev::ubox (scaled radius, integral strides and thr * side^2 precomputed on the
//     host with bad.cpp's float expressions), the two boxes of a pair go through the affine map as ONE packed-fp32
//     sequence (v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations, two per instruction);
//   * the level's image pointer / pitch / size ride in the Affine record (one dependent load before the window loads);
//   * the blur's column pass writes a u8 plane (2 pixels per ds_write_b16 instead of two ds_write_b32); the integral's
//     row pass reads a row as three conflict-free ds_read_b128 and unpacks with SDWA adds; I aliases the dead hb;
//   * a wave packs its 64 bits with two scalar bit reversals and one 8-byte store.
// Measured and dropped: blur item lists limited to the disc the boxes can reach (71 % of the pixels, but the 3-row apron
// leaves 195 of 216 row-pass items: no wave is saved and the compacted order costs bank conflicts).
// ================================================================================================
__global__ __launch_bounds__(256) void bad_det_kernel(
    const int* __restrict__ d_count, int n, const BadParamsDev* __restrict__ P, const Affine* __restrict__ aff,
    float taps0, float taps1, float taps2, float taps3, uint8_t* __restrict__ desc, size_t desc_pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 48, IP = S + 1;
    constexpr int PIX_OFF = 3248, HB_OFF = PIX_OFF + S * S;      // raw [0, 3248) | pix [3248, 5552) | hb / I [5552, ...)
    static_assert(HB_OFF % 16 == 0 && PIX_OFF % 16 == 0, "LDS regions are 16-byte aligned");

    const int tid = threadIdx.x;
    const int count = d_count ? min(*d_count, n) : n;
    if ((int)blockIdx.x >= count) return;
    const int kid = xcd_chunked(blockIdx.x, count);
    const int lane = tid & 63;

    const uint4 q0 = P->ubox[tid], q1 = P->ubox[256 + tid];     // requested first: they do not depend on the keypoint
    const Affine A = aff[kid];                                   // uniform address: scalar loads
    const uint8_t* img = A.img; const int pitch = A.pitch, rows = A.rows, cols = A.cols;
    const int nbits = P->nbits;
    const bool fits = A.S != 0;
    const int wx0 = A.wx0, wy0 = A.wy0;

    uint8_t* raw = smem;
    uint8_t* pix = smem + PIX_OFF;
    float* hb = reinterpret_cast<float*>(smem + HB_OFF);
    int* I = reinterpret_cast<int*>(smem + HB_OFF);              // (S+1) x (S+1), aliases hb (dead after the column pass)

    if (fits) {
        const bool inside = wx0 + S <= cols && wy0 + S <= rows;  // frames smaller than the window: zero beyond the frame
        efx_blur_window_lds<256>(img, pitch, rows, cols, wx0, wy0, S, raw, hb, taps0, taps1, taps2, taps3, tid,
            [&](int r0, int i, int c, uint32_t pk) {
                if (!inside) {
                    const bool rin = (wy0 + r0 + i) < rows;
                    pk &= ((rin && (wx0 + c) < cols) ? 0xffu : 0u) | ((rin && (wx0 + c + 1) < cols) ? 0xff00u : 0u);
                }
                *reinterpret_cast<uint16_t*>(pix + r0 * S + c + i * S) = (uint16_t)pk;
            });
        __syncthreads();
        // window-local integral: row prefix from the u8 plane (wave 0; the zero border is written by wave 1), then column prefix
        if (tid < S) {
            const uint4* row = reinterpret_cast<const uint4*>(pix + tid * S);
            const uint4 w0 = row[0], w1 = row[1], w2 = row[2];
            const uint32_t w[12] = { w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w };
            int* p = I + (tid + 1) * IP + 1;
            int run = 0;
#pragma unroll
            for (int c = 0; c < S; c++) { run += (int)((w[c >> 2] >> (8 * (c & 3))) & 0xffu); p[c] = run; }
        } else if (tid >= 64 && tid < 64 + IP) {
            I[tid - 64] = 0; I[(tid - 64) * IP] = 0;
        }
        __syncthreads();
        if (tid < S) {
            int* p = I + IP + 1 + tid;
            int run = 0;
#pragma unroll
            for (int r = 0; r < S; r++) { run += p[r * IP]; p[r * IP] = run; }
        }
    }
    __syncthreads();

    const bool border = (A.border & 1) != 0;
    const int fw = cols + 1, fh = rows + 1;       // integral image dims of the full frame
    const bool wide_store = ((((uintptr_t)desc) | desc_pitch) & 7u) == 0;
    const int wbase = -(wy0 * IP + wx0) * 4;

    for (int b0 = 0; b0 < nbits; b0 += 256) {
        const int b = b0 + tid;
        bool bit = false;
        if (fits) {
            if (border) {
                const uint2 bq = P->box[b];
                const float x1f = (float)(bq.x & 31u), x2f = (float)((bq.x >> 5) & 31u);
                const float y1f = (float)((bq.x >> 10) & 31u), y2f = (float)((bq.x >> 15) & 31u);
                const int cx1 = (int)((A.m00 * x1f + A.m01 * y1f + A.m02) + 0.5f);
                const int cy1 = (int)((A.m10 * x1f + A.m11 * y1f + A.m12) + 0.5f);
                const int cx2 = (int)((A.m00 * x2f + A.m01 * y2f + A.m02) + 0.5f);
                const int cy2 = (int)((A.m10 * x2f + A.m11 * y2f + A.m12) + 0.5f);
                const int r = (int)((A.s * (float)(bq.x >> 20)) + 0.5f);
                const float thr = __uint_as_float(bq.y);
                // computeBadResponse, bad.cpp:166-251: boxes clamped to the frame, float means
                int ax1 = cx1 - r; if (ax1 < 0) ax1 = 0; else if (ax1 >= fw - 1) ax1 = fw - 2;
                int ay1 = cy1 - r; if (ay1 < 0) ay1 = 0; else if (ay1 >= fh - 1) ay1 = fh - 2;
                int ax2 = cx1 + r + 1; if (ax2 <= 0) ax2 = 1; else if (ax2 >= fw) ax2 = fw - 1;
                int ay2 = cy1 + r + 1; if (ay2 <= 0) ay2 = 1; else if (ay2 >= fh) ay2 = fh - 1;
                int lx1 = clampi(ax1 - wx0, 0, S), ly1 = clampi(ay1 - wy0, 0, S);
                int lx2 = clampi(ax2 - wx0, 0, S), ly2 = clampi(ay2 - wy0, 0, S);
                const float sum1 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
                const int area1 = (ay2 - ay1) * (ax2 - ax1);
                const float avg1 = sum1 / (float)area1;

                int bx1 = cx2 - r; if (bx1 < 0) bx1 = 0; else if (bx1 >= fw - 1) bx1 = fw - 2;
                int by1 = cy2 - r; if (by1 < 0) by1 = 0; else if (by1 >= fh - 1) by1 = fh - 2;
                int bx2 = cx2 + r + 1; if (bx2 <= 0) bx2 = 1; else if (bx2 >= fw) bx2 = fw - 1;
                int by2 = cy2 + r + 1; if (by2 <= 0) by2 = 1; else if (by2 >= fh) by2 = fh - 1;
                lx1 = clampi(bx1 - wx0, 0, S); ly1 = clampi(by1 - wy0, 0, S);
                lx2 = clampi(bx2 - wx0, 0, S); ly2 = clampi(by2 - wy0, 0, S);
                const float sum2 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
                const int area2 = (by2 - by1) * (bx2 - bx1);
                const float avg2 = sum2 / (float)area2;
                bit = (avg1 - avg2) <= thr;
            } else {
                // integer fast path, bad.cpp:365-393.  Not within 27 px of the frame edge: the window is not clamped by
                // the frame and R >= s * reach + 1 puts every tap inside it (no clamps, spec S9 is vacuous here).
                const uint4 q = b0 == 0 ? q0 : q1;
                const efx_f32x2 xs = { (float)(q.x & 0xffu), (float)((q.x >> 16) & 0xffu) };          // x1, x2
                const efx_f32x2 ys = { (float)((q.x >> 8) & 0xffu), (float)(q.x >> 24) };             // y1, y2
                // transform, bad.cpp:151-155: ((m00 x + m01 y) + m02) + 0.5f, truncated -- both boxes per instruction
                const efx_f32x2 half2 = { 0.5f, 0.5f };
                const efx_f32x2 cxf = (((efx_f32x2)(A.m00) * xs + (efx_f32x2)(A.m01) * ys) + (efx_f32x2)(A.m02)) + half2;
                const efx_f32x2 cyf = (((efx_f32x2)(A.m10) * xs + (efx_f32x2)(A.m11) * ys) + (efx_f32x2)(A.m12)) + half2;
                const int cx1 = (int)cxf.x, cx2 = (int)cxf.y, cy1 = (int)cyf.x, cy2 = (int)cyf.y;
                // byte address of integral entry (cy - r' - wy0, cx - r' - wx0): (cy * IP + cx) * 4 + [(-r') (IP + 1) 4 + wbase]
                const int pbase = (int)q.y * (IP + 1) + wbase;
                const int side4 = (int)(q.z & 0xffffu), sideIP4 = (int)(q.z >> 16);
                const int a_tl = (cy1 * IP + cx1) * 4 + pbase, b_tl = (cy2 * IP + cx2) * 4 + pbase;
                const unsigned char* Ib = smem + HB_OFF;
                auto at = [&](int off) -> int { return *reinterpret_cast<const int*>(Ib + off); };
                const int area_resp = at(a_tl) + at(a_tl + side4 + sideIP4) - at(a_tl + side4) - at(a_tl + sideIP4)
                                    - at(b_tl) - at(b_tl + side4 + sideIP4) + at(b_tl + side4) + at(b_tl + sideIP4);
                bit = (float)area_resp <= __uint_as_float(q.w);
            }
        }
        // 64 consecutive box pairs -> 8 bytes, bit i -> byte i / 8, MSB first (bad.cpp:349,368)
        const unsigned long long m = __ballot(bit);
        uint8_t* out = desc + (size_t)kid * desc_pitch + (b0 >> 3) + ((tid >> 6) << 3);
        if (wide_store) {
            if (lane == 0) {
                const uint32_t lo = __builtin_bswap32(__brev((uint32_t)m)), hi = __builtin_bswap32(__brev((uint32_t)(m >> 32)));
                *reinterpret_cast<uint2*>(out) = make_uint2(lo, hi);
            }
        } else if ((lane & 7) == 0) {
            const unsigned v = (unsigned)(m >> lane) & 0xffu;
            out[lane >> 3] = (uint8_t)(__brev(v) >> 24);
        }
    }
}

// ================================================================================================
// computeAsync on detector-sized keypoints WITHOUT the blur (size 31, describer scale such that the window is 48 x 48:
// BASELINE config C3, cuda_efficient_features.cpp:220-223 -> cuda_bad.cpp:46-70 -> computeBADKernel cuda_bad.cu:246-316).
// No blur means no phase that wants 256 lanes, so ONE WAVE owns a keypoint and a 256-thread workgroup runs four of them
// independently: no workgroup barrier anywhere (a wave's LDS operations complete in order), and the waves of a CU sit in
// different phases -- window loads (vector memory), prefix sums (VALU + LDS writes), box gathers (LDS reads) overlap.
//   window   48 rows x 13 aligned dwords through a buffer resource (range-checked: zero beyond the image), row-coalesced
//            (13 consecutive lanes = one row), into LDS rows of 13 dwords (odd pitch: conflict-free row-per-lane reads)
//   rows     lane r: its row as 13 ds_read_b32 + v_alignbyte, exclusive prefix on SDWA byte adds, written as 25 packed u16
//            pairs (a row prefix is at most 48 * 255): P'[r][x] = sum of the row's pixels left of column x
//   columns  lane j < 25: columns 2j, 2j + 1 of the integral J (pitch 50 ints, so the pair is one aligned ds_write_b64);
//            one ds_read_b32 per row.  P' aliases the SECOND half of J: row r + 1 of J never reaches a P' row that is still
//            to be read (200 (r + 2) <= 5000 + 100 (r + 1)), so a wave needs 9.8 KB and a CU holds 16 waves
//   boxes    the per-pair table of the detector path (BadParamsDev::ubox), 64 pairs per step, the wave's ballot is the
//            descriptor word; lane i keeps word i and the descriptor leaves as one 8-byte store per lane
// ================================================================================================
#define BAD_RAW_JP 50                                     // ints per integral row
#define BAD_RAW_P_OFF 5000                                // byte offset of the u16 row-prefix plane inside the integral's storage
#define BAD_RAW_WAVE_LDS 9808                             // 49 * 50 * 4 = 9800, rounded to 16

__global__ __launch_bounds__(256) void bad_raw_kernel(
    const int* __restrict__ d_count, int n, const BadParamsDev* __restrict__ P, const Affine* __restrict__ aff,
    uint8_t* __restrict__ desc, size_t desc_pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 48, JP = BAD_RAW_JP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = d_count ? min(*d_count, n) : n;
    const int ngroups = (count + 3) >> 2;
    if ((int)blockIdx.x >= ngroups) return;
    const int kid = xcd_chunked(blockIdx.x, ngroups) * 4 + wave;      // neighbouring keypoints share an XCD's L2
    if (kid >= count) return;                                         // wave-uniform

    unsigned char* wbuf = smem + wave * BAD_RAW_WAVE_LDS;
    uint32_t* rawdw = reinterpret_cast<uint32_t*>(wbuf);               // 48 x 13 dwords
    uint32_t* Pq = reinterpret_cast<uint32_t*>(wbuf + BAD_RAW_P_OFF);  // 48 x 25 dwords (50 u16)
    int* J = reinterpret_cast<int*>(wbuf);                             // 49 x 50 ints

    const Affine A = aff[kid];                                         // wave-uniform address: scalar loads
    const uint8_t* img = A.img; const int pitch = A.pitch, rows = A.rows, cols = A.cols;
    const int nbits = P->nbits;
    const bool fits = A.S != 0;
    const int wx0 = A.wx0, wy0 = A.wy0;

    if (fits) {
        const bool aligned = ((((uintptr_t)img) | (uintptr_t)pitch) & 3u) == 0;
        const int off = aligned ? (wx0 & 3) : 0;
        if (aligned) {
            // rows are pitch bytes apart and at least roundup4(cols) of them are memory we may read (our own levels are
            // padded; a caller's 4-byte aligned image has pitch >= roundup4(cols))
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, (rows - 1) * pitch + ((cols + 3) & ~3), 0x00020000);
            const int gbase = wy0 * pitch + (wx0 & ~3);
#pragma unroll
            for (int it = 0; it < 10; it++) {
                const int idx = it * 64 + lane;
                const int r = idx / 13, k = idx - r * 13;
                if (idx < S * 13) rawdw[idx] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, gbase + r * pitch + 4 * k, 0, 0);
            }
        } else {
            // caller's image with an unaligned base or pitch: bytes, zero beyond the frame
            uint8_t* rawb = wbuf;
            for (int idx = lane; idx < S * 52; idx += 64) {
                const int r = idx / 52, c = idx - r * 52;
                const int gy = wy0 + r, gx = wx0 + c;
                rawb[idx] = (gy < rows && gx < cols) ? img[(size_t)gy * pitch + gx] : (uint8_t)0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- row prefix (lane = row) ----
        uint32_t out[25];
        if (lane < S) {
            uint32_t d[13], w[12];
#pragma unroll
            for (int k = 0; k < 13; k++) d[k] = rawdw[lane * 13 + k];
#pragma unroll
            for (int k = 0; k < 12; k++) w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], off);
            if (!(wx0 + S <= cols && wy0 + S <= rows)) {
                // frames smaller than the window: zero beyond the frame (the dword loads may have picked up the next row)
                const int vc = min(max(cols - wx0, 0), S);
                const bool rin = (wy0 + lane) < rows;
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const int left = vc - 4 * k;
                    const uint32_t m = !rin || left <= 0 ? 0u : (left >= 4 ? 0xffffffffu : ((1u << (8 * left)) - 1u));
                    w[k] &= m;
                }
            }
            int run = 0;
#pragma unroll
            for (int j = 0; j < 25; j++) {
                const int lo = run;
                if (2 * j < S) run += (int)((w[(2 * j) >> 2] >> (8 * ((2 * j) & 3))) & 0xffu);
                const int hi = run;
                if (2 * j + 1 < S) run += (int)((w[(2 * j + 1) >> 2] >> (8 * ((2 * j + 1) & 3))) & 0xffu);
                out[j] = (uint32_t)lo | ((uint32_t)hi << 16);
            }
        }
        // every lane has read its raw row (program order) before the prefix plane -- which does not overlap the raw rows --
        // is written
        if (lane < S) {
#pragma unroll
            for (int j = 0; j < 25; j++) Pq[lane * 25 + j] = out[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- column prefix (lane = column pair) ----
        if (lane < 25) {
            *reinterpret_cast<uint2*>(J + 2 * lane) = make_uint2(0u, 0u);
            int run0 = 0, run1 = 0;
#pragma unroll
            for (int r0 = 0; r0 < S; r0 += 8) {
                uint32_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = Pq[(r0 + i) * 25 + lane];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    run0 += (int)(v[i] & 0xffffu); run1 += (int)(v[i] >> 16);
                    *reinterpret_cast<uint2*>(J + (r0 + i + 1) * JP + 2 * lane) = make_uint2((uint32_t)run0, (uint32_t)run1);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    const bool border = (A.border & 1) != 0;
    const int fw = cols + 1, fh = rows + 1;
    const int wbase = -(wy0 * JP + wx0) * 4;
    uint32_t mlo = 0u, mhi = 0u;
    for (int it = 0; it * 64 < nbits; it++) {
        const int b = it * 64 + lane;
        bool bit = false;
        if (fits) {
            if (border) {
                bit = bad_border_bit(A, P->box[b], J, JP, S, wx0, wy0, fw, fh);
            } else {
                // integer fast path, bad.cpp:365-393; the window holds every tap (bad_det_kernel has the argument)
                const uint4 q = P->ubox[b];
                const efx_f32x2 xs = { (float)(q.x & 0xffu), (float)((q.x >> 16) & 0xffu) };          // x1, x2
                const efx_f32x2 ys = { (float)((q.x >> 8) & 0xffu), (float)(q.x >> 24) };             // y1, y2
                const efx_f32x2 half2 = { 0.5f, 0.5f };
                const efx_f32x2 cxf = (((efx_f32x2)(A.m00) * xs + (efx_f32x2)(A.m01) * ys) + (efx_f32x2)(A.m02)) + half2;
                const efx_f32x2 cyf = (((efx_f32x2)(A.m10) * xs + (efx_f32x2)(A.m11) * ys) + (efx_f32x2)(A.m12)) + half2;
                const int cx1 = (int)cxf.x, cx2 = (int)cxf.y, cy1 = (int)cyf.x, cy2 = (int)cyf.y;
                const int pbase = (int)q.y * (JP + 1) + wbase;                // q.y = -4 r'
                const int side4 = (int)(q.z & 0xffffu), sideJ4 = side4 * JP;
                const int a_tl = (cy1 * JP + cx1) * 4 + pbase, b_tl = (cy2 * JP + cx2) * 4 + pbase;
                auto at = [&](int o) -> int { return *reinterpret_cast<const int*>(wbuf + o); };
                const int area_resp = at(a_tl) + at(a_tl + side4 + sideJ4) - at(a_tl + side4) - at(a_tl + sideJ4)
                                    - at(b_tl) - at(b_tl + side4 + sideJ4) + at(b_tl + side4) + at(b_tl + sideJ4);
                bit = (float)area_resp <= __uint_as_float(q.w);
            }
        }
        const unsigned long long m = __ballot(bit);
        if (lane == it) { mlo = (uint32_t)m; mhi = (uint32_t)(m >> 32); }
    }
    // bit i -> byte i / 8, MSB first (bad.cpp:349,368): lane i holds bits 64 i .. 64 i + 63
    if (lane * 64 < nbits) {
        uint8_t* o = desc + (size_t)kid * desc_pitch + lane * 8;
        const uint32_t lo = __builtin_bswap32(__brev(mlo)), hi = __builtin_bswap32(__brev(mhi));
        if (((((uintptr_t)desc) | desc_pitch) & 7u) == 0) *reinterpret_cast<uint2*>(o) = make_uint2(lo, hi);
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) { o[k] = (uint8_t)(lo >> (8 * k)); o[4 + k] = (uint8_t)(hi >> (8 * k)); }
        }
    }
}

} // namespace

void efx_gaussian_taps_host(float taps[7])
{
    // cv::getGaussianKernel(7, 2) (spec S6): exp(-(i-3)^2 / (2 sigma^2)) normalised in double, stored as float
    double e[7], sum = 0;
    for (int i = 0; i < 7; i++) {
        const double x = (double)(i - 3);
        e[i] = exp(-(x * x) / (2.0 * 2.0 * 2.0));
        sum += e[i];
    }
    for (int i = 0; i < 7; i++) taps[i] = (float)(e[i] / sum);
}

hipError_t efx_launch_bad(const DescribeLaunch& a, const BadParamsDev* d_params, float reach, hipStream_t stream)
{
    if (a.n <= 0) return hipSuccess;
    const float max_size = a.max_size > 0.f ? a.max_size : (float)EFX_PATCH_SIZE;
    const int S = efx_bad_smax_for(max_size, a.scale_factor, reach);
    size_t lds = (size_t)(S + 1) * (S + 1) * 4;
    if (a.blur) {
        const BlurGeom bg(S);
        if (bg.raw_bytes() > lds) lds = bg.raw_bytes();
        lds = (lds + 15) & ~(size_t)15;
        lds += bg.hb_bytes();
    }
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024 - 64) return hipErrorInvalidValue;    // keypoint window does not fit in LDS
    float t[7];
    efx_gaussian_taps_host(t);
    Affine* aff = static_cast<Affine*>(a.bad_affine);
    const int sfixed = (S == 48 && a.uniform_size) ? 48 : 0;
    if (!a.affine_ready)         // detectAndCompute: angle_kernel has left the records (DetectLaunch::bad_affine)
        hipLaunchKernelGGL(bad_affine_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a.kp4, a.kps5, a.kps5_pitch, a.kp_level, a.d_table, a.img0, a.pitch0, a.pyramid, a.rows0, a.cols0,
                           a.d_count, a.n, a.scale_factor, reach, S, sfixed, aff);
    if (a.blur) {
        if (S == 48 && a.uniform_size && max_size == (float)EFX_PATCH_SIZE && a.kp_level && a.bad_det_tables) {
            // detector keypoints: the per-pair table of BadParamsDev was built for exactly this s; LDS: raw | pix | hb / I
            const size_t lds_det = 3248 + 48 * 48 + BlurGeom(48).hb_bytes();
            hipLaunchKernelGGL(bad_det_kernel, dim3(a.n), dim3(256), lds_det, stream, a.d_count, a.n, d_params, aff, t[0], t[1], t[2], t[3],
                               a.desc, a.desc_pitch);
            return hipGetLastError();
        }
        if (S == 48 && a.uniform_size) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<true, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((bad_kernel<true, 48>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                               a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                               a.desc, a.desc_pitch);
            return hipGetLastError();
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((bad_kernel<true, 0>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                           a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                           a.desc, a.desc_pitch);
    } else {
        if (S == 48 && a.uniform_size && max_size == (float)EFX_PATCH_SIZE && a.bad_det_tables && !a.bad_no_raw) {
            // computeAsync on detector-sized keypoints: a wave per keypoint, four keypoints per workgroup
            hipLaunchKernelGGL(bad_raw_kernel, dim3((a.n + 3) / 4), dim3(256), 4 * BAD_RAW_WAVE_LDS, stream, a.d_count, a.n, d_params, aff,
                               a.desc, a.desc_pitch);
            return hipGetLastError();
        }
        if (S == 48 && a.uniform_size) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<false, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((bad_kernel<false, 48>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                               a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                               a.desc, a.desc_pitch);
            return hipGetLastError();
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((bad_kernel<false, 0>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                           a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                           a.desc, a.desc_pitch);
    }
    return hipGetLastError();
}
