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
#include <algorithm>
#include "blur_window.h"
#include "bad_affine.h"
#include <stdlib.h>

namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }


// computeBadResponse for a keypoint near the frame border (bad.cpp:166-251): boxes clamped to the frame, float means.
// I: window-local integral with IP ints per row (zero beyond the frame), window [wx0, wx0 + S) x [wy0, wy0 + S);
// fw x fh: integral dimensions of the whole frame.
// T = uint16_t: the integral is kept modulo 2^16 (every box sum is below 2^16: bad_raw_kernel), the sums are reduced likewise.
template <class T>
__device__ __forceinline__ bool bad_border_bit(const Affine& A, uint2 bq, const T* I, int IP, int S, int wx0, int wy0, int fw, int fh)
{
    constexpr int M = sizeof(T) == 2 ? 0xffff : -1;
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
    const float sum1 = (float)(((int)I[ly1 * IP + lx1] + (int)I[ly2 * IP + lx2] - (int)I[ly1 * IP + lx2] - (int)I[ly2 * IP + lx1]) & M);
    const int area1 = (ay2 - ay1) * (ax2 - ax1);
    const float avg1 = sum1 / (float)area1;

    int bx1 = cx2 - r; if (bx1 < 0) bx1 = 0; else if (bx1 >= fw - 1) bx1 = fw - 2;
    int by1 = cy2 - r; if (by1 < 0) by1 = 0; else if (by1 >= fh - 1) by1 = fh - 2;
    int bx2 = cx2 + r + 1; if (bx2 <= 0) bx2 = 1; else if (bx2 >= fw) bx2 = fw - 1;
    int by2 = cy2 + r + 1; if (by2 <= 0) by2 = 1; else if (by2 >= fh) by2 = fh - 1;
    lx1 = clampi(bx1 - wx0, 0, S); ly1 = clampi(by1 - wy0, 0, S);
    lx2 = clampi(bx2 - wx0, 0, S); ly2 = clampi(by2 - wy0, 0, S);
    const float sum2 = (float)(((int)I[ly1 * IP + lx1] + (int)I[ly2 * IP + lx2] - (int)I[ly1 * IP + lx2] - (int)I[ly2 * IP + lx1]) & M);
    const int area2 = (by2 - by1) * (bx2 - bx1);
    const float avg2 = sum2 / (float)area2;
    return (avg1 - avg2) <= thr;
}

// The integral of the detector-sized kernels (bad_det_kernel, bad_raw_kernel): u16 entries modulo 2^16, BAD_J_PITCH per row
#define BAD_J_PITCH 50

// One box pair of the integer fast path (bad.cpp:365-393) from the per-pair table BadParamsDev::ubox: both boxes go through
// the affine map as ONE packed-fp32 sequence (bad.cpp:151-155: ((m00 x + m01 y) + m02) + 0.5f, truncated -- the same IEEE
// operations, two per instruction); Jb = byte address of the modulo-2^16 integral, wbase = -(wy0 JP + wx0) 2.
//   byte address of entry (cy - r' - wy0, cx - r' - wx0) = (cy JP + cx) 2 + [.y + wbase],  .y = -2 r' (JP + 1)
//   .z = 2 side | (2 JP side) << 16: byte strides to the box's right / lower corners;  .w = bits of thr * side^2
struct BadTaps { int a_tl, b_tl, side2, sideJ2; };        // byte offsets into the integral of one box pair
__device__ __forceinline__ BadTaps bad_ubox_taps(const Affine& A, uint4 q, int wbase)
{
    constexpr int JP = BAD_J_PITCH;
    const efx_f32x2 xs = { (float)(q.x & 0xffu), (float)((q.x >> 16) & 0xffu) };          // x1, x2
    const efx_f32x2 ys = { (float)((q.x >> 8) & 0xffu), (float)(q.x >> 24) };             // y1, y2
    const efx_f32x2 half2 = { 0.5f, 0.5f };
    const efx_f32x2 cxf = (((efx_f32x2)(A.m00) * xs + (efx_f32x2)(A.m01) * ys) + (efx_f32x2)(A.m02)) + half2;
    const efx_f32x2 cyf = (((efx_f32x2)(A.m10) * xs + (efx_f32x2)(A.m11) * ys) + (efx_f32x2)(A.m12)) + half2;
    const int cx1 = (int)cxf.x, cx2 = (int)cxf.y, cy1 = (int)cyf.x, cy2 = (int)cyf.y;
    const int pbase = (int)q.y + wbase;
    BadTaps t;
    t.side2 = (int)(q.z & 0xffffu); t.sideJ2 = (int)(q.z >> 16);
    // 24-bit multiplies (v_mul_lo_u32 is quarter rate): level coordinates are below 2^15
    t.a_tl = (__mul24(cy1, JP) + cx1) * 2 + pbase; t.b_tl = (__mul24(cy2, JP) + cx2) * 2 + pbase;
    return t;
}
__device__ __forceinline__ bool bad_taps_bit(const BadTaps& t, uint32_t thr_bits, const unsigned char* Jb)
{
    auto at = [&](int o) -> int { return (int)*reinterpret_cast<const uint16_t*>(Jb + o); };
    const int sa = (at(t.a_tl) + at(t.a_tl + t.side2 + t.sideJ2) - at(t.a_tl + t.side2) - at(t.a_tl + t.sideJ2)) & 0xffff;
    const int sb = (at(t.b_tl) + at(t.b_tl + t.side2 + t.sideJ2) - at(t.b_tl + t.side2) - at(t.b_tl + t.sideJ2)) & 0xffff;
    return (float)(sa - sb) <= __uint_as_float(thr_bits);
}
__device__ __forceinline__ bool bad_ubox_bit(const Affine& A, uint4 q, const unsigned char* Jb, int wbase)
{
    return bad_taps_bit(bad_ubox_taps(A, q, wbase), q.w, Jb);
}
// The same with the integral's LDS byte address folded into the taps' base, held in a VECTOR register (vbase = LDS address of
// the plane + wbase): per box pair the wave-uniform base would otherwise be added from a scalar register nine times, and a
// scalar source halves the rate of v_add_u32 (profiles/r04_valu_rate.txt; 98 of bad_raw_kernel<8>'s 1762 VALU instructions)
__device__ __forceinline__ bool bad_ubox_bit_v(const Affine& A, uint4 q, int vbase)
{
    const BadTaps t = bad_ubox_taps(A, q, vbase);
    typedef __attribute__((address_space(3))) const uint16_t lds_u16;
    auto at = [&](int o) -> int { return (int)*reinterpret_cast<lds_u16*>((uintptr_t)(uint32_t)o); };
    const int sa = (at(t.a_tl) + at(t.a_tl + t.side2 + t.sideJ2) - at(t.a_tl + t.side2) - at(t.a_tl + t.sideJ2)) & 0xffff;
    const int sb = (at(t.b_tl) + at(t.b_tl + t.side2 + t.sideJ2) - at(t.b_tl + t.side2) - at(t.b_tl + t.sideJ2)) & 0xffff;
    return (float)(sa - sb) <= __uint_as_float(q.w);
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
//     row pass reads a row as three conflict-free ds_read_b128 and unpacks with SDWA adds; the integral is kept modulo
//     2^16 (round 3: box sums are below 2^16), two columns per dword, column prefix in place on v_pk_add_u16 -- half the
//     LDS traffic of the int32 integral; it aliases the dead hb;
//   * a wave packs its 64 bits with two scalar bit reversals and one 8-byte store.
// Measured and dropped: blur item lists limited to the disc the boxes can reach (71 % of the pixels, but the 3-row apron
// leaves 195 of 216 row-pass items: no wave is saved and the compacted order costs bank conflicts).
// ================================================================================================
// INVESTIGATION (-DBAD_DET_STOP=n builds, tools/microbench/bad_phases.sh): the kernel returns after phase n, so that the
// instruction counters of successive builds give the phases' shares (results are NOT valid)
#ifdef BAD_DET_STOP
#define BAD_DET_PHASE(n) do { if (BAD_DET_STOP == (n)) return; } while (0)
#else
#define BAD_DET_PHASE(n) do { } while (0)
#endif
__global__ __launch_bounds__(256) void bad_det_kernel(
    const int* __restrict__ d_count, int n, const BadParamsDev* __restrict__ P, const Affine* __restrict__ aff,
    float taps0, float taps1, float taps2, float taps3, uint8_t* __restrict__ desc, size_t desc_pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 48, JP = BAD_J_PITCH, JD = JP / 2;
    constexpr int PIX_OFF = 3248, HB_OFF = PIX_OFF + S * S;      // raw [0, 3248) | pix [3248, 5552) | hb / J [5552, ...)
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
    // The window-local integral, MODULO 2^16 (every box of the table has at most 16 x 16 pixels -- BadParamsDev::ubox_max_side,
    // checked by the launcher -- so a box sum is below 2^16 and (tl + br - tr - bl) mod 2^16 IS the sum): 49 rows of 50 u16,
    // two columns per dword; aliases hb (dead after the blur's column pass)
    uint32_t* Jd = reinterpret_cast<uint32_t*>(smem + HB_OFF);
    const uint16_t* J = reinterpret_cast<const uint16_t*>(smem + HB_OFF);

    // A workgroup's life is a chain of dependent phases (record -> window loads -> blur rows -> blur columns -> integral rows
    // -> integral columns -> boxes), a CU holds eight workgroups, and the kernel takes (keypoints / 2048) x that life time:
    // what shortens the chain shortens the kernel, whatever the instruction counts (the int32 -> u16 integral halved the
    // integral's LDS traffic and changed nothing).  So: the box taps' addresses -- they need the record, not the pixels --
    // are worked out while the window loads are in flight; the integral's prefix chains are cut in two (two lanes per row,
    // two lanes per column pair, the second adds the first one's total).
    const int wbase = -(wy0 * JP + wx0) * 2;
    const bool border = (A.border & 1) != 0;
    BadTaps t0 = { 0, 0, 0, 0 }, t1 = { 0, 0, 0, 0 };
    if (fits) {
        const bool inside = wx0 + S <= cols && wy0 + S <= rows;  // frames smaller than the window: zero beyond the frame
        efx_blur_window_lds<256>(img, pitch, rows, cols, wx0, wy0, S, raw, hb, taps0, taps1, taps2, taps3, tid,
            [&](int r0, int i, int c, uint32_t pk) {
                if (!inside) {
                    const bool rin = (wy0 + r0 + i) < rows;
                    pk &= ((rin && (wx0 + c) < cols) ? 0xffu : 0u) | ((rin && (wx0 + c + 1) < cols) ? 0xff00u : 0u);
                }
                *reinterpret_cast<uint16_t*>(pix + r0 * S + c + i * S) = (uint16_t)pk;
            },
            [&]() {
                // pinned BEHIND the window loads: the taps are pure arithmetic and the same on every path of the loader, so
                // the compiler hoists them above it -- where they wait for the table's loads before the window's loads have
                // even been issued -- unless something they depend on is defined here
                int wb = wbase;
                asm volatile("" : "+s"(wb) : : "memory");
                if (!border) { t0 = bad_ubox_taps(A, q0, wb); if (nbits > 256) t1 = bad_ubox_taps(A, q1, wb); }
            });
#ifdef BAD_DET_STOP
        if (BAD_DET_STOP < 2) return;                           // the blur returned early (blur_window.h)
#endif
        BAD_DET_PHASE(2);                                       // record, window loads, taps, blur
        __syncthreads();
        // row prefix from the u8 plane: P'[r][x] = sum of row r left of column x, x = 0 .. 49, as 25 packed pairs into plane
        // row r + 1.  Two lanes per row (waves 0 and 1: 96 lanes), 24 pixels each; the second adds the first one's total
        // (its neighbour lane: DPP row_shr:1).  Wave 2 writes the zero row.
        if (tid < 2 * S) {
            const int row = tid >> 1, half = tid & 1;
            const uint2* src = reinterpret_cast<const uint2*>(pix + row * S + 24 * half);
            const uint2 w0 = src[0], w1 = src[1], w2 = src[2];
            const uint32_t w[6] = { w0.x, w0.y, w1.x, w1.y, w2.x, w2.y };
            uint32_t out[12];
            int run = 0;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const int lo = run;
                run += (int)((w[(2 * j) >> 2] >> (8 * ((2 * j) & 3))) & 0xffu);
                const int hi = run;
                run += (int)((w[(2 * j + 1) >> 2] >> (8 * ((2 * j + 1) & 3))) & 0xffu);
                out[j] = (uint32_t)lo | ((uint32_t)hi << 16);
            }
            const int left = __builtin_amdgcn_update_dpp(0, run, 0x111, 0xf, 0xf, true);     // row_shr:1: the total of lane - 1
            uint32_t* p = Jd + (row + 1) * JD + 12 * half;
            if (half) {
                const u16x2 off = { (unsigned short)left, (unsigned short)left };
#pragma unroll
                for (int j = 0; j < 12; j++) p[j] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, out[j]) + off));
                p[12] = (uint32_t)(left + run);                   // x = 48: the row total (x = 49 is padding)
            } else {
#pragma unroll
                for (int j = 0; j < 12; j++) p[j] = out[j];
            }
        } else if (tid >= 128 && tid < 128 + JD) {
            Jd[tid - 128] = 0u;
        }
        BAD_DET_PHASE(3);                                       // + row prefix
        __syncthreads();
        // column prefix IN PLACE (v_pk_add_u16, modulo 2^16): wave 0, lane = column pair + 32 x (upper / lower 24 rows); the
        // lower half adds the upper half's total (lane - 32: ds_bpermute)
        if (tid < 64) {
            const int cp = tid & 31, seg = tid >> 5;
            uint32_t v[24];
            u16x2 run = { 0, 0 };
            if (cp < JD) {
#pragma unroll
                for (int i = 0; i < 24; i++) v[i] = Jd[(24 * seg + i + 1) * JD + cp];
#pragma unroll
                for (int i = 0; i < 24; i++) { run = run + __builtin_bit_cast(u16x2, v[i]); v[i] = __builtin_bit_cast(uint32_t, run); }
            }
            const uint32_t upper = (uint32_t)__builtin_amdgcn_ds_bpermute(((tid - 32) & 63) * 4, (int)__builtin_bit_cast(uint32_t, run));
            if (cp < JD) {
                const u16x2 off = seg ? __builtin_bit_cast(u16x2, upper) : (u16x2){ 0, 0 };
#pragma unroll
                for (int i = 0; i < 24; i++)
                    Jd[(24 * seg + i + 1) * JD + cp] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, v[i]) + off));
            }
        }
    }
    BAD_DET_PHASE(4);                                           // + column prefix
    __syncthreads();

    const int fw = cols + 1, fh = rows + 1;       // integral image dims of the full frame
    const bool wide_store = ((((uintptr_t)desc) | desc_pitch) & 7u) == 0;

    for (int b0 = 0; b0 < nbits; b0 += 256) {
        const int b = b0 + tid;
        bool bit = false;
        if (fits) {
            if (border) {
                bit = bad_border_bit<uint16_t>(A, P->box[b], J, JP, S, wx0, wy0, fw, fh);
            } else {
                // integer fast path, bad.cpp:365-393.  Not within 27 px of the frame edge: the window is not clamped by
                // the frame and R >= s * reach + 1 puts every tap inside it (no clamps, spec S9 is vacuous here).
                bit = bad_taps_bit(b0 == 0 ? t0 : t1, b0 == 0 ? q0.w : q1.w, smem + HB_OFF);
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
// different phases -- window loads (vector memory), prefix sums (VALU + LDS stores), box gathers (LDS reads) overlap.
// The integral is kept MODULO 2^16: a box of the table has at most 16 x 16 pixels (BadParamsDev::ubox_max_side, checked by
// the launcher), so every box sum is below 2^16 and (tl + br - tr - bl) mod 2^16 IS the sum.  Half the LDS (4.9 KB per
// wave: 32 waves per CU instead of 16), and the column prefix runs on packed pairs.
//   window   48 rows x 13 aligned dwords through a buffer resource (range-checked: zero beyond the image), row-coalesced
//            (16 lanes per row), into LDS rows of 13 dwords (odd pitch: conflict-free row-per-lane reads)
//   rows     lane r: its row as 13 ds_read_b32 + v_alignbyte, exclusive prefix on SDWA byte adds, written as 25 packed u16
//            pairs into plane row r + 1: P'[r][x] = sum of the row's pixels left of column x (x = 0 .. 49)
//   columns  lane j < 25: columns 2j, 2j + 1, IN PLACE: row r + 1 <- row r + row r + 1 (one ds_read_b32, one v_pk_add_u16,
//            one ds_write_b32 per row); row 0 is zero.  The staged window aliases the plane (it is in registers by then)
//   boxes    the per-pair table of the detector path (BadParamsDev::ubox, loaded before anything that depends on the
//            keypoint), 64 pairs per step, ds_read_u16 gathers; the wave's ballot is the descriptor word, lane i keeps
//            word i and the descriptor leaves as one 8-byte store per lane
// Measured (MI355X, C3: 40 000 keypoints of a 4K frame, profiles/r03_c3_*): the kernel is bound by the LDS pipe -- the
// random box gathers cost 3.4 cycles per 32 lanes in bank conflicts -- with the VALU 60 % busy beside it.
// ================================================================================================
#ifndef BAD_RAW_CB
#define BAD_RAW_CB 8                 // rows the column prefix reads ahead of its adds (divides 24)
#endif
#define BAD_RAW_JP BAD_J_PITCH                             // u16 entries per integral row (25 dwords)
#define BAD_RAW_WAVE_LDS 4912                             // 49 * 50 * 2 = 4900, rounded to 16

template <int NIT>            // descriptor words of 64 bits: 4 (BAD256) or 8 (BAD512)
__global__ __launch_bounds__(256) void bad_raw_kernel(
    const int* __restrict__ d_count, int n, const BadParamsDev* __restrict__ P, const Affine* __restrict__ aff,
    uint8_t* __restrict__ desc, size_t desc_pitch, int batched, size_t aff_stride, const FrameOut counts, const FrameDesc descs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 48, JP = BAD_RAW_JP, JD = JP / 2;
    if (batched) {       // frame blockIdx.y of a batched detectAndCompute: its count, records and descriptor matrix
        d_count = counts.count[blockIdx.y]; desc = descs.desc[blockIdx.y]; aff += blockIdx.y * aff_stride;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = d_count ? min(*d_count, n) : n;
    const int ngroups = (count + 3) >> 2;
    // (Round 6, measured and dropped: a grid capped at ~32 K workgroups per launch whose workgroups take the groups g, g + gridDim.x,
    // ... -- no tens of thousands of empty workgroups when a frame yields a fraction of the capacity -- cost the 8K frame rate 9 %:
    // the loop around this body changes its schedule.)
    if ((int)blockIdx.x >= ngroups) return;
    const int kid = xcd_chunked(blockIdx.x, ngroups) * 4 + wave;      // neighbouring keypoints share an XCD's L2
    if (kid >= count) return;                                         // wave-uniform

    unsigned char* wbuf = smem + wave * BAD_RAW_WAVE_LDS;
    uint32_t* rawdw = reinterpret_cast<uint32_t*>(wbuf);               // 48 x 13 dwords, dead before the plane is written
    uint32_t* Jd = reinterpret_cast<uint32_t*>(wbuf);                  // 49 x 25 dwords: the integral, two u16 columns per dword
    const uint16_t* J = reinterpret_cast<const uint16_t*>(wbuf);

    // the per-pair table does not depend on the keypoint: requested first, its latency hides behind the window loads
    uint4 q[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) q[it] = P->ubox[it * 64 + lane];
    const Affine A = aff[kid];                                         // wave-uniform address: scalar loads
    const uint8_t* img = A.img; const int pitch = A.pitch, rows = A.rows, cols = A.cols;
    const bool fits = A.S != 0;
    const int wx0 = A.wx0, wy0 = A.wy0;

    if (fits) {
        const bool aligned = ((((uintptr_t)img) | (uintptr_t)pitch) & 3u) == 0;
        const int off = aligned ? (wx0 & 3) : 0;
        if (aligned) {
            // rows are pitch bytes apart and at least roundup4(cols) of them are memory we may read (our own levels are
            // padded; a caller's 4-byte aligned image has pitch >= roundup4(cols)); beyond the image the range check gives 0
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, (rows - 1) * pitch + ((cols + 3) & ~3), 0x00020000);
            // 16 lanes per row (13 of them load), 4 rows per step: no division, one offset add per step
            const int k = lane & 15, r4 = lane >> 4;
            int goff = wy0 * pitch + (wx0 & ~3) + r4 * pitch + 4 * k;
            uint32_t* dst = rawdw + r4 * 13 + k;
#pragma unroll
            for (int it = 0; it < 12; it++) {
                if (k < 13) dst[it * 52] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff, 0, 0);
                goff += 4 * pitch;
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
        uint32_t out[JD];
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
            for (int j = 0; j < JD; j++) {
                const int lo = run;
                if (2 * j < S) run += (int)((w[(2 * j) >> 2] >> (8 * ((2 * j) & 3))) & 0xffu);
                const int hi = run;
                if (2 * j + 1 < S) run += (int)((w[(2 * j + 1) >> 2] >> (8 * ((2 * j + 1) & 3))) & 0xffu);
                out[j] = (uint32_t)lo | ((uint32_t)hi << 16);
            }
        }
        // every lane has its raw row in registers (program order; a wave's LDS operations complete in order) before the
        // plane, which aliases the staged window, is written
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < S) {
#pragma unroll
            for (int j = 0; j < JD; j++) Jd[(lane + 1) * JD + j] = out[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- column prefix in place (lane = column pair), modulo 2^16 per column.  Round 5: in TWO halves side by side -- lanes 0 .. 24
        //      run down plane rows 1 .. 24, lanes 32 .. 56 down rows 25 .. 48 (24 dependent steps instead of 48: the phase was 15 of
        //      the kernel's 62 us, tools/experiments/README.md) --, then rows 25 .. 48 receive row 24's totals, two rows per pass on
        //      50 lanes, no chain.  Integer sums modulo 2^16: the order does not matter. ----
        {
            const int half = lane >> 5, col = lane & 31;
            if (col < JD) {
                if (half == 0) Jd[col] = 0u;
                const int rb = 1 + (S / 2) * half;
                u16x2 run = { 0, 0 };
#pragma unroll
                for (int r0 = 0; r0 < S / 2; r0 += BAD_RAW_CB) {
                    uint32_t v[BAD_RAW_CB];
#pragma unroll
                    for (int i = 0; i < BAD_RAW_CB; i++) v[i] = Jd[(rb + r0 + i) * JD + col];
#pragma unroll
                    for (int i = 0; i < BAD_RAW_CB; i++) {
                        run = run + __builtin_bit_cast(u16x2, v[i]);
                        Jd[(rb + r0 + i) * JD + col] = __builtin_bit_cast(uint32_t, run);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int g = lane >= JD ? 1 : 0, c = lane - JD * g;
            if (lane < 2 * JD) {
                const u16x2 tot = __builtin_bit_cast(u16x2, Jd[(S / 2) * JD + c]);
                uint32_t v[S / 4];
#pragma unroll
                for (int j = 0; j < S / 4; j++) v[j] = Jd[(S / 2 + 1 + g + 2 * j) * JD + c];
#pragma unroll
                for (int j = 0; j < S / 4; j++) Jd[(S / 2 + 1 + g + 2 * j) * JD + c] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, v[j]) + tot));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    const bool border = (A.border & 1) != 0;
    const int fw = cols + 1, fh = rows + 1;
    const int wbase = -(wy0 * JP + wx0) * 2;
    int vbase = (int)(uint32_t)(uintptr_t)wbuf + wbase;     // LDS byte address of the plane (low half of the generic address) + wbase
    asm volatile("" : "+v"(vbase));
    uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int b = it * 64 + lane;
        bool bit = false;
        if (fits) {
            if (border) {
                bit = bad_border_bit<uint16_t>(A, P->box[b], J, JP, S, wx0, wy0, fw, fh);
            } else {
                // integer fast path, bad.cpp:365-393; the window holds every tap (bad_det_kernel has the argument)
                bit = bad_ubox_bit_v(A, q[it], vbase);
            }
        }
        const unsigned long long m = __ballot(bit);
        if (lane == it) { mlo = (uint32_t)m; mhi = (uint32_t)(m >> 32); }
    }
    // bit i -> byte i / 8, MSB first (bad.cpp:349,368): lane i holds bits 64 i .. 64 i + 63
    if (lane < NIT) {
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

// ================================================================================================
// Whole-level Gaussian for detectAndCompute (round 4; spec S6, cuda_efficient_features.cpp:193,302-306: the reference blurs
// every level before it describes on it).  Until round 3 every keypoint blurred its own 54 x 54 window inside bad_det_kernel
// (no blurred copy of the pyramid): 18.7 lane-instructions per blurred pixel -- aprons, idle lanes of the item shapes, one
// workgroup per keypoint -- against the 17.5 of a streaming blur whose instructions are the CHEAP ones: profiles/
// r04_valu_rate.txt shows v_fma_f32 / v_mul_f32 at 2.2 cycles per wave64 instruction and v_pk_fma_f32 / conversions at 4.2.
// So the levels are blurred once, as images, and a wave describes a keypoint on the blurred level (bad_raw_kernel) instead
// of a workgroup (bad_det_kernel: 1550 wave-instructions per keypoint, bad_raw_kernel: 630).
//
// One WAVE owns a strip of 256 columns (a lane: 4 adjacent pixels = one dword) and BLV_ROWS output rows, and walks down:
//   load    the lane's dword of the next input row and its two neighbours (px -4 .. 7; the same 128-byte lines as the
//           neighbouring lanes': L1 hits), through a buffer resource (range check, no predicates)
//   rows    10 x v_cvt_f32_ubyteN straight from the three dwords, 4 outputs x 7 taps: acc = fma(v_j, tap_j, acc), j = 0 .. 6,
//           from tap_0 * v_0 (spec S6: one rounding per tap), plain v_fma_f32 (full rate; the packed form is no faster per
//           FMA and would need the pixels in register pairs)
//   ring    the row-pass results of the last 7 rows stay in registers (28 floats, the loop is unrolled by 7: static names)
//   columns output row y from rows y - 3 .. y + 3 of the ring, same FMA order, v_cvt_pk_u8_f32 (round half even, saturate),
//           one dword store
// No LDS, no barrier; BORDER_REFLECT_101: the row index is reflected (scalar), strips that touch the left / right edge
// of the level -- and levels whose base or pitch is not 4-byte aligned (a caller's level 0) -- fetch their 10 input pixels
// as bytes at reflected columns (wave-uniform branch).  Bit-exact with efx_blur_window_lds (same expression per pixel).
// ================================================================================================
#ifndef BLV_GROUPS
#define BLV_GROUPS 9                     // a task of the dword path: 9 groups of 7 output rows (the ring's period)
#endif
#ifndef BLV_WAVES
#define BLV_WAVES 7                      // waves per SIMD the kernel is compiled for (72 VGPRs)
#endif
#ifndef BLV_TAPS
#define BLV_TAPS 7                       // INVESTIGATION builds: fewer taps in the row pass = fewer VALU instructions, same memory traffic
#endif
#ifndef BLV_PF
#define BLV_PF 1                         // rows the loads run ahead of the arithmetic (2: same speed, 7 more VGPRs -> spills at 72)
#endif
#define BLV_ROWS (7 * BLV_GROUPS)
#define BLV_ROWS_BYTES 16                // ... of the byte path (narrow or unaligned levels): short tasks, its loads are not prefetched
// Strips of a level on the dword path (cols >= 512): `n_reg` regular strips at x = 256 s, then ONE strip anchored at the
// level's right edge, x = xs + 4 lane with xs = ((cols - 1) & ~3) - 252: its last lane's dword holds the level's last pixel, so
// the BORDER_REFLECT_101 pixels of the right edge are a fixed pattern of register moves in lanes 62 / 63 (by r = number of
// valid pixels in that dword), as the left edge's are in lane 0 of strip 0.  Regular strips do not store columns >= xs
// (the anchored strip owns them), so none of the columns they do store needs a pixel beyond the level.
struct BlurLevel { const uint8_t* src; uint8_t* dst; int spitch, dpitch, rows, cols, bytes, n_reg, nstrips, xs, task_end, src_img0; };
struct BlurLevelsArgs { int nlevels, total, nr; BlurLevel lv[EFX_MAX_LEVELS]; };     // nr: output rows of a dword-path task (7 .. BLV_ROWS)

// the generic form: 10 input pixels per lane as byte loads at reflected columns (any alignment, any size)
__device__ __forceinline__ void blur_task_bytes(const BlurLevel& L, const uint8_t* Lsrc, uint8_t* Ldst, int strip, int chunk, int lane, const float (&tp)[7])
{
    constexpr int NR = BLV_ROWS_BYTES;
    const int rows = L.rows, cols = L.cols, spitch = L.spitch, dpitch = L.dpitch;
    const int x = strip * 256 + lane * 4;
    const int y0 = chunk * NR;
    const int nout = min(NR, rows - y0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Lsrc), 0, (rows - 1) * spitch + cols, 0x00020000);
    int cidx[10];
#pragma unroll
    for (int k = 0; k < 10; k++) cidx[k] = efx_reflect101(x - 3 + k, cols);
    auto rowpass = [&](int yy, float (&o)[4]) {
        const int rbase = efx_reflect101(yy, rows) * spitch;
        float v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = (float)__builtin_amdgcn_raw_buffer_load_b8(rsrc, rbase + cidx[k], 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float acc = v[i] * tp[0];
#pragma unroll
            for (int jt = 1; jt < 7; jt++) acc = __builtin_fmaf(v[i + jt], tp[jt], acc);
            o[i] = acc;
        }
    };
    float R[7][4];
#pragma unroll
    for (int i = 0; i < 6; i++) rowpass(y0 - 3 + i, R[i]);
    uint8_t* drow = Ldst + (size_t)y0 * dpitch + x;
    for (int base = 0; base < nout; base += 7) {
#pragma unroll
        for (int u = 0; u < 7; u++) {
            if (base + u < nout) {                          // wave-uniform
                rowpass(y0 + base + u + 3, R[(u + 6) % 7]);
                uint32_t pk = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float acc = R[u % 7][i] * tp[0];
#pragma unroll
                    for (int jt = 1; jt < 7; jt++) acc = __builtin_fmaf(R[(u + jt) % 7][i], tp[jt], acc);
                    pk = __builtin_amdgcn_cvt_pk_u8_f32(acc, i, pk);
                }
                // destination rows are padded to 256 bytes: the dword of a row's last pixels is inside the row
                if (x < cols) *reinterpret_cast<uint32_t*>(drow + (size_t)(base + u) * dpitch) = pk;
            }
        }
    }
}

// the dword path: straight-line rows (no branch but the loop's), loads two rows ahead, stores through a range-checked resource
__device__ __forceinline__ void blur_task_dwords(const BlurLevel& L, const uint8_t* Lsrc, uint8_t* Ldst, int strip, int chunk, int lane, const float (&tp)[7], int nr)
{
    const int rows = L.rows, cols = L.cols, spitch = L.spitch, dpitch = L.dpitch;
    const bool anchored = strip == L.n_reg;                // wave-uniform
    const bool leftmost = strip == 0;
    const int x = (anchored ? L.xs : strip * 256) + lane * 4;
    const int y0 = chunk * nr;
    const int nout = min(nr, rows - y0);
    // a row's dwords up to roundup4(cols) are memory we may read (own levels: padded rows; a caller's aligned level 0: its pitch
    // is a multiple of 4); beyond that -- and left of the level, where the offset wraps -- the range check returns 0
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Lsrc), 0, (rows - 1) * spitch + ((cols + 3) & ~3), 0x00020000);
    // rows of this task only, and (regular strips) columns left of the anchored strip only: everything else is an offset the
    // range check drops.  Destination rows are padded to 256 bytes: the dword of a row's last pixels is inside the row
    const __amdgpu_buffer_rsrc_t dsrc = __builtin_amdgcn_make_buffer_rsrc(Ldst, 0, (y0 + nout) * dpitch, 0x00020000);
    const int dx = (anchored || x < L.xs) ? x : 0x40000000;
    const int r = cols - ((cols - 1) & ~3);                // valid pixels in the level's last dword: 1 .. 4
    struct Raw { uint32_t a, b, c; };
    auto fetch = [&](int yy) {
        int ry = yy < 0 ? -yy : (yy > rows - 1 ? 2 * (rows - 1) - yy : yy);      // REFLECT_101, rows >= 16; wave-uniform
        ry = max(min(ry, rows - 1), 0);                     // rows fetched ahead of a level's last ones: any valid row
        const int goff = ry * spitch + x - 4;
        Raw q;
#ifdef BLV_NOLOAD                                            // INVESTIGATION builds (tools/microbench/blur_sweep.sh): results invalid
        q.a = (uint32_t)goff; q.b = (uint32_t)goff * 3u; q.c = (uint32_t)goff * 5u;
#else
        q.a = __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff, 0, 0);
        q.b = __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff + 4, 0, 0);
        q.c = __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff + 8, 0, 0);
#endif
        return q;
    };
    auto rowpass = [&](const Raw& q, float (&o)[4]) {
        float v[10];
        v[0] = (float)((q.a >> 8) & 0xffu); v[1] = (float)((q.a >> 16) & 0xffu); v[2] = (float)(q.a >> 24);
        v[3] = (float)(q.b & 0xffu); v[4] = (float)((q.b >> 8) & 0xffu); v[5] = (float)((q.b >> 16) & 0xffu); v[6] = (float)(q.b >> 24);
        v[7] = (float)(q.c & 0xffu); v[8] = (float)((q.c >> 8) & 0xffu); v[9] = (float)((q.c >> 16) & 0xffu);
        if (__builtin_expect(leftmost, 0)) {                // px -3, -2, -1 = px 3, 2, 1 (wave-uniform branch)
            const bool l0 = lane == 0;
            v[0] = l0 ? v[6] : v[0]; v[1] = l0 ? v[5] : v[1]; v[2] = l0 ? v[4] : v[2];
        }
        if (__builtin_expect(anchored, 0)) {
            // lane 63 holds the level's last pixel at input index kmax = r + 2, lane 62 at r + 6: input k in (kmax, kmax + 3] is
            // the mirror image 2 kmax - k (outputs beyond the level are not stored and need nothing)
            const bool l63 = lane == 63, l62 = lane == 62;
            if (r == 1) { v[4] = l63 ? v[2] : v[4]; v[5] = l63 ? v[1] : v[5]; v[6] = l63 ? v[0] : v[6]; v[8] = l62 ? v[6] : v[8]; v[9] = l62 ? v[5] : v[9]; }
            else if (r == 2) { v[5] = l63 ? v[3] : v[5]; v[6] = l63 ? v[2] : v[6]; v[7] = l63 ? v[1] : v[7]; v[9] = l62 ? v[7] : v[9]; }
            else if (r == 3) { v[6] = l63 ? v[4] : v[6]; v[7] = l63 ? v[3] : v[7]; v[8] = l63 ? v[2] : v[8]; }
            else { v[7] = l63 ? v[5] : v[7]; v[8] = l63 ? v[4] : v[8]; v[9] = l63 ? v[3] : v[9]; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float acc = v[i] * tp[0];                      // == fma(tp[0], v, 0) exactly
#pragma unroll
            for (int jt = 1; jt < (BLV_TAPS); jt++) acc = __builtin_fmaf(v[i + jt], tp[jt], acc);
            o[i] = acc;
        }
    };
    float R[7][4];
    Raw pf[BLV_PF];
#pragma unroll
    for (int k = 0; k < BLV_PF; k++) pf[k] = fetch(y0 - 3 + k);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const Raw cur = pf[0];
#pragma unroll
        for (int k = 0; k + 1 < BLV_PF; k++) pf[k] = pf[k + 1];
        pf[BLV_PF - 1] = fetch(y0 - 3 + i + BLV_PF);
        rowpass(cur, R[i]);
    }
    int doff = y0 * dpitch + dx;
    for (int base = 0; base < nout; base += 7) {
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const Raw cur = pf[0];
#pragma unroll
            for (int k = 0; k + 1 < BLV_PF; k++) pf[k] = pf[k + 1];
            pf[BLV_PF - 1] = fetch(y0 + base + u + 3 + BLV_PF);
            rowpass(cur, R[(u + 6) % 7]);
            __builtin_amdgcn_sched_barrier(0);              // row pass, then column pass: interleaved by the scheduler they need 72 VGPRs
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float acc = R[u % 7][i] * tp[0];
#pragma unroll
                for (int jt = 1; jt < 7; jt++) acc = __builtin_fmaf(R[(u + jt) % 7][i], tp[jt], acc);
                pk = __builtin_amdgcn_cvt_pk_u8_f32(acc, i, pk);
            }
#ifdef BLV_NOSTORE
            if (pk == 0x12345679u) __builtin_amdgcn_raw_buffer_store_b32(pk, dsrc, doff, 0, 0);
#else
            __builtin_amdgcn_raw_buffer_store_b32(pk, dsrc, doff, 0, 0);
#endif
            doff += dpitch;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Two kernels, one per path: in ONE kernel the register allocation is the larger path's plus a few, and the dword path then
// spills at seven waves per SIMD -- a kernel with scratch costs ~25 us of dispatch stall per launch on this runtime (round 4: the
// HIP-event pair around the launch read 88 us where the kernel itself ran 57)
template <bool BYTES>
__global__ __launch_bounds__(256, BYTES ? 7 : BLV_WAVES) void blur_levels_kernel(const BlurLevelsArgs A, float tp0, float tp1, float tp2, float tp3,
                                                                                    const FrameIn in, size_t pyr_stride, size_t blur_stride)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = xcd_chunked(blockIdx.x, gridDim.x) * 4 + wave;
    if (task >= A.total) return;
    int l = 0;
    while (l + 1 < A.nlevels && task >= A.lv[l].task_end) l++;
    const BlurLevel& L = A.lv[l];
    const int t = task - (l ? A.lv[l - 1].task_end : 0);
    const int chunk = t / L.nstrips, strip = t - chunk * L.nstrips;
    // the taps live in VECTOR registers: a full-rate VALU instruction (v_fmac_f32: 2 cycles per wave64) with a scalar-register
    // source runs at HALF rate (4.1 cycles; profiles/r04_valu_rate.txt "(sgpr)" rows) -- left to the compiler the 64 FMAs of a
    // row read the taps from SGPRs and the kernel took 68 instead of 57 us
    asm volatile("" : "+v"(tp0), "+v"(tp1), "+v"(tp2), "+v"(tp3));
    const float tp[7] = { tp0, tp1, tp2, tp3, tp2, tp1, tp0 };
    // this frame's (blockIdx.y) source level and blurred copy
    const uint8_t* Lsrc = L.src_img0 ? in.img0[blockIdx.y] : L.src + blockIdx.y * pyr_stride;
    uint8_t* Ldst = L.dst + blockIdx.y * blur_stride;
    if (BYTES) blur_task_bytes(L, Lsrc, Ldst, strip, chunk, lane, tp);
    else blur_task_dwords(L, Lsrc, Ldst, strip, chunk, lane, tp, A.nr);
}

hipError_t efx_launch_blur_levels(const LevelTable& H, const uint8_t* img0, int pitch0, const uint8_t* pyramid, uint8_t* blurred,
                                  int blur0_pitch, size_t blur_levels_off, const ProfRec& prof_rec, hipStream_t stream,
                                  int nframes, const FrameIn& in_frames, const FrameStride& fs)
{
    const int NF = nframes > 0 ? nframes : 1;
    FrameIn in = in_frames;
    if (NF == 1) in.img0[0] = img0;
    uintptr_t img_bits = 0;              // level 0's alignment must hold for every frame's image
    for (int f = 0; f < NF; f++) img_bits |= reinterpret_cast<uintptr_t>(in.img0[f]);
    // Rows per task of the dword path: 63 on large pyramids (a task recomputes 6 rows of the row pass: 10 %), fewer on small ones
    // -- an FHD pyramid has 400 tasks of 63 rows for 1024 SIMDs, and a lone wave issues one VALU instruction per 4 cycles: the
    // kernel then took 25 us of a 110 us call.  The smallest multiple of 7 that still leaves ~3000 tasks, but not below 14
    BlurLevelsArgs A[2] = {};            // [0] the dword path, [1] the byte path (narrow or unaligned levels)
    for (int nr = BLV_ROWS; nr >= 14; nr -= 7) {
        A[0] = BlurLevelsArgs{}; A[1] = BlurLevelsArgs{};
        A[0].nr = A[1].nr = nr;
        for (int l = 0; l < H.nlevels; l++) {
            const LevelDev& L = H.lv[l];
            if (!L.active || L.rows <= 0 || L.cols <= 0) continue;
            const uint8_t* src = l == 0 ? img0 : pyramid + L.img_off;
            const int spitch = l == 0 ? pitch0 : L.pitch;
            const bool aligned = (((l == 0 ? img_bits : (uintptr_t)src) | (uintptr_t)spitch) & 3u) == 0;
            const bool bytes = !(aligned && L.cols >= 512 && L.rows >= 16);
            BlurLevelsArgs& T = A[bytes ? 1 : 0];
            BlurLevel& B = T.lv[T.nlevels++];
            B.src = src; B.spitch = spitch; B.src_img0 = l == 0 ? 1 : 0;
            B.dst = l == 0 ? blurred : blurred + blur_levels_off + L.img_off;
            B.dpitch = l == 0 ? blur0_pitch : L.pitch;
            B.rows = L.rows; B.cols = L.cols;
            B.bytes = bytes ? 1 : 0;
            if (bytes) {
                B.nstrips = (L.cols + 255) / 256;
                T.total += B.nstrips * ((L.rows + BLV_ROWS_BYTES - 1) / BLV_ROWS_BYTES);
            } else {
                B.xs = ((L.cols - 1) & ~3) - 252;
                B.n_reg = (B.xs + 255) / 256;                   // regular strips cover [0, xs)
                B.nstrips = B.n_reg + 1;
                T.total += B.nstrips * ((L.rows + nr - 1) / nr);
            }
            B.task_end = T.total;
        }
        if ((A[0].total + A[1].total) * NF >= 3072) break;
    }
    if (A[0].total + A[1].total == 0) return hipSuccess;
    float t[7];
    efx_gaussian_taps_host(t);
    const bool prof = prof_rec.begin(11, stream);
    for (int k = 0; k < 2; k++) {
        if (A[k].total == 0) continue;
        const int nblk = ((A[k].total + 3) / 4 + EFX_NXCD - 1) / EFX_NXCD * EFX_NXCD;     // xcd_chunked wants whole rounds
        if (k == 0) hipLaunchKernelGGL(blur_levels_kernel<false>, dim3(nblk, NF), dim3(256), 0, stream, A[k], t[0], t[1], t[2], t[3], in, fs.pyramid, fs.blurred);
        else hipLaunchKernelGGL(blur_levels_kernel<true>, dim3(nblk, NF), dim3(256), 0, stream, A[k], t[0], t[1], t[2], t[3], in, fs.pyramid, fs.blurred);
    }
    prof_rec.end(prof, 11, stream);
    return hipGetLastError();
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
    if (a.blur && a.level_blurred) {
        // the records point at BLURRED level images (efx_launch_blur_levels ran on this stream): a wave per keypoint, no blur
        // of its own.  detect_common sets level_blurred only under bad_raw_kernel's conditions (S == 48, box edges <= 16)
        const int batched = a.nframes > 1 ? 1 : 0;
        const int NF = batched ? a.nframes : 1;
        const int raw_grid = (a.n + 3) / 4;
        if (a.nbits == 256)
            hipLaunchKernelGGL(bad_raw_kernel<4>, dim3(raw_grid, NF), dim3(256), 4 * BAD_RAW_WAVE_LDS, stream, a.d_count, a.n, d_params, aff,
                               a.desc, a.desc_pitch, batched, a.aff_stride, a.counts, a.descs);
        else
            hipLaunchKernelGGL(bad_raw_kernel<8>, dim3(raw_grid, NF), dim3(256), 4 * BAD_RAW_WAVE_LDS, stream, a.d_count, a.n, d_params, aff,
                               a.desc, a.desc_pitch, batched, a.aff_stride, a.counts, a.descs);
        return hipGetLastError();
    }
    if (a.blur) {
        if (S == 48 && a.uniform_size && max_size == (float)EFX_PATCH_SIZE && a.kp_level && a.bad_det_tables == 2) {
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
        if (S == 48 && a.uniform_size && max_size == (float)EFX_PATCH_SIZE && a.bad_det_tables == 2 && !a.bad_no_raw) {
            // computeAsync on detector-sized keypoints: a wave per keypoint, four keypoints per workgroup
            if (a.nbits == 256)
                hipLaunchKernelGGL(bad_raw_kernel<4>, dim3((a.n + 3) / 4), dim3(256), 4 * BAD_RAW_WAVE_LDS, stream, a.d_count, a.n, d_params, aff,
                                   a.desc, a.desc_pitch, 0, 0, a.counts, a.descs);
            else
                hipLaunchKernelGGL(bad_raw_kernel<8>, dim3((a.n + 3) / 4), dim3(256), 4 * BAD_RAW_WAVE_LDS, stream, a.d_count, a.n, d_params, aff,
                                   a.desc, a.desc_pitch, 0, 0, a.counts, a.descs);
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
