// hashsift_kernels.hip -- HashSIFT descriptor for gfx950: PatchSIFT 129-vectors + bf16 matrix-core projection.
//
// Arithmetic: the reference CPU descriptor, modules/efficient_features/src/hash_sift.cpp
//   rectifyPatch / warpAffineLinear :68-138, HistBin :162-184, distribute :193-198,
//   computePatchSIFT :200-331, normalize :150-160, matmulAndSign :353-378.
//
// MI355X design
//   * a lane per keypoint writes its record (rectifying affine map, level image, window: hs_record_kernel), then one
//     workgroup per keypoint; the (optionally Gaussian-blurred, spec S6) window the rotated 32x32 patch can touch is
//     staged in LDS, so detectAndCompute needs no global blur pass;
//   * the 6x6x10 gradient histogram is accumulated in 15.17 FIXED POINT with LDS integer atomics, one lane per pixel; the
//     two orientation bins a pixel votes for in a cell are two 32-bit counters packed into ONE 64-bit atomic (a bin
//     collects < 2^15, so 17 fractional bits never carry into the neighbour); a vote is floor(v * 2^17 + 1/2).  Integer
//     addition is associative, so the result does not depend on the order the lanes arrive in (the reference's CUDA
//     kernel adds floats with shared-memory atomics, cuda_hash_sift.cu:282-289, and is order-nondeterministic).
//     Against the CPU loop's sequentially rounded float sums (hash_sift.cpp:233-290) 1.4e-5 of the 129-vector elements
//     differ by one unit and 2e-5 of the descriptor bytes (40 000 keypoints; the reference's own GPU-vs-CPU bound is 1e-4
//     of the bytes, tests/descriptor_test.cpp:72; the 32.32 counters of round 1 gave 1.5e-6 at twice the atomics:
//     DESIGN.md section 3 has the trade);
//   * the Gaussian pixel weights expf(...) (30x30) and orientation bin + magnitude of a gradient (511x511 integer
//     gradients: scaleO*atan2f(dy,dx), sqrtf) are tables computed on the host with the same libm calls the CPU code
//     makes; cosf/sinf of the keypoint angle are glibc's (glibc_sincosf.h, bit-identical to the host libm);
//   * projection T = R[N x 129] . W^T is the one real GEMM of the path: v_mfma_f32_32x32x16_bf16 with W split on the host
//     into three bf16 terms whose sum is W exactly and R (integers 0..255, and 1) exact in bf16, so every product is
//     exact in the fp32 accumulator; fused with the sign test and MSB-first bit packing (no T matrix in HBM, no separate
//     binarize pass as in cuda_hash_sift.cu:414-435).

#include "efx_device.h"
#include "blur_window.h"
#include "glibc_sincosf.h"
#include <stdlib.h>

#define HS_KPAD 132   // 129 padded to a multiple of 4 floats (row pitch of the fp32 weights)
#ifndef HS_PP
#define HS_PP 36      // row pitch of the rectified patch in LDS (bytes)
#endif
#define HS_KB 144     // K of the bf16 projection: 129 padded to 9 MFMA steps of 16

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int efx_u32x2 __attribute__((ext_vector_type(2)));

// Per-keypoint record of the PatchSIFT kernel: the rectifying affine map (rectifyPatch, hash_sift.cpp:111-132), the
// keypoint's pyramid level and the window of it the 32 x 32 patch can touch.  One LANE per keypoint computes it
// (hs_record_kernel): the cosf / sinf restatement is ~600 instructions of double arithmetic, which inside the descriptor
// kernel ran on one lane of a workgroup while 255 waited (24 M wave-instructions per 40 000 keypoints).
struct HsRec {
    float m00, m01, m02, m10, m11, m12;
    int wx0, wy0;
    const uint8_t* img;
    int pitch, rows, cols;
    int S;                       // window edge; 0: the window does not fit the LDS plan (blurred: descriptor of zeros, as before; raw: gathers from memory)
    int pad[2];
};
static_assert(sizeof(HsRec) == 64, "one record per 64-byte line");

// Frames of a batched describe (blockIdx.y; nframes <= 1: the scalar arguments as they are): frame f's keypoints, records and
// 129-vectors lie f * kp_stride entries into their arrays, its pyramid f * pyr_stride bytes into the pyramid buffer
struct HsBatch { int nframes; size_t kp_stride, pyr_stride; FrameIn imgs; FrameOut counts; FrameDesc descs; };

__global__ __launch_bounds__(256) void hs_record_kernel(
    const uint8_t* __restrict__ img0, int pitch0, int rows0, int cols0,
    const uint8_t* __restrict__ pyramid, const LevelTable* __restrict__ T,
    const float4* __restrict__ kp4, const uint8_t* __restrict__ kps5, size_t kps5_pitch, const int* __restrict__ kp_level, const int* __restrict__ d_count, int n,
    float crop_scale, int smax, int sfixed, int dword_rows, HsRec* __restrict__ rec, const HsBatch hb)
{
    if (hb.nframes > 1) {
        const size_t f = blockIdx.y;
        img0 = hb.imgs.img0[f]; pyramid += f * hb.pyr_stride; kp4 += f * hb.kp_stride; kp_level += f * hb.kp_stride;
        d_count = hb.counts.count[f]; rec += f * hb.kp_stride;
    }
    const int count = d_count ? min(*d_count, n) : n;
    const int kid = blockIdx.x * 256 + threadIdx.x;
    if (kid >= count) return;
    const float4 kp = efx_load_keypoint(kp4, kps5, kps5_pitch, kid);
    HsRec A;
    A.img = img0; A.pitch = pitch0; A.rows = rows0; A.cols = cols0;
    if (kp_level) {
        const int l = kp_level[kid];
        if (l > 0) { const LevelDev& L = T->lv[l]; A.img = pyramid + L.img_off; A.pitch = L.pitch; A.rows = L.rows; A.cols = L.cols; }
        else { A.rows = T->lv[0].rows; A.cols = T->lv[0].cols; }
    }
    const float px = kp.x, py = kp.y, size = kp.z, angle = kp.w;
    // rectifyPatch, hash_sift.cpp:111-132
    const float PI_1 = (float)3.1415926535897932384626433832795;
    const float s = crop_scale * size / (0.5f * (float)(32 + 32));
    const float theta = PI_1 * angle / 180;
    // cosf / sinf exactly as the CPU code gets them from libm (glibc's sincosf restated in glibc_sincosf.h and checked
    // against the host libm for every float up to 11); larger angles: the rounded double result
    float c1 = 1.f, s1 = 0.f;
    if (angle >= 0) {
        if (theta < EFX_GLIBC_SINCOSF_MAX) { c1 = efx_glibc_sincosf(theta, 1); s1 = efx_glibc_sincosf(theta, 0); }
        else { c1 = (float)cos((double)theta); s1 = (float)sin((double)theta); }
    }
    const float cost = s * c1;
    const float sint = s * s1;
    A.m00 = +cost; A.m01 = -sint; A.m02 = (-cost + sint) * (float)32 / 2.f + px;
    A.m10 = +sint; A.m11 = +cost; A.m12 = (-sint - cost) * (float)32 / 2.f + py;
    // window the patch can touch
    const float sg = fabsf(crop_scale * size / 32.f);
    const int R = (int)floorf(sg * 22.63f + 2.01f);     // >= 22.63 sg + 1: floor(u) and floor(u) + 1 of every patch pixel are inside
    int S = sfixed ? sfixed : 2 * R + 2;
    bool fits = sfixed ? (2 * R + 2 == sfixed) : (S <= smax && S > 0);
    if (dword_rows && ((((uintptr_t)A.img) | (uintptr_t)A.pitch) & 3u)) fits = false;      // the raw window is staged as aligned dwords
    if (!fits && !sfixed) S = smax;
    const int ix = (int)floorf(px), iy = (int)floorf(py);
    A.wx0 = min(max(ix - R, 0), max(A.cols - S, 0));
    A.wy0 = min(max(iy - R, 0), max(A.rows - S, 0));
    A.S = fits ? S : 0;
    A.pad[0] = 0; A.pad[1] = 0;
    rec[kid] = A;
}

// floor(x + 0.5) of a non-negative float below 2^31 in one instruction (the sum is formed exactly, not in float)
__device__ __forceinline__ uint32_t hs_round_half_up(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return (uint32_t)r;
}

// LDS plan (dynamic): BLUR [ raw | hb | win S*S u8 ] (blur_window.h); otherwise the raw window, rows of WP bytes.
// SF != 0: every keypoint is known to need exactly an SF x SF window (detector keypoints: size 31, crop scale 1 ->
// 48), so the blur's index arithmetic (divisions by the group / column-pair counts) folds to constants.
// What the kernel's time is made of (tools/microbench/hs_stage.py, 40 000 keypoints): workgroup turnaround 12 us, the
// warp's arithmetic 26 us, its pixel fetches 61 us when they are gathers from memory -- the vector cache looks up one
// 128-byte line per clock, and the 64 pixels of a wave lie in 12+ image rows whatever the lane mapping -- which is why
// the window is staged in LDS by row-coalesced loads first (50 line look-ups per keypoint instead of 450); votes 50 us.
// Everything that does not depend on the pixel is hoisted: a thread's four patch pixels share their column (warp: x,
// m00 x, m10 x; votes: x, cf, column offset), the 2^17 fixed-point scale rides in the weight table (stored in the threads' own order:
// one coalesced load), the magnitude sqrtf(dx^2 + dy^2) in the orientation table, rounding to fixed point is one instruction.
template <bool BLUR, int SF>
__global__ __launch_bounds__(256) void patch_sift_kernel(
    const HsRec* __restrict__ rec, const int* __restrict__ d_count, int n, int smax,
    const float* __restrict__ vote_weight16 /*4 x 256: weight x 2^17 of pixel k of thread t*/,
    const float2* __restrict__ grad_lut /*511*511 {orientation bin, magnitude}*/,
    float taps0, float taps1, float taps2, float taps3,
    uint16_t* __restrict__ responses /* n x HS_KB bf16 */, float* __restrict__ dbg_responses /* n x 129 or null */, int dbg_arg, const HsBatch frames)
{
    const int dbg = EFX_DBG(dbg_arg);
    if (frames.nframes > 1) {
        const size_t f = blockIdx.y;
        d_count = frames.counts.count[f]; rec += f * frames.kp_stride; responses += f * frames.kp_stride * HS_KB;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Patch rows HS_PP bytes apart.  With 32 the gradient reads of a wave -- 16 cells x 4 pixels: x = 8 (cell & 3) + w, y = 8 (cell >> 2) --
    // met in FOUR banks, four rows deep (rows 8 apart are 256 bytes = one bank row apart); 36 puts the 16 cells' dwords in 16
    // different banks, and the warp's 8 x 8 stores likewise (-DHS_PP=32: the old pitch, A/B)
    __shared__ __attribute__((aligned(4))) uint8_t s_patch[32 * HS_PP];
    // Histogram in 15.17 fixed point (order-independent integer sums; a bin collects at most 64 pixel-weights x 361 of
    // magnitude < 2^15, so 17 fractional bits fill an unsigned 32-bit counter).  A pixel votes for TWO adjacent orientation bins of
    // each of four cells: the pair goes out as ONE 64-bit LDS atomic on two packed 32-bit counters (a counter stays below
    // 2^32, so nothing carries from the low into the high one).  Slot p of a cell holds the pair (p, p + 1), p = 0 .. 8; a
    // bin is the sum of its two homes.  Half the atomics of one counter per bin.
    __shared__ unsigned long long s_h64[6 * 6 * 9];
    __shared__ float s_rf[32];
    __shared__ int s_roff[32];                                           // byte offset of the cell row / column in s_h64

    // neighbouring keypoints (canonical order) on the same XCD: their windows share L2 lines; chunked over the count
    const int count = d_count ? min(*d_count, n) : n;
    if ((int)blockIdx.x >= count) return;
    const int kid = xcd_chunked(blockIdx.x, count);
    const int tid = threadIdx.x;

    const HsRec A = rec[kid];                                            // uniform address: scalar loads
    const uint8_t* img = A.img; const int pitch = A.pitch, rows = A.rows, cols = A.cols;
    for (int i = tid; i < 6 * 6 * 9; i += 256) s_h64[i] = 0ull;          // ordered before the votes by the barriers below
    // HistBin rows/cols (hash_sift.cpp:162-184), kpScale = 1/6; cellw == cellh: one table serves rows and columns
    if (tid >= 64 && tid < 64 + 32) {
        const int i = tid - 64;
        const float kp_scale = 1.f / 6;
        const float cellh = 3.f * (kp_scale * (float)32 * 0.5f);
        const float scaleR = 1.f / cellh;
        const float bin = scaleR * ((float)i - 0.5f * (float)32) + ((float)(4 / 2) - 0.5f);
        const int bi = (int)floorf(bin);
        s_roff[i] = (bi + 1) * 9 * 8; s_rf[i] = bin - (float)bi;
    }
    const bool staged = A.S != 0;                                        // the window is in LDS
    const bool fits = !BLUR || staged;
    const int S = SF ? SF : (A.S ? A.S : smax);
    const int wx0 = A.wx0, wy0 = A.wy0;
    const BlurGeom bg(S);
    uint8_t* raw = smem;
    float* hb = reinterpret_cast<float*>(smem + bg.raw_bytes());
    const int WP = 4 * ((3 + S + 3) >> 2);                               // window row pitch: a multiple of 4 (the warp reads dword pairs)
    const int wofs = BLUR ? (int)(bg.raw_bytes() + bg.hb_bytes()) : (wx0 & 3);      // byte offset of window pixel (0, 0) in smem
    uint8_t* win = smem + wofs;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, rows * pitch, 0x00020000);

    if (BLUR && fits) {
        efx_blur_window_lds<256>(img, pitch, rows, cols, wx0, wy0, S, raw, hb, taps0, taps1, taps2, taps3, tid,
            [&](int r0, int i, int c, uint32_t pk) {
                *reinterpret_cast<uint16_t*>(win + r0 * WP + c + i * WP) = (uint16_t)pk;         // c is even
            });
    }
    if (!BLUR && staged) {
        // raw window rows as aligned dwords (16 lanes per row); a frame smaller than the window: rows beyond it read as 0
        // (range check of the buffer resource), columns beyond it hold the next row's bytes -- neither is ever used
        const int ndw = WP >> 2, base = wy0 * pitch + (wx0 & ~3);
        for (int r = tid >> 4; r < S; r += 16)
            for (int jj = tid & 15; jj < ndw; jj += 16)
                *reinterpret_cast<uint32_t*>(smem + r * WP + 4 * jj) = __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + r * pitch + 4 * jj, 0, 0);
    }
    __syncthreads();
    if (dbg == 5) return;

    // warpAffineLinear, hash_sift.cpp:68-109.  Thread (wave w, lane i) takes column x = 8 w + (i & 7) of rows
    // (i >> 3) + 8 k: a wave covers an 8 x 8 block of the patch per step.  Both pixels of an image row come from ONE
    // 2-byte load at any byte alignment (global: through a buffer resource, 32-bit offsets; blurred window: LDS).
    {
        const int x = 8 * (tid >> 6) + (tid & 7), y0 = (tid >> 3) & 7;
        const float ax = A.m00 * (float)x, bx = A.m10 * (float)x;
        const int wbase = wofs - wy0 * WP - wx0;                          // smem byte of image pixel (0, 0), were the window that large
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float fy = (float)(y0 + 8 * k);
            const float u = ax + A.m01 * fy + A.m02;
            const float v = bx + A.m11 * fy + A.m12;
            const float fu = floorf(u), fv = floorf(v);
            const int ui = (int)fu, vi = (int)fv;
            uint8_t val = 0;
            if (fits && (unsigned)ui < (unsigned)(cols - 1) && (unsigned)vi < (unsigned)(rows - 1)) {      // ui >= 0, ui + 1 < cols, ...
                uint32_t top, bot;
                if (staged) {
                    // Inside the frame means inside the window: R covers floor(u), floor(u) + 1 of every patch pixel, and the
                    // window is only ever shifted to stay in the frame.  The pixel pair may straddle a dword: both dwords in
                    // one ds_read2_b32, the pair shifted down (an unaligned ds_read_u16 works, at ~50 clocks per wave).
                    const int pb = vi * WP + ui + wbase;
                    const uint32_t* pw = reinterpret_cast<const uint32_t*>(smem + (pb & ~3));
                    top = __builtin_amdgcn_alignbyte(pw[1], pw[0], pb & 3);
                    bot = __builtin_amdgcn_alignbyte(pw[(WP >> 2) + 1], pw[WP >> 2], pb & 3);
                } else {
                    const int off = vi * pitch + ui;
                    top = __builtin_amdgcn_raw_buffer_load_b16(rsrc, off, 0, 0);
                    bot = __builtin_amdgcn_raw_buffer_load_b16(rsrc, off, pitch, 0);
                }
                const float p00 = (float)(top & 0xffu), p01 = (float)((top >> 8) & 0xffu);
                const float p10 = (float)(bot & 0xffu), p11 = (float)((bot >> 8) & 0xffu);
                const float du = u - fu, dv = v - fv;
                const float t0 = (1 - du) * p00 + du * p01;
                const float t1 = (1 - du) * p10 + du * p11;
                const float t2 = (1 - dv) * t0 + dv * t1;
                int iv = (int)(t2 + 0.5f);
                if (iv > 255) iv = 255;
                val = (uint8_t)iv;
            }
            s_patch[(y0 + 8 * k) * HS_PP + x] = val;
        }
    }
    __syncthreads();

    if (dbg == 1) return;
    // gradients, magnitude, orientation of the 30x30 interior (hash_sift.cpp:244-260) and the trilinear vote
    // (distribute, hash_sift.cpp:193-198, 262-290): a lane per pixel, 4 fixed-point atomic adds (two orientation bins each).
    // Lane -> pixel mapping: neighbouring lanes take pixels of DIFFERENT 8x8 cells (16 cells, then the next pixel of each
    // cell), so the atomics of one wave instruction spread over many histogram bins; smooth (blurred) patches, where
    // neighbouring pixels vote for the same bins, otherwise serialise on a few LDS addresses.  A thread's four pixels are
    // (x, y0 + 2 k).
    {
        const int cell = tid & 15, w = tid >> 4;
        const int x = 8 * (cell & 3) + (w & 7), y0 = 8 * (cell >> 2) + (w >> 3);        // patch pixel (x + 1, y + 1)
        if (x < 30) {
            const float cf = s_rf[x + 1];
            const int coff = s_roff[x + 1];
            const __amdgpu_buffer_rsrc_t lut_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(grad_lut), 0, 511 * 511 * 8, 0x00020000);
            unsigned long long sink = 0;                                                // EFX_DEBUG_HS=6 only
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int y = y0 + 2 * k;
                if (y >= 30) continue;
                const uint8_t* pc = s_patch + (y + 1) * HS_PP + (x + 1);
                const int idx = (int)pc[1] - (int)pc[-1], idy = (int)pc[-HS_PP] - (int)pc[HS_PP];
                // scaleO * atan2f(dy, dx) and sqrtf(dx^2 + dy^2): dx, dy are integers in [-255, 255], so the host tabulates
                // the CPU code's own libm results for all 511 x 511 gradients (hash_sift.cpp:254-258)
                const efx_u32x2 gw = __builtin_amdgcn_raw_buffer_load_b64(lut_rsrc, (idy * 511 + idx + 255 * 512) * 8, 0, 0);
                const float2 g = dbg == 7 ? make_float2(3.f + 0.01f * (float)(idx + idy), (float)(idx * idx + idy * idy))
                                          : make_float2(__uint_as_float(gw.x), __uint_as_float(gw.y));
                const float mag = (dbg == 8 ? 32768.f : vote_weight16[k * 256 + tid]) * g.y;   // x 2^17: exact, commutes with every product below
                const float fo = floorf(g.x);
                const float of = g.x - fo;
                const int oi = (int)fo & 7;                                             // bins -4 .. 4 -> 4 .. 7, 0 .. 4
                const float rf = s_rf[y + 1];
                const float v1 = rf * mag, v0 = mag - v1;
                const float v01 = cf * v0, v00 = v0 - v01;
                const float v11 = cf * v1, v10 = v1 - v11;
                const float a4[4] = { v00, v01, v10, v11 };
                unsigned char* home = reinterpret_cast<unsigned char*>(s_h64) + 6 * s_roff[y + 1] + coff + 8 * oi;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float b1 = of * a4[q], b0 = a4[q] - b1;
                    // 15.17 fixed point, rounded half up (contributions are >= 0).  A bin holds hundreds of gray levels, so 2^-18
                    // per vote is ~1e-7 relative.
                    const unsigned long long pair = (unsigned long long)hs_round_half_up(b0) | ((unsigned long long)hs_round_half_up(b1) << 32);
                    if (dbg == 6) sink += pair;
                    else atomicAdd(reinterpret_cast<unsigned long long*>(home + ((q >> 1) * 6 + (q & 1)) * 72), pair);
                }
            }
            if (dbg == 6) s_h64[tid] = sink;
        }
    }
    __syncthreads();
    if (dbg == 2 || tid >= 64) return;
    // ---- the rest is one wave's work, lane t owns elements t and t + 64 of the 128-vector: no LDS round trips, no barriers
    // fixed point -> float, circular fold of the 10 orientation bins into 8 (hash_sift.cpp:293-308: ph[0] += ph[8], ph[1] += ph[9])
    auto bin_of = [&](const unsigned long long* hc, int k) -> float {
        const uint32_t v = (k < 9 ? (uint32_t)hc[k] : 0u) + (k >= 1 ? (uint32_t)(hc[k - 1] >> 32) : 0u);      // pairs (k, k+1) and (k-1, k)
        return (float)v * (1.f / 131072.f);              // one rounding (u32 -> f32), the power of two is exact
    };
    auto element = [&](int e) -> float {
        const int r = e >> 5, c = (e >> 3) & 3, k = e & 7;
        const unsigned long long* hc = s_h64 + ((r + 1) * 6 + (c + 1)) * 9;
        float v = bin_of(hc, k);
        if (k < 2) v += bin_of(hc, k + 8);
        return v;
    };
    float d0 = element(tid), d1 = element(tid + 64);
    if (dbg == 3) { s_patch[tid] = (uint8_t)(d0 + d1); return; }
    // L2 normalise, clip at 0.2, renormalise, x512 -> uchar (hash_sift.cpp:311-330).  The CPU code adds the 128 squares
    // serially; here a fixed tree does (element i with i + 64, then the butterfly 32, 16, .. 1: every lane ends with the
    // same sum).  The order is part of the device arithmetic the tests' CPU model reproduces bit for bit; against the
    // serial sum the norm moves by ~1e-7 relative.
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        float t = d0 * d0 + d1 * d1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t = t + __shfl_xor(t, off, 64);
        float norm = sqrtf(t);
        if (norm < 1.1920929e-07f) norm = 1.1920929e-07f;          // FLT_EPSILON
        const float scale = 1.f / norm;
        d0 *= scale; d1 *= scale;
        if (pass == 0) { d0 = d0 < 0.2f ? d0 : 0.2f; d1 = d1 < 0.2f ? d1 : 0.2f; }
    }
    if (dbg == 4) { s_patch[tid] = (uint8_t)(d0 + d1); return; }
    // the 129-vector {1, 128 x uchar} as bf16 (exact: integers <= 255 need 8 mantissa bits), K padded to HS_KB with zeros
    uint16_t* out = responses + (size_t)kid * HS_KB;
    float q0 = rintf(512.f * d0), q1 = rintf(512.f * d1);             // saturate_cast<uchar>: cvRound + clamp
    q0 = q0 < 0.f ? 0.f : (q0 > 255.f ? 255.f : q0);
    q1 = q1 < 0.f ? 0.f : (q1 > 255.f ? 255.f : q1);
    out[1 + tid] = (uint16_t)(__float_as_uint(q0) >> 16);
    out[65 + tid] = (uint16_t)(__float_as_uint(q1) >> 16);
    if (tid < HS_KB - 128) out[tid == 0 ? 0 : 128 + tid] = tid == 0 ? (uint16_t)0x3f80 : (uint16_t)0;     // element 0, padding 129 .. 143
    if (dbg_responses) {
        dbg_responses[(size_t)kid * 129 + 1 + tid] = q0;
        dbg_responses[(size_t)kid * 129 + 65 + tid] = q1;
        if (tid == 0) dbg_responses[(size_t)kid * 129] = 1.f;
    }
}

// ================================================================================================
// Projection + sign + pack.  T[i][j] = sum_k R[i][k] * W[j][k]  (matmulAndSign, hash_sift.cpp:353-378).
// The one real GEMM of the path, on the bf16 matrix cores: v_mfma_f32_32x32x16_bf16 (16x the rate of the fp32 MFMA the
// first version used).  R is integer valued (0..255 and the leading 1): exact in bf16.  W is split on the host into three
// bf16 terms whose sum is W exactly, so every product R * W_t is exact in the fp32 accumulator and T differs from the
// fp32 FMA chain only by the order / rounding of the accumulation (well inside the stated 2e-3 tolerance on T).
// A wave owns 32 output bits and keeps their 3 x 9 weight operands (W terms x K steps) in registers for its whole life; the
// four waves of a workgroup (128 adjacent bits) walk 64-keypoint tiles of R that the workgroup stages in LDS once.  Per tile
// and wave: 54 MFMAs on two alternating accumulators, sign + pack in the lanes, one 16-byte store per keypoint -- no T
// matrix in HBM, no separate binarize pass (cuda_hash_sift.cu:414-435).  One wave per SIMD (the kernel wants ~270 registers).
// Operand layout: lane l holds row / column (l & 31) and the 8 consecutive k of half (l >> 5) of the K step, for A and B
// alike, so whatever order the hardware contracts the 16 k of a step in, A and B elements meet at equal k.
// Round 3 (C4, 40 000 keypoints x 512 bits, tools/microbench/hs_timing.sh): 26.8 -> 22.3 us.  What it was made of: a
// one-accumulator MFMA chain (dependency-paced: 28 % of the matrix pipe), 32 ballots per tile funnelled through SGPRs, and --
// once those were gone -- v_cmp / s_nop / v_cndmask triples the compiler made of the sign test.  Per tile now: stage +
// fetch 0.33 us, LDS reads + MFMAs 0.8 us (the pipe's own 54 x 32 cycles), pack 0.3 us, barrier + store 0.2 us; launch and
// prologue (weights) 5.8 us.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// lane `lane` (a constant) of `old` := the wave-uniform `sval`
#define efx_writelane(sval, old, lane) ([&]() { uint32_t o_ = (old); asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(o_) : "s"(sval), "n"(lane)); return o_; }())

#ifndef HS_PROJ_WAVES
#define HS_PROJ_WAVES 1          // waves per SIMD the kernel is compiled for (register budget 512 / waves)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HS_PROJ_WAVES, HS_PROJ_WAVES))) void project_sign_kernel(const uint16_t* __restrict__ Rm, const uint16_t* __restrict__ Wb,
                                                           const int* __restrict__ d_count, int n, int nbits,
                                                           uint8_t* __restrict__ desc, size_t desc_pitch, float* __restrict__ dbg_T, const HsBatch hb)
{
    if (hb.nframes > 1) {
        const size_t f = blockIdx.y;
        d_count = hb.counts.count[f]; Rm += f * hb.kp_stride * HS_KB; desc = hb.descs.desc[f];
    }
    const int count = d_count ? min(*d_count, n) : n;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // the waves that share a row group (one per column tile) run on ONE XCD: they read the same rows of R through one L2
    const int gw = xcd_chunked(blockIdx.x, gridDim.x) * 4 + wid;   // logical global wave index
    const int ntn = nbits >> 5;                                   // 32-bit column tiles (a multiple of 4: a workgroup's four
    const int n0 = (gw % ntn) * 32;                               // waves own 128 adjacent bits of the same row group)
    const int mgroup = gw / ntn, nmgroups = (gridDim.x * 4) / ntn;
    const int li = lane & 31, lk = lane >> 5;
    constexpr int KS = HS_KB / 16;                                // 9 K steps
    // The product is formed TRANSPOSED, T^T = W R^T: the weights are the MFMA's A operand (rows = output bits), the keypoints
    // its B operand (columns), so lane (li, lk) ends up with 16 projections of ONE keypoint -- rows i = (r & 3) + 8 (r >> 2)
    // + 4 lk of the C tile in register r -- and packs their signs itself, two VALU operations per value, instead of 32
    // ballots per tile funnelled through SGPRs (round 2: a third of the kernel's time).  The packed word of a keypoint is
    // bit b = r + 16 lk; a descriptor byte holds its 8 bits MSB first (hash_sift.cpp:367-374), i.e. descriptor bit p is bit
    // 8 (p / 8) + 7 - p % 8 of the little-endian word, so MFMA row i carries W row n0 + p(b(i)):
    const int r_of_li = (li & 3) + 4 * (li >> 3), bword = r_of_li + 16 * ((li >> 2) & 1);
    const int wrow = n0 + 8 * (bword >> 3) + 7 - (bword & 7);
    // this wave's weights: b[t][ks] = W_t[wrow][16 ks + 8 lk .. + 8]
    bf16x8 b[3][KS];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const uint4* p = reinterpret_cast<const uint4*>(Wb + ((size_t)t * nbits + wrow) * HS_KB + 8 * lk);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) b[t][ks] = __builtin_bit_cast(bf16x8, p[2 * ks]);
    }
    // A workgroup's four waves multiply the SAME rows of R (by four column tiles), so a 64-row tile is fetched once per
    // workgroup into LDS (round 3; every wave used to fetch its 32 rows for itself).  Thread t fetches 16-byte pieces t,
    // t + 256, ... of the tile (64 rows x 18 pieces); the next tile's pieces are in flight while this one is multiplied.  LDS
    // rows are 304 bytes apart: the 16 lanes of a ds_read_b128 group land on 16 distinct 4-bank slots.
    // Each wave multiplies the tile's two 32-row halves into TWO accumulators, alternating: an MFMA on the accumulator of
    // the MFMA before it is dependency-paced (the one-accumulator chain of round 2 ran at 28 % of the matrix pipe: 75 cycles
    // per 32-cycle instruction, profiles/r03_hashsift_pmc.txt), two alternating chains issue back to back.
    const int stiles = (count + 63) >> 6;
    constexpr int APITCH = 304, NPIECE = 64 * 18, NFETCH = (NPIECE + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char s_a[2][64 * APITCH];
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[2][64][4];      // [buffer][row of the tile][wave]: 128 bits per row
    auto fetch_tile = [&](int st, uint4 (&dst)[NFETCH]) {
#pragma unroll
        for (int u = 0; u < NFETCH; u++) {
            const int idx = (int)threadIdx.x + 256 * u;
            const int row = idx / 18, piece = idx - row * 18;
            const int arow = max(min(st * 64 + row, count - 1), 0);
            dst[u] = make_uint4(0u, 0u, 0u, 0u);
            if (idx < NPIECE) dst[u] = *reinterpret_cast<const uint4*>(Rm + (size_t)arow * HS_KB + 8 * piece);
        }
    };
    // 16-byte stores of a row's 128 bits need a 16-byte aligned destination
    const bool wide = desc != nullptr && (((reinterpret_cast<uintptr_t>(desc) | desc_pitch) & 15u) == 0);
    auto stage_tile = [&](int b_, const uint4 (&src)[NFETCH]) {
#pragma unroll
        for (int u = 0; u < NFETCH; u++) {
            const int idx = (int)threadIdx.x + 256 * u;
            const int row = idx / 18, piece = idx - row * 18;
            if (idx < NPIECE) *reinterpret_cast<uint4*>(&s_a[b_][row * APITCH + 16 * piece]) = src[u];
        }
    };
    // ONE barrier per tile.  While tile i is multiplied out of buffer b, the same instruction stream moves tile i + 1 from
    // registers into buffer b ^ 1 and requests tile i + 2 (a handful of LDS stores and loads between 54 MFMAs: they ride in
    // the matrix pipe's shadow); the tile's sign words go to s_bits[b]; after the barrier 64 threads store them.
#ifdef HS_PROJ_TIMING
#define HSP_T(i) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 0) hsp_t[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
    unsigned long long hsp_t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int hsp_iter = 0;
#else
#define HSP_T(i) do { } while (0)
#endif
    HSP_T(0);
    uint4 nxt[NFETCH];
    if (mgroup >= stiles) return;                                     // workgroup-uniform
    fetch_tile(mgroup, nxt);
    stage_tile(0, nxt);
    if (mgroup + nmgroups < stiles) fetch_tile(mgroup + nmgroups, nxt);
    __syncthreads();
    HSP_T(1);
    int buf = 0;
    for (int st = mgroup; st < stiles; st += nmgroups, buf ^= 1) {
        HSP_T(2);
        const bool more = st + nmgroups < stiles;
        if (more) stage_tile(buf ^ 1, nxt);          // its readers (tile i - 1) are past the barrier that ended tile i - 1
        if (st + 2 * nmgroups < stiles) fetch_tile(st + 2 * nmgroups, nxt);
        HSP_T(3);
        const unsigned char* a0p = &s_a[buf][li * APITCH + 16 * lk];
        const unsigned char* a1p = a0p + 32 * APITCH;
        f32x16 acc0 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, acc1 = acc0;
        // smallest terms first: the fp32 accumulators round them in before the large ones arrive
#pragma unroll
        for (int t = 2; t >= 0; t--)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a0p + 32 * ks));
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a1p + 32 * ks));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[t][ks], a0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[t][ks], a1, acc1, 0, 0, 0);
            }
        // C layout 32x32: col = lane & 31 (keypoint), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output bit, permuted
        // as above).  sign(T > 0) = clamp((int)bits, 0, 1): one v_med3_i32, then one v_lshl_or_b32 per value.
        const int m0 = st * 64;
        HSP_T(4);
        uint32_t word[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const f32x16& acc = h ? acc1 : acc0;
            uint32_t half = 0u;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                // clamp((int)bits, 0, 1) as ONE v_med3_i32, written by hand: the compiler makes v_cmp + s_nop + v_cndmask + v_or3
                // of it (a VALU write of VCC needs wait states before the select reads it): 0.62 -> 0.3 us per tile
                int sgn;
                asm("v_med3_i32 %0, %1, 0, 1" : "=v"(sgn) : "v"(__float_as_uint(acc[r])));
                half |= (uint32_t)sgn << r;
            }
            if (dbg_T) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int i = m0 + 32 * h + li, bb = r + 16 * lk;
                    if (i < count) dbg_T[(size_t)i * nbits + n0 + 8 * (bb >> 3) + 7 - (bb & 7)] = acc[r];
                }
            }
            const uint32_t other = (uint32_t)__shfl_xor((int)half, 32, 64);       // the keypoint's other 16 bits (lane ^ 32)
            word[h] = half | (other << 16);                                         // valid in lanes 0 .. 31 (lk == 0)
        }
        HSP_T(5);
        if (wide) {
            // the workgroup's four waves hold 128 adjacent bits of the same 64 rows: one 16-byte store per row
            if (lane < 32) { s_bits[buf][lane][wid] = word[0]; s_bits[buf][32 + lane][wid] = word[1]; }
            __syncthreads();
            HSP_T(6);
            if (threadIdx.x < 64 && m0 + (int)threadIdx.x < count) {
                const uint4 v = *reinterpret_cast<const uint4*>(&s_bits[buf][threadIdx.x][0]);
                *reinterpret_cast<uint4*>(desc + (size_t)(m0 + threadIdx.x) * desc_pitch + (n0 & ~127) / 8) = v;
            }
        } else {
            if (desc != nullptr && lane < 32) {
                if (m0 + lane < count) *reinterpret_cast<unsigned*>(desc + (size_t)(m0 + lane) * desc_pitch + n0 / 8) = word[0];
                if (m0 + 32 + lane < count) *reinterpret_cast<unsigned*>(desc + (size_t)(m0 + 32 + lane) * desc_pitch + n0 / 8) = word[1];
            }
            __syncthreads();                        // the tile staged for the next trip is complete
        }
#ifdef HS_PROJ_TIMING
        HSP_T(7);
        if (blockIdx.x == 0 && threadIdx.x == 0 && hsp_iter++ < 3)
            printf("proj wg0 ticks(10ns): prologue %llu | stage+fetch %llu | lds+mfma %llu | pack %llu | bits+barrier %llu | store %llu\n",
                   hsp_t[1] - hsp_t[0], hsp_t[3] - hsp_t[2], hsp_t[4] - hsp_t[3], hsp_t[5] - hsp_t[4], hsp_t[6] - hsp_t[5], hsp_t[7] - hsp_t[6]);
#endif
    }
}

} // namespace

static int hs_smax_for(float max_size, float crop_scale)
{
    const float sg = fabsf(crop_scale * max_size / 32.f);
    const int R = (int)floorf(sg * 22.63f + 2.01f);
    return 2 * R + 2;
}

hipError_t efx_launch_hashsift(const DescribeLaunch& a, const HashSiftDev& h, hipStream_t stream)
{
    if (a.n <= 0) return hipSuccess;
    const float max_size = a.max_size > 0.f ? a.max_size : (float)EFX_PATCH_SIZE;
    int S = hs_smax_for(max_size, a.scale_factor);
    size_t lds = 0;
    if (a.blur) {
        const BlurGeom bg(S);
        lds = bg.raw_bytes() + bg.hb_bytes() + (size_t)S * (4 * ((3 + S + 3) >> 2)) + 4;         // + 4: the warp reads dword pairs
        lds = (lds + 15) & ~(size_t)15;
        if (lds > 140 * 1024) return hipErrorInvalidValue;
    } else {
        if (S > 120) S = 120;                                  // larger keypoints gather from memory (8 workgroups per CU fit up to here)
        lds = (size_t)S * (4 * ((3 + S + 3) >> 2)) + 16;
    }
    float t[7];
    efx_gaussian_taps_host(t);
    const int dbg = a.dbg_hs;
    const float* mag = h.W + (size_t)h.nbits * HS_KPAD;    // the pixel weights (x 2^17, in the vote loop's thread order) are stored behind W
    const float2* lut = reinterpret_cast<const float2*>(mag + 1024);      // followed by the 511x511 {orientation bin, magnitude} table
    static_assert(sizeof(HsRec) == EFX_HS_REC_BYTES, "scratch sizing in efx_api.cpp");
    HsRec* rec = static_cast<HsRec*>(h.records);
    const bool fixed48 = a.blur && S == 48 && a.uniform_size;
    HsBatch hb = {};
    const int NF = a.nframes > 1 ? a.nframes : 1;
    if (NF > 1) { hb.nframes = NF; hb.kp_stride = a.kp_stride; hb.pyr_stride = a.pyr_stride; hb.imgs = a.imgs; hb.counts = a.counts; hb.descs = a.descs; }
    hipLaunchKernelGGL(hs_record_kernel, dim3((a.n + 255) / 256, NF), dim3(256), 0, stream, a.img0, a.pitch0, a.rows0, a.cols0,
                       a.pyramid, a.d_table, a.kp4, a.kps5, a.kps5_pitch, a.kp_level, a.d_count, a.n, a.scale_factor, S, fixed48 ? 48 : 0, a.blur ? 0 : 1, rec, hb);
    if (fixed48) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sift_kernel<true, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((patch_sift_kernel<true, 48>), dim3(a.n, NF), dim3(256), lds, stream, rec, a.d_count, a.n, S, mag, lut,
                           t[0], t[1], t[2], t[3], h.responses, h.dbg_responses, dbg, hb);
    } else if (a.blur) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sift_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((patch_sift_kernel<true, 0>), dim3(a.n, NF), dim3(256), lds, stream, rec, a.d_count, a.n, S, mag, lut,
                           t[0], t[1], t[2], t[3], h.responses, h.dbg_responses, dbg, hb);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sift_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((patch_sift_kernel<false, 0>), dim3(a.n, NF), dim3(256), lds, stream, rec, a.d_count, a.n, S, mag, lut,
                           t[0], t[1], t[2], t[3], h.responses, h.dbg_responses, dbg, hb);
    }
    if (a.desc || h.dbg_T) {
        // persistent waves: HS_PROJ_WAVES per SIMD (the weights occupy 108 VGPRs), each owning one 32-bit column tile
        const int ntn = h.nbits / 32;
        int nblk = (256 * HS_PROJ_WAVES) / ntn * ntn;          // one workgroup (four waves) per CU and wave slot, a multiple of the column tiles
        const int need = (((a.n + 63) / 64) * ntn + 3) / 4;     // never more waves than (row tile, column tile) pairs
        if (nblk > need) nblk = (need + ntn - 1) / ntn * ntn;
        hipLaunchKernelGGL(project_sign_kernel, dim3(nblk, NF), dim3(256), 0, stream,
                           h.responses, h.Wb, a.d_count, a.n, h.nbits, a.desc, a.desc_pitch, h.dbg_T, hb);
    }
    return hipGetLastError();
}
