// hashsift_kernels.hip -- HashSIFT descriptor for gfx950: PatchSIFT 129-vectors + fp32 MFMA projection.
//
// Arithmetic: the reference CPU descriptor, modules/efficient_features/src/hash_sift.cpp
//   rectifyPatch / warpAffineLinear :68-138, HistBin :162-184, distribute :193-198,
//   computePatchSIFT :200-331, normalize :150-160, matmulAndSign :353-378.
//
// MI355X design
//   * one workgroup per keypoint; the (optionally Gaussian-blurred, spec S6) window the rotated 32x32
//     patch can touch is staged in LDS, so detectAndCompute needs no global blur pass;
//   * the 6x6x10 gradient histogram is accumulated in 32.32 FIXED POINT with LDS integer atomics, one lane per
//     pixel: integer addition is associative, so the result does not depend on the order the lanes arrive in
//     (the reference's CUDA kernel adds floats with shared-memory atomics, cuda_hash_sift.cu:282-289, and is
//     order-nondeterministic), and it is at least as accurate as the CPU loop's sequentially rounded float
//     sums (hash_sift.cpp:233-290): the two differ by a last-place rounding that changes 1.5e-6 of the
//     129-vector elements by one unit (measured over 40 000 keypoints), 1/60 of the stated tolerance;
//   * the Gaussian pixel weights expf(...) (30x30) and the orientation bins scaleO*atan2f(dy,dx) (511x511
//     integer gradients) are tables computed on the host with the same libm calls the CPU code makes;
//     sqrtf is IEEE; cosf/sinf of the keypoint angle are glibc's (glibc_sincosf.h, bit-identical to the host libm);
//   * projection T = R[N x 129] . W^T is the one real GEMM of the path: v_mfma_f32_32x32x2_f32 (exact fp32
//     FMA chain), fused with the sign test and MSB-first bit packing (no T matrix in HBM, no separate
//     binarize pass as in cuda_hash_sift.cu:414-435).

#include "efx_device.h"
#include "blur_window.h"
#include "glibc_sincosf.h"
#include <stdlib.h>

#define HS_KPAD 132   // 129 padded to a multiple of 4 floats (row pitch of the fp32 weights)
#define HS_KB 144     // K of the bf16 projection: 129 padded to 9 MFMA steps of 16

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AffineF { float m00, m01, m02, m10, m11, m12, pad0, pad1; };

// LDS plan (dynamic, BLUR only): [ raw | hb | win S*S u8 ] (blur_window.h)
// SF != 0: every keypoint is known to need exactly an SF x SF window (detector keypoints: size 31, crop scale 1 ->
// 48), so the blur's index arithmetic (divisions by the group / column-pair counts) folds to constants
template <bool BLUR, int SF>
#ifndef HS_NT
#define HS_NT 256
#endif
__global__ __launch_bounds__(HS_NT) void patch_sift_kernel(
    const uint8_t* __restrict__ img0, int pitch0, int rows0, int cols0,
    const uint8_t* __restrict__ pyramid, const LevelTable* __restrict__ T,
    const float4* __restrict__ kp4, const int* __restrict__ kp_level, const int* __restrict__ d_count, int n,
    float crop_scale, int smax, const float* __restrict__ mag_scale /*30*30*/, const float* __restrict__ obin_lut /*511*511*/,
    float taps0, float taps1, float taps2, float taps3,
    uint16_t* __restrict__ responses /* n x HS_KB bf16 */, float* __restrict__ dbg_responses /* n x 129 or null */, int dbg_arg)
{
    const int dbg = EFX_DBG(dbg_arg);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ AffineF s_aff;
    __shared__ uint8_t s_patch[32 * 32];
    // Histogram in 16.16 fixed point (order-independent integer sums).  A pixel votes for TWO adjacent orientation bins of
    // each of four cells: the pair goes out as ONE 64-bit LDS atomic on two packed 32-bit counters (a counter stays below
    // 2^31, so nothing carries from the low into the high one).  Per cell 5 words hold the pairs (0,1) (2,3) .. (8,9), 4 more
    // the pairs (1,2) (3,4) (5,6) (7,8); a bin is the sum of its two homes.  Half the atomics of one counter per bin.
    __shared__ unsigned long long s_h64[6 * 6 * 9];
    __shared__ float s_hist[6 * 6 * 10];
    __shared__ float s_desc[128];
    __shared__ float s_rf[32], s_cf[32];
    __shared__ int s_ri[32], s_ci[32];
    __shared__ float s_scale;

    // neighbouring keypoints (canonical order) on the same XCD: their windows share L2 lines; chunked over the count
    const int count = d_count ? min(*d_count, n) : n;
    if ((int)blockIdx.x >= count) return;
    const int kid = xcd_chunked(blockIdx.x, count);
    const int tid = threadIdx.x;

    const float4 kp = kp4[kid];
    const uint8_t* img = img0; int pitch = pitch0, rows = rows0, cols = cols0;
    if (kp_level) {
        const int l = kp_level[kid];
        if (l > 0) { const LevelDev& L = T->lv[l]; img = pyramid + L.img_off; pitch = L.pitch; rows = L.rows; cols = L.cols; }
        else { rows = T->lv[0].rows; cols = T->lv[0].cols; }
    }
    const float px = kp.x, py = kp.y, size = kp.z, angle = kp.w;
    for (int i = tid; i < 6 * 6 * 9; i += HS_NT) s_h64[i] = 0ull;      // ordered before the votes by the barriers below

    // rectifyPatch, hash_sift.cpp:111-132
    if (tid == 0) {
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
        AffineF A;
        A.m00 = +cost; A.m01 = -sint; A.m02 = (-cost + sint) * (float)32 / 2.f + px;
        A.m10 = +sint; A.m11 = +cost; A.m12 = (-sint - cost) * (float)32 / 2.f + py;
        A.pad0 = 0; A.pad1 = 0;
        s_aff = A;
    }
    // HistBin rows/cols (hash_sift.cpp:162-184), kpScale = 1/6
    if (tid >= 64 && tid < 64 + 32) {
        const int i = tid - 64;
        const float kp_scale = 1.f / 6;
        const float cellh = 3.f * (kp_scale * (float)32 * 0.5f);
        const float scaleR = 1.f / cellh;
        const float bin = scaleR * ((float)i - 0.5f * (float)32) + ((float)(4 / 2) - 0.5f);
        const int bi = (int)floorf(bin);
        s_ri[i] = bi; s_rf[i] = bin - (float)bi;
        s_ci[i] = bi; s_cf[i] = bin - (float)bi;       // cellw == cellh, same formula for columns
    }


    // window the patch can touch
    const float sg = fabsf(crop_scale * size / 32.f);
    int R = (int)floorf(sg * 22.63f + 2.01f);            // >= 22.63 sg + 1: floor(u) and floor(u) + 1 of every patch pixel are inside
    int S = SF ? SF : 2 * R + 2;
    const bool fits = !BLUR || (SF ? (2 * R + 2 == SF) : (S <= smax && S > 0));
    if (!fits && !SF) S = smax;
    const int ix = (int)floorf(px), iy = (int)floorf(py);
    const int wx0 = min(max(ix - R, 0), max(cols - S, 0));
    const int wy0 = min(max(iy - R, 0), max(rows - S, 0));
    // LDS plan (BLUR): [ raw | hb | win S x S u8 ], raw / hb as blur_window.h lays them out.  Without the blur the warp
    // below gathers its 4 bytes per patch pixel from memory: staging the raw window in LDS first was measured SLOWER
    // (staging 28 us + warp 108 us against 88 us per 40 000 keypoints) -- the kernel's bottleneck is the LDS pipe (the
    // histogram atomics), the gathers go through the texture path, which is otherwise idle.
    const BlurGeom bg(S);
    uint8_t* raw = smem;
    float* hb = reinterpret_cast<float*>(smem + bg.raw_bytes());
    uint8_t* win = smem + bg.raw_bytes() + bg.hb_bytes();

    if (BLUR && fits) {
        efx_blur_window_lds<HS_NT>(img, pitch, rows, cols, wx0, wy0, S, raw, hb, taps0, taps1, taps2, taps3, tid,
            [&](int r, int c, int q0, int q1) {
                *reinterpret_cast<uint16_t*>(win + r * S + c) = (uint16_t)(q0 | (q1 << 8));      // S and c are even
            });
    }
    __syncthreads();
    if (dbg == 5) return;

    // warpAffineLinear, hash_sift.cpp:68-109
    {
        const AffineF A = s_aff;
        for (int i = tid; i < 1024; i += HS_NT) {
            const int y = i >> 5, x = i & 31;
            const float u = A.m00 * (float)x + A.m01 * (float)y + A.m02;
            const float v = A.m10 * (float)x + A.m11 * (float)y + A.m12;
            uint8_t val = 0;
            const int ui = (int)floorf(u), vi = (int)floorf(v);
            if (fits && ui >= 0 && ui + 1 < cols && vi >= 0 && vi + 1 < rows) {
                float p00, p01, p10, p11;
                if (BLUR) {
                    const int lx = min(max(ui - wx0, 0), S - 2), ly = min(max(vi - wy0, 0), S - 2);
                    const uint8_t* p = win + ly * S + lx;
                    p00 = (float)p[0]; p01 = (float)p[1]; p10 = (float)p[S]; p11 = (float)p[S + 1];
                } else {
                    const uint8_t* p = img + (size_t)vi * pitch + ui;
                    p00 = (float)p[0]; p01 = (float)p[1]; p10 = (float)p[pitch]; p11 = (float)p[pitch + 1];
                }
                const float du = u - (float)ui, dv = v - (float)vi;
                const float t0 = (1 - du) * p00 + du * p01;
                const float t1 = (1 - du) * p10 + du * p11;
                const float t2 = (1 - dv) * t0 + dv * t1;
                int iv = (int)(t2 + 0.5f);
                if (iv > 255) iv = 255;
                val = (uint8_t)iv;
            }
            s_patch[i] = val;
        }
    }
    __syncthreads();

    if (dbg == 1) return;
    // gradients, magnitude, orientation of the 30x30 interior (hash_sift.cpp:244-260) and the trilinear vote
    // (distribute, hash_sift.cpp:193-198, 262-290): a lane per pixel, 8 fixed-point atomic adds
    {
        // Lane -> pixel mapping: neighbouring lanes take pixels of DIFFERENT 8x8 cells (16 cells, then the next pixel
        // of each cell), so the atomics of one wave instruction spread over many histogram bins; smooth (blurred)
        // patches, where neighbouring pixels vote for the same bins, otherwise serialise on a few LDS addresses.
        for (int j = tid; j < 1024; j += HS_NT) {
            const int cell = j & 15, w = j >> 4;
            const int x = 8 * (cell & 3) + (w & 7), y = 8 * (cell >> 2) + (w >> 3);           // patch pixel (x+1, y+1)
            if (x >= 30 || y >= 30) continue;
            const int i = y * 30 + x;
            const uint8_t* pc = s_patch + (y + 1) * 32 + (x + 1);
            const int idx = (int)pc[1] - (int)pc[-1], idy = (int)pc[-32] - (int)pc[32];
            const float dx = (float)idx, dy = (float)idy;
            const float mag = mag_scale[i] * sqrtf(dx * dx + dy * dy);
            // scaleO * atan2f(dy, dx): dx, dy are integers in [-255, 255], so the host tabulates the CPU
            // code's own libm result for all 511 x 511 gradients (hash_sift.cpp:254,258)
            const float obin = obin_lut[(idy + 255) * 511 + (idx + 255)];
            int oi = (int)floorf(obin);
            const float of = obin - (float)oi;
            if (oi < 0) oi += 8;
            if (oi >= 8) oi -= 8;
            const float rf = s_rf[y + 1], cf = s_cf[x + 1];
            const int ri = s_ri[y + 1], ci = s_ci[x + 1];
            const float v1 = rf * mag, v0 = mag - v1;
            const float v01 = cf * v0, v00 = v0 - v01;
            const float v11 = cf * v1, v10 = v1 - v11;
            const float a4[4] = { v00, v01, v10, v11 };
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float b1 = of * a4[q], b0 = a4[q] - b1;
                // 16.16 fixed point, rounded to nearest (contributions are >= 0; a bin collects at most 64 pixel-weights
                // x 361 of magnitude: < 2^15).  A bin holds hundreds of gray levels, so 2^-17 per vote is ~1e-7 relative.
                const unsigned long long pair = (unsigned long long)(uint32_t)(b0 * 65536.f + 0.5f) |
                                                ((unsigned long long)(uint32_t)(b1 * 65536.f + 0.5f) << 32);
                atomicAdd(s_h64 + ((ri + 1 + (q >> 1)) * 6 + (ci + 1 + (q & 1))) * 9 + ((oi & 1) ? 5 : 0) + (oi >> 1), pair);
            }
        }
    }
    __syncthreads();
    if (dbg == 2) return;
    for (int i = tid; i < 360; i += HS_NT) {
        const int cell = i / 10, k = i - cell * 10;
        const unsigned long long* hc = s_h64 + cell * 9;
        uint32_t v;
        if (k & 1) v = (uint32_t)(hc[k >> 1] >> 32) + (k < 9 ? (uint32_t)hc[5 + (k >> 1)] : 0u);      // pairs (k-1, k) and (k, k+1)
        else v = (uint32_t)hc[k >> 1] + (k >= 2 ? (uint32_t)(hc[5 + (k >> 1) - 1] >> 32) : 0u);        // pairs (k, k+1) and (k-1, k)
        s_hist[i] = (float)((double)v * (1.0 / 65536.0));
    }
    __syncthreads();
    if (dbg == 3) return;
    // circular fold + copy (hash_sift.cpp:293-308)
    if (tid < 16) {
        const int r = tid >> 2, c = tid & 3;
        float* ph = s_hist + ((r + 1) * 6 + (c + 1)) * 10;
        ph[0] += ph[8];
        ph[1] += ph[9];
        for (int k = 0; k < 8; k++) s_desc[(r * 4 + c) * 8 + k] = ph[k];
    }
    __syncthreads();
    // L2 normalise, clip at 0.2, renormalise, x512 -> uchar (hash_sift.cpp:311-330).  The CPU code adds the 128 squares
    // serially; here a fixed tree does (element i with i + 64, then the butterfly 32, 16, .. 1 inside one wave): 10
    // instructions instead of a 256-instruction chain on one lane.  The order is part of the device arithmetic the tests'
    // CPU model reproduces bit for bit; against the serial sum the norm moves by ~1e-7 relative.
    for (int pass = 0; pass < 2; pass++) {
        if (tid < 64) {
            const float d0 = s_desc[tid], d1 = s_desc[tid + 64];
            float t = d0 * d0 + d1 * d1;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) t = t + __shfl_xor(t, off, 64);
            if (tid == 0) {
                float norm = sqrtf(t);
                if (norm < 1.1920929e-07f) norm = 1.1920929e-07f;          // FLT_EPSILON
                s_scale = 1.f / norm;
            }
        }
        __syncthreads();
        if (tid < 128) {
            float v = s_desc[tid] * s_scale;
            if (pass == 0) v = v < 0.2f ? v : 0.2f;
            s_desc[tid] = v;
        }
        __syncthreads();
    }
    if (dbg == 4) return;
    // the 129-vector {1, 128 x uchar} as bf16 (exact: integers <= 255 need 8 mantissa bits), K padded to HS_KB with zeros
    uint16_t* out = responses + (size_t)kid * HS_KB;
    if (tid < 128) {
        float v = rintf(512.f * s_desc[tid]);                           // saturate_cast<uchar>: cvRound + clamp
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        out[1 + tid] = (uint16_t)(__float_as_uint(v) >> 16);
        if (dbg_responses) dbg_responses[(size_t)kid * 129 + 1 + tid] = v;
    }
    if (tid >= 128 && tid < 128 + HS_KB - 128) {
        const int k = tid == 128 ? 0 : tid;                             // element 0 and the padding 129 .. 143
        out[k] = tid == 128 ? (uint16_t)0x3f80 : (uint16_t)0;
        if (tid == 128 && dbg_responses) dbg_responses[(size_t)kid * 129] = 1.f;
    }
}

// ================================================================================================
// Projection + sign + pack.  T[i][j] = sum_k R[i][k] * W[j][k]  (matmulAndSign, hash_sift.cpp:353-378).
// The one real GEMM of the path, on the bf16 matrix cores: v_mfma_f32_32x32x16_bf16 (16x the rate of the fp32 MFMA the
// first version used).  R is integer valued (0..255 and the leading 1): exact in bf16.  W is split on the host into three
// bf16 terms whose sum is W exactly, so every product R * W_t is exact in the fp32 accumulator and T differs from the
// fp32 FMA chain only by the order / rounding of the accumulation (well inside the stated 2e-3 tolerance on T).
// A wave owns 32 output bits and keeps their 3 x 9 B-operands (W terms x K steps) in registers for its whole life, then
// walks 32-keypoint tiles of R: 9 sixteen-byte loads per lane, 27 MFMAs, sign test by ballot, packed bits out -- no T
// matrix in HBM, no separate binarize pass (cuda_hash_sift.cu:414-435).
// Operand layout: lane l holds row / column (l & 31) and the 8 consecutive k of half (l >> 5) of the K step, for A and B
// alike, so whatever order the hardware contracts the 16 k of a step in, A and B elements meet at equal k.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void project_sign_kernel(const uint16_t* __restrict__ Rm, const uint16_t* __restrict__ Wb,
                                                           const int* __restrict__ d_count, int n, int nbits,
                                                           uint8_t* __restrict__ desc, size_t desc_pitch, float* __restrict__ dbg_T)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[2][32][4];     // [buffer][row of the tile][wave]: 128 bits per row
    const int count = d_count ? min(*d_count, n) : n;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // the waves that share a row group (one per column tile) run on ONE XCD: they read the same rows of R through one L2
    const int gw = xcd_chunked(blockIdx.x, gridDim.x) * 4 + wid;   // logical global wave index
    const int ntn = nbits >> 5;                                   // 32-bit column tiles (a multiple of 4: a workgroup's four
    const int n0 = (gw % ntn) * 32;                               // waves own 128 adjacent bits of the same row group)
    const int mgroup = gw / ntn, nmgroups = (gridDim.x * 4) / ntn;
    const int li = lane & 31, lk = lane >> 5;
    constexpr int KS = HS_KB / 16;                                // 9 K steps
    // this wave's weights: b[t][ks] = W_t[n0 + li][16 ks + 8 lk .. + 8]
    bf16x8 b[3][KS];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const uint4* p = reinterpret_cast<const uint4*>(Wb + ((size_t)t * nbits + n0 + li) * HS_KB + 8 * lk);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) b[t][ks] = __builtin_bit_cast(bf16x8, p[2 * ks]);
    }
    const int mtiles = (count + 31) >> 5;
    // the next row tile of R is in flight while this one is multiplied; two waves per SIMD cover the rest of the latency
    auto load_tile = [&](int mt, uint4 (&dst)[KS]) {
        const int arow = max(min(mt * 32 + li, count - 1), 0);
        const uint4* pa = reinterpret_cast<const uint4*>(Rm + (size_t)arow * HS_KB + 8 * lk);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) dst[ks] = pa[2 * ks];
    };
    uint4 nxt[KS];
    load_tile(mgroup, nxt);
    // 16-byte stores of a row's 128 bits need a 16-byte aligned destination
    const bool wide = desc != nullptr && (((reinterpret_cast<uintptr_t>(desc) | desc_pitch) & 15u) == 0);
    int buf = 0;
    for (int mt = mgroup; mt < mtiles; mt += nmgroups, buf ^= 1) {
        const int m0 = mt * 32;
        bf16x8 a[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) a[ks] = __builtin_bit_cast(bf16x8, nxt[ks]);
        if (mt + nmgroups < mtiles) load_tile(mt + nmgroups, nxt);
        f32x16 acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        // smallest terms first: the fp32 accumulator rounds them in before the large ones arrive
#pragma unroll
        for (int t = 2; t >= 0; t--)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[t][ks], acc, 0, 0, 0);
        // C layout 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  The sign bits of a row are a
        // ballot half; lane `row` collects its row's 32 bits (bit j -> byte j/8, bit 7 - j%8: MSB first, hash_sift.cpp:367-374)
        uint32_t mine = 0u;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2);
            const unsigned long long ba = __ballot(acc[r] > 0.f);
            if (dbg_T) {
                const int i = m0 + row + 4 * lk;
                if (i < count) dbg_T[(size_t)i * nbits + n0 + li] = acc[r];
            }
            const uint32_t w_lo = __builtin_bswap32(__brev((unsigned)ba)), w_hi = __builtin_bswap32(__brev((unsigned)(ba >> 32)));
            mine = lane == row ? w_lo : mine;
            mine = lane == row + 4 ? w_hi : mine;
        }
        if (desc != nullptr) {
            if (wide) {
                // the workgroup's four waves hold 128 adjacent bits of the same 32 rows: one 16-byte store per row
                if (lane < 32) s_bits[buf][lane][wid] = mine;
                __syncthreads();
                if (threadIdx.x < 32 && m0 + (int)threadIdx.x < count) {
                    const uint4 v = *reinterpret_cast<const uint4*>(&s_bits[buf][threadIdx.x][0]);
                    *reinterpret_cast<uint4*>(desc + (size_t)(m0 + threadIdx.x) * desc_pitch + (n0 & ~127) / 8) = v;
                }
            } else if (lane < 32 && m0 + lane < count) {
                *reinterpret_cast<unsigned*>(desc + (size_t)(m0 + lane) * desc_pitch + n0 / 8) = mine;
            }
        }
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
    const int S = hs_smax_for(max_size, a.scale_factor);
    size_t lds = 0;
    if (a.blur) {
        const BlurGeom bg(S);
        lds = bg.raw_bytes() + bg.hb_bytes() + (size_t)S * S;
        lds = (lds + 15) & ~(size_t)15;
        if (lds > 140 * 1024) return hipErrorInvalidValue;
    }
    float t[7];
    efx_gaussian_taps_host(t);
    const int dbg = a.dbg_hs;
    const float* mag = h.W + (size_t)h.nbits * HS_KPAD;    // the 30x30 weight table is stored behind W
    const float* lut = mag + 900;                          // followed by the 511x511 orientation-bin table
    if (a.blur && S == 48 && a.uniform_size) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sift_kernel<true, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((patch_sift_kernel<true, 48>), dim3(a.n), dim3(HS_NT), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0,
                           a.pyramid, a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, mag, lut,
                           t[0], t[1], t[2], t[3], h.responses, h.dbg_responses, dbg);
    } else if (a.blur) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sift_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((patch_sift_kernel<true, 0>), dim3(a.n), dim3(HS_NT), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0,
                           a.pyramid, a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, mag, lut,
                           t[0], t[1], t[2], t[3], h.responses, h.dbg_responses, dbg);
    } else {
        hipLaunchKernelGGL((patch_sift_kernel<false, 0>), dim3(a.n), dim3(HS_NT), 0, stream, a.img0, a.pitch0, a.rows0, a.cols0,
                           a.pyramid, a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, mag, lut,
                           t[0], t[1], t[2], t[3], h.responses, h.dbg_responses, dbg);
    }
    if (a.desc || h.dbg_T) {
        // persistent waves: 2 per SIMD (the weights occupy 108 VGPRs), each owning one 32-bit column tile
        const int ntn = h.nbits / 32;
        int nblk = 512 / ntn * ntn;                            // 2048 waves on 1024 SIMDs, a multiple of the column tiles
        const int need = (((a.n + 31) / 32) * ntn + 3) / 4;     // never more waves than (row tile, column tile) pairs
        if (nblk > need) nblk = (need + ntn - 1) / ntn * ntn;
        hipLaunchKernelGGL(project_sign_kernel, dim3(nblk), dim3(256), 0, stream,
                           h.responses, h.Wb, a.d_count, a.n, h.nbits, a.desc, a.desc_pitch, h.dbg_T);
    }
    return hipGetLastError();
}
