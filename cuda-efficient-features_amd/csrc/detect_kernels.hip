// detect_kernels.hip -- pyramid + FAST-9 + Harris + radius-NMS + top-N + IC-angle for gfx950 (wave64).
//
// What the kernels compute follows the reference's CUDA detector
//   modules/cuda_efficient_features/src/cuda_fast.cu:33-222            (FAST-9 segment test)
//   modules/cuda_efficient_features/src/cuda_efficient_features.cu:62-248 (NMS predicate, Harris, IC angle, scaling)
//   modules/cuda_efficient_features/src/cuda_efficient_features.cpp:136-321 (pyramid, quotas, border, flow)
// with the spec decisions of DESIGN.md (canonical order, deterministic cap / ties, integer Harris sums,
// shared atan2).  How it is computed is MI355X-first: the pyramid by waves that walk down strips of columns (several levels per
// launch), FAST-9 on LDS tiles of ALL levels in one launch, Harris a lane per corner, the suppression a wave per tile on per-cell
// maxima; all counts stay on the device (no host synchronisation anywhere, the reference does 16 per frame) and NOTHING is
// allocated there -- corners travel in fixed per-tile slots and lie at canonical ranks in arrays of the reference's own 10 % cap
// (round 6) --; compaction is ballot / bitmap based and deterministic; every kernel takes the tiles of several same-sized frames in
// one launch (frame = blockIdx.y).
//
// Compile with -ffp-contract=off: the float expressions below must not be fused.

#include <atomic>
#include "efx_device.h"
#include "bad_affine.h"
#include <algorithm>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
// INVESTIGATION (EFX_DEBUG_BUILD builds only): EFX_TRACE=1 names every launch on stderr, waits for it
// and calls a digest hook (efx_api.cpp) -- how the "16 processes share the GPU" discrepancy was traced to stores of one XCD
// missing from memory after fast_kernel's first run on freshly allocated buffers
#ifdef EFX_DEBUG_BUILD
static const bool g_trace = getenv("EFX_TRACE") != nullptr;
void (*efx_trace_hook)(const DetectLaunch&, const char*) = nullptr;
#define EFX_TRACE_POINT(name) do { if (g_trace) { hipError_t te_ = hipStreamSynchronize(stream); fprintf(stderr, "efx-trace done: %s (%s)\n", name, hipGetErrorString(te_)); fflush(stderr); if (efx_trace_hook) efx_trace_hook(a, name); } } while (0)
#else
#define EFX_TRACE_POINT(name) do { } while (0)
#endif

namespace {

__device__ __forceinline__ int lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int v) { return efx_wave_incl_scan(v); }

// exclusive scan over a block of NW waves; `scratch` holds NW+1 ints of LDS. Returns the exclusive prefix,
// *total receives the block sum. Contains two barriers.
template <int NW>
__device__ __forceinline__ int block_excl_scan(int v, int* scratch, int* total)
{
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63) scratch[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        int w = lane < NW ? scratch[lane] : 0;
        const int wi = wave_incl_scan(w);
        if (lane < NW) scratch[lane] = wi - w;     // exclusive prefix of the wave totals
        if (lane == NW - 1) scratch[NW] = wi;
    }
    __syncthreads();
    const int r = scratch[wid] + incl - v;
    *total = scratch[NW];
    __syncthreads();                               // scratch may be reused by the caller
    return r;
}

// The same for four waves with ONE barrier: every wave reads the four wave totals (one 16-byte LDS read) and adds up the
// ones before it (wave-uniform selects).  `scratch`: four ints, 16-byte aligned, not reused by the caller before its
// next barrier.
__device__ __forceinline__ int block_excl_scan4(int v, int* scratch, int* total)
{
    const int lane = lane_id();
    const int wid = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63) scratch[wid] = incl;
    __syncthreads();
    const int4 t = *reinterpret_cast<const int4*>(scratch);
    *total = (t.x + t.y) + (t.z + t.w);
    const int before = (wid > 0 ? t.x : 0) + (wid > 1 ? t.y : 0) + (wid > 2 ? t.z : 0);
    return before + incl - v;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// FAST-9 segment test on an LDS tile (cuda_fast.cu:33-222).  c points at the centre pixel, P = LDS pitch.
// Circle order as cuda_fast.cu:179-207: k=0 at (0,+3) walking towards +x.
// ------------------------------------------------------------------------------------------------
// The test is only ever run on pixels that passed the compass quick test of fast_kernel: the compass points say which
// polarity can succeed, so only ONE 16-bit ring mask is built (dark pixels are mirrored, x -> 255 - x, so both
// polarities share one instruction stream); the other polarity is tested in a second pass only for the rare pixels
// whose compass points allow both.
template <int P>
__device__ __forceinline__ bool fast9_survivor_lds(const uint8_t* c, int t)
{
    const int p = c[0];
    const int v[16] = { c[3 * P],    c[3 * P + 1],  c[2 * P + 2],  c[P + 3],
                        c[3],        c[-P + 3],     c[-2 * P + 2], c[-3 * P + 1],
                        c[-3 * P],   c[-3 * P - 1], c[-2 * P - 2], c[-P - 3],
                        c[-3],       c[P - 3],      c[2 * P - 2],  c[3 * P - 1] };
    const bool bp = min(max(v[0], v[8]), max(v[4], v[12])) > p + t;     // two neighbouring compass points brighter
    const bool dp = max(min(v[0], v[8]), min(v[4], v[12])) < p - t;     // ... darker
    // Ring pixels k and k + 8 share a register (packed 16-bit lanes); s * v + c < 0 -- one v_pk_mad_i16 -- is the test of
    // both: brighter (s = -1, c = p + t: v > p + t) or darker (s = +1, c = t - p: v < p - t).  The sign bits land at bit k /
    // 16 + k of the mask (round 3: ~35 instead of ~60 instructions per polarity).
    i16x2 pv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) pv[k] = __builtin_bit_cast(i16x2, (uint32_t)v[k] | ((uint32_t)v[k + 8] << 16));
    auto arc9 = [&](bool dark) -> bool {
        const short sg = dark ? (short)1 : (short)-1, cc = dark ? (short)(t - p) : (short)(p + t);
        const i16x2 s2 = { sg, sg }, c2 = { cc, cc };
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t d = __builtin_bit_cast(uint32_t, (i16x2)(pv[k] * s2 + c2));
            m |= (d >> (15 - k)) & (0x00010001u << k);
        }
        m = (m & 0xffu) | ((m >> 8) & 0xff00u);                            // ring order, k = 0 at bit 0
        m |= m << 16;
        m &= m >> 1; m &= m >> 2; m &= m >> 4; m &= m >> 1;              // runs of >= 9
        return (m & 0xffffu) != 0;
    };
    bool res = arc9(!bp);
    const bool again = bp && dp && !res;
    if (__builtin_amdgcn_ballot_w64(again) != 0ull) {
        if (again) res = arc9(true);
    }
    return res;
}

// Harris response, spec S4 (cuda_efficient_features.cu:99-139): exact int32 sums of the 49 Sobel products,
// then one fixed, uncontracted float formula.  The 9x9 footprint is pulled as 9 rows x 3 aligned dwords and
// re-aligned with v_alignbyte (27 LDS reads instead of 81 byte reads); the Sobel sums share the pairwise row /
// column sums, and the products use 24-bit multiplies (|d| <= 1020).
__device__ __forceinline__ float harris_from_sums(int sxx, int sxy, int syy)
{
    const float SCALE = 1.f / (float)(4 * 7 * 255);
    const float K = SCALE * SCALE;
    const float a = (float)sxx * K, b = (float)syy * K, cc = (float)sxy * K;
    const float det = a * b - cc * cc;
    const float tr = a + b;
    return det - 0.04f * tr * tr;
}

// same sums from byte loads (level-0 images whose base or pitch is not 4-byte aligned)
__device__ __forceinline__ float harris_bytes(const uint8_t* c, int P)
{
    int sxx = 0, sxy = 0, syy = 0;
    for (int iy = -3; iy <= 3; iy++)
        for (int ix = -3; ix <= 3; ix++) {
            const uint8_t* q = c + iy * P + ix;
            const int dx = ((int)q[-P + 1] + 2 * (int)q[1] + (int)q[P + 1]) - ((int)q[-P - 1] + 2 * (int)q[-1] + (int)q[P - 1]);
            const int dy = ((int)q[P - 1] + 2 * (int)q[P] + (int)q[P + 1]) - ((int)q[-P - 1] + 2 * (int)q[-P] + (int)q[-P + 1]);
            sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
        }
    return harris_from_sums(sxx, sxy, syy);
}


// Harris response of the 7x7 window around a corner (calcResponse, cuda_efficient_features.cu:99-139).
// p0 = top-left byte of the 9x9 footprint, row pitch P bytes; both 4-byte aligned up to the offset sh.
__device__ __forceinline__ float harris_rows(const uint8_t* p0, int P)
{
    // 9x9 footprint.  All intermediate sums fit 16 bits (|dx|, |dy| <= 1020), so the Sobel arithmetic runs on
    // packed pairs of neighbouring columns (v_pk_add_u16 / v_pk_sub_i16) and the three moment sums are
    // v_dot2_i32_i16 accumulations: exact integers, same values as the scalar form.
    const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(p0) & 3u);
    const uint8_t* wb = p0 - sh;
    // Row by row (rolling, to keep the live register set small): E[j] = columns (2j, 2j+1) for j < 4,
    // E[4] = (8, 7); O[j] = columns (2j+1, 2j+2); H[j] = horizontal 1-2-1 sums at columns (2j+1, 2j+2);
    // D[j] = E[j+1] - E[j] = horizontal differences p[x+1] - p[x-1] at the same columns (round 3: the vertical 1-2-1 of
    // the differences instead of the difference of vertical 1-2-1 sums: 12 instead of 14 instructions per row).
    u16x2 Hm2[4], Hm1[4], Dm2[4], Dm1[4];      // H(row-2), H(row-1), D(row-2), D(row-1)
    const u16x2 two2 = { 2, 2 };
    int sxx = 0, sxy = 0, syy = 0;
#pragma unroll
    for (int row = 0; row < 9; row++) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(wb + (size_t)row * P);
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        const uint32_t a = __builtin_amdgcn_alignbyte(w1, w0, sh), b = __builtin_amdgcn_alignbyte(w2, w1, sh);
        const uint32_t t = w2 >> (8 * sh);
        u16x2 E[5], O[4], H[4], D[4];
        E[0] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0c010c00u));
        E[1] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0c030c02u));
        E[2] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0c050c04u));
        E[3] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0c070c06u));
        E[4] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(t, b, 0x0c030c04u));
        O[0] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0c020c01u));
        O[1] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0c040c03u));
        O[2] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0c060c05u));
        O[3] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(t, b, 0x0c040c03u));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // H = 2 * O + (E[j] + E[j+1]) as one v_pk_mad_u16 (the compiler turns the doubling into a separate shift)
            // two sums of bytes per register, neither above 510: ONE 32-bit add (full rate) does both (v_pk_add_u16: half rate)
            const u16x2 e = __builtin_bit_cast(u16x2, __builtin_bit_cast(uint32_t, E[j]) + __builtin_bit_cast(uint32_t, E[j + 1]));
            asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(H[j]) : "v"(O[j]), "v"(two2), "v"(e));
            D[j] = E[j + 1] - E[j];                               // the high half of D[3] is p7 - p7 = 0: ix = 7 does not exist
        }
        if (row >= 2) {
            // output row r = row - 1: dx = D(r-1) + 2 D(r) + D(r+1), dy = H(r+1) - H(r-1)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u16x2 dsum = Dm2[j] + D[j];
                u16x2 dxu;
                asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(dxu) : "v"(Dm1[j]), "v"(two2), "v"(dsum));
                const i16x2 dx = __builtin_bit_cast(i16x2, dxu);
                uint32_t dyu = __builtin_bit_cast(uint32_t, (u16x2)(H[j] - Hm2[j]));
                if (j == 3) dyu &= 0xffffu;                      // ix = 7 does not exist
                const i16x2 dy = __builtin_bit_cast(i16x2, dyu);
                sxx = __builtin_amdgcn_sdot2(dx, dx, sxx, false);
                sxy = __builtin_amdgcn_sdot2(dx, dy, sxy, false);
                syy = __builtin_amdgcn_sdot2(dy, dy, syy, false);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { Hm2[j] = Hm1[j]; Hm1[j] = H[j]; Dm2[j] = Dm1[j]; Dm1[j] = D[j]; }
    }
    return harris_from_sums(sxx, sxy, syy);
}

// The frame's counters are zeroed by workgroup 0 of the frame's first kernel (a memset command of its own costs ~5 us
// on the critical path of every frame): the small per-frame struct and the per-row sums.
__device__ __forceinline__ void efx_zero_counters(Counters* __restrict__ c, RowCtr* __restrict__ rows, int nrows, int tid, int nthreads)
{
    int* w = reinterpret_cast<int*>(c);
    for (int i = tid; i < (int)(sizeof(Counters) / sizeof(int)); i += nthreads) w[i] = 0;
    for (int i = tid; i < nrows; i += nthreads) { rows[i].cand = 0; rows[i].surv = 0; }
}

// Frame of a batched pyramid launch (blockIdx.y): the source level is the caller's image of that frame (src_is_img0) or the
// frame's copy of a pyramid level; destinations are pyramid levels.
struct FramePyr { FrameIn in; size_t stride; int src_is_img0; RowCtr* zrows; size_t rows_stride; };
// ... of a batched detector launch: the caller's images and the distances between the frames' copies of the context's buffers
struct FrameSet { FrameIn in; FrameStride fs; };

// Four horizontally adjacent outputs of the bilinear resize (spec S5), the arithmetic all three resize kernels share.
// ra / rb: the two source rows in LDS, lc[k]: the LDS column of output k's left source pixel (its right neighbour is the
// next byte), wa / wb: the x weights, wy0 / wy1: the y weights.  Per output the expression of the reference, term by term, as
// nvcc's default contraction builds it (spec S5, round 4: the weight product is a rounded multiply, pixel x weight is fused
// into the accumulator -- explicit fmaf, the translation unit itself is compiled with -ffp-contract=off).
// Measured and dropped (round 3, tools/microbench/bench_ab.sh): two outputs per instruction on packed fp32 lanes
// (v_pk_mul_f32 / v_pk_add_f32: 42 instead of 64 VALU instructions per four outputs) -- kernel 11.10 vs 10.97 us, bench
// line 106.6 vs 106.8 Mkeypoints/s; a pixel pair as ONE unaligned 16-bit LDS read -- 40 us instead of 11 (unaligned
// ds_read_u16 is a slow path).
__device__ __forceinline__ uint32_t resize_quad(const uint8_t* ra, const uint8_t* rb, const int (&lc)[4], const float (&wa)[4], const float (&wb)[4],
                                                float wy0, float wy1)
{
    uint32_t packed = 0;
    // the right neighbour is read through an offset the compiler cannot see: it would merge the two byte reads into ONE
    // unaligned ds_read_u16, which is a slow path of the LDS (measured: 40 us instead of 11 for a level)
    int one = 1;
    asm volatile("" : "+v"(one));
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint8_t* pa = ra + lc[k];
        const uint8_t* pb = rb + lc[k];
        float out = (float)pa[0] * (wa[k] * wy0);                    // == fma(p, w, 0.f) exactly
        out = __builtin_fmaf((float)pa[one], wb[k] * wy0, out);
        out = __builtin_fmaf((float)pb[0], wa[k] * wy1, out);
        out = __builtin_fmaf((float)pb[one], wb[k] * wy1, out);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(out, k, packed);     // rint (half even) + saturate + pack
    }
    return packed;
}

// The same four outputs with a quarter of the LDS instructions (round 3): the source pixels of outputs 0, 1 lie within 8
// bytes of the dword that holds the first of them, those of outputs 2, 3 likewise (scale factors up to 4), so a row costs two
// ds_read2_b32 instead of eight ds_read_u8, and one v_perm_b32 per window lines the four bytes up (left / right pixel of
// output k, left / right of output k + 1) for v_cvt_f32_ubyte0..3.  The windows' offsets and byte selectors do not depend on
// the row: resize_windows() once per lane and tile.  ra / rb must be 4-byte aligned (LDS row pitches are multiples of 4).
struct ResizeWin { int offA, offB; uint32_t selA, selB; bool ok; };
__device__ __forceinline__ ResizeWin resize_windows(const int (&lc)[4])
{
    ResizeWin w;
    w.offA = lc[0] & ~3; w.offB = lc[2] & ~3;
    const int a0 = lc[0] - w.offA, a1 = lc[1] - w.offA, b0 = lc[2] - w.offB, b1 = lc[3] - w.offB;
    w.selA = (uint32_t)a0 | ((uint32_t)(a0 + 1) << 8) | ((uint32_t)a1 << 16) | ((uint32_t)(a1 + 1) << 24);
    w.selB = (uint32_t)b0 | ((uint32_t)(b0 + 1) << 8) | ((uint32_t)b1 << 16) | ((uint32_t)(b1 + 1) << 24);
    w.ok = a1 >= a0 && a1 + 1 <= 7 && b1 >= b0 && b1 + 1 <= 7;
    return w;
}
__device__ __forceinline__ uint32_t resize_quad_win(const uint8_t* ra, const uint8_t* rb, const ResizeWin& w, const float (&wa)[4], const float (&wb)[4],
                                                    float wy0, float wy1)
{
#ifdef EFX_RESIZE_NO_WIN                                     // INVESTIGATION builds (bench_ab.sh): the byte reads
    {
        const int a0 = (int)(w.selA & 7u), a1 = (int)((w.selA >> 16) & 7u), b0 = (int)(w.selB & 7u), b1 = (int)((w.selB >> 16) & 7u);
        const int lc[4] = { w.offA + a0, w.offA + a1, w.offB + b0, w.offB + b1 };
        return resize_quad(ra, rb, lc, wa, wb, wy0, wy1);
    }
#endif
    const uint32_t* qa = reinterpret_cast<const uint32_t*>(ra + w.offA);
    const uint32_t* qb = reinterpret_cast<const uint32_t*>(rb + w.offA);
    const uint32_t* ta = reinterpret_cast<const uint32_t*>(ra + w.offB);
    const uint32_t* tb = reinterpret_cast<const uint32_t*>(rb + w.offB);
    const uint32_t a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1], c0 = ta[0], c1 = ta[1], d0 = tb[0], d1 = tb[1];
    // bytes of pr[h][0]: left, right pixel of output 2h, left, right of output 2h + 1, in source row a; pr[h][1]: row b
    const uint32_t pr[2][2] = { { __builtin_amdgcn_perm(a1, a0, w.selA), __builtin_amdgcn_perm(b1, b0, w.selA) },
                                { __builtin_amdgcn_perm(c1, c0, w.selB), __builtin_amdgcn_perm(d1, d0, w.selB) } };
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t pa = pr[k >> 1][0] >> (16 * (k & 1)), pb = pr[k >> 1][1] >> (16 * (k & 1));
        float out = (float)(pa & 0xffu) * (wa[k] * wy0);              // == fma(p, w, 0.f) exactly
        out = __builtin_fmaf((float)((pa >> 8) & 0xffu), wb[k] * wy0, out);
        out = __builtin_fmaf((float)(pb & 0xffu), wa[k] * wy1, out);
        out = __builtin_fmaf((float)((pb >> 8) & 0xffu), wb[k] * wy1, out);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(out, k, packed);     // rint (half even) + saturate + pack
    }
    return packed;
}

// ================================================================================================
// Kernel R: one 64x64 tile of pyramid level s+1 per workgroup, bilinear from level s (spec S5; the
// cv::cuda::resize call of calcImagePyramid, cuda_efficient_features.cpp:154).  The source footprint of the
// tile is staged in LDS with aligned dword loads.  A lane produces 4 horizontally adjacent outputs (their
// source columns and x-weights are row-invariant and stay in registers), 16 lanes cover a row, the 256
// threads cover 16 rows per pass; v_cvt_pk_u8_f32 rounds (half to even), saturates and packs, so a lane
// stores one dword per row.
// ================================================================================================
template <int NT>
__global__ __launch_bounds__(NT) void resize_kernel(
    const uint8_t* __restrict__ src, int spitch, int rows, int cols, int aligned,
    uint8_t* __restrict__ dst, int dpitch, int drows, int dcols, float fx, float fy, int tiles_x, int tiles_y, int lpitch, int ytab_off,
    Counters* __restrict__ zero, int zero_levels, const FramePyr fp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    src = fp.src_is_img0 ? fp.in.img0[blockIdx.y] : src + blockIdx.y * fp.stride;
    dst += blockIdx.y * fp.stride;
    if (zero && blockIdx.x == 0) efx_zero_counters(zero + blockIdx.y, fp.zrows + blockIdx.y * fp.rows_stride, zero_levels, tid, NT);
    const int tile = xcd_chunked(blockIdx.x, tiles_x * tiles_y);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int ox0 = tx * EFX_TILE, oy0 = ty * EFX_TILE;
    const int ox1 = min(ox0 + EFX_TILE, dcols), oy1 = min(oy0 + EFX_TILE, drows);
    // source footprint [sx0, sx1] x [sy0, sy1] (inclusive, after the +1 neighbour and the clamps)
    const int sx0 = min((int)floorf((float)ox0 * fx), cols - 1), sy0 = min((int)floorf((float)oy0 * fy), rows - 1);
    const int sx1 = min(min((int)floorf((float)(ox1 - 1) * fx), cols - 1) + 1, cols - 1);
    const int sy1 = min(min((int)floorf((float)(oy1 - 1) * fy), rows - 1) + 1, rows - 1);
    const int ax0 = sx0 & ~3;                              // LDS column 0 <-> source column ax0
    const int ndw = ((sx1 - ax0) >> 2) + 1, nrow = sy1 - sy0 + 1;
    {   // 32 lanes per source row (ndw <= 32 for scale factors up to ~1.9), 8 rows per pass; wider rows loop.
        // 4-byte aligned sources go through a raw buffer resource (no per-load bounds tests): every dword that holds a
        // pixel of the footprint lies inside [0, (rows-1)*pitch + roundup4(cols)); bytes beyond `cols` are never used.
        const int j0 = tid & 31, r0 = tid >> 5;
        if (aligned) {
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (rows - 1) * spitch + ((cols + 3) & ~3), 0x00020000);
            for (int j = j0; j < ndw; j += 32)
                for (int r = r0; r < nrow; r += NT / 32)
                    *reinterpret_cast<uint32_t*>(smem + r * lpitch + 4 * j) =
                        __builtin_amdgcn_raw_buffer_load_b32(rsrc, (sy0 + r) * spitch + ax0 + 4 * j, 0, 0);
        } else {
            for (int j = j0; j < ndw; j += 32) {
                const int gx = ax0 + 4 * j;
                for (int r = r0; r < nrow; r += NT / 32) {
                    const uint8_t* p = src + (size_t)(sy0 + r) * spitch;
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) if (gx + b < cols) v |= (uint32_t)p[gx + b] << (8 * b);
                    *reinterpret_cast<uint32_t*>(smem + r * lpitch + 4 * j) = v;
                }
            }
        }
        // per output row of the tile: LDS offsets of its two source rows and the y weights (the same for all lanes of
        // a row, so computed once per workgroup instead of once per lane and row)
        int4* ytab = reinterpret_cast<int4*>(smem + ytab_off);
        if (tid < EFX_TILE) {
            const int oy = min(oy0 + tid, drows - 1);
            const float sy = (float)oy * fy;
            int y1 = (int)floorf(sy);
            if (y1 > rows - 1) y1 = rows - 1;
            const int y2 = y1 + 1;
            const int y2r = y2 < rows - 1 ? y2 : rows - 1;
            ytab[tid] = make_int4((y1 - sy0) * lpitch, (y2r - sy0) * lpitch, __float_as_int(efx_s5_w_hi(oy, fy, sy, y2)), __float_as_int(efx_s5_w_lo(oy, fy, sy, y1)));
        }
    }
    __syncthreads();
    // The +1 neighbour of the last source column is that column itself (spec S5 clamp): tiles that touch it get one
    // replicated byte column, so the inner loop always reads the pair (x1, x1 + 1) from one base address.
    if (sx1 == cols - 1) {
        for (int r = tid; r < nrow; r += NT) smem[r * lpitch + (cols - ax0)] = smem[r * lpitch + (cols - 1 - ax0)];
        __syncthreads();
    }
    const int cq = tid & 15, rq = tid >> 4;               // 16 lanes x 4 outputs per row, 16 rows per pass
    const int oxq = ox0 + 4 * cq;
    if (oxq >= ox1) return;
    float wx0[4], wx1[4]; int lc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ox = min(oxq + k, dcols - 1);
        const float sx = (float)ox * fx;
        int x1 = (int)floorf(sx);
        if (x1 > cols - 1) x1 = cols - 1;
        const int x2 = x1 + 1;
        wx0[k] = efx_s5_w_hi(ox, fx, sx, x2); wx1[k] = efx_s5_w_lo(ox, fx, sx, x1);
        lc[k] = x1 - ax0;                                   // the clamped +1 neighbour is the next LDS byte
    }
    const bool full4 = oxq + 4 <= ox1 && ((((uintptr_t)dst) | (uintptr_t)dpitch) & 3u) == 0;
    const ResizeWin win = resize_windows(lc);
    const bool use_win = __ballot(!win.ok) == 0ull;
    const int4* ytab = reinterpret_cast<const int4*>(smem + ytab_off);
    for (int oy = oy0 + rq; oy < oy1; oy += NT / 16) {
        const int4 yt = ytab[oy - oy0];
        const float wy0 = __int_as_float(yt.z), wy1 = __int_as_float(yt.w);
        const uint8_t* ra = smem + yt.x;
        const uint8_t* rb = smem + yt.y;
        const uint32_t packed = use_win ? resize_quad_win(ra, rb, win, wx0, wx1, wy0, wy1) : resize_quad(ra, rb, lc, wx0, wx1, wy0, wy1);
        uint8_t* d = dst + (size_t)oy * dpitch + oxq;
        if (full4) *reinterpret_cast<uint32_t*>(d) = packed;
        else
            for (int k = 0; k < 4; k++) if (oxq + k < ox1) d[k] = (uint8_t)(packed >> (8 * k));
    }
}

// ================================================================================================
// Kernel R': the same tiles as resize_kernel, STREAMED.  Measured on the 8K level 1: staging the footprint alone costs
// 22 us, the arithmetic alone 21 us, the two in sequence 33 us -- a workgroup spends most of its life waiting for its
// own loads.  Here the grid is only as large as the chip holds at once; a workgroup walks through tiles and issues the
// (range-checked, branch-free) buffer loads of its NEXT tile into registers before it computes the current one from
// LDS, so the memory latency is hidden behind arithmetic.  The barriers in the loop order LDS traffic only
// (__syncthreads() would also wait for the loads in flight).
// The kernel is bound by VALU issue, and a third of its instructions used to be per-tile set-up (tile geometry, the x
// weights of a lane's four columns, the row table: float expressions that do not depend on the pixels).  They now come
// from the resize plan the host builds once per geometry with the same expressions (ResizePlanLevel, efx_api.cpp): a
// scalar load per tile, three 16-byte loads per lane, one per row.
// Requirements checked by the host: aligned source, footprint <= 32 dwords x 80 rows (scale factors up to ~1.22; others
// use resize_kernel).  LDS: 80 rows of RS_LP bytes (a lane stages dword j0 of rows r0 + 8 p unconditionally: lanes and
// rows outside the footprint read beyond the resource or unused pixels and land in padding), then the row table.
// ================================================================================================
__device__ __forceinline__ void efx_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#define RS_LP 132                                           // LDS row pitch: 33 dwords (rows one bank apart)
#define RS_ROWS 80
#define RS_YTAB (RS_LP * RS_ROWS)                           // 10560: 16-byte aligned

__global__ __launch_bounds__(256) void resize_stream_kernel(
    const uint8_t* __restrict__ src, int spitch, int rows, int cols,
    uint8_t* __restrict__ dst, int dpitch, int drows, int dcols, int tiles_x, int tiles_y,
    const int* __restrict__ xtab, int W, const int4* __restrict__ ytab_g, const int4* __restrict__ ttab,
    Counters* __restrict__ zero, int zero_levels, const FramePyr fp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = RS_ROWS / 8;                         // 8 rows per pass
    const int tid = threadIdx.x;
    src = fp.src_is_img0 ? fp.in.img0[blockIdx.y] : src + blockIdx.y * fp.stride;
    dst += blockIdx.y * fp.stride;
    if (zero && blockIdx.x == 0) efx_zero_counters(zero + blockIdx.y, fp.zrows + blockIdx.y * fp.rows_stride, zero_levels, tid, 256);
    // XCD x owns the x-th contiguous eighth of the tiles, its workgroups stride through it
    const int ntiles = tiles_x * tiles_y;
    const int Wg = gridDim.x / EFX_NXCD, xcd = blockIdx.x % EFX_NXCD, wg = blockIdx.x / EFX_NXCD;
    const int cq_ = ntiles / EFX_NXCD, cr_ = ntiles % EFX_NXCD;
    const int c0 = xcd < cr_ ? xcd * (cq_ + 1) : cr_ * (cq_ + 1) + (xcd - cr_) * cq_;
    const int c1 = c0 + cq_ + (xcd < cr_ ? 1 : 0);
    int tile = c0 + wg;
    if (tile >= c1) return;

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (rows - 1) * spitch + ((cols + 3) & ~3), 0x00020000);
    const int j0 = tid & 31, r0 = tid >> 5;
    const int lane_off = r0 * spitch + 4 * j0;              // this lane's dword within a footprint
    uint8_t* stage = smem + r0 * RS_LP + 4 * j0;
    uint32_t pf[NP];
    auto issue = [&](const int4 g) {
        // lanes beyond the footprint's dwords read beyond the resource: the hardware range check returns 0, no branch.  The
        // row step is part of the per-lane offset: the check looks at that offset alone (a scalar offset is added to the
        // address unchecked -- until round 3 the rows below the last image row were read from whatever follows the image)
        const int base = j0 < (g.z & 0xff) ? g.x * spitch + g.y + lane_off : 0x7ffffff0;
#pragma unroll
        for (int p = 0; p < NP; p++) pf[p] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 8 * p * spitch, 0, 0);
    };

    int4 g = ttab[tile];
    issue(g);
    const int cq = tid & 15, rq = tid >> 4;                 // 16 lanes x 4 outputs per row, 16 rows per pass
    const bool dst4 = ((((uintptr_t)dst) | (uintptr_t)dpitch) & 3u) == 0;
    for (;;) {
        const int sy0 = g.x, ax0 = g.y;
        const int ox0 = (g.w & 0xffff) * EFX_TILE, oy0 = (g.w >> 16) * EFX_TILE;
        const int ox1 = min(ox0 + EFX_TILE, dcols), oy1 = min(oy0 + EFX_TILE, drows);
        // ---- the prefetched footprint and the per-row table -> LDS ----
#pragma unroll
        for (int p = 0; p < NP; p++) *reinterpret_cast<uint32_t*>(stage + 8 * p * RS_LP) = pf[p];
        if (tid < EFX_TILE) {
            int4 yt = ytab_g[oy0 + tid];
            yt.x = (yt.x - sy0) * RS_LP; yt.y = (yt.y - sy0) * RS_LP;
            reinterpret_cast<int4*>(smem + RS_YTAB)[tid] = yt;
        }
        // x weights and source columns of this lane's four outputs (the clamped +1 neighbour is the next LDS byte)
        const int oxq = ox0 + 4 * cq;
        const int4 x1 = *reinterpret_cast<const int4*>(xtab + oxq);
        const float4 wx0 = *reinterpret_cast<const float4*>(xtab + W + oxq);
        const float4 wx1 = *reinterpret_cast<const float4*>(xtab + 2 * W + oxq);
        efx_lds_barrier();
        if (g.z >> 16) {                                    // replicated +1 neighbour of the last source column (S5 clamp)
            const int nrow = (g.z >> 8) & 0xff;
            for (int r = tid; r < nrow; r += 256) smem[r * RS_LP + (cols - ax0)] = smem[r * RS_LP + (cols - 1 - ax0)];
            efx_lds_barrier();
        }
        // ---- next tile's loads go out now and land while this tile is computed ----
        const int tnext = tile + Wg;
        const bool more = tnext < c1;
        int4 gn = g;
        if (more) { gn = ttab[tnext]; issue(gn); }

        // ---- this tile: the arithmetic of resize_kernel ----
        if (oxq < ox1) {
            const int lc[4] = { x1.x - ax0, x1.y - ax0, x1.z - ax0, x1.w - ax0 };
            const float wa[4] = { wx0.x, wx0.y, wx0.z, wx0.w }, wb[4] = { wx1.x, wx1.y, wx1.z, wx1.w };
            const bool full4 = oxq + 4 <= ox1 && dst4;
            const ResizeWin win = resize_windows(lc);             // always valid here: the host sends scale factors up to ~1.22 only
            const int4* ytab = reinterpret_cast<const int4*>(smem + RS_YTAB);
            for (int oy = oy0 + rq; oy < oy1; oy += 16) {
                const int4 yt = ytab[oy - oy0];
                const float wy0 = __int_as_float(yt.z), wy1 = __int_as_float(yt.w);
                const uint8_t* ra = smem + yt.x;
                const uint8_t* rb = smem + yt.y;
                const uint32_t packed = resize_quad_win(ra, rb, win, wa, wb, wy0, wy1);
                uint8_t* d = dst + (size_t)oy * dpitch + oxq;
                if (full4) *reinterpret_cast<uint32_t*>(d) = packed;
                else
                    for (int k = 0; k < 4; k++) if (oxq + k < ox1) d[k] = (uint8_t)(packed >> (8 * k));
            }
        }
        if (!more) break;
        efx_lds_barrier();                                  // every read of this tile's LDS is done
        g = gn; tile = tnext;
    }
}

// ================================================================================================
// Kernel R-rows (round 5): SEVERAL pyramid levels per launch for large frames, by waves that walk DOWN the image.
// The tiled kernels above are bound by neither VALU issue (46 % of its ceiling) nor HBM (0.30): a workgroup stages a footprint,
// meets at barriers, runs four short row iterations and starts over, and every level is a launch with its own ramp and tail
// (7 x 10 us at 8K for 170 MB of traffic and ~30 us of issue time).  Here a WAVE owns a strip of `own` columns of level s + 1
// (a lane: four adjacent outputs = one dword of the destination; the lanes behind the owned ones compute halo columns) and a
// chunk of rows, and visits the source rows of level s it needs ONE BY ONE, top to bottom:
//   * source row r arrives by LDS-DMA (buffer_load ... lds: lane j's dword j of the footprint straight into one of the wave's
//     RW_D LDS slots, RW_D rows ahead, no registers, no workgroup barrier anywhere); every lane takes the two 8-byte windows
//     that hold the source pixel pairs of its four outputs (the byte gather of resize_quad_win) and converts them to float
//     ONCE -- a source row serves two destination rows five times out of six at scale 1.2;
//   * when r is the lower source row of a destination row (a bit mask from the host says so) that row is made: 16 rounded
//     weight products + the 4-term FMA chain of spec S5 per lane on full-rate fp32 instructions with VECTOR operands,
//     v_cvt_pk_u8_f32, one range-checked dword store;
//   * that row of level s + 1 also goes to an LDS row of its own, and the same two steps produce level s + 2 from it -- and so on
//     up to NLEV levels (a strip owns the column groups of level k + 1 whose first source column it owns at level k, so every
//     store is a whole dword and the levels are partitioned exactly; the halo columns / one halo row per level and chunk that
//     the next level's +1 neighbours need are computed and not stored).
// No level of a launch but the last is re-read from memory, and the chain is 2 -- 4 launches instead of 7 (efx_api.cpp,
// build_rows_plan: which levels share a launch).  Per pixel the arithmetic is resize_quad_win's, term by term: bit-identical
// levels (tests: the chain_rows mode).
// What the first version taught: the CU's ONE scalar unit is the bound of a loop like this -- with ~60 scalar instructions per
// source row (row / slot bookkeeping, per-row "is a destination row due" compares on v_readlane values, M0 save / restore
// around every LDS-DMA) the first launch took 21 us with loads, stores AND arithmetic compiled out, and neither the
// prefetch depth nor the number of waves changed that.  Hence: everything that can be decided ahead is a table
// (RowsPlanLaunch, spec S5's float expressions): which source rows complete a destination row (one 64-bit mask per level and
// chunk), column / row weights, strips; rows beyond the chunk fall outside the buffer resource (no compares); the loop is
// unrolled by the slot count, so slots and the upper / lower roles of the two converted source rows are static.  What is left
// is VALU issue: ~80 % busy while the waves live (SQ counters), plus ~5 us per launch that no wave sees.  The host checks what
// the kernel relies on (footprint <= 128 dwords, windows valid, <= 64 source rows per chunk, no clamped +1 neighbours: a
// geometry that fails goes through the tiled kernels).
// ================================================================================================
#ifndef RW_DBG
#define RW_DBG 0                                            // investigation builds: 1 no stores, 2 no arithmetic
#endif
#define RW_LDS_A 528                                        // 128 dwords + the reach of a window read
#define RW_LDS_B 272                                        // 64 dwords + the same
typedef __attribute__((address_space(3))) unsigned char efx_lds_uchar;
typedef __attribute__((address_space(3))) float efx_lds_float;

struct RowsLevelArgs { uint8_t* dst; int pitch, rows, cols; const int* x; const int4* y; int W; };
struct RowsArgs {
    const uint8_t* src; int spitch, srows, scols;
    RowsLevelArgs lv[RW_MAXLEV];
    const int4* strips; const int4* chunks;
    int nstrips, ntasks, own;
};

// One dword per lane from a buffer straight into LDS (LDS-DMA: no VGPR destination, so nothing the compiler could copy or
// wait for): lane i's dword lands at lds_dst + 4 i (+ 256 for the second load, whose immediate offset counts on both sides).
// The compiler does not count these loads -- the kernel waits for them itself (rows_wait_vm).  M0 is written in the statement
// that reads it and not restored: nothing else in this kernel uses it (checked in the ISA: tools/isa_dump.sh,
// tests/test_isa_checks.py).  (ADVICE r5 asked for "m0" in the clobber list: this compiler treats M0 as a RESERVED register --
// "clobbering them may lead to undefined behaviour", -Winline-asm -- so the ISA test stays the guard.)
__device__ __forceinline__ void rows_dma2(const __amdgpu_buffer_rsrc_t rsrc, int voff0, int voff1, uint32_t lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %3, 0 offen lds\n\tbuffer_load_dword %1, %3, 0 offen offset:256 lds"
                 : : "v"(voff0), "v"(voff1), "s"(lds_dst), "s"(rsrc) : "memory");
}
template <int N> __device__ __forceinline__ void rows_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "i"(N) : "memory"); }

// the source pixel pairs of a lane's four outputs from an LDS row: the two 8-byte windows (rows_read), then [left, right] x 4 as floats
struct RowsRaw { uint32_t a0, a1, b0, b1; };
__device__ __forceinline__ RowsRaw rows_read(const unsigned char* row, const ResizeWin& w)
{
    const uint32_t* qa = reinterpret_cast<const uint32_t*>(row + w.offA);
    const uint32_t* qb = reinterpret_cast<const uint32_t*>(row + w.offB);
    RowsRaw r; r.a0 = qa[0]; r.a1 = qa[1]; r.b0 = qb[0]; r.b1 = qb[1];
    return r;
}
__device__ __forceinline__ void rows_cvt(const RowsRaw& r, const ResizeWin& w, float (&f)[8])
{
    const uint32_t pa = __builtin_amdgcn_perm(r.a1, r.a0, w.selA), pb = __builtin_amdgcn_perm(r.b1, r.b0, w.selB);
    f[0] = (float)(pa & 0xffu); f[1] = (float)((pa >> 8) & 0xffu); f[2] = (float)((pa >> 16) & 0xffu); f[3] = (float)(pa >> 24);
    f[4] = (float)(pb & 0xffu); f[5] = (float)((pb >> 8) & 0xffu); f[6] = (float)((pb >> 16) & 0xffu); f[7] = (float)(pb >> 24);
    // the row's pixels are in registers at this point (the conversions cannot sink below it): its LDS bytes may be overwritten
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
}
__device__ __forceinline__ void rows_conv(const unsigned char* row, const ResizeWin& w, float (&f)[8])
{
    const uint32_t* qa = reinterpret_cast<const uint32_t*>(row + w.offA);
    const uint32_t* qb = reinterpret_cast<const uint32_t*>(row + w.offB);
    const uint32_t a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1];
    const uint32_t pa = __builtin_amdgcn_perm(a1, a0, w.selA), pb = __builtin_amdgcn_perm(b1, b0, w.selB);
    f[0] = (float)(pa & 0xffu); f[1] = (float)((pa >> 8) & 0xffu); f[2] = (float)((pa >> 16) & 0xffu); f[3] = (float)(pa >> 24);
    f[4] = (float)(pb & 0xffu); f[5] = (float)((pb >> 8) & 0xffu); f[6] = (float)((pb >> 16) & 0xffu); f[7] = (float)(pb >> 24);
    // the row's pixels are in registers at this point (the conversions cannot sink below it): its LDS bytes may be overwritten
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
}

// four outputs from the converted pairs of the upper (fa) and lower (fb) source row: resize_quad_win's expression (spec S5)
__device__ __forceinline__ uint32_t rows_quad(const float (&fa)[8], const float (&fb)[8], const float (&wa)[4], const float (&wb)[4],
                                              float wy0, float wy1)
{
    if (RW_DBG & 2) return __float_as_uint(fa[0] + fb[7] + wy0 + wy1);
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float out = fa[2 * k] * (wa[k] * wy0);                       // == fma(p, w, 0.f) exactly
        out = __builtin_fmaf(fa[2 * k + 1], wb[k] * wy0, out);
        out = __builtin_fmaf(fb[2 * k], wa[k] * wy1, out);
        out = __builtin_fmaf(fb[2 * k + 1], wb[k] * wy1, out);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(out, k, packed);     // rint (half even) + saturate + pack
    }
    return packed;
}

__device__ __forceinline__ void rows_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// per-level state of a wave (every index into these arrays is a compile-time constant after inlining: registers, not memory)
template <int NLEV>
struct RowsState {
    float wa[NLEV][4], wb[NLEV][4];                          // x weights of the lane's four columns
    ResizeWin win[NLEV];                                     // where their source pixel pairs sit in the level's source row
    uint32_t yw[NLEV];                                       // y weights of the chunk's rows: LDS byte address (8 bytes per row) of the row after next
    float wy0[NLEV], wy1[NLEV];                              // ... of the NEXT row of the level: fetched (one broadcast LDS read) when the previous row is made
    int off[NLEV];                                           // store offset (a value the range check drops for lanes that own nothing)
    int pitch[NLEV];                                         // (in VECTOR registers: an SGPR source halves the rate of v_add_u32)
    __amdgpu_buffer_rsrc_t rsrc[NLEV];
    unsigned long long mask[NLEV];
    float g[NLEV][2][8];                                     // levels >= 1: converted upper / lower source row
};

// the y weights of level K's next row: every lane reads the same 8 bytes (a broadcast), a whole row ahead of their use
template <int K, int NLEV>
__device__ __forceinline__ void rows_next_weights(RowsState<NLEV>& S)
{
    const efx_lds_float* wy = (const efx_lds_float*)(uintptr_t)S.yw[K];
    S.wy0[K] = wy[0]; S.wy1[K] = wy[1];
    S.yw[K] += 8;
}

// the row of level K - 1 just made (`packed`, K >= 1) is the next source row of level K.  (Measured and dropped, round 5: the
// levels >= 1 ONE SOURCE ROW BEHIND -- windows requested when a row is made, converted when the next one arrives, as the source
// rows of the first level are handled in the main loop: bit-identical, 18.6 / 13.6 us per launch against 18.3 / 13.1.)
template <int K, int NLEV>
__device__ __forceinline__ void rows_push(RowsState<NLEV>& S, unsigned char* rowbuf, int lane, uint32_t packed, const bool (&due)[NLEV])
{
    if constexpr (K < NLEV) {
        unsigned char* row = rowbuf + (K - 1) * RW_LDS_B;
        *reinterpret_cast<uint32_t*>(row + 4 * lane) = packed;
        rows_lds_order();
#pragma unroll
        for (int q = 0; q < 8; q++) S.g[K][0][q] = S.g[K][1][q];
        rows_conv(row, S.win[K], S.g[K][1]);
        if (due[K]) {
            const uint32_t pk = rows_quad(S.g[K][0], S.g[K][1], S.wa[K], S.wb[K], S.wy0[K], S.wy1[K]);
            rows_next_weights<K>(S);
            if (!(RW_DBG & 1) || pk == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b32(pk, S.rsrc[K], S.off[K], 0, 0);
            S.off[K] += S.pitch[K];
            rows_push<K + 1, NLEV>(S, rowbuf, lane, pk, due);
        }
    }
}

template <int NLEV>
__global__ __launch_bounds__(256) void resize_rows_kernel(const RowsArgs A, Counters* __restrict__ zero, int zero_levels, const FramePyr fp)
{
    constexpr int LDS_WAVE = RW_D * RW_LDS_A + (NLEV - 1) * RW_LDS_B + NLEV * 512;      // source slots | a row of every level but the last | y weights
    static_assert((LDS_WAVE & 15) == 0 && (RW_D & 1) == 0 && NLEV >= 1 && NLEV <= RW_MAXLEV, "LDS rows: 16-byte aligned; an even number of slots");
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[4 * LDS_WAVE];
    if (zero && blockIdx.x == 0) efx_zero_counters(zero + blockIdx.y, fp.zrows + blockIdx.y * fp.rows_stride, zero_levels, threadIdx.x, 256);
    const size_t foff = blockIdx.y * fp.stride;              // this frame's copy of the pyramid
    const uint8_t* const srcA = fp.src_is_img0 ? fp.in.img0[blockIdx.y] : A.src + foff;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = xcd_chunked(blockIdx.x, gridDim.x) * 4 + wave;
    if (task >= A.ntasks) return;
    const int chunk = task / A.nstrips, strip = task - chunk * A.nstrips;
    // (wave-uniform by construction; said so explicitly: the LDS-DMA statements need their descriptor and M0 in scalar registers)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int4* st = A.strips + RW_STRIP_INT4 * strip;
    const int4* ck = A.chunks + RW_CHUNK_INT4 * chunk;
    const int4 st0 = st[0], ck0 = ck[0];
    const int ax0 = uni(st0.x), nd = uni(st0.y);
    const int a_first = uni(ck0.x), a_last = uni(ck0.y), na_pad = uni(ck0.z);
    unsigned char* rowA = s_rows + wave * LDS_WAVE;          // RW_D slots of one source row each
    unsigned char* rowbuf = rowA + RW_D * RW_LDS_A;          // one row of each level but the last

    RowsState<NLEV> S;
#pragma unroll
    for (int k = 0; k < NLEV; k++) {
        const RowsLevelArgs& L = A.lv[k];
        // the lane's four columns: level 0 of the launch by position in the strip, the others by column group
        const int4 sk = k == 0 ? make_int4(0, 0, 0, 0) : st[k];
        const int col0 = k == 0 ? strip * A.own + 4 * lane : 4 * (uni(sk.x) + lane);
        const int nown = k == 0 ? A.own / 4 : uni(sk.y), ncomp = k == 0 ? 64 : uni(sk.z);
        const int org = k == 0 ? ax0 : (k == 1 ? strip * A.own : 4 * uni(st[k - 1].x));      // source column at LDS byte 0 of its source row
        const bool act = lane < ncomp;
        const int4 x1 = *reinterpret_cast<const int4*>(L.x + col0);
        const float4 w0 = *reinterpret_cast<const float4*>(L.x + L.W + col0);
        const float4 w1 = *reinterpret_cast<const float4*>(L.x + 2 * L.W + col0);
        const int lc[4] = { act ? x1.x - org : 0, act ? x1.y - org : 0, act ? x1.z - org : 0, act ? x1.w - org : 0 };
        S.win[k] = resize_windows(lc);
        S.wa[k][0] = w0.x; S.wa[k][1] = w0.y; S.wa[k][2] = w0.z; S.wa[k][3] = w0.w;
        S.wb[k][0] = w1.x; S.wb[k][1] = w1.y; S.wb[k][2] = w1.z; S.wb[k][3] = w1.w;
        const int4 ckk = ck[1 + k];
        const int first = uni(ckk.x), store_end = uni(ckk.y);
        S.rsrc[k] = __builtin_amdgcn_make_buffer_rsrc(L.dst + foff, 0, store_end * L.pitch, 0x00020000);     // rows from store_end on: dropped
        { int pv = L.pitch; asm volatile("" : "+v"(pv)); S.pitch[k] = pv; }
        // lanes that own no column of the level (halo lanes, lanes beyond the last column): an offset the range check drops
        S.off[k] = (lane < nown && col0 < L.cols) ? first * L.pitch + col0 : 0x7ffffff0;
        const int4 yt = L.y[first + lane];                   // row table of this chunk: lane i has row first + i -> the wave's LDS table
        unsigned char* ywk = rowbuf + (NLEV - 1) * RW_LDS_B + k * 512;
        *reinterpret_cast<int2*>(ywk + 8 * lane) = make_int2(yt.z, yt.w);
        { uint32_t a = (uint32_t)(uintptr_t)(efx_lds_uchar*)ywk; asm volatile("" : "+v"(a)); S.yw[k] = a; }      // (a vector register: see pitch)
        const int4 mk = ck[1 + RW_MAXLEV + k / 2];
        S.mask[k] = (k & 1) ? (((unsigned long long)(uint32_t)mk.w << 32) | (uint32_t)mk.z) : (((unsigned long long)(uint32_t)mk.y << 32) | (uint32_t)mk.x);
#pragma unroll
        for (int q = 0; q < 8; q++) { S.g[k][0][q] = 0.f; S.g[k][1][q] = 0.f; }
    }

    rows_lds_order();                                        // the weight tables are in LDS
    { auto first_w = [&](auto kc) { rows_next_weights<decltype(kc)::value>(S); };
      first_w(std::integral_constant<int, 0>());
      if constexpr (NLEV > 1) first_w(std::integral_constant<int, 1>());
      if constexpr (NLEV > 2) first_w(std::integral_constant<int, 2>());
      if constexpr (NLEV > 3) first_w(std::integral_constant<int, 3>()); }

    // ---- source rows: dword j of the footprint by lane j (and j + 64), RW_D rows ahead, straight into the wave's LDS slots ----
    // Branch-free: the resource ends with the chunk's last source row, lanes beyond the footprint carry an offset beyond
    // every resource -- the hardware range check drops those loads (no traffic, zeros land) -- so the number of loads in flight
    // is the same at every row and the wait below is a constant.
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(srcA), 0, a_last * A.spitch + ((A.scols + 3) & ~3), 0x00020000);
    int vo0 = a_first * A.spitch + ax0 + 4 * lane;
    int vo1 = lane + 64 < nd ? vo0 : 0x7ffffff0;            // (the second load's + 256 is its immediate offset)
    int spitch_v = A.spitch;
    asm volatile("" : "+v"(spitch_v));                      // vector register: v_add_u32 with an SGPR source is half rate
    const uint32_t ldsA = (uint32_t)(uintptr_t)(efx_lds_uchar*)s_rows + (uint32_t)wave * LDS_WAVE;
#pragma unroll
    for (int u = 0; u < RW_D; u++) { rows_dma2(rsrcA, vo0, vo1, ldsA + u * RW_LDS_A); vo0 += spitch_v; vo1 += spitch_v; }

    float f0[8], f1[8];                                      // converted source rows of the even / odd slots
#pragma unroll
    for (int i = 0; i < 8; i++) { f0[i] = 0.f; f1[i] = 0.f; }

    // A source row's window reads are issued ONE ITERATION AHEAD of their conversion, so that the LDS round trip runs beside the
    // previous row's arithmetic instead of in front of this row's (round 5, last pass).
    // Source row i has landed in its slot when at most the loads of the rows behind it are in flight (stores issued since count
    // as well: the waits are conservative by the rows they stand for).
    rows_wait_vm<2 * (RW_D - 1)>();
    RowsRaw raw = rows_read(rowA, S.win[0]);
    for (int i0 = 0; i0 < na_pad; i0 += RW_D) {
#pragma unroll
        for (int u = 0; u < RW_D; u++) {
            float (&fcur)[8] = (u & 1) ? f1 : f0;
            float (&fprev)[8] = (u & 1) ? f0 : f1;
            rows_cvt(raw, S.win[0], fcur);                              // source row i0 + u (read in the previous iteration)
            rows_wait_vm<2 * (RW_D - 2)>();                             // row i0 + u + 1 has landed (rows up to i0 + u + RW_D - 1 are issued)
            raw = rows_read(rowA + ((u + 1) % RW_D) * RW_LDS_A, S.win[0]);
            rows_dma2(rsrcA, vo0, vo1, ldsA + u * RW_LDS_A);            // the slot's next row (beyond the chunk: dropped)
            vo0 += spitch_v; vo1 += spitch_v;
            bool due[NLEV];
#pragma unroll
            for (int k = 0; k < NLEV; k++) { due[k] = (S.mask[k] & 1ull) != 0; S.mask[k] >>= 1; }
            if (due[0]) {
                // the row's y weights sit in vector registers (a scalar source halves the rate of v_mul / v_fma), fetched by ONE
                // broadcast LDS read when the previous row was made (first versions: two v_readlane + two v_mov per row and level)
                const uint32_t packed = rows_quad(fprev, fcur, S.wa[0], S.wb[0], S.wy0[0], S.wy1[0]);
                rows_next_weights<0>(S);
                if (!(RW_DBG & 1) || packed == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b32(packed, S.rsrc[0], S.off[0], 0, 0);
                S.off[0] += S.pitch[0];
                rows_push<1, NLEV>(S, rowbuf, lane, packed, due);
            }
        }
    }
}

// ================================================================================================
// Kernel R2: a whole small pyramid in one launch ("tower"; frames up to EFX_TOWER_MAX_PX pixels, see plan_tower).  The
// per-level kernel above is launch- and latency-bound once the levels are small (7 dependent launches cost a third of
// an FHD frame's detectAndCompute).  Here a workgroup owns one
// TT x TT tile of the TOP level and produces, for every level s0+1 .. top, the pixels that tile descends from: the
// region of level s0 it needs is staged in LDS once, level s+1 is computed from level s LDS -> LDS (ping-pong) and
// the pixels the workgroup OWNS are written to the pyramid.  Ownership partitions every level: the owned range of tile
// t at level s starts at the source column of the first owned column at level s+1 (x1 of the resize), so ranges are
// contiguous and disjoint; the few columns / rows beyond it that the +1 neighbour of the next level needs are
// recomputed by both neighbours (same arithmetic, same bits, not written twice).  Per pixel the arithmetic is
// resize_kernel's, so the levels are bit-identical whichever kernel makes them.
// ================================================================================================
struct TowerArgs {
    int s0, top;                 // source level, last level produced
    int tt;                      // tile edge at the top level
    int tiles_x, tiles_y;        // tiles of the top level
    int bufA, bufB;              // byte offsets / sizes: [bufA | bufB | ytab0 | ytab1]
    int ytab_off, ytab_rows;
    int aligned0;                // the source level may be staged with dword loads
};

// source column (row) of destination column o of a level with `n` source columns: resize_kernel's x1
__device__ __forceinline__ int tower_src(int o, float f, int n) { const int v = (int)floorf((float)o * f); return v > n - 1 ? n - 1 : v; }

template <int NT>
__global__ __launch_bounds__(NT) void pyramid_tower_kernel(const LevelTable* __restrict__ T, const uint8_t* __restrict__ img0, int pitch0,
                                                            uint8_t* __restrict__ pyramid, TowerArgs A, Counters* __restrict__ zero, const FramePyr fp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    img0 = fp.in.img0[blockIdx.y];
    pyramid += blockIdx.y * fp.stride;
    if (zero && blockIdx.x == 0) efx_zero_counters(zero + blockIdx.y, fp.zrows + blockIdx.y * fp.rows_stride, T->total_rows, threadIdx.x, NT);
    __shared__ int s_rng[EFX_MAX_LEVELS][8];        // per level: lox, hix, ownlox, ownhix, loy, hiy, ownloy, ownhiy
    const int tid = threadIdx.x;
    const int tile = xcd_chunked(blockIdx.x, A.tiles_x * A.tiles_y);
    const int tx = tile % A.tiles_x, ty = tile / A.tiles_x;

    // ---- ranges, top level down (uniform; two threads: x and y) ----
    if (tid < 2) {
        const bool isx = tid == 0;
        const int t = isx ? tx : ty, nt = isx ? A.tiles_x : A.tiles_y;
        const LevelDev& Lt = T->lv[A.top];
        const int ntop = isx ? Lt.cols : Lt.rows;
        int ownlo = t * A.tt, ownhi = min((t + 1) * A.tt, ntop);      // [ownlo, ownhi)
        int lo = ownlo, hi = ownhi - 1;                                // computed range, inclusive
        int* r = &s_rng[A.top][isx ? 0 : 4];
        r[0] = lo; r[1] = hi; r[2] = ownlo; r[3] = ownhi;
        for (int s = A.top - 1; s >= A.s0; s--) {
            const LevelDev& D = T->lv[s + 1];
            const LevelDev& S = T->lv[s];
            const float f = isx ? D.fx : D.fy;
            const int n = isx ? S.cols : S.rows;
            const int nownlo = t == 0 ? 0 : tower_src(ownlo, f, n);
            const int nownhi = t == nt - 1 ? n : tower_src(ownhi, f, n);
            int nlo = tower_src(lo, f, n);
            if (isx) nlo &= ~3;                                        // columns: dword-aligned region origin
            int nhi = min(tower_src(hi, f, n) + 1, n - 1);
            nhi = max(nhi, nownhi - 1);
            lo = nlo; hi = nhi; ownlo = nownlo; ownhi = nownhi;
            r = &s_rng[s][isx ? 0 : 4];
            r[0] = lo; r[1] = hi; r[2] = ownlo; r[3] = ownhi;
        }
    }
    __syncthreads();

    // every LDS access below is smem + integer offset: pointers picked from an array would decay to FLAT accesses
    auto ytab_off = [&](int which) -> int { return A.ytab_off + which * A.ytab_rows * 16; };
    auto buf_off = [&](int which) -> int { return which ? A.bufB : A.bufA; };

    // per-row source offsets and y weights of the phase that makes level s+1 (same expressions as resize_kernel)
    auto fill_ytab = [&](int s, int yt_off, int spitch_l) {
        int4* yt = reinterpret_cast<int4*>(smem + yt_off);
        const LevelDev& D = T->lv[s + 1];
        const LevelDev& S = T->lv[s];
        const int loy1 = s_rng[s + 1][4], hiy1 = s_rng[s + 1][5], loy0 = s_rng[s][4];
        for (int i = tid; i <= hiy1 - loy1; i += NT) {
            const int oy = loy1 + i;
            const float sy = (float)oy * D.fy;
            int y1 = (int)floorf(sy);
            if (y1 > S.rows - 1) y1 = S.rows - 1;
            const int y2 = y1 + 1;
            const int y2r = y2 < S.rows - 1 ? y2 : S.rows - 1;
            yt[i] = make_int4((y1 - loy0) * spitch_l, (y2r - loy0) * spitch_l, __float_as_int(efx_s5_w_hi(oy, D.fy, sy, y2)), __float_as_int(efx_s5_w_lo(oy, D.fy, sy, y1)));
        }
    };
    // LDS row pitch of a level's region: its columns plus the replicated +1 neighbour of the last one, in dwords
    auto pitch_of = [&](int s) -> int { return (s_rng[s][1] - s_rng[s][0] + 2 + 3) & ~3; };

    // ---- the region of level s0 -> LDS ----
    {
        const LevelDev& S = T->lv[A.s0];
        const uint8_t* src = A.s0 == 0 ? img0 : pyramid + S.img_off;
        const int spitch = A.s0 == 0 ? pitch0 : S.pitch;
        const int lox = s_rng[A.s0][0], hix = s_rng[A.s0][1], loy = s_rng[A.s0][4], hiy = s_rng[A.s0][5];
        const int lp = pitch_of(A.s0);
        const int ndw = ((hix - lox) >> 2) + 1, nrow = hiy - loy + 1;
        uint8_t* dst = smem + buf_off(0);
        if (A.aligned0) {
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (S.rows - 1) * spitch + ((S.cols + 3) & ~3), 0x00020000);
            for (int i = tid; i < ndw * nrow; i += NT) {
                const int r = i / ndw, j = i - r * ndw;
                *reinterpret_cast<uint32_t*>(dst + r * lp + 4 * j) = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (loy + r) * spitch + lox + 4 * j, 0, 0);
            }
        } else {
            for (int i = tid; i < ndw * nrow; i += NT) {
                const int r = i / ndw, j = i - r * ndw;
                const uint8_t* p = src + (size_t)(loy + r) * spitch;
                const int gx = lox + 4 * j;
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) if (gx + b < S.cols) v |= (uint32_t)p[gx + b] << (8 * b);
                *reinterpret_cast<uint32_t*>(dst + r * lp + 4 * j) = v;
            }
        }
        fill_ytab(A.s0, ytab_off(0), lp);
        __syncthreads();
        // the +1 neighbour of the last source column is that column itself (spec S5 clamp)
        if (hix == S.cols - 1) {
            for (int r = tid; r < nrow; r += NT) dst[r * lp + (S.cols - lox)] = dst[r * lp + (S.cols - 1 - lox)];
            __syncthreads();
        }
    }

    // ---- level s+1 from level s, LDS -> LDS (+ the owned pixels -> pyramid) ----
    for (int s = A.s0, ph = 0; s < A.top; s++, ph++) {
        const LevelDev& D = T->lv[s + 1];
        const LevelDev& S = T->lv[s];
        const int sb = buf_off(ph & 1), db = buf_off((ph & 1) ^ 1);
        const int lox0 = s_rng[s][0];
        const int lox1 = s_rng[s + 1][0], hix1 = s_rng[s + 1][1], ownlox = s_rng[s + 1][2], ownhix = s_rng[s + 1][3];
        const int loy1 = s_rng[s + 1][4], hiy1 = s_rng[s + 1][5], ownloy = s_rng[s + 1][6], ownhiy = s_rng[s + 1][7];
        const int lp1 = pitch_of(s + 1);
        const int W = hix1 - lox1 + 2, H = hiy1 - loy1 + 1;         // computed columns incl. the replicated one
        const int ngrp = (W + 3) >> 2;
        // column groups of 4 outputs: the smallest power of two that covers them -> rows per pass
        int sh = 1;
        while ((1 << sh) < ngrp && (2 << sh) <= NT) sh++;
        const int cq = tid & ((1 << sh) - 1), rq = tid >> sh, rstep = NT >> sh;
        const int yt = ytab_off(ph & 1);
        uint8_t* gdst = pyramid + D.img_off;
        const bool keep = s + 1 < A.top;                               // the top level feeds nothing
        for (int g = cq; g < ngrp; g += (1 << sh)) {
            const int oxq = lox1 + 4 * g;
            float wx0[4], wx1[4]; int lc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ox = min(oxq + k, hix1);                      // beyond the range: the last column again (replication)
                const float sx = (float)ox * D.fx;
                int x1 = (int)floorf(sx);
                if (x1 > S.cols - 1) x1 = S.cols - 1;
                const int x2 = x1 + 1;
                wx0[k] = efx_s5_w_hi(ox, D.fx, sx, x2); wx1[k] = efx_s5_w_lo(ox, D.fx, sx, x1);
                lc[k] = x1 - lox0;
            }
            const bool colfull = oxq >= ownlox && oxq + 4 <= ownhix;
            const ResizeWin win = resize_windows(lc);             // always valid: plan_tower refuses scale factors above 3.5
            for (int i = rq; i < H; i += rstep) {
                const int4 t4 = *reinterpret_cast<const int4*>(smem + yt + i * 16);
                const float wy0 = __int_as_float(t4.z), wy1 = __int_as_float(t4.w);
                const uint8_t* ra = smem + sb + t4.x;
                const uint8_t* rb = smem + sb + t4.y;
                const uint32_t packed = resize_quad_win(ra, rb, win, wx0, wx1, wy0, wy1);
                if (keep) *reinterpret_cast<uint32_t*>(smem + db + i * lp1 + 4 * g) = packed;
                const int oy = loy1 + i;
                if (oy >= ownloy && oy < ownhiy) {
                    uint8_t* d = gdst + (size_t)oy * D.pitch + oxq;
                    if (colfull) *reinterpret_cast<uint32_t*>(d) = packed;
                    else
                        for (int k = 0; k < 4; k++) if (oxq + k >= ownlox && oxq + k < ownhix) d[k] = (uint8_t)(packed >> (8 * k));
                }
            }
        }
        if (keep) fill_ytab(s + 1, ytab_off((ph & 1) ^ 1), lp1);
        __syncthreads();
    }
}

// tile + halo -> LDS: 72 rows x 72 bytes as 9 x 8-byte pieces per row (the global address is only 4-byte
// aligned: x0 - 4), LDS row pitch 80 B.
// 4-byte aligned images go through a raw buffer resource: rows above / below the image fall outside the resource's
// byte range and read as 0 in hardware, so the loop carries no bounds checks.  Columns left / right of the image wrap
// into the neighbouring row instead; those halo bytes are never consumed, because every pixel that is tested lies at
// least 15 px inside the frame (createMask, cuda_efficient_features.cpp:176-182) and the tests reach 3 px.
typedef unsigned int efx_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ efx_u32x2 efx_six_bits(efx_u32x2 v)
{
    return efx_u32x2{ (v.x >> 2) & 0x3f3f3f3fu, (v.y >> 2) & 0x3f3f3f3fu };
}

// s_q6 (same layout as s_tile): every byte's upper six bits, (v >> 2) & 0x3f -- what fast_kernel's quick test computes on
template <int NT>
__device__ __forceinline__ void load_tile_lds(uint32_t* s_tile, uint32_t* s_q6, const uint8_t* __restrict__ src, int spitch, int rows, int cols,
                                              bool aligned, int x0, int y0, int tid)
{
    if (aligned) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (rows - 1) * spitch + cols, 0x00020000);
        const int base = (y0 - EFX_HALO) * spitch + x0 - EFX_HALO;
        if (NT == 256) {
            // 9 pieces x 28 rows per pass (252 of the 256 threads), three passes: a thread's piece and row are worked out
            // once, a pass adds 28 rows -- one addition for the global offset, an immediate for the LDS address
            // (round 3: the running index cost a division by 9 and three multiply-adds per piece, 36 instructions per wave
            // and tile; now 12)
            const int r0 = tid / 9, c8 = tid - r0 * 9;
            if (tid < 252) {
                const int goff = base + r0 * spitch + c8 * 8;
                uint8_t* l = reinterpret_cast<uint8_t*>(s_tile) + r0 * EFX_LP + c8 * 8;
                // (the row step goes into the per-lane offset: the hardware range check looks at that offset alone, not at a
                // scalar one added to it, and the halo rows above / below the image rely on the check)
                const efx_u32x2 v0 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, goff, 0, 0);
                const efx_u32x2 v1 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, goff + 28 * spitch, 0, 0);
                const efx_u32x2 v2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, goff + 56 * spitch, 0, 0);
                *reinterpret_cast<efx_u32x2*>(l) = v0;
                *reinterpret_cast<efx_u32x2*>(l + 28 * EFX_LP) = v1;
                if (r0 + 56 < EFX_LT) *reinterpret_cast<efx_u32x2*>(l + 56 * EFX_LP) = v2;
                uint8_t* q = reinterpret_cast<uint8_t*>(s_q6) + r0 * EFX_LP + c8 * 8;
                *reinterpret_cast<efx_u32x2*>(q) = efx_six_bits(v0);
                *reinterpret_cast<efx_u32x2*>(q + 28 * EFX_LP) = efx_six_bits(v1);
                if (r0 + 56 < EFX_LT) *reinterpret_cast<efx_u32x2*>(q + 56 * EFX_LP) = efx_six_bits(v2);
            }
            return;
        }
        for (int i = tid; i < EFX_LT * 9; i += NT) {
            const int r = i / 9, c8 = i - r * 9;
            const efx_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, base + r * spitch + c8 * 8, 0, 0);
            *reinterpret_cast<efx_u32x2*>(reinterpret_cast<uint8_t*>(s_tile) + r * EFX_LP + c8 * 8) = v;
            *reinterpret_cast<efx_u32x2*>(reinterpret_cast<uint8_t*>(s_q6) + r * EFX_LP + c8 * 8) = efx_six_bits(v);
        }
        return;
    }
    // byte path (caller's level-0 image with an unaligned base or pitch): zero outside the image
    for (int i = tid; i < EFX_LT * 9; i += NT) {
        const int r = i / 9, c8 = i - r * 9;
        const int gy = y0 - EFX_HALO + r;
        const int gx = x0 - EFX_HALO + c8 * 8;
        uint2 v = make_uint2(0u, 0u);
        if (gy >= 0 && gy < rows) {
            const uint8_t* p = src + (size_t)gy * spitch;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int xa = gx + b, xb = gx + 4 + b;
                if (xa >= 0 && xa < cols) v.x |= (uint32_t)p[xa] << (8 * b);
                if (xb >= 0 && xb < cols) v.y |= (uint32_t)p[xb] << (8 * b);
            }
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(s_tile) + r * EFX_LP + c8 * 8) = v;
        *reinterpret_cast<efx_u32x2*>(reinterpret_cast<uint8_t*>(s_q6) + r * EFX_LP + c8 * 8) = efx_six_bits(efx_u32x2{ v.x, v.y });
    }
}

// level and tile coordinates of a global tile index: one packed word per tile behind the level table
// (efx_pack_tile, written by the host with the geometry)
__device__ __forceinline__ void efx_tile_of(const LevelTable* T, int gt, int& l, int& tx, int& ty)
{
    const uint32_t v = reinterpret_cast<const uint32_t*>(T + 1)[gt];
    l = (int)(v & ((1u << EFX_TILE_LEVEL_BITS) - 1u)); tx = (int)((v >> EFX_TILE_LEVEL_BITS) & 1023u); ty = (int)(v >> (EFX_TILE_LEVEL_BITS + 10));
}

// ================================================================================================
// Kernel A: FAST-9 for every 64x64 tile of every pyramid level in ONE launch (no per-level tails).
//   tile+halo -> LDS | compass quick test on all pixels (4x4 block per lane, packed 16-bit lanes)
//   | single-polarity segment test on the survivors -> corner bitmap | canonical enumeration (cell-major)
//   | append xy to the level's corner array + tile header (responses: harris_kernel)
// Algorithmic HBM bytes: every level is read once.  Bound by VALU issue.
// ================================================================================================
#ifndef EFX_FAST_QUICK16
#define EFX_FAST_QUICK16 0       // 1: the packed 16-bit quick test of rounds 3 - 5 (A/B builds)
#endif
__global__ __launch_bounds__(256) void fast_kernel(
    const LevelTable* __restrict__ T, const uint8_t* __restrict__ img0, int pitch0, int aligned0,
    const uint8_t* __restrict__ pyramid, int threshold, const uint8_t* __restrict__ mask, int mask_pitch,
    unsigned char* __restrict__ slots, uint16_t* __restrict__ tcount, RowCtr* __restrict__ rowsum, TileHdr* __restrict__ hdr_all, int dbg_arg, const FrameSet F)
{
    const int dbg = EFX_DBG(dbg_arg);
    {   // this frame's buffers (blockIdx.y)
        const size_t f = blockIdx.y;
        img0 = F.in.img0[f]; pyramid += f * F.fs.pyramid; slots += f * F.fs.slots; tcount += f * F.fs.hdr; rowsum += f * F.fs.rows;
        hdr_all += f * F.fs.hdr;
    }
    __shared__ __attribute__((aligned(16))) uint32_t s_tile[EFX_LT * (EFX_LP / 4)];
    __shared__ unsigned long long s_bitmap[EFX_TILE];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[EFX_TILE * EFX_TILE];     // quick-test survivors (block << 5 | bit), read by the full test and by the append
    __shared__ __attribute__((aligned(16))) int s_scan[8];     // two one-barrier scans: [0, 4) and [4, 8)
    __shared__ uint16_t s_rowoff[256];                          // phase 3: corners before row rr of cell c (canonical order), index 16 c + rr
    __shared__ int s_celloff[EFX_CELLS_PER_TILE + 1];

    const int tid = threadIdx.x;
    // heaviest tiles first: the upper pyramid levels have the densest corners, so they must not form the tail
    const int gt = T->total_tiles - 1 - xcd_interleaved(blockIdx.x, T->total_tiles);
    int l, tx, ty;
    efx_tile_of(T, gt, l, tx, ty);
    const LevelDev& L = T->lv[l];
    if (!L.active) return;
    const int tile = gt - L.tile_base;
    const int rows = L.rows, cols = L.cols;
    const uint8_t* src = l == 0 ? img0 : pyramid + L.img_off;
    const int spitch = l == 0 ? pitch0 : L.pitch;
    const bool aligned = l == 0 ? aligned0 != 0 : true;
    const int x0 = tx * EFX_TILE, y0 = ty * EFX_TILE;
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(s_tile);
    TileHdr* hdr = hdr_all + L.tile_base;

    // ---- phase 0: tile + halo -> LDS.  72 rows x 72 bytes as 9 x 8-byte pieces per row (the global address is
    //      only 4-byte aligned: x0 - 4), LDS row pitch 80 B. ----
    if (tid < EFX_TILE) s_bitmap[tid] = 0ull;
    // the six-bit copy of the tile the quick test reads lies where the survivor list will: the list is written behind the barrier
    // of phase 1's scan, i.e. after every wave's last read of the copy
    uint32_t* s_q6 = reinterpret_cast<uint32_t*>(s_list);
    static_assert(sizeof(s_list) >= EFX_LT * EFX_LP, "the six-bit tile copy aliases the survivor list");
    load_tile_lds<256>(s_tile, s_q6, src, spitch, rows, cols, aligned, x0, y0, tid);
    __syncthreads();
    if (dbg & 1) return;

    int total = 0;
    {
        const int lane = tid & 63, wid = tid >> 6;
        // ---- phase 1: quick test on all pixels.  A lane owns a 4x4 pixel block and pulls the 18 dwords it needs (10 rows of
        //      its own dword column, the dwords left and right of its four rows).  A 9-arc always contains two neighbouring
        //      compass points of the same polarity (cuda_fast.cu:193-197 has the weaker opposing-pair form); pixels that
        //      pass go to one list per tile. ----
        const int cg = lane & 15, rg = lane >> 4;
        const int bx = cg * 4, by = wid * 16 + rg * 4;
        unsigned qm = 0;
        {
            const int gx0 = x0 + bx, gy0 = y0 + by;
            // validity of the 16 pixels of the block (border mask, .cpp:176-182) in the layout of qm below: column c of
            // the block in byte c, row j at bit 7 - j of the byte
            unsigned xm = 0xf0f0f0f0u, ym = 0xf0f0f0f0u;
            if (x0 < EFX_HALF_PATCH || x0 + EFX_TILE > cols - EFX_HALF_PATCH || y0 < EFX_HALF_PATCH || y0 + EFX_TILE > rows - EFX_HALF_PATCH) {
                // only tiles that touch the 15-px border build the masks (workgroup-uniform branch)
                xm = 0; ym = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if ((gx0 + i) >= EFX_HALF_PATCH && (gx0 + i) < cols - EFX_HALF_PATCH) xm |= 0xf0u << (8 * i);
                    if ((gy0 + i) >= EFX_HALF_PATCH && (gy0 + i) < rows - EFX_HALF_PATCH) ym |= 0x80808080u >> i;
                }
            }
#if EFX_FAST_QUICK16
            // (the form of rounds 3 - 5, kept for A/B runs: exact compass test, two pixels per packed 16-bit instruction)
            const uint32_t* trow = s_tile + (by + EFX_HALO - 3) * (EFX_LP / 4) + cg;
            uint32_t R[10][3];
#pragma unroll
            for (int i = 0; i < 10; i++) {
                R[i][0] = trow[i * (EFX_LP / 4) + 0]; R[i][1] = trow[i * (EFX_LP / 4) + 1]; R[i][2] = trow[i * (EFX_LP / 4) + 2];
            }
            // Two pixels per instruction on packed 16-bit lanes (v_perm_b32 widens byte pairs, v_pk_max/min_i16).  Two
            // neighbouring compass points brighter than p+t <=> min(max(N,S), max(E,W)) > p+t, and the mirrored form for
            // darker; either one <=> max(bright - p, p - dark) > t, i.e. the SIGN of t - max(..) (all values within +-255).
            // The sign bits of a row's four columns are the top bits of four bytes: one v_perm_b32 gathers them, one
            // shift + one v_and_or_b32 drops them into qm (round 3: 91 instead of 112 instructions per 16 pixels).
            //   C[r][h]: columns (2h, 2h+1) of the block in footprint row r;  E/W: the pixels 3 to the right / left
            const i16x2 thr2 = { (short)threshold, (short)threshold };
            i16x2 C[10][2];
#pragma unroll
            for (int r = 0; r < 10; r++) {
                C[r][0] = __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(0u, R[r][1], 0x0c010c00u));
                C[r][1] = __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(0u, R[r][1], 0x0c030c02u));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t d0 = R[j + 3][0], d1 = R[j + 3][1], d2 = R[j + 3][2];
                const i16x2 E[2] = { __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(d2, d1, 0x0c040c03u)),      // x+3 of columns 0,1
                                     __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(d2, d1, 0x0c060c05u)) };    // x+3 of columns 2,3
                const i16x2 W[2] = { __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(d1, d0, 0x0c020c01u)),      // x-3 of columns 0,1
                                     __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(d1, d0, 0x0c040c03u)) };    // x-3 of columns 2,3
                uint32_t sgn[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const i16x2 p2 = C[j + 3][h], cn = C[j][h], cs = C[j + 6][h];
                    const i16x2 bright = __builtin_elementwise_min(__builtin_elementwise_max(cs, cn), __builtin_elementwise_max(E[h], W[h]));
                    const i16x2 dark = __builtin_elementwise_max(__builtin_elementwise_min(cs, cn), __builtin_elementwise_min(E[h], W[h]));
                    const i16x2 m = __builtin_elementwise_max((i16x2)(bright - p2), (i16x2)(p2 - dark));
                    sgn[h] = __builtin_bit_cast(uint32_t, (i16x2)(thr2 - m));      // bit 15 / 31 set <=> the pixel passes
                }
                // bytes 1, 3 of sgn[0] (columns 0, 1) and of sgn[1] (columns 2, 3) -> bytes 0 .. 3
                const uint32_t f = __builtin_amdgcn_perm(sgn[1], sgn[0], 0x07050301u);
                qm |= (f >> j) & (0x80808080u >> j);
            }
#else
            // FOUR pixels per instruction, on the full-rate 32-bit adds and logic (round 6).  The quick test only has to keep every
            // corner, so it runs on the pixels' upper six bits: with q(v) = v >> 2,  n > p + t  implies  q(n) - q(p) >= tq  for
            // tq = ceil((t - 2) / 4) = (t + 1) >> 2  (q(n) >= (n - 3) / 4, q(p) <= p / 4), and likewise for darker.  Six-bit values
            // leave two spare bits per byte, so a whole dword of four neighbouring pixels is compared with ONE 32-bit add or subtract
            // and no carry ever crosses a byte:  q(n) + (128 - tq - q(p))  lies in [1, 191] and has bit 7 set  <=>  q(n) - q(p) >= tq;
            // (q(p) + 128 - tq) - q(n)  likewise for darker.  Two neighbouring compass points of one polarity: (N | S) & (E | W) on
            // those bits.  The six-bit values come from a second copy of the tile in LDS, converted once where the tile is loaded
            // (load_tile_lds; the copy lies where the survivor list will).  The packed 16-bit form this replaces took 10 half-rate
            // instructions per pixel PAIR plus the perms that widen the bytes (profiles/r06_valu_rate.txt: 4.1 cycles against 2.2);
            // this one takes 17 full-rate ones and two v_alignbyte per FOUR pixels.  It passes what the exact test passes at a
            // threshold of ~t - 2.5, so a few more pixels (+5 %) reach the 16-point test of phase 2, which decides as before: the
            // results are the same bit for bit.
            uint32_t k1 = 0x01010101u * (uint32_t)(128 - min((threshold + 1) >> 2, 64));
            asm volatile("" : "+v"(k1));        // in a VECTOR register: a full-rate instruction with a scalar source runs at half rate
            const uint32_t* qrow = s_q6 + (by + EFX_HALO - 3) * (EFX_LP / 4) + cg;
            uint32_t Q[10];
#pragma unroll
            for (int i = 0; i < 10; i++) Q[i] = qrow[i * (EFX_LP / 4) + 1];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // the pixels 3 to the right / left of the block's four columns in footprint row j + 3
                const uint32_t d0 = qrow[(j + 3) * (EFX_LP / 4) + 0], d2 = qrow[(j + 3) * (EFX_LP / 4) + 2];
                const uint32_t qe = __builtin_amdgcn_alignbyte(d2, Q[j + 3], 3), qw = __builtin_amdgcn_alignbyte(Q[j + 3], d0, 1);
                const uint32_t qn = Q[j], qs = Q[j + 6];
                const uint32_t br = k1 - Q[j + 3], dk = k1 + Q[j + 3];
                const uint32_t f = (((qn + br) | (qs + br)) & ((qe + br) | (qw + br))) | (((dk - qn) | (dk - qs)) & ((dk - qe) | (dk - qw)));
                qm |= (f >> j) & (0x80808080u >> j);
            }
#endif
            qm &= xm & ym;
            if (dbg & 2) qm = 0;
        }
        // survivors of the four waves go to ONE list, so the full test below runs on packed lanes (a tile has ~100
        // survivors: two wave passes instead of four quarter-full ones)
        const int qcnt = __popc(qm);
        int nq;
        {
            int pos = block_excl_scan4(qcnt, s_scan, &nq);
            // an entry names the lane's block (tid) and the bit; phase 2 decodes it once per survivor -- this loop runs as
            // often as the fullest block of the wave has survivors, for all 64 lanes (round 3: 14 -> 8 instructions per trip)
            const unsigned tag = (unsigned)tid << 5;
            while (qm) {
                const int b = __ffs(qm) - 1;                 // column b >> 3, row 7 - (b & 7) of the block
                qm &= qm - 1;
                s_list[pos++] = (uint16_t)(tag | (unsigned)b);
            }
        }
        __syncthreads();
        // ---- phase 2: full 16-point segment test on the survivors, one lane per pixel; corners set their bit in
        //      the 64x64 bitmap ----
        for (int idx = tid; idx < ((dbg & 8) ? 0 : nq); idx += 256) {
            const int e = s_list[idx];
            const int blk = e >> 5, b = e & 31;              // block (blk & 15, blk >> 4) of 4 x 4 pixels, see phase 1
            const int lx = ((blk & 15) << 2) + (b >> 3), ly = ((blk >> 4) << 2) + 7 - (b & 7);
            bool corner = fast9_survivor_lds<EFX_LP>(tb + (ly + EFX_HALO) * EFX_LP + lx + EFX_HALO, threshold);
            if (corner && mask) {
                // spec S12: the level-0 mask is sampled where the keypoint will be reported (scalePoints, .cu:236-248)
                const int sx = min((int)(short)(L.scale * (float)(x0 + lx) + 0.5f), T->lv[0].cols - 1);
                const int sy = min((int)(short)(L.scale * (float)(y0 + ly) + 0.5f), T->lv[0].rows - 1);
                corner = mask[(size_t)sy * mask_pitch + sx] != 0;
            }
            if (corner)
                atomicOr(reinterpret_cast<unsigned*>(s_bitmap) + ly * 2 + (lx >> 5), 1u << (lx & 31));
        }
        __syncthreads();

        // ---- phase 3: canonical enumeration (spec S1): cell-major, raster inside the 16x16 cell.  Thread 16 c + rr counts
        //      row rr of cell c; the scan gives the number of corners before that row ----
        const int cell = tid >> 4, rr = tid & 15;
        const int cy = cell >> 2, cx = cell & 3;
        const int brow = cy * 16 + rr;
        const unsigned bits = (unsigned)(s_bitmap[brow] >> (cx * 16)) & 0xffffu;
        const int pre = block_excl_scan4(__popc(bits), s_scan + 4, &total);
        s_rowoff[tid] = (uint16_t)pre;
        if (rr == 0) s_celloff[cell] = pre;
        if (tid == 0) {
            s_celloff[EFX_CELLS_PER_TILE] = total;
            // The tile's count goes to the compact per-tile array and into its tile row's sum: a tile's canonical rank in the level
            // is the sum of the rows above + the counts left of it in its row (harris_kernel) -- no allocation, no scan pass, and a
            // frame of any density fits (round 6; until then a corner arena sized for a density, and void frames beyond it)
            tcount[gt] = (uint16_t)total;
            if (total > 0) __hip_atomic_fetch_add(&rowsum[L.row_base + ty].cand, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        TileHdr* h = hdr + tile;
        if (tid <= EFX_CELLS_PER_TILE) h->cell_off[tid] = (uint16_t)s_celloff[tid];
        if (tid == 32) { h->cand_start = 0; h->surv_count = 0; h->out_off = 0; }
        if (total == 0) return;

        // ---- phase 4: the corners to the tile's 512-byte slot (responses: harris_kernel).  Up to 256 of them as a list of 16-bit
        //      tile coordinates (x | y << 6) in canonical order: the lanes walk the survivor list of phase 1 again; a survivor whose
        //      bit is set is a corner and its place is the count before its row + the set bits left of it in the row.  More (a tile
        //      of a very dense frame): the 64 x 64 bitmap itself, harris_kernel enumerates it ----
        if (total <= EFX_SLOT_LIST) {
            uint16_t* slot = reinterpret_cast<uint16_t*>(slots + (size_t)gt * EFX_SLOT_BYTES);
            for (int idx = tid; idx < nq; idx += 256) {
                const int e = s_list[idx];
                const int blk = e >> 5, b = e & 31;
                const int lx = ((blk & 15) << 2) + (b >> 3), ly = ((blk >> 4) << 2) + 7 - (b & 7);
                const unsigned word = reinterpret_cast<const unsigned*>(s_bitmap)[ly * 2 + (lx >> 5)];
                if ((word >> (lx & 31)) & 1u) {
                    const unsigned rowbits = (word >> (lx & 16)) & 0xffffu;            // the row of the corner's 16 x 16 cell
                    const int k = (int)s_rowoff[((((ly >> 4) << 2) + (lx >> 4)) << 4) + (ly & 15)] + __popc(rowbits & ((1u << (lx & 15)) - 1u));
                    slot[k] = (uint16_t)(lx | (ly << 6));
                }
            }
        } else if (tid < EFX_TILE) {
            reinterpret_cast<unsigned long long*>(slots + (size_t)gt * EFX_SLOT_BYTES)[tid] = s_bitmap[tid];
        }
    }
}

// ================================================================================================
// Kernel A2: Harris responses.  One wave per tile; a lane takes one corner and gathers its 9x9 footprint
// straight from the level image (L2 / Infinity Cache hits: fast_kernel has just read it) as one 12-byte load per
// row -- measured faster than staging the tile in LDS again (117 vs 130 us per 8K frame).
// Also leaves the strongest corner of every 16x16 cell (the quick test of the NMS kernel).
// Round 6: the kernel also PLACES the tile's corners: the tile's canonical rank in its level is the sum of the tile rows above
// it (RowCtr, summed by fast_kernel's atomics) + the counts of the tiles left of it -- loads that depend on the tile's
// coordinates only, i.e. they travel with the header's -- and the records go to cand[rank + k].  The level's array holds exactly
// cap = cvRound(0.1 w h) records (.cpp:252): corners of rank >= cap do not exist (spec S2: the first `cap` in canonical order),
// so nothing is allocated and no frame can overflow anything.
// ================================================================================================
// inclusive scan inside each 16-lane row of the wave (DPP row_shr, no LDS)
__device__ __forceinline__ int efx_row16_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);     // row_shr:8
    return v;
}

// NW waves per tile (round 3, as nms_kernel): frames whose tiles do not fill the chip by themselves spread a tile's corners
// over several waves (the corners are independent; the per-cell maxima meet in LDS atomics).
// One tile's corners by the whole workgroup: responses, records at cand[rank + k], the tile's sixteen cell maxima.
template <int NW>
__device__ __forceinline__ void harris_one_tile(const LevelDev& L, int gt, int tx, int ty, int total, int rank, const uint8_t* __restrict__ src, int spitch,
                                                bool aligned, const unsigned char* __restrict__ slots, Corner* __restrict__ cand_all,
                                                Corner* __restrict__ cmax_all, TileHdr& h, int dbg, int lane)
{
    __shared__ unsigned long long s_cellmax[EFX_CELLS_PER_TILE];
    __shared__ unsigned s_celltie[EFX_CELLS_PER_TILE];     // largest response key that two corners of the cell share
    __shared__ uint16_t s_xy[EFX_SLOT_LIST];               // bitmap tiles: a chunk of the corners' tile coordinates in canonical order
    __shared__ unsigned s_enum[4][64];                     // ... and the enumeration state of row r's four segments
    auto empty_cells = [&]() {
        // a tile without (valid) corners: sixteen empty cell maxima (smooth frames: most tiles; see nms_kernel)
        if (lane < EFX_CELLS_PER_TILE) {
            Corner best; best.xy = 0xffffffffu; best.resp = -3.0e38f;
            cmax_all[L.cmax_base + (size_t)(ty * 4 + (lane >> 2)) * (L.tiles_x * 4) + tx * 4 + (lane & 3)] = best;
        }
    };
    if (total == 0) { empty_cells(); return; }              // no barrier (workgroup-uniform)
    if (lane == 0) h.cand_start = (uint32_t)rank;           // nms_kernel: where this tile's (and its neighbours') lists start
    const int n_valid = min(max(L.cap - rank, 0), total);   // the cap in canonical order (spec S2; cuda_fast.cu:245)
    if (n_valid == 0) { empty_cells(); return; }
    Corner* cand = cand_all + L.cand_base + (size_t)rank;
    __syncthreads();                                        // (a workgroup that takes several tiles: the previous one's cell maxima are out)
    if (lane < EFX_CELLS_PER_TILE) { s_cellmax[lane] = 0ull; s_celltie[lane] = 0u; }
    __syncthreads();

    // one corner: response, record, cell maximum
    auto corner = [&](int k, unsigned lxy) {
        const int x = tx * EFX_TILE + (int)(lxy & 63u), y = min(ty * EFX_TILE + (int)(lxy >> 6), L.rows - 1);
        const uint32_t xy = (uint32_t)x | ((uint32_t)(ty * EFX_TILE + (int)(lxy >> 6)) << 16);
        const uint8_t* c = src + (size_t)y * spitch + x;
        const float resp = (dbg & 4) ? 1.f : (aligned ? harris_rows(c - 4 * spitch - 4, spitch) : harris_bytes(c, spitch));
        Corner rec; rec.xy = xy; rec.resp = resp;
        cand[k] = rec;                                      // whole records, consecutive lanes: full-line stores
        // strongest corner of the 16x16 cell: 64-bit max of (response key, xy).  A corner that finds its own response
        // already there has an equal twin in the cell; if that response ends up being the cell's maximum, the NMS quick
        // test must not treat the stored corner as the only one of that strength (equal responses suppress each other).
        const int cell = (int)((lxy >> 10) & 3u) * 4 + (int)((lxy >> 4) & 3u);
        const unsigned long long key = (efx_select_key(0u, resp) & 0xffffffff00000000ull) | xy;
        const unsigned long long old = atomicMax(&s_cellmax[cell], key);
        if ((unsigned)(old >> 32) == (unsigned)(key >> 32)) atomicMax(&s_celltie[cell], (unsigned)(key >> 32));
    };

    if (total <= EFX_SLOT_LIST) {
        // the common case: fast_kernel left the tile coordinates as a list in canonical order
        const uint16_t* slot = reinterpret_cast<const uint16_t*>(slots + (size_t)gt * EFX_SLOT_BYTES);
        for (int k = lane; k < n_valid; k += 64 * NW) corner(k, (unsigned)slot[k] & 0xfffu);
    } else {
        // a tile of a very dense frame: the slot holds the 64 x 64 bitmap.  Lane r of wave 0 owns row r, i.e. one row of four
        // cells; it pops its bits cell column by cell column into an LDS chunk of 256 corners at their canonical places (a
        // segment's first place = the cell's offset + the bits of the rows above it in the cell), the workgroup computes the
        // chunk, and so on: every bit is popped once
        // (the lanes' enumeration state -- what is left of a row's four segments, the next place of each -- rests in LDS between the
        // chunks: held in registers across the Harris arithmetic it cost the whole kernel a wave of occupancy per SIMD)
        if (lane < 64) {
            const unsigned long long rb = reinterpret_cast<const unsigned long long*>(slots + (size_t)gt * EFX_SLOT_BYTES)[lane];
            const uint2 co = *reinterpret_cast<const uint2*>(&h.cell_off[(lane >> 4) * 4]);      // the offsets of this row's four cells
            const int coff[4] = { (int)(co.x & 0xffffu), (int)(co.x >> 16), (int)(co.y & 0xffffu), (int)(co.y >> 16) };
#pragma unroll
            for (int cx = 0; cx < 4; cx++) {
                const unsigned sg = (unsigned)(rb >> (16 * cx)) & 0xffffu;
                const int c = __popc(sg);
                s_enum[cx][lane] = sg | ((unsigned)(coff[cx] + efx_row16_incl_scan(c) - c) << 16);
            }
        }
        for (int c0 = 0; c0 < n_valid; c0 += EFX_SLOT_LIST) {
            const int end = min(c0 + EFX_SLOT_LIST, n_valid);
            if (lane < 64) {
#pragma unroll
                for (int cx = 0; cx < 4; cx++) {
                    unsigned sg = s_enum[cx][lane] & 0xffffu;
                    int bs = (int)(s_enum[cx][lane] >> 16);
                    while (__ballot(sg != 0u && bs < end) != 0ull) {
                        if (sg != 0u && bs < end) {
                            const int b = __ffs(sg) - 1;
                            sg &= sg - 1u;
                            if (bs >= c0) s_xy[bs - c0] = (uint16_t)((16 * cx + b) | (lane << 6));
                            bs++;
                        }
                    }
                    s_enum[cx][lane] = sg | ((unsigned)bs << 16);
                }
            }
            __syncthreads();
            for (int k = c0 + lane; k < end; k += 64 * NW) corner(k, (unsigned)s_xy[k - c0]);
            __syncthreads();
        }
    }
    __syncthreads();
    if (lane < EFX_CELLS_PER_TILE) {
        const unsigned long long m = s_cellmax[lane];
        Corner best; best.xy = 0xffffffffu; best.resp = -3.0e38f;
        if (m != 0ull) {
            uint32_t u = (uint32_t)(m >> 32);
            u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;          // inverse of the order-preserving map
            best.resp = __uint_as_float(u); best.xy = (uint32_t)m;
            if (s_celltie[lane] == (uint32_t)(m >> 32)) best.xy |= EFX_CMAX_TIE;      // y < 2^15: bit 31 is free
        }
        cmax_all[L.cmax_base + (size_t)(ty * 4 + (lane >> 2)) * (L.tiles_x * 4) + tx * 4 + (lane & 3)] = best;
    }
}

// Frames with the statistics of photographs have 8 .. 20 FAST corners per 64 x 64 tile, not ~110: a wave per tile then runs one
// round of Harris arithmetic for a dozen busy lanes.  Such frames are taken EFX_PACK_TPW TILES PER WAVE (harris_packed_kernel: a
// one-wave workgroup per group of consecutive tiles of a level): a tile's canonical rank is the first tile's rank + the counts of
// the group's tiles before it (one scan), the group's corners stand side by side -- corner n of the group is corner n - (corners
// of the tiles before its tile) of its tile -- and go through the arithmetic 64 per round, full rounds whatever the tiles' own
// counts are; the cell maxima meet in LDS.  Same records, same places: bit-identical, so which kernel a launch takes is a matter
// of speed alone -- the host picks it from the corner density of the context's PREVIOUS frame (a hint select_kernel leaves in host
// memory; video frames resemble their predecessors; efx_api.cpp), EFX_PACK=0 / 1 pins it.  On corner-rich frames the packed form
// is the slower one (63 -> 113 us: few, long waves, nothing hides their load latency).
// What it buys is modest, and measured why (round 6, 1/f^1.3 8K frame, tools/microbench/pack_ab.sh): 33 -> 26.5 us.  With the Harris
// arithmetic and its pixel loads compiled out the two forms take 17 and 12 us -- the per-corner part costs the same 14 .. 18 us
// packed or not: 328 000 scattered corners x 9 footprint rows are ~3 M line requests that miss L1 (on the corner-rich frame
// neighbouring corners share their lines), i.e. the kernel is bound by the gather from L2 there, not by half-empty waves.
// (Built and dropped on the way: both kinds of workgroup in every launch, each level's deciding for themselves from the row sums --
// the 25 500 workgroups that only find out that they are not needed cost a memory round trip each: 41 us instead of 33.)
__global__ __launch_bounds__(64) void harris_packed_kernel(
    const LevelTable* __restrict__ T, const uint8_t* __restrict__ img0, int pitch0, int aligned0,
    const uint8_t* __restrict__ pyramid, const unsigned char* __restrict__ slots, const uint16_t* __restrict__ tcount, const RowCtr* __restrict__ rows,
    Corner* __restrict__ cand_all, Corner* __restrict__ cmax_all, TileHdr* __restrict__ hdr_all, int dbg_arg, const FrameSet F)
{
    const int dbg = EFX_DBG(dbg_arg);
    {
        const size_t f = blockIdx.y;
        img0 = F.in.img0[f]; pyramid += f * F.fs.pyramid; slots += f * F.fs.slots; tcount += f * F.fs.hdr; rows += f * F.fs.rows;
        cand_all += f * F.fs.cand; cmax_all += f * F.fs.cmax; hdr_all += f * F.fs.hdr;
    }
    __shared__ unsigned long long s_cm[EFX_PACK_TPW][EFX_CELLS_PER_TILE];
    __shared__ unsigned s_ct[EFX_PACK_TPW][EFX_CELLS_PER_TILE];
    __shared__ int s_P[EFX_PACK_TPW + 1];                  // corners of the group before tile j (valid ones: below the level's cap)
    __shared__ int s_rk[EFX_PACK_TPW];                     // the tiles' canonical ranks
    __shared__ uint32_t s_txy[EFX_PACK_TPW];               // tx | ty << 16
    const int lane = threadIdx.x;
    int g = (int)blockIdx.x;
    int l = 0;
    while (l + 1 < T->nlevels && g >= T->lv[l].pack_groups) { g -= T->lv[l].pack_groups; l++; }
    const LevelDev& L = T->lv[l];
    if (!L.active || g >= L.pack_groups) return;
    const int ntiles = L.tiles_x * L.tiles_y;
    const int t0 = ntiles - EFX_PACK_TPW * (g + 1);        // densest (last) tiles first; the last group may start below tile 0
    const int tb = max(t0, 0);                             // the group's first tile
    const uint8_t* src = l == 0 ? img0 : pyramid + L.img_off;
    const int spitch = l == 0 ? pitch0 : L.pitch;
    const bool aligned = l == 0 ? aligned0 != 0 : true;
    // the first tile's canonical rank by the whole wave (as harris_kernel), the tiles' own counts by lanes 0 .. 15: one round trip
    const int t = t0 + lane;
    const bool tile_ok = lane < EFX_PACK_TPW && t >= 0;
    const int total = tile_ok ? min((int)tcount[L.tile_base + t], EFX_TILE * EFX_TILE) : 0;
    int btx, bty, bl;
    efx_tile_of(T, L.tile_base + tb, bl, btx, bty);
    int rank0;
    {
        const RowCtr* rp = rows + L.row_base;
        const uint16_t* tc = tcount + L.tile_base + bty * L.tiles_x;
        const int ym = max(bty - 1, 0), xm = max(btx - 1, 0);
        const int v0 = rp[min(lane, ym)].cand, v1 = rp[min(lane + 64, ym)].cand;
        const int w0 = (int)tc[min(lane, xm)], w1 = (int)tc[min(lane + 64, xm)];
        int r = (lane < bty ? v0 : 0) + (lane + 64 < bty ? v1 : 0) + (lane < btx ? w0 : 0) + (lane + 64 < btx ? w1 : 0);
        for (int q = lane + 128; q < bty; q += 64) r += rp[q].cand;
        for (int q = lane + 128; q < btx; q += 64) r += (int)tc[q];
        rank0 = __builtin_amdgcn_readlane(wave_incl_scan(r), 63);
    }
    const int rank = rank0 + wave_incl_scan(total) - total; // (lanes beyond the group hold 0)
    const int n_valid = min(max(L.cap - rank, 0), total);   // the cap in canonical order (spec S2; cuda_fast.cu:245)
    const int pincl = wave_incl_scan(n_valid);
    const int sum = __builtin_amdgcn_readlane(pincl, 63);
    uint32_t txy = 0;
    if (tile_ok) {
        int l2, tx, ty;
        efx_tile_of(T, L.tile_base + t, l2, tx, ty);
        txy = (uint32_t)tx | ((uint32_t)ty << 16);
        if (total > 0) hdr_all[L.tile_base + t].cand_start = (uint32_t)rank;     // nms_kernel: where the tile's lists start
    }
    if (__ballot(total > EFX_SLOT_LIST) != 0ull) {
        // a tile whose slot holds its bitmap (more than 256 corners): the group tile by tile, the whole wave each
        for (int j = 0; j < EFX_PACK_TPW; j++) {
            if (t0 + j < 0) continue;
            const uint32_t jxy = (uint32_t)__builtin_amdgcn_readlane((int)txy, j);
            const int jgt = L.tile_base + t0 + j;
            harris_one_tile<1>(L, jgt, (int)(jxy & 0xffffu), (int)(jxy >> 16), __builtin_amdgcn_readlane(total, j), __builtin_amdgcn_readlane(rank, j),
                               src, spitch, aligned, slots, cand_all, cmax_all, hdr_all[jgt], dbg, lane);
        }
        return;
    }
    if (lane < EFX_PACK_TPW) { s_P[lane] = pincl - n_valid; s_rk[lane] = rank; s_txy[lane] = txy; }
    if (lane == 0) s_P[EFX_PACK_TPW] = sum;
    static_assert(EFX_PACK_TPW == 4 || EFX_PACK_TPW == 8 || EFX_PACK_TPW == 16, "a power of two: the tile search; whole rounds of cells per wave");
#pragma unroll
    for (int q = 0; q < EFX_PACK_TPW * EFX_CELLS_PER_TILE / 64; q++) { (&s_cm[0][0])[lane + 64 * q] = 0ull; (&s_ct[0][0])[lane + 64 * q] = 0u; }
    __syncthreads();
    for (int n = lane; n < sum; n += 64) {
        // the tile that holds corner n of the group: the last j with s_P[j] <= n
        int j = 0;
#pragma unroll
        for (int step = EFX_PACK_TPW / 2; step >= 1; step >>= 1) if (s_P[j + step] <= n) j += step;
        const int k = n - s_P[j];
        const uint32_t jxy = s_txy[j];
        const int jtx = (int)(jxy & 0xffffu), jty = (int)(jxy >> 16);
        const unsigned lxy = (unsigned)reinterpret_cast<const uint16_t*>(slots + (size_t)(L.tile_base + t0 + j) * EFX_SLOT_BYTES)[k] & 0xfffu;
        const int x = jtx * EFX_TILE + (int)(lxy & 63u), yy = jty * EFX_TILE + (int)(lxy >> 6), y = min(yy, L.rows - 1);
        const uint32_t xy = (uint32_t)x | ((uint32_t)yy << 16);
        const uint8_t* c = src + (size_t)y * spitch + x;
        const float resp = (dbg & 4) ? 1.f : (aligned ? harris_rows(c - 4 * spitch - 4, spitch) : harris_bytes(c, spitch));
        Corner rec; rec.xy = xy; rec.resp = resp;
        cand_all[L.cand_base + (size_t)s_rk[j] + k] = rec;
        const int cell = (int)((lxy >> 10) & 3u) * 4 + (int)((lxy >> 4) & 3u);
        const unsigned long long key = (efx_select_key(0u, resp) & 0xffffffff00000000ull) | xy;
        const unsigned long long old = atomicMax(&s_cm[j][cell], key);
        if ((unsigned)(old >> 32) == (unsigned)(key >> 32)) atomicMax(&s_ct[j][cell], (unsigned)(key >> 32));
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EFX_PACK_TPW * EFX_CELLS_PER_TILE / 64; q++) {
        const int e = lane + 64 * q, j = e >> 4, cell = e & 15;
        if (t0 + j < 0) continue;
        const unsigned long long m = s_cm[j][cell];
        Corner best; best.xy = 0xffffffffu; best.resp = -3.0e38f;
        if (m != 0ull) {
            uint32_t u = (uint32_t)(m >> 32);
            u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;          // inverse of the order-preserving map
            best.resp = __uint_as_float(u); best.xy = (uint32_t)m;
            if (s_ct[j][cell] == (uint32_t)(m >> 32)) best.xy |= EFX_CMAX_TIE;
        }
        const uint32_t jxy = s_txy[j];
        cmax_all[L.cmax_base + (size_t)((int)(jxy >> 16) * 4 + (cell >> 2)) * (L.tiles_x * 4) + (int)(jxy & 0xffffu) * 4 + (cell & 3)] = best;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void harris_kernel(
    const LevelTable* __restrict__ T, const uint8_t* __restrict__ img0, int pitch0, int aligned0,
    const uint8_t* __restrict__ pyramid, const unsigned char* __restrict__ slots, const uint16_t* __restrict__ tcount, const RowCtr* __restrict__ rows,
    Corner* __restrict__ cand_all, Corner* __restrict__ cmax_all, TileHdr* __restrict__ hdr_all, int dbg_arg, const FrameSet F)
{
    const int dbg = EFX_DBG(dbg_arg);
    {
        const size_t f = blockIdx.y;
        img0 = F.in.img0[f]; pyramid += f * F.fs.pyramid; slots += f * F.fs.slots; tcount += f * F.fs.hdr; rows += f * F.fs.rows;
        cand_all += f * F.fs.cand; cmax_all += f * F.fs.cmax; hdr_all += f * F.fs.hdr;
    }
    __shared__ int s_rank;
    const int lane = threadIdx.x;                           // 0 .. 64 NW - 1: a corner slot of the round, not the hardware lane

    const int gt = T->total_tiles - 1 - xcd_interleaved(blockIdx.x, T->total_tiles);      // densest tiles first
    int l, tx, ty;
    efx_tile_of(T, gt, l, tx, ty);
    const LevelDev& L = T->lv[l];
    if (!L.active) return;
    const uint8_t* src = l == 0 ? img0 : pyramid + L.img_off;
    const int spitch = l == 0 ? pitch0 : L.pitch;
    const bool aligned = l == 0 ? aligned0 != 0 : true;
    TileHdr& h = hdr_all[gt];
    const int total = min((int)h.cell_off[EFX_CELLS_PER_TILE], EFX_TILE * EFX_TILE);
    // the tile's canonical rank: loads that need the tile's coordinates only (one memory round trip, beside the header's)
    int rank = 0;
    if (lane < 64) {
        // two tile rows and two tiles per lane are requested unconditionally (indices clamped), beside the header: ONE round trip.
        // (A loop with a data-dependent trip count waits for every load where it is issued: measured 7 us of this kernel's 61.)
        const RowCtr* rp = rows + L.row_base;
        const uint16_t* tc = tcount + L.tile_base + ty * L.tiles_x;
        const int ym = max(ty - 1, 0), xm = max(tx - 1, 0);
        const int v0 = rp[min(lane, ym)].cand, v1 = rp[min(lane + 64, ym)].cand;
        const int w0 = (int)tc[min(lane, xm)], w1 = (int)tc[min(lane + 64, xm)];
        rank = (lane < ty ? v0 : 0) + (lane + 64 < ty ? v1 : 0) + (lane < tx ? w0 : 0) + (lane + 64 < tx ? w1 : 0);
        for (int i = lane + 128; i < ty; i += 64) rank += rp[i].cand;       // levels of more than 8192 pixels per side
        for (int i = lane + 128; i < tx; i += 64) rank += (int)tc[i];
    }
    if (total > 0) {
        if (lane < 64) rank = __builtin_amdgcn_readlane(wave_incl_scan(rank), 63);      // DPP steps, no LDS round trips
        if (NW > 1) {
            if (lane == 0) s_rank = rank;
            __syncthreads();
            rank = s_rank;
        }
    }
    harris_one_tile<NW>(L, gt, tx, ty, total, rank, src, spitch, aligned, slots, cand_all, cmax_all, h, dbg, lane);
}

// ================================================================================================
// Kernel C: radius non-max suppression (radiusSuppressionKernel + IsMaxPoint, .cu:62-97, 202-216).
// The work is latency bound (a few dozen corners per tile, dependent look-ups), so it is laid out for
// maximum waves in flight and short dependency chains: ONE WAVE per 64x64 tile, ~3 KB of LDS.
//   prologue the headers of the 3x3 neighbouring tiles go to LDS (every later list look-up reads them there);
//   phase A  one lane per corner: quick test against the strongest corner of each neighbouring cell (the
//            per-cell maxima are written by harris_kernel); this suppresses ~95 % of the corners with 9 loads;
//            the corners that pass ("hard") are collected in an LDS list across rounds;
//   phase B  the hard corners are scanned exactly, 8 lanes per corner, 8 corners at a time: the lanes walk the
//            corner lists of those neighbouring cells whose strongest corner is not weaker (IsMaxPoint);
//   survivors are compacted by ballot in canonical order and appended to the level's survivor array.
// ================================================================================================
#define NMS_HCAP 256
#ifndef EFX_NMS_WIDE_TILES
#define EFX_NMS_WIDE_TILES 2048      // frames with at most this many tiles (FHD and below) run four waves per tile
#endif
#ifndef EFX_NMS_MID_TILES
#define EFX_NMS_MID_TILES 8192       // ... up to this many (4K): EFX_NMS_MID_NW waves per tile
#endif
#ifndef EFX_NMS_MID_NW
#define EFX_NMS_MID_NW 2
#endif

// NW waves per tile (round 3).  One wave per tile fills the chip when a frame has tens of thousands of tiles (8K: 25 500);
// a 4K frame has 6 400 and an FHD frame 1 650, and then the kernel takes as long as its densest tile's ONE wave needs for
// all of that tile's rounds.  With NW = 4 the rounds of 64 corners go round-robin over four waves, each with its own list
// of hard corners (the exact scans are wave-local as before); the survivor bits meet in s_keep, wave 0 compacts.
template <int NW>
__global__ __launch_bounds__(64 * NW) void nms_kernel(const LevelTable* __restrict__ T, TileHdr* __restrict__ hdr,
                                                 const Corner* __restrict__ cand_all, const Corner* __restrict__ cmax_all,
                                                 Corner* __restrict__ surv_all, RowCtr* __restrict__ rows, int* __restrict__ hist,
                                                 Counters* __restrict__ cnt, int radius, int dbg_arg, const FrameStride fs)
{
    const int dbg = EFX_DBG(dbg_arg);
    {
        const size_t f = blockIdx.y;
        hdr += f * fs.hdr; cand_all += f * fs.cand; cmax_all += f * fs.cmax; surv_all += f * fs.cand; rows += f * fs.rows; hist += f * fs.hist;
        cnt += f;
    }
    __shared__ Corner s_hme_all[NW][NMS_HCAP];
    __shared__ uint16_t s_hidx_all[NW][NMS_HCAP];
    __shared__ uint16_t s_hneed_all[NW][NMS_HCAP];         // block_radius 1: the cells (bit q = cell q of the 3x3) that hold a rival
    __shared__ unsigned long long s_keep[64];            // survivor bits of round r (64 rounds = 4096 corners = a full tile)
    __shared__ int s_void;                               // a wave found a record that is not of this tile: nothing is written
    __shared__ __attribute__((aligned(64))) uint32_t s_nb[9][16];   // TileHdr of the 3x3 neighbouring tiles
    __shared__ Corner s_cm[6][6];                        // cell maxima of the tile's 4x4 cells + one ring (block_radius 1)

    const int gt = T->total_tiles - 1 - xcd_interleaved(blockIdx.x, T->total_tiles);      // densest (upper-level) tiles first
    int l, tx, ty;
    efx_tile_of(T, gt, l, tx, ty);
    const LevelDev& L = T->lv[l];
    if (!L.active) return;
    const int tile = gt - L.tile_base;
    TileHdr* hl = hdr + L.tile_base;
    const Corner* cand = cand_all + L.cand_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    Corner* s_hme = s_hme_all[wv]; uint16_t* s_hidx = s_hidx_all[wv]; uint16_t* s_hneed = s_hneed_all[wv];
    if (cnt->sum.overflow) return;                          // void frame (a record failed its range check: DESIGN.md section 7)

    // A wave beyond the tile's corners (most tiles of a sparse frame have fewer than 64 * NW) leaves at once: its slot is
    // free after one load instead of after the tile.  (s_barrier waits for the waves of the workgroup that have not
    // terminated; the prologue below is wave 0's.)
    if (NW > 1 && wv > 0 && 64 * wv >= (int)hl[tile].cell_off[EFX_CELLS_PER_TILE]) return;
    // A tile without corners has nothing to suppress and nothing to report (fast_kernel left its survivor fields at zero): no
    // prologue, no barriers.  Corner-rich frames have no such tiles; on natural-image statistics (1 / f^1.3: 13 corners per tile
    // on average) they are a good part of the 25 500, and the kernel's time there was its per-tile fixed cost (round 5).
    if (hl[tile].cell_off[EFX_CELLS_PER_TILE] == 0) return;
    if (tid < 64) s_keep[tid] = 0ull;
    if (tid == 0) s_void = 0;
    // nine headers of 64 bytes: eight lanes x 8 bytes per header, eight headers in the first pass, the ninth in a second one
    // (round 3: 16 lanes x 4 bytes and three passes with a division by three each)
    if (tid < 64) {
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int t = (tid >> 3) + 8 * pass, w2 = tid & 7;
            if (t < 9) {
                const int dyt = (t * 11) >> 5, dxt = t - 3 * dyt;            // t / 3, t % 3 for t < 9
                const int ntx = tx - 1 + dxt, nty = ty - 1 + dyt;
                uint2 v = make_uint2(0u, 0u);
                if (ntx >= 0 && ntx < L.tiles_x && nty >= 0 && nty < L.tiles_y)
                    v = reinterpret_cast<const uint2*>(&hl[nty * L.tiles_x + ntx])[w2];
                *reinterpret_cast<uint2*>(&s_nb[t][2 * w2]) = v;
            }
        }
    }
    // (the per-cell maxima are taken over the corners below the 10 % cap only -- harris_kernel computes no others --, so a cell
    // maximum is always a valid suppressor; until round 6 they included corners beyond the cap and had to be screened here)
    // the cell maxima are fetched in the same memory round trip as the headers (both only need the tile's coordinates)
    // a cell beyond the grid is an EMPTY cell (harris_kernel's encoding: no coordinate, the lowest response): the
    // neighbourhood of IsMaxPoint is clipped to the grid (.cu:70-73), so such a cell holds nothing -- and phase A may rely on
    // "a neighbouring cell's maximum is never this corner itself"
    const int gw_ = (L.cols + EFX_CELL - 1) / EFX_CELL, gh_ = (L.rows + EFX_CELL - 1) / EFX_CELL;
    const int cyu = ty * 4 - 1 + tid / 6, cxu = tx * 4 - 1 + tid % 6;
    const int cy = min(max(cyu, 0), gh_ - 1), cx = min(max(cxu, 0), gw_ - 1);
    Corner cm; cm.xy = 0xffffffffu; cm.resp = -3.0e38f;
    const bool cell_exists = cyu == cy && cxu == cx;
    if (tid < 36 && cell_exists) cm = cmax_all[L.cmax_base + (size_t)cy * (L.tiles_x * 4) + cx];
    __syncthreads();                                     // s_nb is read below
    {
        // the nine headers address the corner array: a count beyond a tile's 4096 pixels voids the frame
        // (every wave evaluates the same nine headers: the exit is workgroup-uniform)
        bool bad = false;
        if (lane < 9) {
            const TileHdr* nh = reinterpret_cast<const TileHdr*>(&s_nb[lane][0]);
            const unsigned tot = nh->cell_off[EFX_CELLS_PER_TILE];
            bad = tot > (unsigned)(EFX_TILE * EFX_TILE);      // (list ranges are clipped to the level's cap where they are used)
        }
        if (__ballot(bad) != 0ull) {
            if (tid == 0) efx_raise_overflow(T, cnt);
            return;
        }
    }
    if (tid < 36) s_cm[tid / 6][tid % 6] = cm;
    __syncthreads();
    // header of the tile that holds cell (bx, by): the LDS copy when it is a neighbour (always, up to radius 64)
    auto nhdr = [&](int bx, int by) -> const TileHdr* {
        const int ntx = bx >> 2, nty = by >> 2;
        const int dx = ntx - tx, dy = nty - ty;
        if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1) return reinterpret_cast<const TileHdr*>(&s_nb[(dy + 1) * 3 + dx + 1][0]);
        return &hl[nty * L.tiles_x + ntx];
    };
    const TileHdr& h = *reinterpret_cast<const TileHdr*>(&s_nb[4][0]);
    // a tile's corners start at its canonical rank in the level; those of rank >= cap do not exist (spec S2; cuda_fast.cu:245)
    const unsigned own_start = min(h.cand_start, (unsigned)L.cap);
    const Corner* own = cand + own_start;

    const int n_own = h.cell_off[EFX_CELLS_PER_TILE];
    const int n_valid = min(L.cap - (int)own_start, n_own);
    if (dbg == 1) return;

    const int image_radius = radius * radius;           // cvCeil(radius * radius), .cu:291
    const int block_radius = (radius + EFX_CELL - 1) / EFX_CELL;   // cvCeil(radius / CELL_SIZE), .cu:292
    const int gw = (L.cols + EFX_CELL - 1) / EFX_CELL, gh = (L.rows + EFX_CELL - 1) / EFX_CELL;
    const int gwp = L.tiles_x * 4;                      // row pitch of the per-cell maxima table
    const Corner* cmax = cmax_all + L.cmax_base;
    const bool quick_ok = block_radius <= 2;
    const int span = 2 * block_radius + 1;
    const int ncell = span * span;
    const int grp = lane >> 3, sub = lane & 7;

    // one cell's list [nb, ne) walked by the 8 lanes of a group, four entries per lane in flight (a long list is a chain of
    // dependent trips to L2 otherwise: dense frames).  Entries past the end are clamped onto the last one (harmless repeats).
    auto walk_list = [&](unsigned nbase, int nb, int ne, const Corner& m, int mx, int my, bool& kill) {
        const Corner* lst = cand + (size_t)nbase;
        if (__ballot(ne - nb > 8) == 0ull) {
            // every list of this step fits one entry per lane (sparse frames: nearly always): one load, no clamping
            const int j = nb + sub;
            if (j < ne) {
                const Corner o = lst[j];
                const i16x2 d = __builtin_bit_cast(i16x2, m.xy) - __builtin_bit_cast(i16x2, o.xy);
                kill |= (o.xy != m.xy && m.resp <= o.resp && __builtin_amdgcn_sdot2(d, d, 0, false) < image_radius);
            }
            return;
        }
        for (int j = nb + sub; j < ne; j += 32) {
            Corner o[4];
#pragma unroll
            for (int u = 0; u < 4; u++) o[u] = lst[min(j + 8 * u, ne - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const i16x2 d = __builtin_bit_cast(i16x2, m.xy) - __builtin_bit_cast(i16x2, o[u].xy);
                kill |= (o[u].xy != m.xy && m.resp <= o[u].resp && __builtin_amdgcn_sdot2(d, d, 0, false) < image_radius);
            }
        }
    };

    // phase B over the hard corners collected so far
    auto scan_hard = [&](int nh) {
        for (int h0 = 0; h0 < nh; h0 += 8) {
            const int hi = h0 + grp;
            const bool act = hi < nh;
            Corner m; m.xy = 0; m.resp = 0.f;
            if (act) m = s_hme[hi];
            const int mx = m.xy & 0xffff, my = m.xy >> 16;
            const int bx1 = mx / EFX_CELL, by1 = my / EFX_CELL;
            const int cx0 = max(bx1 - block_radius, 0), cx1 = min(bx1 + block_radius, gw - 1);
            const int cy0 = max(by1 - block_radius, 0), cy1 = min(by1 + block_radius, gh - 1);
            bool kill = false;
            if (block_radius == 1) {
                // common case: phase A already knows which of the 9 cells hold a rival (s_hneed); lane `sub` fetches the
                // list range of cell `sub`, lane 0 also that of cell 8, straight from the headers in LDS
                const int needm = act ? (int)s_hneed[hi] : 0;
                int lb[2] = { 0, 0 }, le[2] = { 0, 0 }; unsigned lbase[2] = { 0u, 0u };
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int ci = 8 * q + sub;
                    if (ci < 9 && ((needm >> ci) & 1)) {
                        const int bx = min(max(bx1 - 1 + (ci % 3), 0), gw - 1), by = min(max(by1 - 1 + (ci / 3), 0), gh - 1);
                        const TileHdr* nh2 = reinterpret_cast<const TileHdr*>(&s_nb[((by >> 2) - ty + 1) * 3 + ((bx >> 2) - tx + 1)][0]);
                        const int c = (by & 3) * 4 + (bx & 3);
                        lb[q] = nh2->cell_off[c];
                        le[q] = nh2->cell_off[c + 1];
                        lbase[q] = min(nh2->cand_start, (unsigned)L.cap);
                        le[q] = min(le[q], L.cap - (int)lbase[q]);                         // corners beyond the cap do not exist
                    }
                }
                unsigned gneed = ((unsigned)(__ballot(le[0] > lb[0]) >> (grp * 8)) & 0xffu) |
                                 (((unsigned)(__ballot(le[1] > lb[1]) >> (grp * 8)) & 0x01u) << 8);
                while (__ballot(gneed != 0u) != 0ull) {
                    // the corner's own cell first: its rivals are the closest ones
                    const int i = (gneed & 16u) ? 4 : (gneed ? __ffs(gneed) - 1 : 0);
                    const int src = (lane & 56) + (i & 7);
                    const int b0 = __shfl(lb[0], src, 64), b1 = __shfl(lb[1], src, 64);
                    const int e0 = __shfl(le[0], src, 64), e1 = __shfl(le[1], src, 64);
                    const unsigned a0 = (unsigned)__shfl((int)lbase[0], src, 64), a1 = (unsigned)__shfl((int)lbase[1], src, 64);
                    const int nb = i < 8 ? b0 : b1, ne = i < 8 ? e0 : e1;
                    const unsigned nbase = i < 8 ? a0 : a1;
                    if (gneed) walk_list(nbase, nb, ne, m, mx, my, kill);
                    gneed &= ~(1u << i);
                    // one suppressor is enough: a group whose corner is dead stops walking (most hard corners of a
                    // dense tile die in the first list)
                    if (((unsigned)(__ballot(kill) >> (grp * 8)) & 0xffu) != 0u) gneed = 0u;
                }
            } else
            // cells of the neighbourhood in chunks of 16: lane `sub` fetches the list ranges of cells c0+sub and
            // c0+8+sub, then the 8 lanes walk every list that is left together
            for (int c0 = 0; c0 < ncell; c0 += 16) {
                int lb[2] = { 0, 0 }, le[2] = { 0, 0 }; unsigned lbase[2] = { 0u, 0u };
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int ci = c0 + 8 * q + sub;
                    const int oy = ci / span, ox = ci - oy * span;
                    const int bx = bx1 - block_radius + ox, by = by1 - block_radius + oy;
                    if (act && ci < ncell && bx >= cx0 && bx <= cx1 && by >= cy0 && by <= cy1) {
                        const TileHdr* nh2 = nhdr(bx, by);
                        const int c = (by & 3) * 4 + (bx & 3);
                        lbase[q] = min(nh2->cand_start, (unsigned)L.cap);
                        const int nn = L.cap - (int)lbase[q];
                        lb[q] = nh2->cell_off[c];
                        le[q] = nh2->cell_off[c + 1];
                        if (le[q] > nn) le[q] = nn;
                        // a cell whose strongest corner is weaker than this corner cannot suppress it: skip the cell
                        if (quick_ok && le[q] > lb[q] && cmax[by * gwp + bx].resp < m.resp) le[q] = lb[q];
                        // nor can a cell whose nearest pixel is not inside the radius
                        const int ex = max(max(bx * EFX_CELL - mx, mx - (bx * EFX_CELL + EFX_CELL - 1)), 0);
                        const int ey = max(max(by * EFX_CELL - my, my - (by * EFX_CELL + EFX_CELL - 1)), 0);
                        if (ex * ex + ey * ey >= image_radius) le[q] = lb[q];
                    }
                }
                // bit i of gneed <-> cell c0+i still has a list to walk (group-uniform)
                unsigned gneed = ((unsigned)(__ballot(le[0] > lb[0]) >> (grp * 8)) & 0xffu) |
                                 (((unsigned)(__ballot(le[1] > lb[1]) >> (grp * 8)) & 0xffu) << 8);
                while (__ballot(gneed != 0u) != 0ull) {
                    const int i = gneed ? __ffs(gneed) - 1 : 0;
                    const int src = (lane & 56) + (i & 7);
                    const int b0 = __shfl(lb[0], src, 64), b1 = __shfl(lb[1], src, 64);
                    const int e0 = __shfl(le[0], src, 64), e1 = __shfl(le[1], src, 64);
                    const unsigned a0 = (unsigned)__shfl((int)lbase[0], src, 64), a1 = (unsigned)__shfl((int)lbase[1], src, 64);
                    const int nb = i < 8 ? b0 : b1, ne = i < 8 ? e0 : e1;
                    const unsigned nbase = i < 8 ? a0 : a1;
                    if (gneed) walk_list(nbase, nb, ne, m, mx, my, kill);
                    gneed &= gneed - 1u;
                    if (((unsigned)(__ballot(kill) >> (grp * 8)) & 0xffu) != 0u) gneed = 0u;       // one suppressor is enough
                }
                if (__ballot(act && !kill) == 0ull) break;       // every corner of this batch is dead
            }
            const unsigned gk = (unsigned)(__ballot(kill) >> (grp * 8)) & 0xffu;
            if (act && sub == 0 && gk == 0u) {
                const int idx = s_hidx[hi];
                atomicOr(&s_keep[idx >> 6], 1ull << (idx & 63));
            }
        }
    };

    // LDS traffic of one wave is ordered; the hard-corner lists are wave-local: wave-level fences are all the loop needs
    auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    int nh = 0;
    bool foreign = false;
    for (int k0 = 64 * wv; k0 < n_valid; k0 += 64 * NW) {
        const int k = k0 + lane;
        bool hard = false, sure = false;
        int need = 0x1ff;                                     // neighbour cells the exact scan has to walk (all, unless phase A knows better)
        Corner me; me.xy = 0; me.resp = 0.f;
        if (k < n_valid) me = own[k];
        // range check: a record that is not of this tile would index LDS and the arrays out of range;
        // the frame is void (wave-uniform exit: the workgroup is this wave)
        if (__ballot(k < n_valid && ((int)((me.xy & 0xffff) >> 6) != tx || (int)(me.xy >> 22) != ty)) != 0ull) {
            if (lane == 0) { efx_raise_overflow(T, cnt); s_void = 1; }
            foreign = true;
            break;                                           // this wave stops; the workgroup meets at the barrier below
        }
        if (k < n_valid) {
            const int mx = me.xy & 0xffff, my = me.xy >> 16;
            const int bx1 = mx / EFX_CELL, by1 = my / EFX_CELL;
            hard = true;
            if (quick_ok) {
                if (block_radius == 1) {
                    // the common case as straight-line code: 9 loads, then branch-free compares (this kernel is bound
                    // by instruction issue, scalar exec-mask bookkeeping included)
                    Corner o[9];
                    const int cj = bx1 - tx * 4, ci = by1 - ty * 4;      // this corner's cell inside the tile, 0..3
#pragma unroll
                    for (int q = 0; q < 9; q++) o[q] = s_cm[ci + q / 3][cj + q % 3];
                    bool kill = false;
                    int rival = 0;
                    // A neighbouring cell matters only if its nearest pixel is inside the radius (round 3): with a small
                    // radius most corners are too far from most of the eight neighbours -- radius 5: 2.3 cells on
                    // average instead of 9 -- and dense frames stop walking lists that cannot hold a suppressor.
                    // gx[k], gy[k]: squared gap to the cell column / row k - 1 (0 for the own cell)
                    const int px = mx & (EFX_CELL - 1), py = my & (EFX_CELL - 1);
                    const int gx[3] = { (px + 1) * (px + 1), 0, (EFX_CELL - px) * (EFX_CELL - px) };
                    const int gy[3] = { (py + 1) * (py + 1), 0, (EFX_CELL - py) * (EFX_CELL - py) };
                    // The kernel is bound by instruction issue and this loop was 60 % of its VALU instructions (round 3,
                    // tools/microbench/stage_insts.sh: 14.8 M of 24.3 M): the squared distance is one v_pk_sub_i16 + one
                    // v_dot2_i32_i16 on the packed coordinate word (|dx|, |dy| < 2^15; an empty cell's 0x7fffffff gives
                    // a huge distance and a response no corner is below), the conditions stay booleans (compares into
                    // lane masks, combined on the scalar unit) instead of 0 / 1 integers.
                    const i16x2 mev = __builtin_bit_cast(i16x2, me.xy);
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        const uint32_t oxy = o[q].xy & ~EFX_CMAX_TIE;
                        const i16x2 d = mev - __builtin_bit_cast(i16x2, oxy);
                        const int d2 = __builtin_amdgcn_sdot2(d, d, 0, false);
                        if (q == 4) {
                            // the corner's own cell: its maximum may be this corner.  Then the exact scan must still walk the
                            // cell when another corner of it has the same response (ties suppress)
                            const bool other = oxy != me.xy;
                            const bool ge = other && me.resp <= o[q].resp;
                            if (ge || (!other && (int)o[q].xy < 0)) rival |= 1 << q;
                            kill = kill || (ge && d2 < image_radius);
                        } else {
                            // a neighbouring cell's maximum is another corner (cells beyond the grid are empty, see the prologue)
                            const bool ge = me.resp <= o[q].resp;
                            // cell q holds a corner at least as strong: the exact scan must walk it
                            if (ge && gx[q % 3] + gy[q / 3] < image_radius) rival |= 1 << q;
                            kill = kill || (ge && d2 < image_radius);
                        }
                    }
                    hard = !kill;
                    need = rival;
                    // stronger than every neighbouring cell maximum: nothing in the neighbourhood can suppress it, so
                    // it survives without the exact scan (most true survivors are such local maxima)
                    sure = rival == 0;
                } else {
                    const int minx = max(bx1 - block_radius, 0), maxx = min(bx1 + block_radius, gw - 1);
                    const int miny = max(by1 - block_radius, 0), maxy = min(by1 + block_radius, gh - 1);
                    for (int by = miny; by <= maxy; by++)
                        for (int bx = minx; bx <= maxx; bx++) {
                            const Corner o = cmax[by * gwp + bx];
                            const uint32_t oxy = o.xy & ~EFX_CMAX_TIE;
                            const int dx = mx - (int)(oxy & 0xffff), dy = my - (int)(oxy >> 16);
                            if (oxy != me.xy && me.resp <= o.resp && dx * dx + dy * dy < image_radius) hard = false;
                        }
                }
            }
        }
        const unsigned long long sm = __ballot(sure);
        if (lane == 0 && sm) atomicOr(&s_keep[k0 >> 6], sm);
        hard = hard && !sure;
        const unsigned long long hm = __ballot(hard);
        if (hard) {
            const int pos = nh + __popcll(hm & ((1ull << lane) - 1ull));
            s_hme[pos] = me;
            s_hidx[pos] = (uint16_t)k;
            s_hneed[pos] = (uint16_t)need;
        }
        nh += __popcll(hm);
        if (nh > NMS_HCAP - 64 || k0 + 64 * NW >= n_valid) {
            wave_sync();
            if (dbg != 2) scan_hard(nh);
            nh = 0;
            wave_sync();
        }
    }
    (void)foreign;
    __syncthreads();                                     // every wave's survivor bits are in s_keep
    if (dbg == 3 || wv != 0 || s_void) return;
    // lane r holds the survivor ballot of round r
    const unsigned long long my_round_mask = s_keep[lane];
    const int nsurv = __shfl(wave_incl_scan(__popcll(my_round_mask)), 63, 64);
    // second pass: the survivors in canonical order, at the tile's own place in the level's index space (round 6: no allocation --
    // a tile has at most as many survivors as corners, and its corners' places are its own); their number joins the tile row's sum
    // and every survivor one bin of the level's key histogram (select_kernel finds the quota's threshold bin there without a pass
    // over the survivors; its leader leaves the histogram zero again for the next frame)
    Corner* surv = surv_all + L.cand_base + own_start;
    int* lhist = hist + (size_t)l * EFX_HIST_BINS;
    int base = 0, round = 0;
    for (int k0 = 0; k0 < n_valid; k0 += 64, round++) {
        const unsigned long long m = (unsigned long long)(unsigned)__shfl((int)(my_round_mask & 0xffffffffu), round, 64) |
                                     ((unsigned long long)(unsigned)__shfl((int)(my_round_mask >> 32), round, 64) << 32);
        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
        if ((m >> lane) & 1ull) {
            const Corner c = own[k0 + lane];
            surv[slot] = c;
            if (dbg < 4)      // (debug builds, EFX_DEBUG = 64 / 80: without the histogram / the row sums -- selection invalid, timing only)
            __hip_atomic_fetch_add(&lhist[efx_hist_word((uint32_t)(efx_select_key(c.xy, c.resp) >> (64 - EFX_HIST_BITS)))], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        base += __popcll(m);
    }
    if (lane == 0) {
        hl[tile].surv_count = (uint32_t)nsurv;
        if (nsurv > 0 && dbg < 5) __hip_atomic_fetch_add(&rows[L.row_base + ty].surv, nsurv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ================================================================================================
// Kernel D: per-level quota (limitPoints, .cu:344-358, spec S3) + output offsets.
// Round 6: the kernel is spread over the chip (until then: ONE workgroup of 1024 threads and 132 KB of LDS per level, i.e. eight
// CUs of 256, 27 us alone and up to 330 us when it had to wait for a free CU behind other frames' kernels):
//   leaders   workgroup l < nlevels: the level's survivor count (sum of its tile rows' sums) and, when it exceeds the quota, the
//             bin b* of the key histogram that holds the quota-th largest key -- nms_kernel has filled the histogram, so no pass
//             over the survivors is needed: 64 bins per thread as 32 coalesced loads, one scan, the owner's bins again;
//   counters  the other workgroups, a lane per tile (256 tiles each): they wait for their level's leader (the leaders have the
//             lowest workgroup numbers, are dispatched first and wait for nobody), count the tile's survivors above b*, append
//             the keys OF b* -- a few dozen on real frames -- to the level's list, and take every survivor's count out of the
//             histogram again (it is zero when the frame is done, without a pass that clears it);
//   last      the counting workgroup that finishes a level last (a counter per level) ranks the list in LDS -> the exact
//             threshold key, and scans the tiles' counts (+ the list's selected keys) in canonical order: the output offsets.
// Everything that crosses workgroups inside the launch is a relaxed device-scope atomic access (a word that is its own flag,
// counts, list keys): measured on this part, an acquire / release FENCE costs microseconds each (a last workgroup that fenced and
// read eight leaders' flags with acquire loads spent 30 us there), a device-scope load or store costs a memory round trip.
// A bin with more than EFX_SEL_LIST_CAP keys (synthetic frames with thousands of equal responses): the last workgroup runs
// 12-bit radix passes over the level's tiles itself (slow, exact).
// ================================================================================================
#define SEL_NT 256
#define SEL_ILP 8                     // survivors a counting lane has in flight
#define SEL_SEG 8192                  // tiles whose counts (16 bits each) the last workgroup holds in LDS at a time
// INVESTIGATION (-DEFX_SEL_TIMING builds only): the leader of level 0 and the workgroup that finishes level 0 print their phase
// times (10 ns ticks of the constant-frequency counter)
#ifdef EFX_SEL_TIMING
#define SEL_TL(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) sel_t[i] = wall_clock64(); } while (0)
#define SEL_TF(i) do { if (fl == 0 && blockIdx.y == 0 && tid == 0) sel_t[i] = wall_clock64(); } while (0)
#define SEL_TC(i) do { if (blockIdx.y == 0 && tid == 0) sel_t[i] = wall_clock64(); } while (0)
#else
#define SEL_TL(i) do { } while (0)
#define SEL_TF(i) do { } while (0)
#define SEL_TC(i) do { } while (0)
#endif

// device-scope accesses that are coherent per location without a fence
template <class V> __device__ __forceinline__ V efx_ld(const V* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class V> __device__ __forceinline__ void efx_st(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the level's published word A: bit 63 ready | bits 32..47 bin + 1 | bits 0..31 keys in the bin; word B: remaining << 32 | min(n, quota).
// A bounded wait (the leaders precede the counters in dispatch order and never wait themselves, so it always ends; the bound only
// keeps a broken build from hanging the GPU)
__device__ __forceinline__ unsigned long long efx_wait_word(const unsigned long long* w)
{
    for (int spin = 0; spin < (1 << 22); spin++) {
        const unsigned long long v = efx_ld(w);
        if (v >> 63) return v;
        __builtin_amdgcn_s_sleep(2);
    }
    return (1ull << 63) | ((unsigned long long)(EFX_HIST_BINS + 1) << 32);      // "nothing is selected"
}

__global__ __launch_bounds__(SEL_NT) void select_kernel(const LevelTable* __restrict__ T, TileHdr* __restrict__ hdr,
                                                        const Corner* __restrict__ surv_all, const RowCtr* __restrict__ rows,
                                                        int* __restrict__ hist, unsigned long long* __restrict__ sel_list,
                                                        uint32_t* __restrict__ nsel, Counters* __restrict__ cnt,
                                                        int capacity, const FrameOut out, const FrameStride fs)
{
    {
        const size_t f = blockIdx.y;
        hdr += f * fs.hdr; surv_all += f * fs.cand; rows += f * fs.rows; hist += f * fs.hist; sel_list += f * fs.list; nsel += f * fs.hdr; cnt += f;
    }
    int* const d_count = out.count[blockIdx.y];
    __shared__ __attribute__((aligned(16))) int s_scan[8];
    __shared__ int s_own, s_want, s_bin, s_inbin, s_rem, s_last[EFX_MAX_LEVELS], s_m;
    __shared__ unsigned long long s_thresh;
    __shared__ unsigned long long s_keys[EFX_SEL_LIST_CAP];      // the last workgroup's list (16 KB)
    __shared__ int s_cnt[SEL_SEG / 2];                           // ... a segment of the tiles' counts, two per word (16 KB; a tile has at most
                                                                 // 4096 + EFX_SEL_LIST_CAP selected survivors); the slow path's 4096-bin histogram
    __shared__ int s_sub[256];
    const int tid = threadIdx.x, nl = T->nlevels;
    constexpr int TOPSH = 64 - EFX_HIST_BITS;
    constexpr int PER = EFX_HIST_BINS / SEL_NT;                  // bins a leader thread owns
    static_assert(PER == 64 || PER == 128, "leader: the owner's bins are re-read by one wave; owners == threads");
    static_assert((EFX_HIST_BINS / 32) % PER == 0, "leader: a line's words belong to owners BINS / 32 / PER apart");
#ifdef EFX_SEL_TIMING
    unsigned long long sel_t[12] = { 0 };
#endif

    if ((int)blockIdx.x < nl) {
        // ---------------- leader of level l ----------------
        const int l = blockIdx.x;
        const LevelDev& L = T->lv[l];
        SEL_TL(0);
        int part = 0, partc = 0;
        for (int i = tid; i < L.tiles_y; i += SEL_NT) { part += rows[L.row_base + i].surv; partc += rows[L.row_base + i].cand; }
        int n, nc;
        (void)block_excl_scan4(part, s_scan, &n);
        (void)block_excl_scan4(partc, s_scan + 4, &nc);
        SEL_TL(1);
        const bool none = !L.active || L.quota <= 0 || cnt->sum.overflow != 0;      // nothing is selected (void frame: N = 0)
        if (tid == 0) { s_bin = none ? EFX_HIST_BINS : -1; s_inbin = 0; s_rem = 0; s_own = 0; s_want = 0; }
        __syncthreads();
        if (!none && n > L.quota) {
            const int* lh = hist + (size_t)l * EFX_HIST_BINS;
            const int own = SEL_NT - 1 - tid;                   // bins PER own .. PER own + PER - 1: thread 0 owns the top
            // The whole histogram as coalesced int4 loads (all requested before the first is used); word w of line n is bin
            // w (BINS / 32) + n (efx_hist_word), whose owner is that bin / PER: the counts meet in s_sub[owner]
            s_sub[tid] = 0;
            __syncthreads();
            {
                constexpr int NV = EFX_HIST_BINS / 4 / SEL_NT;   // int4 per thread
                int4 x[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) x[k] = reinterpret_cast<const int4*>(lh)[k * SEL_NT + tid];
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const int i4 = k * SEL_NT + tid, line = i4 >> 3, w0 = (i4 & 7) * 4;
                    const int o0 = (w0 * (EFX_HIST_BINS / 32) + line) / PER;           // owner of word w0; the next words': + BINS / 32 / PER each
                    if (x[k].x) atomicAdd(&s_sub[o0], x[k].x);
                    if (x[k].y) atomicAdd(&s_sub[o0 + EFX_HIST_BINS / 32 / PER], x[k].y);
                    if (x[k].z) atomicAdd(&s_sub[o0 + 2 * (EFX_HIST_BINS / 32 / PER)], x[k].z);
                    if (x[k].w) atomicAdd(&s_sub[o0 + 3 * (EFX_HIST_BINS / 32 / PER)], x[k].w);
                }
            }
            __syncthreads();
            const int sum = s_sub[own];
            SEL_TL(2);
            int tot;
            const int before = block_excl_scan4(sum, s_scan, &tot);
            if (before < L.quota && L.quota <= before + sum) { s_own = own; s_want = L.quota - before; }      // exactly one thread
            __syncthreads();
            if (tid < 64) {
                // the owner's bins from the top, PER / 64 per lane
                const int want = s_want;
                int cb[PER / 64 > 0 ? PER / 64 : 1], bb[PER / 64 > 0 ? PER / 64 : 1], tot2 = 0;
#pragma unroll
                for (int j = 0; j < (PER >= 64 ? PER / 64 : 1); j++) {
                    const int idx = tid * (PER >= 64 ? PER / 64 : 1) + j;              // 0 = the top bin
                    bb[j] = s_own * PER + PER - 1 - idx;
                    cb[j] = idx < PER ? lh[efx_hist_word((uint32_t)bb[j])] : 0;
                    tot2 += cb[j];
                }
                int excl = wave_incl_scan(tot2) - tot2;
#pragma unroll
                for (int j = 0; j < (PER >= 64 ? PER / 64 : 1); j++) {
                    if (excl < want && want <= excl + cb[j]) { s_bin = bb[j]; s_inbin = cb[j]; s_rem = want - excl; }
                    excl += cb[j];
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            SelLevel& S = cnt->sel[l];
            const int kmin = none ? 0 : min(n, L.quota);
            cnt->sum.surv[l] = n; cnt->sum.cand[l] = nc;
            // the frame's corner density for the host: the context's next launch takes its sparse or its dense form by it (harris_kernel)
            if (l == 0 && blockIdx.y == 0 && T->host_hint) __hip_atomic_store(T->host_hint, nc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            efx_st(&S.pub[1], ((unsigned long long)(uint32_t)s_rem << 32) | (uint32_t)kmin);
            // word B has ARRIVED before word A leaves: an explicit wait for the store's acknowledgement.  (Until the last session of
            // round 6 a workgroup-scope release fence stood here and before the `done` counter below, "the stores have been
            // acknowledged" -- on this target that fence compiles to no wait for global stores at all; see the counting workgroups.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            efx_st(&S.pub[0], (1ull << 63) | ((unsigned long long)(uint32_t)(s_bin + 1) << 32) | (uint32_t)s_inbin);
        }
        SEL_TL(3);
        // the level's histogram is read by this workgroup only: it leaves it zero for the next frame (64 KB of coalesced stores behind
        // the publication, off everybody's path; withdrawing every survivor's count in the counting pass -- the first form -- put
        // 110 000 more atomics on a natural frame's few hundred hot lines: 12 us of its 19)
        if (n > 0) {
            int4* z = reinterpret_cast<int4*>(hist + (size_t)l * EFX_HIST_BINS);
#pragma unroll
            for (int k = 0; k < EFX_HIST_BINS / 4 / SEL_NT; k++) z[k * SEL_NT + tid] = make_int4(0, 0, 0, 0);
        }
#ifdef EFX_SEL_TIMING
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
            printf("select leader l0 ticks(10ns): rows %llu | hist %llu | scan+bins+publish %llu | n %d bin %d in_bin %d rem %d | start tick %llu\n",
                   sel_t[1] - sel_t[0], sel_t[2] - sel_t[1], sel_t[3] - sel_t[2], n, s_bin, s_inbin, s_rem, sel_t[0]);
#endif
        return;
    }

    // ---------------- counting workgroup: a lane per tile ----------------
    const int w = (int)blockIdx.x - nl;
    SEL_TC(0);
    const int gt = w * EFX_SEL_WG_TILES + tid;
    const bool valid = gt < T->total_tiles;
    int l = 0, tx = 0, ty = 0;
    if (valid) efx_tile_of(T, gt, l, tx, ty);
    const LevelDev& L = T->lv[l];
    unsigned start = 0; int sc = 0;
    if (valid && L.active) {
        start = min(hdr[gt].cand_start, (unsigned)L.cap);
        sc = min((int)min(hdr[gt].surv_count, (uint32_t)(EFX_TILE * EFX_TILE)), L.cap - (int)start);
    }
    const Corner* q = surv_all + L.cand_base + start;
    // the first survivors are requested before the wait for the leader: most tiles have no more
    Corner s[SEL_ILP];
#pragma unroll
    for (int u = 0; u < SEL_ILP; u++) s[u] = q[min(u, max(sc - 1, 0))];
    SEL_TC(1);
    int bin = EFX_HIST_BINS, in_bin = 0, rem = 0;
    if (valid) {
        const unsigned long long a = efx_wait_word(&cnt->sel[l].pub[0]);
        bin = (int)((a >> 32) & 0xffffu) - 1; in_bin = (int)(uint32_t)a;
        rem = (int)(efx_ld(&cnt->sel[l].pub[1]) >> 32);
    }
    SEL_TC(2);
    const bool all_bin = rem == in_bin, listed = !all_bin && in_bin <= EFX_SEL_LIST_CAP;
    unsigned long long* lst = sel_list + (size_t)l * EFX_SEL_LIST_CAP;
    const int lane64 = tid & 63;
    int c = 0;
    {
        int scmax = sc;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) scmax = max(scmax, __shfl_xor(scmax, d, 64));
        for (int j0 = 0; j0 < scmax; j0 += SEL_ILP) {
            if (j0 > 0) {
#pragma unroll
                for (int u = 0; u < SEL_ILP; u++) s[u] = q[min(j0 + u, max(sc - 1, 0))];       // unconditional loads: one round trip per step
            }
#pragma unroll
            for (int u = 0; u < SEL_ILP; u++) {
                const unsigned long long key = efx_select_key(s[u].xy, s[u].resp);
                const int kb = j0 + u < sc ? (int)(key >> TOPSH) : -2;
                if (kb > bin || (kb == bin && all_bin)) c++;
                // the keys of the threshold's bin go to the level's list: one returning atomic per level and wave step, not per key
                // (a natural frame has ~500 of them: 5 us of same-word atomics)
                const bool app = kb == bin && listed;
                unsigned long long am = __ballot(app);
                while (am != 0ull) {
                    // (the lanes of a wave may hold tiles of two levels where levels meet: one round per level)
                    const int lead = __ffsll((long long)am) - 1;
                    const int ll = __shfl(l, lead, 64);
                    const unsigned long long mine = __ballot(app && l == ll);
                    int base = 0;
                    if (lane64 == lead) base = atomicAdd(&cnt->sel[ll].list_n, __popcll(mine));
                    base = __shfl(base, lead, 64);
                    if (app && l == ll) {
                        const int pos = base + __popcll(mine & ((1ull << lane64) - 1ull));
                        if (pos < EFX_SEL_LIST_CAP) efx_st(&lst[pos], key);
                    }
                    am &= ~mine;
                }
            }
        }
    }
    if (bin == EFX_HIST_BINS) c = 0;
    if (valid) efx_st(&nsel[gt], (uint32_t)c);
    SEL_TC(3);
    // ---- which levels does this workgroup complete? ----
    if (tid < EFX_MAX_LEVELS) s_last[tid] = 0;
    // Every thread's device-scope stores (its tile's count, the list keys) must have ARRIVED before the workgroup adds itself to the
    // level's `done` counter: whoever completes the level reads them right behind that addition.  A workgroup-scope release fence
    // does not do it -- on gfx950 it waits for LDS traffic only, the stores may still be on their way when thread 0's atomic goes
    // out, and nothing orders the two on the way to memory.  Found by the determinism soak (tools/microbench/soak_diag.py): one
    // 8K frame in ~500 000 had a few tiles' counts read as the PREVIOUS frame's offsets by the workgroup that scans the level, so
    // every later tile of the level was emitted beyond the capacity, i.e. not at all (round6.md section 11).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int l0, l1, tx_, ty_;
        efx_tile_of(T, w * EFX_SEL_WG_TILES, l0, tx_, ty_);
        efx_tile_of(T, min(w * EFX_SEL_WG_TILES + EFX_SEL_WG_TILES - 1, T->total_tiles - 1), l1, tx_, ty_);
        for (int i = l0; i <= l1; i++)
            if (atomicAdd(&cnt->sel[i].done, 1) == T->lv[i].sel_wgs - 1) s_last[i] = 1;
    }
    __syncthreads();
    for (int fl = 0; fl < nl; fl++) {
        if (!s_last[fl]) continue;                              // workgroup-uniform
        // ---------------- last workgroup of level fl: threshold key, counts, scan ----------------
        SEL_TF(4);
        const LevelDev& F = T->lv[fl];
        const int ntiles = F.tiles_x * F.tiles_y;
        // every leader's share of N (the leaders wait for nobody): lane i asks for level i
        int base = 0, all = 0;
        {
            int k = 0;
            if (tid < nl) { (void)efx_wait_word(&cnt->sel[tid].pub[0]); k = (int)(uint32_t)efx_ld(&cnt->sel[tid].pub[1]); }
            int kb4 = tid < fl ? k : 0;
            (void)block_excl_scan4(kb4, s_scan, &base);
            (void)block_excl_scan4(k, s_scan + 4, &all);
        }
        const unsigned long long fa = efx_ld(&cnt->sel[fl].pub[0]);
        const int fbin = (int)((fa >> 32) & 0xffffu) - 1, fin = (int)(uint32_t)fa, frem = (int)(efx_ld(&cnt->sel[fl].pub[1]) >> 32);
        SEL_TF(5);
        unsigned long long thresh = 0ull;
        int m = 0;                                              // listed keys in s_keys
        bool slow = false;
        if (fbin == EFX_HIST_BINS) thresh = ~0ull;              // nothing is selected (no key reaches this value)
        else if (fbin < 0) thresh = 0ull;                       // every survivor is
        else if (frem == fin) thresh = (unsigned long long)fbin << TOPSH;      // every key of the bin is wanted: its lower edge
        else if (fin <= EFX_SEL_LIST_CAP) {
            m = min(efx_ld(&cnt->sel[fl].list_n), EFX_SEL_LIST_CAP);          // == fin
            const unsigned long long* gl = sel_list + (size_t)fl * EFX_SEL_LIST_CAP;
            {
                unsigned long long lk[EFX_SEL_LIST_CAP / SEL_NT];
#pragma unroll
                for (int r = 0; r < EFX_SEL_LIST_CAP / SEL_NT; r++) lk[r] = efx_ld(&gl[min(r * SEL_NT + tid, EFX_SEL_LIST_CAP - 1)]);
#pragma unroll
                for (int r = 0; r < EFX_SEL_LIST_CAP / SEL_NT; r++) if (r * SEL_NT + tid < m) s_keys[r * SEL_NT + tid] = lk[r];
            }
            __syncthreads();
            if (m <= SEL_NT) {
                // rank by counting: the key with exactly frem - 1 larger keys (keys are unique)
                if (tid < m) {
                    const unsigned long long mine = s_keys[tid];
                    int larger = 0;
                    for (int j = 0; j < m; j++) larger += s_keys[j] > mine ? 1 : 0;
                    if (larger == frem - 1) s_thresh = mine;
                }
                __syncthreads();
                thresh = s_thresh;
            } else {
                // MSB-first radix select over the LDS list, 8-bit digits below the decided bits
                unsigned long long prefix = (unsigned long long)fbin;
                int decided = EFX_HIST_BITS, remaining = frem;
                while (decided < 64) {
                    const int width = (64 - decided) < 8 ? (64 - decided) : 8;
                    const int shift = 64 - decided - width;
                    s_sub[tid] = 0;
                    __syncthreads();
                    for (int i = tid; i < m; i += SEL_NT) {
                        const unsigned long long k = s_keys[i];
                        if ((k >> (64 - decided)) == prefix) atomicAdd(&s_sub[(int)((k >> shift) & ((1u << width) - 1))], 1);
                    }
                    __syncthreads();
                    if (tid < 64) {
                        // wave 0: lane t owns bins [255 - 4 t - 3, 255 - 4 t]
                        int loc[4]; int sum = 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) { loc[j] = s_sub[255 - (tid * 4 + j)]; sum += loc[j]; }
                        int before = wave_incl_scan(sum) - sum;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (before < remaining && remaining <= before + loc[j]) { s_bin = 255 - (tid * 4 + j); s_rem = remaining - before; s_m = loc[j]; }
                            before += loc[j];
                        }
                    }
                    __syncthreads();
                    prefix = (prefix << width) | (unsigned long long)s_bin;
                    const int cnt_bin = s_m;
                    remaining = s_rem;
                    decided += width;
                    __syncthreads();
                    if (remaining == cnt_bin) { prefix = decided < 64 ? (prefix << (64 - decided)) : prefix; decided = 64; }
                }
                thresh = prefix;
            }
        } else {
            // thousands of keys in one bin: 12-bit radix passes over the level's survivors, below the bits already decided
            slow = true;
            int* s_hist = s_cnt;                                 // 4096 bins
            unsigned long long prefix = (unsigned long long)fbin;
            int decided = EFX_HIST_BITS, remaining = frem;
            while (decided < 64) {
                const int width = (64 - decided) < 12 ? (64 - decided) : 12;
                const int shift = 64 - decided - width;
                for (int i = tid; i < 4096; i += SEL_NT) s_hist[i] = 0;
                __syncthreads();
                for (int t = tid; t < ntiles; t += SEL_NT) {
                    const TileHdr& th = hdr[F.tile_base + t];
                    const unsigned st = min(th.cand_start, (unsigned)F.cap);
                    const int n2 = min((int)min(th.surv_count, (uint32_t)(EFX_TILE * EFX_TILE)), F.cap - (int)st);
                    const Corner* q2 = surv_all + F.cand_base + st;
                    for (int j = 0; j < n2; j++) {
                        const unsigned long long k = efx_select_key(q2[j].xy, q2[j].resp);
                        if ((k >> (64 - decided)) == prefix) atomicAdd(&s_hist[(int)((k >> shift) & ((1u << width) - 1))], 1);
                    }
                }
                __syncthreads();
                // walk the bins from the top: thread t owns bins [hi - 16 t - 15, hi - 16 t]
                const int nb = 1 << width;
                int sum = 0;
                for (int j = 0; j < 16; j++) { const int b = nb - 1 - (tid * 16 + j); sum += b >= 0 ? s_hist[b] : 0; }
                int tot;
                int before = block_excl_scan4(sum, s_scan, &tot);
                for (int j = 0; j < 16; j++) {
                    const int b = nb - 1 - (tid * 16 + j);
                    const int cb = b >= 0 ? s_hist[b] : 0;
                    if (b >= 0 && before < remaining && remaining <= before + cb) { s_bin = b; s_rem = remaining - before; s_m = cb; }
                    before += cb;
                }
                __syncthreads();
                prefix = (prefix << width) | (unsigned long long)s_bin;
                const int cnt_bin = s_m;
                remaining = s_rem;
                decided += width;
                __syncthreads();
                if (remaining == cnt_bin) { prefix = decided < 64 ? (prefix << (64 - decided)) : prefix; decided = 64; }
            }
            thresh = prefix;                 // exactly `quota` keys are >= thresh (keys are unique)
        }
        SEL_TF(6);
        // ---- the tiles' output offsets: exclusive scan of the counts in canonical tile order, SEL_SEG tiles at a time in LDS ----
        TileHdr* hl = hdr + F.tile_base;
        const uint32_t* ns = nsel + F.tile_base;
        int running = 0;
        for (int seg = 0; seg < ntiles; seg += SEL_SEG) {
            const int nseg = min(SEL_SEG, ntiles - seg);
            __syncthreads();
            // coalesced device-scope loads, 32 independent ones per thread: all requested (indices clamped into the level) before the
            // first is used -- a load under a condition is waited for on the spot (measured: 10 us per segment instead of 1).  Tiles
            // 2 j and 2 j + 1 of the segment share word j
            {
                constexpr int NL = SEL_SEG / 2 / SEL_NT;        // words per thread
                uint32_t lo[NL], hi[NL];
#pragma unroll
                for (int r = 0; r < NL; r++) {
                    const int t = 2 * (r * SEL_NT + tid);
                    lo[r] = efx_ld(&ns[min(seg + t, ntiles - 1)]);      // (plain loads measured the same: 23.3 against 22.8 us)
                    hi[r] = efx_ld(&ns[min(seg + t + 1, ntiles - 1)]);
                }
#pragma unroll
                for (int r = 0; r < NL; r++) {
                    const int t = 2 * (r * SEL_NT + tid);
                    s_cnt[r * SEL_NT + tid] = (int)((t < nseg ? lo[r] : 0u) | ((t + 1 < nseg ? hi[r] : 0u) << 16));
                }
            }
            __syncthreads();
            // the list's selected keys (the counting pass left the threshold bin's keys out) join their tiles' counts
            for (int i = tid; i < m; i += SEL_NT) {
                const unsigned long long k = s_keys[i];
                if (k >= thresh) {
                    const uint32_t xy = 0xffffffffu - (uint32_t)k;      // the key's low word is ~xy
                    const int t = (int)((xy >> 16) >> 6) * F.tiles_x + (int)((xy & 0xffffu) >> 6) - seg;
                    if (t >= 0 && t < nseg) atomicAdd(&s_cnt[t >> 1], 1 << (16 * (t & 1)));
                }
            }
            if (slow) {
                for (int t = tid; t < nseg; t += SEL_NT) {
                    const TileHdr& th = hl[seg + t];
                    const unsigned st = min(th.cand_start, (unsigned)F.cap);
                    const int n2 = min((int)min(th.surv_count, (uint32_t)(EFX_TILE * EFX_TILE)), F.cap - (int)st);
                    const Corner* q2 = surv_all + F.cand_base + st;
                    int add = 0;
                    for (int j = 0; j < n2; j++) {
                        const unsigned long long k = efx_select_key(q2[j].xy, q2[j].resp);
                        add += ((int)(k >> TOPSH) == fbin && k >= thresh) ? 1 : 0;
                    }
                    if (add) atomicAdd(&s_cnt[t >> 1], add << (16 * (t & 1)));
                }
            }
            __syncthreads();
            constexpr int CW = SEL_SEG / 2 / SEL_NT;             // words per thread: tiles 2 CW tid .. 2 CW tid + 2 CW - 1
            int v[2 * CW], local = 0;
#pragma unroll
            for (int r = 0; r < CW; r++) {
                const uint32_t wv = (uint32_t)s_cnt[tid * CW + r];
                v[2 * r] = (int)(wv & 0xffffu); v[2 * r + 1] = (int)(wv >> 16);
                local += v[2 * r] + v[2 * r + 1];
            }
            int tot;
            int pre = running + base + block_excl_scan4(local, s_scan, &tot);
#pragma unroll
            for (int r = 0; r < 2 * CW; r++) {
                const int t = tid * 2 * CW + r;
                if (t < nseg) nsel[F.tile_base + seg + t] = (uint32_t)pre;      // the counts' words now hold the offsets (emit_kernel)
                pre += v[r];
            }
            running += tot;
        }
        SEL_TF(7);
        if (tid == 0) {
            cnt->thresh[fl] = thresh;
            cnt->level_out_base[fl] = base;
            cnt->sum.kept[fl] = running;
            if (fl == 0) {
                const int n = all < capacity ? all : capacity;
                cnt->level_out_base[nl] = all;
                cnt->sum.n_out = n;
                if (d_count) *d_count = n;
            }
        }
        SEL_TF(8);
#ifdef EFX_SEL_TIMING
        if (fl == 0 && blockIdx.y == 0 && tid == 0)
            printf("select last-of-l0 (wg %d) ticks(10ns): tile loads %llu | wait %llu | count %llu | done %llu | leaders %llu | rank %llu | scan+offsets %llu | end tick %llu\n",
                   w, sel_t[1] - sel_t[0], sel_t[2] - sel_t[1], sel_t[3] - sel_t[2], sel_t[4] - sel_t[3], sel_t[5] - sel_t[4], sel_t[6] - sel_t[5],
                   sel_t[7] - sel_t[6], sel_t[8]);
#endif
        __syncthreads();
    }
}

// Spec S7: deterministic double-precision atan2 (octant reduction + odd Taylor series), the same
// arithmetic as the CPU checker.  IEEE +,-,*,/ only; no contraction.
__device__ __forceinline__ float atan2_deg(int m01, int m10)
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
#pragma unroll
    for (int k = 23; k >= 0; k--) {
        const double ck = 1.0 / (double)(2 * k + 1);
        s = (k & 1 ? -ck : ck) + u2 * s;
    }
    double a = base + u * s;
    if (ay > ax) a = PI / 2 - a;
    if (x < 0) a = PI - a;
    if (y < 0) a = -a;
    if (a < 0) a = a + 2 * PI;
    return (float)(a * (180.0 / PI));
}

// ================================================================================================
// Kernel E: emit the selected survivors in canonical order: scalePoints (.cu:236-248), 5xN output rows,
// and the level-local float4 list for the describers (convertKeypointsKernel, .cu:250-263).
// Kernel F (angle_kernel) then fills in the IC angle (calcAngles, .cu:376-390), one wave per keypoint.
// ================================================================================================
// (Four tiles per 256-thread workgroup were measured: 10.6 against 9.6 us.  The kernel is a chain of dependent loads --
// header, threshold, survivors -- per wave, not a workgroup-launch-rate problem.)
#ifndef EMIT_TPW
#define EMIT_TPW 4
#endif
__global__ __launch_bounds__(64) void emit_kernel(const LevelTable* __restrict__ T, const TileHdr* __restrict__ hdr,
                                                  const Corner* __restrict__ surv_all, const Counters* __restrict__ cnt,
                                                  const uint8_t* __restrict__ img0, int pitch0, const uint8_t* __restrict__ pyramid,
                                                  size_t kps_pitch, int capacity,
                                                  float4* __restrict__ kp4, int* __restrict__ kp_level, const uint32_t* __restrict__ out_off_all,
                                                  const FrameOut out, const FrameStride fs)
{
    uint8_t* const kps = out.kps[blockIdx.y];
    out_off_all += blockIdx.y * fs.hdr;
    hdr += blockIdx.y * fs.hdr; surv_all += blockIdx.y * fs.cand; cnt += blockIdx.y; kp4 += blockIdx.y * fs.kp; kp_level += blockIdx.y * fs.kp;
    // EMIT_TPW tiles per wave, 64 / EMIT_TPW lanes each (round 3): a tile has three survivors on average and the kernel is a
    // chain of dependent loads per wave (tile word -> header -> survivors), so its time is the number of waves the chip must
    // cycle through: 25 500 one-tile waves took 3.1 rounds of the chip's 8192 wave slots
    constexpr int LPT = 64 / EMIT_TPW;                      // lanes per tile
    const int lane = threadIdx.x, part = lane / LPT, sub = lane % LPT;
    const int gt = EMIT_TPW * (int)blockIdx.x + part;
    const bool tile_ok = gt < T->total_tiles;
    int l = 0, tx = 0, ty = 0;
    if (tile_ok) efx_tile_of(T, gt, l, tx, ty);
    const LevelDev& L = T->lv[l];
    const bool act = tile_ok && L.active;               // (a void frame's threshold is above every key: nothing is selected)
    const unsigned start = act ? min(hdr[gt].cand_start, (unsigned)L.cap) : 0u;
    const int sc = act ? min((int)min(hdr[gt].surv_count, (uint32_t)(EFX_TILE * EFX_TILE)), L.cap - (int)start) : 0;
    const int out_off = act ? (int)out_off_all[gt] : 0;
    int sc_max = sc;
#pragma unroll
    for (int d = LPT; d < 64; d <<= 1) sc_max = max(sc_max, __shfl_xor(sc_max, d, 64));
#ifdef EFX_EMIT_DIAG
    // INVESTIGATION builds (tools/microbench/soak_diag.py with a capacity of 50 000): what this workgroup saw, in the unused columns
    // 40 000 + blockIdx.x of the keypoint matrix: row 0 the four tiles' survivor counts (a byte each), row 1 the first tile's output
    // offset, row 3 its raw header count, row 4 a marker.  (How the select_kernel hand-off race was found: round6.md section 11.)
    if (kps && capacity >= 40000 + (int)gridDim.x) {
        const int c0 = min(__shfl(sc, 0, 64), 255), c1 = min(__shfl(sc, LPT, 64), 255), c2 = min(__shfl(sc, 2 * LPT, 64), 255), c3 = min(__shfl(sc, 3 * LPT, 64), 255);
        if (lane == 0) {
            const size_t col = 40000 + (size_t)blockIdx.x;
            *reinterpret_cast<uint32_t*>(kps + 0 * kps_pitch + 4 * col) = (uint32_t)c0 | ((uint32_t)c1 << 8) | ((uint32_t)c2 << 16) | ((uint32_t)c3 << 24);
            *reinterpret_cast<uint32_t*>(kps + 1 * kps_pitch + 4 * col) = (uint32_t)out_off;
            *reinterpret_cast<uint32_t*>(kps + 3 * kps_pitch + 4 * col) = act ? hdr[gt].surv_count : 0xffffffffu;
            *reinterpret_cast<uint32_t*>(kps + 4 * kps_pitch + 4 * col) = 0x5eed0000u | (uint32_t)(blockIdx.x & 0xffffu);
        }
    }
#endif
    if (sc_max == 0) return;
    const unsigned long long thresh = cnt->thresh[l];
    const Corner* q = surv_all + L.cand_base + start;

    int running = 0;
    for (int i0 = 0; i0 < sc_max; i0 += LPT) {
        const int i = i0 + sub;
        Corner c; c.xy = 0; c.resp = 0.f;
        bool sel = false;
        if (i < sc) { c = q[i]; sel = efx_select_key(c.xy, c.resp) >= thresh; }
        const unsigned m = (unsigned)(__ballot(sel) >> (LPT * part)) & (unsigned)((1ull << LPT) - 1ull);     // this tile's lanes
        const int rank = __popc(m & ((1u << sub) - 1u));
        const int out = out_off + running + rank;
        running += __popc(m);
        if (sel && out < capacity) {
            const int x = c.xy & 0xffff, y = c.xy >> 16;
            const short sx = (short)(L.scale * (float)x + 0.5f);
            const short sy = (short)(L.scale * (float)y + 0.5f);
            if (kps) {
                *reinterpret_cast<uint32_t*>(kps + 0 * kps_pitch + 4 * (size_t)out) = (uint32_t)(uint16_t)sx | ((uint32_t)(uint16_t)sy << 16);
                *reinterpret_cast<float*>(kps + 1 * kps_pitch + 4 * (size_t)out) = c.resp;
                *reinterpret_cast<int*>(kps + 3 * kps_pitch + 4 * (size_t)out) = l;
                *reinterpret_cast<float*>(kps + 4 * kps_pitch + 4 * (size_t)out) = L.scale * (float)EFX_PATCH_SIZE;
            }
            kp4[out] = make_float4((float)x, (float)y, (float)EFX_PATCH_SIZE, 0.f);     // angle: angle_kernel
            kp_level[out] = l;
        }
    }
}

// ================================================================================================
// Kernel F: IC_Angle (cuda_efficient_features.cu:141-172), one wave per keypoint.  Lane = column dx of the
// radius-15 disc (31 columns), loop over the 31 rows: every row is one coalesced 31-byte read.  The integer
// moments are order-independent, the wave reduction gives exactly the reference's m_01 / m_10.
// ================================================================================================
#define ANGLE_KP 8              // keypoints per workgroup (two per wave)
// TAIL: the workgroup also turns its ANGLE_KP moments into angles (and records) itself, on ANGLE_KP lanes of one wave -- the
// form for small frames, where a second launch costs more than those ~1100 serial instructions (FHD: 7 us against 4.8 + 4.6)
template <bool TAIL>
__global__ __launch_bounds__(ANGLE_KP * 32) void angle_kernel(const LevelTable* __restrict__ T, int capacity,
                                                    const uint8_t* __restrict__ img0, int pitch0, const uint8_t* __restrict__ pyramid,
                                                    float4* __restrict__ kp4, const int* __restrict__ kp_level,
                                                    int want_kps, size_t kps_pitch,
                                                    Affine* __restrict__ aff, float bad_scale, float bad_reach, int bad_smax, int bad_sfixed,
                                                    const uint8_t* __restrict__ rec_img0, int rec_pitch0, const uint8_t* __restrict__ rec_levels, int rec_blurred,
                                                    const FrameSet F, const FrameOut out)
{
    const size_t fr = blockIdx.y;
    const int* const d_count = out.count[fr];
    uint8_t* const kps = want_kps ? out.kps[fr] : nullptr;
    img0 = F.in.img0[fr]; pyramid += fr * F.fs.pyramid; kp4 += fr * F.fs.kp; kp_level += fr * F.fs.kp;
    if (aff) aff += fr * F.fs.kp;
    // the image the describer's records refer to: the frame's raw levels, or its blurred copies
    rec_img0 = rec_blurred ? rec_img0 + fr * F.fs.blurred : img0;
    rec_levels = rec_blurred ? rec_levels + fr * F.fs.blurred : pyramid;
    // The intensity-centroid moments m01, m10 of every keypoint's patch; two keypoints per wave (31 of each 32 lanes hold
    // one patch column), ANGLE_KP per workgroup.  The moments are left in the .z / .w words of the keypoint's kp4 entry
    // (integer bits); angle_tail_kernel turns them into the angle.  (Until round 3 the double-precision atan2 / cos / sin
    // tail, ~1100 instructions, ran here on ANGLE_KP lanes of one wave per workgroup; now one LANE per keypoint of a dense
    // launch runs it: 6.0 -> 4.3 + 0.2 M wave-instructions, 19.0 -> 15.9 us.)
    // The kernel is bound by the bytes it makes the memory system fetch (31-byte rows at arbitrary alignment: 87 MB per 8K
    // frame for 38 MB of pixels).  Measured and dropped (round 3): the patch as range-checked dwords, 8 lanes along a row,
    // three v_dot4_u32_u8 per dword against (dx + 16), (dy + 16) and 1 inside the disc -- 2.1 M wave-instructions instead of
    // 4.3, but 20.6 us and the same frame rate; one lane per patch ROW -- 3.0 M, every load touches 62 cache lines: 33 us.
    // Neighbouring keypoints (canonical order) stay on the same XCD: their patches share L2 lines
    const int lane = threadIdx.x & 31;
    const int count = min(*d_count, capacity);
    const int ngroups = (count + ANGLE_KP - 1) / ANGLE_KP;   // the grid is sized for the capacity
    if ((int)blockIdx.x >= ngroups) return;
    const int group = xcd_chunked(blockIdx.x, ngroups);  // chunked over the groups that exist: all XCDs busy at any count
    const int kid = group * ANGLE_KP + (threadIdx.x >> 5);
    const bool act = kid < count;
    const float4 kp = act ? kp4[kid] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int l = act ? kp_level[kid] : 0;
    const uint8_t* img = l == 0 ? img0 : pyramid + T->lv[l].img_off;
    const int pitch = l == 0 ? pitch0 : T->lv[l].pitch;
    const int x = (int)kp.x, y = (int)kp.y;
    int m01 = 0, m10 = 0;
    if (act && lane < 31) {
        const int dx = lane - EFX_HALF_PATCH;
        const int adx = dx < 0 ? -dx : dx;
        const uint8_t* c = img + (size_t)y * pitch + x + dx;
        // all 31 row loads are issued first (the disc test only masks the accumulation), so one memory round trip
        int v[31];
#pragma unroll
        for (int i = 0; i < 31; i++) v[i] = c[(i - EFX_HALF_PATCH) * pitch];
#pragma unroll
        for (int i = 0; i < 31; i++) {
            // U_MAX, cuda_efficient_features.cu:143
            const int U_MAX[16] = { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3 };
            const int dy = i - EFX_HALF_PATCH;
            const int ady = dy < 0 ? -dy : dy;
            const int vv = adx <= U_MAX[ady] ? v[i] : 0;
            m10 += dx * vv;
            m01 += dy * vv;
        }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { m10 += __shfl_xor(m10, d, 64); m01 += __shfl_xor(m01, d, 64); }
    if (!TAIL) {
        if (act && lane == 0) *reinterpret_cast<int2*>(&kp4[kid].z) = make_int2(m10, m01);
        return;
    }
    // the double-precision atan2 (spec S7) is ~100 instructions: the 8 keypoints of the workgroup share one pass of it
    __shared__ int s_m[ANGLE_KP][2];
    if (lane == 0) { s_m[threadIdx.x >> 5][0] = m01; s_m[threadIdx.x >> 5][1] = m10; }
    __syncthreads();
    const int k8 = group * ANGLE_KP + threadIdx.x;
    if (threadIdx.x < ANGLE_KP && k8 < count) {
        const float angle = atan2_deg(s_m[threadIdx.x][0], s_m[threadIdx.x][1]);
        kp4[k8].w = angle;
        if (kps) *reinterpret_cast<float*>(kps + 2 * kps_pitch + 4 * (size_t)k8) = angle;
        if (aff) {
            // the BAD describer's record of this keypoint, while its angle is in a register (saves bad_affine_kernel's launch)
            float4 kq = kp4[k8]; kq.w = angle;
            const int lv = kp_level[k8];
            const LevelDev& L = T->lv[lv];
            aff[k8] = efx_bad_affine(kq, lv == 0 ? rec_img0 : rec_levels + L.img_off, lv == 0 ? rec_pitch0 : L.pitch, L.rows, L.cols, lv,
                                     bad_scale, bad_reach, bad_smax, bad_sfixed);
        }
    }
}

// The angle from the moments (calcAngles, .cu:376-390; the double-precision atan2 of spec S7), one lane per keypoint; the
// kp4 entry gets its size and angle, the caller's matrix its angle row and -- behind a BAD describer -- the keypoint's
// record (rectifyBoxes' double cos / sin, bad_affine.h), which saves bad_affine_kernel's launch.
__global__ __launch_bounds__(64) void angle_tail_kernel(const LevelTable* __restrict__ T, int capacity,
                                                        const uint8_t* __restrict__ img0, int pitch0, const uint8_t* __restrict__ pyramid,
                                                        float4* __restrict__ kp4, const int* __restrict__ kp_level,
                                                        int want_kps, size_t kps_pitch,
                                                        Affine* __restrict__ aff, float bad_scale, float bad_reach, int bad_smax, int bad_sfixed,
                                                        const uint8_t* __restrict__ rec_img0, int rec_pitch0, const uint8_t* __restrict__ rec_levels, int rec_blurred,
                                                        const FrameSet F, const FrameOut out)
{
    const size_t fr = blockIdx.y;
    const int* const d_count = out.count[fr];
    uint8_t* const kps = want_kps ? out.kps[fr] : nullptr;
    img0 = F.in.img0[fr]; pyramid += fr * F.fs.pyramid; kp4 += fr * F.fs.kp; kp_level += fr * F.fs.kp;
    if (aff) aff += fr * F.fs.kp;
    rec_img0 = rec_blurred ? rec_img0 + fr * F.fs.blurred : img0;
    rec_levels = rec_blurred ? rec_levels + fr * F.fs.blurred : pyramid;
    const int count = min(*d_count, capacity);
    const int k = (int)blockIdx.x * 64 + (int)threadIdx.x;
    if (k >= count) return;
    float4 kq = kp4[k];
    const float angle = atan2_deg(__float_as_int(kq.w), __float_as_int(kq.z));
    kq.z = (float)EFX_PATCH_SIZE; kq.w = angle;
    kp4[k] = kq;
    if (kps) *reinterpret_cast<float*>(kps + 2 * kps_pitch + 4 * (size_t)k) = angle;
    if (aff) {
        const int lv = kp_level[k];
        const LevelDev& L = T->lv[lv];
        aff[k] = efx_bad_affine(kq, lv == 0 ? rec_img0 : rec_levels + L.img_off, lv == 0 ? rec_pitch0 : L.pitch, L.rows, L.cols, lv,
                                bad_scale, bad_reach, bad_smax, bad_sfixed);
    }
}

__global__ void convert_keypoints_kernel(const uint8_t* __restrict__ kps, size_t kps_pitch, int n, float4* __restrict__ kp4)
{
    // convertKeypointsKernel, cuda_efficient_features.cu:250-263: size hard-wired to PATCH_SIZE
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t loc = *reinterpret_cast<const uint32_t*>(kps + 4 * (size_t)i);
    const float ang = *reinterpret_cast<const float*>(kps + 2 * kps_pitch + 4 * (size_t)i);
    const short x = (short)(loc & 0xffff), y = (short)(loc >> 16);
    kp4[i] = make_float4((float)x, (float)y, (float)EFX_PATCH_SIZE, ang);
}

__global__ void provided_keypoints_kernel(const LevelTable* __restrict__ T, const uint8_t* __restrict__ kps, size_t kps_pitch, int n,
                                         float4* __restrict__ kp4, int* __restrict__ kp_level)
{
    // spec S13: inverse of scalePoints (cuda_efficient_features.cu:236-248) for scale >= 1
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t loc = *reinterpret_cast<const uint32_t*>(kps + 4 * (size_t)i);
    const float ang = *reinterpret_cast<const float*>(kps + 2 * kps_pitch + 4 * (size_t)i);
    const int oct = *reinterpret_cast<const int*>(kps + 3 * kps_pitch + 4 * (size_t)i);
    const short x = (short)(loc & 0xffff), y = (short)(loc >> 16);
    const bool ok = oct >= 0 && oct < T->nlevels && T->lv[oct].rows > 0 && T->lv[oct].cols > 0;
    const float sc = ok ? T->lv[oct].scale : 1.f;
    kp4[i] = make_float4((float)(int)((float)x / sc + 0.5f), (float)(int)((float)y / sc + 0.5f), (float)EFX_PATCH_SIZE, ang);
    kp_level[i] = ok ? oct : 0;
}

__global__ void zero_invalid_descriptors_kernel(const LevelTable* __restrict__ T, const uint8_t* __restrict__ kps, size_t kps_pitch, int n,
                                                uint8_t* __restrict__ desc, size_t desc_pitch, int nbytes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int oct = *reinterpret_cast<const int*>(kps + 3 * kps_pitch + 4 * (size_t)i);
    const bool ok = oct >= 0 && oct < T->nlevels && T->lv[oct].rows > 0 && T->lv[oct].cols > 0;
    if (!ok) for (int b = 0; b < nbytes; b++) desc[(size_t)i * desc_pitch + b] = 0;
}

__global__ void copy2d_kernel(const uint8_t* __restrict__ src, size_t spitch, uint8_t* __restrict__ dst, size_t dpitch, int rows, int cols)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x < cols && y < rows) dst[(size_t)y * dpitch + x] = src[(size_t)y * spitch + x];
}

} // namespace

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
// Plan of the tower launch (pyramid_tower_kernel): which level it starts from, its tile size and its LDS layout.  The
// ranges are the kernel's own recurrence evaluated for every tile column / row, so the sizes are exact maxima.
#ifndef EFX_TOWER_MAX_PX
#define EFX_TOWER_MAX_PX 6000000      // larger frames: the row-walking chain (resize_rows_kernel).  Round 5, 4K (8.3 Mpx), reference protocol,
                                      // tower / chain: detect 0.143 / 0.145 ms, detectAndCompute BAD512 0.203 / 0.194, HashSIFT512 0.256 / 0.249; FHD: 0.084 / 0.095 detect
#endif
static inline int tower_src_host(int o, float f, int n) { const int v = (int)floorf((float)o * f); return v > n - 1 ? n - 1 : v; }

static bool plan_tower(const LevelTable& H, int last, const uint8_t* img0, int pitch0, bool no_tower, long long max_px, int nframes, TowerArgs* out, size_t* lds_out)
{
    if (last < 2) return false;
    // Measured (MI355X, one stream, sync per call): the tower beats the launch chain when the WHOLE pyramid is small
    // (FHD 0.166 -> 0.147 ms, 4K 0.270 -> 0.259 ms per detectAndCompute); fusing only the upper levels of a large frame
    // (8K levels 4..7) does not pay: those levels are as large as a 4K pyramid and the tower recomputes ~1.7x the pixels.
    // (a batch of frames is judged by ALL its pixels: the tower recomputes ~1.7x the pixels and only pays while the chain's three
    // launches would be latency-bound -- sixteen FHD frames are an 8K frame's worth of work)
    if ((long long)H.lv[0].rows * H.lv[0].cols * nframes > (max_px > 0 ? max_px : (long long)EFX_TOWER_MAX_PX)) return false;      // (EFX_TOWER_MAX_PX in the environment: tests)
    if (no_tower) return false;                               // EFX_NO_TOWER (tests): exercise the per-level kernels on small frames
    for (int s = 1; s <= last; s++) if (H.lv[s].fx > 3.5f || H.lv[s].fy > 3.5f) return false;     // resize_quad_win: source columns of neighbouring outputs within 8 bytes
    // Tile edge of the top level.  The kernel is a chain of dependent levels, so a workgroup's time hardly shrinks with
    // its tile; what counts is how the workgroups (1024 threads, at most two per CU) spread over the 256 CUs.  Measured
    // (tower time per edge, tools/microbench/tower_tt.sh): VGA 12: 14.1 us (180 WGs), 32: 19.2 (30); 720p 20: 16.4 (198),
    // 32: 19.7 (84), 12: 19.1 (510); FHD 28: 20.6 (220), 32: 22.6 (170), 24: 25.6 (299); 2.7K 36: 28.0 (252), 32: 32.9 (336);
    // 4K 36: 41 (510), 32: 48.4 (646 = a full round and a mostly empty one), 40: 44.5 (432).  Hence: the smallest edge
    // whose tiles fit ONE workgroup per CU; when that needs an edge above 40, the smallest edge from 32 up that fits two
    // per CU; otherwise 32.  EFX_TOWER_TT pins the edge (investigation).
    static const int tt_pin = getenv("EFX_TOWER_TT") ? atoi(getenv("EFX_TOWER_TT")) : 0;
    const LevelDev& Ltop = H.lv[last];
    auto tiles_of = [&](int tt) { return ((Ltop.cols + tt - 1) / tt) * ((Ltop.rows + tt - 1) / tt); };
    int cand[16], ncand = 0;
    if (tt_pin > 0) cand[ncand++] = tt_pin;
    else {
        for (int tt = 12; tt <= 40 && ncand == 0; tt += 4) if (tiles_of(tt) <= 256) cand[ncand++] = tt;
        for (int tt = 32; tt <= 48 && ncand == 0; tt += 4) if (tiles_of(tt) <= 512) cand[ncand++] = tt;
        if (ncand == 0 || cand[0] != 32) cand[ncand++] = 32;                // the fallback when the LDS plan of the first choice is too large
    }
    int pick = -1;
    TowerArgs best; size_t best_lds = 0;
    for (int ci = 0; ci < ncand && pick < 0; ci++)
    for (int s0 = 0; s0 == 0; s0++) {
        const LevelDev& Lt = H.lv[last];
        const int tt = cand[ci];
        const int ntx = (Lt.cols + tt - 1) / tt, nty = (Lt.rows + tt - 1) / tt;
        int maxw[EFX_MAX_LEVELS] = { 0 }, maxh[EFX_MAX_LEVELS] = { 0 };
        for (int dim = 0; dim < 2; dim++) {
            const int nt = dim == 0 ? ntx : nty;
            for (int t = 0; t < nt; t++) {
                const int ntop = dim == 0 ? Lt.cols : Lt.rows;
                int ownhi = std::min((t + 1) * tt, ntop);
                int lo = t * tt, hi = ownhi - 1;
                int* mx = dim == 0 ? maxw : maxh;
                mx[last] = std::max(mx[last], hi - lo + 1);
                for (int s = last - 1; s >= s0; s--) {
                    const float f = dim == 0 ? H.lv[s + 1].fx : H.lv[s + 1].fy;
                    const int n = dim == 0 ? H.lv[s].cols : H.lv[s].rows;
                    const int nownhi = t == nt - 1 ? n : tower_src_host(ownhi, f, n);
                    int nlo = tower_src_host(lo, f, n);
                    if (dim == 0) nlo &= ~3;
                    int nhi = std::min(tower_src_host(hi, f, n) + 1, n - 1);
                    nhi = std::max(nhi, nownhi - 1);
                    lo = nlo; hi = nhi; ownhi = nownhi;
                    mx[s] = std::max(mx[s], hi - lo + 1);
                }
            }
        }
        size_t bufA = 0, bufB = 0;
        int yrows = 0;
        for (int s = s0; s <= last; s++) {
            const size_t bytes = (size_t)((maxw[s] + 1 + 3) & ~3) * maxh[s];
            if (((s - s0) & 1) == 0) bufA = std::max(bufA, bytes); else bufB = std::max(bufB, bytes);
            if (s > s0) yrows = std::max(yrows, maxh[s]);
        }
        bufA = (bufA + 15) & ~(size_t)15; bufB = (bufB + 15) & ~(size_t)15;
        const size_t lds = bufA + bufB + 2 * (size_t)yrows * 16;
        if (lds > 78 * 1024) break;                           // this edge does not fit two workgroups per CU
        const uint8_t* src = s0 == 0 ? img0 : nullptr;
        const int spitch = s0 == 0 ? pitch0 : H.lv[s0].pitch;
        TowerArgs t;
        t.s0 = s0; t.top = last; t.tt = tt; t.tiles_x = ntx; t.tiles_y = nty;
        t.bufA = 0; t.bufB = (int)bufA; t.ytab_off = (int)(bufA + bufB); t.ytab_rows = yrows;
        t.aligned0 = s0 > 0 ? 1 : ((((uintptr_t)src | (uintptr_t)spitch) & 3u) == 0 && (H.lv[0].cols & 3) == 0);
        best = t; best_lds = lds; pick = ci;
    }
    if (best_lds == 0) return false;                          // too deep for this scale factor: one launch per level
    *out = best; *lds_out = best_lds;
    return true;
}

static hipError_t efx_launch_detect_impl(const DetectLaunch& a, hipStream_t stream, bool* forked_out);

hipError_t efx_launch_detect(const DetectLaunch& a, hipStream_t stream)
{
    bool forked = false;
    const hipError_t e = efx_launch_detect_impl(a, stream, &forked);
    // A failure between the fork and the join would leave the side stream's blur unjoined: a later inline blur of the same
    // context could then write `blurred` concurrently.  Wait for it here, on the failure path only.
    if (e != hipSuccess && forked && a.side) { (void)hipStreamSynchronize(a.side); (void)hipGetLastError(); }
    return e;
}

static hipError_t efx_launch_detect_impl(const DetectLaunch& a, hipStream_t stream, bool* forked_out)
{
    const LevelTable& H = *a.h_table;
    hipError_t e = hipSuccess;
    // frames of this launch (blockIdx.y of every kernel); a single-frame caller fills only the scalar fields
    const int B = a.nframes > 0 ? a.nframes : 1;
    if (B > EFX_MAX_BATCH) return hipErrorInvalidValue;
    FrameSet F;
    F.in = a.in; F.fs = a.fs;
    FrameOut out = a.out;
    if (B == 1) { F.in.img0[0] = a.img0; out.kps[0] = static_cast<uint8_t*>(a.d_keypoints); out.count[0] = a.d_count; }
    FramePyr fp;
    fp.in = F.in; fp.stride = a.fs.pyramid; fp.src_is_img0 = 0; fp.zrows = a.rows; fp.rows_stride = a.fs.rows;
    // alignment decisions about level 0 hold for every frame of the launch: the address bits of all the images, OR-ed
    uintptr_t img_bits = 0;
    for (int f = 0; f < B; f++) img_bits |= reinterpret_cast<uintptr_t>(F.in.img0[f]);
    // several waves per tile (harris_kernel, nms_kernel) only while the tiles of the whole launch do not fill the chip
    const long long launch_tiles = (long long)H.total_tiles * B;

    // pyramid chain (calcImagePyramid, .cpp:136-157): level s+1 from level s.  The large lower levels get one launch each;
    // the small upper levels (launch- and latency-bound one by one) are produced by ONE tower launch from level s0.
    int last = 0;                                              // last level with pixels
    while (last + 1 < H.nlevels && H.lv[last + 1].rows > 0 && H.lv[last + 1].cols > 0 && H.lv[last].rows > 0 && H.lv[last].cols > 0) last++;
    TowerArgs tw;
    size_t tw_lds = 0;
    const bool use_tower = plan_tower(H, last, reinterpret_cast<const uint8_t*>(img_bits), a.pitch0, a.knobs.no_tower != 0, a.knobs.tower_max_px, B, &tw, &tw_lds);
    const int chain_end = use_tower ? tw.s0 : last;           // levels 1 .. chain_end by the per-level kernel
    // the counters are zeroed by the first pyramid kernel; a single-level "pyramid" has none: memset command
    bool zeroed = false;
    if (chain_end == 0 && !use_tower) {
        e = hipMemsetAsync(a.counters, 0, sizeof(Counters) * (size_t)B, stream);
        if (e == hipSuccess && a.rows) e = hipMemsetAsync(a.rows, 0, sizeof(RowCtr) * (B > 1 ? a.fs.rows * (size_t)B : (size_t)H.total_rows), stream);
        if (e != hipSuccess) return e;
        zeroed = true;
    }
    for (int s = 0; s < chain_end; s++) {
        const LevelDev& L = H.lv[s];
        const LevelDev& N = H.lv[s + 1];
        const uint8_t* src = s == 0 ? a.img0 : a.pyramid + L.img_off;
        const int spitch = s == 0 ? a.pitch0 : L.pitch;
        // dword staging reads up to roundup4(cols) bytes of a row: always inside our own (padded) pyramid levels, inside a
        // caller's image only when its width is a multiple of 4 (otherwise the byte path)
        const int aligned = (((s == 0 ? img_bits : (uintptr_t)src) | (uintptr_t)spitch) & 3u) == 0 && (s > 0 || (L.cols & 3) == 0);
        // round 5: several levels per launch by waves walking down strips (resize_rows_kernel); EFX_NO_RESIZE_ROWS: the tiled kernels
        const RowsPlanLaunch* RP = (a.rows_plan && a.rplan && !a.knobs.no_resize_rows) ? &a.rows_plan[s] : nullptr;
        if (RP && RP->nlev > 0 && aligned && s + RP->nlev <= chain_end) {
            RowsArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.src = src; ra.spitch = spitch; ra.srows = L.rows; ra.scols = L.cols;
            for (int k = 0; k < RP->nlev; k++) {
                const LevelDev& D = H.lv[s + 1 + k];
                ra.lv[k].dst = a.pyramid + D.img_off; ra.lv[k].pitch = D.pitch; ra.lv[k].rows = D.rows; ra.lv[k].cols = D.cols;
                ra.lv[k].x = reinterpret_cast<const int*>(a.rplan + RP->x_off[k]); ra.lv[k].y = reinterpret_cast<const int4*>(a.rplan + RP->y_off[k]);
                ra.lv[k].W = RP->W[k];
            }
            ra.strips = reinterpret_cast<const int4*>(a.rplan + RP->strip_off); ra.chunks = reinterpret_cast<const int4*>(a.rplan + RP->chunk_off);
            ra.nstrips = RP->nstrips; ra.ntasks = RP->nstrips * RP->nchunks; ra.own = RP->own;
            const int nblk = (ra.ntasks + 3) / 4;
            const bool prof = a.prof.begin(100 + s, stream);
            Counters* zc = zeroed ? nullptr : a.counters;
            fp.src_is_img0 = s == 0;
            switch (RP->nlev) {
            case 1: hipLaunchKernelGGL(resize_rows_kernel<1>, dim3(nblk, B), dim3(256), 0, stream, ra, zc, H.total_rows, fp); break;
            case 2: hipLaunchKernelGGL(resize_rows_kernel<2>, dim3(nblk, B), dim3(256), 0, stream, ra, zc, H.total_rows, fp); break;
            case 3: hipLaunchKernelGGL(resize_rows_kernel<3>, dim3(nblk, B), dim3(256), 0, stream, ra, zc, H.total_rows, fp); break;
            default: hipLaunchKernelGGL(resize_rows_kernel<4>, dim3(nblk, B), dim3(256), 0, stream, ra, zc, H.total_rows, fp); break;
            }
            zeroed = true;
            a.prof.end(prof, 100 + s, stream);
            EFX_TRACE_POINT("resize_rows");
            s += RP->nlev - 1;
            continue;
        }
        const int sw = (int)ceilf((float)(EFX_TILE - 1) * N.fx) + 8, sh = (int)ceilf((float)(EFX_TILE - 1) * N.fy) + 3;
        const int lpitch = (sw + 3) & ~3;
        const int ytab_off = ((lpitch * sh) + 15) & ~15;          // per-row table behind the tile
        const size_t lds = (size_t)ytab_off + EFX_TILE * 16;
        if (lds > 64 * 1024) return hipErrorInvalidValue;
        const bool prof = a.prof.begin(100 + s, stream);
        const int ntiles = N.tiles_x * N.tiles_y;
        const bool no_stream = a.knobs.no_resize_stream != 0;     // EFX_NO_RESIZE_STREAM (tests): force the one-tile-per-workgroup kernel
        const ResizePlanLevel* R = a.rplan_lv ? &a.rplan_lv[s + 1] : nullptr;
        fp.src_is_img0 = s == 0;
        if (aligned && R && R->W != 0 && a.rplan && !no_stream) {
            // streamed variant: a grid the chip holds at once (8 workgroups of 256 threads per CU), a multiple of the 8 XCDs
            static std::atomic<int> s_slots_a{0};
            int s_slots = s_slots_a.load(std::memory_order_relaxed);
            if (s_slots == 0) {
                int dev = 0, cus = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
                s_slots = cus * 8;
                s_slots_a.store(s_slots, std::memory_order_relaxed);
            }
            const int per_xcd = std::min(s_slots / EFX_NXCD, (ntiles + EFX_NXCD - 1) / EFX_NXCD);
            hipLaunchKernelGGL(resize_stream_kernel, dim3(per_xcd * EFX_NXCD, B), dim3(256), RS_YTAB + EFX_TILE * 16, stream, src, spitch, L.rows, L.cols,
                               a.pyramid + N.img_off, N.pitch, N.rows, N.cols, N.tiles_x, N.tiles_y,
                               reinterpret_cast<const int*>(a.rplan + R->x_off), R->W, reinterpret_cast<const int4*>(a.rplan + R->y_off),
                               reinterpret_cast<const int4*>(a.rplan + R->t_off), zeroed ? nullptr : a.counters, H.total_rows, fp);
        } else
        hipLaunchKernelGGL((resize_kernel<256>), dim3(ntiles, B), dim3(256), lds, stream, src, spitch, L.rows, L.cols, aligned,
                           a.pyramid + N.img_off, N.pitch, N.rows, N.cols, N.fx, N.fy, N.tiles_x, N.tiles_y, lpitch, ytab_off,
                           zeroed ? nullptr : a.counters, H.total_rows, fp);
        zeroed = true;
        a.prof.end(prof, 100 + s, stream);
        EFX_TRACE_POINT("resize");
    }
    if (use_tower) {
        const bool prof = a.prof.begin(100 + tw.s0, stream);
        if (tw_lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pyramid_tower_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tw_lds);
        hipLaunchKernelGGL((pyramid_tower_kernel<1024>), dim3(tw.tiles_x * tw.tiles_y, B), dim3(1024), tw_lds, stream, a.d_table, a.img0,
                           a.pitch0, a.pyramid, tw, zeroed ? nullptr : a.counters, fp);
        zeroed = true;
        a.prof.end(prof, 100 + tw.s0, stream);
        EFX_TRACE_POINT("tower");
    }
    if (a.pyramid_only) return hipGetLastError();
    // BAD behind detectAndCompute: every active level blurred once, as an image (bad_kernel.hip: blur_levels_kernel); the
    // describer's records (angle kernels below) point at the copies.  The blur needs the pyramid only: on the call's stream right
    // here, or on the context's side stream beside the detector kernels (fork point: DetectLaunch::blur_fork)
    bool blur_pending = a.blurred != nullptr, forked = false;
    const int fork_point = (a.side && a.ev_fork && a.ev_join && a.blur_fork >= 1 && a.blur_fork <= 3) ? a.blur_fork : 0;
    auto launch_blur = [&](int point) -> hipError_t {
        if (!blur_pending || point != fork_point) return hipSuccess;
        blur_pending = false;
        hipStream_t st = stream;
        if (point > 0) {
            hipError_t e2 = hipEventRecord(a.ev_fork, stream);
            if (e2 == hipSuccess) e2 = hipStreamWaitEvent(a.side, a.ev_fork, 0);
            if (e2 != hipSuccess) return e2;
            st = a.side; forked = true; *forked_out = true;
        }
        hipError_t e2 = efx_launch_blur_levels(H, a.img0, a.pitch0, a.pyramid, a.blurred, a.blur0_pitch, a.blur_levels_off, a.prof, st, B, F.in, a.fs);
        if (e2 == hipSuccess && forked) e2 = hipEventRecord(a.ev_join, a.side);
        EFX_TRACE_POINT("blur");
        return e2;
    };
    e = launch_blur(0);
    if (e == hipSuccess) e = launch_blur(1);
    if (e != hipSuccess) return e;
    {
        const int aligned0 = ((img_bits | (uintptr_t)a.pitch0) & 3u) == 0;
        bool prof = a.prof.begin(0, stream);
        hipLaunchKernelGGL(fast_kernel, dim3(H.total_tiles, B), dim3(256), 0, stream, a.d_table, a.img0, a.pitch0, aligned0,
                           a.pyramid, a.threshold, a.mask, a.mask_pitch, a.slots, a.tcount, a.rows, a.hdr, a.knobs.dbg & 15, F);
        a.prof.end(prof, 0, stream);
        EFX_TRACE_POINT("fast");
        prof = a.prof.begin(1, stream);
        int pack_groups = 0;                       // packed launch (a.pack_harris): a workgroup per four tiles of a level
        for (int s = 0; s < H.nlevels; s++) pack_groups += H.lv[s].pack_groups;
        if (a.pack_harris && pack_groups > 0)
            hipLaunchKernelGGL(harris_packed_kernel, dim3(pack_groups, B), dim3(64), 0, stream, a.d_table, a.img0, a.pitch0, aligned0, a.pyramid,
                               a.slots, a.tcount, a.rows, a.cand, a.cmax, a.hdr, a.knobs.dbg & 15, F);
        else if (launch_tiles <= EFX_NMS_WIDE_TILES)
            hipLaunchKernelGGL(harris_kernel<4>, dim3(H.total_tiles, B), dim3(256), 0, stream, a.d_table, a.img0, a.pitch0, aligned0, a.pyramid,
                               a.slots, a.tcount, a.rows, a.cand, a.cmax, a.hdr, a.knobs.dbg & 15, F);
        else if (launch_tiles <= EFX_NMS_MID_TILES)
            hipLaunchKernelGGL(harris_kernel<EFX_NMS_MID_NW>, dim3(H.total_tiles, B), dim3(64 * EFX_NMS_MID_NW), 0, stream, a.d_table, a.img0, a.pitch0, aligned0, a.pyramid,
                               a.slots, a.tcount, a.rows, a.cand, a.cmax, a.hdr, a.knobs.dbg & 15, F);
        else
            hipLaunchKernelGGL(harris_kernel<1>, dim3(H.total_tiles, B), dim3(64), 0, stream, a.d_table, a.img0, a.pitch0, aligned0, a.pyramid,
                               a.slots, a.tcount, a.rows, a.cand, a.cmax, a.hdr, a.knobs.dbg & 15, F);
        a.prof.end(prof, 1, stream);
        EFX_TRACE_POINT("harris");
        e = launch_blur(2);
        if (e != hipSuccess) return e;
    }
    bool prof = a.prof.begin(2, stream);
    // several waves per tile when the tiles alone do not fill the chip (256 CUs x 32 waves)
    if (launch_tiles <= EFX_NMS_WIDE_TILES)
        hipLaunchKernelGGL(nms_kernel<4>, dim3(H.total_tiles, B), dim3(256), 0, stream, a.d_table, a.hdr, a.cand, a.cmax, a.surv,
                           a.rows, a.hist, a.counters, a.nonmax_radius, a.knobs.dbg >> 4, a.fs);
    else if (launch_tiles <= EFX_NMS_MID_TILES)
        hipLaunchKernelGGL(nms_kernel<EFX_NMS_MID_NW>, dim3(H.total_tiles, B), dim3(64 * EFX_NMS_MID_NW), 0, stream, a.d_table, a.hdr, a.cand, a.cmax, a.surv,
                           a.rows, a.hist, a.counters, a.nonmax_radius, a.knobs.dbg >> 4, a.fs);
    else
        hipLaunchKernelGGL(nms_kernel<1>, dim3(H.total_tiles, B), dim3(64), 0, stream, a.d_table, a.hdr, a.cand, a.cmax, a.surv,
                           a.rows, a.hist, a.counters, a.nonmax_radius, a.knobs.dbg >> 4, a.fs);
    a.prof.end(prof, 2, stream);
    EFX_TRACE_POINT("nms");
    e = launch_blur(3);
    if (e != hipSuccess) return e;
    prof = a.prof.begin(3, stream);
    hipLaunchKernelGGL(select_kernel, dim3(H.nlevels + (H.total_tiles + EFX_SEL_WG_TILES - 1) / EFX_SEL_WG_TILES, B), dim3(SEL_NT), 0, stream, a.d_table, a.hdr,
                       a.surv, a.rows, a.hist, a.sel_list, a.nsel, a.counters, a.capacity, out, a.fs);
    EFX_TRACE_POINT("select");
    hipLaunchKernelGGL(emit_kernel, dim3((H.total_tiles + EMIT_TPW - 1) / EMIT_TPW, B), dim3(64), 0, stream, a.d_table, a.hdr, a.surv, a.counters,
                       a.img0, a.pitch0, a.pyramid, a.kps_pitch, a.capacity, a.kp4, a.kp_level, a.nsel, out, a.fs);
    EFX_TRACE_POINT("emit");
    if (a.capacity > 0) {
        // the image the describer's records refer to: the raw levels, or their blurred copies (blur_levels_kernel above)
        const uint8_t* rec_img0 = a.blurred ? a.blurred : a.img0;
        const int rec_pitch0 = a.blurred ? a.blur0_pitch : a.pitch0;
        const uint8_t* rec_levels = a.blurred ? a.blurred + a.blur_levels_off : a.pyramid;
        const int rec_blurred = a.blurred ? 1 : 0;
        int nmax = 0;
        for (int s = 0; s < H.nlevels; s++) if (H.lv[s].active) nmax += H.lv[s].quota;
        if (nmax > a.capacity) nmax = a.capacity;
        if (nmax > 0 && use_tower) {
            // small frames (the ones whose pyramid is one tower launch): angles and records in the same launch
            hipLaunchKernelGGL(angle_kernel<true>, dim3((nmax + ANGLE_KP - 1) / ANGLE_KP, B), dim3(ANGLE_KP * 32), 0, stream, a.d_table, a.capacity,
                               a.img0, a.pitch0, a.pyramid, a.kp4, a.kp_level, 1, a.kps_pitch,
                               static_cast<Affine*>(a.bad_affine), a.bad_scale, a.bad_reach, a.bad_smax, a.bad_sfixed, rec_img0, rec_pitch0, rec_levels, rec_blurred, F, out);
        } else if (nmax > 0) {
            hipLaunchKernelGGL(angle_kernel<false>, dim3((nmax + ANGLE_KP - 1) / ANGLE_KP, B), dim3(ANGLE_KP * 32), 0, stream, a.d_table, a.capacity,
                               a.img0, a.pitch0, a.pyramid, a.kp4, a.kp_level, 0, 0, nullptr, 0.f, 0.f, 0, 0, nullptr, 0, nullptr, 0, F, out);
            hipLaunchKernelGGL(angle_tail_kernel, dim3((nmax + 63) / 64, B), dim3(64), 0, stream, a.d_table, a.capacity,
                               a.img0, a.pitch0, a.pyramid, a.kp4, a.kp_level, 1, a.kps_pitch,
                               static_cast<Affine*>(a.bad_affine), a.bad_scale, a.bad_reach, a.bad_smax, a.bad_sfixed, rec_img0, rec_pitch0, rec_levels, rec_blurred, F, out);
        }
    }
    a.prof.end(prof, 3, stream);
    EFX_TRACE_POINT("angle");
    if (forked) {                       // the describer (next on this stream) reads the blurred levels
        e = hipStreamWaitEvent(stream, a.ev_join, 0);
        if (e != hipSuccess) return e;
        *forked_out = false;            // joined
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return e;        // the host reads the summary (N, per-level counts) on demand: fetch_summary(), efx_api.cpp
}

#ifdef EFX_DEBUG_BUILD
// INVESTIGATION (rounds 1-3: the 16-process discrepancy): reran harris_kernel / nms_kernel on the buffers of
// the last frame.  The stages now leave per-row sums and a key histogram that a second run would add to again: not supported.
hipError_t efx_debug_rerun_stages(const DetectLaunch&, int, hipStream_t) { return hipErrorNotSupported; }
#endif

hipError_t efx_launch_convert_keypoints(const void* d_keypoints, size_t kps_pitch, int n, float4* kp4, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(convert_keypoints_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                       (const uint8_t*)d_keypoints, kps_pitch, n, kp4);
    return hipGetLastError();
}

hipError_t efx_launch_provided_keypoints(const LevelTable* d_table, const void* d_keypoints, size_t kps_pitch, int n, float4* kp4,
                                         int* kp_level, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(provided_keypoints_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_table, (const uint8_t*)d_keypoints,
                       kps_pitch, n, kp4, kp_level);
    return hipGetLastError();
}

hipError_t efx_launch_zero_invalid_descriptors(const LevelTable* d_table, const void* d_keypoints, size_t kps_pitch, int n,
                                               uint8_t* desc, size_t desc_pitch, int nbytes, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(zero_invalid_descriptors_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_table,
                       (const uint8_t*)d_keypoints, kps_pitch, n, desc, desc_pitch, nbytes);
    return hipGetLastError();
}

hipError_t efx_launch_copy2d(const uint8_t* src, size_t spitch, uint8_t* dst, size_t dpitch, int rows, int cols, hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(copy2d_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, stream, src, spitch, dst, dpitch, rows, cols);
    return hipGetLastError();
}
