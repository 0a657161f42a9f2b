// match_kernels.hip -- brute-force Hamming matcher on the device (the consumer of the descriptors).
// Semantics of cv::BFMatcher(NORM_HAMMING) as the reference's samples use it:
//   knnMatch(query, train, k = 2)          samples/sample_image_sequence.cpp:81, 114-115
//   match() with crossCheck = true         samples/sample_feature_matching.cpp:99-101
// Distance = number of differing bits; ties are resolved towards the lower train index (the order in which a
// sequential scan with a strict `<` meets them).
//
// MI355X design: one lane owns one query descriptor (8 or 16 dwords in VGPRs); the train descriptors are read
// through the scalar cache (the address is wave-uniform, so every lane xors against SGPR operands) and each
// 32-bit word costs one v_xor + one v_bcnt_u32 (popcount with accumulate).  The train set is split into chunks
// along grid.y for occupancy; a second kernel merges the per-chunk best-2 lists.

#include <atomic>
#include "efx_device.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct Best2 { int d0, i0, d1, i1; };

template <int NW>   // dwords per descriptor: 8 (256 bit) or 16 (512 bit)
__global__ __launch_bounds__(256) void knn2_kernel(const uint8_t* __restrict__ query, size_t q_pitch, int nq,
                                                   const uint8_t* __restrict__ train, size_t t_pitch, int nt,
                                                   int chunk, Best2* __restrict__ partial)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * chunk, t1 = min(t0 + chunk, nt);
    uint32_t q[NW];
    const uint32_t* qp = reinterpret_cast<const uint32_t*>(query + (size_t)min(qi, nq - 1) * q_pitch);
#pragma unroll
    for (int k = 0; k < NW; k++) q[k] = qp[k];
    Best2 b; b.d0 = 0x7fffffff; b.i0 = -1; b.d1 = 0x7fffffff; b.i1 = -1;
    for (int t = t0; t < t1; t++) {
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(train + (size_t)t * t_pitch);   // wave-uniform
        int d = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) d += __popc(q[k] ^ tp[k]);
        if (d < b.d0) { b.d1 = b.d0; b.i1 = b.i0; b.d0 = d; b.i0 = t; }
        else if (d < b.d1) { b.d1 = d; b.i1 = t; }
    }
    if (qi < nq) partial[(size_t)blockIdx.y * nq + qi] = b;
}

__global__ void knn2_merge_kernel(const Best2* __restrict__ partial, int nq, int nchunks, int* __restrict__ idx, int* __restrict__ dist)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    Best2 b; b.d0 = 0x7fffffff; b.i0 = -1; b.d1 = 0x7fffffff; b.i1 = -1;
    // chunks are visited in train order and candidates inside a chunk are already ordered, so a strict `<`
    // keeps the lower train index on ties
    for (int c = 0; c < nchunks; c++) {
        const Best2 p = partial[(size_t)c * nq + qi];
        const int cd[2] = { p.d0, p.d1 }, ci[2] = { p.i0, p.i1 };
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (ci[j] < 0) continue;
            if (cd[j] < b.d0) { b.d1 = b.d0; b.i1 = b.i0; b.d0 = cd[j]; b.i0 = ci[j]; }
            else if (cd[j] < b.d1) { b.d1 = cd[j]; b.i1 = ci[j]; }
        }
    }
    idx[2 * qi] = b.i0; idx[2 * qi + 1] = b.i1;
    dist[2 * qi] = b.i0 >= 0 ? b.d0 : -1; dist[2 * qi + 1] = b.i1 >= 0 ? b.d1 : -1;
}

// ================================================================================================
// The same search on the int8 matrix cores (large query / train sets).  With every bit expanded to a byte of +-1, the dot
// product of two descriptors is nbits - 2 * hamming: the brute-force distance matrix is an int8 GEMM, exact in the i32
// accumulator, and v_mfma_i32_32x32x32_i8 does 32 x 32 x 32 of its terms per instruction (the popcount kernel above spends
// 2 VALU instructions per 32 bit-pairs of ONE query).  Layout: the TRAINS are the M operand (a 32-row tile staged in LDS per
// workgroup step, shared by the eight waves), the QUERIES the N operand (32 per wave, kept in registers for the wave's
// life), so that a lane of the C tile holds 16 trains of ONE query and keeps that query's best two in registers: no
// cross-lane traffic until the two row-halves of a query are merged at the very end.  A tile whose maximum does not beat
// any lane's second best (almost all of them, after the first few) is dismissed with 8 v_max3 per lane.
// ================================================================================================
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// bits -> bytes of +1 / -1 (bit j of a descriptor -> byte j; the order only has to be the same for queries and trains);
// rows [n, n_pad) are zero (they contribute nothing to a dot product; padded trains are masked by index)
__global__ void expand_pm1_kernel(const uint8_t* __restrict__ src, size_t pitch, int n, int n_pad, int nbytes, uint8_t* __restrict__ dst)
{
    const int groups = nbytes >> 1;                        // 16 bits -> 16 bytes per thread
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_pad * groups) return;
    const int row = (int)(i / groups), g = (int)(i - (size_t)row * groups);
    uint4 out = make_uint4(0u, 0u, 0u, 0u);
    if (row < n) {
        const uint8_t* p = src + (size_t)row * pitch + 2 * g;
        const uint32_t bits = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t ones = (((bits >> (4 * k)) & 15u) * 0x00204081u) & 0x01010101u;     // bit -> 0 / 1 per byte
            w[k] = ones | ((ones ^ 0x01010101u) * 0xffu);                                       // 1 -> 0x01, 0 -> 0xff
        }
        out = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4*>(dst + ((size_t)row * groups + g) * 16) = out;
}

// bits -> FP4 (E2M1) nibbles of +1.0 (0x2) / -1.0 (0xA), two per byte (round 4: the MX matrix-core path below); bit j of a
// descriptor -> nibble j, the same order for queries and trains; rows [n, n_pad) are zero nibbles (0.0)
__global__ void expand_fp4_kernel(const uint8_t* __restrict__ src, size_t pitch, int n, int n_pad, int nbytes, uint8_t* __restrict__ dst)
{
    const int groups = nbytes >> 1;                        // 16 bits -> 8 bytes per thread
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_pad * groups) return;
    const int row = (int)(i / groups), g = (int)(i - (size_t)row * groups);
    uint2 out = make_uint2(0u, 0u);
    if (row < n) {
        const uint8_t* p = src + (size_t)row * pitch + 2 * g;
        uint32_t w[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint32_t x = p[k];                              // bit j -> bit 4 j
            x = (x | (x << 12)) & 0x000f000fu;
            x = (x | (x << 6)) & 0x03030303u;
            x = (x | (x << 3)) & 0x11111111u;
            w[k] = 0x22222222u | ((~x & 0x11111111u) << 3);  // 1 -> 0x2 (+1.0), 0 -> 0xA (-1.0)
        }
        out = make_uint2(w[0], w[1]);
    }
    *reinterpret_cast<uint2*>(dst + ((size_t)row * groups + g) * 8) = out;
}

// (dot product, train index) a better than b: larger dot (smaller distance), then lower index
// The best-two bookkeeping compares INTEGER keys.  The int8 kernel's accumulators are integers already; the FP4 kernel's fp32
// accumulators start at KNN_FP4_BIAS instead of 0, so that every value is a positive float (a dot product is within +-512, and
// exact), and positive floats order like their bit patterns.  Integer max / compare have no NaN cases to canonicalise
// (fmaxf costs a v_max_f32 x, x per operand), and -- unlike an inline-asm v_max3_f32, round 4's first form of the group
// filter -- they are instructions the compiler knows read the MFMA's result registers: it places the wait states behind the
// last MFMA itself (the asm form read stale accumulators and lost ~1 in 300 best-two updates on tie-heavy sets:
// tools/microbench/match_fuzz.py found it, the parity tests' random sets had not).
#define KNN_FP4_BIAS 1024.f
__device__ __forceinline__ int knn_key(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int knn_key(int v) { return v; }
__device__ __forceinline__ int knn_max4(int a, int b, int c, int d)
{
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
template <class T> __device__ __forceinline__ bool knn_better(T da, int ia, T db, int ib) { return da > db || (da == db && (unsigned)ia < (unsigned)ib); }
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NBITS: descriptor bits (256 or 512); NB: bytes of an expanded descriptor = NBITS (int8) or NBITS / 2 (FP4); CB: 32-query column
// blocks per wave (the A fragment a wave reads from LDS serves CB MFMAs: the kernel is bound by those reads); NW waves per
// workgroup: NW * CB * 32 = 256 queries.
// FP4 (round 4): the same GEMM on the MX matrix cores -- v_mfma_f32_32x32x64_f8f6f4 with both operands FP4 (E2M1 holds +-1.0
// exactly; no block scales: the unscaled form), 64 terms of K per instruction at the int8 instruction's issue cost (32 cycles)
// and 16 bytes per lane and operand: half the MFMAs AND half the LDS bytes per descriptor pair.  The fp32 accumulator is exact
// (|dot| <= 512).  The lane layout of the operands is the int8 form's with two elements per byte (row = lane & 31, K range by
// lane >> 5), the C layout is the shape's; which K index a nibble stands for does not matter as long as queries and trains agree.
// DB (round 5): two LDS copies of the train tile, taking turns -- a step then has ONE barrier (behind its staging stores): when a
// wave passes the barrier of step t + 1 every wave has finished multiplying step t, so step t + 2 may overwrite that copy.
template <int NBITS, int CB, int NW, bool FP4, bool DB = false>
__global__ __launch_bounds__(NW * 64) void knn2_mfma_kernel(const uint8_t* __restrict__ xq, int nq, const uint8_t* __restrict__ xt, int nt,
                                                            int tiles_per_chunk, Best2* __restrict__ partial)
{
    constexpr int NT = NW * 64;
    constexpr int NB = FP4 ? NBITS / 2 : NBITS;
    constexpr int KS = NB / 32;                            // K steps: 32 bytes of a row per MFMA (32 int8 or 64 FP4 terms)
    constexpr int LP = NB + 16;                            // LDS row pitch: rows 4 banks apart
    constexpr int NPIECE = 32 * NB / 16;                   // 16-byte pieces of a 32-row tile
    constexpr int NPF = (NPIECE + NT - 1) / NT;            // ... per thread
    static_assert(NW * CB == 8, "256 queries per workgroup");
    typedef typename std::conditional<FP4, float, int>::type acc_t;
    typedef typename std::conditional<FP4, f32x16, i32x16>::type accv_t;
    const int LOWEST = FP4 ? 0 : -0x7fffffff;             // key below every real one (FP4: +0.0f)
    __shared__ __attribute__((aligned(16))) uint8_t s_tiles[(DB ? 2 : 1) * 32 * LP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 256 + wave * (32 * CB);    // this wave's queries: CB column blocks of 32
    i32x4 bq[CB][KS];
#pragma unroll
    for (int c = 0; c < CB; c++) {
        const i32x4* p = reinterpret_cast<const i32x4*>(xq + (size_t)(q0 + 32 * c + li) * NB + 16 * lh);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) bq[c][ks] = p[2 * ks];
    }
    const int ntiles = (nt + 31) >> 5;
    const int tile0 = blockIdx.y * tiles_per_chunk, tile1 = min(tile0 + tiles_per_chunk, ntiles);
    int bd0[CB], bd1[CB];                                  // keys (knn_key) of the best two
    int bi0[CB], bi1[CB];
#pragma unroll
    for (int c = 0; c < CB; c++) { bd0[c] = bd1[c] = LOWEST; bi0[c] = bi1[c] = -1; }

    // a tile is 32 x NB bytes = 32 * NB / 16 sixteen-byte pieces, NPF per thread, rows contiguous in memory
    uint4 pf[NPF];
    auto fetch = [&](int tile) {
        const uint4* src = reinterpret_cast<const uint4*>(xt + (size_t)tile * 32 * NB);
#pragma unroll
        for (int j = 0; j < NPF; j++) if (NPIECE % NT == 0 || tid + NT * j < NPIECE) pf[j] = src[tid + NT * j];
    };
    if (tile0 < tile1) fetch(tile0);
    for (int tile = tile0; tile < tile1; tile++) {
        uint8_t* s_tile = s_tiles + (DB ? ((tile - tile0) & 1) * 32 * LP : 0);
#pragma unroll
        for (int j = 0; j < NPF; j++) {
            const int piece = tid + NT * j, row = piece / (NB / 16), col = piece - row * (NB / 16);
            if (NPIECE % NT == 0 || piece < NPIECE) *reinterpret_cast<uint4*>(s_tile + row * LP + 16 * col) = pf[j];
        }
        __syncthreads();
        if (tile + 1 < tile1) fetch(tile + 1);             // in flight while this tile is multiplied
        accv_t acc[CB];
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[c][r] = FP4 ? (acc_t)KNN_FP4_BIAS : (acc_t)0;
        const uint8_t* arow = s_tile + li * LP + 16 * lh;
#ifndef EFX_MATCH_NO_PRELOAD
        if constexpr (FP4 && CB == 1) {
            // round 5: all KS fragments requested up front (32 more VGPRs: 114, still four waves per SIMD), the MFMAs back to back
            // as they arrive (s_waitcnt lgkmcnt(7) .. (0)) instead of two reads - two MFMAs - two reads: 0.425 -> 0.410 ms per
            // 40 000 x 40 000 x 512-bit call, same box (tools/microbench/match_ab.sh).  Holding three workgroups per CU instead
            // (85 registers) spills.  -DEFX_MATCH_NO_PRELOAD: the one-ahead form below
            i32x4 af[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) af[ks] = *reinterpret_cast<const i32x4*>(arow + 32 * ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const i32x8 a8 = { af[ks][0], af[ks][1], af[ks][2], af[ks][3], 0, 0, 0, 0 };
                const i32x8 b8 = { bq[0][ks][0], bq[0][ks][1], bq[0][ks][2], bq[0][ks][3], 0, 0, 0, 0 };
                acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[0], 4, 4, 0, 0, 0, 0);
            }
        } else
#endif
        {
        // one A fragment ahead; the scheduling barrier keeps the compiler from hoisting all KS fragments (4 VGPRs each)
        i32x4 a = *reinterpret_cast<const i32x4*>(arow);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 an = *reinterpret_cast<const i32x4*>(arow + 32 * (ks + 1 < KS ? ks + 1 : ks));
#pragma unroll
            for (int c = 0; c < CB; c++) {
                if constexpr (FP4) {
                    const i32x8 a8 = { a[0], a[1], a[2], a[3], 0, 0, 0, 0 };
                    const i32x8 b8 = { bq[c][ks][0], bq[c][ks][1], bq[c][ks][2], bq[c][ks][3], 0, 0, 0, 0 };
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[c], 4, 4, 0, 0, 0, 0);   // cbsz / blgp 4: FP4 E2M1; scales 0: the unscaled form
                } else {
                    acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[c][ks], acc[c], 0, 0, 0);
                }
            }
            a = an;
            if (CB > 1) __builtin_amdgcn_sched_barrier(0);
        }
        }
        // C layout: column (query) = lane & 31, row (train) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): increasing with reg
        const int t0 = tile * 32 + 4 * lh;
        const bool partial_tile = tile * 32 + 32 > nt;     // padded trains must not compete
#pragma unroll
        for (int c = 0; c < CB; c++) {
            int key[16];
#pragma unroll
            for (int r = 0; r < 16; r++) key[r] = knn_key(acc[c][r]);
            if (partial_tile) {
#pragma unroll
                for (int r = 0; r < 16; r++) if (t0 + (r & 3) + 8 * (r >> 2) >= nt) key[r] = LOWEST;
            }
            // four registers (four trains per lane) are looked at only if their maximum beats some lane's second best, and then
            // a register only if it does (knn2_fp4_kernel has the reason)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int m = knn_max4(key[4 * g], key[4 * g + 1], key[4 * g + 2], key[4 * g + 3]);
                if (__builtin_expect(__ballot(m > bd1[c]) == 0ull, 1)) continue;
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; r++) {
                    const int d = key[r];
                    if (__ballot(d > bd1[c]) == 0ull) continue;
                    const int ti = t0 + (r & 3) + 8 * (r >> 2);
                    // trains arrive in increasing index order, so a strict > keeps the lower index on ties
                    if (d > bd0[c]) { bd1[c] = bd0[c]; bi1[c] = bi0[c]; bd0[c] = d; bi0[c] = ti; }
                    else if (d > bd1[c]) { bd1[c] = d; bi1[c] = ti; }
                }
            }
        }
        if (!DB) __syncthreads();                          // every wave is done with the tile before it is overwritten
    }
    // the two row-halves of a query (lanes l and l + 32) merge their best two; dot -> Hamming distance
#pragma unroll
    for (int c = 0; c < CB; c++) {
        const int od0 = __shfl_xor(bd0[c], 32, 64), od1 = __shfl_xor(bd1[c], 32, 64);
        const int oi0 = __shfl_xor(bi0[c], 32, 64), oi1 = __shfl_xor(bi1[c], 32, 64);
        int d0 = bd0[c], d1 = bd1[c];
        int i0 = bi0[c], i1 = bi1[c];
        const int cd[2] = { od0, od1 }; const int ci[2] = { oi0, oi1 };
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (ci[j] < 0) continue;
            if (i0 < 0 || knn_better(cd[j], ci[j], d0, i0)) { d1 = d0; i1 = i0; d0 = cd[j]; i0 = ci[j]; }
            else if (i1 < 0 || knn_better(cd[j], ci[j], d1, i1)) { d1 = cd[j]; i1 = ci[j]; }
        }
        const int q = q0 + 32 * c + li;
        if (lh == 0 && q < nq) {
            Best2 b;
            // key -> dot -> distance: dot = bits - 2 * distance (exact in either accumulator)
            auto dot_of = [](int key) -> int { return FP4 ? (int)(__builtin_bit_cast(float, key) - KNN_FP4_BIAS) : key; };
            b.d0 = i0 >= 0 ? (NBITS - dot_of(d0)) >> 1 : 0x7fffffff; b.i0 = i0;
            b.d1 = i1 >= 0 ? (NBITS - dot_of(d1)) >> 1 : 0x7fffffff; b.i1 = i1;
            partial[(size_t)blockIdx.y * nq + q] = b;
        }
    }
}

// The FP4 kernel with TT train tiles (TT x 32 trains) per step: the MX instruction halved the matrix time of a tile (8 MFMAs of
// 32 cycles for 512 bits), and what was left -- staging, two barriers, the best-two bookkeeping of a tile -- then took three
// quarters of a step.  TT tiles share one staging pass and one pair of barriers, and their TT accumulators are independent
// MFMA chains (a wave's eight MFMAs on ONE accumulator wait for each other).  Eight waves x 32 queries, CB = 1.
template <int NBITS, int TT>
__global__ __launch_bounds__(512) void knn2_fp4_kernel(const uint8_t* __restrict__ xq, int nq, const uint8_t* __restrict__ xt, int nt,
                                                       int tiles_per_chunk, Best2* __restrict__ partial)
{
    constexpr int NT = 512;
    constexpr int NB = NBITS / 2;
    constexpr int KS = NB / 32;
    constexpr int LP = NB + 16;
    constexpr int NPIECE = TT * 32 * NB / 16;              // 16-byte pieces of a step's TT tiles (rows contiguous in memory)
    constexpr int NPF = (NPIECE + NT - 1) / NT;
    const int LOWEST = 0;                                  // key (knn_key: the bits of a positive float) below every real one
    // two copies of a step's tiles taking turns: ONE barrier per step (see knn2_mfma_kernel's DB form; round 5)
    constexpr bool DBUF = KS * TT <= 8;                    // (the 512-bit two-tile investigation form keeps one copy and two barriers: registers)
    __shared__ __attribute__((aligned(16))) uint8_t s_tiles[(DBUF ? 2 : 1) * TT * 32 * LP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 256 + wave * 32;
    i32x4 bq[KS];
    {
        const i32x4* p = reinterpret_cast<const i32x4*>(xq + (size_t)(q0 + li) * NB + 16 * lh);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) bq[ks] = p[2 * ks];
    }
    const int ntiles = (nt + 31) >> 5;                     // tiles that hold trains; the buffer is padded to whole steps
    const int tile0 = blockIdx.y * tiles_per_chunk, tile1 = min(tile0 + tiles_per_chunk, ntiles);
    int bd0 = LOWEST, bd1 = LOWEST;
    int bi0 = -1, bi1 = -1;
    uint4 pf[NPF];
    auto fetch = [&](int tile) {
        const uint4* src = reinterpret_cast<const uint4*>(xt + (size_t)tile * 32 * NB);
#pragma unroll
        for (int j = 0; j < NPF; j++) if (NPIECE % NT == 0 || tid + NT * j < NPIECE) pf[j] = src[tid + NT * j];
    };
    if (tile0 < tile1) fetch(tile0);
    for (int tile = tile0; tile < tile1; tile += TT) {
        uint8_t* s_tile = s_tiles + (DBUF ? (((tile - tile0) / TT) & 1) * (TT * 32 * LP) : 0);
#pragma unroll
        for (int j = 0; j < NPF; j++) {
            const int piece = tid + NT * j, row = piece / (NB / 16), col = piece - row * (NB / 16);
            if (NPIECE % NT == 0 || piece < NPIECE) *reinterpret_cast<uint4*>(s_tile + row * LP + 16 * col) = pf[j];
        }
        __syncthreads();
        if (tile + TT < tile1) fetch(tile + TT);           // in flight while these tiles are multiplied
        f32x16 acc[TT];
#pragma unroll
        for (int tt = 0; tt < TT; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tt][r] = KNN_FP4_BIAS;
        const uint8_t* arow = s_tile + li * LP + 16 * lh;
        if constexpr (DBUF) {
            // every fragment of the step requested up front (32 VGPRs), the MFMAs back to back as they arrive (round 5, as knn2_mfma_kernel)
            i32x4 af[KS][TT];
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int tt = 0; tt < TT; tt++) af[ks][tt] = *reinterpret_cast<const i32x4*>(arow + tt * 32 * LP + 32 * ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const i32x8 b8 = { bq[ks][0], bq[ks][1], bq[ks][2], bq[ks][3], 0, 0, 0, 0 };
#pragma unroll
                for (int tt = 0; tt < TT; tt++) {
                    const i32x8 a8 = { af[ks][tt][0], af[ks][tt][1], af[ks][tt][2], af[ks][tt][3], 0, 0, 0, 0 };
                    acc[tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[tt], 4, 4, 0, 0, 0, 0);
                }
            }
        } else {
            // (512 bits x two tiles, the EFX_MATCH_TT = 2 investigation form: sixteen fragments would spill)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const i32x8 b8 = { bq[ks][0], bq[ks][1], bq[ks][2], bq[ks][3], 0, 0, 0, 0 };
#pragma unroll
                for (int tt = 0; tt < TT; tt++) {
                    const i32x4 a = *reinterpret_cast<const i32x4*>(arow + tt * 32 * LP + 32 * ks);
                    const i32x8 a8 = { a[0], a[1], a[2], a[3], 0, 0, 0, 0 };
                    acc[tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[tt], 4, 4, 0, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TT; tt++) {                   // increasing train index: a strict > keeps the lower index on ties
            const int tb = (tile + tt) * 32;
            if (tb >= nt) break;                            // a padding tile of the last step (wave-uniform)
            const int t0 = tb + 4 * lh;
            int key[16];
#pragma unroll
            for (int r = 0; r < 16; r++) key[r] = knn_key(acc[tt][r]);
            if (tb + 32 > nt) {
#pragma unroll
                for (int r = 0; r < 16; r++) if (t0 + (r & 3) + 8 * (r >> 2) >= nt) key[r] = LOWEST;
            }
            // A register holds one train per lane (64 pairs).  It is looked at only if it beats SOME lane's second best.  (Until
            // round 4 the tile's maximum was tested once and all 16 registers then went through the update -- with the trains
            // split into chunks for occupancy a wave sees few thousand of them, two thirds of its tiles held a new best-two for
            // one of its 64 queries, and the update sequence was half of the kernel's time: 252 M VALU instructions for
            // 40 000 x 40 000 descriptors.  Then one compare + scalar branch per register -- whose common case, "nothing
            // here", was a TAKEN branch, 16 per tile.)  Now two levels: the maximum of four registers (v_max3_f32 + v_max_f32)
            // against the second bests, falling through when nothing beats them; the four registers one by one only otherwise.
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int m = knn_max4(key[4 * g], key[4 * g + 1], key[4 * g + 2], key[4 * g + 3]);
                if (__builtin_expect(__ballot(m > bd1) == 0ull, 1)) continue;
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; r++) {
                    const int d = key[r];
                    if (__ballot(d > bd1) == 0ull) continue;
                    const int ti = t0 + (r & 3) + 8 * (r >> 2);
                    if (d > bd0) { bd1 = bd0; bi1 = bi0; bd0 = d; bi0 = ti; }
                    else if (d > bd1) { bd1 = d; bi1 = ti; }
                }
            }
        }
        // (DBUF: no barrier here: the next step writes the OTHER copy, and this one is not written again before every wave has passed the next barrier)
        if (!DBUF) __syncthreads();
    }
    // the two row-halves of a query (lanes l and l + 32) merge their best two; dot -> Hamming distance
    {
        const int od0 = __shfl_xor(bd0, 32, 64), od1 = __shfl_xor(bd1, 32, 64);
        const int oi0 = __shfl_xor(bi0, 32, 64), oi1 = __shfl_xor(bi1, 32, 64);
        int d0 = bd0, d1 = bd1;
        int i0 = bi0, i1 = bi1;
        const int cd[2] = { od0, od1 }; const int ci[2] = { oi0, oi1 };
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (ci[j] < 0) continue;
            if (i0 < 0 || knn_better(cd[j], ci[j], d0, i0)) { d1 = d0; i1 = i0; d0 = cd[j]; i0 = ci[j]; }
            else if (i1 < 0 || knn_better(cd[j], ci[j], d1, i1)) { d1 = cd[j]; i1 = ci[j]; }
        }
        const int q = q0 + li;
        if (lh == 0 && q < nq) {
            Best2 b;
            b.d0 = i0 >= 0 ? (NBITS - (int)(__builtin_bit_cast(float, d0) - KNN_FP4_BIAS)) >> 1 : 0x7fffffff; b.i0 = i0;
            b.d1 = i1 >= 0 ? (NBITS - (int)(__builtin_bit_cast(float, d1) - KNN_FP4_BIAS)) >> 1 : 0x7fffffff; b.i1 = i1;
            partial[(size_t)blockIdx.y * nq + q] = b;
        }
    }
}

// crossCheck: query i matches train j iff j is i's nearest train and i is j's nearest query
__global__ void crosscheck_kernel(const int* __restrict__ q2t, const int* __restrict__ t2q, int nq, int* __restrict__ match)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    const int j = q2t[2 * qi];
    match[qi] = (j >= 0 && t2q[2 * j] == qi) ? j : -1;
}

} // namespace

// int8 matrix-core path: scratch_x holds the expanded (+-1 bytes) queries and trains, efx_knn2_mfma_scratch() bytes
size_t efx_knn2_mfma_scratch(int nq, int nt, int desc_bytes)
{
    const size_t nb = (size_t)desc_bytes * 8;
    return ((size_t)((nq + 255) & ~255) + (size_t)((nt + 127) & ~127)) * nb;      // trains padded to whole steps of up to four 32-row tiles
}

// tiles per step of the FP4 kernel: measured (40 000 x 40 000): a 256-bit tile is four MFMAs, two tiles amortise its barriers;
// a 512-bit step of two tiles is slower than two steps of one (0.467 against 0.436 ms)
static int knn2_fp4_tt(int desc_bytes)
{
    static const int tt_env = [] { const char* v = getenv("EFX_MATCH_TT"); return v ? atoi(v) : 0; }();     // INVESTIGATION knob
    return (tt_env == 1 || tt_env == 2) ? tt_env : (desc_bytes == 32 ? 2 : 1);      // anything else: the default
}

// Workgroups of the matrix-core kernel the chip holds at once (CUs x resident workgroups per CU, from the runtime's occupancy
// calculation for the variant efx_launch_knn2_mfma will launch): the host cuts the trains into as many chunks as fill ONE such
// round with (query block, chunk) pairs
int efx_knn2_mfma_resident_workgroups(int desc_bytes, int fp4)
{
    // per device (CU count and occupancy differ between devices), written by whichever thread asks first: atomics
    static std::atomic<int> cache[16][2][2][3];
    const int tt = fp4 ? knn2_fp4_tt(desc_bytes) : 0, b = desc_bytes == 32 ? 0 : 1;
    int dev0 = 0;
    if (hipGetDevice(&dev0) != hipSuccess) { (void)hipGetLastError(); dev0 = 0; }
    std::atomic<int>* slot = (dev0 >= 0 && dev0 < 16) ? &cache[dev0][b][fp4 ? 1 : 0][tt] : nullptr;
    if (slot) { const int c = slot->load(std::memory_order_relaxed); if (c > 0) return c; }
    const void* f;
    if (!fp4) f = desc_bytes == 32 ? reinterpret_cast<const void*>(&knn2_mfma_kernel<256, 1, 8, false>) : reinterpret_cast<const void*>(&knn2_mfma_kernel<512, 1, 8, false>);
    else if (tt == 2) f = desc_bytes == 32 ? reinterpret_cast<const void*>(&knn2_fp4_kernel<256, 2>) : reinterpret_cast<const void*>(&knn2_fp4_kernel<512, 2>);
    else {
        // the 512-bit variant the launcher will pick: double-buffered unless EFX_MATCH_NO_DB (read once per process there, too), so
        // that the chunks are sized by the occupancy of the kernel that runs
        static const bool no_db = getenv("EFX_MATCH_NO_DB") != nullptr;
        f = desc_bytes == 32 ? reinterpret_cast<const void*>(&knn2_mfma_kernel<256, 1, 8, true>)
          : no_db ? reinterpret_cast<const void*>(&knn2_mfma_kernel<512, 1, 8, true>) : reinterpret_cast<const void*>(&knn2_mfma_kernel<512, 1, 8, true, true>);
    }
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, 512, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (slot) slot->store(per_cu * cus, std::memory_order_relaxed);
    return per_cu * cus;
}

hipError_t efx_launch_knn2_mfma(const uint8_t* query, size_t q_pitch, int nq, const uint8_t* train, size_t t_pitch, int nt,
                                int desc_bytes, void* scratch_x, void* scratch, int nchunks, int* idx, int* dist, hipStream_t stream, int fp4)
{
    if (nq <= 0) return hipSuccess;
    const int nb = fp4 ? desc_bytes * 4 : desc_bytes * 8, nq_pad = (nq + 255) & ~255, nt_pad = (nt + 127) & ~127;
    uint8_t* xq = static_cast<uint8_t*>(scratch_x);
    uint8_t* xt = xq + (size_t)nq_pad * nb;
    const size_t gq = (size_t)nq_pad * (desc_bytes / 2), gt = (size_t)nt_pad * (desc_bytes / 2);
    if (fp4) {
        hipLaunchKernelGGL(expand_fp4_kernel, dim3((unsigned)((gq + 255) / 256)), dim3(256), 0, stream, query, q_pitch, nq, nq_pad, desc_bytes, xq);
        hipLaunchKernelGGL(expand_fp4_kernel, dim3((unsigned)((gt + 255) / 256)), dim3(256), 0, stream, train, t_pitch, nt, nt_pad, desc_bytes, xt);
    } else {
        hipLaunchKernelGGL(expand_pm1_kernel, dim3((unsigned)((gq + 255) / 256)), dim3(256), 0, stream, query, q_pitch, nq, nq_pad, desc_bytes, xq);
        hipLaunchKernelGGL(expand_pm1_kernel, dim3((unsigned)((gt + 255) / 256)), dim3(256), 0, stream, train, t_pitch, nt, nt_pad, desc_bytes, xt);
    }
    const int ntiles = (nt + 31) / 32;
    int tpc = (ntiles + nchunks - 1) / nchunks;
    tpc = (tpc + 3) & ~3;                                   // whole steps of the FP4 kernel (up to four tiles)
    const int chunks = (ntiles + tpc - 1) / tpc;
    Best2* partial = static_cast<Best2*>(scratch);
    const dim3 grid(nq_pad / 256, chunks);
    // eight waves of 32 queries each (two column blocks per wave read half as much LDS per MFMA but run at half the
    // occupancy: 1.6 against 1.0 ms; a double-buffered tile with one barrier per step: 1.3 ms)
    if (fp4) {
        const int tt = knn2_fp4_tt(desc_bytes);
        if (tt == 2) {
            if (desc_bytes == 32) hipLaunchKernelGGL((knn2_fp4_kernel<256, 2>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
            else hipLaunchKernelGGL((knn2_fp4_kernel<512, 2>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
        } else {
            // 512 bit: the double-buffered form (one barrier per step; round 5: 0.433 -> 0.414 ms per 40 000 x 40 000 call on one box;
            // EFX_MATCH_NO_DB: the two-barrier form, A/B).  Measured and dropped on the way: two query blocks per wave (half the LDS
            // fragment reads per MFMA, but 180 VGPRs with spills: 0.70 ms)
            static const bool no_db = getenv("EFX_MATCH_NO_DB") != nullptr;
            if (desc_bytes == 32) hipLaunchKernelGGL((knn2_mfma_kernel<256, 1, 8, true>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
            else if (!no_db) hipLaunchKernelGGL((knn2_mfma_kernel<512, 1, 8, true, true>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
            else hipLaunchKernelGGL((knn2_mfma_kernel<512, 1, 8, true>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
        }
    } else {
        if (desc_bytes == 32) hipLaunchKernelGGL((knn2_mfma_kernel<256, 1, 8, false>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
        else hipLaunchKernelGGL((knn2_mfma_kernel<512, 1, 8, false>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
    }
    hipLaunchKernelGGL(knn2_merge_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, partial, nq, chunks, idx, dist);
    return hipGetLastError();
}

hipError_t efx_launch_knn2(const uint8_t* query, size_t q_pitch, int nq, const uint8_t* train, size_t t_pitch, int nt,
                           int desc_bytes, void* scratch, int nchunks, int* idx, int* dist, hipStream_t stream)
{
    if (nq <= 0) return hipSuccess;
    const int chunk = (nt + nchunks - 1) / (nchunks > 0 ? nchunks : 1);
    Best2* partial = static_cast<Best2*>(scratch);
    const dim3 grid((nq + 255) / 256, nchunks);
    if (desc_bytes == 32)
        hipLaunchKernelGGL(knn2_kernel<8>, grid, dim3(256), 0, stream, query, q_pitch, nq, train, t_pitch, nt, chunk, partial);
    else
        hipLaunchKernelGGL(knn2_kernel<16>, grid, dim3(256), 0, stream, query, q_pitch, nq, train, t_pitch, nt, chunk, partial);
    hipLaunchKernelGGL(knn2_merge_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, partial, nq, nchunks, idx, dist);
    return hipGetLastError();
}

hipError_t efx_launch_crosscheck(const int* q2t, const int* t2q, int nq, int* match, hipStream_t stream)
{
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(crosscheck_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, q2t, t2q, nq, match);
    return hipGetLastError();
}
