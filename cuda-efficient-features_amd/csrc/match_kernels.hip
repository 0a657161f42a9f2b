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

#include "efx_device.h"
#include <stdlib.h>

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

// (dot product, train index) a better than b: larger dot (smaller distance), then lower index
__device__ __forceinline__ bool knn_better(int da, int ia, int db, int ib) { return da > db || (da == db && (unsigned)ia < (unsigned)ib); }

// NB: bytes of an expanded descriptor = bits (256 or 512); CB: 32-query column blocks per wave (the A fragment a wave
// reads from LDS serves CB MFMAs: the kernel is bound by those reads); NW waves per workgroup: NW * CB * 32 = 256 queries
template <int NB, int CB, int NW>
__global__ __launch_bounds__(NW * 64) void knn2_mfma_kernel(const uint8_t* __restrict__ xq, int nq, const uint8_t* __restrict__ xt, int nt,
                                                            int tiles_per_chunk, Best2* __restrict__ partial)
{
    constexpr int NT = NW * 64;
    constexpr int KS = NB / 32;                            // K steps of the 32x32x32 MFMA
    constexpr int LP = NB + 16;                            // LDS row pitch: rows 4 banks apart
    constexpr int NPF = 32 * NB / 16 / NT;                 // 16-byte loads per thread that stage a 32-row tile
    static_assert(NW * CB == 8 && NPF >= 1, "256 queries per workgroup");
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[32 * LP];
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
    int bd0[CB], bi0[CB], bd1[CB], bi1[CB];
#pragma unroll
    for (int c = 0; c < CB; c++) { bd0[c] = bd1[c] = -0x7fffffff; bi0[c] = bi1[c] = -1; }

    // a tile is 32 x NB bytes = 32 * NB / 16 sixteen-byte pieces, NPF per thread, rows contiguous in memory
    uint4 pf[NPF];
    auto fetch = [&](int tile) {
        const uint4* src = reinterpret_cast<const uint4*>(xt + (size_t)tile * 32 * NB);
#pragma unroll
        for (int j = 0; j < NPF; j++) pf[j] = src[tid + NT * j];
    };
    if (tile0 < tile1) fetch(tile0);
    for (int tile = tile0; tile < tile1; tile++) {
#pragma unroll
        for (int j = 0; j < NPF; j++) {
            const int piece = tid + NT * j, row = piece / (NB / 16), col = piece - row * (NB / 16);
            *reinterpret_cast<uint4*>(s_tile + row * LP + 16 * col) = pf[j];
        }
        __syncthreads();
        if (tile + 1 < tile1) fetch(tile + 1);             // in flight while this tile is multiplied
        const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        i32x16 acc[CB];
#pragma unroll
        for (int c = 0; c < CB; c++) acc[c] = zero;
        const uint8_t* arow = s_tile + li * LP + 16 * lh;
        // one A fragment ahead; the scheduling barrier keeps the compiler from hoisting all KS fragments (4 VGPRs each)
        i32x4 a = *reinterpret_cast<const i32x4*>(arow);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const i32x4 an = *reinterpret_cast<const i32x4*>(arow + 32 * (ks + 1 < KS ? ks + 1 : ks));
#pragma unroll
            for (int c = 0; c < CB; c++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[c][ks], acc[c], 0, 0, 0);
            a = an;
            if (CB > 1) __builtin_amdgcn_sched_barrier(0);
        }
        // C layout: column (query) = lane & 31, row (train) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): increasing with reg
        const int t0 = tile * 32 + 4 * lh;
        const bool partial_tile = tile * 32 + 32 > nt;     // padded trains must not compete
#pragma unroll
        for (int c = 0; c < CB; c++) {
            if (partial_tile) {
#pragma unroll
                for (int r = 0; r < 16; r++) if (t0 + (r & 3) + 8 * (r >> 2) >= nt) acc[c][r] = -0x7fffffff;
            }
            int m = -0x7fffffff;
#pragma unroll
            for (int r = 0; r < 16; r++) m = max(m, acc[c][r]);
            if (__ballot(m > bd1[c]) == 0ull) continue;    // nobody's second best is beaten: the usual case after the first tiles
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int d = acc[c][r], ti = t0 + (r & 3) + 8 * (r >> 2);
                // trains arrive in increasing index order, so a strict > keeps the lower index on ties
                if (d > bd0[c]) { bd1[c] = bd0[c]; bi1[c] = bi0[c]; bd0[c] = d; bi0[c] = ti; }
                else if (d > bd1[c]) { bd1[c] = d; bi1[c] = ti; }
            }
        }
        __syncthreads();                                   // every wave is done with the tile before it is overwritten
    }
    // the two row-halves of a query (lanes l and l + 32) merge their best two; dot -> Hamming distance
#pragma unroll
    for (int c = 0; c < CB; c++) {
        const int od0 = __shfl_xor(bd0[c], 32, 64), oi0 = __shfl_xor(bi0[c], 32, 64);
        const int od1 = __shfl_xor(bd1[c], 32, 64), oi1 = __shfl_xor(bi1[c], 32, 64);
        int d0 = bd0[c], i0 = bi0[c], d1 = bd1[c], i1 = bi1[c];
        const int cd[2] = { od0, od1 }, ci[2] = { oi0, oi1 };
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (ci[j] < 0) continue;
            if (i0 < 0 || knn_better(cd[j], ci[j], d0, i0)) { d1 = d0; i1 = i0; d0 = cd[j]; i0 = ci[j]; }
            else if (i1 < 0 || knn_better(cd[j], ci[j], d1, i1)) { d1 = cd[j]; i1 = ci[j]; }
        }
        const int q = q0 + 32 * c + li;
        if (lh == 0 && q < nq) {
            Best2 b;
            b.d0 = i0 >= 0 ? (NB - d0) >> 1 : 0x7fffffff; b.i0 = i0;
            b.d1 = i1 >= 0 ? (NB - d1) >> 1 : 0x7fffffff; b.i1 = i1;
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
    return ((size_t)((nq + 255) & ~255) + (size_t)((nt + 31) & ~31)) * nb;
}

hipError_t efx_launch_knn2_mfma(const uint8_t* query, size_t q_pitch, int nq, const uint8_t* train, size_t t_pitch, int nt,
                                int desc_bytes, void* scratch_x, void* scratch, int nchunks, int* idx, int* dist, hipStream_t stream)
{
    if (nq <= 0) return hipSuccess;
    const int nb = desc_bytes * 8, nq_pad = (nq + 255) & ~255, nt_pad = (nt + 31) & ~31;
    uint8_t* xq = static_cast<uint8_t*>(scratch_x);
    uint8_t* xt = xq + (size_t)nq_pad * nb;
    const size_t gq = (size_t)nq_pad * (desc_bytes / 2), gt = (size_t)nt_pad * (desc_bytes / 2);
    hipLaunchKernelGGL(expand_pm1_kernel, dim3((unsigned)((gq + 255) / 256)), dim3(256), 0, stream, query, q_pitch, nq, nq_pad, desc_bytes, xq);
    hipLaunchKernelGGL(expand_pm1_kernel, dim3((unsigned)((gt + 255) / 256)), dim3(256), 0, stream, train, t_pitch, nt, nt_pad, desc_bytes, xt);
    const int ntiles = nt_pad / 32, tpc = (ntiles + nchunks - 1) / nchunks;
    const int chunks = (ntiles + tpc - 1) / tpc;
    Best2* partial = static_cast<Best2*>(scratch);
    const dim3 grid(nq_pad / 256, chunks);
    // eight waves of 32 queries each (two column blocks per wave read half as much LDS per MFMA but run at half the
    // occupancy: 1.6 against 1.0 ms; a double-buffered tile with one barrier per step: 1.3 ms)
    if (desc_bytes == 32) hipLaunchKernelGGL((knn2_mfma_kernel<256, 1, 8>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
    else hipLaunchKernelGGL((knn2_mfma_kernel<512, 1, 8>), grid, dim3(512), 0, stream, xq, nq, xt, nt, tpc, partial);
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
