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

// crossCheck: query i matches train j iff j is i's nearest train and i is j's nearest query
__global__ void crosscheck_kernel(const int* __restrict__ q2t, const int* __restrict__ t2q, int nq, int* __restrict__ match)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    const int j = q2t[2 * qi];
    match[qi] = (j >= 0 && t2q[2 * j] == qi) ? j : -1;
}

} // namespace

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
