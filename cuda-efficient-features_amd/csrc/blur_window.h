// blur_window.h -- the 7x7 sigma-2 Gaussian of detectAndCompute (spec S6; createGaussianFilter(CV_8UC1, 7x7, 2, 2,
// BORDER_REFLECT_101), cuda_efficient_features.cpp:193,305) applied to one keypoint's S x S window in LDS.
// Shared by the BAD and HashSIFT describers: neither needs a global blur pass over the pyramid.
//
// Geometry for a window of S pixels (S even):  G = ceil(S/8) groups of 8 outputs,  HP = 8G floats per hb row,
//   RP = S + 6 raw rows / valid raw columns (3-px apron),  RPB = 4 * ceil((S + 12) / 4) raw row pitch in bytes
//   (room for the dword-alignment slack).   LDS: raw RP x RPB bytes, hb (8G + 6) x HP floats.
//   efx_blur_lds_bytes(S) = bytes of [raw | hb] with hb 16-byte aligned behind raw.
#pragma once
#include "efx_device.h"

struct BlurGeom {
    int G, HP, RP, RPB;
    __host__ __device__ explicit BlurGeom(int S) : G((S + 7) >> 3), HP(((S + 7) >> 3) * 8), RP(S + 6), RPB(((S + 12 + 3) >> 2) << 2) {}
    __host__ __device__ size_t raw_bytes() const { return ((size_t)RP * RPB + 15) & ~(size_t)15; }
    __host__ __device__ size_t hb_bytes() const { return (size_t)(8 * G + 6) * HP * 4; }
};

#ifdef __HIPCC__
typedef float efx_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int efx_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

// Blurs the window [wx0, wx0 + S) x [wy0, wy0 + S) of `img`.  All NT threads of the workgroup must call it (it
// contains two barriers; the caller synchronises before reading what `store` wrote).  store(r, c, q0, q1) receives the
// blurred, rounded (half-even) and saturated pixels of row r at columns c and c + 1 (c even, r < S).
template <int NT, class Store>
__device__ __forceinline__ void efx_blur_window_lds(const uint8_t* __restrict__ img, int pitch, int rows, int cols, int wx0, int wy0,
                                                    int S, uint8_t* raw, float* hb, float taps0, float taps1, float taps2, float taps3,
                                                    int tid, Store store)
{
    const BlurGeom g_(S);
    const int G = g_.G, HP = g_.HP, RP = g_.RP, RPB = g_.RPB;
    // ---- raw window with a 3-px apron -> LDS.  Interior + 4-byte aligned images: aligned dword loads (row start
    //      rounded down to 4, byte offset `off` kept); otherwise bytes with REFLECT_101.
    const bool interior = (wx0 - 3 >= 0) && (wx0 + S + 3 <= cols) && (wy0 - 3 >= 0) && (wy0 + S + 3 <= rows);
    const bool fastld = interior && ((((uintptr_t)img) | (uintptr_t)pitch) & 3u) == 0;
    const int off = fastld ? ((wx0 - 3) & 3) : 0;
    if (fastld) {
        const int ndw = (off + RP + 3) >> 2;
        const int sh = ndw <= 16 ? 4 : (ndw <= 32 ? 5 : 6);      // lanes per row: 16 / 32 / 64
        const int j = tid & ((1 << sh) - 1), r0 = tid >> sh, rstep = NT >> sh;
        const uint8_t* base = img + (size_t)(wy0 - 3) * pitch + ((wx0 - 3) & ~3);
        for (int jj = j; jj < ndw; jj += (1 << sh))
            for (int r = r0; r < RP; r += rstep)
                *reinterpret_cast<uint32_t*>(raw + r * RPB + 4 * jj) =
                    *reinterpret_cast<const uint32_t*>(base + (size_t)r * pitch + 4 * jj);
    } else {
        for (int r = tid >> 6; r < RP; r += NT / 64) {
            const int gy = efx_reflect101(wy0 - 3 + r, rows);
            const uint8_t* src = img + (size_t)gy * pitch;
            for (int c = tid & 63; c < RP; c += 64) raw[r * RPB + c] = src[efx_reflect101(wx0 - 3 + c, cols)];
        }
    }
    __syncthreads();
    const float tp[7] = { taps0, taps1, taps2, taps3, taps2, taps1, taps0 };
    // ---- row pass: u8 -> float, acc = fma(tap_j, v_j, acc) for j = 0..6.  An item is 8 consecutive outputs of TWO
    //      adjacent rows, so every FMA is a v_pk_fma_f32 on a (row r, row r+1) register pair and every raw byte is
    //      read from LDS once.  RP = S + 6 is even.
    {
        const int nitems = (RP >> 1) * G;
        for (int it = tid; it < nitems; it += NT) {
            const int rp = it / G, g = it - rp * G;
            const uint32_t* wa = reinterpret_cast<const uint32_t*>(raw + (2 * rp) * RPB + 8 * g);
            const uint32_t* wb = wa + (RPB >> 2);
            const uint32_t a0 = wa[0], a1 = wa[1], a2 = wa[2], a3 = wa[3], a4 = wa[4];
            const uint32_t c0 = wb[0], c1 = wb[1], c2 = wb[2], c3 = wb[3], c4 = wb[4];
            const uint32_t ba[4] = { __builtin_amdgcn_alignbyte(a1, a0, off), __builtin_amdgcn_alignbyte(a2, a1, off),
                                     __builtin_amdgcn_alignbyte(a3, a2, off), __builtin_amdgcn_alignbyte(a4, a3, off) };
            const uint32_t bb[4] = { __builtin_amdgcn_alignbyte(c1, c0, off), __builtin_amdgcn_alignbyte(c2, c1, off),
                                     __builtin_amdgcn_alignbyte(c3, c2, off), __builtin_amdgcn_alignbyte(c4, c3, off) };
            efx_f32x2 v[14];
#pragma unroll
            for (int k = 0; k < 14; k++) {
                v[k].x = (float)((ba[k >> 2] >> (8 * (k & 3))) & 0xff);
                v[k].y = (float)((bb[k >> 2] >> (8 * (k & 3))) & 0xff);
            }
            efx_f32x2 o[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                efx_f32x2 acc = v[i] * tp[0];                        // == fma(tp[0], v, 0) exactly
#pragma unroll
                for (int jt = 1; jt < 7; jt++) acc = __builtin_elementwise_fma(v[i + jt], (efx_f32x2)(tp[jt]), acc);
                o[i] = acc;
            }
            float4* d0 = reinterpret_cast<float4*>(hb + (2 * rp) * HP + 8 * g);
            float4* d1 = reinterpret_cast<float4*>(hb + (2 * rp + 1) * HP + 8 * g);
            d0[0] = make_float4(o[0].x, o[1].x, o[2].x, o[3].x); d0[1] = make_float4(o[4].x, o[5].x, o[6].x, o[7].x);
            d1[0] = make_float4(o[0].y, o[1].y, o[2].y, o[3].y); d1[1] = make_float4(o[4].y, o[5].y, o[6].y, o[7].y);
        }
    }
    __syncthreads();
    // ---- column pass: float -> u8 (round half even, saturate: v_cvt_pk_u8_f32).  An item is 8 consecutive rows of
    //      TWO adjacent columns (S is even): 14 ds_read_b64, 56 v_pk_fma_f32.
    {
        const int ncp = S >> 1;
        const int nitems = ncp * G;
        for (int it = tid; it < nitems; it += NT) {
            const int rg = it / ncp, cp = it - rg * ncp;
            const int c = 2 * cp;
            efx_f32x2 v[14];
#pragma unroll
            for (int k = 0; k < 14; k++) v[k] = *reinterpret_cast<const efx_f32x2*>(hb + (8 * rg + k) * HP + c);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int r = 8 * rg + i;
                efx_f32x2 acc = v[i] * tp[0];
#pragma unroll
                for (int jt = 1; jt < 7; jt++) acc = __builtin_elementwise_fma(v[i + jt], (efx_f32x2)(tp[jt]), acc);
                const int q0 = (int)__builtin_amdgcn_cvt_pk_u8_f32(acc.x, 0, 0u);
                const int q1 = (int)__builtin_amdgcn_cvt_pk_u8_f32(acc.y, 0, 0u);
                if (r < S) store(r, c, q0, q1);
            }
        }
    }
}
#endif
