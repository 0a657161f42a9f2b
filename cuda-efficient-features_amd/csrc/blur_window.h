// blur_window.h -- the 7x7 sigma-2 Gaussian of detectAndCompute (spec S6; createGaussianFilter(CV_8UC1, 7x7, 2, 2,
// BORDER_REFLECT_101), cuda_efficient_features.cpp:193,305) applied to one keypoint's S x S window in LDS.
// Shared by the BAD and HashSIFT describers: neither needs a global blur pass over the pyramid.
//
// Item shapes (chosen for lane utilisation at S = 48 with 256 threads, tools/microbench/sweep_define.sh):
//   row pass     2 adjacent rows x RO = 6 consecutive outputs  -> 27 x 8 = 216 items (84 % of the lanes; 8 outputs: 63 %)
//   column pass  2 adjacent columns x CR = 5 consecutive rows  -> 24 x 10 = 240 items (94 %; 8 rows: 56 %)
// Geometry for a window of S pixels (S even):  GR = ceil(S/RO) row-pass groups, GC = ceil(S/CR) column-pass groups,
//   HP = RO*GR floats per hb row,  RP = S + 6 raw rows / valid raw columns (3-px apron),  RPB = raw row pitch in bytes
//   (room for the dword-alignment slack of the last group),  hb rows = CR*GC + 6 (rows >= RP are never written; what the
//   column pass computes from them lies below the window and is discarded).
//   LDS: raw RP x RPB bytes, then hb (16-byte aligned).
#pragma once
#include "efx_device.h"

#ifndef EFX_BLUR_RO
#define EFX_BLUR_RO 6
#endif
#ifndef EFX_BLUR_CR
#define EFX_BLUR_CR 5
#endif

struct BlurGeom {
    static constexpr int RO = EFX_BLUR_RO, CR = EFX_BLUR_CR;
    static constexpr int NV = RO + 6;                   // input pixels of one row of a row-pass item
    static constexpr int ND = (NV + 3 + 3) / 4;         // dwords that hold them at any byte alignment
    int GR, GC, HP, RP, RPB, HR;
    __host__ __device__ explicit BlurGeom(int S)
        : GR((S + RO - 1) / RO), GC((S + CR - 1) / CR), HP(RO * ((S + RO - 1) / RO)), RP(S + 6), RPB(0), HR(CR * ((S + CR - 1) / CR) + 6)
    {
        const int need_items = 4 * (((RO * (GR - 1) + 3) >> 2) + ND);     // last group, worst alignment
        const int need_load = 4 * ((3 + RP + 3) >> 2);                    // the staged row incl. alignment slack
        RPB = need_items > need_load ? need_items : need_load;
        if ((RPB & 63) == 0) RPB += 4;                                    // two rows apart must not be 32 banks apart
    }
    __host__ __device__ size_t raw_bytes() const { return ((size_t)RP * RPB + 15) & ~(size_t)15; }
    __host__ __device__ size_t hb_bytes() const { return (size_t)HR * HP * 4; }
};

#include <stdint.h>
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
// contains two barriers; the caller synchronises before reading what `store` wrote).  store(r0, i, c, pk) receives the
// blurred, rounded (half-even) and saturated pixels of row r0 + i (i a compile-time constant after unrolling) at columns
// c and c + 1 (c even, r0 + i < S) as the two low bytes of pk.
struct EfxNoOverlap { __device__ __forceinline__ void operator()() const {} };

// overlap(): work of the caller that does not depend on the window (a describer's per-keypoint set-up); it is placed between
// the issue of the window's global loads and their first use, i.e. it runs while the loads are in flight.
template <int NT, class Store, class Overlap = EfxNoOverlap>
__device__ __forceinline__ void efx_blur_window_lds(const uint8_t* __restrict__ img, int pitch, int rows, int cols, int wx0, int wy0,
                                                    int S, uint8_t* raw, float* hb, float taps0, float taps1, float taps2, float taps3,
                                                    int tid, Store store, Overlap overlap = Overlap())
{
    const BlurGeom g_(S);
    constexpr int RO = BlurGeom::RO, CR = BlurGeom::CR, NV = BlurGeom::NV, ND = BlurGeom::ND;
    static_assert(RO % 2 == 0, "hb rows are written as float2");
    const int GR = g_.GR, GC = g_.GC, HP = g_.HP, RP = g_.RP, RPB = g_.RPB;
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
        if (sh == 4 && RP <= 4 * rstep) {
            // the common shape (S = 48: 54 rows of 15 dwords, 16 rows per step): all of a thread's loads are issued before
            // the caller's overlap() work and before the first store.  Through a buffer resource: 32-bit offsets, no
            // predicates (rows / dwords past the window are inside the image or range-checked to zero, and not stored), and
            // -- unlike loads through the generic pointer -- counted in issue order, so that the overlap() work waits for
            // ITS operands only, not for these loads
            // (the last row counts up to roundup4(cols) only: a caller's allocation may end there -- ADVICE r3; bad_raw_kernel uses the same bound)
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, (rows - 1) * pitch + ((cols + 3) & ~3), 0x00020000);
            int goff = (wy0 - 3 + r0) * pitch + ((wx0 - 3) & ~3) + 4 * j;
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, goff, 0, 0);
                goff += rstep * pitch;
            }
            overlap();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = r0 + k * rstep;
                if (j < ndw && r < RP) *reinterpret_cast<uint32_t*>(raw + r * RPB + 4 * j) = v[k];
            }
        } else {
            overlap();
            for (int jj = j; jj < ndw; jj += (1 << sh))
                for (int r = r0; r < RP; r += rstep)
                    *reinterpret_cast<uint32_t*>(raw + r * RPB + 4 * jj) =
                        *reinterpret_cast<const uint32_t*>(base + (size_t)r * pitch + 4 * jj);
        }
    } else {
        overlap();
        for (int r = tid >> 6; r < RP; r += NT / 64) {
            const int gy = efx_reflect101(wy0 - 3 + r, rows);
            const uint8_t* src = img + (size_t)gy * pitch;
            for (int c = tid & 63; c < RP; c += 64) raw[r * RPB + c] = src[efx_reflect101(wx0 - 3 + c, cols)];
        }
    }
#if defined(BAD_DET_STOP)                                    // INVESTIGATION builds (bad_kernel.hip): stop after the window loads / the row pass
#define EFX_BLUR_STOP(n) do { if (BAD_DET_STOP == (n)) return; } while (0)
#else
#define EFX_BLUR_STOP(n) do { } while (0)
#endif
    EFX_BLUR_STOP(0);
    __syncthreads();
    const float tp[7] = { taps0, taps1, taps2, taps3, taps2, taps1, taps0 };
    // ---- row pass: u8 -> float, acc = fma(tap_j, v_j, acc) for j = 0..6.  An item is RO consecutive outputs of TWO
    //      adjacent rows, so every FMA is a v_pk_fma_f32 on a (row r, row r+1) register pair and every raw byte is
    //      read from LDS once.  RP = S + 6 is even.
    {
        const int nitems = (RP >> 1) * GR;
        for (int it = tid; it < nitems; it += NT) {
            const int rp = it / GR, g = it - rp * GR;
            const int sb = RO * g + off;                            // first input byte of the item within the raw row
            const uint32_t* wa = reinterpret_cast<const uint32_t*>(raw + (2 * rp) * RPB) + (sb >> 2);
            const uint32_t* wb = wa + (RPB >> 2);
            const int sh = sb & 3;
            uint32_t a[ND], c[ND], ba[ND - 1], bb[ND - 1];
#pragma unroll
            for (int k = 0; k < ND; k++) { a[k] = wa[k]; c[k] = wb[k]; }
#pragma unroll
            for (int k = 0; k < ND - 1; k++) {
                ba[k] = __builtin_amdgcn_alignbyte(a[k + 1], a[k], sh);
                bb[k] = __builtin_amdgcn_alignbyte(c[k + 1], c[k], sh);
            }
            efx_f32x2 v[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                v[k].x = (float)((ba[k >> 2] >> (8 * (k & 3))) & 0xff);
                v[k].y = (float)((bb[k >> 2] >> (8 * (k & 3))) & 0xff);
            }
            efx_f32x2 o[RO];
#pragma unroll
            for (int i = 0; i < RO; i++) {
                efx_f32x2 acc = v[i] * tp[0];                        // == fma(tp[0], v, 0) exactly
#pragma unroll
                for (int jt = 1; jt < 7; jt++) acc = __builtin_elementwise_fma(v[i + jt], (efx_f32x2)(tp[jt]), acc);
                o[i] = acc;
            }
            // o[i] = (row 2 rp, row 2 rp + 1) of column RO g + i: neighbouring columns of a row sit in different register
            // pairs, so the stores are written dword by dword (ds_write2_b32 takes any two registers; a float2 store would
            // cost a v_mov per value to line the registers up)
            // (written as ds_write2_b32 by hand: the compiler merges adjacent dword stores into ds_write_b64 + moves)
            const uint32_t a0 = (uint32_t)(uintptr_t)(hb + (2 * rp) * HP + RO * g);       // LDS byte address: low half of the generic address
            const uint32_t a1 = (uint32_t)(uintptr_t)(hb + (2 * rp + 1) * HP + RO * g);
#pragma unroll
            for (int i = 0; i < RO / 2; i++) {
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(a0), "v"(o[2 * i].x), "v"(o[2 * i + 1].x), "n"(2 * i), "n"(2 * i + 1) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(a1), "v"(o[2 * i].y), "v"(o[2 * i + 1].y), "n"(2 * i), "n"(2 * i + 1) : "memory");
            }
            // the compiler does not count hand-written LDS operations: without this the barrier below is not preceded by
            // a wait for them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    EFX_BLUR_STOP(1);
    __syncthreads();
    // ---- column pass: float -> u8 (round half even, saturate: v_cvt_pk_u8_f32).  An item is CR consecutive rows of
    //      TWO adjacent columns (S is even): CR + 6 ds_read_b64, 7 CR v_pk_fma_f32.
    {
        const int ncp = S >> 1;
        const int nitems = ncp * GC;
        for (int it = tid; it < nitems; it += NT) {
            const int rg = it / ncp, cp = it - rg * ncp;
            const int c = 2 * cp;
            efx_f32x2 v[CR + 6];
#pragma unroll
            for (int k = 0; k < CR + 6; k++) v[k] = *reinterpret_cast<const efx_f32x2*>(hb + (CR * rg + k) * HP + c);
#pragma unroll
            for (int i = 0; i < CR; i++) {
                const int r = CR * rg + i;
                efx_f32x2 acc = v[i] * tp[0];
#pragma unroll
                for (int jt = 1; jt < 7; jt++) acc = __builtin_elementwise_fma(v[i + jt], (efx_f32x2)(tp[jt]), acc);
                // both pixels through one pack chain: the store receives them packed (low byte: column c, next: c + 1;
                // the upper half is zero), with the row split into the item's first row and the unrolled offset
                const uint32_t pk = __builtin_amdgcn_cvt_pk_u8_f32(acc.y, 1, __builtin_amdgcn_cvt_pk_u8_f32(acc.x, 0, 0u));
                if (r < S) store(CR * rg, i, c, pk);
            }
        }
    }
}
#endif
