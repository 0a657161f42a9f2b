// glibc_sincosf.h -- cosf / sinf exactly as the reference's CPU HashSIFT gets them from libm.
//
// hash_sift.cpp:121-122 rotates the patch with cosf(theta) / sinf(theta).  Those are not correctly rounded: glibc >= 2.28
// (third-party dependency, absent from /root/reference; the build host has Ubuntu GLIBC 2.35) implements them with the
// ARM optimized-routines algorithm -- argument reduction by multiples of pi/2 and two polynomials, all in double, one
// final rounding to float -- whose result differs from round(cos(double)) for a few per cent of the arguments.  One ulp of
// cos moves a rectified patch pixel across an integer boundary now and then, which on a border keypoint of a
// high-contrast image changes a dozen histogram entries.  This header restates the published algorithm (constants and
// operation order of sincosf.h / sincosf_data.c, "fast" path) so that the device rotates patches with the very numbers
// the CPU code uses; tests/test_sincosf_restatement.py checks it against the host libm for EVERY float in [2^-13, 11):
// bit-identical.  The range stops at 11 (angles up to 630 degrees; callers fall back to a rounded double cos beyond):
// up to there the reduction's n * (pi/2) is exact (n <= 7, the constant has three trailing zero bits), so the result is
// the same whether libm was built with fused multiply-add (x86-64 ifunc variants) or not; beyond, the two builds differ.
// Compile without FP contraction (the Makefile passes -ffp-contract=off).
//
// Upstream: Arm Optimized Routines math/sincosf.h, sincosf_data.c -- Copyright (c) 2018, Arm Limited, SPDX MIT -- as
// adopted by the GNU C Library (sysdeps/ieee754/flt-32/s_sincosf.h, Copyright (C) 2018-2022 Free Software Foundation,
// Inc., LGPL-2.1-or-later).  Full notices: THIRD_PARTY_NOTICES.md next to this file.
#define EFX_GLIBC_SINCOSF_MAX 11.0f
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define EFX_SC_HD __host__ __device__ __forceinline__
#else
#define EFX_SC_HD static inline
#endif

struct efx_sincos_tab { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };

EFX_SC_HD uint32_t efx_sc_abstop12(float x)
{
    union { float f; uint32_t u; } v; v.f = x;
    return (v.u >> 20) & 0x7ffu;
}

// polynomial of one quadrant: sine for even n, cosine for odd n
EFX_SC_HD float efx_sc_poly(double x, double x2, const efx_sincos_tab& p, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = p.s2 + x2 * p.s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * p.s1;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = p.c3 + x2 * p.c4;
    const double c1 = p.c0 + x2 * p.c1;
    const double x6 = x4 * x2;
    const double c = c1 + x4 * p.c2;
    return (float)(c + x6 * c2);
}

// want_cos != 0: cosf(y), else sinf(y); valid for 0 <= y < EFX_GLIBC_SINCOSF_MAX
EFX_SC_HD float efx_glibc_sincosf(float y, int want_cos)
{
    const efx_sincos_tab t0 = { { 1.0, -1.0, -1.0, 1.0 }, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
                                0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
                                -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 };
    const efx_sincos_tab t1 = { { 1.0, -1.0, -1.0, 1.0 }, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
                                -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
                                -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 };
    double x = (double)y;
    if (efx_sc_abstop12(y) < efx_sc_abstop12(0x1.921FB6p-1f)) {           // |y| < pi/4
        if (efx_sc_abstop12(y) < efx_sc_abstop12(0x1p-12f)) return want_cos ? 1.0f : y;
        return efx_sc_poly(x, x * x, t0, want_cos ? 1 : 0);
    }
    // reduce_fast: the quadrant ends up in bits 24..31 of the scaled product
    const double r = x * t0.hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * t0.hpi;
    const double s = t0.sign[n & 3];
    return efx_sc_poly(x * s, x * x, (n & 2) ? t1 : t0, want_cos ? (n ^ 1) : n);
}
