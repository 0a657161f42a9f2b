#!/usr/bin/env python3
"""Generates tools/microbench/valu_rate.cpp: the VALU issue-rate microbenchmark (gfx950), round 4.

What changed against the round-3 version:
  * every operand is a hand-assigned VGPR, so the register BANK (index mod 4) of each source is known: the "clean" form
    gives the three sources of an instruction three different banks, the "same" form puts them all in one bank;
  * waves per SIMD are controlled (1 / 2 / 4 / 8: one 256-thread workgroup = one wave per SIMD, dynamic LDS sized so that
    exactly W workgroups fit a CU) and CHECKED from HW_ID of every wave;
  * cycles come from s_memtime deltas of the waves themselves (shader clock), the clock from s_memtime / s_memrealtime
    (constant-rate counter) measured DURING the run, and a hipEvent wall time is printed beside them;
  * a dependent single chain gives the latency of the instruction;
  * a sustained leg (many launches back to back) shows what the clock does under a second of dense VALU load.
Run:  python tools/microbench/gen_valu_rate.py && hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_rate.cpp -o tools/microbench/valu_rate
"""
import os

# x copies v4..v7 (banks 0..3), y copies v8..v11, z copies v12..v15, accumulators v16..v23 (banks 0 1 2 3 0 1 2 3).
# pairs: X v[4:5] v[6:7], Y v[8:9] v[10:11], accumulators v[16:17] .. v[30:31]
def regs(i, mode):
    b = i % 4
    if mode == "clean":
        return dict(i=i, d=f"v{16 + i}", a=f"v{4 + (b + 1) % 4}", b=f"v{8 + (b + 2) % 4}", c=f"v{12 + (b + 3) % 4}",
                    D=f"v[{32 + 2 * i}:{33 + 2 * i}]", A=f"v[{4 + 2 * ((i + 1) % 2)}:{5 + 2 * ((i + 1) % 2)}]",
                    B=f"v[{8 + 2 * (i % 2)}:{9 + 2 * (i % 2)}]")
    if mode == "same":
        return dict(i=i, d=f"v{16 + i}", a=f"v{4 + b}", b=f"v{8 + b}", c=f"v{12 + b}",
                    D=f"v[{32 + 2 * i}:{33 + 2 * i}]", A=f"v[{4 + 2 * (i % 2)}:{5 + 2 * (i % 2)}]",
                    B=f"v[{8 + 2 * (i % 2)}:{9 + 2 * (i % 2)}]")
    if mode == "dep":          # one chain, clean banks
        return regs(0, "clean")
    raise ValueError(mode)

CLASSES = [
    # name, template, group
    ("v_mov_b32", "v_mov_b32 {d}, {a}", "move"),
    ("v_add_u32", "v_add_u32 {d}, {a}, {d}", "int32"),
    ("v_sub_u32", "v_sub_u32 {d}, {d}, {a}", "int32"),
    ("v_and_b32", "v_and_b32 {d}, {a}, {d}", "int32"),
    ("v_xor_b32", "v_xor_b32 {d}, {a}, {d}", "int32"),
    ("v_or_b32", "v_or_b32 {d}, {a}, {d}", "int32"),
    ("v_not_b32", "v_not_b32 {d}, {d}", "int32"),
    ("v_subrev_u32", "v_subrev_u32 {d}, {a}, {d}", "int32"),
    ("v_add_u32 (literal)", "v_add_u32 {d}, 0x12345, {d}", "int32"),
    ("v_add_u32 (sgpr)", "v_add_u32 {d}, s47, {d}", "int32"),
    ("v_add_u32_e64", "v_add_u32_e64 {d}, {a}, {d}", "int32"),
    ("v_add_co_u32", "v_add_co_u32 {d}, vcc, {a}, {d}", "int32"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {d}, 1, {d}", "int32"),
    ("v_lshrrev_b32", "v_lshrrev_b32 {d}, 1, {d}", "int32"),
    ("v_ashrrev_i32", "v_ashrrev_i32 {d}, 1, {d}", "int32"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {a}, 2, {d}", "int32 3-op"),
    ("v_add_lshl_u32", "v_add_lshl_u32 {d}, {a}, {d}, 1", "int32 3-op"),
    ("v_xad_u32", "v_xad_u32 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_or3_b32", "v_or3_b32 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_cmp_lt_u32_e32", "v_cmp_lt_u32 vcc, {d}, {a}", "mask"),
    ("v_cndmask_b32_e32", "v_cndmask_b32 {d}, {a}, {d}, vcc", "mask"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {d}, {a}, {d}", "int mul"),
    ("v_min_u32", "v_min_u32 {d}, {a}, {d}", "int32"),
    ("v_max_i32", "v_max_i32 {d}, {a}, {d}", "int32"),
    ("v_add3_u32", "v_add3_u32 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {a}, 3, {d}", "int32 3-op"),
    ("v_and_or_b32", "v_and_or_b32 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_bfe_u32", "v_bfe_u32 {d}, {d}, 8, 8", "int32 3-op"),
    ("v_med3_i32", "v_med3_i32 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_max3_u32", "v_max3_u32 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_min3_u32", "v_min3_u32 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_perm_b32", "v_perm_b32 {d}, {a}, {d}, {b}", "int32 3-op"),
    ("v_alignbyte_b32", "v_alignbyte_b32 {d}, {a}, {d}, 1", "int32 3-op"),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {a}, {d}, 31", "int32 3-op"),
    ("v_sad_u8", "v_sad_u8 {d}, {a}, {b}, {d}", "int32 3-op"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {d}, {a}, {d}", "int32"),
    ("v_mbcnt_lo", "v_mbcnt_lo_u32_b32 {d}, {a}, {d}", "int32"),
    ("v_mul_i32_i24", "v_mul_i32_i24 {d}, {a}, {d}", "int mul"),
    ("v_mad_i32_i24", "v_mad_i32_i24 {d}, {a}, {b}, {d}", "int mul"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {a}, {b}, {d}", "int mul"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {a}, {d}", "int mul"),
    ("v_mul_hi_u32", "v_mul_hi_u32 {d}, {a}, {d}", "int mul"),
    ("v_mad_u64_u32", "v_mad_u64_u32 {D}, s[40:41], {a}, {b}, {D}", "int mul"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {a}, {d}", "packed 16"),
    ("v_pk_sub_i16", "v_pk_sub_i16 {d}, {d}, {a}", "packed 16"),
    ("v_pk_max_u16", "v_pk_max_u16 {d}, {a}, {d}", "packed 16"),
    ("v_pk_min_i16", "v_pk_min_i16 {d}, {a}, {d}", "packed 16"),
    ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 {d}, 1, {d}", "packed 16"),
    ("v_pk_mul_lo_u16", "v_pk_mul_lo_u16 {d}, {a}, {d}", "packed 16"),
    ("v_pk_mad_u16", "v_pk_mad_u16 {d}, {a}, {b}, {d}", "packed 16"),
    ("v_pk_mad_i16", "v_pk_mad_i16 {d}, {a}, {b}, {d}", "packed 16"),
    ("v_dot2c_i32_i16", "v_dot2c_i32_i16 {d}, {a}, {b}", "dot"),
    ("v_dot2_i32_i16", "v_dot2_i32_i16 {d}, {a}, {b}, {d}", "dot"),
    ("v_dot4c_i32_i8", "v_dot4c_i32_i8 {d}, {a}, {b}", "dot"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 {d}, {a}, {b}, {d}", "dot"),
    ("v_add_f32", "v_add_f32 {d}, {b}, {d}", "fp32"),
    ("v_mul_f32", "v_mul_f32 {d}, {a}, {d}", "fp32"),
    ("v_sub_f32", "v_sub_f32 {d}, {d}, {b}", "fp32"),
    ("v_max_f32", "v_max_f32 {d}, {a}, {d}", "fp32"),
    ("v_mul_f32 (literal)", "v_mul_f32 {d}, 0x3f800000, {d}", "fp32"),
    ("v_fma_f32", "v_fma_f32 {d}, {a}, {b}, {d}", "fp32"),
    ("v_fma_f32 (sgpr)", "v_fma_f32 {d}, {a}, s47, {d}", "fp32"),
    ("v_fma_f32 (neg/abs)", "v_fma_f32 {d}, -{a}, |{b}|, {d}", "fp32"),
    ("v_mad_f32? v_fma_mix_f32", "v_fma_mix_f32 {d}, {a}, {b}, {d}", "fp32"),
    ("v_fma_f16", "v_fma_f16 {d}, {a}, {b}, {d}", "fp16"),
    ("v_add_f16", "v_add_f16 {d}, {b}, {d}", "fp16"),
    ("v_pk_fma_f16", "v_pk_fma_f16 {d}, {a}, {b}, {d}", "fp16"),
    ("v_pk_add_f16", "v_pk_add_f16 {d}, {b}, {d}", "fp16"),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 {d}, {d}", "convert"),
    ("v_cvt_f16_f32", "v_cvt_f16_f32 {d}, {d}", "convert"),
    ("v_cvt_f32_f16", "v_cvt_f32_f16 {d}, {d}", "convert"),
    ("v_cvt_pk_f32_fp8", "v_cvt_pk_f32_fp8 {D}, {a}", "convert"),
    ("v_mov_b64", "v_mov_b64 {D}, {A}", "move"),
    ("v_accvgpr_write", "v_accvgpr_write_b32 a{i}, {a}", "move"),
    ("v_accvgpr_read", "v_accvgpr_read_b32 {d}, a{i}", "move"),
    ("v_fmac_f32", "v_fmac_f32 {d}, {a}, {b}", "fp32"),
    ("v_fmac_f32 (sgpr)", "v_fmac_f32 {d}, s47, {b}", "fp32"),
    ("v_fmac_f32 (body of 1024)", "v_fmac_f32 {d}, {a}, {b}", "instruction fetch"),
    ("v_fmac_f32 (sgpr, body 1024)", "v_fmac_f32 {d}, s47, {b}", "instruction fetch"),
    ("v_fma_f32 (body of 1024)", "v_fma_f32 {d}, {a}, {b}, {d}", "instruction fetch"),
    ("v_add_u32 (body of 1024)", "v_add_u32 {d}, {a}, {d}", "instruction fetch"),
    ("v_pk_mad_u16 (body of 1024)", "v_pk_mad_u16 {d}, {a}, {b}, {d}", "instruction fetch"),
    ("v_mad_f32_like(mul+add)", "v_mul_f32 {c}, {a}, {b}\\n v_add_f32 {d}, {c}, {d}", "fp32 (two instructions)"),
    ("v_pk_fma_f32", "v_pk_fma_f32 {D}, {A}, {B}, {D}", "packed fp32"),
    ("v_pk_mul_f32", "v_pk_mul_f32 {D}, {A}, {D}", "packed fp32"),
    ("v_pk_add_f32", "v_pk_add_f32 {D}, {B}, {D}", "packed fp32"),
    ("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 {d}, {d}", "convert"),
    ("v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 {d}, {d}", "convert"),
    ("v_cvt_f32_ubyte3", "v_cvt_f32_ubyte3 {d}, {d}", "convert"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {d}, {d}", "convert"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 {d}, {d}", "convert"),
    ("v_cvt_rpi_i32_f32", "v_cvt_rpi_i32_f32 {d}, {d}", "convert"),
    ("v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 {d}, {a}, 1, {d}", "convert"),
    ("v_cvt_pk_u16_u32", "v_cvt_pk_u16_u32 {d}, {a}, {d}", "convert"),
    ("v_rcp_f32", "v_rcp_f32 {d}, {d}", "transcendental"),
    ("v_sqrt_f32", "v_sqrt_f32 {d}, {d}", "transcendental"),
    ("v_exp_f32", "v_exp_f32 {d}, {d}", "transcendental"),
    ("v_min_u32_sdwa", "v_min_u32_sdwa {d}, {a}, {d} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", "sdwa / dpp"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa {d}, {a}, {d} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD", "sdwa / dpp"),
    ("v_add_u32_dpp", "v_add_u32_dpp {d}, {a}, {d} row_shr:1 row_mask:0xf bank_mask:0xf", "sdwa / dpp"),
    ("v_mov_b32_dpp", "v_mov_b32_dpp {d}, {d} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "sdwa / dpp"),
    ("v_cndmask_b32_e64", "v_cndmask_b32_e64 {d}, {a}, {d}, s[42:43]", "mask"),
    ("v_cmp_lt_u32_e64", "v_cmp_lt_u32_e64 s[44:45], {d}, {a}", "mask"),
    ("v_cmp+v_cndmask", "v_cmp_lt_u32_e64 s[44:45], {d}, {a}\\n v_cndmask_b32_e64 {d}, {a}, {d}, s[44:45]", "mask (two instructions)"),
    ("v_addc_co_u32_e64", "v_addc_co_u32_e64 {d}, s[44:45], {d}, {a}, s[42:43]", "mask"),
    ("v_readlane_b32", "v_readlane_b32 s46, {d}, 3", "lane"),
    ("v_writelane_b32", "v_writelane_b32 {d}, s47, 3", "lane"),
    ("v_add_f64", "v_add_f64 {D}, {B}, {D}", "fp64"),
    ("v_mul_f64", "v_mul_f64 {D}, {A}, {D}", "fp64"),
    ("v_fma_f64", "v_fma_f64 {D}, {A}, {B}, {D}", "fp64"),
]

MODES = ["clean", "same", "dep"]
UNROLL = 4


BIG = {"v_fmac_f32 (body of 1024)": 128, "v_fma_f32 (body of 1024)": 128, "v_add_u32 (body of 1024)": 128, "v_pk_mad_u16 (body of 1024)": 128,
       "v_fmac_f32 (sgpr, body 1024)": 128}


def body(tmpl, mode, unroll=UNROLL):
    lines = []
    for _ in range(unroll):
        for i in range(8):
            r = regs(i, mode)
            lines.append(tmpl.format(**r))
    return "\\n ".join(lines)


def kname(idx, mode):
    return f"k{idx}_{mode}"


HEADER = r'''// GENERATED by tools/microbench/gen_valu_rate.py -- do not edit.  VALU issue-rate microbenchmark for gfx950 (round 4).
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_rate.cpp -o tools/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
struct Rec { unsigned long long t0, t1, r0, r1; unsigned hwid, xcc; unsigned pad[2]; };
typedef void (*kern_t)(Rec*, int);
struct Entry { const char* name; const char* group; int per_body; kern_t clean, same, dep; };
#define PROLOGUE \
    "v_mov_b32 v4, %[x0]\n v_mov_b32 v5, %[x1]\n v_mov_b32 v6, %[x2]\n v_mov_b32 v7, %[x3]\n" \
    "v_mov_b32 v8, %[y0]\n v_mov_b32 v9, %[y1]\n v_mov_b32 v10, %[y2]\n v_mov_b32 v11, %[y3]\n" \
    "v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n" \
    "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n" \
    "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n" \
    "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n" \
    "v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n" \
    "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" \
    "s_mov_b64 s[42:43], 0x55555555\n s_mov_b32 s47, 7\n s_mov_b64 vcc, 0x33333333\n" \
    "s_mov_b32 s48, %[it]\n" \
    "s_getreg_b32 %[hw], hwreg(HW_REG_HW_ID)\n s_getreg_b32 %[xc], hwreg(HW_REG_XCC_ID)\n" \
    "s_barrier\n" \
    "s_memtime s[50:51]\n s_memrealtime s[52:53]\n s_waitcnt lgkmcnt(0)\n" \
    "1:\n "
#define EPILOGUE \
    "\n s_sub_u32 s48, s48, 1\n s_cmp_lg_u32 s48, 0\n s_cbranch_scc1 1b\n" \
    "s_memtime s[54:55]\n s_memrealtime s[56:57]\n s_waitcnt lgkmcnt(0)\n" \
    "s_mov_b64 %[t0], s[50:51]\n s_mov_b64 %[r0], s[52:53]\n s_mov_b64 %[t1], s[54:55]\n s_mov_b64 %[r1], s[56:57]\n"
#define CLOBBERS \
    "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", \
    "v23", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "vcc", "scc", "memory"
#define KERNEL(NAME, BODY) \
__global__ __launch_bounds__(256) void NAME(Rec* rec, int iters) { \
    unsigned long long t0, t1, r0, r1; unsigned hw, xc; \
    /* iters < 0: lane-dependent operands with random mantissas (x in [0.5, 1), y in [0, 2^-20)) instead of the constants 1.0 and 0 */ \
    const bool rnd = iters < 0; const int it_ = rnd ? -iters : iters; \
    unsigned h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u; \
    float x0 = 1.f, x1 = 1.f, x2 = 1.f, x3 = 1.f, y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f; \
    if (rnd) { \
        x0 = __uint_as_float(0x3f000000u | (h & 0x7fffffu)); h = h * 1664525u + 1013904223u; \
        x1 = __uint_as_float(0x3f000000u | (h & 0x7fffffu)); h = h * 1664525u + 1013904223u; \
        x2 = __uint_as_float(0x3f000000u | (h & 0x7fffffu)); h = h * 1664525u + 1013904223u; \
        x3 = __uint_as_float(0x3f000000u | (h & 0x7fffffu)); h = h * 1664525u + 1013904223u; \
        y0 = __uint_as_float(0x35000000u | (h & 0x7fffffu)); h = h * 1664525u + 1013904223u; \
        y1 = __uint_as_float(0x35000000u | (h & 0x7fffffu)); h = h * 1664525u + 1013904223u; \
        y2 = __uint_as_float(0x35000000u | (h & 0x7fffffu)); h = h * 1664525u + 1013904223u; \
        y3 = __uint_as_float(0x35000000u | (h & 0x7fffffu)); } \
    asm volatile(PROLOGUE BODY EPILOGUE \
        : [t0] "=&s"(t0), [t1] "=&s"(t1), [r0] "=&s"(r0), [r1] "=&s"(r1), [hw] "=&s"(hw), [xc] "=&s"(xc) \
        : [it] "s"(it_), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3) : CLOBBERS); \
    if ((threadIdx.x & 63) == 0) { Rec r; r.t0 = t0; r.t1 = t1; r.r0 = r0; r.r1 = r1; r.hwid = hw; r.xcc = xc; r.pad[0] = r.pad[1] = 0; \
        rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = r; } }
'''

HOST = r'''
struct Result { double cyc_simd_med, cyc_simd_min, cyc_wave_med, sclk_mhz, cyc_event; int w_ok, w_bad; };
static int g_wall_khz = 100000;
static Result run(kern_t k, int W, int iters, int per_iter, Rec* d_rec, std::vector<Rec>& h) {
    // exactly W workgroups (one wave per SIMD each) per CU: dynamic LDS sized so that W fit and W + 1 do not
    const int lds = W == 1 ? 96 * 1024 : W == 2 ? 64 * 1024 : W == 4 ? 40 * 1024 : 20 * 1024;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int nblk = 256 * W;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, 0, d_rec, 64); (void)hipDeviceSynchronize();
    Result best{}; best.cyc_simd_med = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, 0, d_rec, iters); (void)hipEventRecord(b);
        (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        if (hipGetLastError() != hipSuccess) { fprintf(stderr, "launch failed\n"); exit(1); }
        h.resize((size_t)nblk * 4); (void)hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
        const double n = (double)iters * per_iter;
        std::map<unsigned long long, std::vector<const Rec*>> simd;
        std::vector<double> per_wave, clk;
        for (const Rec& r : h) {
            // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
            unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 32) | (r.hwid & 0xff30u);
            simd[key].push_back(&r);
            per_wave.push_back((double)(r.t1 - r.t0) / n);
            if (r.r1 > r.r0) clk.push_back((double)(r.t1 - r.t0) / (double)(r.r1 - r.r0) * g_wall_khz * 1e-3);
        }
        std::vector<double> per_simd; int ok = 0, bad = 0;
        for (auto& kv : simd) {
            unsigned long long lo = ~0ull, hi = 0;
            for (const Rec* r : kv.second) { lo = std::min(lo, r->t0); hi = std::max(hi, r->t1); }
            per_simd.push_back((double)(hi - lo) / (n * kv.second.size()));
            if ((int)kv.second.size() == W) ok++; else bad++;
        }
        std::sort(per_simd.begin(), per_simd.end()); std::sort(per_wave.begin(), per_wave.end()); std::sort(clk.begin(), clk.end());
        Result r; r.cyc_simd_med = per_simd[per_simd.size() / 2]; r.cyc_simd_min = per_simd[0]; r.cyc_wave_med = per_wave[per_wave.size() / 2];
        r.sclk_mhz = clk.empty() ? 0 : clk[clk.size() / 2]; r.w_ok = ok; r.w_bad = bad;
        r.cyc_event = ms * 1e-3 * r.sclk_mhz * 1e6 / (n * W);
        if (r.cyc_simd_med < best.cyc_simd_med) best = r;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return best;
}
static const char* rate_class(double c) {
    if (c < 2.6) return "full (2)"; if (c < 4.6) return "half (4)"; if (c < 9.5) return "quarter (8)"; if (c < 18) return "1/8 (16)"; return "slower";
}
int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;
    const bool calib = only && !strcmp(only, "--calib");
    Rec* d_rec; (void)hipMalloc(&d_rec, sizeof(Rec) * 256 * 8 * 4);
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    (void)hipDeviceGetAttribute(&g_wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("# device %s, %d CUs, attribute clock %.3f GHz, wall-clock counter %.1f MHz\n", p.name, p.multiProcessorCount, khz * 1e-6, g_wall_khz * 1e-3);
    printf("# cycles per wave64 instruction on ONE SIMD = (last s_memtime - first s_memtime of the SIMD's waves) / (waves x instructions); median over the\n"
           "# 1024 SIMDs; W = resident waves per SIMD (checked from HW_ID: 'placement' = SIMDs holding exactly W / not); sclk = s_memtime ticks per\n"
           "# s_memrealtime tick, i.e. the shader clock DURING the run; 'event' = the same figure from the hipEvent wall time x that clock.\n"
           "# clean = the sources of an instruction sit in different VGPR banks (index mod 4); same = all in one bank; dep = one dependent chain (latency).\n");
    printf("%-26s %-22s | %7s %7s %7s %7s | %7s | %7s | %7s %9s %-10s %s\n", "instruction", "group", "W=1", "W=2", "W=4", "W=8", "same W8", "dep W1", "sclkMHz", "event W8", "placement", "rate");
    std::vector<Rec> h;
    const int iters = 1200;
    if (calib) {
        // counter calibration (run under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ...): kernels whose
        // VALU pipe is saturated by construction (8 waves per SIMD, nothing but one instruction class), two launches each
        const char* names[] = {"v_add_u32", "v_fma_f32", "v_perm_b32", "v_pk_mad_u16", "v_pk_fma_f32", "v_cvt_f32_ubyte0", "v_rcp_f32"};
        int idx = 0;
        for (const Entry& e : entries) {
            bool take = false;
            for (const char* n : names) take = take || !strcmp(n, e.name);
            if (take) {
                const int lds = 20 * 1024, it = 4800;
                (void)hipFuncSetAttribute((const void*)e.clean, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                for (int l = 0; l < 2; l++) hipLaunchKernelGGL(e.clean, dim3(256 * 8), dim3(256), lds, 0, d_rec, it);
                (void)hipDeviceSynchronize();
                printf("calib k%d_clean = %s: %d wave-instructions per wave, 8 waves per SIMD\n", idx, e.name, it * 32 * e.per_body);
            }
            idx++;
        }
        return 0;
    }
    for (const Entry& e : entries) {
        if (only && !strstr(e.name, only)) continue;
        const int per_iter = 32 * e.per_body;
        double c[4]; Result r8{};
        int ws[4] = {1, 2, 4, 8};
        for (int i = 0; i < 4; i++) { Result r = run(e.clean, ws[i], iters, per_iter, d_rec, h); c[i] = r.cyc_simd_med; if (i == 3) r8 = r; }
        Result rs = run(e.same, 8, iters, per_iter, d_rec, h);
        Result rd = run(e.dep, 1, iters, per_iter, d_rec, h);
        char place[32]; snprintf(place, sizeof place, "%d/%d", r8.w_ok, r8.w_bad);
        printf("%-26s %-22s | %7.2f %7.2f %7.2f %7.2f | %7.2f | %7.2f | %7.0f %9.2f %-10s %s\n", e.name, e.group, c[0], c[1], c[2], c[3], rs.cyc_simd_med, rd.cyc_simd_med,
               r8.sclk_mhz, r8.cyc_event, place, rate_class(c[3]));
        fflush(stdout);
    }
    // data dependence: the same kernels with lane-dependent random mantissas in the operands (the table above uses the constants
    // 1.0 and 0: the least switching activity there is).  If the rate or the clock moves, the ceiling is a power figure
    if (!only) {
        printf("# random operands (x in [0.5, 1), y tiny, per lane): cycles per wave-instruction at W = 8, clock during the run; 200 launches back to back, last one read\n");
        for (const Entry& e : entries) {
            if (strcmp(e.name, "v_add_u32") && strcmp(e.name, "v_fma_f32") && strcmp(e.name, "v_fmac_f32") && strcmp(e.name, "v_mul_f32") && strcmp(e.name, "v_pk_fma_f32") &&
                strcmp(e.name, "v_pk_mad_u16") && strcmp(e.name, "v_perm_b32") && strcmp(e.name, "v_cvt_f32_ubyte1") && strcmp(e.name, "v_dot2c_i32_i16")) continue;
            for (int rnd = 0; rnd < 2; rnd++) {
                const int W = 8, lds = 20 * 1024, nblk = 256 * W, it = 2400; const double n = (double)it * 32 * e.per_body;
                (void)hipFuncSetAttribute((const void*)e.clean, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                for (int l = 0; l < 200; l++) hipLaunchKernelGGL(e.clean, dim3(nblk), dim3(256), lds, 0, d_rec, rnd ? -it : it);
                (void)hipDeviceSynchronize(); h.resize((size_t)nblk * 4); (void)hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
                std::map<unsigned long long, std::vector<const Rec*>> simd; std::vector<double> clk, per;
                for (const Rec& r : h) { simd[((unsigned long long)(r.xcc & 0xf) << 32) | (r.hwid & 0xff30u)].push_back(&r); clk.push_back((double)(r.t1 - r.t0) / (double)(r.r1 - r.r0) * g_wall_khz * 1e-3); }
                for (auto& kv : simd) { unsigned long long lo = ~0ull, hi = 0; for (const Rec* r : kv.second) { lo = std::min(lo, r->t0); hi = std::max(hi, r->t1); } per.push_back((double)(hi - lo) / (n * kv.second.size())); }
                std::sort(clk.begin(), clk.end()); std::sort(per.begin(), per.end());
                printf("%-22s %-8s %6.2f cycles  %5.0f MHz  -> %6.1f G wave-instr/s per chip\n", e.name, rnd ? "random" : "constant", per[per.size() / 2], clk[clk.size() / 2],
                       1024.0 * clk[clk.size() / 2] * 1e-3 / per[per.size() / 2]);
            }
        }
    }
    // sustained leg: what the clock does under ~1 s of dense VALU work (the bench's timed region is that long)
    if (!only) {
        printf("# sustained: 400 back-to-back launches at W = 8 (about 1 s); shader clock during launches 0, 100, 200, 399\n");
        for (const Entry& e : entries) {
            if (strcmp(e.name, "v_add_u32") && strcmp(e.name, "v_fma_f32") && strcmp(e.name, "v_pk_fma_f32") && strcmp(e.name, "v_pk_mad_u16") && strcmp(e.name, "v_perm_b32")) continue;
            const int W = 8, lds = 20 * 1024, nblk = 256 * W, it = 2400;
            (void)hipFuncSetAttribute((const void*)e.clean, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            printf("%-26s", e.name);
            for (int l = 0; l < 400; l++) {
                hipLaunchKernelGGL(e.clean, dim3(nblk), dim3(256), lds, 0, d_rec, it);
                if (l == 0 || l == 100 || l == 200 || l == 399) {
                    (void)hipDeviceSynchronize(); h.resize((size_t)nblk * 4); (void)hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
                    std::vector<double> clk;
                    for (const Rec& r : h) clk.push_back((double)(r.t1 - r.t0) / (double)(r.r1 - r.r0) * g_wall_khz * 1e-3);
                    std::sort(clk.begin(), clk.end());
                    printf("  [%3d] %5.0f MHz", l, clk[clk.size() / 2]);
                }
            }
            printf("\n"); fflush(stdout);
        }
    }
    return 0;
}
'''


def main():
    out = [HEADER]
    ents = []
    for idx, (name, tmpl, group) in enumerate(CLASSES):
        unroll = BIG.get(name, UNROLL)
        per_body = (tmpl.count("\\n") + 1) * unroll // UNROLL
        for mode in MODES:
            out.append(f'KERNEL({kname(idx, mode)}, "{body(tmpl, mode, unroll)}")')
        ents.append(f'    {{"{name}", "{group}", {per_body}, {kname(idx, "clean")}, {kname(idx, "same")}, {kname(idx, "dep")}}},')
    out.append("static const Entry entries[] = {")
    out.extend(ents)
    out.append("};")
    out.append(HOST)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_rate.cpp")
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
