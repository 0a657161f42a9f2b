// Issue-rate microbenchmark for the VALU instructions the detector / descriptor kernels lean on (gfx950).
// Each kernel runs ITER iterations of 8 independent copies of one instruction per wave; with 8 waves per SIMD
// resident the measured time gives cycles per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.cpp -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define BODY8(ASM) \
    asm volatile(ASM "\n" : "+v"(a0) : "v"(x), "v"(y)); asm volatile(ASM "\n" : "+v"(a1) : "v"(x), "v"(y)); \
    asm volatile(ASM "\n" : "+v"(a2) : "v"(x), "v"(y)); asm volatile(ASM "\n" : "+v"(a3) : "v"(x), "v"(y)); \
    asm volatile(ASM "\n" : "+v"(a4) : "v"(x), "v"(y)); asm volatile(ASM "\n" : "+v"(a5) : "v"(x), "v"(y)); \
    asm volatile(ASM "\n" : "+v"(a6) : "v"(x), "v"(y)); asm volatile(ASM "\n" : "+v"(a7) : "v"(x), "v"(y));
#define KERNEL(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(unsigned* out) { \
    unsigned x = threadIdx.x * 2654435761u, y = blockIdx.x + 12345u; \
    unsigned a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7; \
    for (int i = 0; i < ITER; i++) { BODY8(ASM) } \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = 1; }
KERNEL(k_add, "v_add_u32 %0, %1, %0")
KERNEL(k_add3, "v_add3_u32 %0, %1, %2, %0")
KERNEL(k_mul24, "v_mul_i32_i24 %0, %1, %0")
KERNEL(k_mad24, "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %1, %0")
KERNEL(k_perm, "v_perm_b32 %0, %1, %0, %2")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %1, %0, 1")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %1, %0")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")
KERNEL(k_pk_lshl, "v_pk_lshlrev_b16 %0, 1, %0")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0")
KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %1, %2")
KERNEL(k_dot4c, "v_dot4c_i32_i8 %0, %1, %2")
KERNEL(k_min_sdwa, "v_min_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
KERNEL(k_min, "v_min_u32 %0, %1, %0")
KERNEL(k_med3, "v_med3_i32 %0, %1, %2, %0")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %1, 3, %0")
KERNEL(k_sad, "v_sad_u8 %0, %1, %2, %0")
KERNEL(k_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
// round 3: what exactly is slow about a select -- the mask read, the encoding, or the compare that feeds it?
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %1, %0, s[20:21]")
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(k_addc_s, "v_addc_co_u32_e64 %0, s[22:23], %0, %1, s[20:21]")
KERNEL(k_cmp_vcc, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL(k_cmp_s, "v_cmp_lt_u32_e64 s[20:21], %0, %1")
KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc")
KERNEL(k_cmp_addc, "v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(k_readlane, "v_readlane_b32 s24, %0, 3")
KERNEL(k_writelane, "v_writelane_b32 %0, s24, 3")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
// packed fp32 needs register pairs
__global__ __launch_bounds__(256) void k_pk_fma_f32(unsigned* out) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x = {(float)threadIdx.x, 1.5f}, y = {0.999f, 1.001f};
    f2 a0 = x, a1 = x + 1.f, a2 = x + 2.f, a3 = x + 3.f, a4 = x + 4.f, a5 = x + 5.f, a6 = x + 6.f, a7 = x + 7.f;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a4) : "v"(x), "v"(y)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a5) : "v"(x), "v"(y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(x), "v"(y)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a7) : "v"(x), "v"(y));
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 0.12345f) out[0] = 1;
}
template <class K> void run(const char* name, K k, unsigned* d, double clk_ghz) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int nblk = 256 * 8;    // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, 0, d); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) { hipEventRecord(a); hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, 0, d); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    // per SIMD: 8 waves x ITER x 8 instructions
    const double instr = 8.0 * ITER * 8;
    printf("%-14s %8.1f us  -> %5.2f cycles per wave-instruction (at %.2f GHz)\n", name, best * 1e3, best * 1e-3 * clk_ghz * 1e9 / instr, clk_ghz);
}
int main() {
    unsigned* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6; printf("clock %.3f GHz\n", ghz);
#define R(k) run(#k, k, d, ghz)
    R(k_add); R(k_add3); R(k_mul24); R(k_mad24); R(k_mullo); R(k_perm); R(k_alignbyte); R(k_pk_add_u16); R(k_pk_sub_i16); R(k_pk_lshl);
    R(k_pk_mad_u16); R(k_dot2c); R(k_dot4c); R(k_min_sdwa); R(k_min); R(k_med3); R(k_fma); R(k_cvt_ub); R(k_cvt_pk_u8); R(k_bfe); R(k_lshl_or);
    R(k_sad); R(k_dpp); R(k_cndmask); R(k_cndmask_s); R(k_addc); R(k_addc_s); R(k_cmp_vcc); R(k_cmp_s); R(k_cmp_cnd); R(k_cmp_addc); R(k_readlane); R(k_writelane); R(k_alignbit); R(k_pk_fma_f32);
    return 0;
}
