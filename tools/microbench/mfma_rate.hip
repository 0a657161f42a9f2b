// MFMA issue-rate probe (GPU box): hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// v_mfma_f32_32x32x64_f8f6f4 with FP4 operands: cycles per instruction per SIMD for (a) four independent accumulators,
// (b) one dependent chain, (c) one chain with an s_waitcnt between the links, at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc)
{
    i32x8 a = { (int)threadIdx.x, 0x22222222, 0x2a2a2a2a, 0x22a2a222, 0, 0, 0, 0 }, b = { 0x22222222, (int)threadIdx.x * 3, 0x2a2aa2a2, 0x2222a2a2, 0, 0, 0, 0 };
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
                if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float* out; long long* cyc; hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
        for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
            const int grid = 256 * wgs_per_cu;      // 256-thread workgroups: one wave per SIMD each
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double mf = 4.0 * iters * wgs_per_cu;      // MFMAs per SIMD
            printf("mode %d (%s) waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD, wave clock64 ticks per own MFMA %.1f, %.0f TFLOP/s\n", mode,
                   mode == 0 ? "4 accumulators" : mode == 1 ? "one chain" : "one chain + s_waitcnt", wgs_per_cu, ms, ms * 1e6 / mf, (double)c / (4.0 * iters),
                   mf * 1024 * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
