// Round 6 probe: n single-wave workgroups add 4 counts each (lanes 0..3) into a 16 384-bin histogram whose hit bins are `span`
// CONSECUTIVE bins (responses of a natural frame crowd into ~100 neighbouring bins), under three bin -> word layouts:
//   0  word = (b & 31) << 9 | b >> 5          neighbouring bins 2 KB apart (32 planes)
//   1  word = (b & 31) * 544 + (b >> 5)       ... 2 KB + one line apart
//   2  word = (b & 511) << 5 | b >> 9         neighbouring bins in neighbouring 128-byte lines
// Build: hipcc -O3 --offload-arch=gfx950 hist_layout.cpp -o hist_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <functional>
__device__ __forceinline__ unsigned word_of(unsigned b, int layout)
{
    return layout == 0 ? ((b & 31u) << 9) | (b >> 5) : layout == 1 ? (b & 31u) * 544u + (b >> 5) : ((b & 511u) << 5) | (b >> 9);
}
__global__ void k_hist(int* h, int layout, unsigned base, unsigned span)
{
    if (threadIdx.x < 4) {
        unsigned x = blockIdx.x * 4u + threadIdx.x;
        x = x * 2654435761u; x ^= x >> 15;                       // which of the hit bins
        __hip_atomic_fetch_add(h + word_of(base + x % span, layout), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_empty(int*) {}
static float timeit(const std::function<void()>& f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 20; i++) f();
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 20;
}
int main()
{
    int* d; (void)hipMalloc(&d, 1 << 20); (void)hipMemset(d, 0, 1 << 20);
    const int n = 25500;
    printf("n=%d workgroups x 4 updates; empty launch %.1f us\n", n, timeit([&]{ hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), 0, 0, d); }));
    for (unsigned span : { 8u, 32u, 100u, 400u, 2000u })
        printf("  %4u hit bins: planes %.1f us | padded planes %.1f us | line-major %.1f us\n", span,
               timeit([&]{ hipLaunchKernelGGL(k_hist, dim3(n), dim3(64), 0, 0, d, 0, 9000u, span); }),
               timeit([&]{ hipLaunchKernelGGL(k_hist, dim3(n), dim3(64), 0, 0, d, 1, 9000u, span); }),
               timeit([&]{ hipLaunchKernelGGL(k_hist, dim3(n), dim3(64), 0, 0, d, 2, 9000u, span); }));
    return 0;
}
