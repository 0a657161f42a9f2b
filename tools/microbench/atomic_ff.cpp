// Round 6 probe: cost of device-scope atomics that do NOT return a value ("fire and forget": histogram / row-sum updates) against
// returning ones, by how many distinct 128-byte lines the updates of a launch hit.  n single-wave workgroups, one atomic each
// (lane 0), line = workgroup % L.  Build: hipcc -O3 --offload-arch=gfx950 atomic_ff.cpp -o /tmp/atomic_ff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <functional>
__global__ void k_empty(int*) {}
__global__ void k_ret(int* p, int L) { __shared__ int st; if (threadIdx.x == 0) st = atomicAdd(p + 32 * (blockIdx.x % L), 1); __syncthreads(); if (st == -5) p[1] = 1; }
__global__ void k_ff(int* p, int L) { if (threadIdx.x == 0) __hip_atomic_fetch_add(p + 32 * (blockIdx.x % L), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every lane one update: 64 n updates over L lines, word inside the line by lane
__global__ void k_ff_lanes(int* p, int L) { __hip_atomic_fetch_add(p + 32 * ((blockIdx.x * 7 + threadIdx.x) % L) + (threadIdx.x & 31), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static float timeit(const std::function<void()>& f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; i++) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 20;
}
int main()
{
    int* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    for (int n : { 8160, 25500, 65536 }) {
        printf("n=%d empty %.1f us\n", n, timeit([&]{ hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), 0, 0, d); }));
        for (int L : { 1, 8, 64, 512, 4096 })
            printf("  lines %4d: returning %.1f us | fire-and-forget %.1f us | 64 lanes each, fire-and-forget %.1f us\n", L,
                   timeit([&]{ hipLaunchKernelGGL(k_ret, dim3(n), dim3(64), 0, 0, d, L); }),
                   timeit([&]{ hipLaunchKernelGGL(k_ff, dim3(n), dim3(64), 0, 0, d, L); }),
                   timeit([&]{ hipLaunchKernelGGL(k_ff_lanes, dim3(n), dim3(64), 0, 0, d, L); }));
    }
    return 0;
}
