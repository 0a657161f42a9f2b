#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int* p) { if (threadIdx.x == 1000) p[0] = 1; }
template<int LDSB>
__global__ void k_lds(int* p) { __shared__ int s[LDSB/4]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (s[(threadIdx.x+1)&255] == 12345) p[0] = 1; }
__global__ void k_atomic(int* p) { __shared__ int st; if (threadIdx.x == 0) st = atomicAdd(p, 1); __syncthreads(); if (st == -5) p[1] = 1; }
__global__ void k_atomic8(int* p) { __shared__ int st; if (threadIdx.x == 0) st = atomicAdd(p + 32 * (blockIdx.x & 7), 1); __syncthreads(); if (st == -5) p[1] = 1; }
__global__ void k_barriers(int* p) { __shared__ int s[256]; int v = threadIdx.x; for (int i = 0; i < 20; i++) { s[threadIdx.x] = v; __syncthreads(); v += s[(threadIdx.x + 1) & 255]; __syncthreads(); } if (v == 12345) p[0] = 1; }
template <class F> float timeit(F f, int reps = 5) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize(); float best = 1e9; for (int r = 0; r < reps; r++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; } return best * 1000; }
int main() {
  int* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  for (int n : {8160, 25500, 40000}) {
    printf("n=%d empty %.1f us | lds1k %.1f | lds16k %.1f | lds26k %.1f | atomic %.1f | atomic8 %.1f | barriers40 %.1f\n", n,
      timeit([&]{ hipLaunchKernelGGL(k_empty, dim3(n), dim3(256), 0, 0, d); }),
      timeit([&]{ hipLaunchKernelGGL(k_lds<1024>, dim3(n), dim3(256), 0, 0, d); }),
      timeit([&]{ hipLaunchKernelGGL(k_lds<16384>, dim3(n), dim3(256), 0, 0, d); }),
      timeit([&]{ hipLaunchKernelGGL(k_lds<26624>, dim3(n), dim3(256), 0, 0, d); }),
      timeit([&]{ hipLaunchKernelGGL(k_atomic, dim3(n), dim3(256), 0, 0, d); }),
      timeit([&]{ hipLaunchKernelGGL(k_atomic8, dim3(n), dim3(256), 0, 0, d); }),
      timeit([&]{ hipLaunchKernelGGL(k_barriers, dim3(n), dim3(256), 0, 0, d); }));
  }
  return 0;
}
