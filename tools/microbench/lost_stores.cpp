// Stand-alone reproducer for the "16 processes share one GPU" discrepancy: no library code.
// Each iteration: poison a buffer, let a small grid store one tagged word per thread (8-byte stride, like Corner::xy of
// fast_kernel), wait for the stream, copy the buffer to the host and check every word.  Run N copies at once:
//   for i in $(seq 16); do ./lost_stores 20000 & done; wait
// Third argument 1: free and re-allocate the buffers every iteration (what a test loop that creates a new context per frame
// does); 0: allocate once.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/lost_stores.cpp -o tools/microbench/lost_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ __launch_bounds__(256) void tag_kernel(unsigned* out, unsigned tag, int* counter)
{
    __shared__ int s_start;
    // a chunk per workgroup from a global counter (arbitrary placement, as the corner arenas are filled)
    if (threadIdx.x == 0) s_start = atomicAdd(counter, 256);
    __syncthreads();
    out[2 * (s_start + threadIdx.x)] = (tag << 16) | (blockIdx.x & 0xffff);
}
__global__ void busy_kernel(unsigned* p, int n) { for (int i = 0; i < n; i++) p[blockIdx.x * 64 + (threadIdx.x & 63)] += i; }

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, nwg = argc > 2 ? atoi(argv[2]) : 100, churn = argc > 3 ? atoi(argv[3]) : 0;
    unsigned *d = nullptr, *scratch = nullptr, *pad[6] = {}; int* cnt = nullptr;
    CK(hipMalloc(&d, (size_t)nwg * 256 * 8)); CK(hipMalloc(&scratch, 1 << 20)); CK(hipMalloc(&cnt, 4));
    std::vector<unsigned> h((size_t)nwg * 256 * 2);
    long bad_iters = 0, bad_words = 0;
    for (int it = 0; it < iters; it++) {
        if (churn) {
            // the allocation pattern of one library context: a dozen buffers from kilobytes to a few megabytes
            CK(hipFree(d)); CK(hipFree(scratch)); CK(hipFree(cnt));
            for (int k = 0; k < 6; k++) if (pad[k]) CK(hipFree(pad[k]));
            for (int k = 0; k < 6; k++) CK(hipMalloc(&pad[k], (size_t)(4096 << (k * 2 % 11)) + 256 * (it % 7)));
            CK(hipMalloc(&d, (size_t)nwg * 256 * 8)); CK(hipMalloc(&scratch, 1 << 20)); CK(hipMalloc(&cnt, 4));
            // ... and its two small synchronous uploads from pageable host memory (level table, describer parameters)
            static std::vector<unsigned> table(3000, 7u);
            CK(hipMemcpy(pad[0], table.data(), 4096, hipMemcpyHostToDevice)); CK(hipMemcpy(pad[3], table.data(), 12000, hipMemcpyHostToDevice));
        }
        CK(hipMemsetAsync(d, 0xFF, h.size() * 4, 0)); CK(hipMemsetAsync(cnt, 0, 4, 0));
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(256), 0, 0, scratch, 50);
        hipLaunchKernelGGL(tag_kernel, dim3(nwg), dim3(256), 0, 0, d, (unsigned)(it & 0x7fff), cnt);
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(256), 0, 0, scratch, 50);
        CK(hipStreamSynchronize(0));
        CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
        long w = 0;
        for (size_t i = 0; i < h.size(); i += 2) if ((h[i] >> 16) != (unsigned)(it & 0x7fff)) w++;
        if (w) { bad_iters++; bad_words += w; if (bad_iters <= 3) printf("iteration %d: %ld of %zu words not stored (first word %08x)\n", it, w, h.size() / 2, h[0]); }
    }
    printf("iterations %d bad_iterations %ld missing_words %ld\n", iters, bad_iters, bad_words);
    return 0;
}
