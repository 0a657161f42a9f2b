// first_frame -- ONE process, ONE context, ONE frame.
// The case the process-wide block cache cannot cover: the first frame of the first context of a fresh process runs on
// freshly mapped device memory.  tools/microbench/first_frame_stress.py starts many of these at once and counts the
// results that differ from the oracle's digest.
// usage: first_frame <gray.raw> <rows> <cols> <nfeatures> <expected fnv64 hex | 0> [extra frames]   exit 0: equal (or printed), 3: differs
//        extra frames > 0: after the first (checked) frame the same context keeps running frames -- background load for the stress
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/first_frame.cpp -Iinclude -Lcuda-efficient-features_amd -lefx_hip
//        -Wl,-rpath,'$ORIGIN/../../cuda-efficient-features_amd' -o tools/microbench/first_frame
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "efx.h"

static unsigned long long fnv(const void* p, size_t n, unsigned long long h)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: first_frame <gray.raw> <rows> <cols> <nfeatures> <expected hex | 0>\n"); return 2; }
    const int rows = atoi(argv[2]), cols = atoi(argv[3]), nf = atoi(argv[4]);
    const unsigned long long want = strtoull(argv[5], nullptr, 16);
    std::vector<unsigned char> img((size_t)rows * cols);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(img.data(), 1, img.size(), f) != img.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    unsigned char* d_img = nullptr; void* d_kps = nullptr; unsigned char* d_desc = nullptr; int* d_cnt = nullptr;
    const size_t kpitch = (size_t)nf * 4;
    if (hipMalloc(&d_img, img.size()) != hipSuccess || hipMalloc(&d_kps, kpitch * 5) != hipSuccess ||
        hipMalloc(&d_desc, (size_t)nf * 32) != hipSuccess || hipMalloc(&d_cnt, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
    hipMemset(d_kps, 0, kpitch * 5); hipMemset(d_desc, 0, (size_t)nf * 32);
    efx_params p; efx_default_params(&p);
    p.nfeatures = nf; p.descriptor_type = EFX_BAD_256;
    efx_context* ctx = nullptr;
    if (efx_create(&p, &ctx) != EFX_OK) { fprintf(stderr, "efx_create: %s\n", efx_last_error(nullptr)); return 2; }
    int rc = efx_detect_and_compute_async(ctx, d_img, rows, cols, (size_t)cols, d_kps, kpitch, d_desc, 32, nf, d_cnt, nullptr);
    if (rc != EFX_OK) { fprintf(stderr, "detectAndCompute: %s\n", efx_last_error(ctx)); return 2; }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "device error: %s\n", hipGetErrorString(hipGetLastError())); return 4; }
    int n = 0;
    hipMemcpy(&n, d_cnt, 4, hipMemcpyDeviceToHost);
    std::vector<unsigned char> kps(kpitch * 5), desc((size_t)nf * 32);
    hipMemcpy(kps.data(), d_kps, kps.size(), hipMemcpyDeviceToHost);
    hipMemcpy(desc.data(), d_desc, desc.size(), hipMemcpyDeviceToHost);
    unsigned long long h = fnv(&n, 4, 1469598103934665603ull);
    for (int r = 0; r < 5; r++) h = fnv(kps.data() + r * kpitch, (size_t)n * 4, h);
    h = fnv(desc.data(), (size_t)n * 32, h);
    if (want == 0) { printf("%016llx %d\n", h, n); return 0; }
    if (h != want) { printf("DIFFERS n %d digest %016llx\n", n, h); return 3; }
    const long extra = argc > 6 ? atol(argv[6]) : 0;
    for (long i = 0; i < extra; i++) {
        if (efx_detect_and_compute_async(ctx, d_img, rows, cols, (size_t)cols, d_kps, kpitch, d_desc, 32, nf, d_cnt, nullptr) != EFX_OK) return 2;
        if ((i & 7) == 7 && hipDeviceSynchronize() != hipSuccess) return 4;
    }
    return 0;
}
