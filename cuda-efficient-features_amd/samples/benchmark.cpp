// benchmark.cpp -- the reference's timing harness (samples/sample_benchmark.cpp:39-52, 104-142) on the C++ facade:
// 1 warm-up + N timed iterations of the *Async call followed by a stream synchronise, input resident on the device.
// The reference reads a JPEG; this image has no OpenCV, so a seeded synthetic frame (filled rectangles + noise)
// of the requested size is generated instead.
//   efx_benchmark [width height] [--max-keypoints N] [--descriptor-type 0|1] [--descriptor-bits 256|512]
//                 [--benchmark-type 0|1|2] [--num-iterations N]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../host/efficient_features.hpp"

// Protocol of the reference's harness: the first call warms up and is not counted, the mean of the next `n` is reported.
struct Timing { double mean_ms, min_ms, max_ms; };

template <class Call>
static Timing time_calls(int n, Call call)
{
    using clock = std::chrono::steady_clock;
    call();                                                         // warm-up: allocations, first-launch costs
    std::vector<double> ms((size_t)n);
    for (double& m : ms) {
        const clock::time_point begin = clock::now();
        call();
        m = std::chrono::duration<double, std::milli>(clock::now() - begin).count();
    }
    Timing t{ 0, ms.empty() ? 0 : ms[0], 0 };
    for (double m : ms) { t.mean_ms += m / n; t.min_ms = m < t.min_ms ? m : t.min_ms; t.max_ms = m > t.max_ms ? m : t.max_ms; }
    return t;
}

static std::vector<uint8_t> synth(int w, int h, uint32_t seed)
{
    std::vector<uint8_t> img((size_t)w * h, 128);
    auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    const int nshapes = (int)(700.0 * w * h / 1e6);   // every level above its quota, FAST corners below the 10 % cap
    for (int i = 0; i < nshapes; i++) {
        const int sc[5] = { 10, 18, 32, 56, 96 };
        const int s = sc[rnd() % 5];
        const int rw = s / 2 + (int)(rnd() % (unsigned)s), rh = s / 2 + (int)(rnd() % (unsigned)s);
        const int x0 = (int)(rnd() % (unsigned)w), y0 = (int)(rnd() % (unsigned)h);
        const uint8_t v = (uint8_t)(rnd() & 255);
        for (int y = y0; y < y0 + rh && y < h; y++) memset(&img[(size_t)y * w + x0], v, (size_t)((x0 + rw < w ? rw : w - x0)));
    }
    for (auto& p : img) { const int v = (int)p + (int)(rnd() % 7) - 3; p = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    return img;
}

int main(int argc, char** argv)
{
    int w = 3840, h = 2160, nfeatures = 10000, descType = 0, descBits = 256, benchType = 0, niter = 100;
    int pos = 0;
    for (int i = 1; i < argc; i++) {
        auto opt = [&](const char* name, int& dst) { if (!strcmp(argv[i], name) && i + 1 < argc) { dst = atoi(argv[++i]); return true; } return false; };
        if (opt("--max-keypoints", nfeatures) || opt("--descriptor-type", descType) || opt("--descriptor-bits", descBits) ||
            opt("--benchmark-type", benchType) || opt("--num-iterations", niter)) continue;
        if (pos == 0) { w = atoi(argv[i]); pos++; } else if (pos == 1) { h = atoi(argv[i]); pos++; }
    }
    try {
        const auto dt = descType == 0 ? (descBits == 256 ? efx::EfficientFeatures::BAD_256 : efx::EfficientFeatures::BAD_512)
                                      : (descBits == 256 ? efx::EfficientFeatures::HASH_SIFT_256 : efx::EfficientFeatures::HASH_SIFT_512);
        auto feature = efx::EfficientFeatures::create(nfeatures);
        feature->setDescriptorType(dt);
        const char* benchStr[] = { "detect-and-compute", "detect-only", "compute-only" };
        printf("=== configurations ===\nimage size      : [%d x %d]\ndescriptor type : %s\ndescriptor bits : %d\nmax keypoints   : %d\nbenchmark type  : %s\n\n",
               w, h, descType == 0 ? "BAD" : "HashSIFT", descBits, nfeatures, benchStr[benchType]);
        const std::vector<uint8_t> gray = synth(w, h, 12345);
        efx::DeviceMatrix d_gray, d_keypoints, d_descriptors;
        d_gray.create(h, w, 1);
        if (hipMemcpy2D(d_gray.data(), d_gray.step, gray.data(), (size_t)w, (size_t)w, (size_t)h, hipMemcpyHostToDevice) != hipSuccess) return 1;
        const efx::DeviceImage img{ static_cast<const uint8_t*>(d_gray.data()), h, w, d_gray.step };
        hipStream_t stream;
        if (hipStreamCreate(&stream) != hipSuccess) return 1;
        const auto wait = [&] { (void)hipStreamSynchronize(stream); };
        int n_computed = 0;
        if (benchType == 2) {                                       // compute-only works on one detection's keypoints
            feature->detectAsync(img, d_keypoints, stream);
            wait();
            n_computed = feature->lastCount();
        }
        const Timing t = time_calls(niter, [&] {
            switch (benchType) {
            case 0: feature->detectAndComputeAsync(img, d_keypoints, d_descriptors, false, stream); break;
            case 1: feature->detectAsync(img, d_keypoints, stream); break;
            default: feature->computeAsync(img, d_keypoints, n_computed, d_descriptors, stream); break;
            }
            wait();
        });
        printf("%5d keypoints found.\nprocessing time: %.3f[milli sec]   (min %.3f, max %.3f over %d calls)\n",
               feature->lastCount(), t.mean_ms, t.min_ms, t.max_ms, niter);
        efx_level_stats st[32]; int nl = 0;
        if (efx_last_level_stats(feature->handle(), st, 32, &nl) == EFX_OK)
            for (int l = 0; l < nl; l++) printf("level %d: %d FAST corners, %d after NMS, %d kept\n", l, st[l].n_candidates, st[l].n_after_nms, st[l].n_kept);
    } catch (const efx::Exception& e) {
        fprintf(stderr, "efx error %d: %s\n", e.code, e.what());
        return 2;
    }
    return 0;
}
