// facade_check.cpp -- exercises the C++ facade (host/efficient_features.hpp) end to end on the GPU: the calls a user of the
// reference's classes makes (samples/sample_feature_extraction.cpp:83-105, sample_feature_matching.cpp:85-101), plus the
// mask / useProvidedKeypoints / uploader additions.  Prints "facade ok" and returns 0 when every property holds.
#include "../host/efficient_features.hpp"

#include <cstdio>
#include <cstring>
#include <vector>

static std::vector<uint8_t> synth(int w, int h, uint32_t seed)
{
    std::vector<uint8_t> img((size_t)w * h, 128);
    auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int i = 0; i < (int)(700.0 * w * h / 1e6); i++) {
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

#define REQUIRE(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main()
{
    try {
        const int w = 800, h = 600;
        const std::vector<uint8_t> gray = synth(w, h, 4242);
        auto feature = efx::EfficientFeatures::create(3000);
        feature->setDescriptorType(efx::EfficientFeatures::BAD_256);
        const efx::HostImage img{ gray.data(), h, w, (size_t)w };

        // detectAndCompute, then the same keypoints through useProvidedKeypoints: identical descriptors (spec S13)
        std::vector<efx::KeyPoint> kps; std::vector<uint8_t> desc;
        feature->detectAndCompute(img, kps, desc);
        REQUIRE(kps.size() > 300 && desc.size() == kps.size() * 32);
        std::vector<efx::KeyPoint> kps2 = kps; std::vector<uint8_t> desc2;
        feature->detectAndCompute(img, kps2, desc2, true);
        REQUIRE(desc2 == desc);

        // mask: only the left half; every keypoint must lie there, and an all-set mask changes nothing (spec S12)
        std::vector<uint8_t> mask((size_t)w * h, 0);
        for (int y = 0; y < h; y++) memset(&mask[(size_t)y * w], 255, (size_t)w / 2);
        std::vector<efx::KeyPoint> km; std::vector<uint8_t> dm;
        feature->detectAndCompute(img, efx::HostImage{ mask.data(), h, w, (size_t)w }, km, dm);
        REQUIRE(!km.empty());
        for (const auto& k : km) REQUIRE(k.x < w / 2);
        std::fill(mask.begin(), mask.end(), 1);
        std::vector<efx::KeyPoint> ka; std::vector<uint8_t> da;
        feature->detectAndCompute(img, efx::HostImage{ mask.data(), h, w, (size_t)w }, ka, da);
        REQUIRE(da == desc);

        // uploader (BGR frame whose channels all equal the gray frame -> the same gray) + async path + matcher
        std::vector<uint8_t> bgr((size_t)w * h * 3);
        for (size_t i = 0; i < gray.size(); i++) bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = gray[i];
        efx::Uploader up;
        efx::DeviceMatrix dk, dd;
        const efx::DeviceImage dimg = up.upload(bgr.data(), h, w, (size_t)w * 3, 3);
        feature->detectAndComputeAsync(dimg, dk, dd);
        REQUIRE(hipStreamSynchronize(nullptr) == hipSuccess);
        const int n = feature->lastCount();
        REQUIRE(n == (int)kps.size());                     // (3735 + 19235 + 9798) v + 16384 >> 15 == v
        efx::BFMatcher matcher(true);
        std::vector<efx::DMatch> matches;
        matcher.match(dd, n, dd, n, 32, matches);
        REQUIRE((int)matches.size() > n * 9 / 10);         // self match: (almost) every descriptor is its own mutual nearest
        int self = 0;
        for (const auto& m : matches) { REQUIRE(m.distance == 0); self += m.queryIdx == m.trainIdx; }
        REQUIRE(self > n * 9 / 10);
        efx::BFMatcher knn;
        std::vector<std::vector<efx::DMatch>> kn;
        knn.knnMatch(dd, n, dd, n, 32, kn);
        REQUIRE((int)kn.size() == n && kn[0].size() == 2 && kn[0][0].distance == 0 && kn[0][0].distance <= kn[0][1].distance);
        // the frame three times as ONE batched call (one launch of every kernel for the three frames): every frame's N and
        // descriptor bytes equal the single call's
        {
            std::vector<efx::DeviceImage> frames(3, dimg);
            std::vector<efx::DeviceMatrix> bk, bd;
            int* d_counts = nullptr;
            REQUIRE(hipMalloc(&d_counts, 3 * sizeof(int)) == hipSuccess);
            feature->detectAndComputeBatchAsync(frames, bk, bd, { d_counts, d_counts + 1, d_counts + 2 });
            REQUIRE(hipStreamSynchronize(nullptr) == hipSuccess);
            int hc[3] = { 0, 0, 0 };
            REQUIRE(hipMemcpy(hc, d_counts, sizeof(hc), hipMemcpyDeviceToHost) == hipSuccess);
            std::vector<uint8_t> one((size_t)n * 32), other((size_t)n * 32);
            REQUIRE(hipMemcpy2D(one.data(), 32, dd.data(), dd.step, 32, n, hipMemcpyDeviceToHost) == hipSuccess);
            for (int f = 0; f < 3; f++) {
                REQUIRE(hc[f] == n);
                REQUIRE(hipMemcpy2D(other.data(), 32, bd[f].data(), bd[f].step, 32, n, hipMemcpyDeviceToHost) == hipSuccess);
                REQUIRE(other == one);
            }
            (void)hipFree(d_counts);
        }
        printf("facade ok: %d keypoints, %d masked, %d cross-checked self matches, batch of 3 equal\n", n, (int)km.size(), (int)matches.size());
        return 0;
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
}
