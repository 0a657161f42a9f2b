// efficient_features.hpp -- C++ host facade over the C ABI (include/efx.h), mirroring the reference classes
//   cv::cuda::EfficientFeatures   modules/cuda_efficient_features/include/cuda_efficient_features.h:28-98
//   cv::cuda::BAD / HashSIFT      modules/cuda_efficient_features/include/cuda_efficient_descriptors.h:27-126
// with the same method names, argument meaning and error behaviour (bad arguments throw, like CV_Assert /
// CV_Error).  OpenCV is not available in this image, so the facade carries its own small types:
//   efx::KeyPoint    == cv::KeyPoint fields the reference fills (convert, cuda_efficient_features.cpp:323-349)
//   efx::HostImage   == a CV_8UC1 cv::Mat view            efx::DeviceImage == a CV_8UC1 cv::cuda::GpuMat view
//   efx::DeviceMatrix== an owning cv::cuda::GpuMat stand-in (hipMalloc'd, rows x cols x elemSize, pitched)
// INTEGRATION.md shows the 30-line adapter that turns this into a real cv::Feature2D subclass when OpenCV exists.
// Header-only; link with libefx_hip.so and the HIP runtime.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/efx.h"

namespace efx {

class Exception : public std::runtime_error {          // cv::Exception stand-in
public:
    Exception(int code_, const std::string& msg) : std::runtime_error(msg), code(code_) {}
    int code;
};

using KeyPoint = efx_keypoint;

struct HostImage { const uint8_t* data; int rows, cols; size_t step; };
struct DeviceImage { const uint8_t* data; int rows, cols; size_t step; };

class DeviceMatrix {                                     // grow-only, like GpuMat::create + DeviceBuffer
public:
    DeviceMatrix() = default;
    DeviceMatrix(const DeviceMatrix&) = delete;
    DeviceMatrix& operator=(const DeviceMatrix&) = delete;
    DeviceMatrix(DeviceMatrix&& o) noexcept : rows(o.rows), cols(o.cols), elemSize(o.elemSize), step(o.step), data_(o.data_), bytes_(o.bytes_)
    {
        o.data_ = nullptr; o.bytes_ = 0; o.rows = o.cols = 0;
    }
    ~DeviceMatrix() { release(); }
    void create(int rows_, int cols_, int elem_size)
    {
        const size_t step_ = ((size_t)cols_ * elem_size + 255) / 256 * 256;
        const size_t need = step_ * (size_t)(rows_ > 0 ? rows_ : 1);
        if (need > bytes_) {
            release();
            if (hipMalloc(&data_, need) != hipSuccess) throw Exception(EFX_ERR_NOMEM, "hipMalloc failed");
            bytes_ = need;
        }
        rows = rows_; cols = cols_; step = step_; elemSize = elem_size;
    }
    void release() { if (data_) (void)hipFree(data_); data_ = nullptr; bytes_ = 0; rows = cols = 0; }
    void* data() const { return data_; }
    bool empty() const { return rows == 0 || cols == 0; }
    int rows = 0, cols = 0, elemSize = 1;
    size_t step = 0;
private:
    void* data_ = nullptr;
    size_t bytes_ = 0;
};

class EfficientFeatures {
public:
    static const int LOCATION_ROW = 0, RESPONSE_ROW = 1, ANGLE_ROW = 2, OCTAVE_ROW = 3, SIZE_ROW = 4, ROWS_COUNT = 5;
    enum DescriptorType { BAD_256, BAD_512, HASH_SIFT_256, HASH_SIFT_512 };

    static std::shared_ptr<EfficientFeatures> create(int nfeatures = 5000, float scaleFactor = 1.2f, int nlevels = 8,
                                                     int firstLevel = 0, int fastThreshold = 20, int nonmaxRadius = 15,
                                                     DescriptorType dtype = HASH_SIFT_256)
    {
        efx_params p{ nfeatures, scaleFactor, nlevels, firstLevel, fastThreshold, nonmaxRadius, (int)dtype };
        efx_context* c = nullptr;
        const int rc = efx_create(&p, &c);
        if (rc != EFX_OK) throw Exception(rc, efx_last_error(nullptr));
        return std::shared_ptr<EfficientFeatures>(new EfficientFeatures(c));
    }
    ~EfficientFeatures() { efx_destroy(ctx_); if (count_) (void)hipFree(count_); }

    // ---- synchronous, host images (the cv::Mat branches) ----
    void detect(const HostImage& image, std::vector<KeyPoint>& keypoints)
    {
        const int cap = getMaxFeatures();
        keypoints.resize((size_t)(cap > 0 ? cap : 1));
        int n = 0;
        check(efx_detect(ctx_, image.data, image.rows, image.cols, image.step, keypoints.data(), cap, &n));
        keypoints.resize((size_t)n);
    }
    void compute(const HostImage& image, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& descriptors)
    {
        descriptors.assign(keypoints.size() * (size_t)descriptorSize(), 0);
        if (keypoints.empty()) return;                                    // release() branch, cuda_bad.cpp:51-56
        check(efx_compute(ctx_, image.data, image.rows, image.cols, image.step, keypoints.data(), (int)keypoints.size(),
                          descriptors.data(), (size_t)descriptorSize()));
    }
    // Feature2D::detectAndCompute(image, mask, keypoints, descriptors, useProvidedKeypoints).  The reference ignores the
    // mask and asserts !useProvidedKeypoints (.cpp:225-229); here both work.  mask.data == nullptr: none.
    void detectAndCompute(const HostImage& image, const HostImage& mask, std::vector<KeyPoint>& keypoints,
                          std::vector<uint8_t>& descriptors, bool useProvidedKeypoints = false)
    {
        int n = (int)keypoints.size();
        if (useProvidedKeypoints) {
            descriptors.assign(keypoints.size() * (size_t)descriptorSize(), 0);
            check(efx_detect_and_compute_ex(ctx_, image.data, image.rows, image.cols, image.step, nullptr, 0, keypoints.data(),
                                            descriptors.data(), (size_t)descriptorSize(), n, &n, 1));
            return;
        }
        const int cap = getMaxFeatures();
        keypoints.resize((size_t)(cap > 0 ? cap : 1));
        descriptors.assign(keypoints.size() * (size_t)descriptorSize(), 0);
        check(efx_detect_and_compute_ex(ctx_, image.data, image.rows, image.cols, image.step, mask.data, mask.step, keypoints.data(),
                                        descriptors.data(), (size_t)descriptorSize(), cap, &n, 0));
        keypoints.resize((size_t)n);
        descriptors.resize((size_t)n * descriptorSize());
    }
    void detectAndCompute(const HostImage& image, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& descriptors,
                          bool useProvidedKeypoints = false)
    {
        detectAndCompute(image, HostImage{ nullptr, 0, 0, 0 }, keypoints, descriptors, useProvidedKeypoints);
    }

    // ---- asynchronous, device images.  `keypoints` becomes a 5 x nfeatures float matrix (capacity, not exact N:
    //      no host sync happens); lastCount() gives N after the stream was synchronised. ----
    void detectAsync(const DeviceImage& image, DeviceMatrix& keypoints, hipStream_t stream = nullptr)
    {
        const int cap = getMaxFeatures();
        keypoints.create(ROWS_COUNT, cap > 0 ? cap : 1, 4);
        check(efx_detect_async(ctx_, image.data, image.rows, image.cols, image.step, keypoints.data(), keypoints.step, cap,
                               countPtr(), stream));
    }
    // detectAndComputeAsync(image, mask, keypoints, descriptors, useProvidedKeypoints, stream), cuda_efficient_features.h:66-73.
    // useProvidedKeypoints: the first nProvided columns of `keypoints` are described (spec S13), nothing is detected.
    void detectAndComputeAsync(const DeviceImage& image, const DeviceImage& mask, DeviceMatrix& keypoints, DeviceMatrix& descriptors,
                               bool useProvidedKeypoints = false, hipStream_t stream = nullptr, int nProvided = 0)
    {
        if (useProvidedKeypoints) {
            descriptors.create(nProvided > 0 ? nProvided : 1, descriptorSize(), 1);
            check(efx_compute_provided_async(ctx_, image.data, image.rows, image.cols, image.step, keypoints.data(), keypoints.step,
                                             nProvided, static_cast<uint8_t*>(descriptors.data()), descriptors.step, stream));
            return;
        }
        const int cap = getMaxFeatures();
        keypoints.create(ROWS_COUNT, cap > 0 ? cap : 1, 4);
        descriptors.create(cap > 0 ? cap : 1, descriptorSize(), 1);
        check(efx_detect_and_compute_masked_async(ctx_, image.data, image.rows, image.cols, image.step, mask.data, mask.step,
                                                  keypoints.data(), keypoints.step, static_cast<uint8_t*>(descriptors.data()),
                                                  descriptors.step, cap, countPtr(), stream));
    }
    void detectAndComputeAsync(const DeviceImage& image, DeviceMatrix& keypoints, DeviceMatrix& descriptors,
                               bool useProvidedKeypoints = false, hipStream_t stream = nullptr, int nProvided = 0)
    {
        detectAndComputeAsync(image, DeviceImage{ nullptr, 0, 0, 0 }, keypoints, descriptors, useProvidedKeypoints, stream, nProvided);
    }
    // keypoints: 5 x n matrix in the detector's layout (size forced to 31: cuda_efficient_features.cu:260)
    void computeAsync(const DeviceImage& image, const DeviceMatrix& keypoints, int n, DeviceMatrix& descriptors, hipStream_t stream = nullptr)
    {
        descriptors.create(n > 0 ? n : 1, descriptorSize(), 1);
        check(efx_compute_async(ctx_, image.data, image.rows, image.cols, image.step, keypoints.data(), keypoints.step, n,
                                static_cast<uint8_t*>(descriptors.data()), descriptors.step, stream));
    }
    int lastCount() const { int n = 0; check(efx_last_count(ctx_, &n)); return n; }
    // void frames this object has reported: 0 since round 6 -- the scratch arenas hold the reference's own 10 % candidate cap, so a
    // frame of any corner density is complete on the first call (include/efx.h, "scratch arenas"); kept for earlier callers
    int overflowEvents() const { return efx_overflow_events(ctx_); }
    // The loop of samples/sample_image_sequence.cpp:70-105 over frames of one size as ONE call: the frames go through one launch of
    // every kernel (up to 16 per launch chain), each frame's results equal detectAndComputeAsync's on it.  keypoints[i] /
    // descriptors[i] are created here (5 x maxFeatures, maxFeatures x descriptorSize); counts[i]: device ints receiving N_i.
    void detectAndComputeBatchAsync(const std::vector<DeviceImage>& images, std::vector<DeviceMatrix>& keypoints,
                                    std::vector<DeviceMatrix>& descriptors, const std::vector<int*>& counts, hipStream_t stream = nullptr)
    {
        const size_t n = images.size();
        if (n == 0) return;
        if (counts.size() != n) throw Exception(EFX_ERR_BAD_ARG, "one device count per frame");
        keypoints.resize(n); descriptors.resize(n);
        const int cap = getMaxFeatures();
        std::vector<const uint8_t*> img(n); std::vector<void*> kp(n); std::vector<uint8_t*> de(n);
        for (size_t i = 0; i < n; i++) {
            if (images[i].rows != images[0].rows || images[i].cols != images[0].cols || images[i].step != images[0].step)
                throw Exception(EFX_ERR_BAD_ARG, "the frames of a batch have one size and pitch");
            keypoints[i].create(ROWS_COUNT, cap > 0 ? cap : 1, 4);
            descriptors[i].create(cap > 0 ? cap : 1, descriptorSize(), 1);
            img[i] = images[i].data; kp[i] = keypoints[i].data(); de[i] = static_cast<uint8_t*>(descriptors[i].data());
        }
        efx_context* ctxs[1] = { ctx_ };
        void* streams[1] = { stream };
        check(efx_detect_and_compute_batch_async(ctxs, streams, 1, img.data(), (int)n, images[0].rows, images[0].cols, images[0].step,
                                                 kp.data(), keypoints[0].step, de.data(), descriptors[0].step, cap, counts.data()));
    }
    // device blocks of destroyed objects are cached process-wide (at most EFX_BLOCK_CACHE_MB, default 1 GB): give them back
    static size_t trimMemory() { return efx_trim_memory(); }

    // convert (cuda_efficient_features.cpp:323-349): downloads the 5 x n matrix and fills KeyPoints
    void convert(const DeviceMatrix& gpu_keypoints, int n, std::vector<KeyPoint>& keypoints) const
    {
        keypoints.resize((size_t)n);
        if (n == 0) return;
        std::vector<uint8_t> tmp(gpu_keypoints.step * ROWS_COUNT);
        if (hipMemcpy(tmp.data(), gpu_keypoints.data(), tmp.size(), hipMemcpyDeviceToHost) != hipSuccess)
            throw Exception(EFX_ERR_HIP, "download failed");
        check(efx_convert(tmp.data(), gpu_keypoints.step, n, keypoints.data()));
    }

    int descriptorSize() const { return efx_descriptor_size(ctx_); }
    int descriptorType() const { return efx_descriptor_dtype(ctx_); }
    int defaultNorm() const { return efx_default_norm(ctx_); }

    void setMaxFeatures(int v) { check(efx_set_max_features(ctx_, v)); }       int getMaxFeatures() const { return efx_get_max_features(ctx_); }
    void setScaleFactor(float v) { check(efx_set_scale_factor(ctx_, v)); }     float getScaleFactor() const { return efx_get_scale_factor(ctx_); }
    void setNLevels(int v) { check(efx_set_nlevels(ctx_, v)); }                int getNLevels() const { return efx_get_nlevels(ctx_); }
    void setFirstLevel(int v) { check(efx_set_first_level(ctx_, v)); }         int getFirstLevel() const { return efx_get_first_level(ctx_); }
    void setFastThreshold(int v) { check(efx_set_fast_threshold(ctx_, v)); }   int getFastThreshold() const { return efx_get_fast_threshold(ctx_); }
    void setNonmaxRadius(int v) { check(efx_set_nonmax_radius(ctx_, v)); }     int getNonmaxRadius() const { return efx_get_nonmax_radius(ctx_); }
    void setDescriptorType(DescriptorType v) { check(efx_set_descriptor_type(ctx_, (int)v)); }
    DescriptorType getDescriptorType() const { return (DescriptorType)efx_get_descriptor_type(ctx_); }

    efx_context* handle() const { return ctx_; }

private:
    explicit EfficientFeatures(efx_context* c) : ctx_(c) {}
    void check(int rc) const { if (rc != EFX_OK) throw Exception(rc, efx_last_error(ctx_)); }
    int* countPtr()
    {
        if (!count_ && hipMalloc(reinterpret_cast<void**>(&count_), sizeof(int)) != hipSuccess) throw Exception(EFX_ERR_NOMEM, "hipMalloc failed");
        return count_;
    }
    efx_context* ctx_;
    int* count_ = nullptr;
};

// cv::cuda::BAD / cv::cuda::HashSIFT (EfficientDescriptorsAsync, cuda_efficient_descriptors.h:27-57)
class EfficientDescriptorsAsync {
public:
    enum { SIZE_512_BITS = 100, SIZE_256_BITS = 101 };
    virtual ~EfficientDescriptorsAsync() { efx_describer_destroy(d_); }
    void compute(const HostImage& image, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& descriptors)
    {
        descriptors.assign(keypoints.size() * (size_t)descriptorSize(), 0);
        if (keypoints.empty()) return;
        check(efx_describer_compute(d_, image.data, image.rows, image.cols, image.step, keypoints.data(), (int)keypoints.size(),
                                    descriptors.data(), (size_t)descriptorSize()));
    }
    void computeAsync(const DeviceImage& image, const DeviceMatrix& keypoints5xN, int n, DeviceMatrix& descriptors, hipStream_t stream = nullptr)
    {
        descriptors.create(n > 0 ? n : 1, descriptorSize(), 1);
        check(efx_describer_compute_async(d_, image.data, image.rows, image.cols, image.step, keypoints5xN.data(), keypoints5xN.step, n,
                                          static_cast<uint8_t*>(descriptors.data()), descriptors.step, stream));
    }
    int descriptorSize() const { return efx_describer_descriptor_size(d_); }
    int descriptorType() const { return 0; }
    int defaultNorm() const { return 6; }
protected:
    explicit EfficientDescriptorsAsync(efx_describer* d) : d_(d) {}
    void check(int rc) const { if (rc != EFX_OK) throw Exception(rc, efx_describer_last_error(d_)); }
    efx_describer* d_;
};

class BAD : public EfficientDescriptorsAsync {
public:
    static std::shared_ptr<BAD> create(float scaleFactor, int nbits = SIZE_256_BITS)
    {
        efx_describer* d = nullptr;
        const int rc = efx_bad_create(scaleFactor, nbits, &d);
        if (rc != EFX_OK) throw Exception(rc, efx_describer_last_error(nullptr));
        return std::shared_ptr<BAD>(new BAD(d));
    }
private:
    explicit BAD(efx_describer* d) : EfficientDescriptorsAsync(d) {}
};

class HashSIFT : public EfficientDescriptorsAsync {
public:
    static std::shared_ptr<HashSIFT> create(float croppingScale, int nbits = SIZE_256_BITS)
    {
        efx_describer* d = nullptr;
        const int rc = efx_hashsift_create(croppingScale, nbits, &d);
        if (rc != EFX_OK) throw Exception(rc, efx_describer_last_error(nullptr));
        return std::shared_ptr<HashSIFT>(new HashSIFT(d));
    }
private:
    explicit HashSIFT(efx_describer* d) : EfficientDescriptorsAsync(d) {}
};


// cv::BFMatcher(NORM_HAMMING) as the samples use it (sample_feature_matching.cpp:99-101, sample_image_sequence.cpp:81)
struct DMatch { int queryIdx, trainIdx, distance; };
class BFMatcher {
public:
    explicit BFMatcher(bool crossCheck = false) : cross_(crossCheck)
    {
        if (efx_matcher_create(&m_) != EFX_OK) throw Exception(EFX_ERR_HIP, efx_matcher_last_error(nullptr));
    }
    ~BFMatcher() { efx_matcher_destroy(m_); }
    BFMatcher(const BFMatcher&) = delete;
    // device descriptors in, host matches out (one stream synchronisation)
    void knnMatch(const DeviceMatrix& query, int nq, const DeviceMatrix& train, int nt, int descBytes,
                  std::vector<std::vector<DMatch>>& matches, hipStream_t stream = nullptr)
    {
        idx_.create(1, 2 * (nq > 0 ? nq : 1), 4); dist_.create(1, 2 * (nq > 0 ? nq : 1), 4);      // tight nq x 2 ints
        check(efx_match_knn2_async(m_, static_cast<const uint8_t*>(query.data()), query.step, nq, static_cast<const uint8_t*>(train.data()),
                                   train.step, nt, descBytes, static_cast<int*>(idx_.data()), static_cast<int*>(dist_.data()), stream));
        std::vector<int> hi((size_t)nq * 2), hd((size_t)nq * 2);
        if (hipStreamSynchronize(stream) != hipSuccess ||
            hipMemcpy(hi.data(), idx_.data(), hi.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hd.data(), dist_.data(), hd.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
            throw Exception(EFX_ERR_HIP, "download failed");
        matches.assign((size_t)nq, {});
        for (int i = 0; i < nq; i++)
            for (int k = 0; k < 2; k++)
                if (hi[2 * i + k] >= 0) matches[i].push_back(DMatch{ i, hi[2 * i + k], hd[2 * i + k] });
    }
    void match(const DeviceMatrix& query, int nq, const DeviceMatrix& train, int nt, int descBytes, std::vector<DMatch>& matches,
               hipStream_t stream = nullptr)
    {
        matches.clear();
        if (!cross_) {
            std::vector<std::vector<DMatch>> knn;
            knnMatch(query, nq, train, nt, descBytes, knn, stream);
            for (auto& v : knn) if (!v.empty()) matches.push_back(v[0]);
            return;
        }
        idx_.create(1, nq > 0 ? nq : 1, 4); dist_.create(1, nq > 0 ? nq : 1, 4);
        check(efx_match_crosscheck_async(m_, static_cast<const uint8_t*>(query.data()), query.step, nq, static_cast<const uint8_t*>(train.data()),
                                         train.step, nt, descBytes, static_cast<int*>(idx_.data()), static_cast<int*>(dist_.data()), stream));
        std::vector<int> hi((size_t)nq), hd((size_t)nq);
        if (hipStreamSynchronize(stream) != hipSuccess ||
            hipMemcpy(hi.data(), idx_.data(), hi.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hd.data(), dist_.data(), hd.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
            throw Exception(EFX_ERR_HIP, "download failed");
        for (int i = 0; i < nq; i++) if (hi[i] >= 0) matches.push_back(DMatch{ i, hi[i], hd[i] });
    }
private:
    void check(int rc) const { if (rc != EFX_OK) throw Exception(rc, efx_matcher_last_error(m_)); }
    efx_matcher* m_ = nullptr;
    bool cross_;
    DeviceMatrix idx_, dist_;
};

// getInputMat's upload as a double-buffered stage (cuda_efficient_features.cpp:71-84): host frame (1, 3 or 4 channels)
// -> device gray frame ordered on `stream`; upload k+1 overlaps the work enqueued for frame k.
class Uploader {
public:
    Uploader() { if (efx_uploader_create(&u_) != EFX_OK) throw Exception(EFX_ERR_HIP, efx_uploader_last_error(nullptr)); }
    ~Uploader() { efx_uploader_destroy(u_); }
    Uploader(const Uploader&) = delete;
    DeviceImage upload(const uint8_t* data, int rows, int cols, size_t step, int channels, hipStream_t stream = nullptr)
    {
        const uint8_t* d = nullptr; size_t pitch = 0;
        const int rc = efx_upload_gray_async(u_, data, rows, cols, step, channels, &d, &pitch, stream);
        if (rc != EFX_OK) throw Exception(rc, efx_uploader_last_error(u_));
        return DeviceImage{ d, rows, cols, pitch };
    }
private:
    efx_uploader* u_ = nullptr;
};

} // namespace efx
