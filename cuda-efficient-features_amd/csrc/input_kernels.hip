// input_kernels.hip -- the step before the hot path: colour -> gray on the device.
//
// Reference: the GPU class only accepts CV_8UC1 (cuda_efficient_features.cpp:228); colour frames are converted on the
// host with cv::cvtColor(COLOR_BGR2GRAY / COLOR_BGRA2GRAY) by the CPU describers (bad.cpp:268-281,
// hash_sift.cpp:51-66) and the samples (sample_common.cpp:35-45).  cvtColor is third-party arithmetic (OpenCV >= 4.6,
// not in the reference tree); spec S11 fixes it to OpenCV's 8-bit fixed-point form
//     gray = (3735 * B + 19235 * G + 9798 * R + 16384) >> 15        (BY15 / GY15 / RY15, shift 15, round to nearest).
// HBM bound: 3 or 4 bytes read + 1 byte written per pixel; one lane converts 4 neighbouring pixels (dword store).

#include "efx_device.h"
#include <math.h>

namespace {

__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r)
{
    return (3735u * b + 19235u * g + 9798u * r + 16384u) >> 15;
}

template <int CH>
__global__ __launch_bounds__(256) void cvt_gray_kernel(const uint8_t* __restrict__ src, size_t spitch, int rows, int cols,
                                                       uint8_t* __restrict__ dst, size_t dpitch, int aligned)
{
    const int y = blockIdx.y;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= cols) return;
    const uint8_t* s = src + (size_t)y * spitch + (size_t)x * CH;
    uint8_t* d = dst + (size_t)y * dpitch + x;
    if (aligned && x + 4 <= cols) {
        uint32_t g4;
        if (CH == 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(s);
            g4 = gray_of(v.x & 0xff, (v.x >> 8) & 0xff, (v.x >> 16) & 0xff) | (gray_of(v.y & 0xff, (v.y >> 8) & 0xff, (v.y >> 16) & 0xff) << 8) |
                 (gray_of(v.z & 0xff, (v.z >> 8) & 0xff, (v.z >> 16) & 0xff) << 16) | (gray_of(v.w & 0xff, (v.w >> 8) & 0xff, (v.w >> 16) & 0xff) << 24);
        } else {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(s);        // 12 bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            g4 = gray_of(w0 & 0xff, (w0 >> 8) & 0xff, (w0 >> 16) & 0xff) | (gray_of(w0 >> 24, w1 & 0xff, (w1 >> 8) & 0xff) << 8) |
                 (gray_of((w1 >> 16) & 0xff, w1 >> 24, w2 & 0xff) << 16) | (gray_of((w2 >> 8) & 0xff, (w2 >> 16) & 0xff, w2 >> 24) << 24);
        }
        *reinterpret_cast<uint32_t*>(d) = g4;
    } else {
        for (int k = 0; k < 4 && x + k < cols; k++) d[k] = (uint8_t)gray_of(s[CH * k], s[CH * k + 1], s[CH * k + 2]);
    }
}

// ---- HPatches exporter helper: intensity-centroid angles of arbitrary circular patches ----
// ICAngles, samples/hpatches_description.cpp:128-162 (the sample does this on the host with OpenMP): one wave per
// keypoint, lane l takes the rows v = l, l + 64, ... of the patch; integer moments; cv::fastAtan2's polynomial.
struct UMaxTable { int half; int umax[130]; };

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    // cv::fastAtan2 (OpenCV >= 4.6, mathfuncs_core.simd.hpp), same operation order as the oracle restatement
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__global__ __launch_bounds__(256) void ic_angles_kernel(const uint8_t* __restrict__ img, size_t pitch, int rows, int cols,
                                                        float4* __restrict__ kp4, int n, UMaxTable t)
{
    const int kid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (kid >= n) return;
    const float4 kp = kp4[kid];
    const int cx = (int)floorf(kp.x), cy = (int)floorf(kp.y);
    auto px = [&](int x, int y) -> int { return (x >= 0 && x < cols && y >= 0 && y < rows) ? (int)img[(size_t)y * pitch + x] : 0; };
    int m01 = 0, m10 = 0;
    for (int v = lane; v <= t.half; v += 64) {
        if (v == 0) {
            for (int u = -t.half; u <= t.half; ++u) m10 += u * px(cx + u, cy);
        } else {
            int v_sum = 0;
            const int d = t.umax[v];
            for (int u = -d; u <= d; ++u) {
                const int vp = px(cx + u, cy + v), vm = px(cx + u, cy - v);
                v_sum += vp - vm;
                m10 += u * (vp + vm);
            }
            m01 += v * v_sum;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { m01 += __shfl_xor(m01, o, 64); m10 += __shfl_xor(m10, o, 64); }
    if (lane == 0) kp4[kid].w = fast_atan2_deg((float)m01, (float)m10);
}

} // namespace

hipError_t efx_launch_ic_angles(const uint8_t* img, size_t pitch, int rows, int cols, float4* kp4, int n, int patch_size, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    UMaxTable t;
    const int half = patch_size / 2;
    if (half < 1 || half + 2 > 130) return hipErrorInvalidValue;
    t.half = half;
    // calcUMax, samples/hpatches_description.cpp:107-126
    for (int v = 0; v < 130; v++) t.umax[v] = 0;
    const int vmax = (int)floor((double)((float)half * sqrtf(2.f) / 2 + 1));
    const int vmin = (int)ceil((double)((float)half * sqrtf(2.f) / 2));
    for (int v = 0; v <= vmax; ++v) t.umax[v] = (int)lrint(sqrt((double)half * half - (double)v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    hipLaunchKernelGGL(ic_angles_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, img, pitch, rows, cols, kp4, n, t);
    return hipGetLastError();
}

hipError_t efx_launch_cvt_gray(const uint8_t* src, size_t spitch, int rows, int cols, int channels, uint8_t* dst, size_t dpitch,
                               hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const int aligned = ((((uintptr_t)src) | spitch | ((uintptr_t)dst) | dpitch) & 3u) == 0 && (channels == 3 || ((((uintptr_t)src) | spitch) & 15u) == 0);
    const dim3 grid((cols + 1023) / 1024, rows);
    if (channels == 4)
        hipLaunchKernelGGL((cvt_gray_kernel<4>), grid, dim3(256), 0, stream, src, spitch, rows, cols, dst, dpitch, aligned);
    else
        hipLaunchKernelGGL((cvt_gray_kernel<3>), grid, dim3(256), 0, stream, src, spitch, rows, cols, dst, dpitch, aligned);
    return hipGetLastError();
}
