// input_kernels.hip -- the step before the hot path: colour -> gray on the device.
//
// Reference: the GPU class only accepts CV_8UC1 (cuda_efficient_features.cpp:228); colour frames are converted on the
// host with cv::cvtColor(COLOR_BGR2GRAY / COLOR_BGRA2GRAY) by the CPU describers (bad.cpp:268-281,
// hash_sift.cpp:51-66) and the samples (sample_common.cpp:35-45).  cvtColor is third-party arithmetic (OpenCV >= 4.6,
// not in the reference tree); spec S11 fixes it to OpenCV's 8-bit fixed-point form
//     gray = (3735 * B + 19235 * G + 9798 * R + 16384) >> 15        (BY15 / GY15 / RY15, shift 15, round to nearest).
// HBM bound: 3 or 4 bytes read + 1 byte written per pixel; one lane converts 4 neighbouring pixels (dword store).

#include "efx_device.h"

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

} // namespace

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
