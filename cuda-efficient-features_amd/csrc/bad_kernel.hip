// bad_kernel.hip -- BAD (box average difference) descriptor for gfx950.
//
// Arithmetic: the reference CPU descriptor, modules/efficient_features/src/bad.cpp:86-157 (border test,
// rectifyBoxes), :166-251 (clamped float path), :320-405 (integer fast path), bit i -> byte i/8 MSB first.
//
// MI355X design: no global integral image and no global blur pass (the reference does both per level:
// cuda_bad.cu:350-363, cuda_efficient_features.cpp:305).  One workgroup per keypoint stages the S x S
// window that can contain all boxes of that keypoint into LDS (~3 KB of u8 from L2 instead of a
// 4 B/px global integral), optionally applies the 7x7 sigma-2 Gaussian of detectAndCompute in LDS
// (spec S6, identical op order to the oracle), builds a window-local integral image in LDS and evaluates
// all box pairs from it.  Box sums are exact integers, so the local integral gives the same sums as the
// global one; the float expressions are evaluated exactly as in bad.cpp (compile with -ffp-contract=off).

#include "efx_device.h"
#include "blur_window.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Everything about a keypoint that is uniform over its workgroup, computed once by one lane of bad_affine_kernel and
// read back as scalars: the affine map (rectifyBoxes), the LDS window origin / size and the border flag.
struct __attribute__((aligned(16))) Affine {
    float m00, m01, m02, m10, m11, m12, s;
    int wx0, wy0, S;            // window [wx0, wx0 + S) x [wy0, wy0 + S); S == 0: the keypoint does not fit (zero descriptor)
    int border;                 // isKeypointInTheBorder (bad.cpp:86-103)
    int level;                  // pyramid level of the keypoint (0 in single-image mode)
};

// rectifyBoxes, bad.cpp:115-147: the patch -> image affine map of every keypoint, one lane per keypoint (the double
// cos/sin of bad.cpp:138-139 is ~400 instructions: far too long to run on one lane of a per-keypoint workgroup)
__global__ __launch_bounds__(256) void bad_affine_kernel(const float4* __restrict__ kp4, const int* __restrict__ kp_level,
                                                         const LevelTable* __restrict__ T, int rows0, int cols0,
                                                         const int* __restrict__ d_count, int n,
                                                         float scale_factor, float reach, int smax, int sfixed, Affine* __restrict__ aff)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int count = d_count ? min(*d_count, n) : n;
    if (i >= count) return;
    const float4 kp = kp4[i];
    const float x = kp.x, y = kp.y, size = kp.z, angle = kp.w;
    Affine A;
    const float s = scale_factor * size / (0.5f * (float)(32 + 32));
    if (angle == -1) {
        A.m00 = s; A.m01 = 0.0f; A.m02 = -0.5f * s * (float)32 + x;
        A.m10 = 0.0f; A.m11 = s; A.m12 = -s * 0.5f * (float)32 + y;
    } else {
        const float cosine = (angle >= 0) ? (float)cos((double)angle * 0.017453292519943295) : 1.f;
        const float sine = (angle >= 0) ? (float)sin((double)angle * 0.017453292519943295) : 0.f;
        A.m00 = s * cosine; A.m01 = -s * sine;
        A.m02 = (-s * cosine + s * sine) * (float)32 * 0.5f + x;
        A.m10 = s * sine; A.m11 = s * cosine;
        A.m12 = (-s * sine - s * cosine) * (float)32 * 0.5f + y;
    }
    A.s = s;
    int rows = rows0, cols = cols0, l = 0;
    if (kp_level) { l = kp_level[i]; rows = T->lv[l].rows; cols = T->lv[l].cols; }
    A.level = l;
    // window geometry: every (clamped) box coordinate of this keypoint lies in [wx0, wx0+S] x [wy0, wy0+S]
    const float sg = scale_factor * size / 32.f;
    // R >= |sg| * reach + 1 covers every box: a centre rounds to within 0.5 of its exact position, a radius grows by at
    // most 0.5, the far integral coordinate is one more, and x - floor(x) < 1
    const int R = (int)floorf(fabsf(sg) * reach + 2.01f);
    const int Srt = 2 * R + 2;
    const bool fits = sfixed ? (Srt == sfixed) : (Srt <= smax && Srt > 0);
    const int S = sfixed ? sfixed : (fits ? Srt : smax);
    const int ix = (int)floorf(x), iy = (int)floorf(y);
    A.wx0 = min(max(ix - R, 0), max(cols - S, 0));
    A.wy0 = min(max(iy - R, 0), max(rows - S, 0));
    A.S = fits ? S : 0;                                // keypoint larger than the caller's max_size: zero descriptor
    // isKeypointInTheBorder, bad.cpp:86-103
    const float sb = scale_factor * size / (float)(32 + 32);
    const float bw = (float)32 * sb * 1.75f, bh = (float)32 * sb * 1.75f;
    A.border = ((x < bw || x + bw >= (float)cols) || (y < bh || y + bh >= (float)rows)) ? 1 : 0;
    aff[i] = A;
}

// LDS plan (dynamic): [ I: (S+1)^2 int32, aliased by raw: (S+6) x RPB u8 | hb: HR x HP float ]   (BlurGeom, blur_window.h)
// SF != 0: every keypoint is known to need exactly an SF x SF window (size 31, scale 1 -> 48: the detector's
// keypoints), so all index arithmetic and loop bounds fold to constants.
template <bool BLUR, int SF>
__global__ __launch_bounds__(256) void bad_kernel(
    const uint8_t* __restrict__ img0, int pitch0, int rows0, int cols0,
    const uint8_t* __restrict__ pyramid, const LevelTable* __restrict__ T,
    const float4* __restrict__ kp4, const int* __restrict__ kp_level, const int* __restrict__ d_count, int n,
    float scale_factor, int smax, const BadParamsDev* __restrict__ P, const Affine* __restrict__ aff, float taps0, float taps1, float taps2, float taps3,
    uint8_t* __restrict__ desc, size_t desc_pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // the grid is sized for the capacity; only the first `count` workgroups have a keypoint.  Neighbouring keypoints go
    // to the same XCD (their windows share L2 lines): chunked over the COUNT, so that a frame with few keypoints still
    // uses all eight XCDs
    const int count = d_count ? min(*d_count, n) : n;
    if ((int)blockIdx.x >= count) return;
    const int kid = xcd_chunked(blockIdx.x, count);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;

    const Affine A = aff[kid];                                   // uniform address: scalar loads
    const uint8_t* img = img0; int pitch = pitch0, rows = rows0, cols = cols0;
    if (kp_level) {
        const int l = A.level;
        if (l > 0) { const LevelDev& L = T->lv[l]; img = pyramid + L.img_off; pitch = L.pitch; rows = L.rows; cols = L.cols; }
        else { rows = T->lv[0].rows; cols = T->lv[0].cols; }
    }
    const int nbits = P->nbits;
    const bool fits = A.S != 0;
    const int S = SF ? SF : (fits ? A.S : smax);
    const int wx0 = A.wx0, wy0 = A.wy0;

    int* I = reinterpret_cast<int*>(smem);                       // (S+1) x (S+1)
    const int IP = S + 1;
    const BlurGeom bg(S);
    uint8_t* raw = smem;                                         // aliases I (dead before I is written)
    size_t ibytes = (size_t)IP * IP * 4;
    if (BLUR && bg.raw_bytes() > ibytes) ibytes = bg.raw_bytes();
    float* hb = reinterpret_cast<float*>(smem + ((ibytes + 15) & ~(size_t)15));

    if (fits) {
        if (BLUR) {
            // 7x7 sigma-2 Gaussian of the window (spec S6) on packed fp32 FMAs; the quantised pixels go straight into
            // the integral's source (zero outside the frame)
            efx_blur_window_lds<256>(img, pitch, rows, cols, wx0, wy0, S, raw, hb, taps0, taps1, taps2, taps3, tid,
                [&](int r, int c, int q0, int q1) {
                    const bool rin = (wy0 + r) < rows;
                    I[(r + 1) * IP + (c + 1)] = (rin && (wx0 + c) < cols) ? q0 : 0;
                    I[(r + 1) * IP + (c + 2)] = (rin && (wx0 + c + 1) < cols) ? q1 : 0;
                });
        } else {
            for (int r = wid; r < S; r += 4) {
                const int gy = wy0 + r;
                const uint8_t* src = img + (size_t)(gy < rows ? gy : 0) * pitch;
                for (int c = lane; c < S; c += 64) {
                    const int gx = wx0 + c;
                    I[(r + 1) * IP + (c + 1)] = (gy < rows && gx < cols) ? (int)src[gx] : 0;
                }
            }
        }
        for (int i = tid; i <= S; i += 256) { I[i] = 0; I[i * IP] = 0; }
        __syncthreads();
        // window-local integral: row prefix, then column prefix
        for (int r = tid; r < S; r += 256) {
            int* p = I + (r + 1) * IP + 1;
            int run = 0;
            for (int c = 0; c < S; c++) { run += p[c]; p[c] = run; }
        }
        __syncthreads();
        for (int c = tid; c < S; c += 256) {
            int* p = I + IP + 1 + c;
            int run = 0;
            for (int r = 0; r < S; r++) { run += p[r * IP]; p[r * IP] = run; }
        }
    }
    __syncthreads();

    const bool border = A.border != 0;
    const int fw = cols + 1, fh = rows + 1;       // integral image dims of the full frame

    for (int b0 = 0; b0 < nbits; b0 += 256) {
        const int b = b0 + tid;
        bool bit = false;
        if (b < nbits && fits) {
            const uint2 bq = P->box[b];
            const float x1f = (float)(bq.x & 31u), x2f = (float)((bq.x >> 5) & 31u);
            const float y1f = (float)((bq.x >> 10) & 31u), y2f = (float)((bq.x >> 15) & 31u);
            // transform, bad.cpp:151-155: CV_ROUNDNUM(x) = (int)(x + 0.5f)
            const int cx1 = (int)((A.m00 * x1f + A.m01 * y1f + A.m02) + 0.5f);
            const int cy1 = (int)((A.m10 * x1f + A.m11 * y1f + A.m12) + 0.5f);
            const int cx2 = (int)((A.m00 * x2f + A.m01 * y2f + A.m02) + 0.5f);
            const int cy2 = (int)((A.m10 * x2f + A.m11 * y2f + A.m12) + 0.5f);
            const int r = (int)((A.s * (float)(bq.x >> 20)) + 0.5f);
            const float thr = __uint_as_float(bq.y);
            if (border) {
                // computeBadResponse, bad.cpp:166-251: boxes clamped to the frame, float means
                int ax1 = cx1 - r; if (ax1 < 0) ax1 = 0; else if (ax1 >= fw - 1) ax1 = fw - 2;
                int ay1 = cy1 - r; if (ay1 < 0) ay1 = 0; else if (ay1 >= fh - 1) ay1 = fh - 2;
                int ax2 = cx1 + r + 1; if (ax2 <= 0) ax2 = 1; else if (ax2 >= fw) ax2 = fw - 1;
                int ay2 = cy1 + r + 1; if (ay2 <= 0) ay2 = 1; else if (ay2 >= fh) ay2 = fh - 1;
                int lx1 = clampi(ax1 - wx0, 0, S), ly1 = clampi(ay1 - wy0, 0, S);
                int lx2 = clampi(ax2 - wx0, 0, S), ly2 = clampi(ay2 - wy0, 0, S);
                const float sum1 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
                const int area1 = (ay2 - ay1) * (ax2 - ax1);
                const float avg1 = sum1 / (float)area1;

                int bx1 = cx2 - r; if (bx1 < 0) bx1 = 0; else if (bx1 >= fw - 1) bx1 = fw - 2;
                int by1 = cy2 - r; if (by1 < 0) by1 = 0; else if (by1 >= fh - 1) by1 = fh - 2;
                int bx2 = cx2 + r + 1; if (bx2 <= 0) bx2 = 1; else if (bx2 >= fw) bx2 = fw - 1;
                int by2 = cy2 + r + 1; if (by2 <= 0) by2 = 1; else if (by2 >= fh) by2 = fh - 1;
                lx1 = clampi(bx1 - wx0, 0, S); ly1 = clampi(by1 - wy0, 0, S);
                lx2 = clampi(bx2 - wx0, 0, S); ly2 = clampi(by2 - wy0, 0, S);
                const float sum2 = (float)(I[ly1 * IP + lx1] + I[ly2 * IP + lx2] - I[ly1 * IP + lx2] - I[ly2 * IP + lx1]);
                const int area2 = (by2 - by1) * (bx2 - bx1);
                const float avg2 = sum2 / (float)area2;
                bit = (avg1 - avg2) <= thr;
            } else {
                // integer fast path, bad.cpp:365-393 (tap coordinates clamped into the frame: spec S9)
                // The window lies inside the frame (or, for frames smaller than the window, the pixels beyond the
                // frame are zero, so the integral is constant there): clamping to the frame and then to the
                // window equals clamping to the window.
                // Detector keypoints (SF != 0: size 31, scale <= 1, not within 27 px of the frame edge on this path): the
                // window is never clamped by the frame and R >= s * reach + 1 puts every tap inside it, so no clamp.
                auto wc = [&](int v) -> int { return SF ? v : clampi(v, 0, S); };
                const int lax1 = wc(cx1 - r - wx0), lay1 = wc(cy1 - r - wy0);
                const int lax2 = wc(cx1 + r + 1 - wx0), lay2 = wc(cy1 + r + 1 - wy0);
                const int lbx1 = wc(cx2 - r - wx0), lby1 = wc(cy2 - r - wy0);
                const int lbx2 = wc(cx2 + r + 1 - wx0), lby2 = wc(cy2 + r + 1 - wy0);
                const int side = 1 + (r << 1);
                const int area_resp = I[lay1 * IP + lax1] + I[lay2 * IP + lax2] - I[lay1 * IP + lax2] - I[lay2 * IP + lax1]
                                    - I[lby1 * IP + lbx1] - I[lby2 * IP + lbx2] + I[lby1 * IP + lbx2] + I[lby2 * IP + lbx1];
                bit = (float)area_resp <= (thr * (float)(side * side));
            }
        }
        // 8 consecutive box pairs -> one byte, MSB first (bad.cpp:349,368)
        const unsigned long long m = __ballot(bit);
        if ((lane & 7) == 0 && b < nbits) {
            const unsigned v = (unsigned)(m >> lane) & 0xffu;
            desc[(size_t)kid * desc_pitch + (b >> 3)] = (uint8_t)(__brev(v) >> 24);
        }
    }
}

} // namespace

void efx_gaussian_taps_host(float taps[7])
{
    // cv::getGaussianKernel(7, 2) (spec S6): exp(-(i-3)^2 / (2 sigma^2)) normalised in double, stored as float
    double e[7], sum = 0;
    for (int i = 0; i < 7; i++) {
        const double x = (double)(i - 3);
        e[i] = exp(-(x * x) / (2.0 * 2.0 * 2.0));
        sum += e[i];
    }
    for (int i = 0; i < 7; i++) taps[i] = (float)(e[i] / sum);
}

static int bad_smax_for(float max_size, float scale_factor, float reach)
{
    const float sg = fabsf(scale_factor * max_size / 32.f);
    const int R = (int)floorf(sg * reach + 2.01f);
    return 2 * R + 2;
}

hipError_t efx_launch_bad(const DescribeLaunch& a, const BadParamsDev* d_params, float reach, hipStream_t stream)
{
    if (a.n <= 0) return hipSuccess;
    const float max_size = a.max_size > 0.f ? a.max_size : (float)EFX_PATCH_SIZE;
    const int S = bad_smax_for(max_size, a.scale_factor, reach);
    size_t lds = (size_t)(S + 1) * (S + 1) * 4;
    if (a.blur) {
        const BlurGeom bg(S);
        if (bg.raw_bytes() > lds) lds = bg.raw_bytes();
        lds = (lds + 15) & ~(size_t)15;
        lds += bg.hb_bytes();
    }
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024 - 64) return hipErrorInvalidValue;    // keypoint window does not fit in LDS
    float t[7];
    efx_gaussian_taps_host(t);
    Affine* aff = static_cast<Affine*>(a.bad_affine);
    const int sfixed = (S == 48 && a.uniform_size) ? 48 : 0;
    hipLaunchKernelGGL(bad_affine_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a.kp4, a.kp_level, a.d_table, a.rows0, a.cols0,
                       a.d_count, a.n, a.scale_factor, reach, S, sfixed, aff);
    if (a.blur) {
        if (S == 48 && a.uniform_size) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<true, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((bad_kernel<true, 48>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                               a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                               a.desc, a.desc_pitch);
            return hipGetLastError();
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((bad_kernel<true, 0>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                           a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                           a.desc, a.desc_pitch);
    } else {
        if (S == 48 && a.uniform_size) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<false, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((bad_kernel<false, 48>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                               a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                               a.desc, a.desc_pitch);
            return hipGetLastError();
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bad_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((bad_kernel<false, 0>), dim3(a.n), dim3(256), lds, stream, a.img0, a.pitch0, a.rows0, a.cols0, a.pyramid,
                           a.d_table, a.kp4, a.kp_level, a.d_count, a.n, a.scale_factor, S, d_params, aff, t[0], t[1], t[2], t[3],
                           a.desc, a.desc_pitch);
    }
    return hipGetLastError();
}
