// bad_affine.h -- the per-keypoint record of the BAD describers: the patch -> image affine map of rectifyBoxes
// (modules/efficient_features/src/bad.cpp:115-147), the LDS window that holds every box of the keypoint, the border flag
// (isKeypointInTheBorder, bad.cpp:86-103) and the level's image.  One LANE per keypoint computes it (the double cos / sin of
// bad.cpp:138-139 is ~400 instructions: far too long to run on one lane of a per-keypoint workgroup): bad_affine_kernel for
// the stand-alone describers, the tail of angle_kernel for the detector's keypoints (no launch of its own).
#pragma once
#include "efx_device.h"

// Everything about a keypoint that is uniform over its workgroup, read back as scalars by the describing workgroup.
struct __attribute__((aligned(16))) Affine {
    float m00, m01, m02, m10, m11, m12, s;
    int wx0, wy0, S;            // window [wx0, wx0 + S) x [wy0, wy0 + S); S == 0: the keypoint does not fit (zero descriptor)
    int border;                 // isKeypointInTheBorder (bad.cpp:86-103)
    int level;                  // pyramid level of the keypoint (0 in single-image mode)
    // the level's image, so that the describing workgroup needs ONE dependent load (this record) before its window loads
    const uint8_t* img; int pitch, rows, cols, pad;
};
static_assert(sizeof(Affine) == 80, "Affine is 80 bytes (DescribeLaunch::bad_affine scratch)");

#ifdef __HIPCC__
// kp = {x, y, size, angle}; img / pitch / rows / cols: the image the keypoint lives on; level: its pyramid level
__device__ __forceinline__ Affine efx_bad_affine(float4 kp, const uint8_t* img, int pitch, int rows, int cols, int level,
                                                 float scale_factor, float reach, int smax, int sfixed)
{
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
    A.img = img; A.pitch = pitch; A.pad = 0;
    A.level = level; A.rows = rows; A.cols = cols;
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
    return A;
}
#endif

// window edge for keypoints of at most max_size pixels (host side of the same formula)
static inline int efx_bad_smax_for(float max_size, float scale_factor, float reach)
{
    const float sg = fabsf(scale_factor * max_size / 32.f);
    const int R = (int)floorf(sg * reach + 2.01f);
    return 2 * R + 2;
}
