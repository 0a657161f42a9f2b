/*
 * efx_oracle.h -- CPU ORACLE for the detect / describe hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm, used as the checker by tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  Nothing in the product path
 * (cuda-efficient-features_amd/) may include, link or call it.
 *
 * PARITY STATUS
 *   - Descriptors (BAD, HashSIFT): restate /root/reference/modules/efficient_features/src/{bad,hash_sift}.cpp
 *     line by line (citations at each function).  That code needs OpenCV (absent from this image), so it
 *     cannot be compiled here without writing stand-in headers, which the build rules forbid, and the
 *     reference's own test data (tests/data submodule) is absent.  PINNED ON ONE KNOWN-ANSWER VECTOR PER
 *     DESCRIPTOR TYPE: SURVEY.md Appendix B records FNV-1a-32 hashes of the reference CPU code's output
 *     (LCG-noise 640x480 image, 200 keypoints) for BAD256/BAD512/HashSIFT256/HashSIFT512; this oracle
 *     reproduces all four bit for bit (tests/test_oracle_pins.py).  Nothing else exists to pin against.
 *   - Detector: **parity unpinned**.  It exists in the reference only as CUDA (cuda_fast.cu,
 *     cuda_efficient_features.cu) on top of cv::cuda::resize / createGaussianFilter (not in the tree); no
 *     golden keypoints exist anywhere.  The restatement follows those files; every place where the
 *     reference is nondeterministic or delegates to OpenCV is decided in DESIGN.md ("Spec decisions") and
 *     implemented identically here and in the HIP kernels.
 */
#ifndef EFX_ORACLE_H
#define EFX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFXO_PATCH_SIZE 31      /* cuda_efficient_features.cpp:33 */
#define EFXO_HALF_PATCH 15      /* cuda_efficient_features.cpp:34 */
#define EFXO_CELL 16            /* cuda_efficient_features.cu:35 */
#define EFXO_TILE 64            /* canonical-order tile (spec S1) */
#define EFXO_MAX_LEVELS 32

/* ---------------- descriptors (reference CPU module) ---------------- */

/* cv::integral for CV_8U -> CV_32S, (rows+1)x(cols+1), first row/col zero (bad.cpp:286). */
void efxo_integral(const uint8_t* img, int rows, int cols, int stride, int32_t* out);

/* cv::BAD::compute (bad.cpp:254-405). kps: n x {x, y, size, angle}. boxes: nbits x {x1,x2,y1,y2,r}.
 * desc: n x (nbits/8) bytes, MSB first. */
void efxo_bad_compute(const uint8_t* img, int rows, int cols, int stride,
                      const float* kps, int n, float scale_factor,
                      const int32_t* boxes, const float* thresholds, int nbits,
                      uint8_t* desc);

/* rectifyPatch + warpAffineLinear (hash_sift.cpp:68-138): 32x32 u8 patch of one keypoint. */
void efxo_hashsift_patch(const uint8_t* img, int rows, int cols, int stride,
                         const float* kp4, float crop_scale, uint8_t* patch /*32*32*/);

/* computePatchSIFTs (hash_sift.cpp:200-351): responses n x 129 (element 0 == 1). */
void efxo_hashsift_responses(const uint8_t* img, int rows, int cols, int stride,
                             const float* kps, int n, float crop_scale, float* responses);

/* matmulAndSign (hash_sift.cpp:353-378). W: nbits x 129 float (already converted from double).
 * T (optional, may be NULL): n x nbits pre-threshold values. desc: n x nbits/8. */
void efxo_hashsift_project(const float* responses, int n, const float* W, int nbits,
                           float* T, uint8_t* desc);

/* cv::HashSIFT::compute (hash_sift.cpp:399-426). */
void efxo_hashsift_compute(const uint8_t* img, int rows, int cols, int stride,
                           const float* kps, int n, float crop_scale,
                           const float* W, int nbits, uint8_t* desc);

/* ---------------- detector (reference CUDA module, restated) ---------------- */

/* calcImagePyramid sizes/scales (cuda_efficient_features.cpp:136-157). */
void efxo_pyramid_geometry(int rows, int cols, float scale_factor, int nlevels,
                           int* lrows, int* lcols, float* scales);

/* calcNumFeaturesPerLevel (cuda_efficient_features.cpp:159-174). */
void efxo_level_quotas(int total, float scale_factor, int nlevels, int* quotas);

/* cv::cuda::resize INTER_LINEAR as specified in DESIGN.md S5 (call site cuda_efficient_features.cpp:154). */
int efxo_s5_fused_weights(void);      /* 1: built with -DEFX_S5_FUSED_WEIGHTS=1 (the other reading of spec S5's weights) */
void efxo_resize_linear(const uint8_t* src, int srows, int scols, int sstride,
                        uint8_t* dst, int drows, int dcols, int dstride);

/* 7x7 sigma=2 Gaussian, BORDER_REFLECT_101, spec S6 (call site cuda_efficient_features.cpp:193,305). */
void efxo_gaussian_taps(float taps[7]);
void efxo_gaussian7(const uint8_t* src, int rows, int cols, int sstride, uint8_t* dst, int dstride);

/* FAST-9/16 segment test at one pixel (cuda_fast.cu:33-222); 1 if corner. Needs a 3-px margin. */
int efxo_fast9_at(const uint8_t* img, int stride, int x, int y, int threshold);
/* the 9-arc predicate the FAST test applies to a 16-bit ring mask (== the reference's c_table lookup, cuda_fast.cu:160-166) */
int efxo_has_arc9(unsigned mask16);
/* the U_MAX row table efxo_ic_angle walks (cuda_efficient_features.cu:143), 17 ints */
void efxo_ic_umax(int* out17);

/* All FAST corners inside [border, cols-border) x [border, rows-border) in raster order
 * (cuda_fast.cu:168-222 + mask cuda_efficient_features.cpp:176-182). xy: pairs (x,y).
 * Returns the number found (never more than max_out are written). */
int efxo_fast9_detect(const uint8_t* img, int rows, int cols, int stride, int threshold, int border,
                      int16_t* xy, int max_out);

/* Harris response, spec S4 (cuda_efficient_features.cu:99-139). */
float efxo_harris(const uint8_t* img, int stride, int x, int y);

/* IC_Angle in degrees, spec S7 (cuda_efficient_features.cu:141-172). */
float efxo_ic_angle(const uint8_t* img, int stride, int x, int y);

/* The shared deterministic atan2 (degrees in [0,360]) used by efxo_ic_angle. */
float efxo_atan2_deg(int m01, int m10);

typedef struct {
    int nfeatures;       /* 5000  */
    float scale_factor;  /* 1.2f  */
    int nlevels;         /* 8     */
    int first_level;     /* 0     */
    int fast_threshold;  /* 20    */
    int nonmax_radius;   /* 15    */
} efxo_params;

typedef struct {
    int n_candidates[EFXO_MAX_LEVELS];   /* FAST corners found (before the 10% cap) */
    int n_after_cap[EFXO_MAX_LEVELS];
    int n_after_nms[EFXO_MAX_LEVELS];
    int n_kept[EFXO_MAX_LEVELS];         /* after the per-level quota */
} efxo_stats;

/* EfficientFeaturesImpl::detectAndComputeAsync (cuda_efficient_features.cpp:225-321).
 * kps_out: 5 x capacity floats, row-major with row stride = capacity:
 *   row0 = short2(x,y) bit-packed, row1 = response, row2 = angle(deg), row3 = int32 octave, row4 = size.
 * desc_type: -1 none, 0 BAD256, 1 BAD512, 2 HASHSIFT256, 3 HASHSIFT512 (cuda_efficient_features.h:39-45).
 * params_a / params_b: for BAD the box table (int32) and thresholds (float); for HashSIFT W (float) and NULL.
 * desc_out: capacity x (nbits/8). lvl_xy_out (optional): 2 x capacity int16 level-local coordinates.
 * Returns N (<= capacity) or -1 on bad arguments. Order: level ascending, then canonical order (spec S1). */
int efxo_detect_and_compute(const uint8_t* img, int rows, int cols, int stride,
                            const efxo_params* p, int desc_type,
                            const void* params_a, const void* params_b,
                            float* kps_out, uint8_t* desc_out, int16_t* lvl_xy_out, int capacity,
                            efxo_stats* stats);

/* Pyramid level s (for tests); returns 0 on success. dst must hold lrows*lcols bytes (tight stride). */
int efxo_pyramid_level(const uint8_t* img, int rows, int cols, int stride, float scale_factor, int level,
                       uint8_t* dst);

/* spec S12: detectAndCompute honouring a level-0 mask (NULL = no mask) */
int efxo_detect_and_compute_masked(const uint8_t* img, int rows, int cols, int stride, const uint8_t* mask, int mstride,
                                   const efxo_params* p, int desc_type,
                                   const void* params_a, const void* params_b,
                                   float* kps_out, uint8_t* desc_out, int16_t* lvl_xy_out, int capacity,
                                   efxo_stats* stats);
/* spec S13: detectAndCompute(useProvidedKeypoints = true): describe the given 5 x n keypoint matrix */
int efxo_compute_provided(const uint8_t* img, int rows, int cols, int stride, const efxo_params* p, int desc_type,
                          const void* params_a, const void* params_b, const float* kps, int capacity, int n, uint8_t* desc_out);

/* threads used by the row / candidate / keypoint loops (OpenMP; 1 = single-threaded like the reference CPU module) */
void efxo_set_threads(int n);
int efxo_get_threads(void);

/* CPU model of the HIP kernel's fixed-point histogram sums (test infrastructure for the HashSIFT tolerance) */
/* mode: bit 0 = fixed-point histogram, bit 1 = tree-ordered norms: isolates the two differences of the device arithmetic */
void efxo_hashsift_responses_model(const uint8_t* img, int rows, int cols, int stride,
                                   const float* kps, int n, float crop_scale, int mode, float* responses);
void efxo_hashsift_responses_fixedpoint(const uint8_t* img, int rows, int cols, int stride,
                                        const float* kps, int n, float crop_scale, float* responses);

/* HPatches exporter helpers (samples/hpatches_description.cpp:107-162) */
void efxo_calc_umax(int patch_size, int* umax);
float efxo_fast_atan2(float y, float x);
void efxo_ic_angles(const uint8_t* img, int rows, int cols, int stride, float* kp4, int n, int patch_size);

/* spec S11: BGR / BGRA -> gray (cv::cvtColor 8-bit fixed point) */
void efxo_bgr2gray(const uint8_t* src, int rows, int cols, int sstride, int channels, uint8_t* dst, int dstride);

#ifdef __cplusplus
}
#endif
#endif
