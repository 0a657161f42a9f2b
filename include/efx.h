/*
 * efx.h -- C ABI of the MI355X-native efficient-features library (libefx_hip.so).
 *
 * This is the drop-in boundary for the detect / describe hot path of fixstars/cuda-efficient-features:
 * the entry points are what a binding of the reference's Feature2D facade would call.  Each declaration
 * cites the reference interface it replaces (paths relative to the reference repository).
 *
 *   - plain pointers and sizes only; device pointers are ordinary HIP device addresses;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - every function returns an efx_status; efx_last_error() gives the message of the last failure;
 *   - a context is NOT re-entrant: one context per (thread, stream, device), exactly like an
 *     EfficientFeaturesImpl instance (cuda_efficient_features.cpp:391-403).  Parameter tables are
 *     per-context (the reference's process-global __constant__ tables, cuda_bad.cu:49-50, are not copied).
 *
 * Keypoint matrix layout ("5xN"), identical to the reference (cuda_efficient_features.h:32-37):
 *   row 0 LOCATION  short2 (x, y) bit-packed in 4 bytes      row 3 OCTAVE   int32
 *   row 1 RESPONSE  float                                    row 4 SIZE     float
 *   row 2 ANGLE     float, degrees
 * Rows are `kps_pitch` BYTES apart.  Descriptors are N x (nbits/8) bytes, rows `desc_pitch` bytes apart,
 * type u8, norm HAMMING (cuda_bad.cpp:82-84, cuda_hash_sift.cpp:149-151).
 */
#ifndef EFX_H
#define EFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFX_VERSION 100

typedef enum efx_status {
    EFX_OK = 0,
    EFX_ERR_BAD_ARG = -1,       /* CV_Assert / CV_Error(StsBadArg) in the reference */
    EFX_ERR_UNSUPPORTED = -2,
    EFX_ERR_HIP = -3,           /* a HIP runtime call failed (the reference only printf's: cuda_macro.h:23-28) */
    EFX_ERR_NO_DEVICE = -4,
    EFX_ERR_NOMEM = -5,
    EFX_ERR_OVERFLOW = -6       /* the frame is VOID (N = 0): records in the context's scratch arenas failed their range checks on
                                 * the device.  Since version 100 / round 6 NO frame content can cause this -- the arenas hold the
                                 * reference's own 10 % candidate cap and nothing is allocated on the device ("scratch arenas" at
                                 * efx_detect_async); the code remains for the platform anomaly of DESIGN.md section 7 */
} efx_status;

/* cuda_efficient_features.h:39-45 */
typedef enum efx_descriptor_type {
    EFX_BAD_256 = 0,
    EFX_BAD_512 = 1,
    EFX_HASH_SIFT_256 = 2,
    EFX_HASH_SIFT_512 = 3
} efx_descriptor_type;

/* cuda_efficient_descriptors.h:75-78, 109-112 (BADSize / HashSIFTSize) */
enum { EFX_SIZE_512_BITS = 100, EFX_SIZE_256_BITS = 101 };

/* cuda_efficient_features.h:32-37 */
enum { EFX_LOCATION_ROW = 0, EFX_RESPONSE_ROW = 1, EFX_ANGLE_ROW = 2, EFX_OCTAVE_ROW = 3, EFX_SIZE_ROW = 4,
       EFX_ROWS_COUNT = 5 };

/* Arguments of EfficientFeatures::create (cuda_efficient_features.h:47-48). */
typedef struct efx_params {
    int nfeatures;          /* 5000 */
    float scale_factor;     /* 1.2f */
    int nlevels;            /* 8    */
    int first_level;        /* 0    */
    int fast_threshold;     /* 20   */
    int nonmax_radius;      /* 15   */
    int descriptor_type;    /* EFX_HASH_SIFT_256 */
} efx_params;

/* cv::KeyPoint as the facade hands it out (EfficientFeaturesImpl::convert, cuda_efficient_features.cpp:323-349). */
typedef struct efx_keypoint {
    float x, y;       /* pt */
    float size;
    float angle;      /* degrees, -1 = not applicable */
    float response;
    int octave;
    int class_id;     /* always -1 */
} efx_keypoint;

typedef struct efx_context efx_context;       /* EfficientFeaturesImpl */
typedef struct efx_describer efx_describer;   /* cuda::BAD / cuda::HashSIFT stand-alone describers */

/* Per-stage device counters of the last detect call (diagnostics; no reference equivalent). */
typedef struct efx_level_stats {
    int n_candidates;   /* FAST corners found              */
    int n_after_nms;    /* survivors of radius suppression */
    int n_kept;         /* after the per-level quota       */
} efx_level_stats;

/* ------------------------------------------------------------------------------------------------ */
/* life cycle                                                                                        */

void efx_default_params(efx_params* p);                         /* the defaults of create(), .h:47-48 */
int efx_create(const efx_params* p, efx_context** out);         /* EfficientFeatures::create, .cpp:406-411 */
int efx_destroy(efx_context* ctx);
/* Device memory the context holds (pyramid, tile headers, corner / survivor arenas, keypoint lists), in bytes. */
size_t efx_device_bytes(const efx_context* ctx);                              /* ~EfficientFeatures, .cpp:413-415 */
/* Destroyed (and regrown) contexts and describers hand their device blocks to a process-wide cache that later contexts draw
   from (no hipMalloc / hipFree in a create-destroy loop; the reference's DeviceBuffer arena, src/device_buffer.cpp:29-69, is
   per object).  efx_trim_memory returns the cached blocks to the driver and reports their bytes; efx_cached_bytes reports
   them.  The cache holds at most EFX_BLOCK_CACHE_MB (environment, default 1024); EFX_NO_BLOCK_CACHE=1 disables it.  Cached
   blocks are invisible to other allocators of the process: a deployment that shares the GPU with one (PyTorch, ...) calls
   efx_trim_memory() after tearing contexts down.  Blocks are waited for and freed on the device they were allocated on. */
size_t efx_trim_memory(void);
size_t efx_cached_bytes(void);
const char* efx_last_error(const efx_context* ctx);             /* ctx may be NULL: last create() error */
int efx_version(void);

/* getters / setters, cuda_efficient_features.h:78-97, impl .cpp:355-379 */
int efx_set_max_features(efx_context* ctx, int v);       int efx_get_max_features(const efx_context* ctx);
int efx_set_scale_factor(efx_context* ctx, float v);     float efx_get_scale_factor(const efx_context* ctx);
int efx_set_nlevels(efx_context* ctx, int v);            int efx_get_nlevels(const efx_context* ctx);
int efx_set_first_level(efx_context* ctx, int v);        int efx_get_first_level(const efx_context* ctx);
int efx_set_fast_threshold(efx_context* ctx, int v);     int efx_get_fast_threshold(const efx_context* ctx);
int efx_set_nonmax_radius(efx_context* ctx, int v);      int efx_get_nonmax_radius(const efx_context* ctx);
int efx_set_descriptor_type(efx_context* ctx, int v);    int efx_get_descriptor_type(const efx_context* ctx);

int efx_descriptor_size(const efx_context* ctx);   /* bytes per descriptor, .cpp:351 */
int efx_descriptor_dtype(const efx_context* ctx);  /* 0 == CV_8U, .cpp:352 */
int efx_default_norm(const efx_context* ctx);      /* 6 == cv::NORM_HAMMING, .cpp:353 */

/* ------------------------------------------------------------------------------------------------ */
/* asynchronous device entry points (inputs and outputs resident in HBM)                             */

/* detectAsync (cuda_efficient_features.cpp:215-218).  d_image: rows x cols u8, `pitch` bytes per row.
 * d_keypoints: 5 x capacity matrix (layout above).  d_count: device int receiving N (<= capacity).
 * The mask argument of the reference is accepted and ignored there (.cpp:225-250); it has no parameter here.
 * No host synchronisation happens inside; once the stream has been synchronised, efx_last_count() and
 * efx_last_level_stats() fetch N and the per-level counts from the device (one small blocking copy per call).
 *
 * Scratch arenas (every detect entry point).  The reference keeps at most cvRound(0.1 w h) FAST corners per pyramid level
 * (cuda_efficient_features.cpp:252, cuda_fast.cu:216-219,245: which ones is a race there; here the first ones in canonical order,
 * DESIGN.md S2).  The context's corner and survivor arrays hold exactly that many records per level and nothing is allocated on the
 * device, so a frame of ANY corner density is complete on the first call -- there is no "void frame, repeat the call" any more
 * (rounds 2-5 sized the arenas for a corner density of 1/8 and returned N = 0 once for denser frames).  A context's device memory
 * depends on the frame size only (8K: ~350 MB with a BAD describer), never on what the frames show. */
int efx_detect_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                     void* d_keypoints, size_t kps_pitch, int capacity, int* d_count, void* stream);

/* detectAndComputeAsync (cuda_efficient_features.cpp:225-321) with useProvidedKeypoints == false. */
int efx_detect_and_compute_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                 void* d_keypoints, size_t kps_pitch,
                                 uint8_t* d_descriptors, size_t desc_pitch,
                                 int capacity, int* d_count, void* stream);

/* computeAsync with the 5xN GPU keypoint matrix (cuda_efficient_features.cpp:220-223 ->
 * getKeypointsMat :102-115 -> convertKeypointsKernel .cu:250-263): size is forced to 31, the OCTAVE and
 * SIZE rows are ignored, no blur is applied. */
int efx_compute_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                      const void* d_keypoints, size_t kps_pitch, int n,
                      uint8_t* d_descriptors, size_t desc_pitch, void* stream);

/* computeAsync for keypoints already packed as float4 {x, y, size, angle} on the device
 * (the std::vector<KeyPoint> branch of getKeypointsMat, cuda_efficient_features.cpp:116-128).
 * max_size: an upper bound of the `size` fields (sizes the LDS window; 0 = 31). */
int efx_compute_kp4_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                          const float* d_kp4, int n, float max_size,
                          uint8_t* d_descriptors, size_t desc_pitch, void* stream);

/* N of the last detect / detectAndCompute on this context (valid after the stream was synchronised); behind a batched call: of the
 * batch's LAST frame.  (EFX_ERR_OVERFLOW: see the status code -- not reachable through frame content.) */
int efx_last_count(const efx_context* ctx, int* n);
/* Times efx_last_count / efx_last_level_stats reported a void frame on this context (EFX_ERR_OVERFLOW).  0 in every run since
 * round 6; kept for callers of the earlier contract. */
int efx_overflow_events(const efx_context* ctx);
/* Diagnostics: streams the context currently tracks for its release waits (a block a regrow or the destructor hands back
 * waits for exactly these; the stream of the call that triggered a regrow stays tracked).  No reference counterpart. */
int efx_tracked_streams(const efx_context* ctx);
/* Per-level counters of the last detect call (valid after the stream was synchronised). */
int efx_last_level_stats(const efx_context* ctx, efx_level_stats* stats, int max_levels, int* nlevels);

/* ------------------------------------------------------------------------------------------------ */
/* synchronous host entry points (upload, run, download), the cv::Mat branches of the facade         */

/* detect (cuda_efficient_features.cpp:197-201): h_image is host memory; keypoints returned as structs. */
int efx_detect(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
               efx_keypoint* keypoints, int capacity, int* n);

/* compute (cuda_efficient_features.cpp:203-206): keypoints given as structs (size and angle honoured). */
int efx_compute(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
                const efx_keypoint* keypoints, int n, uint8_t* h_descriptors, size_t desc_pitch);

/* detectAndCompute (cuda_efficient_features.cpp:208-213). */
int efx_detect_and_compute(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
                           efx_keypoint* keypoints, uint8_t* h_descriptors, size_t desc_pitch,
                           int capacity, int* n);

/* convert (cuda_efficient_features.cpp:323-349): HOST 5xN matrix -> keypoint structs. */
int efx_convert(const void* h_keypoints, size_t kps_pitch, int n, efx_keypoint* out);

/* ------------------------------------------------------------------------------------------------ */
/* stand-alone describers: cuda::BAD::create / cuda::HashSIFT::create                                 */
/* (cuda_efficient_descriptors.h:89,120; cuda_bad.cpp:36-98; cuda_hash_sift.cpp:95-167)             */

int efx_bad_create(float scale_factor, int nbits /* EFX_SIZE_*_BITS */, efx_describer** out);
int efx_hashsift_create(float cropping_scale, int nbits /* EFX_SIZE_*_BITS */, efx_describer** out);
int efx_describer_destroy(efx_describer* d);
int efx_describer_descriptor_size(const efx_describer* d);
const char* efx_describer_last_error(const efx_describer* d);

/* EfficientDescriptorsAsync::computeAsync, float4 keypoints on the device. */
int efx_describer_compute_kp4_async(efx_describer* d, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                    const float* d_kp4, int n, float max_size,
                                    uint8_t* d_descriptors, size_t desc_pitch, void* stream);
/* EfficientDescriptorsAsync::computeAsync, 5xN keypoint matrix on the device (size forced to 31). */
int efx_describer_compute_async(efx_describer* d, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                const void* d_keypoints, size_t kps_pitch, int n,
                                uint8_t* d_descriptors, size_t desc_pitch, void* stream);
/* EfficientDescriptorsAsync::compute, host image + keypoint structs. */
int efx_describer_compute(efx_describer* d, const uint8_t* h_image, int rows, int cols, size_t pitch,
                          const efx_keypoint* keypoints, int n, uint8_t* h_descriptors, size_t desc_pitch);

/* HashSIFT intermediate results for tolerance-based parity tests: the 129-vectors
 * (computePatchSIFTs, hash_sift.cpp:333-351) and the pre-threshold projections T (matmulAndSign :353-378).
 * d_responses: n x 129 floats (may be NULL); d_T: n x nbits floats (may be NULL). */
int efx_describer_hashsift_debug_async(efx_describer* d, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                       const float* d_kp4, int n, float max_size,
                                       float* d_responses, float* d_T, void* stream);

/* ------------------------------------------------------------------------------------------------ */
/* brute-force Hamming matcher: cv::BFMatcher(NORM_HAMMING) as the samples use it   */

typedef struct efx_matcher efx_matcher;      /* owns the scratch buffers; one per (thread, stream) */
int efx_matcher_create(efx_matcher** out);
int efx_matcher_destroy(efx_matcher* m);
const char* efx_matcher_last_error(const efx_matcher* m);

/* knnMatch(query, train, matches, 2) (samples/sample_image_sequence.cpp:114-115).  Descriptors are device
 * matrices of desc_bytes (32 or 64) per row.  d_idx / d_dist: nq x 2 ints, nearest first; the distance is the
 * number of differing bits; ties go to the lower train index; -1 where fewer than two train rows exist. */
int efx_match_knn2_async(efx_matcher* m, const uint8_t* d_query, size_t q_pitch, int nq,
                         const uint8_t* d_train, size_t t_pitch, int nt, int desc_bytes,
                         int* d_idx, int* d_dist, void* stream);

/* BFMatcher::create(NORM_HAMMING, crossCheck = true)->match (samples/sample_feature_matching.cpp:99-101):
 * d_match[i] = j if train j is query i's nearest and query i is train j's nearest, else -1;
 * d_dist[i] (may be NULL) = their distance. */
int efx_match_crosscheck_async(efx_matcher* m, const uint8_t* d_query, size_t q_pitch, int nq,
                               const uint8_t* d_train, size_t t_pitch, int nt, int desc_bytes,
                               int* d_match, int* d_dist, void* stream);

/* Batched variant (SURVEY 8b "batched variants (..., nframes) for roofline-sized launches"; the loop of
 * samples/sample_image_sequence.cpp:70-105): nframes independent frames of one size in one call.  Frame i belongs to context
 * ctxs[i % nctx] and stream streams[i % nctx]; the frames of ONE context go through ONE launch of every kernel of the path (frame =
 * blockIdx.y; up to EFX_MAX_BATCH = 16 per launch chain, more in several chains), so a batch of small frames fills the chip like one
 * large frame: 16 FHD frames per launch chain run at 1.9 x the frame rate of 16 single-frame calls.  Every
 * frame's results equal those of efx_detect_and_compute_async on that frame, bit for bit.  d_descriptors may be NULL (detect
 * only); either every frame has a keypoint / descriptor matrix or none.  A context holds its intermediate buffers once per frame of
 * its largest batch (FHD: 27 MB per frame, 4K: 92 MB, 8K: 350 MB).  The describers run batched too (BAD on the blurred levels, HashSIFT's three kernels).  efx_last_count / efx_last_level_stats refer to the batch's last frame of that context.
 * EFX_NO_BATCH=1 (read when a context is created): one single-frame call per frame.  Stops at the first error. */
int efx_detect_and_compute_batch_async(efx_context* const* ctxs, void* const* streams, int nctx,
                                       const uint8_t* const* d_images, int nframes, int rows, int cols, size_t pitch,
                                       void* const* d_keypoints, size_t kps_pitch,
                                       uint8_t* const* d_descriptors, size_t desc_pitch, int capacity, int* const* d_counts);

/* ------------------------------------------------------------------------------------------------ */
/* mask and useProvidedKeypoints: arguments the reference accepts but ignores / asserts  */

/* detectAsync / detectAndComputeAsync honouring `mask` (cuda_efficient_features.cpp:225-250 takes `_mask` and never
 * reads it).  d_mask: 8-bit, rows x cols, non-zero = allowed, NULL = no mask.  A FAST corner of pyramid level s at level
 * coordinates (x, y) exists only if the mask is non-zero at the pixel the keypoint is reported at,
 * ((short)(scale_s x + 0.5f), (short)(scale_s y + 0.5f)) (DESIGN.md spec S12).  d_descriptors may be NULL (detect only). */
int efx_detect_and_compute_masked_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                        const uint8_t* d_mask, size_t mask_pitch, void* d_keypoints, size_t kps_pitch,
                                        uint8_t* d_descriptors, size_t desc_pitch, int capacity, int* d_count, void* stream);
/* detectAndComputeAsync(useProvidedKeypoints = true) (the reference asserts it is false, :229): no detection; the n
 * keypoints of the 5xN matrix are described exactly as detectAndCompute would have described them -- on the blurred
 * pyramid level `octave`, at level coordinates (int)(x / scale + 0.5f), size 31, angle as given (DESIGN.md spec S13).
 * Keypoints whose octave is not a pyramid level get a zero descriptor. */
int efx_compute_provided_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                               const void* d_keypoints, size_t kps_pitch, int n,
                               uint8_t* d_descriptors, size_t desc_pitch, void* stream);
/* Feature2D::detectAndCompute(image, mask, keypoints, descriptors, useProvidedKeypoints) on host buffers
 * (cuda_efficient_features.cpp:208-213).  use_provided_keypoints: *n keypoints are read, only descriptors are written. */
int efx_detect_and_compute_ex(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
                              const uint8_t* h_mask, size_t mask_pitch, efx_keypoint* keypoints,
                              uint8_t* h_descriptors, size_t desc_pitch, int capacity, int* n, int use_provided_keypoints);

/* ------------------------------------------------------------------------------------------------ */
/* input stage: colour -> gray and host -> device upload overlapped with compute   */

/* cv::cvtColor(COLOR_BGR2GRAY / COLOR_BGRA2GRAY) of an 8-bit image already on the device: what the CPU describers
 * (bad.cpp:268-281, hash_sift.cpp:51-66) and the samples (sample_common.cpp:35-45) do on the host before calling
 * the GPU class, which itself only takes CV_8UC1 (cuda_efficient_features.cpp:228).  channels = 3 (BGR) or 4 (BGRA);
 * gray = (3735 B + 19235 G + 9798 R + 16384) >> 15 (OpenCV's 8-bit fixed-point form, DESIGN.md spec S11). */
int efx_cvt_gray_async(const uint8_t* d_src, int rows, int cols, size_t src_pitch, int channels,
                       uint8_t* d_gray, size_t gray_pitch, void* stream);

/* Page-locked host memory (cv::cuda::HostMem): frames decoded into it upload by DMA without a staging copy. */
int efx_host_alloc(size_t bytes, void** out);
int efx_host_free(void* p);

/* getInputMat's upload (cuda_efficient_features.cpp:71-84) as a double-buffered stage: efx_upload_gray_async copies a
 * host frame (1, 3 or 4 channels, 8 bit) to one of two device slots on an internal copy stream and makes `stream` wait
 * for it (plus the colour conversion, which runs on `stream`); *d_gray / *gray_pitch is the device gray frame, valid
 * for work enqueued on `stream` until the second-next upload on this uploader.  Upload k+1 overlaps with whatever was
 * enqueued on `stream` for frame k.  Pageable host memory is staged through pinned chunks (host memcpy bound);
 * memory from efx_host_alloc goes by DMA directly. */
typedef struct efx_uploader efx_uploader;
int efx_uploader_create(efx_uploader** out);
int efx_uploader_destroy(efx_uploader* u);
const char* efx_uploader_last_error(const efx_uploader* u);
int efx_upload_gray_async(efx_uploader* u, const uint8_t* h_image, int rows, int cols, size_t pitch, int channels,
                          const uint8_t** d_gray, size_t* gray_pitch, void* stream);
/* Slot reuse contract.  By default the uploader assumes that everything that reads frame k was enqueued on the `stream`
 * frame k was uploaded for, BEFORE efx_upload_gray_async is called for frame k+1.  A caller that pipelines differently
 * (uploads ahead, or consumes on another stream) calls efx_uploader_release(u, d_gray_k, s) after enqueuing the last
 * reader of frame k on stream s: the slot is then recycled only behind that point.
 * efx_uploader_wait_uploaded blocks the host until frame d_gray's host-to-device copy has completed, i.e. until the
 * caller may overwrite the (page-locked) host buffer it came from. */
int efx_uploader_release(efx_uploader* u, const uint8_t* d_gray, void* stream);
int efx_uploader_wait_uploaded(efx_uploader* u, const uint8_t* d_gray);

/* EfficientDescriptors::compute with a colour host image (bad.cpp:268-281 accepts 8UC1 / 8UC3 / 8UC4). */
int efx_describer_compute_color(efx_describer* d, const uint8_t* h_image, int rows, int cols, size_t pitch, int channels,
                                const efx_keypoint* keypoints, int n, uint8_t* h_descriptors, size_t desc_pitch);

/* ------------------------------------------------------------------------------------------------ */
/* HPatches exporter helpers: samples/hpatches_description.cpp                     */

/* ICAngles (hpatches_description.cpp:128-162): intensity-centroid angle of the circular patch of `patch_size` pixels
 * around (floor(x), floor(y)) of every keypoint, cv::fastAtan2 of the integer moments, degrees in [0, 360).  d_kp4 is
 * n x {x, y, size, angle} floats on the device; only `angle` is written.  Pixels outside the image count as 0. */
int efx_ic_angles_async(const uint8_t* d_image, int rows, int cols, size_t pitch, float* d_kp4, int n, int patch_size, void* stream);
int efx_ic_angles(const uint8_t* h_image, int rows, int cols, size_t pitch, efx_keypoint* keypoints, int n, int patch_size);
/* saveDescriptors (hpatches_description.cpp:76-105): n descriptors -> text, one line each, bits MSB first separated by
 * commas.  Returns the number of characters (n * nbytes * 16); out == NULL only sizes; -1 on bad arguments. */
long efx_descriptors_to_csv(const uint8_t* h_descriptors, int n, int nbytes, size_t desc_pitch, char* out, size_t out_capacity);

/* ------------------------------------------------------------------------------------------------ */
/* introspection used by the parity tests (no reference equivalent)                                  */

/* Per-launch timing of the pipeline's kernels with HIP events on the caller's stream: efx_profile_enable
 * allocates `max_launches` event pairs (0 disables); every following detect / detectAndCompute call records one
 * pair per timed launch until the pairs are used up.  efx_profile_read (after the stream was synchronised)
 * returns the elapsed milliseconds and a code for each recorded launch and rewinds the recorder.  Codes:
 * 0 fast_kernel, 1 harris_kernel, 2 nms_kernel, 3 select + emit + angle kernels, 10 describe stage (BAD or
 * HashSIFT kernels), 100+s resize of pyramid level s+1. */
int efx_profile_enable(efx_context* ctx, int max_launches);
/* Record events only on every `stride`-th detect call (an event pair costs a few microseconds of stream idle time). */
int efx_profile_set_stride(efx_context* ctx, int stride);
/* Which launch groups are timed (default: all).  Bit 0 fast_kernel, 1 harris_kernel, 2 nms_kernel, 3 select + emit + angle,
 * 4 the describe stage, 5 the pyramid kernels.  Every timed launch costs a few microseconds of stream idle time, so a
 * caller that measures inside a throughput run times only what it reports. */
int efx_profile_set_groups(efx_context* ctx, unsigned groups);
int efx_profile_read(efx_context* ctx, float* ms, int* level, int capacity, int* n);

/* Geometry of pyramid level `level` for a rows x cols frame with the context's parameters
 * (calcImagePyramid, cuda_efficient_features.cpp:136-157). */
int efx_level_geometry(const efx_context* ctx, int rows, int cols, int level, int* lrows, int* lcols, float* scale);
/* Copies pyramid level `level` of the LAST processed frame into d_dst (tight rows x cols, dst_pitch bytes). */
int efx_copy_level_async(efx_context* ctx, int level, uint8_t* d_dst, size_t dst_pitch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EFX_H */
