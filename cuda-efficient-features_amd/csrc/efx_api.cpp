// efx_api.cpp -- C ABI (include/efx.h) over the HIP kernels.  Host-side orchestration of
// EfficientFeaturesImpl::detectAndComputeAsync (cuda_efficient_features.cpp:225-321) re-thought for MI355X:
// all per-level counts stay on the device, every stage is one launch over all levels where possible, and the
// context owns grow-only HBM buffers sized for the frame (the reference's DeviceBuffer arena,
// device_buffer.cpp:29-69, without its mid-pipeline host syncs).

#include "../../include/efx.h"
#include "efx_device.h"
#include "bad_affine.h"

// match_kernels.hip (declared here, not in efx_device.h: that header is one of the headline kernels' stamped sources, profiles/rNN_counters.json)
int efx_knn2_mfma_resident_workgroups(int desc_bytes, int fp4);

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <algorithm>
#include <string>
#include <mutex>
#include <vector>

// learned parameter blobs, embedded by params_embed.S
extern "C" {
extern const unsigned char efx_blob_bad256[], efx_blob_bad512[], efx_blob_hashsift256[], efx_blob_hashsift512[];
}

#define HS_KPAD 132
#define EFX_PACK_AVG 24           // harris_kernel's sparse form: at most this many FAST corners per level-0 tile in the context's last frame
#define EFX_PACK_MIN_TILES 8192   // ... and a launch beyond the several-waves-per-tile regime (EFX_NMS_MID_TILES, detect_kernels.hip)
#define HS_KB 144             // K of the bf16 projection: 129 padded to 9 MFMA steps of 16

namespace {

thread_local std::string g_create_error;

// Process-wide cache of device blocks.  A context's buffers come from it and go back to it; memory returns to the driver
// only through efx_trim_memory().  Contexts that are created and destroyed in a loop therefore run on the SAME device
// pages: no hipMalloc / hipFree (each an implicit device-wide synchronisation and a page-table change) on that path, and
// the one situation in which the 16-process stress showed wrong frames -- the first kernels on freshly mapped memory while
// many processes oversubscribe the GPU, DESIGN.md section 7 -- does not arise after a process's first context.
struct BlockCache {
    struct Block { void* p; size_t bytes; int dev; };      // blocks belong to the device that was current when they were allocated
    std::mutex m;
    std::vector<Block> free_blocks;
    // smallest cached block of at least n bytes that wastes at most 1/16 (or no block: allocate)
    void* take(size_t n, size_t* got, int dev)
    {
        std::lock_guard<std::mutex> g(m);
        int best = -1;
        for (int i = 0; i < (int)free_blocks.size(); i++)
            if (free_blocks[i].dev == dev && free_blocks[i].bytes >= n && free_blocks[i].bytes <= n + n / 16 + 256 && (best < 0 || free_blocks[i].bytes < free_blocks[best].bytes)) best = i;
        if (best < 0) return nullptr;
        void* p = free_blocks[best].p; *got = free_blocks[best].bytes;
        free_blocks.erase(free_blocks.begin() + best);
        return p;
    }
    // the cache keeps at most EFX_BLOCK_CACHE_MB (default 1024: three 8K contexts; 0 switches it off): beyond that the oldest
    // blocks go back to the driver.  Memory held here is invisible to other allocators of the process (PyTorch's caching
    // allocator, ...): efx_trim_memory() returns it, INTEGRATION.md section 4
    void give(void* p, size_t bytes, int dev)
    {
        static const size_t cap = [] { const char* v = getenv("EFX_BLOCK_CACHE_MB"); return (size_t)(v ? atoll(v) : 1024) << 20; }();
        std::lock_guard<std::mutex> g(m);
        free_blocks.push_back({ p, bytes, dev });
        size_t t = 0;
        for (const Block& b : free_blocks) t += b.bytes;
        while (t > cap && !free_blocks.empty()) {
            const Block b = free_blocks.front();
            free_blocks.erase(free_blocks.begin());
            t -= b.bytes;
            free_on(b.p, b.dev);
        }
    }
    // hipFree on the device the block lives on (a process may drive several GPUs)
    static void free_on(void* p, int dev)
    {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        (void)hipFree(p);
        if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
    }
    size_t trim()
    {
        std::lock_guard<std::mutex> g(m);
        size_t t = 0;
        for (const Block& b : free_blocks) { t += b.bytes; free_on(b.p, b.dev); }
        free_blocks.clear();
        return t;
    }
    size_t cached() { std::lock_guard<std::mutex> g(m); size_t t = 0; for (const Block& b : free_blocks) t += b.bytes; return t; }
};
static BlockCache& block_cache() { static BlockCache* c = new BlockCache; return *c; }      // never destroyed: contexts may outlive statics

// Who waits before a block is handed back.  A context knows the streams it has launched on: while one of its calls (or its
// destructor) is on the stack, a block it releases waits for THOSE streams only -- a regrow under the overflow path used to
// stall every stream of the process (hipDeviceSynchronize in every release).  Owners that do not track streams (stand-alone
// describers, the matcher, the uploader) keep the device-wide wait.
struct Quiesce { void (*fn)(void*); void* arg; bool done; };
thread_local Quiesce* tl_quiesce = nullptr;

struct DevBuf {                     // grow-only device allocation
    void* p = nullptr;
    size_t bytes = 0;
    int dev = 0;                    // device the block lives on
    hipError_t reserve(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        release();
        size_t got = 0;
        (void)hipGetDevice(&dev);
        p = block_cache().take(n, &got, dev);
        if (p) { bytes = got; return hipSuccess; }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {                       // out of memory with blocks of other sizes cached: give them back, once
            (void)hipGetLastError();
            if (block_cache().trim() > 0) e = hipMalloc(&p, n);
        }
        if (e == hipSuccess) bytes = n; else p = nullptr;
        // EFX_POISON=1 (tests): fill every new allocation with a pattern, so that a kernel that reads memory nobody wrote
        // fails the parity tests deterministically instead of once in 10^5 frames (fresh allocations are zero pages,
        // recycled ones hold a previous frame's data)
        static const bool poison = getenv("EFX_POISON") != nullptr;
        if (e == hipSuccess && poison) { e = hipMemset(p, 0xA5, n); if (e == hipSuccess) e = hipDeviceSynchronize(); }   // before any non-blocking stream touches it
        return e;
    }
    // The block may still be in use by kernels in flight ON ITS OWN DEVICE (hipFree used to wait for them implicitly): wait
    // there -- not on whatever device is current: one process may drive several GPUs -- then cache it.
    void release()
    {
        if (!p) return;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        if (tl_quiesce) { if (!tl_quiesce->done) { tl_quiesce->fn(tl_quiesce->arg); tl_quiesce->done = true; } }
        else (void)hipDeviceSynchronize();
        static const bool no_cache = getenv("EFX_NO_BLOCK_CACHE") != nullptr;
        if (no_cache) (void)hipFree(p); else block_cache().give(p, bytes, dev);
        if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
        p = nullptr; bytes = 0;
    }
};

struct Describer {                  // cuda::BAD / cuda::HashSIFT state
    int dbg_hs = 0;                 // EFX_DEBUG_HS, read when the describer is created (EFX_DEBUG_BUILD builds only)
    int no_raw = 0;                 // EFX_BAD_NO_RAW (variant knob), read when the describer is created
    int ubox_max_side = 0;          // BAD: largest box edge of the detector-keypoint table
    size_t hs_wb_off = 0;           // HashSIFT: byte offset of the bf16 weight terms inside `params`
    int kind = 0;                   // 0 BAD, 1 HashSIFT
    int nbits = 256;
    float scale = 1.f;              // BAD scaleFactor / HashSIFT croppingScale
    float reach = 0.f;              // BAD: max (centre distance + radius) in patch units
    DevBuf params;                  // BadParamsDev, or W (nbits x 132) + 30x30 weight table
    DevBuf responses;               // HashSIFT scratch, n x 144 bf16; BAD scratch, n x 80 bytes (per-keypoint affine map + window geometry)
    DevBuf kp4;                     // float4 keypoints for the stand-alone / compute paths
    DevBuf img, desc;               // staging for the host entry points
    std::string err;
    void release_all() { params.release(); responses.release(); kp4.release(); img.release(); desc.release(); }
    ~Describer() { release_all(); }
};

int set_err(std::string& dst, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    dst = buf;
    return code;
}

#define HIP_TRY(errstr, call)                                                                         \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return set_err(errstr, EFX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

int nbits_from_enum(int e) { return e == EFX_SIZE_512_BITS ? 512 : (e == EFX_SIZE_256_BITS ? 256 : 0); }

int describer_init(Describer& d, int kind, int nbits, float scale)
{
    d.kind = kind; d.nbits = nbits; d.scale = scale;
    d.dbg_hs = efx_read_knobs().dbg_hs;
    d.no_raw = getenv("EFX_BAD_NO_RAW") != nullptr;
    if (kind == 0) {
        // BAD_Impl ctor, bad.cpp:300-317 / loadBoxPairParams, cuda_bad.cu:318-334 (per-instance here)
        const unsigned char* blob = nbits == 256 ? efx_blob_bad256 : efx_blob_bad512;
        const int32_t* boxes = reinterpret_cast<const int32_t*>(blob);
        const float* thr = reinterpret_cast<const float*>(blob + (size_t)nbits * 5 * 4);
        BadParamsDev* h = new (std::nothrow) BadParamsDev;
        if (!h) return set_err(d.err, EFX_ERR_NOMEM, "out of host memory");
        memset(h, 0, sizeof(*h));
        h->nbits = nbits;
        float reach = 0.f;
        for (int i = 0; i < nbits; i++) {
            const int x1 = boxes[5 * i + 0], x2 = boxes[5 * i + 1], y1 = boxes[5 * i + 2], y2 = boxes[5 * i + 3], r = boxes[5 * i + 4];
            uint32_t tb; memcpy(&tb, &thr[i], 4);
            h->box[i] = make_uint2((uint32_t)(x1 | (x2 << 5) | (y1 << 10) | (y2 << 15) | (r << 20)), tb);
            const float d1 = sqrtf((float)((x1 - 16) * (x1 - 16) + (y1 - 16) * (y1 - 16))) + (float)r;
            const float d2 = sqrtf((float)((x2 - 16) * (x2 - 16) + (y2 - 16) * (y2 - 16))) + (float)r;
            reach = fmaxf(reach, fmaxf(d1, d2));
        }
        h->reach = reach;
        d.reach = reach;
        // the keypoint-independent part of every box pair for the detector's keypoints (size 31): bad.cpp:151-155,393
        {
            const float sz = (float)EFX_PATCH_SIZE;
            const float su = scale * sz / (0.5f * (float)(32 + 32));            // == Affine.s of bad_affine_kernel
            h->ubox_s = su;
            h->ubox_max_side = 0;
            for (int i = 0; i < nbits; i++) {
                const int x1 = boxes[5 * i + 0], x2 = boxes[5 * i + 1], y1 = boxes[5 * i + 2], y2 = boxes[5 * i + 3], r = boxes[5 * i + 4];
                const int rs = (int)((su * (float)r) + 0.5f);
                const int side = 1 + (rs << 1);
                if (side > h->ubox_max_side) h->ubox_max_side = side;
                const float ts = thr[i] * (float)(side * side);
                uint32_t tsb; memcpy(&tsb, &ts, 4);
                // byte offsets in the detector-sized kernels' integral: u16 entries, 50 per row (BAD_J_PITCH, bad_kernel.hip)
                h->ubox[i] = make_uint4((uint32_t)(x1 | (y1 << 8) | (x2 << 16) | (y2 << 24)), (uint32_t)(-2 * rs * (50 + 1)),
                                        (uint32_t)(2 * side) | ((uint32_t)(2 * 50 * side) << 16), tsb);
            }
        }
        d.ubox_max_side = h->ubox_max_side;
        hipError_t e = d.params.reserve(sizeof(BadParamsDev));
        if (e == hipSuccess) e = hipMemcpy(d.params.p, h, sizeof(BadParamsDev), hipMemcpyHostToDevice);
        delete h;
        if (e != hipSuccess) return set_err(d.err, EFX_ERR_HIP, "BAD parameter upload failed: %s", hipGetErrorString(e));
    } else {
        // HashSIFTImpl ctor, hash_sift.cpp:384-397: Mat(nbits,129,CV_64F).convertTo(CV_32F)
        const double* w64 = reinterpret_cast<const double*>(nbits == 256 ? efx_blob_hashsift256 : efx_blob_hashsift512);
        std::vector<float> w((size_t)nbits * HS_KPAD + 1024 + 2 * 511 * 511, 0.f);
        for (int j = 0; j < nbits; j++)
            for (int k = 0; k < 129; k++) w[(size_t)j * HS_KPAD + k] = (float)w64[(size_t)j * 129 + k];
        // Gaussian pixel weights of computePatchSIFT (hash_sift.cpp:220-224,247): host expf, same call as the CPU code.  Stored
        // x 2^17 (exact): the kernel's histogram is 15.17 fixed point and a power of two commutes with all its products
        const float kp_scale = 1.f / 6;
        const float kp_radius = kp_scale * (float)32 * 0.5f;
        const float kernel_sigma = 0.5f * (float)4 * 3.f * kp_radius;
        const float dist_scale = -1.f / ((float)2 * kernel_sigma * kernel_sigma);
        const float cx = 0.5f * (float)30, cy = 0.5f * (float)30;
        // Laid out in the vote loop's order (patch_sift_kernel): thread t owns the pixels (x, y0 + 2 k), k = 0 .. 3, of its cell
        for (int t = 0; t < 256; t++)
            for (int k = 0; k < 4; k++) {
                const int cell = t & 15, wq = t >> 4;
                const int x = 8 * (cell & 3) + (wq & 7), y = 8 * (cell >> 2) + (wq >> 3) + 2 * k;
                if (x >= 30 || y >= 30) continue;
                const float ddx = (float)x - cx, ddy = (float)y - cy;
                w[(size_t)nbits * HS_KPAD + k * 256 + t] = 131072.f * expf(dist_scale * (ddx * ddx + ddy * ddy));
            }
        // orientation bin scaleO * atan2f(dy, dx) and magnitude sqrtf(dx^2 + dy^2) of every integer gradient
        // (hash_sift.cpp:171,254-258), interleaved: one 8-byte gather per pixel
        {
            const float PI_2 = (float)6.283185307179586476925286766559;
            const float scaleO = (float)8 / PI_2;
            float* lut = w.data() + (size_t)nbits * HS_KPAD + 1024;
            for (int dy = -255; dy <= 255; dy++)
                for (int dx = -255; dx <= 255; dx++) {
                    const float fdx = (float)dx, fdy = (float)dy;
                    float* e = lut + 2 * ((size_t)(dy + 255) * 511 + (dx + 255));
                    e[0] = scaleO * atan2f(fdy, fdx);
                    e[1] = sqrtf(fdx * fdx + fdy * fdy);
                }
        }
        // The projection runs on the bf16 matrix cores with W split EXACTLY into three bf16 terms, W = W1 + W2 + W3
        // (8 + 8 + 8 mantissa bits; each term is the round-to-nearest-even bf16 of what is left): the 129-vector is integer
        // valued (0..255, and the leading 1) and therefore exact in bf16, every product is exact in the fp32 accumulator.
        // Layout behind the float tables: [term][bit][HS_KB] bf16, K padded with zeros.
        const size_t wb_off = (w.size() + 3) & ~(size_t)3;         // 16-byte aligned: the kernel loads 8 bf16 at a time
        w.resize(wb_off + (size_t)3 * nbits * HS_KB / 2, 0.f);
        {
            uint16_t* wb = reinterpret_cast<uint16_t*>(w.data() + wb_off);
            auto bf16_rne = [](float f) -> uint16_t {
                uint32_t u; memcpy(&u, &f, 4);
                u += 0x7fffu + ((u >> 16) & 1u);                   // finite inputs only
                return (uint16_t)(u >> 16);
            };
            auto bf16_f = [](uint16_t b) -> float { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
            for (int j = 0; j < nbits; j++)
                for (int k = 0; k < 129; k++) {
                    float rest = w[(size_t)j * HS_KPAD + k];
                    for (int t = 0; t < 3; t++) {
                        const uint16_t b = bf16_rne(rest);
                        wb[((size_t)t * nbits + j) * HS_KB + k] = b;
                        rest = rest - bf16_f(b);                   // exact: the difference fits the float mantissa
                    }
                }
        }
        d.hs_wb_off = wb_off * sizeof(float);
        hipError_t e = d.params.reserve(w.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d.params.p, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) return set_err(d.err, EFX_ERR_HIP, "HashSIFT weight upload failed: %s", hipGetErrorString(e));
    }
    return EFX_OK;
}

// one describe call: keypoints as float4 on the device
int describer_run(Describer& d, std::string& err, DescribeLaunch a, float* dbg_resp, float* dbg_T, hipStream_t stream)
{
    a.scale_factor = d.scale;
    a.nbits = d.nbits;
    a.dbg_hs = d.dbg_hs;
    if (a.n <= 0) return EFX_OK;
    const bool prof = a.prof.begin(10, stream);
    struct ProfEnd { const ProfRec& p; bool on; hipStream_t st; ~ProfEnd() { p.end(on, 10, st); } } prof_end{a.prof, prof, stream};
    if (d.kind == 0) {
        // (behind a batched detect the records of all frames are already there: frame f's at frame_affine_off / f * aff_stride)
        HIP_TRY(err, d.responses.reserve(((size_t)a.n + a.frame_affine_off + (a.nframes > 1 ? (size_t)(a.nframes - 1) * a.aff_stride : 0)) * sizeof(Affine)));
        a.bad_affine = static_cast<Affine*>(d.responses.p) + a.frame_affine_off;
        a.bad_det_tables = d.ubox_max_side <= 16 ? 2 : 1;      // describer_init builds ubox for d.scale and size 31
        a.bad_no_raw = d.no_raw;
        hipError_t e = efx_launch_bad(a, static_cast<const BadParamsDev*>(d.params.p), d.reach, stream);
        if (e == hipErrorInvalidValue) return set_err(err, EFX_ERR_UNSUPPORTED, "keypoint size %.1f needs a window larger than the 160 KB LDS", a.max_size);
        if (e != hipSuccess) return set_err(err, EFX_ERR_HIP, "BAD launch failed: %s", hipGetErrorString(e));
    } else {
        if (a.desc && ((((uintptr_t)a.desc) | a.desc_pitch) & 3u))
            return set_err(err, EFX_ERR_BAD_ARG, "HashSIFT descriptors need a 4-byte aligned base and pitch");
        for (int f = 1; f < a.nframes; f++)
            if (((uintptr_t)a.descs.desc[f]) & 3u) return set_err(err, EFX_ERR_BAD_ARG, "HashSIFT descriptors need a 4-byte aligned base and pitch");
        // 129-vectors, then the per-keypoint records; behind a batched detect: nframes x kp_stride of each
        const size_t nrec = a.nframes > 1 ? (size_t)a.nframes * a.kp_stride : (size_t)a.n;
        HIP_TRY(err, d.responses.reserve(nrec * (HS_KB * sizeof(uint16_t) + EFX_HS_REC_BYTES)));
        HashSiftDev h;
        h.nbits = d.nbits;
        h.W = static_cast<const float*>(d.params.p);
        h.Wb = reinterpret_cast<const uint16_t*>(static_cast<const unsigned char*>(d.params.p) + d.hs_wb_off);
        h.responses = static_cast<uint16_t*>(d.responses.p);
        h.records = static_cast<unsigned char*>(d.responses.p) + nrec * HS_KB * sizeof(uint16_t);     // 288 n: 32-byte aligned
        h.dbg_responses = dbg_resp;
        h.dbg_T = dbg_T;
        hipError_t e = efx_launch_hashsift(a, h, stream);
        if (e == hipErrorInvalidValue) return set_err(err, EFX_ERR_UNSUPPORTED, "keypoint size %.1f needs a window larger than the LDS", a.max_size);
        if (e != hipSuccess) return set_err(err, EFX_ERR_HIP, "HashSIFT launch failed: %s", hipGetErrorString(e));
    }
    return EFX_OK;
}

} // namespace

// where the level blur runs when EFX_BLUR_FORK does not say: -1 = decided per call (detect_common): on the side stream behind
// nms_kernel (3) when the call's stream has nothing pending -- a caller that waits for every frame, whose chip is mostly idle
// under select / emit / angle: 8K one-call latency 0.447 -> 0.426 ms -- and inline (0) otherwise: with frames in flight the side
// stream's two event edges and the slots the blur takes from the next frame's kernels cost 2.5 % of the throughput
#ifndef EFX_BLUR_FORK_DEFAULT
#define EFX_BLUR_FORK_DEFAULT (-1)
#endif

EfxKnobs efx_read_knobs()
{
    EfxKnobs k = {};
    k.no_tower = getenv("EFX_NO_TOWER") != nullptr;
    k.no_resize_stream = getenv("EFX_NO_RESIZE_STREAM") != nullptr;
    k.no_resize_rows = getenv("EFX_NO_RESIZE_ROWS") != nullptr;     // the tiled per-level kernels instead of resize_rows_kernel
    k.tower_max_px = getenv("EFX_TOWER_MAX_PX") ? atoll(getenv("EFX_TOWER_MAX_PX")) : 0;   // 0: the built-in limit of the tower launch (detect_kernels.hip)
    k.no_level_blur = getenv("EFX_NO_LEVEL_BLUR") != nullptr;
    { const char* f = getenv("EFX_BLUR_FORK"); k.blur_fork = f ? atoi(f) : EFX_BLUR_FORK_DEFAULT; }   // DetectLaunch::blur_fork
    k.blur_fork_min_px = getenv("EFX_BLUR_FORK_MIN_PX") ? atoll(getenv("EFX_BLUR_FORK_MIN_PX")) : 0;  // 0: the built-in gate of the per-call fork decision
    k.no_batch = getenv("EFX_NO_BATCH") != nullptr;
    { const char* f = getenv("EFX_PACK"); k.pack = f ? atoi(f) : -1; }      // harris_kernel four tiles per wave: 0 never, 1 always, unset: by the last frame's density       // the batched entry point as a loop of single-frame calls (A/B, parity tests)
    // BAD behind detectAndCompute: every keypoint blurs its own window (A/B, parity tests)
    const char* d = getenv("EFX_DEBUG");
    const char* h = getenv("EFX_DEBUG_HS");
#ifdef EFX_DEBUG_BUILD
    k.dbg = d ? atoi(d) : 0;
    k.dbg_hs = h ? atoi(h) : 0;
    if (k.dbg || k.dbg_hs) fprintf(stderr, "efx: DEBUG BUILD with stage knobs EFX_DEBUG=%d EFX_DEBUG_HS=%d: results are NOT valid\n", k.dbg, k.dbg_hs);
#else
    static bool warned = false;
    if ((d || h) && !warned) { warned = true; fprintf(stderr, "efx: EFX_DEBUG / EFX_DEBUG_HS ignored (library built without -DEFX_DEBUG_BUILD)\n"); }
#endif
    return k;
}

namespace {

int cv_round_f(float v) { return (int)lrintf(v); }
int cv_round_d(double v) { return (int)lrint(v); }
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

struct efx_describer { Describer d; };

struct efx_matcher {
    bool no_mfma = getenv("EFX_MATCH_NO_MFMA") != nullptr;        // variant knob (tests: force the popcount kernel), read when the matcher is created
    bool no_fp4 = getenv("EFX_MATCH_NO_FP4") != nullptr;          // ... the int8 matrix-core kernel instead of the FP4 one
    DevBuf scratch, expanded, a_idx, a_dist, b_idx, b_dist;
    std::string err;
    ~efx_matcher() { scratch.release(); expanded.release(); a_idx.release(); a_dist.release(); b_idx.release(); b_dist.release(); }
};

struct efx_context {
    efx_params p;
    EfxKnobs knobs = efx_read_knobs();   // investigation knobs, read once (efx_device.h)
    Describer desc;                 // describer_ (cuda_efficient_features.cpp:402), rebuilt by setDescriptorType
    std::string err;

    // geometry cache
    int g_rows = -1, g_cols = -1;
    efx_params g_p;
    LevelTable h_table;
    DevBuf d_table, pyramid, hdr, cand, cmax, surv, counters, kp4, kp_level, img, kps, descout, count, maskbuf;
    DevBuf slots, tcount, nsel, rowsum, hist, sel_list;            // round 6: per-tile corner slots / counts, per-row sums, key histograms (efx_device.h)
    bool hist_clean = false;        // the key histograms are zero (nms_kernel adds, select_kernel's leaders read and clear: they stay zero across frames
                                    // unless a call failed half-way or the buffer is new)
    DevBuf rplan; ResizePlanLevel rplan_lv[EFX_MAX_LEVELS];        // resize plan (tables of resize_stream_kernel)
    RowsPlanLaunch rows_plan[EFX_MAX_LEVELS];                       // ... and of resize_rows_kernel, by source level
    DevBuf blurred;                 // blurred copies of the pyramid levels for the BAD describer (blur_levels_kernel), on first use
    hipStream_t side = nullptr;     // side stream + fork / join events of the level blur (DetectLaunch::blur_fork), on first use
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int idle_streak = 0;            // consecutive calls that found their stream idle (where the level blur runs: detect_common)
    int g_frames = 0;               // frames the per-frame buffers were reserved for (batched launches: detect_frames)
    int g_plan_frames = 0;          // frames per launch the row-walking pyramid plan was chunked for (build_rows_plan)
    FrameStride fs = {};            // distance between the frames' copies inside the buffers (efx_device.h)
    int last_frames = 1;            // frames of the last detect call (the summary mirror reads the LAST one)
    Summary* h_mirror = nullptr;    // host copy of the last frame's summary, filled on demand by fetch_summary()
    int n_out_max = 0;              // sum of the active levels' quotas
    std::vector<hipStream_t> streams;   // streams this context has launched on since it last waited for them (ctx_quiesce)
    hipStream_t active_stream = nullptr; bool has_active = false;   // the stream of the call on the stack (QuiesceScope)
    int* h_hint = nullptr;          // pinned host word select_kernel leaves the last frame's level-0 corner count in (+ 1; 0: none yet)
    int* d_hint = nullptr;          // ... its device address
    int overflow_events = 0;        // void frames this context has reported (efx_last_count: arena contents that failed their range checks)
    DetectLaunch last_launch;       // investigation (efx_debug_rerun): the last frame's launch arguments
    bool has_frame = false;
    const uint8_t* last_img0 = nullptr; int last_pitch0 = 0;
    // per-launch timing of the pipeline's kernels (efx_profile_*)
    std::vector<hipEvent_t> prof_start, prof_stop;
    std::vector<int> prof_level;
    int prof_count = 0;
    int prof_stride = 1, prof_calls = 0;   // record events on every prof_stride-th detect call only
    unsigned prof_skip = 0;                // launch groups that are not timed (efx_profile_set_groups)

    static void quiesce_cb(void* p)
    {
        efx_context* c = static_cast<efx_context*>(p);
        bool ok = true;
        for (hipStream_t st : c->streams) ok = ok && hipStreamSynchronize(st) == hipSuccess;
        // the context's own side stream is joined into the call's stream by every call that uses it -- unless that call failed
        // between its fork and its join: wait for it by name
        if (c->side) ok = ok && hipStreamSynchronize(c->side) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }      // e.g. a stream the caller has destroyed meanwhile
        c->streams.clear();
        // the call that triggered this wait (a regrow inside detect / compute) launches on its stream AFTER the wait: that
        // stream stays tracked, or the next release (another stream's regrow, the destructor) would not wait for this
        // call's kernels and hand blocks they still use to the process-wide cache
        if (c->has_active) c->streams.push_back(c->active_stream);
    }
    void note_stream(hipStream_t st)
    {
        for (hipStream_t q : streams) if (q == st) return;
        // a caller that makes a stream per call: forget streams that have drained (or no longer exist) before the list grows
        if (streams.size() >= 32) {
            size_t k = 0;
            for (hipStream_t q : streams) {
                const hipError_t e = hipStreamQuery(q);
                if (e == hipErrorNotReady) streams[k++] = q; else if (e != hipSuccess) (void)hipGetLastError();
            }
            streams.resize(k);
        }
        streams.push_back(st);
    }
    ~efx_context()
    {
        Quiesce q = { &efx_context::quiesce_cb, this, false };
        Quiesce* prev = tl_quiesce;
        tl_quiesce = &q;
        desc.release_all();                                // the describer's blocks are this context's: same wait
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        blurred.release();
        rplan.release(); d_table.release(); pyramid.release(); hdr.release(); cand.release(); cmax.release(); surv.release(); counters.release();
        slots.release(); tcount.release(); nsel.release(); rowsum.release(); hist.release(); sel_list.release();
        kp4.release(); kp_level.release(); img.release(); kps.release(); descout.release(); count.release(); maskbuf.release();
        delete h_mirror;
        if (h_hint) (void)hipHostFree(h_hint);
        for (hipEvent_t e : prof_start) (void)hipEventDestroy(e);
        for (hipEvent_t e : prof_stop) (void)hipEventDestroy(e);
        tl_quiesce = prev;
    }
};

namespace {
// while one of these is alive on the calling thread, blocks released by `c` (regrow, describer rebuild) wait for c's streams
struct QuiesceScope {
    Quiesce q; Quiesce* prev;
    efx_context* ctx; hipStream_t prev_stream; bool prev_has;
    QuiesceScope(efx_context* c, hipStream_t stream)
        : q{ &efx_context::quiesce_cb, c, false }, prev(tl_quiesce), ctx(c), prev_stream(c->active_stream), prev_has(c->has_active)
    {
        tl_quiesce = &q;
        c->active_stream = stream; c->has_active = true;
        c->note_stream(stream);
    }
    ~QuiesceScope() { tl_quiesce = prev; ctx->active_stream = prev_stream; ctx->has_active = prev_has; }
};
}

namespace {

int validate_params(const efx_params& p, std::string& err)
{
    if (p.nfeatures < 0) return set_err(err, EFX_ERR_BAD_ARG, "nfeatures must be >= 0");
    if (!(p.scale_factor > 1.0f)) return set_err(err, EFX_ERR_BAD_ARG, "scale_factor must be > 1");
    if (p.nlevels < 1 || p.nlevels > EFX_MAX_LEVELS) return set_err(err, EFX_ERR_BAD_ARG, "nlevels must be in [1, %d]", EFX_MAX_LEVELS);
    if (p.first_level < 0) return set_err(err, EFX_ERR_BAD_ARG, "first_level must be >= 0");
    if (p.fast_threshold < 0 || p.fast_threshold > 255) return set_err(err, EFX_ERR_BAD_ARG, "fast_threshold must be in [0, 255]");
    if (p.nonmax_radius < 0 || p.nonmax_radius > 1024) return set_err(err, EFX_ERR_BAD_ARG, "nonmax_radius must be in [0, 1024]");
    if (p.descriptor_type < EFX_BAD_256 || p.descriptor_type > EFX_HASH_SIFT_512) return set_err(err, EFX_ERR_BAD_ARG, "unknown descriptor type %d", p.descriptor_type);
    return EFX_OK;
}

#ifndef EFX_ROWS_SPLIT_DEFAULT
#define EFX_ROWS_SPLIT_DEFAULT "2,2,3"
#endif
// Tables of resize_rows_kernel (detect_kernels.hip): which levels share a launch (`split`: levels made per launch, from level 1
// up), and per launch the column / row / strip / chunk tables.  Everything the kernel relies on is checked here; a launch whose
// geometry does not fit gets nlev = 0 and its levels go through the tiled per-level kernels.  The float expressions are spec
// S5's (the ones of the plan above).
void build_rows_plan(const LevelTable& T, int nlevels, int nframes, std::vector<int>& blob, RowsPlanLaunch* out)
{
    auto fbits = [](float f) { int i; memcpy(&i, &f, 4); return i; };
    memset(out, 0, sizeof(RowsPlanLaunch) * EFX_MAX_LEVELS);
    static const int wave_target = getenv("EFX_ROWS_WAVES") ? atoi(getenv("EFX_ROWS_WAVES")) : 3072;      // investigation: waves per launch aimed at
    // levels per launch: every launch costs ~5 us that no wave sees plus ~2 us of set-up, while a deeper launch spends more lanes on
    // halo columns and idle upper-level lanes and holds fewer waves (registers).  Measured at 8K, chain us per frame
    // (tools/microbench/rows_split.sh): 2,2,2,1 55.1 | 2,2,3 51.9 | 2,3,2 52.8 | 3,2,2 59.1 | 3,4 58.1 | 4,3 60.5 -- all bit-identical
    const char* split_env = getenv("EFX_ROWS_SPLIT");              // variant knob (A/B, parity tests), read whenever a geometry is built: e.g. "2,2,2,1"
    std::vector<int> split;
    for (const char* p = split_env ? split_env : EFX_ROWS_SPLIT_DEFAULT; *p;) {
        split.push_back(std::max(1, std::min(RW_MAXLEV, atoi(p))));
        while (*p && *p != ',') p++;
        if (*p == ',') p++;
    }
    auto align4 = [&]() { while (blob.size() & 3) blob.push_back(0); };
    size_t si = 0;
    for (int s = 0; s + 1 < nlevels;) {
        int nlev = si < split.size() ? split[si] : split.empty() ? 2 : split.back();
        si++;
        while (nlev > 1 && (s + nlev >= nlevels || T.lv[s + nlev].rows < 1 || T.lv[s + nlev].cols < 4)) nlev--;
        const LevelDev& A = T.lv[s];
        bool ok = A.rows >= 2 && A.cols >= 8;
        for (int k = 1; k <= nlev && ok; k++) {
            const LevelDev& D = T.lv[s + k];
            ok = D.rows >= 1 && D.cols >= 4 && D.fx >= 1.f && D.fx <= 1.9f && D.fy >= 1.f && D.fy <= 1.9f;
        }
        if (!ok) { s += nlev; continue; }
        const size_t mark = blob.size();
        RowsPlanLaunch R;
        memset(&R, 0, sizeof(R));
        // ---- column / row tables ----
        for (int k = 0; k < nlev; k++) {
            const LevelDev& D = T.lv[s + 1 + k];
            const LevelDev& S = T.lv[s + k];
            align4();
            R.W[k] = ((D.cols + 3) & ~3) + 256 + 8;
            R.x_off[k] = (unsigned)(blob.size() * 4);
            blob.resize(blob.size() + 3 * (size_t)R.W[k]);
            int* xt = blob.data() + R.x_off[k] / 4;
            for (int i = 0; i < R.W[k]; i++) {
                const int ox = i < D.cols - 1 ? i : D.cols - 1;
                const float sx = (float)ox * D.fx;
                int x1 = (int)floorf(sx);
                if (x1 > S.cols - 1) x1 = S.cols - 1;
                const int x2 = x1 + 1;
                xt[i] = x1; xt[R.W[k] + i] = fbits(efx_s5_w_hi(ox, D.fx, sx, x2)); xt[2 * R.W[k] + i] = fbits(efx_s5_w_lo(ox, D.fx, sx, x1));
                if (x1 >= S.cols - 1) ok = false;             // a clamped +1 neighbour (fx == 1): the tiled kernels
            }
            align4();
            R.y_off[k] = (unsigned)(blob.size() * 4);
            const int Hh = D.rows + 64;
            blob.resize(blob.size() + 4 * (size_t)Hh);
            int* yt = blob.data() + R.y_off[k] / 4;
            for (int i = 0; i < Hh; i++) {
                const int oy = i < D.rows - 1 ? i : D.rows - 1;
                const float sy = (float)oy * D.fy;
                int y1 = (int)floorf(sy);
                if (y1 > S.rows - 1) y1 = S.rows - 1;
                const int y2 = y1 + 1;
                const int y2r = y2 < S.rows - 1 ? y2 : S.rows - 1;
                yt[4 * i] = y1; yt[4 * i + 1] = y2r; yt[4 * i + 2] = fbits(efx_s5_w_hi(oy, D.fy, sy, y2)); yt[4 * i + 3] = fbits(efx_s5_w_lo(oy, D.fy, sy, y1));
                if (y2r != y1 + 1) ok = false;                // clamped bottom (fy == 1): the tiled kernels
            }
        }
        if (!ok) { blob.resize(mark); s += nlev; continue; }
        auto X = [&](int k) { return blob.data() + R.x_off[k] / 4; };
        auto Y = [&](int k) { return blob.data() + R.y_off[k] / 4; };
        auto cols_of = [&](int k) { return T.lv[s + 1 + k].cols; };
        auto rows_of = [&](int k) { return T.lv[s + 1 + k].rows; };
        // ---- strips: the widest `own` (columns of level s + 1 per strip) whose halo fits the 64 lanes at every level ----
        std::vector<int> stv;
        bool found = false;
        for (int own = nlev == 1 ? 256 : 252; own >= 192 && !found; own -= 4) {
            const int nstrips = (cols_of(0) + own - 1) / own;
            stv.assign(4 * RW_STRIP_INT4 * (size_t)nstrips, 0);
            bool good = true;
            // owned group ranges, bottom up: level k owns the groups whose first source column lies in the columns level k - 1 owns
            std::vector<std::vector<int>> glo(nlev, std::vector<int>(nstrips + 1, 0));
            for (int j = 0; j <= nstrips; j++) glo[0][j] = std::min(j * own, (cols_of(0) + 3) & ~3) / 4;      // (level 0 of the launch: groups of `own` / 4)
            for (int k = 1; k < nlev; k++) {
                const int ngroups = (cols_of(k) + 3) / 4;
                int g = 0;
                for (int j = 0; j < nstrips; j++) {
                    glo[k][j] = g;
                    const int own_end = j == nstrips - 1 ? 0x7fffffff : 4 * glo[k - 1][j + 1];      // columns of level k - 1 this strip owns: up to here
                    while (g < ngroups && X(k)[4 * g] < own_end) g++;
                }
                glo[k][nstrips] = ngroups;
                if (g != ngroups) good = false;
            }
            for (int j = 0; j < nstrips && good; j++) {
                // computed groups, top down: the owned ones plus what the next level's computed groups read
                std::vector<int> ncomp(nlev, 0);
                for (int k = nlev - 1; k >= 1 && good; k--) {
                    int hi = glo[k][j + 1];                                                            // end of the owned groups
                    if (k + 1 < nlev && ncomp[k + 1] > 0) {
                        const int lastcol = std::min(4 * (glo[k + 1][j] + ncomp[k + 1]) - 1, cols_of(k + 1) - 1);
                        hi = std::max(hi, (X(k + 1)[lastcol] + 1) / 4 + 1);                            // its +1 neighbour's group, inclusive
                    }
                    ncomp[k] = hi - glo[k][j];
                    if (ncomp[k] > 64 || ncomp[k] < 0) good = false;
                }
                if (!good) break;
                const int bx0 = j * own;
                // level 0 of the launch computes columns bx0 .. bx0 + 255: enough for level 1's computed groups?
                if (nlev > 1 && ncomp[1] > 0) {
                    const int lastcol = std::min(4 * (glo[1][j] + ncomp[1]) - 1, cols_of(1) - 1);
                    if (X(1)[lastcol] + 1 > bx0 + 255 || X(1)[4 * glo[1][j]] < bx0) good = false;
                }
                // windows of every computing lane at every level (resize_windows: both pixel pairs of an output pair within 8 bytes
                // of an aligned dword), and the 256-column reach of a level's LDS row
                const int ax0 = X(0)[bx0] & ~3;
                const int lastx1 = X(0)[std::min(bx0 + 255, R.W[0] - 1)];
                const int nd = ((std::min(lastx1 + 1, A.cols - 1) - ax0) >> 2) + 1;
                if (nd > 128) good = false;
                for (int k = 0; k < nlev && good; k++) {
                    const int org = k == 0 ? ax0 : k == 1 ? bx0 : 4 * glo[k - 1][j];
                    const int g0 = k == 0 ? bx0 / 4 : glo[k][j], nc = k == 0 ? 64 : ncomp[k];
                    for (int l = 0; l < nc && good; l++) {
                        const int* q = X(k) + 4 * (g0 + l);
                        const int oA = (q[0] - org) & ~3, oB = (q[2] - org) & ~3;
                        if (q[0] < org || q[1] < q[0] || q[1] - org - oA + 1 > 7 || q[3] < q[2] || q[3] - org - oB + 1 > 7) good = false;
                        if (q[3] + 1 - org > (k == 0 ? 511 : 255)) good = false;
                    }
                }
                int* q = stv.data() + 4 * RW_STRIP_INT4 * (size_t)j;
                q[0] = ax0; q[1] = nd;
                for (int k = 1; k < nlev; k++) { q[4 * k] = glo[k][j]; q[4 * k + 1] = glo[k][j + 1] - glo[k][j]; q[4 * k + 2] = ncomp[k]; }
            }
            if (good) { found = true; R.own = own; R.nstrips = nstrips; }
        }
        if (!found) { blob.resize(mark); s += nlev; continue; }
        align4();
        R.strip_off = (unsigned)(blob.size() * 4);
        blob.insert(blob.end(), stv.begin(), stv.end());
        // ---- chunks of rows of the top level: as many as give ~wave_target waves; at most 64 source rows (the masks) per chunk ----
        {
            const int top = nlev - 1;
            // (the frames of a batched launch share the plan: the launch has nframes x these waves, so its chunks are longer and
            // recompute fewer halo rows)
            const int want = std::max(1, (wave_target + R.nstrips * nframes - 1) / (R.nstrips * nframes));
            int rc = (rows_of(top) + want - 1) / want;
            float ftot = 1.f;
            for (int k = 0; k < nlev; k++) ftot *= T.lv[s + 1 + k].fy;
            const int rc_max = std::max(1, (int)((64 - RW_D - 2 * nlev) / ftot) - nlev);
            rc = std::max(std::min(8, rc_max), std::min(rc_max, rc));
            R.nchunks = (rows_of(top) + rc - 1) / rc;
            std::vector<int> cv(4 * RW_CHUNK_INT4 * (size_t)R.nchunks, 0);
            for (int c = 0; c < R.nchunks && ok; c++) {
                const bool lastc = c == R.nchunks - 1;
                int first[RW_MAXLEV], last[RW_MAXLEV], send[RW_MAXLEV];       // rows computed: first .. last; stored: first .. send - 1
                first[top] = c * rc; send[top] = std::min(first[top] + rc, rows_of(top)); last[top] = send[top] - 1;
                int nfirst_up = send[top];                                      // first row of the NEXT chunk at the level above
                for (int k = top - 1; k >= 0; k--) {
                    first[k] = Y(k + 1)[4 * first[k + 1]];                      // y1 of the first row above
                    last[k] = Y(k + 1)[4 * last[k + 1] + 1];                    // lower source row of the last row above
                    send[k] = lastc ? rows_of(k) : Y(k + 1)[4 * nfirst_up];     // where the next chunk starts
                    if (lastc) last[k] = rows_of(k) - 1;                        // the rows below the top level's last source row belong to the level all the same
                    nfirst_up = send[k];
                    if (last[k] < send[k] - 1 || last[k] - first[k] + 1 > 64) ok = false;
                }
                if (last[top] - first[top] + 1 > 64) ok = false;
                if (!ok) break;
                const int a_first = Y(0)[4 * first[0]], a_last = Y(0)[4 * last[0] + 1];
                const int na = a_last - a_first + 1, na_pad = (na + RW_D - 1) / RW_D * RW_D;
                if (na_pad > 64) { ok = false; break; }
                // done[k][r - first[k]]: the source row (index in the chunk) that completes row r of level k
                unsigned long long mask[RW_MAXLEV] = { 0, 0, 0, 0 };
                std::vector<int> done_prev, done;
                for (int k = 0; k < nlev; k++) {
                    done.assign(last[k] - first[k] + 1, 0);
                    for (int r = first[k]; r <= last[k]; r++) {
                        const int lower = Y(k)[4 * r + 1];                      // its lower source row, in level k - 1 (source level for k == 0)
                        int d;
                        if (k == 0) d = lower - a_first;
                        else { if (lower < first[k - 1] || lower > last[k - 1]) { ok = false; break; } d = done_prev[lower - first[k - 1]]; }
                        if (d < 0 || d >= na) { ok = false; break; }
                        done[r - first[k]] = d;
                        if ((mask[k] >> d) & 1ull) { ok = false; break; }      // two rows of a level completed by one source row: the kernel makes one row per bit
                        mask[k] |= 1ull << d;
                    }
                    done_prev = done;
                    if (k > 0 && (mask[k] & ~mask[k - 1])) ok = false;          // a level-k row is made in the iteration that makes its lower source row
                }
                int* q = cv.data() + 4 * RW_CHUNK_INT4 * (size_t)c;
                q[0] = a_first; q[1] = a_last; q[2] = na_pad;
                for (int k = 0; k < nlev; k++) { q[4 * (1 + k)] = first[k]; q[4 * (1 + k) + 1] = send[k]; }
                for (int k = 0; k < RW_MAXLEV; k++) {
                    q[4 * (1 + RW_MAXLEV) + 2 * k] = (int)(uint32_t)mask[k];
                    q[4 * (1 + RW_MAXLEV) + 2 * k + 1] = (int)(uint32_t)(mask[k] >> 32);
                }
            }
            align4();
            R.chunk_off = (unsigned)(blob.size() * 4);
            blob.insert(blob.end(), cv.begin(), cv.end());
        }
        if (!ok) { blob.resize(mark); s += nlev; continue; }
        R.nlev = nlev;
        out[s] = R;
        s += nlev;
    }
}

// pyramid geometry, quotas, caps (calcImagePyramid .cpp:136-157, calcNumFeaturesPerLevel :159-174, :252)
int build_geometry(efx_context* c, int rows, int cols, int nframes = 1)
{
    const efx_params& p = c->p;
    if (c->g_rows == rows && c->g_cols == cols && c->g_frames >= nframes && c->g_plan_frames == nframes && memcmp(&c->g_p, &p, sizeof(p)) == 0) return EFX_OK;
    const size_t NF = (size_t)std::max(nframes, c->g_rows == rows && c->g_cols == cols ? c->g_frames : 1);      // buffers never shrink
    LevelTable& T = c->h_table;
    memset(&T, 0, sizeof(T));
    T.nlevels = p.nlevels;
    if (!c->h_hint) {
        // best effort: without the word every launch takes the dense form
        void* hp = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            c->h_hint = static_cast<int*>(hp); c->d_hint = static_cast<int*>(dp);
            *c->h_hint = 0;
        } else {
            if (hp) (void)hipHostFree(hp);
            (void)hipGetLastError();
        }
    }
    T.host_hint = c->d_hint;

    int quota[EFX_MAX_LEVELS];
    {
        const double factor = (double)(1 / p.scale_factor);       // float division widened to double (.cpp:164)
        double nf = p.nfeatures * (1 - factor) / (1 - pow(factor, p.nlevels));
        int sum = 0;
        for (int s = 0; s < p.nlevels - 1; s++) { quota[s] = cv_round_d(nf); sum += quota[s]; nf *= factor; }
        quota[p.nlevels - 1] = p.nfeatures - sum > 0 ? p.nfeatures - sum : 0;
    }
    float scale = 1.f;
    size_t pyr = 0, ncand = 0, ncmax = 0;
    int tiles = 0, trows = 0;
    for (int s = 0; s < p.nlevels; s++) {
        LevelDev& L = T.lv[s];
        if (s > 0) scale *= p.scale_factor;
        const float inv = 1.f / scale;
        L.rows = s == 0 ? rows : cv_round_f(inv * (float)rows);
        L.cols = s == 0 ? cols : cv_round_f(inv * (float)cols);
        if (L.rows < 0) L.rows = 0;
        if (L.cols < 0) L.cols = 0;
        L.scale = scale;
        L.active = s >= p.first_level && L.rows > 0 && L.cols > 0;
        L.quota = quota[s];
        L.cap = cv_round_d(0.1 * (double)((long long)L.rows * L.cols));
        L.tiles_x = (L.cols + EFX_TILE - 1) / EFX_TILE;
        L.tiles_y = (L.rows + EFX_TILE - 1) / EFX_TILE;
        L.tile_base = tiles;
        tiles += L.tiles_x * L.tiles_y;
        if (s > 0) {
            const LevelDev& P = T.lv[s - 1];
            L.pitch = (int)align_up((size_t)L.cols, 256);
            L.img_off = pyr;
            pyr += (size_t)L.pitch * L.rows;
            if (L.cols > 0 && L.rows > 0 && P.cols > 0 && P.rows > 0) {
                L.fx = (float)(1.0 / ((double)L.cols / (double)P.cols));       // spec S5
                L.fy = (float)(1.0 / ((double)L.rows / (double)P.rows));
            }
        }
        // The level's corner and survivor arrays hold EXACTLY cap records: a tile's corners start at the tile's canonical rank in
        // the level, corners of rank >= cap do not exist (spec S2; the reference keeps cvRound(0.1 w h) per level, .cpp:252), and a
        // tile's survivors sit at its corners' places.  Nothing is allocated on the device, so no frame -- whatever its corner
        // density -- can overflow anything.  (Rounds 2-5: arenas sized for a corner density of 1/8 with atomic chunk allocation;
        // a denser frame was void and the context regrew to 2.1 GB at 8K.)
        L.cand_base = ncand;
        L.cmax_base = ncmax;
        L.row_base = trows;
        ncmax += (size_t)L.tiles_x * 4 * L.tiles_y * 4;
        trows += L.tiles_y;
        if (L.active) ncand += (size_t)L.cap + 64;      // (+ 64: a wave's clamped look-ahead loads stay inside the array)
    }
    // counting workgroups of select_kernel (EFX_SEL_WG_TILES consecutive tiles each) that hold tiles of a level
    for (int s = 0; s < p.nlevels; s++) {
        LevelDev& L = T.lv[s];
        const int nt = L.tiles_x * L.tiles_y;
        L.sel_wg0 = L.tile_base / EFX_SEL_WG_TILES;
        L.sel_wgs = nt > 0 ? (L.tile_base + nt - 1) / EFX_SEL_WG_TILES - L.sel_wg0 + 1 : 0;
        L.pack_groups = L.active ? (nt + EFX_PACK_TPW - 1) / EFX_PACK_TPW : 0;      // harris_packed_kernel's workgroups
    }
    T.total_rows = trows;
    T.total_tiles = tiles;
    // upper bound of N: calcNumFeaturesPerLevel rounds every level up, so the quotas can sum to more than nfeatures
    c->n_out_max = 0;
    for (int s = 0; s < p.nlevels; s++) if (T.lv[s].active) c->n_out_max += T.lv[s].quota;
    if (rows > 32767 || cols > 32767) return set_err(c->err, EFX_ERR_UNSUPPORTED, "image larger than 32767 (short2 coordinates)");

    // the level table is followed by one packed word per tile (efx_pack_tile / efx_tile_of)
    HIP_TRY(c->err, c->d_table.reserve(sizeof(LevelTable) + (size_t)(tiles + 1) * sizeof(uint32_t)));
    // everything below exists once per frame of a batched launch, `fs` apart
    c->fs.pyramid = align_up(pyr + 256, 256);
    c->fs.hdr = (size_t)tiles + 1;
    c->fs.cand = ncand + 64;
    c->fs.slots = ((size_t)tiles + 1) * EFX_SLOT_BYTES;
    c->fs.rows = (size_t)trows + 1;
    c->fs.hist = (size_t)p.nlevels * EFX_HIST_BINS;
    c->fs.list = (size_t)p.nlevels * EFX_SEL_LIST_CAP;
    c->fs.cmax = ncmax + 1;
    HIP_TRY(c->err, c->pyramid.reserve(c->fs.pyramid * NF));
    HIP_TRY(c->err, c->hdr.reserve(c->fs.hdr * NF * sizeof(TileHdr)));
    HIP_TRY(c->err, c->cand.reserve(c->fs.cand * NF * sizeof(Corner)));
    HIP_TRY(c->err, c->surv.reserve(c->fs.cand * NF * sizeof(Corner)));
    HIP_TRY(c->err, c->slots.reserve(c->fs.slots * NF));
    HIP_TRY(c->err, c->tcount.reserve(c->fs.hdr * NF * sizeof(uint16_t)));
    HIP_TRY(c->err, c->nsel.reserve(c->fs.hdr * NF * sizeof(uint32_t)));
    HIP_TRY(c->err, c->rowsum.reserve(c->fs.rows * NF * sizeof(RowCtr)));
    {
        const void* before = c->hist.p; const size_t before_bytes = c->hist.bytes;
        HIP_TRY(c->err, c->hist.reserve(c->fs.hist * NF * sizeof(int)));
        if (c->hist.p != before || c->hist.bytes != before_bytes) c->hist_clean = false;
    }
    c->hist_clean = false;          // (the levels' places in the buffer move with the geometry)
    HIP_TRY(c->err, c->sel_list.reserve(c->fs.list * NF * sizeof(unsigned long long)));
    HIP_TRY(c->err, c->cmax.reserve(c->fs.cmax * NF * sizeof(Corner)));
    HIP_TRY(c->err, c->counters.reserve(sizeof(Counters) * NF));
    HIP_TRY(c->err, c->count.reserve(sizeof(int) * EFX_MAX_BATCH));
    if (!c->h_mirror) {
        c->h_mirror = new (std::nothrow) Summary;
        if (!c->h_mirror) return set_err(c->err, EFX_ERR_NOMEM, "out of host memory");
        memset(c->h_mirror, 0, sizeof(Summary));
    }
    // synchronous upload: geometry changes are rare (first frame / size or parameter change)
    {
        std::vector<unsigned char> blob(sizeof(LevelTable) + (size_t)(tiles + 1) * sizeof(uint32_t), 0);
        memcpy(blob.data(), &T, sizeof(LevelTable));
        uint32_t* info = reinterpret_cast<uint32_t*>(blob.data() + sizeof(LevelTable));
        for (int s = 0; s < p.nlevels; s++) {
            const LevelDev& L = T.lv[s];
            for (int ty = 0; ty < L.tiles_y; ty++)
                for (int tx = 0; tx < L.tiles_x; tx++) info[L.tile_base + ty * L.tiles_x + tx] = efx_pack_tile((uint32_t)s, (uint32_t)tx, (uint32_t)ty);
        }
        HIP_TRY(c->err, hipMemcpy(c->d_table.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    }
    // resize plan: per destination level the column / row / tile tables of resize_stream_kernel (spec S5 expressions)
    {
        std::vector<int> blob;
        auto fbits = [](float f) { int i; memcpy(&i, &f, 4); return i; };
        memset(c->rplan_lv, 0, sizeof(c->rplan_lv));
        for (int s = 1; s < p.nlevels; s++) {
            const LevelDev& P = T.lv[s - 1];
            const LevelDev& N = T.lv[s];
            if (N.rows <= 0 || N.cols <= 0 || P.rows <= 0 || P.cols <= 0) continue;
            const int W = N.tiles_x * EFX_TILE, Hh = N.tiles_y * EFX_TILE;
            ResizePlanLevel& R = c->rplan_lv[s];
            R.W = W;
            R.x_off = (unsigned)(blob.size() * 4);
            blob.resize(blob.size() + 3 * (size_t)W);
            int* xt = blob.data() + R.x_off / 4;
            for (int i = 0; i < W; i++) {
                const int ox = i < N.cols - 1 ? i : N.cols - 1;
                const float sx = (float)ox * N.fx;
                int x1 = (int)floorf(sx);
                if (x1 > P.cols - 1) x1 = P.cols - 1;
                const int x2 = x1 + 1;
                xt[i] = x1; xt[W + i] = fbits(efx_s5_w_hi(ox, N.fx, sx, x2)); xt[2 * W + i] = fbits(efx_s5_w_lo(ox, N.fx, sx, x1));
            }
            R.y_off = (unsigned)(blob.size() * 4);
            blob.resize(blob.size() + 4 * (size_t)Hh);
            int* yt = blob.data() + R.y_off / 4;
            for (int i = 0; i < Hh; i++) {
                const int oy = i < N.rows - 1 ? i : N.rows - 1;
                const float sy = (float)oy * N.fy;
                int y1 = (int)floorf(sy);
                if (y1 > P.rows - 1) y1 = P.rows - 1;
                const int y2 = y1 + 1;
                const int y2r = y2 < P.rows - 1 ? y2 : P.rows - 1;
                yt[4 * i] = y1; yt[4 * i + 1] = y2r; yt[4 * i + 2] = fbits(efx_s5_w_hi(oy, N.fy, sy, y2)); yt[4 * i + 3] = fbits(efx_s5_w_lo(oy, N.fy, sy, y1));
            }
            R.t_off = (unsigned)(blob.size() * 4);
            blob.resize(blob.size() + 4 * (size_t)N.tiles_x * N.tiles_y);
            int* tt = blob.data() + R.t_off / 4;
            for (int ty = 0; ty < N.tiles_y; ty++)
                for (int tx = 0; tx < N.tiles_x; tx++) {
                    const int ox0 = tx * EFX_TILE, oy0 = ty * EFX_TILE;
                    const int ox1 = std::min(ox0 + EFX_TILE, N.cols), oy1 = std::min(oy0 + EFX_TILE, N.rows);
                    const int sx0 = std::min((int)floorf((float)ox0 * N.fx), P.cols - 1), sy0 = std::min((int)floorf((float)oy0 * N.fy), P.rows - 1);
                    const int sx1 = std::min(std::min((int)floorf((float)(ox1 - 1) * N.fx), P.cols - 1) + 1, P.cols - 1);
                    const int sy1 = std::min(std::min((int)floorf((float)(oy1 - 1) * N.fy), P.rows - 1) + 1, P.rows - 1);
                    const int ax0 = sx0 & ~3, ndw = ((sx1 - ax0) >> 2) + 1, nrow = sy1 - sy0 + 1;
                    int* q = tt + 4 * ((size_t)ty * N.tiles_x + tx);
                    q[0] = sy0; q[1] = ax0;
                    q[2] = (ndw & 0xff) | ((nrow & 0xff) << 8) | ((sx1 == P.cols - 1 ? 1 : 0) << 16);
                    q[3] = tx | (ty << 16);
                    if (ndw > 32 || nrow > 80) R.W = 0;          // footprint beyond what the streamed kernel stages: no plan
                }
        }
        build_rows_plan(T, p.nlevels, nframes, blob, c->rows_plan);
        if (!blob.empty()) {
            HIP_TRY(c->err, c->rplan.reserve(blob.size() * 4));
            HIP_TRY(c->err, hipMemcpy(c->rplan.p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
        }
    }
    c->g_rows = rows; c->g_cols = cols; c->g_p = p; c->g_frames = (int)NF; c->g_plan_frames = nframes;
    return EFX_OK;
}

// nframes same-sized frames through ONE launch of every kernel (round 6; SURVEY 8b "batched variants"): frame f reads
// d_images[f] and writes d_keypoints[f] / d_desc[f] / d_counts[f]; everything in between lives in the context's buffers, once per
// frame (build_geometry: FrameStride).  nframes == 1 is the plain detectAsync / detectAndComputeAsync call.
int detect_frames(efx_context* c, int nframes, const uint8_t* const* d_images, int rows, int cols, size_t pitch,
                  void* const* d_keypoints, size_t kps_pitch, uint8_t* const* d_descs, size_t desc_pitch,
                  int capacity, int* const* d_counts, hipStream_t stream, const uint8_t* d_mask = nullptr, size_t mask_pitch = 0)
{
    if (nframes < 1 || nframes > EFX_MAX_BATCH) return set_err(c->err, EFX_ERR_BAD_ARG, "a launch takes 1 .. %d frames", EFX_MAX_BATCH);
    if (d_mask && mask_pitch < (size_t)cols) return set_err(c->err, EFX_ERR_BAD_ARG, "mask must be an 8-bit image of the frame size");
    if (d_mask && nframes > 1) return set_err(c->err, EFX_ERR_BAD_ARG, "a mask belongs to one frame");
    // CV_Assert(_image.type() == CV_8U) etc. (.cpp:228-229)
    if (!d_images || rows <= 0 || cols <= 0 || pitch < (size_t)cols) return set_err(c->err, EFX_ERR_BAD_ARG, "bad image arguments");
    if (capacity < 0) return set_err(c->err, EFX_ERR_BAD_ARG, "capacity must be >= 0");
    const bool want_kps = d_keypoints && d_keypoints[0];
    const bool want_desc = d_descs && d_descs[0];
    for (int f = 0; f < nframes; f++) {
        if (!d_images[f]) return set_err(c->err, EFX_ERR_BAD_ARG, "bad image arguments");
        if (want_kps != (d_keypoints && d_keypoints[f] != nullptr) || want_desc != (d_descs && d_descs[f] != nullptr))
            return set_err(c->err, EFX_ERR_BAD_ARG, "either every frame of a batch has a keypoint / descriptor matrix or none");
    }
    if (want_kps && (kps_pitch < (size_t)capacity * 4 || (kps_pitch & 3))) return set_err(c->err, EFX_ERR_BAD_ARG, "kps_pitch too small or unaligned");
    if (want_desc && desc_pitch < (size_t)efx_descriptor_size(c)) return set_err(c->err, EFX_ERR_BAD_ARG, "desc_pitch smaller than the descriptor");
    int rc = validate_params(c->p, c->err);
    if (rc) return rc;
    QuiesceScope quiesce(c, stream);
    rc = build_geometry(c, rows, cols, nframes);
    if (rc) return rc;
    const size_t NF = (size_t)nframes;
    const int cap_alloc = capacity > 0 ? capacity : 1;
    HIP_TRY(c->err, c->kp4.reserve((size_t)cap_alloc * NF * sizeof(float4)));
    HIP_TRY(c->err, c->kp_level.reserve((size_t)cap_alloc * NF * sizeof(int)));

    DetectLaunch a;
    memset(&a, 0, sizeof(a));
    a.nframes = nframes;
    a.fs = c->fs;
    a.fs.kp = (size_t)cap_alloc;
    for (int f = 0; f < nframes; f++) {
        a.in.img0[f] = d_images[f];
        a.out.kps[f] = want_kps ? static_cast<uint8_t*>(d_keypoints[f]) : nullptr;
        a.out.count[f] = (d_counts && d_counts[f]) ? d_counts[f] : static_cast<int*>(c->count.p) + f;
    }
    a.img0 = d_images[0]; a.pitch0 = (int)pitch;
    a.pyramid = static_cast<uint8_t*>(c->pyramid.p);
    a.d_table = static_cast<const LevelTable*>(c->d_table.p);
    a.h_table = &c->h_table;
    a.hdr = static_cast<TileHdr*>(c->hdr.p);
    a.rplan = static_cast<const unsigned char*>(c->rplan.p); a.rplan_lv = c->rplan_lv; a.rows_plan = c->rows_plan;
    a.cand = static_cast<Corner*>(c->cand.p);
    a.surv = static_cast<Corner*>(c->surv.p);
    a.slots = static_cast<unsigned char*>(c->slots.p);
    a.tcount = static_cast<uint16_t*>(c->tcount.p);
    a.nsel = static_cast<uint32_t*>(c->nsel.p);
    a.rows = static_cast<RowCtr*>(c->rowsum.p);
    a.hist = static_cast<int*>(c->hist.p);
    a.sel_list = static_cast<unsigned long long*>(c->sel_list.p);
    if (!c->hist_clean) {
        // nms_kernel adds to the key histograms, select_kernel's leaders clear them: they are zero between frames, and only a new
        // buffer (or a call that failed half-way) needs clearing
        HIP_TRY(c->err, hipMemsetAsync(c->hist.p, 0, c->hist.bytes, stream));
    }
    c->hist_clean = false;          // ... until this call's launches are all enqueued
    a.cmax = static_cast<Corner*>(c->cmax.p);
    a.counters = static_cast<Counters*>(c->counters.p);
    a.threshold = c->p.fast_threshold;
    a.nonmax_radius = c->p.nonmax_radius;
    a.first_level = c->p.first_level;
    a.knobs = c->knobs;
    a.mask = d_mask; a.mask_pitch = (int)mask_pitch;
    {
        // Sparse or dense form of harris_kernel (the same results either way): by the FAST corners per level-0 tile of the frame this
        // context processed last -- read from host memory without a synchronisation, possibly a frame or two stale -- once the frame
        // is large enough for one wave per tile; EFX_PACK pins it
        const LevelDev& L0 = c->h_table.lv[0];
        const long long t0 = (long long)L0.tiles_x * L0.tiles_y;
        const int hint = c->h_hint ? *(volatile int*)c->h_hint : 0;
        a.pack_harris = c->knobs.pack >= 0 ? c->knobs.pack
                      : (hint > 0 && (long long)c->h_table.total_tiles * nframes > EFX_PACK_MIN_TILES && (long long)(hint - 1) <= EFX_PACK_AVG * t0) ? 1 : 0;
    }
    a.d_keypoints = want_kps ? d_keypoints[0] : nullptr; a.kps_pitch = kps_pitch; a.capacity = capacity;
    a.d_count = a.out.count[0];
    a.kp4 = static_cast<float4*>(c->kp4.p);
    a.kp_level = static_cast<int*>(c->kp_level.p);
    if (!c->prof_start.empty() && (c->prof_calls++ % c->prof_stride) == 0) {
        a.prof.start = c->prof_start.data(); a.prof.stop = c->prof_stop.data(); a.prof.code = c->prof_level.data();
        a.prof.count = &c->prof_count; a.prof.capacity = (int)c->prof_start.size(); a.prof.skip = c->prof_skip;
    }
    // BAD behind detectAndCompute: angle_kernel writes the describer's per-keypoint records (no bad_affine_kernel launch)
    const int n_desc = capacity < c->n_out_max ? capacity : c->n_out_max;     // the bound angle_kernel uses: sum of the active quotas
    bool affine_ready = false, level_blurred = false;
    if (want_desc && capacity > 0 && c->desc.kind == 0 && n_desc > 0) {
        HIP_TRY(c->err, c->desc.responses.reserve((size_t)cap_alloc * NF * sizeof(Affine)));      // frame f's records at f * cap_alloc (FrameStride::kp)
        const int S = efx_bad_smax_for((float)EFX_PATCH_SIZE, c->desc.scale, c->desc.reach);
        a.bad_affine = c->desc.responses.p; a.bad_scale = c->desc.scale; a.bad_reach = c->desc.reach;
        a.bad_smax = S; a.bad_sfixed = S == 48 ? 48 : 0;
        affine_ready = true;
        // the levels blurred as images, a wave per keypoint on them (bad_kernel.hip: blur_levels_kernel + bad_raw_kernel) -- under
        // bad_raw_kernel's conditions: the 48-pixel window, no box edge above 16 (integral modulo 2^16)
        if (S == 48 && c->desc.ubox_max_side <= 16 && !c->desc.no_raw && !c->knobs.no_level_blur) {
            const LevelTable& H = c->h_table;
            size_t lv_bytes = 0;
            for (int l = 1; l < H.nlevels; l++)
                if (H.lv[l].rows > 0) lv_bytes = std::max(lv_bytes, (size_t)H.lv[l].img_off + (size_t)H.lv[l].pitch * H.lv[l].rows);
            const int p0 = (int)align_up((size_t)cols, 256);
            const size_t l0_bytes = H.lv[0].active ? (size_t)p0 * rows : 0;
            a.fs.blurred = align_up(l0_bytes + lv_bytes + 256, 256);
            HIP_TRY(c->err, c->blurred.reserve(a.fs.blurred * NF));
            a.blurred = static_cast<uint8_t*>(c->blurred.p); a.blur0_pitch = p0; a.blur_levels_off = l0_bytes;
            level_blurred = true;
            a.blur_fork = c->knobs.blur_fork;
            if (a.blur_fork < 0) {
                // per call: the side stream only for an idle stream; never inside a stream capture (a query would invalidate it)
                // and not in a profiled call (its event pairs time the kernels of ONE stream)
                a.blur_fork = 0;
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                // ... and only for frames whose blur is long enough to be worth two event edges (~10 us): measured with the
                // reference's call-then-wait protocol, 8K (103 M level pixels) 0.430 -> 0.408 ms, 4K (26 M) 0.195 -> 0.199, FHD 0.115 -> 0.122
                size_t level_px = 0;
                for (int l = 0; l < H.nlevels; l++) level_px += (size_t)H.lv[l].rows * H.lv[l].cols;
                // (EFX_BLUR_FORK_MIN_PX: the gate, for tests that exercise the per-call decision on small frames; read per
                // context with the other knobs -- ADVICE r5: a process-wide static made the path depend on test collection order)
                const size_t min_px = c->knobs.blur_fork_min_px > 0 ? (size_t)c->knobs.blur_fork_min_px : (size_t)50 * 1000 * 1000;
                if (!a.prof.start && level_px >= min_px && nframes == 1) {
                    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
                    else if (cs == hipStreamCaptureStatusNone) {
                        const hipError_t q = hipStreamQuery(stream);
                        // ... and only for a caller that found it idle twice in a row: a throughput loop whose queue runs dry
                        // now and then stays inline
                        if (q == hipSuccess) { if (++c->idle_streak >= 2) a.blur_fork = 3; }
                        else { c->idle_streak = 0; if (q != hipErrorNotReady) (void)hipGetLastError(); }
                    }
                }
            }
            // the side stream and its two events exist from the context's first describing call on (creating them costs ~0.2 ms: not
            // inside the first call that forks); it joins the call's stream before the describer runs, so whoever waits for the
            // call's stream has waited for it too (release waits, the caller's own synchronisation)
            if (!c->side && (c->knobs.blur_fork != 0)) {
                HIP_TRY(c->err, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
                HIP_TRY(c->err, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                HIP_TRY(c->err, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            if (a.blur_fork >= 1 && a.blur_fork <= 3) { a.side = c->side; a.ev_fork = c->ev_fork; a.ev_join = c->ev_join; }
        }
    }
    hipError_t e = efx_launch_detect(a, stream);
    if (e != hipSuccess) return set_err(c->err, EFX_ERR_HIP, "detect launch failed: %s", hipGetErrorString(e));
    c->hist_clean = true;
    c->has_frame = true; c->last_img0 = d_images[0]; c->last_pitch0 = (int)pitch; c->last_frames = nframes;
#ifdef EFX_DEBUG_BUILD
    c->last_launch = a; c->last_launch.prof = ProfRec{};
#endif

    if (want_desc && capacity > 0) {
        // blur + describe per level (.cpp:302-307), here one launch over the keypoints of all levels
        DescribeLaunch dl;
        memset(&dl, 0, sizeof(dl));
        dl.img0 = d_images[0]; dl.pitch0 = (int)pitch; dl.rows0 = rows; dl.cols0 = cols;
        dl.pyramid = a.pyramid; dl.d_table = a.d_table;
        dl.kp4 = a.kp4; dl.kp_level = a.kp_level; dl.d_count = a.d_count;
        dl.n = n_desc;
        dl.affine_ready = affine_ready ? 1 : 0;
        dl.level_blurred = level_blurred ? 1 : 0;
        dl.blur = 1;
        dl.max_size = (float)EFX_PATCH_SIZE;
        dl.uniform_size = 1;
        dl.desc = d_descs[0]; dl.desc_pitch = desc_pitch;
        dl.prof = a.prof;
        if (nframes > 1 && (level_blurred || c->desc.kind == 1)) {
            // every frame's keypoints in one launch of bad_raw_kernel (the records point at each frame's blurred levels), or in one
            // launch of each of the three HashSIFT kernels
            dl.nframes = nframes; dl.aff_stride = (size_t)cap_alloc;
            dl.kp_stride = (size_t)cap_alloc; dl.pyr_stride = a.fs.pyramid; dl.imgs = a.in;
            for (int f = 0; f < nframes; f++) { dl.counts.count[f] = a.out.count[f]; dl.descs.desc[f] = d_descs[f]; }
            rc = describer_run(c->desc, c->err, dl, nullptr, nullptr, stream);
            if (rc) return rc;
        } else {
            // (BAD outside bad_raw_kernel's conditions behind a batch: one describe per frame on the frame's buffers)
            for (int f = 0; f < nframes; f++) {
                DescribeLaunch df = dl;
                df.img0 = d_images[f]; df.pyramid = a.pyramid + f * a.fs.pyramid;
                df.kp4 = a.kp4 + f * a.fs.kp; df.kp_level = a.kp_level + f * a.fs.kp; df.d_count = a.out.count[f];
                df.desc = d_descs[f];
                df.frame_affine_off = (size_t)f * cap_alloc;
                rc = describer_run(c->desc, c->err, df, nullptr, nullptr, stream);
                if (rc) return rc;
            }
        }
    }
    return EFX_OK;
}

int detect_common(efx_context* c, const uint8_t* d_image, int rows, int cols, size_t pitch,
                  void* d_keypoints, size_t kps_pitch, uint8_t* d_desc, size_t desc_pitch,
                  int capacity, int* d_count, hipStream_t stream, const uint8_t* d_mask = nullptr, size_t mask_pitch = 0)
{
    return detect_frames(c, 1, &d_image, rows, cols, pitch, &d_keypoints, kps_pitch, &d_desc, desc_pitch, capacity, &d_count, stream, d_mask, mask_pitch);
}

// detectAndCompute(useProvidedKeypoints = true), spec S13: pyramid only, then blur + describe on the keypoints' levels
int compute_provided(efx_context* c, const uint8_t* d_image, int rows, int cols, size_t pitch,
                     const void* d_keypoints, size_t kps_pitch, int n, uint8_t* d_desc, size_t desc_pitch, hipStream_t stream)
{
    if (!d_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols) return set_err(c->err, EFX_ERR_BAD_ARG, "bad image arguments");
    if (n < 0) return set_err(c->err, EFX_ERR_BAD_ARG, "n must be >= 0");
    if (n == 0) return EFX_OK;
    if (!d_keypoints || kps_pitch < (size_t)n * 4 || (kps_pitch & 3)) return set_err(c->err, EFX_ERR_BAD_ARG, "bad keypoint matrix");
    if (!d_desc || desc_pitch < (size_t)efx_descriptor_size(c)) return set_err(c->err, EFX_ERR_BAD_ARG, "bad descriptor buffer");
    int rc = validate_params(c->p, c->err);
    if (rc) return rc;
    QuiesceScope quiesce(c, stream);
    rc = build_geometry(c, rows, cols);
    if (rc) return rc;
    HIP_TRY(c->err, c->kp4.reserve((size_t)n * sizeof(float4)));
    HIP_TRY(c->err, c->kp_level.reserve((size_t)n * sizeof(int)));
    DetectLaunch a;
    memset(&a, 0, sizeof(a));
    a.img0 = d_image; a.pitch0 = (int)pitch;
    a.pyramid = static_cast<uint8_t*>(c->pyramid.p);
    a.d_table = static_cast<const LevelTable*>(c->d_table.p);
    a.h_table = &c->h_table;
    a.counters = static_cast<Counters*>(c->counters.p);
    a.rows = static_cast<RowCtr*>(c->rowsum.p);
    a.knobs = c->knobs;
    a.pyramid_only = 1;
    hipError_t e = efx_launch_detect(a, stream);
    if (e == hipSuccess)
        e = efx_launch_provided_keypoints(a.d_table, d_keypoints, kps_pitch, n, static_cast<float4*>(c->kp4.p), static_cast<int*>(c->kp_level.p), stream);
    if (e != hipSuccess) return set_err(c->err, EFX_ERR_HIP, "provided-keypoints launch failed: %s", hipGetErrorString(e));
    c->has_frame = true; c->last_img0 = d_image; c->last_pitch0 = (int)pitch;
    DescribeLaunch dl;
    memset(&dl, 0, sizeof(dl));
    dl.img0 = d_image; dl.pitch0 = (int)pitch; dl.rows0 = rows; dl.cols0 = cols;
    dl.pyramid = a.pyramid; dl.d_table = a.d_table;
    dl.kp4 = static_cast<const float4*>(c->kp4.p); dl.kp_level = static_cast<const int*>(c->kp_level.p); dl.d_count = nullptr;
    dl.n = n; dl.blur = 1; dl.max_size = (float)EFX_PATCH_SIZE; dl.uniform_size = 1;
    dl.desc = d_desc; dl.desc_pitch = desc_pitch;
    rc = describer_run(c->desc, c->err, dl, nullptr, nullptr, stream);
    if (rc) return rc;
    e = efx_launch_zero_invalid_descriptors(a.d_table, d_keypoints, kps_pitch, n, d_desc, desc_pitch, efx_descriptor_size(c), stream);
    if (e != hipSuccess) return set_err(c->err, EFX_ERR_HIP, "launch failed: %s", hipGetErrorString(e));
    return EFX_OK;
}

int describe_single(Describer& d, std::string& err, const uint8_t* d_image, int rows, int cols, size_t pitch,
                    const float4* kp4, int n, float max_size, uint8_t* d_desc, size_t desc_pitch,
                    float* dbg_resp, float* dbg_T, hipStream_t stream, int uniform_size = 0)
{
    if (!d_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols) return set_err(err, EFX_ERR_BAD_ARG, "bad image arguments");
    if (n < 0) return set_err(err, EFX_ERR_BAD_ARG, "n must be >= 0");
    if (n == 0) return EFX_OK;           // empty keypoints: nothing to do (cuda_bad.cpp:51-56)
    if (!kp4) return set_err(err, EFX_ERR_BAD_ARG, "null keypoints");
    if (d_desc && desc_pitch < (size_t)(d.nbits / 8)) return set_err(err, EFX_ERR_BAD_ARG, "desc_pitch smaller than the descriptor");
    DescribeLaunch dl;
    memset(&dl, 0, sizeof(dl));
    dl.img0 = d_image; dl.pitch0 = (int)pitch; dl.rows0 = rows; dl.cols0 = cols;
    dl.kp4 = kp4; dl.n = n; dl.blur = 0; dl.max_size = max_size; dl.uniform_size = uniform_size;
    dl.desc = d_desc; dl.desc_pitch = desc_pitch;
    return describer_run(d, err, dl, dbg_resp, dbg_T, stream);
}

int describe_5xn(Describer& d, std::string& err, const uint8_t* d_image, int rows, int cols, size_t pitch,
                 const void* d_keypoints, size_t kps_pitch, int n, uint8_t* d_desc, size_t desc_pitch, hipStream_t stream)
{
    if (n < 0) return set_err(err, EFX_ERR_BAD_ARG, "n must be >= 0");
    if (n == 0) return EFX_OK;
    if (!d_keypoints || kps_pitch < (size_t)n * 4) return set_err(err, EFX_ERR_BAD_ARG, "bad keypoint matrix");   // CV_Assert(rows == 5), .cpp:111
    // getKeypointsMat -> convertKeypointsKernel (.cpp:102-115, .cu:250-263): the describers' record kernels read the 5 x n
    // matrix themselves (size forced to 31), so the conversion is not a launch of its own
    if (!d_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols) return set_err(err, EFX_ERR_BAD_ARG, "bad image arguments");
    if (d_desc && desc_pitch < (size_t)(d.nbits / 8)) return set_err(err, EFX_ERR_BAD_ARG, "desc_pitch smaller than the descriptor");
    DescribeLaunch dl;
    memset(&dl, 0, sizeof(dl));
    dl.img0 = d_image; dl.pitch0 = (int)pitch; dl.rows0 = rows; dl.cols0 = cols;
    dl.kps5 = static_cast<const uint8_t*>(d_keypoints); dl.kps5_pitch = kps_pitch;
    dl.n = n; dl.blur = 0; dl.max_size = (float)EFX_PATCH_SIZE; dl.uniform_size = 1;
    dl.desc = d_desc; dl.desc_pitch = desc_pitch;
    return describer_run(d, err, dl, nullptr, nullptr, stream);
}

int describe_host(Describer& d, std::string& err, const uint8_t* h_image, int rows, int cols, size_t pitch,
                  const efx_keypoint* kps, int n, uint8_t* h_desc, size_t desc_pitch)
{
    if (!h_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols) return set_err(err, EFX_ERR_BAD_ARG, "bad image arguments");
    if (n < 0) return set_err(err, EFX_ERR_BAD_ARG, "n must be >= 0");
    if (n == 0) return EFX_OK;
    if (!kps || !h_desc) return set_err(err, EFX_ERR_BAD_ARG, "null keypoints / descriptors");
    const int nbytes = d.nbits / 8;
    if (desc_pitch < (size_t)nbytes) return set_err(err, EFX_ERR_BAD_ARG, "desc_pitch smaller than the descriptor");
    const size_t ipitch = align_up((size_t)cols, 256);
    HIP_TRY(err, d.img.reserve(ipitch * rows));
    HIP_TRY(err, d.kp4.reserve((size_t)n * sizeof(float4)));
    HIP_TRY(err, d.desc.reserve((size_t)n * nbytes));
    // getKeypointsMat host branch, cuda_efficient_features.cpp:116-128: {pt.x, pt.y, size, angle}
    std::vector<float4> h((size_t)n);
    float max_size = 0.f;
    for (int i = 0; i < n; i++) {
        h[i] = make_float4(kps[i].x, kps[i].y, kps[i].size, kps[i].angle);
        max_size = fmaxf(max_size, fabsf(kps[i].size));
    }
    HIP_TRY(err, hipMemcpy2D(d.img.p, ipitch, h_image, pitch, cols, rows, hipMemcpyHostToDevice));
    HIP_TRY(err, hipMemcpy(d.kp4.p, h.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice));
    int rc = describe_single(d, err, static_cast<const uint8_t*>(d.img.p), rows, cols, ipitch, static_cast<const float4*>(d.kp4.p), n,
                             max_size, static_cast<uint8_t*>(d.desc.p), nbytes, nullptr, nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(err, hipMemcpy2D(h_desc, desc_pitch, d.desc.p, nbytes, nbytes, n, hipMemcpyDeviceToHost));
    return EFX_OK;
}

int ctx_rebuild_describer(efx_context* c)
{
    // createDescriber, cuda_efficient_features.cpp:48-69: scale 1 for every type
    const int t = c->p.descriptor_type;
    const int kind = (t == EFX_BAD_256 || t == EFX_BAD_512) ? 0 : 1;
    const int nbits = (t == EFX_BAD_256 || t == EFX_HASH_SIFT_256) ? 256 : 512;
    int rc = describer_init(c->desc, kind, nbits, 1.f);
    if (rc) c->err = c->desc.err;
    return rc;
}

} // namespace

extern "C" {

int efx_version(void) { return EFX_VERSION; }

void efx_default_params(efx_params* p)
{
    if (!p) return;
    p->nfeatures = 5000; p->scale_factor = 1.2f; p->nlevels = 8; p->first_level = 0;
    p->fast_threshold = 20; p->nonmax_radius = 15; p->descriptor_type = EFX_HASH_SIFT_256;
}

int efx_create(const efx_params* p, efx_context** out)
{
    if (!out) return set_err(g_create_error, EFX_ERR_BAD_ARG, "null output handle");
    *out = nullptr;
    efx_params q;
    if (p) q = *p; else efx_default_params(&q);
    int rc = validate_params(q, g_create_error);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(g_create_error, EFX_ERR_NO_DEVICE, "no HIP device: the efficient-features path has no CPU fallback");
    efx_context* c = new (std::nothrow) efx_context;
    if (!c) return set_err(g_create_error, EFX_ERR_NOMEM, "out of host memory");
    c->p = q;
    memset(&c->g_p, 0, sizeof(c->g_p));
    rc = ctx_rebuild_describer(c);
    if (rc) { g_create_error = c->err; delete c; return rc; }
    *out = c;
    return EFX_OK;
}

int efx_destroy(efx_context* ctx) { delete ctx; return EFX_OK; }

const char* efx_last_error(const efx_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

#define EFX_SETTER(name, field, type)                                                        \
    int efx_set_##name(efx_context* ctx, type v)                                             \
    {                                                                                        \
        if (!ctx) return EFX_ERR_BAD_ARG;                                                    \
        efx_params q = ctx->p; q.field = v;                                                  \
        int rc = validate_params(q, ctx->err);                                               \
        if (rc) return rc;                                                                   \
        ctx->p = q; return EFX_OK;                                                           \
    }                                                                                        \
    type efx_get_##name(const efx_context* ctx) { return ctx ? ctx->p.field : (type)0; }

EFX_SETTER(max_features, nfeatures, int)
EFX_SETTER(scale_factor, scale_factor, float)
EFX_SETTER(nlevels, nlevels, int)
EFX_SETTER(first_level, first_level, int)
EFX_SETTER(fast_threshold, fast_threshold, int)
EFX_SETTER(nonmax_radius, nonmax_radius, int)

int efx_set_descriptor_type(efx_context* ctx, int v)
{
    // setDescriptorType rebuilds the describer (cuda_efficient_features.cpp:373-377)
    if (!ctx) return EFX_ERR_BAD_ARG;
    efx_params q = ctx->p; q.descriptor_type = v;
    int rc = validate_params(q, ctx->err);
    if (rc) return rc;
    if (v == ctx->p.descriptor_type) return EFX_OK;
    ctx->p = q;
    rc = ctx_rebuild_describer(ctx);
    // the blurred copies of the levels (+103 MB at 8K) serve BAD describers only: a context switched to HashSIFT gives them back
    // (ADVICE r4); its streams are waited for first, like any regrow (Quiesce)
    if (rc == EFX_OK && ctx->desc.kind != 0 && ctx->blurred.p) {
        Quiesce q2 = { &efx_context::quiesce_cb, ctx, false };
        Quiesce* prev = tl_quiesce;
        tl_quiesce = &q2;
        ctx->blurred.release();
        tl_quiesce = prev;
    }
    return rc;
}
int efx_get_descriptor_type(const efx_context* ctx) { return ctx ? ctx->p.descriptor_type : -1; }

int efx_descriptor_size(const efx_context* ctx) { return ctx ? ctx->desc.nbits / 8 : 0; }
int efx_descriptor_dtype(const efx_context*) { return 0; /* CV_8U */ }
int efx_default_norm(const efx_context*) { return 6; /* cv::NORM_HAMMING */ }

int efx_detect_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                     void* d_keypoints, size_t kps_pitch, int capacity, int* d_count, void* stream)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    return detect_common(ctx, d_image, rows, cols, pitch, d_keypoints, kps_pitch, nullptr, 0, capacity, d_count, (hipStream_t)stream);
}

int efx_detect_and_compute_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                 void* d_keypoints, size_t kps_pitch, uint8_t* d_descriptors, size_t desc_pitch,
                                 int capacity, int* d_count, void* stream)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    return detect_common(ctx, d_image, rows, cols, pitch, d_keypoints, kps_pitch, d_descriptors, desc_pitch, capacity, d_count, (hipStream_t)stream);
}

int efx_detect_and_compute_masked_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                        const uint8_t* d_mask, size_t mask_pitch, void* d_keypoints, size_t kps_pitch,
                                        uint8_t* d_descriptors, size_t desc_pitch, int capacity, int* d_count, void* stream)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    return detect_common(ctx, d_image, rows, cols, pitch, d_keypoints, kps_pitch, d_descriptors, desc_pitch, capacity, d_count,
                         (hipStream_t)stream, d_mask, mask_pitch);
}

int efx_compute_provided_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                               const void* d_keypoints, size_t kps_pitch, int n, uint8_t* d_descriptors, size_t desc_pitch, void* stream)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    return compute_provided(ctx, d_image, rows, cols, pitch, d_keypoints, kps_pitch, n, d_descriptors, desc_pitch, (hipStream_t)stream);
}

int efx_detect_and_compute_batch_async(efx_context* const* ctxs, void* const* streams, int nctx,
                                       const uint8_t* const* d_images, int nframes, int rows, int cols, size_t pitch,
                                       void* const* d_keypoints, size_t kps_pitch,
                                       uint8_t* const* d_descriptors, size_t desc_pitch, int capacity, int* const* d_counts)
{
    if (!ctxs || nctx <= 0 || nframes < 0 || !d_images || !d_keypoints || !d_counts) return EFX_ERR_BAD_ARG;
    for (int j = 0; j < nctx && j < nframes; j++) if (!ctxs[j]) return EFX_ERR_BAD_ARG;
    // Context j owns the frames j, j + nctx, j + 2 nctx, ...: they go through ONE launch of every kernel, EFX_MAX_BATCH at a time
    // (frame = blockIdx.y; detect_frames).  EFX_NO_BATCH=1: one single-frame call per frame, in frame order (the round-5 form).
    if (ctxs[0]->knobs.no_batch) {
        for (int i = 0; i < nframes; i++) {
            const int rc = detect_common(ctxs[i % nctx], d_images[i], rows, cols, pitch, d_keypoints[i], kps_pitch,
                                         d_descriptors ? d_descriptors[i] : nullptr, desc_pitch, capacity, d_counts[i],
                                         streams ? (hipStream_t)streams[i % nctx] : nullptr);
            if (rc) return rc;
        }
        return EFX_OK;
    }
    // round-robin over the contexts, so that every stream has work early
    const int per_ctx_max = (nframes + nctx - 1) / nctx;
    for (int k0 = 0; k0 < per_ctx_max; k0 += EFX_MAX_BATCH) {
        for (int j = 0; j < nctx; j++) {
            const uint8_t* img[EFX_MAX_BATCH]; void* kps[EFX_MAX_BATCH]; uint8_t* desc[EFX_MAX_BATCH]; int* cnt[EFX_MAX_BATCH];
            int nb = 0;
            for (int k = k0; k < k0 + EFX_MAX_BATCH; k++) {
                const int i = j + k * nctx;
                if (i >= nframes) break;
                img[nb] = d_images[i]; kps[nb] = d_keypoints[i]; desc[nb] = d_descriptors ? d_descriptors[i] : nullptr; cnt[nb] = d_counts[i];
                nb++;
            }
            if (nb == 0) continue;
            const int rc = detect_frames(ctxs[j], nb, img, rows, cols, pitch, kps, kps_pitch, desc, desc_pitch, capacity, cnt,
                                         streams ? (hipStream_t)streams[j] : nullptr);
            if (rc) return rc;
        }
    }
    return EFX_OK;
}

int efx_compute_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                      const void* d_keypoints, size_t kps_pitch, int n, uint8_t* d_descriptors, size_t desc_pitch, void* stream)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    QuiesceScope quiesce(ctx, (hipStream_t)stream);
    return describe_5xn(ctx->desc, ctx->err, d_image, rows, cols, pitch, d_keypoints, kps_pitch, n, d_descriptors, desc_pitch, (hipStream_t)stream);
}

int efx_compute_kp4_async(efx_context* ctx, const uint8_t* d_image, int rows, int cols, size_t pitch,
                          const float* d_kp4, int n, float max_size, uint8_t* d_descriptors, size_t desc_pitch, void* stream)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    QuiesceScope quiesce(ctx, (hipStream_t)stream);
    return describe_single(ctx->desc, ctx->err, d_image, rows, cols, pitch, reinterpret_cast<const float4*>(d_kp4), n, max_size,
                           d_descriptors, desc_pitch, nullptr, nullptr, (hipStream_t)stream);
}

// The last frame's summary (N, per-level counts) lives in device memory (select_kernel writes it).  The host fetches it ON
// DEMAND with a blocking copy after the caller has synchronised the frame's stream.  (Two cheaper transports were tried
// and dropped: an asynchronous 400-byte copy command behind every frame, and select_kernel writing a pinned, mapped host
// buffer directly.  Under 16 processes sharing the GPU, one frame in ~20 000-40 000 read zeros through either -- the whole
// struct with the copy command, one level's row with the direct writes -- while the device data was correct every time.)
static int fetch_summary(const efx_context* ctx)
{
    if (!ctx->h_mirror || !ctx->counters.p) return EFX_ERR_BAD_ARG;
    const Counters* dc = static_cast<const Counters*>(ctx->counters.p) + (ctx->last_frames - 1);     // the last frame of a batched launch
    return hipMemcpy(ctx->h_mirror, &dc->sum, sizeof(Summary), hipMemcpyDeviceToHost) == hipSuccess ? EFX_OK : EFX_ERR_HIP;
}

// A void frame (N = 0): arena contents failed their range checks on the device ("lost stores" under heavy
// oversubscription).  Since round 6 no frame CONTENT can cause it: the arenas hold the reference's own 10 % cap and nothing is
// allocated on the device.
static int check_overflow(const efx_context* ctx)
{
    if (!ctx->h_mirror->overflow) return EFX_OK;
    efx_context* c = const_cast<efx_context*>(ctx);
    c->overflow_events++;
    return set_err(c->err, EFX_ERR_OVERFLOW, "the frame is void: records in the context's scratch arenas failed their range checks (not a property of "
                   "the frame: see DESIGN.md section 7); repeat the call");
}

int efx_last_count(const efx_context* ctx, int* n)
{
    if (!ctx || !n || !ctx->h_mirror || !ctx->has_frame) return EFX_ERR_BAD_ARG;
    const int rc = fetch_summary(ctx);
    if (rc) return rc;
    *n = ctx->h_mirror->n_out;
    return check_overflow(ctx);
}

int efx_overflow_events(const efx_context* ctx) { return ctx ? ctx->overflow_events : 0; }
int efx_tracked_streams(const efx_context* ctx) { return ctx ? (int)ctx->streams.size() : 0; }

size_t efx_trim_memory(void) { (void)hipDeviceSynchronize(); return block_cache().trim(); }
size_t efx_cached_bytes(void) { return block_cache().cached(); }

size_t efx_device_bytes(const efx_context* ctx)
{
    if (!ctx) return 0;
    const DevBuf* b[] = { &ctx->blurred, &ctx->rplan, &ctx->d_table, &ctx->pyramid, &ctx->hdr, &ctx->cand, &ctx->cmax, &ctx->surv, &ctx->counters, &ctx->kp4,
                          &ctx->kp_level, &ctx->img, &ctx->kps, &ctx->descout, &ctx->count, &ctx->maskbuf, &ctx->desc.params,
                          &ctx->desc.responses, &ctx->desc.kp4, &ctx->desc.img, &ctx->desc.desc };
    size_t t = 0;
    for (const DevBuf* d : b) t += d->bytes;
    return t;
}

int efx_last_level_stats(const efx_context* ctx, efx_level_stats* stats, int max_levels, int* nlevels)
{
    if (!ctx || !stats || !ctx->h_mirror || !ctx->has_frame) return EFX_ERR_BAD_ARG;
    const int frc = fetch_summary(ctx);
    if (frc) return frc;
    const int nl = ctx->h_table.nlevels < max_levels ? ctx->h_table.nlevels : max_levels;
    for (int i = 0; i < nl; i++) {
        stats[i].n_candidates = ctx->h_mirror->cand[i];
        stats[i].n_after_nms = ctx->h_mirror->surv[i];
        stats[i].n_kept = ctx->h_mirror->kept[i];
    }
    if (nlevels) *nlevels = nl;
    return check_overflow(ctx);
}

int efx_convert(const void* h_keypoints, size_t kps_pitch, int n, efx_keypoint* out)
{
    // EfficientFeaturesImpl::convert, cuda_efficient_features.cpp:323-349
    if (n < 0 || (n > 0 && (!h_keypoints || !out))) return EFX_ERR_BAD_ARG;
    const unsigned char* b = static_cast<const unsigned char*>(h_keypoints);
    for (int i = 0; i < n; i++) {
        uint32_t loc; float resp, ang, size; int32_t oct;
        memcpy(&loc, b + 0 * kps_pitch + 4 * (size_t)i, 4);
        memcpy(&resp, b + 1 * kps_pitch + 4 * (size_t)i, 4);
        memcpy(&ang, b + 2 * kps_pitch + 4 * (size_t)i, 4);
        memcpy(&oct, b + 3 * kps_pitch + 4 * (size_t)i, 4);
        memcpy(&size, b + 4 * kps_pitch + 4 * (size_t)i, 4);
        out[i].x = (float)(int16_t)(loc & 0xffff);
        out[i].y = (float)(int16_t)(loc >> 16);
        out[i].response = resp; out[i].angle = ang; out[i].octave = oct; out[i].size = size; out[i].class_id = -1;
    }
    return EFX_OK;
}

static int host_detect_impl(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
                            efx_keypoint* keypoints, uint8_t* h_desc, size_t desc_pitch, int capacity, int* n, bool want_desc,
                            const uint8_t* h_mask = nullptr, size_t mask_pitch = 0)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    if (!h_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols) return set_err(ctx->err, EFX_ERR_BAD_ARG, "bad image arguments");
    if (capacity < 0 || !n || (capacity > 0 && !keypoints)) return set_err(ctx->err, EFX_ERR_BAD_ARG, "bad output arguments");
    const int nbytes = efx_descriptor_size(ctx);
    if (want_desc && (!h_desc || desc_pitch < (size_t)nbytes)) return set_err(ctx->err, EFX_ERR_BAD_ARG, "bad descriptor buffer");
    const size_t ipitch = align_up((size_t)cols, 256);
    const size_t kpitch = align_up((size_t)(capacity > 0 ? capacity : 1) * 4, 256);
    HIP_TRY(ctx->err, ctx->img.reserve(ipitch * rows));
    HIP_TRY(ctx->err, ctx->kps.reserve(kpitch * EFX_ROWS_COUNT));
    if (want_desc) HIP_TRY(ctx->err, ctx->descout.reserve((size_t)(capacity > 0 ? capacity : 1) * nbytes));
    HIP_TRY(ctx->err, hipMemcpy2D(ctx->img.p, ipitch, h_image, pitch, cols, rows, hipMemcpyHostToDevice));   // getInputMat upload, .cpp:75-77
    if (h_mask) {
        if (mask_pitch < (size_t)cols) return set_err(ctx->err, EFX_ERR_BAD_ARG, "mask must be an 8-bit image of the frame size");
        HIP_TRY(ctx->err, ctx->maskbuf.reserve(ipitch * rows));
        HIP_TRY(ctx->err, hipMemcpy2D(ctx->maskbuf.p, ipitch, h_mask, mask_pitch, cols, rows, hipMemcpyHostToDevice));
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        int rc = detect_common(ctx, static_cast<const uint8_t*>(ctx->img.p), rows, cols, ipitch, ctx->kps.p, kpitch,
                               want_desc ? static_cast<uint8_t*>(ctx->descout.p) : nullptr, nbytes, capacity, nullptr, nullptr,
                               h_mask ? static_cast<const uint8_t*>(ctx->maskbuf.p) : nullptr, ipitch);
        if (rc) return rc;
        HIP_TRY(ctx->err, hipStreamSynchronize(nullptr));
        if (fetch_summary(ctx) != EFX_OK) return set_err(ctx->err, EFX_ERR_HIP, "summary copy failed");
        // (a void frame -- records that failed their range checks, DESIGN.md section 7 -- is run once more)
        if (check_overflow(ctx) == EFX_OK || attempt == 1) break;
    }
    if (ctx->h_mirror->overflow) return EFX_ERR_OVERFLOW;
    const int cnt = ctx->h_mirror->n_out;
    *n = cnt;
    if (cnt > 0) {
        std::vector<unsigned char> tmp(kpitch * EFX_ROWS_COUNT);
        HIP_TRY(ctx->err, hipMemcpy(tmp.data(), ctx->kps.p, kpitch * EFX_ROWS_COUNT, hipMemcpyDeviceToHost));
        efx_convert(tmp.data(), kpitch, cnt, keypoints);
        if (want_desc) HIP_TRY(ctx->err, hipMemcpy2D(h_desc, desc_pitch, ctx->descout.p, nbytes, nbytes, cnt, hipMemcpyDeviceToHost));
    }
    return EFX_OK;
}

int efx_detect(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
               efx_keypoint* keypoints, int capacity, int* n)
{
    return host_detect_impl(ctx, h_image, rows, cols, pitch, keypoints, nullptr, 0, capacity, n, false);
}

int efx_detect_and_compute(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
                           efx_keypoint* keypoints, uint8_t* h_descriptors, size_t desc_pitch, int capacity, int* n)
{
    return host_detect_impl(ctx, h_image, rows, cols, pitch, keypoints, h_descriptors, desc_pitch, capacity, n, true);
}

// Feature2D::detectAndCompute with its full argument list: (image, mask, keypoints, descriptors, useProvidedKeypoints)
int efx_detect_and_compute_ex(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
                              const uint8_t* h_mask, size_t mask_pitch, efx_keypoint* keypoints, uint8_t* h_descriptors, size_t desc_pitch,
                              int capacity, int* n, int use_provided_keypoints)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    if (!use_provided_keypoints)
        return host_detect_impl(ctx, h_image, rows, cols, pitch, keypoints, h_descriptors, desc_pitch, capacity, n, h_descriptors != nullptr,
                                h_mask, mask_pitch);
    // spec S13: *n keypoints are given; only the descriptors are written
    if (!h_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols) return set_err(ctx->err, EFX_ERR_BAD_ARG, "bad image arguments");
    if (!n || *n < 0 || (*n > 0 && (!keypoints || !h_descriptors))) return set_err(ctx->err, EFX_ERR_BAD_ARG, "bad keypoint / descriptor arguments");
    const int cnt = *n, nbytes = efx_descriptor_size(ctx);
    if (cnt == 0) return EFX_OK;
    if (desc_pitch < (size_t)nbytes) return set_err(ctx->err, EFX_ERR_BAD_ARG, "desc_pitch smaller than the descriptor");
    const size_t ipitch = align_up((size_t)cols, 256), kpitch = align_up((size_t)cnt * 4, 256);
    HIP_TRY(ctx->err, ctx->img.reserve(ipitch * rows));
    HIP_TRY(ctx->err, ctx->kps.reserve(kpitch * EFX_ROWS_COUNT));
    HIP_TRY(ctx->err, ctx->descout.reserve((size_t)cnt * nbytes));
    std::vector<unsigned char> tmp(kpitch * EFX_ROWS_COUNT, 0);
    for (int i = 0; i < cnt; i++) {
        const uint32_t loc = (uint32_t)(uint16_t)(int16_t)keypoints[i].x | ((uint32_t)(uint16_t)(int16_t)keypoints[i].y << 16);
        memcpy(&tmp[0 * kpitch + 4 * (size_t)i], &loc, 4);
        memcpy(&tmp[1 * kpitch + 4 * (size_t)i], &keypoints[i].response, 4);
        memcpy(&tmp[2 * kpitch + 4 * (size_t)i], &keypoints[i].angle, 4);
        memcpy(&tmp[3 * kpitch + 4 * (size_t)i], &keypoints[i].octave, 4);
        memcpy(&tmp[4 * kpitch + 4 * (size_t)i], &keypoints[i].size, 4);
    }
    HIP_TRY(ctx->err, hipMemcpy2D(ctx->img.p, ipitch, h_image, pitch, cols, rows, hipMemcpyHostToDevice));
    HIP_TRY(ctx->err, hipMemcpy(ctx->kps.p, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    int rc = compute_provided(ctx, static_cast<const uint8_t*>(ctx->img.p), rows, cols, ipitch, ctx->kps.p, kpitch, cnt,
                              static_cast<uint8_t*>(ctx->descout.p), nbytes, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx->err, hipStreamSynchronize(nullptr));
    HIP_TRY(ctx->err, hipMemcpy2D(h_descriptors, desc_pitch, ctx->descout.p, nbytes, nbytes, cnt, hipMemcpyDeviceToHost));
    return EFX_OK;
}

int efx_compute(efx_context* ctx, const uint8_t* h_image, int rows, int cols, size_t pitch,
                const efx_keypoint* keypoints, int n, uint8_t* h_descriptors, size_t desc_pitch)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    return describe_host(ctx->desc, ctx->err, h_image, rows, cols, pitch, keypoints, n, h_descriptors, desc_pitch);
}

// ---- stand-alone describers ----
static int describer_create(int kind, float scale, int nbits_enum, efx_describer** out)
{
    if (!out) return set_err(g_create_error, EFX_ERR_BAD_ARG, "null output handle");
    *out = nullptr;
    const int nbits = nbits_from_enum(nbits_enum);
    if (!nbits) return set_err(g_create_error, EFX_ERR_BAD_ARG, "n_bits should be either SIZE_512_BITS or SIZE_256_BITS");   // bad.cpp:316
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(g_create_error, EFX_ERR_NO_DEVICE, "no HIP device: the efficient-features path has no CPU fallback");
    efx_describer* d = new (std::nothrow) efx_describer;
    if (!d) return set_err(g_create_error, EFX_ERR_NOMEM, "out of host memory");
    int rc = describer_init(d->d, kind, nbits, scale);
    if (rc) { g_create_error = d->d.err; delete d; return rc; }
    *out = d;
    return EFX_OK;
}

int efx_bad_create(float scale_factor, int nbits, efx_describer** out) { return describer_create(0, scale_factor, nbits, out); }
int efx_hashsift_create(float cropping_scale, int nbits, efx_describer** out) { return describer_create(1, cropping_scale, nbits, out); }
int efx_describer_destroy(efx_describer* d) { delete d; return EFX_OK; }
int efx_describer_descriptor_size(const efx_describer* d) { return d ? d->d.nbits / 8 : 0; }
const char* efx_describer_last_error(const efx_describer* d) { return d ? d->d.err.c_str() : g_create_error.c_str(); }

int efx_describer_compute_kp4_async(efx_describer* d, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                    const float* d_kp4, int n, float max_size, uint8_t* d_descriptors, size_t desc_pitch, void* stream)
{
    if (!d) return EFX_ERR_BAD_ARG;
    return describe_single(d->d, d->d.err, d_image, rows, cols, pitch, reinterpret_cast<const float4*>(d_kp4), n, max_size,
                           d_descriptors, desc_pitch, nullptr, nullptr, (hipStream_t)stream);
}

int efx_describer_compute_async(efx_describer* d, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                const void* d_keypoints, size_t kps_pitch, int n, uint8_t* d_descriptors, size_t desc_pitch, void* stream)
{
    if (!d) return EFX_ERR_BAD_ARG;
    return describe_5xn(d->d, d->d.err, d_image, rows, cols, pitch, d_keypoints, kps_pitch, n, d_descriptors, desc_pitch, (hipStream_t)stream);
}

int efx_describer_compute(efx_describer* d, const uint8_t* h_image, int rows, int cols, size_t pitch,
                          const efx_keypoint* keypoints, int n, uint8_t* h_descriptors, size_t desc_pitch)
{
    if (!d) return EFX_ERR_BAD_ARG;
    return describe_host(d->d, d->d.err, h_image, rows, cols, pitch, keypoints, n, h_descriptors, desc_pitch);
}

int efx_describer_hashsift_debug_async(efx_describer* d, const uint8_t* d_image, int rows, int cols, size_t pitch,
                                       const float* d_kp4, int n, float max_size, float* d_responses, float* d_T, void* stream)
{
    if (!d) return EFX_ERR_BAD_ARG;
    if (d->d.kind != 1) return set_err(d->d.err, EFX_ERR_BAD_ARG, "not a HashSIFT describer");
    return describe_single(d->d, d->d.err, d_image, rows, cols, pitch, reinterpret_cast<const float4*>(d_kp4), n, max_size,
                           nullptr, 0, d_responses, d_T, (hipStream_t)stream);
}

// ---- HPatches exporter helpers ----
int efx_ic_angles_async(const uint8_t* d_image, int rows, int cols, size_t pitch, float* d_kp4, int n, int patch_size, void* stream)
{
    if (!d_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols || n < 0 || (n > 0 && !d_kp4) || patch_size < 3 || patch_size > 257)
        return set_err(g_create_error, EFX_ERR_BAD_ARG, "bad arguments (patch_size must be in [3, 257])");
    hipError_t e = efx_launch_ic_angles(d_image, pitch, rows, cols, reinterpret_cast<float4*>(d_kp4), n, patch_size, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(g_create_error, EFX_ERR_HIP, "ic_angles launch failed: %s", hipGetErrorString(e));
    return EFX_OK;
}

int efx_ic_angles(const uint8_t* h_image, int rows, int cols, size_t pitch, efx_keypoint* keypoints, int n, int patch_size)
{
    if (!h_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols || n < 0 || (n > 0 && !keypoints))
        return set_err(g_create_error, EFX_ERR_BAD_ARG, "bad arguments");
    if (n == 0) return EFX_OK;
    DevBuf img, kp;
    const size_t ipitch = align_up((size_t)cols, 256);
    std::vector<float4> k((size_t)n);
    for (int i = 0; i < n; i++) k[i] = make_float4(keypoints[i].x, keypoints[i].y, keypoints[i].size, keypoints[i].angle);
    hipError_t e = img.reserve(ipitch * rows);
    if (e == hipSuccess) e = kp.reserve((size_t)n * sizeof(float4));
    if (e == hipSuccess) e = hipMemcpy2D(img.p, ipitch, h_image, pitch, cols, rows, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(kp.p, k.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice);
    int rc = EFX_OK;
    if (e == hipSuccess) rc = efx_ic_angles_async(static_cast<const uint8_t*>(img.p), rows, cols, ipitch, static_cast<float*>(kp.p), n, patch_size, nullptr);
    if (e == hipSuccess && rc == EFX_OK) e = hipMemcpy(k.data(), kp.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost);
    img.release(); kp.release();
    if (rc) return rc;
    if (e != hipSuccess) return set_err(g_create_error, EFX_ERR_HIP, "efx_ic_angles failed: %s", hipGetErrorString(e));
    for (int i = 0; i < n; i++) keypoints[i].angle = k[i].w;
    return EFX_OK;
}

// saveDescriptors, samples/hpatches_description.cpp:76-105: one line per descriptor, bits MSB first, comma separated
long efx_descriptors_to_csv(const uint8_t* h_descriptors, int n, int nbytes, size_t desc_pitch, char* out, size_t out_capacity)
{
    if (n < 0 || nbytes <= 0 || (n > 0 && !h_descriptors) || desc_pitch < (size_t)nbytes) return -1;
    const size_t line = (size_t)nbytes * 16;          // 8 x "b," per byte, the last comma replaced by the newline
    const size_t need = line * (size_t)n;
    if (!out) return (long)need;
    if (out_capacity < need) return -1;
    char* p = out;
    for (int i = 0; i < n; i++) {
        const uint8_t* d = h_descriptors + (size_t)i * desc_pitch;
        for (int j = 0; j < nbytes; j++)
            for (int k = 7; k >= 0; k--) { *p++ = (char)('0' + ((d[j] >> k) & 1)); *p++ = ','; }
        p[-1] = '\n';
    }
    return (long)need;
}

// ---- input stage ----
int efx_cvt_gray_async(const uint8_t* d_src, int rows, int cols, size_t src_pitch, int channels,
                       uint8_t* d_gray, size_t gray_pitch, void* stream)
{
    if (!d_src || !d_gray || rows <= 0 || cols <= 0 || (channels != 3 && channels != 4) || src_pitch < (size_t)cols * channels ||
        gray_pitch < (size_t)cols)
        return set_err(g_create_error, EFX_ERR_BAD_ARG, "Image should be 8UC3 or 8UC4 with valid pitches");   // bad.cpp:279
    hipError_t e = efx_launch_cvt_gray(d_src, src_pitch, rows, cols, channels, d_gray, gray_pitch, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(g_create_error, EFX_ERR_HIP, "cvt_gray launch failed: %s", hipGetErrorString(e));
    return EFX_OK;
}

int efx_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return EFX_ERR_BAD_ARG;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return set_err(g_create_error, EFX_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return EFX_OK;
}

int efx_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? EFX_OK : EFX_ERR_HIP; }

struct efx_uploader {
    static constexpr int NSLOT = 2, NCHUNK = 4;
    static constexpr size_t CHUNK = 4u << 20;
    DevBuf raw[NSLOT], gray[NSLOT];
    hipStream_t copy = nullptr;
    hipEvent_t uploaded[NSLOT] = {}, consumed[NSLOT] = {}, chunk_done[NCHUNK] = {};
    void* staging[NCHUNK] = {};
    bool chunk_busy[NCHUNK] = {};
    bool slot_used[NSLOT] = {};
    bool slot_released[NSLOT] = {};         // consumed[] was recorded explicitly (efx_uploader_release)
    hipStream_t slot_stream[NSLOT] = {};    // the consumer stream the slot's frame was handed to
    unsigned long long frame = 0;
    int slot_of(const uint8_t* d) const
    {
        for (int i = 0; i < NSLOT; i++)
            if (slot_used[i] && d && (d == raw[i].p || d == gray[i].p)) return i;
        return -1;
    }
    std::string err;
    ~efx_uploader()
    {
        if (copy) (void)hipStreamSynchronize(copy);
        for (int i = 0; i < NSLOT; i++) {
            raw[i].release(); gray[i].release();
            if (uploaded[i]) (void)hipEventDestroy(uploaded[i]);
            if (consumed[i]) (void)hipEventDestroy(consumed[i]);
        }
        for (int i = 0; i < NCHUNK; i++) {
            if (chunk_done[i]) (void)hipEventDestroy(chunk_done[i]);
            if (staging[i]) (void)hipHostFree(staging[i]);
        }
        if (copy) (void)hipStreamDestroy(copy);
    }
};

int efx_uploader_create(efx_uploader** out)
{
    if (!out) return set_err(g_create_error, EFX_ERR_BAD_ARG, "null output handle");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(g_create_error, EFX_ERR_NO_DEVICE, "no HIP device");
    efx_uploader* u = new (std::nothrow) efx_uploader;
    if (!u) return set_err(g_create_error, EFX_ERR_NOMEM, "out of host memory");
    hipError_t e = hipStreamCreateWithFlags(&u->copy, hipStreamNonBlocking);
    for (int i = 0; i < efx_uploader::NSLOT && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&u->uploaded[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&u->consumed[i], hipEventDisableTiming);
    }
    for (int i = 0; i < efx_uploader::NCHUNK && e == hipSuccess; i++) e = hipEventCreateWithFlags(&u->chunk_done[i], hipEventDisableTiming);
    if (e != hipSuccess) { set_err(g_create_error, EFX_ERR_HIP, "uploader setup failed: %s", hipGetErrorString(e)); delete u; return EFX_ERR_HIP; }
    *out = u;
    return EFX_OK;
}

int efx_uploader_destroy(efx_uploader* u) { delete u; return EFX_OK; }
const char* efx_uploader_last_error(const efx_uploader* u) { return u ? u->err.c_str() : g_create_error.c_str(); }

int efx_upload_gray_async(efx_uploader* u, const uint8_t* h_image, int rows, int cols, size_t pitch, int channels,
                          const uint8_t** d_gray, size_t* gray_pitch, void* stream_)
{
    if (!u) return EFX_ERR_BAD_ARG;
    if (!h_image || rows <= 0 || cols <= 0 || (channels != 1 && channels != 3 && channels != 4) || pitch < (size_t)cols * channels ||
        !d_gray || !gray_pitch)
        return set_err(u->err, EFX_ERR_BAD_ARG, "Image should be 8UC1, 8UC3 or 8UC4");               // bad.cpp:279
    hipStream_t stream = (hipStream_t)stream_;
    const int slot = (int)(u->frame % efx_uploader::NSLOT);
    const int prev = (int)((u->frame + efx_uploader::NSLOT - 1) % efx_uploader::NSLOT);
    // Implicit contract: the consumers of the previous frame (slot `prev`) were enqueued on the stream THAT frame was handed
    // to, before this call.  A caller that enqueues them later, or elsewhere, says so with efx_uploader_release().
    if (u->frame > 0 && !u->slot_released[prev]) HIP_TRY(u->err, hipEventRecord(u->consumed[prev], u->slot_stream[prev]));
    // this slot was last used by frame - NSLOT: its consumed event was recorded one call ago or by efx_uploader_release
    if (u->slot_used[slot]) HIP_TRY(u->err, hipStreamWaitEvent(u->copy, u->consumed[slot], 0));
    u->slot_released[slot] = false;
    u->slot_stream[slot] = stream;
    const size_t row_bytes = (size_t)cols * channels;
    const size_t dpitch = align_up(row_bytes, 256);
    HIP_TRY(u->err, u->raw[slot].reserve(dpitch * rows));
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, h_image) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    if (pinned) {
        HIP_TRY(u->err, hipMemcpy2DAsync(u->raw[slot].p, dpitch, h_image, pitch, row_bytes, rows, hipMemcpyHostToDevice, u->copy));
    } else {
        // pageable memory: row blocks through a ring of pinned chunks; the host copy of block i+1 overlaps the DMA of block i
        const int rows_per_chunk = (int)std::max<size_t>(1, efx_uploader::CHUNK / row_bytes);
        if (row_bytes > efx_uploader::CHUNK) return set_err(u->err, EFX_ERR_UNSUPPORTED, "image row larger than the staging chunk");
        int c = 0;
        for (int r0 = 0; r0 < rows; r0 += rows_per_chunk, c = (c + 1) % efx_uploader::NCHUNK) {
            const int nr = std::min(rows_per_chunk, rows - r0);
            if (!u->staging[c]) HIP_TRY(u->err, hipHostMalloc(&u->staging[c], efx_uploader::CHUNK, hipHostMallocDefault));
            if (u->chunk_busy[c]) HIP_TRY(u->err, hipEventSynchronize(u->chunk_done[c]));
            uint8_t* st = static_cast<uint8_t*>(u->staging[c]);
            for (int r = 0; r < nr; r++) memcpy(st + (size_t)r * row_bytes, h_image + (size_t)(r0 + r) * pitch, row_bytes);
            HIP_TRY(u->err, hipMemcpy2DAsync(static_cast<uint8_t*>(u->raw[slot].p) + (size_t)r0 * dpitch, dpitch, st, row_bytes, row_bytes, nr,
                                             hipMemcpyHostToDevice, u->copy));
            HIP_TRY(u->err, hipEventRecord(u->chunk_done[c], u->copy));
            u->chunk_busy[c] = true;
        }
    }
    HIP_TRY(u->err, hipEventRecord(u->uploaded[slot], u->copy));
    HIP_TRY(u->err, hipStreamWaitEvent(stream, u->uploaded[slot], 0));
    u->slot_used[slot] = true;
    u->frame++;
    if (channels == 1) {
        *d_gray = static_cast<const uint8_t*>(u->raw[slot].p); *gray_pitch = dpitch;
        return EFX_OK;
    }
    const size_t gpitch = align_up((size_t)cols, 256);
    HIP_TRY(u->err, u->gray[slot].reserve(gpitch * rows));
    hipError_t e = efx_launch_cvt_gray(static_cast<const uint8_t*>(u->raw[slot].p), dpitch, rows, cols, channels,
                                       static_cast<uint8_t*>(u->gray[slot].p), gpitch, stream);
    if (e != hipSuccess) return set_err(u->err, EFX_ERR_HIP, "cvt_gray launch failed: %s", hipGetErrorString(e));
    *d_gray = static_cast<const uint8_t*>(u->gray[slot].p); *gray_pitch = gpitch;
    return EFX_OK;
}

int efx_uploader_release(efx_uploader* u, const uint8_t* d_gray, void* stream)
{
    if (!u) return EFX_ERR_BAD_ARG;
    const int slot = u->slot_of(d_gray);
    if (slot < 0) return set_err(u->err, EFX_ERR_BAD_ARG, "not a frame of this uploader (or already recycled)");
    HIP_TRY(u->err, hipEventRecord(u->consumed[slot], (hipStream_t)stream));
    u->slot_released[slot] = true;
    return EFX_OK;
}

int efx_uploader_wait_uploaded(efx_uploader* u, const uint8_t* d_gray)
{
    if (!u) return EFX_ERR_BAD_ARG;
    const int slot = u->slot_of(d_gray);
    if (slot < 0) return set_err(u->err, EFX_ERR_BAD_ARG, "not a frame of this uploader (or already recycled)");
    HIP_TRY(u->err, hipEventSynchronize(u->uploaded[slot]));
    return EFX_OK;
}

int efx_describer_compute_color(efx_describer* d, const uint8_t* h_image, int rows, int cols, size_t pitch, int channels,
                                const efx_keypoint* keypoints, int n, uint8_t* h_descriptors, size_t desc_pitch)
{
    if (!d) return EFX_ERR_BAD_ARG;
    if (channels == 1) return describe_host(d->d, d->d.err, h_image, rows, cols, pitch, keypoints, n, h_descriptors, desc_pitch);
    if ((channels != 3 && channels != 4) || !h_image || rows <= 0 || cols <= 0 || pitch < (size_t)cols * channels)
        return set_err(d->d.err, EFX_ERR_BAD_ARG, "Image should be 8UC1, 8UC3 or 8UC4");             // bad.cpp:279
    // convertToGray on the host side of the describer (bad.cpp:268-281): spec S11, same integers as the device kernel
    std::vector<uint8_t> gray((size_t)rows * cols);
    for (int y = 0; y < rows; y++) {
        const uint8_t* s = h_image + (size_t)y * pitch;
        uint8_t* g = gray.data() + (size_t)y * cols;
        for (int x = 0; x < cols; x++, s += channels) g[x] = (uint8_t)((3735u * s[0] + 19235u * s[1] + 9798u * s[2] + 16384u) >> 15);
    }
    return describe_host(d->d, d->d.err, gray.data(), rows, cols, (size_t)cols, keypoints, n, h_descriptors, desc_pitch);
}

int efx_profile_set_stride(efx_context* ctx, int stride)
{
    if (!ctx || stride < 1) return EFX_ERR_BAD_ARG;
    ctx->prof_stride = stride; ctx->prof_calls = 0;
    return EFX_OK;
}

int efx_profile_set_groups(efx_context* ctx, unsigned groups)
{
    if (!ctx) return EFX_ERR_BAD_ARG;
    ctx->prof_skip = ~groups & 0x3fu;
    return EFX_OK;
}

int efx_profile_enable(efx_context* ctx, int max_launches)
{
    if (!ctx || max_launches < 0 || max_launches > 65536) return EFX_ERR_BAD_ARG;
    ctx->prof_calls = 0;
    for (hipEvent_t e : ctx->prof_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_stop) (void)hipEventDestroy(e);
    ctx->prof_start.clear(); ctx->prof_stop.clear(); ctx->prof_level.clear(); ctx->prof_count = 0;
    for (int i = 0; i < max_launches; i++) {
        hipEvent_t a, b;
        HIP_TRY(ctx->err, hipEventCreate(&a));
        HIP_TRY(ctx->err, hipEventCreate(&b));
        ctx->prof_start.push_back(a); ctx->prof_stop.push_back(b);
    }
    ctx->prof_level.assign((size_t)max_launches, 0);
    return EFX_OK;
}

int efx_profile_read(efx_context* ctx, float* ms, int* level, int capacity, int* n)
{
    if (!ctx || !n) return EFX_ERR_BAD_ARG;
    const int cnt = ctx->prof_count < capacity ? ctx->prof_count : capacity;
    for (int i = 0; i < cnt; i++) {
        float t = 0.f;
        HIP_TRY(ctx->err, hipEventElapsedTime(&t, ctx->prof_start[i], ctx->prof_stop[i]));
        if (ms) ms[i] = t;
        if (level) level[i] = ctx->prof_level[i];
    }
    *n = cnt;
    ctx->prof_count = 0;
    return EFX_OK;
}

// ---- brute-force Hamming matcher ----
int efx_matcher_create(efx_matcher** out)
{
    if (!out) return set_err(g_create_error, EFX_ERR_BAD_ARG, "null output handle");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(g_create_error, EFX_ERR_NO_DEVICE, "no HIP device: the matcher has no CPU fallback");
    efx_matcher* m = new (std::nothrow) efx_matcher;
    if (!m) return set_err(g_create_error, EFX_ERR_NOMEM, "out of host memory");
    *out = m;
    return EFX_OK;
}
int efx_matcher_destroy(efx_matcher* m) { delete m; return EFX_OK; }
const char* efx_matcher_last_error(const efx_matcher* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

static int match_args_ok(efx_matcher* m, const uint8_t* q, size_t qp, int nq, const uint8_t* t, size_t tp, int nt, int db)
{
    if (db != 32 && db != 64) return set_err(m->err, EFX_ERR_BAD_ARG, "descriptor size must be 32 or 64 bytes");
    if (nq < 0 || nt < 0) return set_err(m->err, EFX_ERR_BAD_ARG, "negative row count");
    if ((nq > 0 && !q) || (nt > 0 && !t)) return set_err(m->err, EFX_ERR_BAD_ARG, "null descriptors");
    if (qp < (size_t)db || tp < (size_t)db || ((qp | tp | (uintptr_t)q | (uintptr_t)t) & 3u))
        return set_err(m->err, EFX_ERR_BAD_ARG, "descriptor rows must be 4-byte aligned and at least desc_bytes apart");
    return EFX_OK;
}

static int knn2_run(efx_matcher* m, const uint8_t* q, size_t qp, int nq, const uint8_t* t, size_t tp, int nt, int db,
                    int* idx, int* dist, hipStream_t stream)
{
    if (nq == 0) return EFX_OK;
    if (nq >= 128 && nt >= 64 && !m->no_mfma) {
        // large sets: the distance matrix as a GEMM on the matrix cores (match_kernels.hip): FP4 (MX) operands, int8 with EFX_MATCH_NO_FP4
        // (query block, train chunk) pairs: ONE round of the workgroups the chip holds of the kernel that will run (two 512-thread
        // workgroups per CU for 512 bits, three for 256) -- fewer chunks mean fewer best-two updates per wave (a wave's updates
        // fall off as 1 / trains seen), a second, partly filled round costs as much as the first; 40 000 x 40 000: 512 bit
        // 2 / 3 / 4 / 6 chunks 0.576 / 0.436 / 0.436 / 0.469 ms, 256 bit 0.376 / 0.295 / 0.277 / 0.323
        int nchunks = efx_knn2_mfma_resident_workgroups(db, m->no_fp4 ? 0 : 1) / ((nq + 255) / 256);
        { static const int env = [] { const char* v = getenv("EFX_MATCH_CHUNKS"); return v ? atoi(v) : 0; }(); if (env > 0) nchunks = env; }   // INVESTIGATION knob
        if (nchunks < 1) nchunks = 1;
        if (nchunks > 64) nchunks = 64;
        const int ntiles = (nt + 31) / 32;
        if (nchunks > ntiles) nchunks = ntiles;
        HIP_TRY(m->err, m->scratch.reserve((size_t)nchunks * nq * 16));
        HIP_TRY(m->err, m->expanded.reserve(efx_knn2_mfma_scratch(nq, nt, db)));
        hipError_t e = efx_launch_knn2_mfma(q, qp, nq, t, tp, nt, db, m->expanded.p, m->scratch.p, nchunks, idx, dist, stream, m->no_fp4 ? 0 : 1);
        if (e != hipSuccess) return set_err(m->err, EFX_ERR_HIP, "knn launch failed: %s", hipGetErrorString(e));
        return EFX_OK;
    }
    // enough (query block, train chunk) pairs to fill the chip: ~4 workgroups per CU
    int nchunks = 1024 / ((nq + 255) / 256);
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 64) nchunks = 64;
    if (nchunks > nt) nchunks = nt > 0 ? nt : 1;
    HIP_TRY(m->err, m->scratch.reserve((size_t)nchunks * nq * 16));
    hipError_t e = efx_launch_knn2(q, qp, nq, t, tp, nt, db, m->scratch.p, nchunks, idx, dist, stream);
    if (e != hipSuccess) return set_err(m->err, EFX_ERR_HIP, "knn launch failed: %s", hipGetErrorString(e));
    return EFX_OK;
}

int efx_match_knn2_async(efx_matcher* m, const uint8_t* d_query, size_t q_pitch, int nq, const uint8_t* d_train, size_t t_pitch, int nt,
                         int desc_bytes, int* d_idx, int* d_dist, void* stream)
{
    if (!m) return EFX_ERR_BAD_ARG;
    int rc = match_args_ok(m, d_query, q_pitch, nq, d_train, t_pitch, nt, desc_bytes);
    if (rc) return rc;
    if (nq > 0 && (!d_idx || !d_dist)) return set_err(m->err, EFX_ERR_BAD_ARG, "null outputs");
    return knn2_run(m, d_query, q_pitch, nq, d_train, t_pitch, nt, desc_bytes, d_idx, d_dist, (hipStream_t)stream);
}

int efx_match_crosscheck_async(efx_matcher* m, const uint8_t* d_query, size_t q_pitch, int nq, const uint8_t* d_train, size_t t_pitch, int nt,
                               int desc_bytes, int* d_match, int* d_dist, void* stream)
{
    if (!m) return EFX_ERR_BAD_ARG;
    int rc = match_args_ok(m, d_query, q_pitch, nq, d_train, t_pitch, nt, desc_bytes);
    if (rc) return rc;
    if (nq == 0) return EFX_OK;
    if (!d_match) return set_err(m->err, EFX_ERR_BAD_ARG, "null outputs");
    HIP_TRY(m->err, m->a_idx.reserve((size_t)nq * 8)); HIP_TRY(m->err, m->a_dist.reserve((size_t)nq * 8));
    HIP_TRY(m->err, m->b_idx.reserve((size_t)(nt > 0 ? nt : 1) * 8)); HIP_TRY(m->err, m->b_dist.reserve((size_t)(nt > 0 ? nt : 1) * 8));
    rc = knn2_run(m, d_query, q_pitch, nq, d_train, t_pitch, nt, desc_bytes, (int*)m->a_idx.p, (int*)m->a_dist.p, (hipStream_t)stream);
    if (rc) return rc;
    rc = knn2_run(m, d_train, t_pitch, nt, d_query, q_pitch, nq, desc_bytes, (int*)m->b_idx.p, (int*)m->b_dist.p, (hipStream_t)stream);
    if (rc) return rc;
    hipError_t e = efx_launch_crosscheck((const int*)m->a_idx.p, (const int*)m->b_idx.p, nq, d_match, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(m->err, EFX_ERR_HIP, "crosscheck launch failed: %s", hipGetErrorString(e));
    if (d_dist) {
        // distance of the kept pairs = first column of the query->train result
        HIP_TRY(m->err, hipMemcpy2DAsync(d_dist, 4, m->a_dist.p, 8, 4, (size_t)nq, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return EFX_OK;
}

int efx_level_geometry(const efx_context* ctx, int rows, int cols, int level, int* lrows, int* lcols, float* scale)
{
    if (!ctx || level < 0 || level >= ctx->p.nlevels) return EFX_ERR_BAD_ARG;
    float s = 1.f;
    for (int i = 1; i <= level; i++) s *= ctx->p.scale_factor;
    const float inv = 1.f / s;
    if (lrows) *lrows = level == 0 ? rows : cv_round_f(inv * (float)rows);
    if (lcols) *lcols = level == 0 ? cols : cv_round_f(inv * (float)cols);
    if (scale) *scale = s;
    return EFX_OK;
}

int efx_copy_level_async(efx_context* ctx, int level, uint8_t* d_dst, size_t dst_pitch, void* stream)
{
    if (!ctx || !ctx->has_frame || level < 0 || level >= ctx->h_table.nlevels || !d_dst) return EFX_ERR_BAD_ARG;
    const LevelDev& L = ctx->h_table.lv[level];
    const uint8_t* src = level == 0 ? ctx->last_img0 : static_cast<const uint8_t*>(ctx->pyramid.p) + L.img_off;
    const size_t sp = level == 0 ? (size_t)ctx->last_pitch0 : (size_t)L.pitch;
    hipError_t e = efx_launch_copy2d(src, sp, d_dst, dst_pitch, L.rows, L.cols, (hipStream_t)stream);
    if (e != hipSuccess) return set_err(ctx->err, EFX_ERR_HIP, "copy failed: %s", hipGetErrorString(e));
    return EFX_OK;
}

} // extern "C"
