// efx_device.h -- device-side data layout shared by the HIP kernels and the C-ABI host code.
//
// HBM layout of one context (all owned by the context, grow-only, reused across frames):
//   pyramid   : levels 1..n-1, u8, row pitch rounded up to 256 B (level 0 is the caller's image, aliased)
//   slots     : one 512-B slot per 64x64 tile: the tile's FAST corners as 16-bit tile coordinates in canonical order (up to 256),
//               or its 64 x 64 corner bitmap (more) -- fixed size, whatever the frame holds (round 6)
//   cand      : per level EXACTLY cap = cvRound(0.1 w h) records {xy, harris} (8 B) (.cpp:252): the level's corners in canonical
//               order; a tile's corners start at its canonical rank, corners of rank >= cap do not exist (spec S2)
//   surv      : per level, the radius-NMS survivors of a tile at the tile's place in `cand`'s index space (same size: no
//               allocation, nothing can overflow)
//   tile_hdr  : one 64-B header per 64x64 tile of every level
//   kp4/lvl   : float4 {x, y, 31, angle} level-local keypoints + their level, input of the describers
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define EFX_TILE 64            // canonical-order tile edge (pixels)
#define EFX_CELL 16            // NMS cell edge (CELL_SIZE, cuda_efficient_features.cu:35)
#define EFX_CELLS_PER_TILE 16
#define EFX_HALO 4             // FAST needs 3, Harris 7x7 of 3x3 Sobel needs 4, resize needs 1
#define EFX_LT (EFX_TILE + 2 * EFX_HALO)   // LDS tile edge (72)
#define EFX_LP 80               // LDS tile row pitch in bytes: 20 dwords, so rows 4 apart are 16 banks apart
#define EFX_MAX_LEVELS 32
#define EFX_HALF_PATCH 15      // cuda_efficient_features.cpp:34
#define EFX_PATCH_SIZE 31      // cuda_efficient_features.cpp:33
#define EFX_NXCD 8
#define EFX_SLOT_LIST 256       // a tile with at most this many FAST corners leaves them as a list in its slot, else as a bitmap
#define EFX_SLOT_BYTES 512      // 256 x u16 == 64 x u64
#define EFX_HIST_BITS 14        // selection histogram: top bits of the 64-bit key (sign, exponent, 5 mantissa bits of the response)
#define EFX_HIST_BINS (1 << EFX_HIST_BITS)
#define EFX_SEL_LIST_CAP 2048   // keys of the threshold's bin that are ranked exactly in LDS (more: the slow radix path of select_kernel)
#define EFX_SEL_WG_TILES 256    // tiles per counting workgroup of select_kernel (lane per tile)
#ifndef EFX_PACK_TPW
#define EFX_PACK_TPW 4          // tiles per wave of harris_packed_kernel (sparse frames; 4 / 8 / 16 measured: 26.5 / 28.6 / 30.0 us on the
                                // 1/f^1.3 8K frame, harris_kernel 33 -- tools/microbench/pack_ab.sh)
#endif
#define EFX_MAX_BATCH 16       // frames of one size a context runs through ONE launch of every kernel (blockIdx.y = frame)

struct LevelDev {
    int rows, cols;
    int pitch;                  // bytes; level 0: patched per call
    int tiles_x, tiles_y;
    int tile_base;              // index of the level's first tile in the global tile arrays
    int cap;                    // cvRound(0.1 * area), cuda_efficient_features.cpp:252
    int quota;                  // calcNumFeaturesPerLevel, cuda_efficient_features.cpp:159-174
    float scale;                // level scale (1.2^s in float)
    float fx, fy;               // resize factors with THIS level as destination: src = dst * f
    int active;                 // s >= firstLevel
    unsigned long long img_off; // byte offset of the level in the pyramid buffer (levels >= 1)
    unsigned long long cand_base;   // entry offset of the level in the cand AND surv arrays: the sum of the lower levels' caps
    unsigned long long cmax_base;   // entry offset of the level in the per-cell maxima table (tiles_x*4 x tiles_y*4)
    int row_base;                   // index of the level's first tile row in the per-row counters (RowCtr)
    int sel_wg0, sel_wgs;           // counting workgroups of select_kernel that hold tiles of this level: first, how many
    int pack_groups;                // groups of sixteen tiles (harris_packed_kernel's workgroups): ceil(tiles / 16), 0 for an inactive level
    int pad2_;
};

struct LevelTable {
    int nlevels;
    int total_tiles;
    int total_rows;                 // tile rows of all levels (RowCtr entries)
    int pad_;
    int* host_hint;                 // pinned, device-mapped host word (null: none): FAST corners of level 0 of the last frame + 1
    LevelDev lv[EFX_MAX_LEVELS];
};

struct __attribute__((aligned(64))) TileHdr {
    uint32_t cand_start;        // canonical rank of the tile's first corner in its level == where its corners (cand) and survivors
                                // (surv) start in the level's arrays (harris_kernel; may exceed the level's cap: then none exist)
    uint32_t surv_count;        // nms_kernel
    uint32_t out_off;           // (unused since round 6: the offsets are a compact array, DetectLaunch::nsel)
    uint32_t pad0;
    uint16_t cell_off[EFX_CELLS_PER_TILE + 1];   // start of each cell's corners inside the tile list (fast_kernel)
    uint16_t pad[7];
};
static_assert(sizeof(TileHdr) == 64, "TileHdr must be 64 bytes");

// Per tile row of a level: the FAST corners and the NMS survivors of its tiles, summed by fire-and-forget atomics (fast_kernel,
// nms_kernel).  One 128-B line per row: atomics on one line serialise at ~11.5 ns each, returning or not
// (tools/microbench/atomic_ff.cpp), so a row's ~120 updates cost ~1.4 us spread over the kernel, a level's 8 000 on one line 94 us.
// A tile's canonical rank = the sums of the rows above it + the counts of the tiles left of it (harris_kernel): no scan pass.
#ifndef EFX_ROWCTR_PAD
#define EFX_ROWCTR_PAD 30
#endif
struct __attribute__((aligned(8))) RowCtr { int cand; int surv; int pad[EFX_ROWCTR_PAD]; };
// select_kernel's per-level state: written by the level's leader workgroup, read by the counting workgroups
struct __attribute__((aligned(32))) SelLevel {
    // published by the leader with two device-scope stores, [1] first: [0] = ready << 63 | (bin + 1) << 32 | keys in the bin, where
    // bin is the threshold bin of the key histogram (-1: every survivor is selected, EFX_HIST_BINS: none); [1] = remaining << 32 |
    // min(n, quota): the largest `remaining` keys of the bin are selected (== keys in the bin: all of them, no ranking), and the
    // level's share of N
    unsigned long long pub[2];
    int list_n;                 // keys of the bin appended to the level's list so far
    int done;                   // counting workgroups of the level that have finished
    int pad[2];
};
struct Summary {                // what the host mirror receives (written by select_kernel)
    int cand[EFX_MAX_LEVELS];
    int surv[EFX_MAX_LEVELS];
    int kept[EFX_MAX_LEVELS];           // after quota
    int n_out;                          // N written to the caller
    int dbg;
    int overflow;                       // arena contents failed their range checks ("lost stores"): the frame is
                                        // void (N = 0).  Since round 6 NO frame content can raise it: the arenas hold the reference's own
                                        // 10 % cap (.cpp:252) and nothing is allocated
};
struct Counters {               // zeroed at the start of every frame (efx_zero_counters)
    int level_out_base[EFX_MAX_LEVELS + 1];
    unsigned long long thresh[EFX_MAX_LEVELS];   // selection threshold key per level
    SelLevel sel[EFX_MAX_LEVELS];
    Summary sum;
};

// Frame-batched launches (round 6; SURVEY 8b "batched variants (..., nframes)"): every kernel of the detect path takes the tiles /
// strips / keypoints of `nframes` same-sized frames in one launch, frame = blockIdx.y.  The frames share the level table, the tile
// words and the resize plans (pure geometry); everything that holds pixels or results exists once per frame, `stride` apart
// inside the context's buffers.  The caller's per-frame pointers travel as by-value kernel arguments (no table upload).
struct FrameIn { const uint8_t* img0[EFX_MAX_BATCH]; };                       // level 0 of every frame (same rows, cols, pitch)
struct FrameOut { uint8_t* kps[EFX_MAX_BATCH]; int* count[EFX_MAX_BATCH]; };  // 5 x capacity matrices (may be null), device N
struct FrameDesc { uint8_t* desc[EFX_MAX_BATCH]; };                           // descriptor matrices
struct FrameStride {            // distance between two frames' copies, in ELEMENTS of the buffer's type (0 is fine for one frame)
    size_t pyramid;             // bytes
    size_t hdr;                 // TileHdr
    size_t cand;                // Corner records (cand and surv are indexed alike)
    size_t slots;               // bytes (EFX_SLOT_BYTES per tile) == 2 x the tile-count words (uint16_t) and 4 x the selected-count words
    size_t rows;                // RowCtr
    size_t hist;                // ints (EFX_HIST_BINS per level)
    size_t list;                // unsigned long long (EFX_SEL_LIST_CAP per level)
    size_t cmax;                // Corner
    size_t kp;                  // float4 / int (kp4, kp_level), and Affine records (bad_affine)
    size_t blurred;             // bytes
};

#if defined(__HIPCC__)
// Keypoint i of a describe call: from the float4 list, or straight from the caller's 5 x n matrix with the rule of
// convertKeypointsKernel (cuda_efficient_features.cu:250-263): x, y from the packed shorts, size forced to 31, angle row
__device__ __forceinline__ float4 efx_load_keypoint(const float4* __restrict__ kp4, const uint8_t* __restrict__ kps5, size_t kps5_pitch, int i)
{
    if (!kps5) return kp4[i];
    const uint32_t loc = *reinterpret_cast<const uint32_t*>(kps5 + 4 * (size_t)i);
    const float ang = *reinterpret_cast<const float*>(kps5 + 2 * kps5_pitch + 4 * (size_t)i);
    return make_float4((float)(short)(loc & 0xffff), (float)(short)(loc >> 16), 31.f, ang);
}

// the frame is void: arena contents that fail their range checks (never a property of the frame itself, see Summary::overflow)
__device__ __forceinline__ void efx_raise_overflow(const LevelTable*, Counters* cnt) { cnt->sum.overflow = 1; }
// word of key-histogram bin b: neighbouring bins lie in neighbouring 128-byte LINES (bin b in line b mod BINS / 32, word b / (BINS /
// 32) of it).  The survivors' responses of one frame crowd into a few hundred neighbouring bins -- ~100 on frames with the
// statistics of photographs -- and device-scope atomics on one line serialise at ~11 ns each: measured for 100 hit bins and 100 000
// updates (tools/microbench/hist_layout.cpp), 45 us with neighbouring bins 2 KB apart (32 lines hit), 14 us this way
__host__ __device__ inline uint32_t efx_hist_word(uint32_t b) { return ((b & (EFX_HIST_BINS / 32 - 1)) << 5) | (b >> (EFX_HIST_BITS - 5)); }
#endif

// One packed word per tile behind the level table: level | tx << 5 | ty << 15.  The level field must hold EFX_MAX_LEVELS
// values; tx, ty < 32768 / 64 = 512 (images are at most 32767 px per side).
#define EFX_TILE_LEVEL_BITS 5
static_assert((1 << EFX_TILE_LEVEL_BITS) >= EFX_MAX_LEVELS, "tile word: level field too narrow for EFX_MAX_LEVELS");
__host__ __device__ inline uint32_t efx_pack_tile(uint32_t level, uint32_t tx, uint32_t ty)
{
    return level | (tx << EFX_TILE_LEVEL_BITS) | (ty << (EFX_TILE_LEVEL_BITS + 10));
}

// one FAST corner / survivor
struct __attribute__((aligned(8))) Corner {
    uint32_t xy;       // x | y << 16, level coordinates
    float resp;        // Harris response (spec S4)
};
// In the per-cell maxima table (Corner* cmax) bit 31 of xy (y < 2^15) says that another corner of the cell has the same
// response as the stored one (equal responses suppress each other: the NMS must then walk the cell's list)
#define EFX_CMAX_TIE 0x80000000u

// 64-bit selection key: response descending, then raster (y, x) ascending (spec S3).
__host__ __device__ inline unsigned long long efx_select_key(uint32_t xy, float resp)
{
    union { float f; uint32_t u; } c;
    c.f = resp;
    uint32_t u = c.u;
    if (u == 0x80000000u) u = 0;                        // -0 == +0
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // order-preserving float -> uint
    const uint32_t x = xy & 0xffffu, y = xy >> 16;
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - ((y << 16) | x));
}

struct BadParamsDev {           // per-context copy of the learned tables (no process-global constants)
    int nbits;
    float reach;                // max over boxes of (centre distance from (16,16) + radius), patch units
    uint2 box[512];             // {x1 | x2<<5 | y1<<10 | y2<<15 | radius<<20, threshold bits}: 8 B per box pair
    // Detector keypoints (size 31, describer scale s.t. the window is 48 x 48): everything about a box pair that does not
    // depend on the keypoint, worked out once on the host with the float expressions of bad.cpp:151-155,393
    //   .x = x1 | y1 << 8 | x2 << 16 | y2 << 24          (v_cvt_f32_ubyteN)
    //   .y = -2 r' (50 + 1)   r' = (int)(s r + 0.5f), the scaled radius: byte offset of the box's top-left entry from its centre's
    //   .z = 2 side | (2 * 50 side) << 16               side = 2 r' + 1; byte strides in the integral (u16 entries, 50 per row)
    //   .w = bits of thr * (float)(side * side)
    uint4 ubox[512];
    float ubox_s;               // the s the table was built for (scale_factor * 31 / 32)
    int ubox_max_side;          // largest box edge 2 r' + 1 of the table (<= 16: every box sum fits 16 bits, bad_raw_kernel)
};

#if defined(__HIPCC__)
// Inclusive scan across the 64 lanes of a wave with DPP moves only (6 VALU ops, no LDS crossbar):
// row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast15 / row_bcast31 across rows (gfx9 / CDNA DPP controls).
__device__ __forceinline__ int efx_wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);     // row_bcast:31 -> rows 2, 3
    return v;
}
#endif

// Investigation knobs.  The stage knobs (EFX_DEBUG, EFX_DEBUG_HS: kernels stop early / skip stages, i.e. WRONG results)
// exist only in builds made with -DEFX_DEBUG_BUILD (make EXTRA=-DEFX_DEBUG_BUILD); a production build ignores the
// variables and the kernels carry no trace of them.  The variant knobs (EFX_NO_TOWER, EFX_NO_RESIZE_STREAM: pick another,
// bit-identical pyramid kernel; used by the parity tests) are read ONCE, when a context is created.
#ifdef EFX_DEBUG_BUILD
#define EFX_DBG(v) (v)
#else
#define EFX_DBG(v) 0
#endif
struct EfxKnobs { int dbg, dbg_hs, no_tower, no_resize_stream, no_level_blur, blur_fork, no_resize_rows, no_batch, pack; long long tower_max_px, blur_fork_min_px; };
EfxKnobs efx_read_knobs();      // efx_api.cpp

// ---- launchers (host side, defined in the .hip files) ----
#ifdef __HIPCC__
// XCD-aware block -> tile mapping: workgroup b runs on XCD b % 8 (observed dispatch order), so give every
// XCD one contiguous run of tiles; neighbouring tiles then share halo lines in the same L2.
__device__ __forceinline__ int xcd_chunked(int bid, int n)
{
    const int q = n / EFX_NXCD, r = n % EFX_NXCD;
    const int xcd = bid % EFX_NXCD, idx = bid / EFX_NXCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// The same idea for work whose cost varies along the index (tiles ordered densest level first): runs of
// EFX_XCD_RUN consecutive items go round-robin over the XCDs, so every XCD sees the same mix of levels while
// neighbouring tiles still share an L2.
#ifndef EFX_XCD_RUN
#define EFX_XCD_RUN 32
#endif
__device__ __forceinline__ int xcd_interleaved(int bid, int n)
{
    const int S = EFX_NXCD * EFX_XCD_RUN;
    if (bid >= (n / S) * S) return bid;
    const int xcd = bid % EFX_NXCD, i = bid / EFX_NXCD;
    return ((i / EFX_XCD_RUN) * EFX_NXCD + xcd) * EFX_XCD_RUN + (i % EFX_XCD_RUN);
}
#endif

// Optional per-launch timing (efx_profile_*): one HIP-event pair around a launch, tagged with a code:
// 0 fast_kernel, 1 harris_kernel, 2 nms_kernel, 3 select+emit+angle, 10 describe (BAD / HashSIFT), 100+s resize of level s+1
struct ProfRec {
    hipEvent_t* start; hipEvent_t* stop; int* code; int* count; int capacity;
    unsigned skip;              // bit g set: launches of group g are not timed (0 fast, 1 harris, 2 nms, 3 select+emit+angle,
                                // 4 describe, 5 pyramid) -- every event pair costs a few microseconds of stream idle time
    static int group_of(int c) { return c >= 100 ? 5 : (c == 10 || c == 11 ? 4 : c); }     // 10 describe, 11 blur_levels_kernel
    bool begin(int c, hipStream_t st) const
    {
        if (!count || *count >= capacity || ((skip >> group_of(c)) & 1u)) return false;
        (void)hipEventRecord(start[*count], st);
        return true;
    }
    void end(bool on, int c, hipStream_t st) const
    {
        if (!on) return;
        (void)hipEventRecord(stop[*count], st);
        code[*count] = c;
        ++*count;
    }
};

// Spec S5 weights of destination index o (source position s = (float)o * f, rounded): of the upper neighbour i2 = i1 + 1 and of
// the lower one i1.  Default: separate subtractions of the rounded product.  -DEFX_S5_FUSED_WEIGHTS=1 (library AND oracle): the
// other thing nvcc's contraction could have made of opencv_contrib's `x2 - src_x` / `src_x - x1` (undecidable in
// this image: no CUDA build, no cv::cuda::resize dump) -- every pyramid kernel and plan takes its weights from here.
#ifndef EFX_S5_FUSED_WEIGHTS
#define EFX_S5_FUSED_WEIGHTS 0
#endif
__host__ __device__ inline float efx_s5_w_hi(int o, float f, float s, int i2)
{
#if EFX_S5_FUSED_WEIGHTS
    return __builtin_fmaf(-(float)o, f, (float)i2);
#else
    (void)o; (void)f; return (float)i2 - s;
#endif
}
__host__ __device__ inline float efx_s5_w_lo(int o, float f, float s, int i1)
{
#if EFX_S5_FUSED_WEIGHTS
    return __builtin_fmaf((float)o, f, -(float)i1);
#else
    (void)o; (void)f; return s - (float)i1;
#endif
}

// Resize plan of one destination level (resize_stream_kernel): everything about a tile / column / row that does not
// depend on the pixels, computed once per geometry on the host with the kernel's own float expressions (spec S5).
//   x table  3 x W words (W = tiles_x * 64): source column x1 | weight of x1 | weight of x1 + 1, per destination column
//   y table  int4 per destination row (tiles_y * 64): source rows y1, min(y1 + 1, rows - 1) | weights as float bits
//   tile table  int4 per tile: sy0 | ax0 | ndw + (nrow << 8) + (touches the last source column << 16) | tx + (ty << 16)
struct ResizePlanLevel { unsigned x_off, y_off, t_off; int W; };      // byte offsets into DetectLaunch::rplan; W == 0: no plan

// Plan of one launch of resize_rows_kernel (round 5, detect_kernels.hip): level s -> s + 1 .. s + nlev by waves that walk down
// strips of `own` columns of level s + 1.  Offsets are bytes into DetectLaunch::rplan; level index k = 0 is level s + 1.
//   x[k]      3 x W[k] words: source column x1 | weight of x1 | weight of x1 + 1 per destination column (clamped beyond the last one)
//   y[k]      int4 per destination row, padded by 64 rows: y1, min(y1 + 1, rows - 1) | the two weights as float bits
//   strips    RW_MAXLEV x int4 per strip: [first source column staged (4-aligned) | dwords of a source row it needs | 0 | 0], then per level
//             k >= 1: [first group of four columns the strip computes (= owns) | groups it owns | groups it computes (halo for level k + 1) | 0]
//   chunks    (1 + RW_MAXLEV + 2) x int4 per chunk of rows: [first / last source row | source rows padded to the kernel's slot count | 0],
//             per level k: [first row computed (= owned) | end of the rows it stores | 0 | 0], then RW_MAXLEV 64-bit masks: bit i of mask k =
//             source row i of the chunk completes a row of level k (and, for k >= 1, mask k - 1 has the bit too)
#define RW_MAXLEV 4
#define RW_STRIP_INT4 RW_MAXLEV
#define RW_CHUNK_INT4 (1 + RW_MAXLEV + RW_MAXLEV / 2)
#ifndef RW_D
#define RW_D 6                  // source rows in flight per wave = LDS slots = the unroll of the kernel's row loop (even); chunks are padded to it
#endif
struct RowsPlanLaunch {
    int nlev;                   // 0: no plan for this source level
    int own;                    // columns of level s + 1 a strip owns (a multiple of 4, <= 256: the other lanes compute halo columns)
    int nstrips, nchunks;
    int W[RW_MAXLEV];
    unsigned x_off[RW_MAXLEV], y_off[RW_MAXLEV], strip_off, chunk_off;
};

struct DetectLaunch {
    int nframes;                // frames in this launch (1 .. EFX_MAX_BATCH); frame 0's pointers are also the scalar fields below
    FrameIn in; FrameOut out; FrameStride fs;
    const unsigned char* rplan; const ResizePlanLevel* rplan_lv;      // device blob, host index by destination level
    const RowsPlanLaunch* rows_plan;                                  // host array indexed by SOURCE level (null: none)
    const uint8_t* img0;        // level 0 (caller's image)
    int pitch0;
    uint8_t* pyramid;           // levels >= 1
    const LevelTable* d_table;  // device copy
    const LevelTable* h_table;  // host copy (same contents)
    TileHdr* hdr;
    Corner* cand;
    unsigned char* slots;       // EFX_SLOT_BYTES per tile: corner list or bitmap (fast_kernel -> harris_kernel)
    uint16_t* tcount;           // FAST corners per tile, compact (the row part of a tile's canonical rank)
    uint32_t* nsel;             // selected survivors per tile (select_kernel's counting pass), then -- same words -- the output index of the
                                // tile's first selected survivor (its scan -> emit_kernel)
    RowCtr* rows;               // per tile row: corner / survivor sums
    int* hist;                  // key histogram of the survivors, EFX_HIST_BINS per level (nms_kernel adds, select_kernel's leaders read and clear)
    unsigned long long* sel_list;   // keys of the threshold's bin, EFX_SEL_LIST_CAP per level
    Corner* cmax;               // strongest corner of every 16x16 cell (quick test of the NMS)
    Corner* surv;
    Counters* counters;
    int threshold;
    int nonmax_radius;
    int first_level;
    EfxKnobs knobs;             // read at context creation (dbg: EFX_DEBUG_BUILD builds only)
    const uint8_t* mask; int mask_pitch;   // optional level-0 mask (spec S12), null = none
    int pyramid_only;           // 1: build the pyramid and stop (detectAndCompute with provided keypoints)
    int pack_harris;            // harris_kernel takes four tiles per wave (sparse frames; bit-identical to the other form)
    // outputs
    void* d_keypoints; size_t kps_pitch; int capacity; int* d_count;
    float4* kp4; int* kp_level;
    // BAD describer behind this detect call: angle_kernel also writes the per-keypoint Affine records (bad_affine.h)
    void* bad_affine; float bad_scale, bad_reach; int bad_smax, bad_sfixed;
    // ... on BLURRED copies of the levels (round 4, bad_kernel.hip: blur_levels_kernel): level 0 at `blurred` with pitch
    // blur0_pitch, level l >= 1 at blurred + blur_levels_off + its pyramid offset with the level's pitch; null: the describers
    // blur per keypoint.  The records' image pointers then refer to the blurred levels
    uint8_t* blurred; int blur0_pitch; size_t blur_levels_off;
    // where the level blur runs: 0 on the call's stream right behind the pyramid; 1 / 2 / 3 on the context's side stream, forked
    // behind the pyramid / harris_kernel / nms_kernel and joined at the end of the detect launches (it needs the pyramid only,
    // and select / emit / angle leave most of the chip idle)
    int blur_fork; hipStream_t side; hipEvent_t ev_fork, ev_join;
    ProfRec prof;                                          // optional HIP-event pairs around the launches
};

hipError_t efx_launch_detect(const DetectLaunch& a, hipStream_t stream);
hipError_t efx_debug_rerun_stages(const DetectLaunch& a, int stages, hipStream_t stream);      // investigation only

struct DescribeLaunch {
    const uint8_t* img0; int pitch0; int rows0, cols0;     // image for level index 0 / single-image mode
    const uint8_t* pyramid; const LevelTable* d_table;     // null in single-image mode
    const float4* kp4; const int* kp_level;                // kp_level null -> all keypoints on img0
    const uint8_t* kps5; size_t kps5_pitch;                // non-null: the keypoints are the caller's 5 x n matrix (computeAsync): the
                                                           // record kernels read it directly (convertKeypointsKernel's rule: size 31), kp4 is unused
    const int* d_count;                                    // device N (null -> use n)
    int n;                                                 // grid size (upper bound of N)
    int blur;                                              // 1: 7x7 sigma-2 Gaussian first (detectAndCompute)
    float scale_factor;                                    // BAD scaleFactor / HashSIFT croppingScale
    float max_size;                                        // upper bound of keypoint size (LDS window)
    int uniform_size;                                      // 1: every keypoint has size == max_size (detector output)
    uint8_t* desc; size_t desc_pitch;
    void* bad_affine;                                      // BAD scratch: n x 80 bytes (per-keypoint affine map + window geometry)
    int nbits;                                             // descriptor bits (256 / 512)
    int bad_det_tables;                                    // 1: BadParamsDev::ubox was built for this describer scale and size 31; 2: and no box edge exceeds 16
    int bad_no_raw;                                        // EFX_BAD_NO_RAW (variant knob, read when the describer is created; parity tests):
                                                           // computeAsync through the generic one-workgroup-per-keypoint kernel
    int affine_ready;                                      // bad_affine already holds this call's records (written by angle_kernel)
    int level_blurred;                                     // ... and they point at blurred level images: describe without a blur (bad_raw_kernel)
    int dbg_hs;                                            // EFX_DEBUG_HS (EFX_DEBUG_BUILD builds only)
    ProfRec prof;
    // batched describe behind a batched detect (bad_raw_kernel on blurred levels only): nframes > 1, frame = blockIdx.y; the
    // records of frame f start at bad_affine + f * aff_stride, its count is counts.count[f], its descriptors descs.desc[f]
    int nframes; size_t aff_stride; FrameOut counts; FrameDesc descs;
    size_t frame_affine_off;                               // a single frame of a batch described on its own: its records start here (Affine entries)
    // batched HashSIFT behind a batched detect: frame f's keypoints at kp4 / kp_level + f * kp_stride, its pyramid at pyramid +
    // f * pyr_stride, its level 0 at imgs.img0[f]; counts / descs as above
    size_t kp_stride, pyr_stride; FrameIn imgs;
};

hipError_t efx_launch_bad(const DescribeLaunch& a, const BadParamsDev* d_params, float reach, hipStream_t stream);
hipError_t efx_launch_blur_levels(const LevelTable& H, const uint8_t* img0, int pitch0, const uint8_t* pyramid, uint8_t* blurred,
                                  int blur0_pitch, size_t blur_levels_off, const ProfRec& prof_rec, hipStream_t stream,
                                  int nframes, const FrameIn& in, const FrameStride& fs);

#define EFX_HS_REC_BYTES 64
struct HashSiftDev {
    int nbits;
    const float* W;             // nbits x 132 (129 padded to 132), fp32, device; the 30x30 / 511x511 tables follow
    const uint16_t* Wb;         // 3 x nbits x 144 bf16: W split into three bf16 terms (project_sign_kernel)
    uint16_t* responses;        // scratch n x 144 bf16: the 129-vectors (integer valued, exact), K padded with zeros
    void* records;              // scratch n x EFX_HS_REC_BYTES: per-keypoint affine map + window (hs_record_kernel)
    float* dbg_responses;       // optional n x 129
    float* dbg_T;               // optional n x nbits
};
hipError_t efx_launch_hashsift(const DescribeLaunch& a, const HashSiftDev& h, hipStream_t stream);

// 5xN keypoint matrix -> float4 {x, y, 31, angle} (convertKeypointsKernel, cuda_efficient_features.cu:250-263)
hipError_t efx_launch_convert_keypoints(const void* d_keypoints, size_t kps_pitch, int n, float4* kp4, hipStream_t stream);
// 5xN keypoint matrix -> level coordinates {x / scale, y / scale, 31, angle} + level (spec S13); descriptors of
// keypoints whose octave is out of range are zeroed afterwards
hipError_t efx_launch_provided_keypoints(const LevelTable* d_table, const void* d_keypoints, size_t kps_pitch, int n, float4* kp4,
                                         int* kp_level, hipStream_t stream);
hipError_t efx_launch_zero_invalid_descriptors(const LevelTable* d_table, const void* d_keypoints, size_t kps_pitch, int n,
                                               uint8_t* desc, size_t desc_pitch, int nbytes, hipStream_t stream);
hipError_t efx_launch_copy2d(const uint8_t* src, size_t spitch, uint8_t* dst, size_t dpitch, int rows, int cols, hipStream_t stream);

void efx_gaussian_taps_host(float taps[7]);

// ICAngles of samples/hpatches_description.cpp:128-162 on n x {x, y, size, angle} keypoints (input_kernels.hip)
hipError_t efx_launch_ic_angles(const uint8_t* img, size_t pitch, int rows, int cols, float4* kp4, int n, int patch_size, hipStream_t stream);

// BGR / BGRA -> gray (input_kernels.hip, spec S11)
hipError_t efx_launch_cvt_gray(const uint8_t* src, size_t spitch, int rows, int cols, int channels, uint8_t* dst, size_t dpitch,
                               hipStream_t stream);

// brute-force Hamming matcher (match_kernels.hip); scratch: nchunks * nq * 16 bytes
// int8 matrix-core variant for large sets: scratch_x = efx_knn2_mfma_scratch() bytes (the +-1 expansions of both sets)
size_t efx_knn2_mfma_scratch(int nq, int nt, int desc_bytes);
hipError_t efx_launch_knn2_mfma(const uint8_t* query, size_t q_pitch, int nq, const uint8_t* train, size_t t_pitch, int nt,
                                int desc_bytes, void* scratch_x, void* scratch, int nchunks, int* idx, int* dist, hipStream_t stream, int fp4);
hipError_t efx_launch_knn2(const uint8_t* query, size_t q_pitch, int nq, const uint8_t* train, size_t t_pitch, int nt,
                           int desc_bytes, void* scratch, int nchunks, int* idx, int* dist, hipStream_t stream);
hipError_t efx_launch_crosscheck(const int* q2t, const int* t2q, int nq, int* match, hipStream_t stream);
