// Internal declarations shared by the HIP translation units of libgsplat_hip.so.
// gfx950 (MI355X) only: wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

constexpr int WAVE = 64;
constexpr uint32_t DEPTH_RANGE = 65536u;   // wasm/wasm.cpp:33
constexpr int KEY_BITS = 17;               // keys are in [0, 65536]
constexpr int RADIX_LO_BITS = 8;           // pass 1 digit: key & 0xff
constexpr int RADIX_HI_BITS = 9;           // pass 2 digit: key >> 8  (0..256)
constexpr int RADIX_LO_BINS = 1 << RADIX_LO_BITS;
constexpr int RADIX_HI_BINS = 1 << RADIX_HI_BITS;

constexpr int TILE = 16;                   // pixels per tile side: one wave per tile, 4 px per lane
constexpr int BIN_TILES = 2;               // tiles per coarse-bin side
constexpr int BIN_PX = TILE * BIN_TILES;   // 32 px: one workgroup (4 waves) per bin

// Camera constants passed by value (kernarg -> SGPRs).
struct CamParams {
    float view[16];
    float proj[16];
    float vp2, vp6, vp10;   // row 2 of viewProj: wasm/wasm.cpp:18-20
    float fx, fy;
    int32_t W, H;
    int32_t sh_on;          // scene carries SH textures
    int32_t band[3];        // Scene.bandsIndices
    int32_t band_px0, band_px1;  // pixel columns [x0, x1) this context composites (multi-GPU band; whole image otherwise)
    int32_t use_fade;       // u_useDepthFade
    float fade;             // u_depthFade
};

// 32-byte projected record consumed by the tile compositor (image coordinates, row 0 = top).
//   vPosition at pixel centre p is (dot(p-c,u), dot(p-c,w)); coverage |vPosition|^2 <= 4.
struct __attribute__((aligned(16))) Record {
    float cx, cy, ux, uy;
    float wx, wy, la;       // la = log2(opacity)
    uint32_t rgb8;          // r | g<<8 | b<<16; or RGB8_IN_SHCOL: the colour is the float triple shcol[splat]
};
constexpr uint32_t RGB8_IN_SHCOL = 0x01000000u;
static_assert(sizeof(Record) == 32, "record must be 32 bytes");

// Packed inclusive pixel bounding box: x = x0 | x1<<16, y = y0 | y1<<16; invisible when x0 > x1.
constexpr uint32_t BBOX_INVISIBLE_X = 1u;  // x0 = 1, x1 = 0
constexpr uint32_t BBOX_INVISIBLE_Y = 1u;

struct SceneSoA {
    const float *px, *py, *pz;
    const uint32_t *cov0, *cov1, *cov2, *rgba;
    const uint32_t *sh_r, *sh_g, *sh_b;  // 8 u32 per SH-carrying splat and channel (null without SH)
    float4* shcol;                       // out: evaluated SH colour per splat (null without SH)
};

// mutable device scene for the on-device build / transforms (k_scene.hip)
struct SceneDev {
    float *px, *py, *pz;
    uint32_t *cov0, *cov1, *cov2, *rgba;
    float4* rot;   // (w, x, y, z) as Scene._rotations stores them
    float4* scl;   // Scene._scales
};
void launch_build_scene(const uint8_t* rows, uint32_t n, const SceneDev& sc, hipStream_t s);
void launch_scene_translate(uint32_t n, const SceneDev& sc, const double* t, hipStream_t s);
void launch_scene_rotate(uint32_t n, const SceneDev& sc, const double* q_xyzw, hipStream_t s);
void launch_scene_scale(uint32_t n, const SceneDev& sc, const double* sv, hipStream_t s);
void launch_scene_limit_box(uint32_t n, const SceneDev& src, const SceneDev& dst, const double* box, uint32_t* block_count,
                            uint32_t* total, hipStream_t s);

// ---- launchers (each enqueues on `s`; none synchronises) ----
void launch_repack_scene(const uint32_t* data, const float* positions, uint32_t n, float* px, float* py, float* pz,
                         uint32_t* cov0, uint32_t* cov1, uint32_t* cov2, uint32_t* rgba, uint32_t* mismatch, hipStream_t s);

void launch_repack_positions(const float* positions, uint32_t n, float* px, float* py, float* pz, hipStream_t s);

// frame_words: the context's per-frame device words, [0] = minDepth, [1] = maxDepth, the rest zero at frame start
void launch_begin_frame(const CamParams& cam, CamParams* dst, uint32_t* frame_words, uint32_t nwords, hipStream_t s);
void launch_project_key(const SceneSoA& sc, uint32_t n, const CamParams* cam_dev, int do_project, int32_t* depth,
                        int2* blk_minmax /* ceil(n/256) */, int32_t* minmax, Record* rec, uint2* bbox, hipStream_t s);

// radix sort of the 17-bit keys; see k_sort.hip
struct SortBuffers {
    const int32_t* depth;      // n
    const int32_t* minmax;     // 2
    uint32_t* keys;            // n   17-bit keys, original order
    uint32_t* keys_tmp;        // n   keys after pass 1
    uint32_t* idx_tmp;         // n   indices after pass 1
    uint32_t* depth_index;     // n   result
    uint32_t* block_hist;      // nblocks * RADIX_HI_BINS
    uint32_t* digit_total;     // RADIX_LO_BINS + RADIX_HI_BINS
    const uint2* cull_bbox;    // band mode: boxes of the projection (empty box = absent from the sort); null = sort all n
    uint32_t* count;           // out: keys the first pass kept (n, or the band's survivors) = entries of depth_index
    uint32_t keys_per_block;
    uint32_t nblocks;
};
void launch_sort(const SortBuffers& b, uint32_t n, hipStream_t s);
// column scan (k_sort.hip), shared with the binning
void launch_column_scan(uint32_t* table, uint32_t* total, int ncols, uint32_t nrows, hipStream_t s);

struct BinGrid {
    int32_t nbx, nby;          // bins across / down the whole image
    int32_t bx_lo, bx_hi;      // bin columns of this context's band [lo, hi)
    int32_t W, H;
};
struct BinBuffers {
    const uint32_t* depth_index; // *count entries
    const uint32_t* count;       // ranks to bin (SortBuffers::count)
    const uint2* bbox;           // n
    uint32_t* table;             // nblocks * nbins  (counts, then per-workgroup offsets inside each bin)
    uint2* blk_counts;           // nblocks: (visible splats, 16x16 tile overlaps) per counting workgroup
    uint32_t* rects;             // n: packed bin rectangle of every rank (count pass -> scatter pass)
    uint32_t* bin_total;         // nbins (zeroed by the caller when n == 0)
    uint32_t* bin_start;         // nbins + 1
    uint32_t* seg_start;         // nbins + 1: first compositor work item of each bin; [nbins] = item count
    uint32_t* items;             // max_items: bin | segment << 16
    uint32_t* list;              // capacity entries (splat indices, depth order inside each bin)
    uint32_t* overflow;          // bit 0: list too small, bit 1: item table too small
    uint64_t* visible;           // V counter
    uint64_t* tile_entries;      // D counter (16x16 tiles overlapped by visible bboxes)
    uint64_t* accum;             // [8] running sums over frames: visible, bin entries, tile entries, frames; [4] = entries the last frame needs;
                                 // sticky: [5] = frames that did not fit (never composited), [6] / [7] = most entries / items one of them needed
    uint64_t* mailbox;           // host-mapped word: accum[5] is stored here whenever it changes
    uint64_t* report;            // [6] per-frame copy for the host: accum[0..4] after this frame, [5] = this frame's bin entries
    uint32_t capacity;
    uint32_t max_items;
    uint32_t seg_len;            // minimum list entries per compositor work item (multiple of 256); k_bin_finalize
                                 // raises it for long lists and publishes the frame's value in *seg_len_dev
    uint32_t* seg_len_dev;
    int32_t items_by_size;       // order the bins' last segments by size class (one frame at a time) or leave them in raster order
    uint32_t* queue;             // the compositor's work-item counter, set to queue_start (= its grid size) by k_bin_finalize
    uint32_t queue_start;
    uint32_t seg_target_items;   // full segments the frame should be cut into at least (long lists -> longer segments)
    uint32_t nblocks;
};
void launch_bin(const BinBuffers& b, const BinGrid& g, uint32_t n, hipStream_t s);

struct BlendBuffers {
    const uint32_t* items;      // work items: bin | segment << 16
    const uint32_t* seg_start;  // nbins + 1
    const uint32_t* bin_start;  // nbins + 1
    const uint32_t* list;
    const Record* rec;
    const uint2* bbox;
    const float4* shcol;        // evaluated SH colours (may be null)
    float4* fb;
    float4* partial;            // max_items * 1024 float4: per-segment (colour, transmittance), slot = seg_start[bin] + segment
    uint32_t* queue;            // device-wide work-item counter, zero at frame start
    uint32_t seg_len;           // host's minimum; >= 0x40000000: one item per bin (early termination mode)
    const uint32_t* seg_len_dev; // the frame's segment length (k_bin_finalize)
    uint32_t grid;              // persistent workgroups launched
    uint32_t capacity;          // entries the list can hold
    uint32_t nsplats;
};
// `between` (may be null) is recorded after k_blend and before k_combine
void launch_blend(const BlendBuffers& b, const BinGrid& g, float early_out_eps, hipStream_t s, hipEvent_t between);
void launch_clear_fb(float4* fb, int32_t W, int32_t H, hipStream_t s);
void launch_to_rgba8(const float4* fb, uint32_t* out, uint32_t npix, hipStream_t s);

// multi-GPU exchange helpers (RGBA8 slabs of the all-gather)
constexpr int MAX_SLABS = 16;
struct SlabEdges { int32_t x0[MAX_SLABS], x1[MAX_SLABS]; };
void launch_pack_band_rgba8(const float4* fb, uint32_t* slab, int W, int H, int x0, int x1, int slab_w, hipStream_t s);
void launch_unpack_slabs_rgba8(const uint32_t* gathered, uint32_t* image, int W, int H, int slab_w, int world,
                               const SlabEdges& e, hipStream_t s);

}  // namespace gsr
