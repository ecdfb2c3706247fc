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

// Bin rectangle of a splat's pixel box inside a context's band of bin columns [bx_lo, bx_hi), packed in 4 bytes:
// x0 | x1 << 8 | y0 << 16 | y1 << 24 (inclusive bin coordinates, x relative to the band; at most 256 bins per axis);
// 1 = nothing to draw.  Written once per splat by k_project_key; k_bin_count gathers them into depth order (4-byte
// gathers; carrying them through the last radix pass instead moved the same cost into that kernel: measured, dropped).
constexpr uint32_t RECT_NONE = 1u;
__host__ __device__ inline uint32_t pack_bin_rect(uint32_t bbx, uint32_t bby, int bx_lo, int bx_hi)
{
    const int px0 = (int)(bbx & 0xffffu), px1 = (int)(bbx >> 16), py0 = (int)(bby & 0xffffu), py1 = (int)(bby >> 16);
    if (px0 > px1) return RECT_NONE;
    const int x0 = (px0 / BIN_PX > bx_lo ? px0 / BIN_PX : bx_lo) - bx_lo;
    const int x1 = (px1 / BIN_PX < bx_hi - 1 ? px1 / BIN_PX : bx_hi - 1) - bx_lo;
    if (x0 > x1) return RECT_NONE;
    return (uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)(py0 / BIN_PX) << 16) | ((uint32_t)(py1 / BIN_PX) << 24);
}

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
void launch_begin_frame(const CamParams& cam, CamParams* dst, uint32_t* frame_words, uint32_t nwords, int32_t* slots, hipStream_t s);
// Frame-wide reductions of k_project_key -- depth min / max over ALL splats (wasm.cpp:14-31), visible splats and the
// 16x16 tiles their boxes overlap (V and D of the byte model) -- go through FRAME_SLOTS accumulators, one 128-byte line
// each: a workgroup folds its values into slot (blockIdx & 63) with four atomics.  ~60 workgroups share a slot over the
// kernel's ~20 us, far from the ~90 same-address atomics per microsecond at which one word saturates (with ONE pair of
// words for 4000 workgroups the kernel stayed alive ~17 us after its last store).  The consumers fold the 64 slots
// themselves (k_quantise_hist, k_bin_finalize): no reduction kernel.
// Blocks are dealt round-robin over the 8 XCDs (observed, used for speed only: MI355X_MICROARCH.md, Workgroup dispatch).
// The kernels that APPEND runs to many output streams (radix digits, bin lists) want neighbouring input blocks on one XCD,
// so that the runs they append to a stream meet in one L2 instead of leaving two XCDs as partial lines: inside every
// group of 64 consecutive blocks, XCD k takes the 8 consecutive blocks [8k, 8k+8).  Groups, not one contiguous range per
// XCD: the input is depth-ordered and the front blocks hold the nearest, largest splats -- a contiguous range per XCD
// gave one XCD all the heavy blocks (C4 binning +17 %).  Bijective for any grid size (the tail keeps its order).
// Measured: k_bin_scatter 330 -> 286 us at 20 M splats, binning -4.5 % on C3 and C4; first radix pass 98 -> 93 us at 20 M.
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t xcd_group_remap(uint32_t bid, uint32_t nwg)
{
    constexpr uint32_t RUN = 8u, G = 8u * RUN;   // blocks per XCD in a group, blocks per group (runs of 16 and 32 measured the same)
    const uint32_t full = nwg - nwg % G;
    if (bid >= full) return bid;
    const uint32_t in = bid % G;
    return (bid - in) + (in % 8u) * RUN + in / 8u;
}
#endif
// -DGSR_BOUNDS (diagnostic build, scripts/build_exp.sh bounds "-DGSR_BOUNDS"; tests/test_gpu_bounds.py runs the parity tests'
// frames on it): every index the kernels derive from device data -- list entries, splat indices, slots, LDS cells -- is
// checked against the extent of what it indexes; a violation is counted per site in the translation unit's g_bounds[]
// (gsr_debug_bounds_*), never trapped: a faulting kernel can take the whole node down.  The stand-in for the GPU
// sanitizers this pool does not offer.  Nothing of it is compiled into the shipped library.
#ifdef GSR_BOUNDS
#define GSR_BOUNDS_DECL(name) __device__ unsigned int g_bounds_##name[8];                                             \
    extern "C" int gsr_debug_bounds_##name(unsigned int* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds_##name), sizeof g_bounds_##name) == hipSuccess ? 0 : -1; }
#define GSR_BOUND(name, site, idx, limit) do { if (!((unsigned long long)(idx) < (unsigned long long)(limit))) atomicAdd(&g_bounds_##name[site], 1u); } while (0)
#else
#define GSR_BOUNDS_DECL(name)
#define GSR_BOUND(name, site, idx, limit) do { } while (0)
#endif

// Compositor work items: four words, (bin | segment << 16, first list entry, end, first partial slot of the bin | its segments << 25).
constexpr uint32_t PROJ_THREADS = 256;
constexpr int FRAME_SLOTS = 64;
constexpr int FRAME_SLOT_WORDS = 32;   // words per slot (one 128-byte line): [0] min depth, [1] max depth, [2] visible, [3] tiles,
                                       // [4] sum of opacity byte x tiles / 16 (the frame's optical depth over the splats' boxes, k_bin_finalize),
                                       // [5] sum of opacity byte x footprint pixels / 256 (the optical mass the splats really carry)
// the depth key and the min / max alone (sort-only frames): camera by value, its own frame slots, the next frame's reset
void launch_depth_key(const SceneSoA& sc, uint32_t n, const CamParams& cam, int32_t* depth, int32_t* slots, int32_t* slots_next, hipStream_t s);
// The projection kernel's launch as a value: its arguments and the pointer array hipLaunchKernel / a graph kernel node take.
// (One struct for both, so that the graph replay rewrites exactly what a direct launch passes: the camera.)
struct ProjectLaunch {
    SceneSoA sc; uint32_t n; CamParams cam; int do_project;   // do_project: 1 = project (render frames), 2 = records and pixel boxes only (read-back)
    int32_t* depth;
    int32_t* slots;      // FRAME_SLOTS * FRAME_SLOT_WORDS, clean at the start of the frame (the finalize step resets them)
    Record* rec; uint2* bbox;   // bbox may be null (frames: nothing on the path reads the pixel boxes)
    uint32_t* rect;      // n: packed bin rectangle per splat
    uint32_t* overflow;  // the frame's overflow word, zeroed by the kernel
    uint32_t* kept;      // band mode: per 256-splat workgroup, the survivors it packed to the front of its depth slots (null: no packing)
    uint8_t* kept_lane;  // band mode: n: the lane (index & 255) a packed slot's splat came from
    void* ptrs[12];
    void bind();
};
const void* project_key_kernel();
dim3 project_key_grid(uint32_t n);
void launch_project_key(ProjectLaunch& a, hipStream_t s);

// radix sort of the 17-bit keys; see k_sort.hip
struct SortBuffers {
    const int32_t* depth;      // n
    const int32_t* slots;      // FRAME_SLOTS partial (min, max) pairs of k_project_key
    int32_t* minmax;           // 2: folded by k_quantise_hist (workgroup 0 stores it for the host / read-backs)
    uint32_t* keys;            // n   17-bit keys, original order
    uint32_t* keys_tmp;        // n   keys after pass 1
    uint32_t* idx_tmp;         // n   indices after pass 1
    uint32_t* depth_index;     // n   result
    uint32_t* block_hist;      // nblocks * RADIX_HI_BINS
    uint32_t* digit_total;     // RADIX_LO_BINS + RADIX_HI_BINS
    const uint32_t* rect;      // per splat: packed bin rectangle of the projection (carried with the keys when rects_out is set)
    const uint32_t* kept;      // band mode (koff set): per 256 splats, the survivors k_project_key packed to the front of their
    const uint8_t* kept_lane;  //   depth slots, and the lane each came from: everybody else is absent from the sort
    uint32_t* koff;            // band mode: n / 256 + 2 words: the survivors in front of every block (k_kept_scan); null otherwise
    uint32_t* count;           // out: keys the first pass kept (n, or the band's survivors) = entries of depth_index
    uint32_t keys_per_block;
    uint32_t nblocks;
    int bucket_order;          // 1: high digit first + one workgroup per bucket (4 launches); 0: LSD (6 launches); see k_sort.hip
    uint32_t* max_bucket;      // out: keys in the frame's largest high-digit bucket (host-mapped word)
    uint32_t* chunk_tab;       // bucket order: 4 x (1 + n / 4096 + 258) words: k_local_sort's work list (k_scatter's first workgroup writes it)
    uint32_t* rect_tmp;        // LSD order with rects_out: the rectangles after the first pass
    uint32_t* rects_out;       // LSD order: out: the packed bin rectangles in depth order (null: not carried; the binning gathers them)
};
void launch_sort(const SortBuffers& b, uint32_t n, hipStream_t s);
// column scan (k_sort.hip), shared with the binning
// (live: the frame holds *live keys or ranks, live_unit of them per table row: the rows behind are neither written nor scanned)
void launch_column_scan(uint32_t* table, uint32_t* total, int ncols, uint32_t nrows, hipStream_t s, const uint32_t* live = nullptr, uint32_t live_unit = 1);

struct BinGrid {
    int32_t nbx, nby;          // bins across / down the whole image
    int32_t bx_lo, bx_hi;      // bin columns of this context's band [lo, hi)
    int32_t W, H;
};
struct BinBuffers {
    const uint32_t* depth_index; // *count entries
    const uint32_t* count;       // ranks to bin (SortBuffers::count)
    uint32_t* table;             // nblocks * nbins  (counts, then per-workgroup offsets inside each bin)
    int32_t* slots;              // FRAME_SLOTS partial (visible splats, 16x16 tile overlaps) sums of k_project_key; the finalize step,
                                 // their last reader in a frame, resets them for the next one
    const uint32_t* rect_idx;    // n: packed bin rectangle of every splat (k_project_key)
    uint32_t* rects;             // n: the same in depth order (count pass -> scatter pass)
    uint32_t rects_sorted;       // 1: the sort has left them there already (SortBuffers::rects_out)
    uint32_t* bin_total;         // nbins (zeroed by the caller when n == 0)
    uint32_t* bin_start;         // nbins + 1
    uint32_t* bin_start_pre;     // nbins + 1: the same starts, computed ahead of the scatter by k_bin_starts (large-grid form)
    uint32_t rounds;             // rounds of 2048 ranks a binning workgroup takes (table rows = ceil(ranks / (2048 * rounds)); > 1 only with big)
    uint32_t big;                // large bin grids: 1 = k_bin_scatter_big (finalize as its first workgroup, rounds of 2048 ranks),
                                 // 2 = the same with rounds of 1024 ranks (half the step loops); 0 = the 64-register kernel (A/B knob)
    uint32_t* seg_start;         // nbins + 1: first compositor work item of each bin; [nbins] = item count
    uint32_t* items;             // max_items x 4 words: (bin | segment << 16, first list entry, end, first partial slot of the bin | its segments << 25)
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
    uint32_t* seg_len_dev;       // [0] the frame's segment length, [1] its number of work items, [2] reserved (0)
    int32_t items_by_size;       // order the bins' last segments by size class (one frame at a time) or leave them in raster order
    uint32_t* queue;             // the compositor's work-item counter, set to queue_start (= its grid size) by k_bin_finalize
    uint32_t queue_start;
    uint32_t seg_target_items;   // full segments the frame should be cut into at least (long lists -> longer segments)
    uint32_t nblocks;
    unsigned long long* bin_mask; // nbins: the compositor's per-bin arrival masks (one bit per segment), zeroed by the finalize step (may be null)
    int32_t long_policy;         // work items of at least seg_len_long entries: 1 always, 0 never, -1 where the frame's optical depth >= long_tau
    uint32_t seg_len_long, long_tau;
    uint32_t npix;               // pixels of this context's band (the optical depth is per pixel)
    uint32_t long_tiles_x2;      // long work items also need this many 16x16 tiles per visible splat, times two (0: no such condition)
    uint32_t long_tau_bin;       // 0: the built-in per-bin thresholds (k_bin_finalize); else: bins from this optical depth on are one item (GSR_LONG_TAU)
    uint32_t long_mass_min;      // a frame that is not dense as a whole: its saturated bins become items only from this optical mass per list entry on (pixels)
    // two-level binning (launch_bin; large bin grids): cells of 4 x 4 bins first, then the cell lists' chunks into the bins
    uint32_t two_level;          // 1: on (nblocks = workgroups of 2048 ranks, rounds = 1; table holds nblocks x (cells + 1) words)
    uint32_t* cell_list;         // 2 x capacity words: (splat index, rectangle in bins) per cell-list entry
    uint32_t* cell_total;        // cells + 1: entries per cell; [cells] = list entries the frame needs
    uint32_t* cell_start;        // cells + 1
    uint32_t* chunk_start;       // cells + 2: first chunk of each cell, the frame's chunks, the frame's need
    uint32_t* chunk_info;        // (capacity / 2048 + cells) x 4: per chunk its cell, first and end entry
    uint32_t* cell_wcnt;         // (capacity / 2048 + cells) x 64 words: per chunk, bin and wave of k_cell_scatter2 one byte: the wave's entries
    uint32_t* cell_table2;       // (capacity / 2048 + cells) x 16: per chunk and bin of its cell: entries, then their first slot
    uint32_t cell_grid;          // workgroups of the level-two kernels (they stride over the frame's chunks)
    uint32_t band;               // 1: a band context (the frame holds fewer ranks than the scene: scans and workgroups stop at *count's rows)
    uint32_t n_max;              // entries the rank-ordered buffers hold (the scene's splats): k_bin_count may load that far before it knows *count
};
void launch_bin(const BinBuffers& b, const BinGrid& g, uint32_t n, hipStream_t s);

struct BlendBuffers {
    const uint32_t* items;      // work items, four words each (BinBuffers::items)
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
    const uint32_t* seg_len_dev; // [0] the frame's segment length, [1] its number of work items (k_bin_finalize)
    uint32_t grid;              // persistent workgroups launched
    uint32_t capacity;          // entries the list can hold
    uint32_t nsplats;
    unsigned long long* bin_mask; // nbins arrival masks, zeroed by the finalize step: the workgroup delivering a bin's last
                                // segment folds the bin inside k_blend; null = the separate k_combine launch does it
    uint32_t saturate;          // 1: quadrants whose pixels can no longer change are skipped (bit-identical; k_blend); 0: A/B knob
    uint32_t sub;               // waves per 16x16 tile: 1 (k_blend, 256-thread workgroups) or 2 (k_blend2: halves a wave's serial walk)
};
// `between` (may be null) is recorded after k_blend and before k_combine
void launch_blend(const BlendBuffers& b, const BinGrid& g, float early_out_eps, hipStream_t s, hipEvent_t between);
void launch_clear_fb(float4* fb, int32_t W, int32_t H, hipStream_t s);
void launch_to_rgba8(const float4* fb, uint32_t* out, uint32_t npix, hipStream_t s);

// multi-GPU exchange helpers (RGBA8 slabs of the all-gather)
constexpr int MAX_SLABS = 16;
struct SlabEdges { int32_t x0[MAX_SLABS], x1[MAX_SLABS]; };
// (the library's own slabs carry SLAB_FLAG_WORDS words behind their H x slab_w pixels: "this band was not composited")
constexpr int SLAB_FLAG_WORDS = 4;   // one flag, padded to 16 bytes
void launch_pack_band_rgba8(const float4* fb, uint32_t* slab, int W, int H, int x0, int x1, int slab_w, hipStream_t s, const uint32_t* overflow = nullptr);
void launch_unpack_slabs_rgba8(const uint32_t* gathered, uint32_t* image, int W, int H, int slab_w, int world,
                               const SlabEdges& e, hipStream_t s, uint32_t* stale = nullptr);

}  // namespace gsr
