// C ABI of libgsplat_hip.so (see include/gsplat_hip.h).  Host side only: owns
// the device buffers, enqueues the per-frame kernels on one HIP stream, and
// never touches a CPU fallback: without a usable AMD GPU every entry point
// fails with GSR_ERR_NO_DEVICE / GSR_ERR_HIP.
#include "../../include/gsplat_hip.h"
#include "gsr_internal.h"

#include <dlfcn.h>
// RCCL: types and constants only -- the entry points are resolved with dlsym (gsr_comm_*), so the library loads on hosts
// without RCCL; and it BUILDS without the header too: the handful of declarations the calls need are repeated here
// (ABI of rccl.h / nccl.h 2.x: an opaque communicator pointer, a 128-byte id, enum values 0 = success, 1 = uint8).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1 } ncclDataType_t;
#endif

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <cstddef>
#include <string>
#include <vector>

using namespace gsr;

namespace {

thread_local std::string g_create_error;

enum Stage { EV_BEGIN = 0, EV_PROJECT, EV_SORT, EV_BIN, EV_BLEND /* after k_blend */, EV_COMBINE /* after k_combine */, EV_COUNT };

struct FrameState {  // small per-frame device words; initialised once (k_begin_frame), then every frame STORES them -- only `overflow`
                     // is accumulated, and zeroed by k_project_key
    int32_t minmax[2];
    uint32_t overflow;
    uint32_t queue;   // compositor work-item counter
    uint64_t visible;
    uint64_t tile_entries;
    uint64_t report[6];  // written by k_bin_finalize for the host: running sums accum[0..4], this frame's bin entries
    uint32_t digit_total[RADIX_LO_BINS + RADIX_HI_BINS];
    uint32_t sorted_count;  // entries of depth_index: n, or the band's survivors (SortBuffers::count)
    uint32_t seg_len;       // the frame's compositor segment length (k_bin_finalize -> k_blend)
    uint32_t n_items;       // and its number of work items (directly behind seg_len: k_blend reads both through one pointer)
    uint32_t spec;          // reserved (0)
};

}  // namespace

struct gsr_ctx {
    int device = 0;
    int cu_count = 256;               // compute units of the device (the compositor's persistent grid is sized from it)
    hipStream_t stream = nullptr;
    std::string error;
    gsr_options opt{};
    int W = 0, H = 0;
    int band_x0 = 0, band_x1 = 0;

    // scene
    uint32_t n = 0;
    float *px = nullptr, *py = nullptr, *pz = nullptr;
    uint32_t *cov0 = nullptr, *cov1 = nullptr, *cov2 = nullptr, *rgba = nullptr;
    // rotations / scales, only for scenes built on the device from .splat rows
    float4 *rotv = nullptr, *sclv = nullptr;
    bool have_rows = false;
    // spherical harmonics (optional)
    uint32_t *sh_r = nullptr, *sh_g = nullptr, *sh_b = nullptr;
    float4* shcol = nullptr;
    uint32_t sh_count = 0;
    int32_t band[3] = {-1, -1, -1};
    // per frame, sized by n
    int32_t* depth = nullptr;
    uint32_t* kept = nullptr;         // band mode: survivors per 256-splat workgroup of k_project_key, packed to the front of its depth slots
    uint8_t* kept_lane = nullptr;     // band mode: the lane a packed slot's splat came from
    uint32_t* koff = nullptr;         // band mode: survivors in front of every workgroup's block (k_kept_scan)
    uint32_t *keys = nullptr, *keys_tmp = nullptr, *idx_tmp = nullptr, *depth_index = nullptr;
    uint32_t* block_hist = nullptr;
    Record* rec = nullptr;
    uint2* bbox = nullptr;
    int32_t* slots = nullptr;     // FRAME_SLOTS x 128 B: partial depth (min, max), visible and tile sums of k_project_key
    uint32_t* rect_idx = nullptr; // per splat: packed bin rectangle (k_project_key); bin_rects holds them in depth order
    uint32_t sort_blocks = 0, sort_kpb = 0;
    // binning
    uint32_t *bin_table = nullptr, *bin_total = nullptr, *bin_start = nullptr, *bin_start_pre = nullptr, *bin_list = nullptr;
    uint32_t bin_rounds = 1;          // rounds of 2048 ranks per binning workgroup (alloc_bins; GSR_BIN_ROUNDS)
    long bin_rounds_env = 0;
    int bin_two_level_env = -1;       // GSR_BIN_TWO_LEVEL: 0 / 1 force the one- / two-level binning (alloc_bins); -1: by the bin grid
    bool bin_two_level = false;
    uint32_t *cell_list = nullptr, *cell_total = nullptr, *cell_start = nullptr, *chunk_start = nullptr, *chunk_info = nullptr, *cell_table2 = nullptr, *cell_wcnt = nullptr;
    uint32_t cell_capacity_alloc = 0, cell_ncells_alloc = 0, cell_grid = 0;
    uint32_t bin_big = 2;             // large bin grids: k_bin_scatter_big (GSR_BIN_BIG=0: the 64-register kernel + k_bin_finalize; 1: 2048-rank rounds)
    uint32_t *seg_start = nullptr, *items = nullptr;
    unsigned long long* bin_mask = nullptr;   // per-bin arrival masks of the compositor (null: separate k_combine launch)
    int items_by_size = 1;            // work items heaviest first (k_bin_finalize); GSR_ITEMS_BY_SIZE
    bool fuse_combine = true;
    bool saturate = true;             // skip quadrants that can no longer change (GSR_SATURATE=0: composite every entry)
    uint32_t long_tau_env = 0;        // GSR_LONG_TAU: the per-bin optical depth (true mass) from which a bin is one work item (0: the built-in rule)
    int long_items = -1;              // -1: long work items where the frame's optical depth says so (LONG_TAU), 0 / 1: pinned (GSR_LONG_ITEMS)
    uint32_t blend_sub = 1;           // compositor waves per 16x16 tile: 1 (k_blend) or 2 (k_blend2); alloc_bins, GSR_BLEND_SUB
    int blend_sub_env = 0;
    uint32_t* bin_rects = nullptr;
    uint32_t* rect_tmp = nullptr;     // the rectangles between the two LSD passes (rect_carry)
    uint32_t* sort_chunk_tab = nullptr; // bucket order: k_local_sort's work list
    int sort_parity = 0;              // which of the two sort-only slot sets the next sort-only frame uses
    bool slots_need_init = true;      // frame words and the three slot sets: initialised once, by the first frame's enqueue
    bool rect_carry = true;           // LSD sort order (large scenes): the packed rectangles travel with the keys (GSR_RECT_CARRY=0: the binning gathers them)
    bool rect_carry_bucket = false;   // ... also in the bucket order (GSR_RECT_CARRY=2; measured: what k_bin_count saves, the two sort kernels
                                      // pay -- C3 sort 35.0 -> 41.8 us, binning 47.6 -> 41.3 us -- so not by default)
    bool rects_sorted_now = false;    // this frame's sort left the rectangles in depth order
    float4* partial = nullptr;
    uint32_t bin_blocks = 0, bin_capacity = 0, bin_table_elems = 0, bin_nbins_alloc = 0;
    uint32_t max_items = 0, seg_len = 0, blend_grid = 2048;
    uint32_t seg_target_items = 5000;
    uint32_t timing_every = 1, frame_no = 0;
    bool sort_culled = false;  // the last sort kept only the band's survivors (depth_index / keys are partial)
    bool bucket_order_now = false;  // sort order of the frame being enqueued (part of the graph signature)
    // frame words
    FrameState* fstate = nullptr;       // device
    FrameState* fstate_host = nullptr;  // pinned
    uint64_t* accum = nullptr;          // device [8]: sums over frames (visible, bin entries, tile entries, frames), [4] entries of
                                        // the last frame, sticky [5] frames that did not fit, [6]/[7] most entries/items one needed
    uint64_t* mailbox = nullptr;        // pinned host words the device stores into: [0] accum[5] (k_bin_finalize), [1] low half: keys in the
                                        // largest high-digit bucket of the last sorted frame (k_local_sort / last LSD pass)
    int sort_order = -1;                // -1: chosen per frame from the reported bucket size; 0: always LSD; 1: always bucket order
    uint64_t* mailbox_dev = nullptr;    // its device address
    uint64_t overflow_seen = 0;         // accum[5] as of the last regrowth
    uint64_t overflow_frames = 0;       // frames that did not fit, since the context was created
    uint64_t dropped_frames = 0;        // of those, frames never composited (later frames had been enqueued before the host noticed)
    uint64_t dropped_unreported = 0;    // dropped frames gsr_sync has not reported yet
    // output
    float4* fb = nullptr;
    uint32_t* fb8 = nullptr;
    size_t fb_pixels = 0;

    // multi-GPU exchange (gsr_comm_init): RCCL communicator, its stream, the RGBA8 slab / gathered slabs / full frame
    ncclComm_t comm = nullptr;
    gsr_allgather_fn comm_fn = nullptr;   // gsr_comm_init_custom: the caller's collective in place of ncclAllGather
    void* comm_fn_user = nullptr;
    bool comm_owned = true;               // false: communicator and exchange stream belong to another context (gsr_comm_share)
    gsr_ctx* comm_leader = nullptr;       // that context; it lists this one in comm_followers and detaches it when it leaves the group first
    std::vector<gsr_ctx*> comm_followers;
    int comm_rank = 0, comm_world = 0, slab_w = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_slab_free = nullptr;
    uint32_t *slab = nullptr, *gathered = nullptr, *frame8 = nullptr;
    SlabEdges comm_edges{};
    bool frame8_valid = false;

    CamParams cam{};
    CamParams cam_frame{};            // the camera of the last rendered frame
    CamParams* cam_dev = nullptr;     // a camera slot in device memory (written by the one-time initialisation only)
    hipEvent_t link_ev[2] = {nullptr, nullptr};  // gsr_stream_order
    // the frame's launch chain replayed as a HIP graph (frames that carry no stage events)
    bool graphs_enabled = true;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_fresh = false;                 // the graph was captured for the frame being enqueued
    hipGraphNode_t graph_project = nullptr;   // the captured chain's projection node: its camera argument is rewritten every replay
    ProjectLaunch proj{};                     // the projection kernel's arguments of the current frame
    std::vector<uint64_t> graph_sig;  // everything the chain's kernel arguments and grids derive from
    bool have_cam = false, have_frame = false, have_sort = false;

    // timing: a ring of event sets so that frames can be enqueued back to back without a host
    // sync per frame; gsr_sync / gsr_get_timings drain the ring
    static constexpr int EV_RING = 128;
    hipEvent_t evring[EV_RING][EV_COUNT]{};
    bool ev_is_render[EV_RING]{};
    int ev_head = 0, ev_pending = 0;
    hipEvent_t* ev = nullptr;  // the set being recorded
    bool ev_valid = false, ev_recorded = false, ev_render = false;
    gsr_timings tm{};
};

namespace {

int fail(gsr_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) {
        c->error = buf;
        // A chain that stopped half way (a failed launch behind k_project_key, a device error) leaves the frame slots and frame
        // words as that frame had them -- partial min / max, counters -- and nothing would ever clean them: their last reader in a
        // frame (the finalize step) did not run.  The next frame starts with the one-time initialisation again.
        if (code == GSR_ERR_HIP) c->slots_need_init = true;
    }
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail((c), GSR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T>
int dev_alloc(gsr_ctx* c, T** p, size_t count)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (!count) count = 1;
    HIP_TRY(c, hipMalloc((void**)p, count * sizeof(T)));
    return GSR_OK;
}

template <class T>
void dev_free(T** p)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
}

BinGrid make_grid(const gsr_ctx* c)
{
    BinGrid g;
    g.W = c->W; g.H = c->H;
    g.nbx = (c->W + BIN_PX - 1) / BIN_PX;
    g.nby = (c->H + BIN_PX - 1) / BIN_PX;
    if (c->band_x1 > c->band_x0) {
        g.bx_lo = c->band_x0 / BIN_PX;
        g.bx_hi = std::min((c->band_x1 + BIN_PX - 1) / BIN_PX, g.nbx);
    } else {
        g.bx_lo = 0; g.bx_hi = g.nbx;
    }
    return g;
}

bool band_is_partial(const gsr_ctx* c)
{
    const BinGrid g = make_grid(c);
    return g.bx_lo > 0 || g.bx_hi < g.nbx;
}

// Compositor work-item granularity: list entries per (bin, segment) item.  0x7fffff00 = one item per
// bin, which early termination needs (a segment cannot see whether earlier ones saturated the bin).
constexpr uint32_t SEG_LEN_MIN = 512;            // shortest segment; k_bin_finalize lengthens it so that the frame is cut
                                                 // into about SEG_TARGET_* full segments (multiples of 256 entries)
constexpr uint32_t SEG_TARGET_EXACT = 5000;      // one frame at a time: concurrency from the frame's own segments (C3: 512)
constexpr uint32_t SEG_TARGET_THROUGHPUT = 1300; // GSR_FLAG_THROUGHPUT: concurrency comes from the other frames in flight
                                                 // (C3: 2048-entry segments; a 1/8-screen band stays at 512)
// Persistent compositor workgroups per CU.  k_blend is built for 7 waves per SIMD (72 VGPRs), so 7 four-wave
// workgroups are resident per CU and the grid must not exceed that: a workgroup that is not resident at launch still
// owns its first work item by index (a heavy one: the queue is ordered heaviest first) and starts it only when a
// resident workgroup exits.  With 8 per CU, one item in eight began at 222 us of a 270 us kernel (in-kernel stamps,
// scripts/blend_stamps.py): k_blend 271 -> 252 us on C3 at 7 per CU.
constexpr uint32_t BLEND_WG_PER_CU_EXACT = 7;
// Contexts that overlap with others' kernels (GSR_FLAG_THROUGHPUT): 6 per CU left a wave slot per SIMD to the other
// contexts and was best while the fold of the partials was a kernel of its own; with the fold inside k_blend 7 is
// (bench.py, three frames in flight, C3: 3324 -> 3400 frames/s, reproducible; C2 -0.8 %, C4 and early-out unchanged).
constexpr uint32_t BLEND_WG_PER_CU_THROUGHPUT = 7;
// Two waves per tile (k_blend2, 512-thread workgroups): three workgroups per CU are resident (6 waves per SIMD).
constexpr uint32_t BLEND_WG_PER_CU_SUB2 = 3;
constexpr uint32_t SUB2_MAX_BINS = 4096, SEG_LEN_MIN_SUB2 = 1024;
constexpr uint32_t BIN_BLOCKS_TARGET = 640, BIN_ROUNDS_MAX = 8;
constexpr uint32_t TWO_LEVEL_MIN_BINS = 4096, CELL_WG_PER_CU = 4;
constexpr uint32_t SEG_LEN_WHOLE_BIN = 0x7fffff00u;

// Sort order.  Up to BUCKET_ORDER_MAX_N splats the radix sort runs high digit first with one workgroup per bucket
// (four launches, k_sort.hip) -- unless the last sorted frame reported a bucket above LOCAL_BUCKET_LIMIT keys: depth
// outliers stretch the key range and can put most of a scene into one bucket, which would serialise in its workgroup.
// Then, for the first frame of a scene, and above BUCKET_ORDER_MAX_N (the average bucket alone needs several chunks)
// it runs the LSD order (six launches).  Same permutation either way.  GSR_SORT_ORDER=lsd|bucket pins it.
constexpr uint32_t BUCKET_ORDER_MAX_N = 3u << 20;
constexpr uint32_t LOCAL_BUCKET_LIMIT = 48u << 10;

// Work-item length.  With the saturation skip of k_blend a work item ends as soon as nothing it could still add can change
// a bit of its pixels, and that needs the item to contain the splats that saturate it: a bin cut into 512-entry segments
// never saturates inside one of them (every segment starts from transmittance 1), a bin processed as one item stops
// after the few thousand entries that matter (C3: 3430 -> 4990 frames/s with three frames in flight, 2670 -> 2945 one at
// a time; C4: 292 -> 856).  Where the scene does not saturate (C2: small splats, 9 % of the entries skipped against 53 %
// on C3 and 85 % on C4; or any thin, low-opacity scene) long items only cost balance (C2: 6670 -> 2780 frames/s).
// k_bin_finalize decides per frame, from a figure the projection already has: the frame's optical depth
//     tau = sum over visible splats of opacity x (16x16 tiles its box overlaps) x 256 / pixels
// (C1 14, C2 74, C3 362, C4 1090): items are at least SEG_LEN_LONG entries (in practice whole bins) from LONG_TAU_* on.
// A function of the frame alone: no feedback from earlier frames, the same frame always takes the same path.
// Where long items start to pay (scripts/tau_crossover.py: the C3 and C2 generators at 0.25 .. 1.6 M splats, 1080p): with
// other frames' kernels filling the gaps, between tau 90 and 145 for both generators (tau 90: 10 390 -> 10 080 frames/s,
// tau 145: 7350 -> 8640, tau 250: 4730 -> 6800); one frame at a time the few long items are the frame's tail and the
// crossover depends on the scene (C3 generator: tau ~ 255, C3 itself +23 %; the C2 generator's small splats still lose
// 8 % at tau 390), so the threshold there stays high.
constexpr uint32_t SEG_LEN_LONG = 32768;   // (16384: C4 k_blend 437 instead of 405 us -- its heaviest bins hold 50-100 k entries; 65536 measures the same)
constexpr uint32_t LONG_TAU_EXACT = 340, LONG_TAU_THROUGHPUT = 120;
// a frame that is not dense as a whole: bins far past saturation become one item only where a list entry carries at least this
// optical mass (pixels): C3 14, C2 8, 2 M tiny splats 1.9 -- one frame at a time a 3000-entry serial walk is the frame's tail
constexpr uint32_t LONG_MASS_MIN_EXACT = 12, LONG_MASS_MIN_THROUGHPUT = 0;
constexpr uint32_t LONG_TILES_X2_EXACT = 9;   // one frame at a time: and at least 4.5 tiles per visible splat (k_bin_finalize)
constexpr uint32_t LONG_TILES_X2_THROUGHPUT = 6;   // with frames in flight: 3 (scripts/policy_check.py: 2 M tiny splats, 1.9 tiles each, tau 264:
                                                   // long items -26 %; the C2 generator, 3.6 tiles each: +10 % at the same tau)

inline bool use_bucket_order(const gsr_ctx* c)
{
    if (c->sort_order >= 0) return c->sort_order == 1;
    if (c->n > BUCKET_ORDER_MAX_N) return false;
    const uint32_t largest = reinterpret_cast<volatile const uint32_t*>(c->mailbox)[2];   // low half of mailbox[1]
    return largest <= LOCAL_BUCKET_LIMIT;   // 0xffffffff until a frame of this scene has reported
}

int alloc_bins(gsr_ctx* c)
{
    if (!c->W) return GSR_OK;
    const BinGrid g = make_grid(c);
    const uint32_t nbins = (uint32_t)((g.bx_hi - g.bx_lo) * g.nby);
    // Ranks per binning workgroup: rounds of 2048.  The count / scan / scatter passes exchange a [workgroup][bin] table; with
    // one round per workgroup it is 80 MB at 5 M splats and 8160 bins.  Large grids (the k_bin_scatter_big form, > 4096 bins)
    // take several rounds per workgroup, keeping about BIN_BLOCKS_TARGET workgroups (C4: 4 rounds, 611 workgroups, 20 MB).
    // Two-level binning (k_bin.hip): grids above TWO_LEVEL_MIN_BINS bins whose cells of 4 x 4 bins number at most 4096.
    const uint32_t ncells = (uint32_t)(((g.bx_hi - g.bx_lo + 3) >> 2) * ((g.nby + 3) >> 2));
    // (its level-two stores address the list with 32-bit byte offsets: lists of 2^30 entries or more take the one-level pass)
    const uint64_t cap_now = c->bin_capacity ? c->bin_capacity : std::max<uint64_t>(6ull * c->n + (1u << 20), 1u << 22);
    c->bin_two_level = ncells <= 4096u && cap_now < (1ull << 30) &&
                       (c->bin_two_level_env >= 0 ? c->bin_two_level_env == 1 : nbins > TWO_LEVEL_MIN_BINS);
    c->bin_rounds = 1;
    if (!c->bin_two_level && c->bin_big && nbins > 4096)
        c->bin_rounds = std::min<uint32_t>(BIN_ROUNDS_MAX, std::max<uint32_t>(1u, ((c->n + 2047u) / 2048u + BIN_BLOCKS_TARGET - 1u) / BIN_BLOCKS_TARGET));
    if (!c->bin_two_level && c->bin_big && nbins > 4096 && c->bin_rounds_env > 0) c->bin_rounds = (uint32_t)c->bin_rounds_env;
    c->bin_blocks = (c->n + 2048u * c->bin_rounds - 1u) / (2048u * c->bin_rounds);
    const size_t table = (size_t)std::max(c->bin_blocks, 1u) * (c->bin_two_level ? ncells + 1u : nbins);
    if (table > c->bin_table_elems) {
        if (int r = dev_alloc(c, &c->bin_table, table)) return r;
        c->bin_table_elems = (uint32_t)table;
    }
    bool items_dirty = false;
    if (nbins > c->bin_nbins_alloc) {
        if (int r = dev_alloc(c, &c->bin_total, nbins)) return r;
        if (int r = dev_alloc(c, &c->bin_start, nbins + 1)) return r;
        if (int r = dev_alloc(c, &c->bin_start_pre, nbins + 1)) return r;
        if (int r = dev_alloc(c, &c->seg_start, nbins + 1)) return r;
        if (c->fuse_combine) {
            if (int r = dev_alloc(c, &c->bin_mask, nbins)) return r;
        }
        c->bin_nbins_alloc = nbins;
        items_dirty = true;
    }
    if (!c->bin_capacity) {
        c->bin_capacity = std::max<uint32_t>(6u * c->n + (1u << 20), 1u << 22);
        if (int r = dev_alloc(c, &c->bin_list, c->bin_capacity)) return r;
        items_dirty = true;
    }
    if (c->bin_two_level) {
        if (ncells > c->cell_ncells_alloc) {
            if (int r = dev_alloc(c, &c->cell_total, ncells + 1)) return r;
            if (int r = dev_alloc(c, &c->cell_start, ncells + 1)) return r;
            if (int r = dev_alloc(c, &c->chunk_start, ncells + 2)) return r;
            c->cell_ncells_alloc = ncells;
            c->cell_capacity_alloc = 0;
        }
        if (c->bin_capacity > c->cell_capacity_alloc) {
            if (int r = dev_alloc(c, &c->cell_list, (size_t)c->bin_capacity * 2)) return r;
            if (int r = dev_alloc(c, &c->cell_table2, ((size_t)c->bin_capacity / 2048u + ncells + 1u) * 16u)) return r;
            if (int r = dev_alloc(c, &c->chunk_info, ((size_t)c->bin_capacity / 2048u + ncells + 1u) * 4u)) return r;
            if (int r = dev_alloc(c, &c->cell_wcnt, ((size_t)c->bin_capacity / 2048u + ncells + 1u) * 64u)) return r;
            c->cell_capacity_alloc = c->bin_capacity;
        }
        // the level-two kernels stride over the frame's chunks: two 16-wave workgroups per CU, twice over
        c->cell_grid = (uint32_t)std::max(c->cu_count, 1) * CELL_WG_PER_CU;
        if (const char* e = getenv("GSR_CELL_GRID")) {   // tuning knob: workgroups of the level-two kernels
            const long v = atol(e);
            if (v >= 1) c->cell_grid = (uint32_t)std::min(v, 65535L);
        }
    }
    const bool throughput = (c->opt.flags & GSR_FLAG_THROUGHPUT) != 0;
    c->seg_len = c->opt.early_out_eps > 0.0f ? SEG_LEN_WHOLE_BIN : SEG_LEN_MIN;
    c->seg_target_items = throughput ? SEG_TARGET_THROUGHPUT : SEG_TARGET_EXACT;
    // Waves per tile.  Two (k_blend2) halve a wave's serial walk over a work item -- the pole of a frame rendered alone, where a
    // wave needs ~560 cycles per entry visit whatever else the chip does -- and pay with occupancy (24 instead of 28 waves
    // per CU) and saturation tests at chunk instead of 64-entry boundaries.  Measured one frame at a time: C3 k_blend 194 ->
    // 147 us, C1 20 -> 15; C2 (short segments) 77 -> 86, with 1024-entry segments 80; C4, whose 8160 bins keep every slot
    // busy: 412 -> 509; three frames in flight, C3: 5280 -> 4410 frames/s.  So: contexts that render one frame at a time, up
    // to SUB2_MAX_BINS bins, with segments of at least 1024 entries.  (Leaving the choice to k_bin_finalize per frame --
    // both kernels launched, the other one returning at once -- cost 4.5 us per frame for the idle launch.)
    c->blend_sub = c->blend_sub_env ? (uint32_t)c->blend_sub_env : (!throughput && nbins <= SUB2_MAX_BINS) ? 2u : 1u;
    if (c->blend_sub >= 2 && c->seg_len != SEG_LEN_WHOLE_BIN) c->seg_len = SEG_LEN_MIN_SUB2;
    c->blend_grid = (c->blend_sub >= 2 ? BLEND_WG_PER_CU_SUB2 : throughput ? BLEND_WG_PER_CU_THROUGHPUT : BLEND_WG_PER_CU_EXACT) * (uint32_t)std::max(c->cu_count, 1);
    if (const char* e = getenv("GSR_SEG_TARGET")) {  // tuning knob: full segments a frame is cut into at least
        const long v = atol(e);
        if (v >= 1) c->seg_target_items = (uint32_t)v;
    }
    if (const char* e = getenv("GSR_BLEND_GRID")) {  // tuning knob: persistent compositor workgroups
        const long v = atol(e);
        if (v >= 1) c->blend_grid = (uint32_t)v;
    }
    if (const char* e = getenv("GSR_SEG_LEN")) {  // tuning knob: entries per compositor work item (multiple of 256)
        const long v = atol(e);
        if (v >= 256 && c->seg_len != SEG_LEN_WHOLE_BIN) c->seg_len = (uint32_t)(v / 256 * 256);
    }
    // segments = work items (each may need a partial slot): one per bin plus one per seg_len entries
    const uint32_t want_segs = nbins + c->bin_capacity / c->seg_len + 16;
    const uint32_t want_items = want_segs;
    if (items_dirty || want_items > c->max_items) {
        c->max_items = want_items;
        if (int r = dev_alloc(c, &c->items, (size_t)c->max_items * 4)) return r;   // (four words per work item: k_bin_finalize)
        if (c->seg_len != SEG_LEN_WHOLE_BIN) {
            if (int r = dev_alloc(c, &c->partial, (size_t)want_segs * BIN_PX * BIN_PX)) return r;
        }
    }
    return GSR_OK;
}

int alloc_fb(gsr_ctx* c)
{
    const size_t np = (size_t)c->W * c->H;
    if (np > c->fb_pixels) {
        if (int r = dev_alloc(c, &c->fb, np)) return r;
        if (int r = dev_alloc(c, &c->fb8, np)) return r;
        c->fb_pixels = np;
    }
    launch_clear_fb(c->fb, c->W, c->H, c->stream);
    return GSR_OK;
}

int finish_frame(gsr_ctx* c);
void comm_release(gsr_ctx* c);
int handle_overflow(gsr_ctx* c, uint64_t* newly);
inline bool overflow_pending(const gsr_ctx* c);

// the frame's device work on the context's stream: frame words reset, projection + depth key, sort, (bin, blend)
static int enqueue_chain(gsr_ctx* c, bool render, bool timing)
{
    hipStream_t s = c->stream;
    int32_t* slots_now = c->slots;
    if (timing) HIP_TRY(c, hipEventRecord(c->ev[EV_BEGIN], s));
    if (c->n) {
        SceneSoA sc{c->px, c->py, c->pz, c->cov0, c->cov1, c->cov2, c->rgba, c->sh_r, c->sh_g, c->sh_b, c->shcol};
        if (render) {
            // (band mode: the workgroups pack their survivors, see k_project_key; the sort below reads the same two arrays)
            const bool pack = band_is_partial(c);
            c->proj = ProjectLaunch{sc, c->n, c->cam, 1, c->depth, c->slots, c->rec, nullptr, c->rect_idx, &c->fstate->overflow,
                                    pack ? c->kept : nullptr, pack ? c->kept_lane : nullptr, {}};
            launch_project_key(c->proj, s);
        }
        else {   // a sort-only frame: its own slots (sets 1 and 2 in turn; set 0 belongs to the render frames and k_begin_frame)
            slots_now = c->slots + (size_t)(1 + c->sort_parity) * FRAME_SLOTS * FRAME_SLOT_WORDS;
            launch_depth_key(sc, c->n, c->cam, c->depth, slots_now, c->slots + (size_t)(2 - c->sort_parity) * FRAME_SLOTS * FRAME_SLOT_WORDS, s);
            c->sort_parity ^= 1;
        }
    }
    if (timing) HIP_TRY(c, hipEventRecord(c->ev[EV_PROJECT], s));
    if (c->n) {
        // band mode (a context that composites only part of the screen): sort and bin only the splats whose box
        // touches the band; the full depthIndex is produced on demand (gsr_read_depth_index)
        const bool cull = render && band_is_partial(c);
        SortBuffers sb{c->depth, slots_now, c->fstate->minmax, c->keys, c->keys_tmp, c->idx_tmp, c->depth_index,
                       c->block_hist, c->fstate->digit_total, c->rect_idx, c->kept, c->kept_lane, cull ? c->koff : nullptr, &c->fstate->sorted_count, c->sort_kpb, c->sort_blocks,
                       c->bucket_order_now ? 1 : 0, reinterpret_cast<uint32_t*>(c->mailbox_dev + 1),
                       c->sort_chunk_tab, c->rect_tmp, (render && c->rect_carry && (c->rect_carry_bucket || !c->bucket_order_now)) ? c->bin_rects : nullptr};
        c->sort_culled = cull;
        c->rects_sorted_now = sb.rects_out != nullptr;
        launch_sort(sb, c->n, s);
    }
    if (timing) HIP_TRY(c, hipEventRecord(c->ev[EV_SORT], s));
    if (render) {
        const BinGrid g = make_grid(c);
        const int nbins = (g.bx_hi - g.bx_lo) * g.nby;
        if (!c->n) {
            HIP_TRY(c, hipMemsetAsync(c->bin_total, 0, sizeof(uint32_t) * nbins, s));
            HIP_TRY(c, hipMemsetAsync(&c->fstate->overflow, 0, sizeof(uint32_t), s));   // (k_project_key zeroes it otherwise)
        }
        BinBuffers bb{c->depth_index, &c->fstate->sorted_count, c->bin_table, c->slots, c->rect_idx, c->bin_rects, (c->n && c->rects_sorted_now) ? 1u : 0u, c->bin_total, c->bin_start, c->bin_start_pre, c->bin_rounds, c->bin_big, c->seg_start,
                      c->items, c->bin_list, &c->fstate->overflow, &c->fstate->visible, &c->fstate->tile_entries,
                      c->accum, c->mailbox_dev, c->fstate->report, c->bin_capacity, c->max_items, c->seg_len, &c->fstate->seg_len, c->items_by_size, &c->fstate->queue, std::min<uint32_t>(c->max_items, c->blend_grid), c->seg_target_items, c->bin_blocks, c->bin_mask,
                      c->seg_len == SEG_LEN_WHOLE_BIN ? 0 : c->long_items >= 0 ? c->long_items : c->saturate ? -1 : 0, SEG_LEN_LONG,
                      (c->opt.flags & GSR_FLAG_THROUGHPUT) ? LONG_TAU_THROUGHPUT : LONG_TAU_EXACT,
                      (uint32_t)((g.bx_hi - g.bx_lo) * BIN_PX) * (uint32_t)c->H,
                      (c->opt.flags & GSR_FLAG_THROUGHPUT) ? LONG_TILES_X2_THROUGHPUT : LONG_TILES_X2_EXACT,
                      c->long_tau_env, (c->opt.flags & GSR_FLAG_THROUGHPUT) ? LONG_MASS_MIN_THROUGHPUT : LONG_MASS_MIN_EXACT,
                      c->bin_two_level ? 1u : 0u, c->cell_list, c->cell_total, c->cell_start, c->chunk_start, c->chunk_info, c->cell_wcnt, c->cell_table2, c->cell_grid, band_is_partial(c) ? 1u : 0u, c->n};
        launch_bin(bb, g, c->n, s);
        if (timing) HIP_TRY(c, hipEventRecord(c->ev[EV_BIN], s));
        BlendBuffers bl{c->items, c->seg_start, c->bin_start, c->bin_list, c->rec, c->bbox, c->shcol, c->fb, c->partial,
                        &c->fstate->queue, c->seg_len, &c->fstate->seg_len, std::min<uint32_t>(c->max_items, c->blend_grid), c->bin_capacity,
                        std::max(c->n, 1u), c->bin_mask, c->saturate ? 1u : 0u, c->blend_sub};
        launch_blend(bl, g, c->opt.early_out_eps, s, (timing && !c->bin_mask) ? c->ev[EV_BLEND] : nullptr);
        if (timing) HIP_TRY(c, hipEventRecord(c->ev[EV_COMBINE], s));
    }
    HIP_TRY(c, hipGetLastError());
    return GSR_OK;
}

static void drop_graph(gsr_ctx* c)
{
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    c->graph_exec = nullptr; c->graph = nullptr; c->graph_project = nullptr;
    c->graph_sig.clear();
}

// Every value the chain's kernel arguments, grids and LDS sizes derive from.  The camera is not among them: it is
// read from c->cam_dev.  A graph captured for one signature is replayed while the signature stays the same.
static std::vector<uint64_t> chain_signature(const gsr_ctx* c)
{
    const BinGrid g = make_grid(c);
    std::vector<uint64_t> v;
    auto P = [&v](const void* p) { v.push_back((uint64_t)(uintptr_t)p); };
    auto U = [&v](uint64_t x) { v.push_back(x); };
    P(c->px); P(c->py); P(c->pz); P(c->cov0); P(c->cov1); P(c->cov2); P(c->rgba); P(c->sh_r); P(c->sh_g); P(c->sh_b); P(c->shcol);
    P(c->depth); P(c->kept); P(c->kept_lane); P(c->koff); P(c->keys); P(c->keys_tmp); P(c->idx_tmp); P(c->depth_index); P(c->block_hist); P(c->fstate);
    P(c->rec); P(c->bbox); P(c->slots); P(c->rect_idx); P(c->bin_table); P(c->bin_rects); P(c->bin_total); P(c->bin_start); P(c->seg_start); P(c->bin_mask);
    P(c->items); P(c->bin_list); P(c->partial); P(c->fb); P(c->accum); P(c->cam_dev);
    U(c->n); U((uint64_t)c->W); U((uint64_t)c->H); U((uint64_t)g.bx_lo); U((uint64_t)g.bx_hi); U(c->sort_kpb); U(c->sort_blocks);
    U(c->bin_capacity); U(c->max_items); U(c->seg_len); U(c->seg_target_items); U(c->blend_grid); U(c->bin_blocks); U(c->bin_rounds); U(c->bin_big); P(c->bin_start_pre);
    U((uint64_t)c->opt.flags); U((uint64_t)(c->opt.early_out_eps * 1e9f)); U(c->bucket_order_now ? 1u : 0u); U(c->saturate ? 1u : 0u); U((uint64_t)c->items_by_size); U((uint64_t)(int64_t)c->long_items); U(c->long_tau_env);
    U(c->blend_sub);
    U(c->bin_two_level ? 1u : 0u); P(c->cell_list); P(c->cell_total); P(c->cell_start); P(c->chunk_start); P(c->chunk_info); P(c->cell_wcnt); P(c->cell_table2); U(c->cell_grid); P(c->rect_tmp); P(c->sort_chunk_tab); U(c->rect_carry ? (c->rect_carry_bucket ? 1u : 2u) : 0u);
    return v;
}

// enqueue one frame: camera into its device slot, then the chain -- as individual launches when the frame carries
// stage events or is sort-only, as one graph launch otherwise (13 launches and a copy become one: the host issues a
// frame in ~12 us instead of ~45 us, which is what a rank of a multi-GPU run or a small scene is bound by)
int enqueue_frame(gsr_ctx* c, bool render)
{
    if (!c->have_cam) return fail(c, GSR_ERR_ARG, "gsr_set_camera has not been called");
    if (render && (!c->W || !c->H)) return fail(c, GSR_ERR_ARG, "framebuffer size is 0");
    hipStream_t s = c->stream;
    if (render && overflow_pending(c)) {
        // an earlier asynchronous frame did not fit: regrow before this one is enqueued.  The frames that overflowed
        // are lost (later frames were already behind them); gsr_sync reports how many.
        uint64_t newly = 0;
        if (int r = handle_overflow(c, &newly)) return r;
        c->dropped_frames += newly;
        c->dropped_unreported += newly;
    }
    // stage timing is sampled: every timing_every-th frame carries the six events (each is a packet the command
    // processor has to retire; on short frames they cost more than they measure)
    const bool timing = c->ev_valid && c->timing_every != 0xffffffffu && (c->frame_no++ % c->timing_every) == 0;   // (0xffffffff: no frame)
    if (timing) {
        if (c->ev_pending == gsr_ctx::EV_RING) { if (int r = finish_frame(c)) return r; }
        const int slot = (c->ev_head + c->ev_pending) % gsr_ctx::EV_RING;
        c->ev = c->evring[slot];
        c->ev_is_render[slot] = render;
    }
    c->cam.W = c->W; c->cam.H = c->H;
    {
        const BinGrid bg = make_grid(c);
        c->cam.band_px0 = bg.bx_lo * BIN_PX;
        c->cam.band_px1 = bg.bx_hi * BIN_PX;
    }
    c->bucket_order_now = use_bucket_order(c);
    c->cam.sh_on = c->sh_count ? 1 : 0;
    c->cam.band[0] = c->band[0]; c->cam.band[1] = c->band[1]; c->cam.band[2] = c->band[2];
    if (render) c->cam_frame = c->cam;   // (gsr_read_records projects once more for this camera to get the pixel boxes)
    // No kernel in front of the frame: the camera is an argument of the projection kernel (k_project_key; k_depth_key in a
    // sort-only frame), the frame slots are left clean by their last reader, the frame words are stored, not accumulated
    // (the overflow word is zeroed by k_project_key).  The context's first frame initialises all of them, once.
    static_assert(offsetof(FrameState, minmax) == 0 && sizeof(FrameState) % 4 == 0, "k_begin_frame initialises the frame words");
    if (c->slots_need_init) {
        for (int k = 0; k < 3; k++)   // the render frames' slot set and the two of the sort-only frames
            launch_begin_frame(c->cam, c->cam_dev, reinterpret_cast<uint32_t*>(c->fstate), (uint32_t)(sizeof(FrameState) / 4),
                               c->slots + (size_t)k * FRAME_SLOTS * FRAME_SLOT_WORDS, s);
        c->slots_need_init = false;
    }

    bool replayed = false;
    if (c->graphs_enabled && render && !timing) {
        std::vector<uint64_t> sig = chain_signature(c);
        if (!c->graph_exec || sig != c->graph_sig) {
            drop_graph(c);
            bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                const int r = enqueue_chain(c, true, false);
                hipGraph_t gph = nullptr;
                ok = (hipStreamEndCapture(s, &gph) == hipSuccess) && r == GSR_OK && gph;
                if (ok && c->n) {   // the node whose camera argument changes from frame to frame
                    size_t nn = 0;
                    ok = hipGraphGetNodes(gph, nullptr, &nn) == hipSuccess && nn > 0;
                    std::vector<hipGraphNode_t> nodes(nn);
                    if (ok) ok = hipGraphGetNodes(gph, nodes.data(), &nn) == hipSuccess;
                    for (size_t k = 0; ok && k < nn && !c->graph_project; k++) {
                        hipGraphNodeType ty;
                        hipKernelNodeParams kp{};
                        if (hipGraphNodeGetType(nodes[k], &ty) == hipSuccess && ty == hipGraphNodeTypeKernel &&
                            hipGraphKernelNodeGetParams(nodes[k], &kp) == hipSuccess && kp.func == project_key_kernel())
                            c->graph_project = nodes[k];
                    }
                    ok = ok && c->graph_project != nullptr;
                }
                if (ok) ok = hipGraphInstantiate(&c->graph_exec, gph, nullptr, nullptr, 0) == hipSuccess;
                if (ok) { c->graph = gph; c->graph_sig = std::move(sig); c->graph_fresh = true; }
                else if (gph) (void)hipGraphDestroy(gph);
            }
            if (!ok) {  // this runtime cannot capture the chain: individual launches from now on
                (void)hipGetLastError();
                drop_graph(c);
                c->graphs_enabled = false;
            }
        }
        if (c->graph_exec) {
            if (c->graph_project && !c->graph_fresh) {   // (a graph captured for this very frame already holds its camera)
                c->proj.cam = c->cam;
                c->proj.bind();
                hipKernelNodeParams kp{};
                kp.func = const_cast<void*>(project_key_kernel());
                kp.gridDim = project_key_grid(c->proj.n);
                kp.blockDim = dim3(PROJ_THREADS);
                kp.sharedMemBytes = 0;
                kp.kernelParams = c->proj.ptrs;
                kp.extra = nullptr;
                if (hipGraphExecKernelNodeSetParams(c->graph_exec, c->graph_project, &kp) != hipSuccess) {
                    (void)hipGetLastError();   // this runtime cannot rewrite the node: individual launches from now on
                    drop_graph(c);
                    c->graphs_enabled = false;
                }
            }
            c->graph_fresh = false;
        }
        if (c->graph_exec) {
            HIP_TRY(c, hipGraphLaunch(c->graph_exec, s));   // (a failure marks the frame slots for re-initialisation: fail())
            c->sort_culled = band_is_partial(c);
            replayed = true;
        }
    }
    if (!replayed) { if (int r = enqueue_chain(c, render, timing)) return r; }
    if (timing) c->ev_pending++;
    c->ev_recorded = timing;
    c->ev_render = render;
    c->have_sort = true;
    c->have_frame = c->have_frame || render;
    return GSR_OK;
}

int finish_frame(gsr_ctx* c)
{
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    while (c->ev_pending > 0) {
        hipEvent_t* ev = c->evring[c->ev_head];
        const bool render = c->ev_is_render[c->ev_head];
        float a = 0, b = 0, d = 0, e = 0, f = 0, t = 0;
        HIP_TRY(c, hipEventElapsedTime(&a, ev[EV_BEGIN], ev[EV_PROJECT]));
        HIP_TRY(c, hipEventElapsedTime(&b, ev[EV_PROJECT], ev[EV_SORT]));
        t = a + b;
        if (render) {
            HIP_TRY(c, hipEventElapsedTime(&d, ev[EV_SORT], ev[EV_BIN]));
            if (c->bin_mask) {   // the fold of multi-segment bins runs inside k_blend: one stage, no event in between
                HIP_TRY(c, hipEventElapsedTime(&e, ev[EV_BIN], ev[EV_COMBINE]));
            } else {
                HIP_TRY(c, hipEventElapsedTime(&e, ev[EV_BIN], ev[EV_BLEND]));
                HIP_TRY(c, hipEventElapsedTime(&f, ev[EV_BLEND], ev[EV_COMBINE]));
            }
            HIP_TRY(c, hipEventElapsedTime(&t, ev[EV_BEGIN], ev[EV_COMBINE]));
        }
        c->tm.ms_project_key = a; c->tm.ms_sort = b; c->tm.ms_bin = d; c->tm.ms_blend = e; c->tm.ms_combine = f; c->tm.ms_total = t;
        c->tm.sum_ms_project_key += a; c->tm.sum_ms_sort += b; c->tm.sum_ms_bin += d; c->tm.sum_ms_blend += e; c->tm.sum_ms_combine += f;
        c->tm.sum_ms_total += t;
        c->tm.frames++;
        c->ev_head = (c->ev_head + 1) % gsr_ctx::EV_RING;
        c->ev_pending--;
    }
    c->ev_recorded = false;
    return GSR_OK;
}

// after a synchronised render: pull the frame words of the last frame (counts for gsr_timings, its overflow word)
int check_frame_words(gsr_ctx* c, bool* overflowed)
{
    // one small copy: the frame words up to and including k_bin_finalize's report
    HIP_TRY(c, hipMemcpyAsync(c->fstate_host, c->fstate, offsetof(FrameState, digit_total), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint64_t* acc = c->fstate_host->report;
    const uint32_t total = (uint32_t)c->fstate_host->report[5];
    c->tm.sum_visible = acc[0]; c->tm.sum_bin_entries = acc[1]; c->tm.sum_tile_entries = acc[2]; c->tm.sum_frames = acc[3];
    c->tm.visible = c->fstate_host->visible;
    c->tm.tile_entries = c->fstate_host->tile_entries;
    c->tm.bin_entries = total;
    c->tm.n = c->n;
    *overflowed = c->fstate_host->overflow != 0;
    return GSR_OK;
}

// true when the device has counted frames that did not fit (k_bin_finalize, sticky accum[5] mirrored into the
// host-mapped mailbox) that the host has not sized the buffers for yet: a plain host read, no copy, no sync
inline bool overflow_pending(const gsr_ctx* c)
{
    return c->mailbox && *reinterpret_cast<volatile const uint64_t*>(c->mailbox) != c->overflow_seen;
}

// Frames did not fit since the host last looked: wait for the stream, regrow the list for the largest of them and
// count them (*newly).  Those frames were not composited: a frame whose lists do not fit publishes no work items, so
// the framebuffer kept the image before it.  The caller decides whether one of them can still be rendered again.
int handle_overflow(gsr_ctx* c, uint64_t* newly)
{
    *newly = 0;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t acc[8];
    HIP_TRY(c, hipMemcpyAsync(acc, c->accum, sizeof acc, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (acc[5] == c->overflow_seen) return GSR_OK;
    *newly = acc[5] - c->overflow_seen;
    c->overflow_seen = acc[5];
    c->overflow_frames += *newly;
    const uint64_t need = acc[6];
    const uint64_t want = need + (need >> 2) + (1u << 20);
    if (want > 0xfffffff0ull) return fail(c, GSR_ERR_OVERFLOW, "bin list would need %llu entries", (unsigned long long)want);
    if (want > c->bin_capacity) {
        c->bin_capacity = (uint32_t)want;
        if (int r = dev_alloc(c, &c->bin_list, c->bin_capacity)) return r;
    }
    c->max_items = 0;
    return alloc_bins(c);
}

}  // namespace

extern "C" {

const char* gsr_last_error(gsr_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int gsr_create(gsr_ctx** out, const gsr_options* opt)
{
    if (!out) return fail(nullptr, GSR_ERR_ARG, "out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, GSR_ERR_NO_DEVICE, "no HIP device is visible (this library has no CPU path)");
    gsr_options o{};
    if (opt) o = *opt;
    if (o.device < 0 || o.device >= count) return fail(nullptr, GSR_ERR_ARG, "device %d out of range (%d visible)", o.device, count);
    if (o.width < 0 || o.height < 0 || o.width > 8192 || o.height > 8192) return fail(nullptr, GSR_ERR_ARG, "bad framebuffer size %dx%d (up to 8192)", o.width, o.height);
    if (!(o.early_out_eps >= 0.0f && o.early_out_eps < 1.0f)) return fail(nullptr, GSR_ERR_ARG, "early_out_eps must be in [0,1)");
    gsr_ctx* c = new gsr_ctx();
    c->device = o.device;
    c->opt = o;
    auto bail = [&](int code) {
        g_create_error = c->error;
        gsr_destroy(c);
        return code;
    };
#define CREATE_TRY(expr)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) { fail(c, GSR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return bail(GSR_ERR_HIP); } \
    } while (0)
    CREATE_TRY(hipSetDevice(c->device));
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) c->cu_count = cus;
    }
    CREATE_TRY(hipMalloc((void**)&c->fstate, sizeof(FrameState)));
    CREATE_TRY(hipMalloc((void**)&c->cam_dev, sizeof(CamParams)));
    CREATE_TRY(hipMalloc((void**)&c->slots, sizeof(int32_t) * 3 * FRAME_SLOTS * FRAME_SLOT_WORDS));   // render frames' set, two sets of the sort-only frames
    CREATE_TRY(hipMemset(c->slots, 0, sizeof(int32_t) * 3 * FRAME_SLOTS * FRAME_SLOT_WORDS));
    c->slots_need_init = true;   // (the first frame's enqueue launches the one-time initialisation: the stream exists by then)
    if (const char* e = getenv("GSR_NO_GRAPH")) c->graphs_enabled = atoi(e) == 0;
    if (const char* e = getenv("GSR_FUSE_COMBINE")) c->fuse_combine = atoi(e) != 0;   // A/B knob: 0 = separate k_combine launch
    if (const char* e = getenv("GSR_SATURATE")) c->saturate = atoi(e) != 0;           // A/B knob: 0 = no saturation skip
    c->items_by_size = (o.flags & GSR_FLAG_THROUGHPUT) ? 0 : 1;
    if (const char* e = getenv("GSR_ITEMS_BY_SIZE")) c->items_by_size = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("GSR_LONG_ITEMS")) c->long_items = atoi(e) != 0 ? 1 : 0; // pins the work-item length policy
    if (const char* e = getenv("GSR_LONG_TAU")) c->long_tau_env = (uint32_t)std::max(0L, atol(e));
    if (const char* e = getenv("GSR_BIN_ROUNDS")) c->bin_rounds_env = std::min(64L, std::max(0L, atol(e)));
    if (const char* e = getenv("GSR_BIN_BIG")) c->bin_big = (uint32_t)std::min(2, std::max(0, atoi(e)));
    if (const char* e = getenv("GSR_BIN_TWO_LEVEL")) c->bin_two_level_env = atoi(e) ? 1 : 0;
    if (const char* e = getenv("GSR_RECT_CARRY")) { c->rect_carry = atoi(e) != 0; c->rect_carry_bucket = atoi(e) == 2; }
    if (const char* e = getenv("GSR_BLEND_SUB")) c->blend_sub_env = atoi(e) == 2 ? 2 : atoi(e) == 1 ? 1 : 0;
    CREATE_TRY(hipMalloc((void**)&c->accum, 8 * sizeof(uint64_t)));
    CREATE_TRY(hipMemset(c->accum, 0, 8 * sizeof(uint64_t)));
    CREATE_TRY(hipHostMalloc((void**)&c->mailbox, 64, hipHostMallocMapped));
    memset(c->mailbox, 0, 64);
    reinterpret_cast<uint32_t*>(c->mailbox)[2] = 0xffffffffu;   // no frame has reported its largest bucket yet
    if (const char* e = getenv("GSR_SORT_ORDER")) c->sort_order = !strcmp(e, "lsd") ? 0 : !strcmp(e, "bucket") ? 1 : -1;
    CREATE_TRY(hipHostGetDevicePointer((void**)&c->mailbox_dev, c->mailbox, 0));
    CREATE_TRY(hipHostMalloc((void**)&c->fstate_host, sizeof(FrameState), hipHostMallocDefault));
    memset(c->fstate_host, 0, sizeof(FrameState));
    c->fstate_host->minmax[0] = 0x7fffffff;            // wasm/wasm.cpp:14
    c->fstate_host->minmax[1] = (int32_t)0x80000000;   // wasm/wasm.cpp:15
    CREATE_TRY(hipMemcpy(c->fstate, c->fstate_host, sizeof(FrameState), hipMemcpyHostToDevice));
    if (o.flags & GSR_FLAG_TIMING) {
        if (const char* e = getenv("GSR_TIMING_EVERY")) c->timing_every = (uint32_t)std::max(1L, atol(e));
        for (auto& set : c->evring)
            for (auto& e : set) CREATE_TRY(hipEventCreate(&e));
        c->ev_valid = true;
    }
#undef CREATE_TRY
    *out = c;
    if (o.width && o.height) {
        int r = gsr_resize(c, o.width, o.height);
        if (r) { *out = nullptr; return bail(r); }
        if (o.band_x1 > o.band_x0) {
            r = gsr_set_band(c, o.band_x0, o.band_x1);
            if (r) { *out = nullptr; return bail(r); }
        }
    }
    return GSR_OK;
}

int gsr_destroy(gsr_ctx* c)
{
    if (!c) return GSR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    dev_free(&c->px); dev_free(&c->py); dev_free(&c->pz);
    dev_free(&c->cov0); dev_free(&c->cov1); dev_free(&c->cov2); dev_free(&c->rgba);
    dev_free(&c->sh_r); dev_free(&c->sh_g); dev_free(&c->sh_b); dev_free(&c->shcol);
    dev_free(&c->rotv); dev_free(&c->sclv);
    dev_free(&c->depth); dev_free(&c->kept); dev_free(&c->kept_lane); dev_free(&c->koff); dev_free(&c->keys); dev_free(&c->keys_tmp); dev_free(&c->idx_tmp); dev_free(&c->depth_index);
    dev_free(&c->block_hist); dev_free(&c->rec); dev_free(&c->bbox); dev_free(&c->slots); dev_free(&c->rect_idx);
    dev_free(&c->bin_table); dev_free(&c->bin_total); dev_free(&c->bin_start); dev_free(&c->bin_start_pre); dev_free(&c->bin_list);
    dev_free(&c->cell_list); dev_free(&c->cell_total); dev_free(&c->cell_start); dev_free(&c->chunk_start); dev_free(&c->chunk_info); dev_free(&c->cell_wcnt); dev_free(&c->cell_table2);
    dev_free(&c->seg_start); dev_free(&c->bin_mask); dev_free(&c->items); dev_free(&c->partial); dev_free(&c->bin_rects); dev_free(&c->rect_tmp); dev_free(&c->sort_chunk_tab);
    drop_graph(c);
    dev_free(&c->cam_dev);
    dev_free(&c->fstate); dev_free(&c->accum); dev_free(&c->fb); dev_free(&c->fb8);
    if (c->fstate_host) (void)hipHostFree(c->fstate_host);
    if (c->mailbox) (void)hipHostFree(c->mailbox);
    for (auto& set : c->evring)
        for (auto& e : set) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->link_ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GSR_OK;
}

}  // extern "C"

namespace {

// (re)allocate everything sized by the splat count; clears SH and per-frame state
int alloc_scene(gsr_ctx* c, uint32_t n, bool with_rows)
{
    c->n = 0; c->have_frame = false; c->have_sort = false; c->have_rows = false;
    if (c->mailbox) reinterpret_cast<volatile uint32_t*>(c->mailbox)[2] = 0xffffffffu;   // a new scene: LSD order until a frame reports
    c->sh_count = 0; c->band[0] = c->band[1] = c->band[2] = -1;
    dev_free(&c->sh_r); dev_free(&c->sh_g); dev_free(&c->sh_b); dev_free(&c->shcol);
    dev_free(&c->rotv); dev_free(&c->sclv);
    int r;
    if ((r = dev_alloc(c, &c->px, n)) || (r = dev_alloc(c, &c->py, n)) || (r = dev_alloc(c, &c->pz, n)) ||
        (r = dev_alloc(c, &c->cov0, n)) || (r = dev_alloc(c, &c->cov1, n)) || (r = dev_alloc(c, &c->cov2, n)) ||
        (r = dev_alloc(c, &c->rgba, n)) || (r = dev_alloc(c, &c->depth, n)) || (r = dev_alloc(c, &c->keys, n)) ||
        (r = dev_alloc(c, &c->keys_tmp, n)) || (r = dev_alloc(c, &c->idx_tmp, n)) ||
        (r = dev_alloc(c, &c->depth_index, n)) || (r = dev_alloc(c, &c->rec, n)) ||
        (r = dev_alloc(c, &c->bin_rects, n)) || (r = dev_alloc(c, &c->rect_idx, n)) || (r = dev_alloc(c, &c->rect_tmp, n)) ||
        (r = dev_alloc(c, &c->sort_chunk_tab, 4 * ((size_t)n / 4096 + 260))) ||
        (r = dev_alloc(c, &c->kept, (size_t)n / PROJ_THREADS + 1)) || (r = dev_alloc(c, &c->kept_lane, n)) ||
        (r = dev_alloc(c, &c->koff, (size_t)n / PROJ_THREADS + 2)))
        return r;
    if (with_rows && ((r = dev_alloc(c, &c->rotv, n)) || (r = dev_alloc(c, &c->sclv, n)))) return r;
    // keys per radix workgroup: the scatter stores runs of keys_per_block / 2^bits keys, so larger scenes take larger
    // blocks (longer runs) while small ones keep enough workgroups to fill the chip.  Measured at 20 M splats, the two
    // scatters: 135 + 126 us with 2048 keys, 99 + 98 us with 4096, 113 + 116 us with 8192 (96 KiB of LDS: one
    // workgroup per CU, nothing overlaps its load and store phases).
    c->sort_kpb = n <= (3u << 20) ? 2048 : 4096;
    if (const char* e = getenv("GSR_SORT_KPB")) {   // tuning knob: 2048, 4096 or 8192
        const long v = atol(e);
        if (v == 2048 || v == 4096 || v == 8192) c->sort_kpb = (uint32_t)v;
    }
    c->sort_blocks = (n + c->sort_kpb - 1) / c->sort_kpb;
    return dev_alloc(c, &c->block_hist, (size_t)std::max(c->sort_blocks, 1u) * RADIX_HI_BINS);
}

SceneDev scene_dev(gsr_ctx* c) { return SceneDev{c->px, c->py, c->pz, c->cov0, c->cov1, c->cov2, c->rgba, c->rotv, c->sclv}; }

int need_rows(gsr_ctx* c)
{
    if (!c->have_rows) return fail(c, GSR_ERR_ARG, "scene transforms need a scene built with gsr_set_scene_rows");
    HIP_TRY(c, hipSetDevice(c->device));
    c->have_frame = false; c->have_sort = false;
    return GSR_OK;
}

}  // namespace

extern "C" {

int gsr_set_scene(gsr_ctx* c, const uint32_t* data, const float* positions, uint32_t n)
{
    if (!c) return GSR_ERR_ARG;
    if (n && (!data || !positions)) return fail(c, GSR_ERR_ARG, "data/positions is NULL");
    if (n > 0x7fffffffu / 8) return fail(c, GSR_ERR_ARG, "too many splats");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int r;
    if ((r = alloc_scene(c, n, false))) return r;
    if (n) {
        uint32_t* d_data = nullptr; float* d_pos = nullptr; uint32_t* d_flag = nullptr;
        if ((r = dev_alloc(c, &d_data, (size_t)n * 8)) || (r = dev_alloc(c, &d_pos, (size_t)n * 3)) || (r = dev_alloc(c, &d_flag, 1))) {
            dev_free(&d_data); dev_free(&d_pos); dev_free(&d_flag);
            return r;
        }
        hipError_t e1 = hipMemcpyAsync(d_data, data, (size_t)n * 32, hipMemcpyHostToDevice, c->stream);
        hipError_t e2 = hipMemcpyAsync(d_pos, positions, (size_t)n * 12, hipMemcpyHostToDevice, c->stream);
        hipError_t e3 = hipMemsetAsync(d_flag, 0, 4, c->stream);
        launch_repack_scene(d_data, d_pos, n, c->px, c->py, c->pz, c->cov0, c->cov1, c->cov2, c->rgba, d_flag, c->stream);
        uint32_t flag = 0;
        hipError_t e4 = hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, c->stream);
        hipError_t e5 = hipStreamSynchronize(c->stream);
        dev_free(&d_data); dev_free(&d_pos); dev_free(&d_flag);
        for (hipError_t e : {e1, e2, e3, e4, e5, hipGetLastError()})
            if (e != hipSuccess) return fail(c, GSR_ERR_HIP, "scene upload failed: %s", hipGetErrorString(e));
        if (flag) return fail(c, GSR_ERR_SCENE, "positions differ from data words 0..2 (Scene.ts:141-143 keeps them equal)");
    }
    c->n = n;
    c->bin_capacity = 0;
    return alloc_bins(c);
}

int gsr_set_scene_rows(gsr_ctx* c, const uint8_t* rows, uint32_t n)
{
    if (!c) return GSR_ERR_ARG;
    if (n && !rows) return fail(c, GSR_ERR_ARG, "rows is NULL");
    if (n > 0x7fffffffu / 8) return fail(c, GSR_ERR_ARG, "too many splats");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int r;
    if ((r = alloc_scene(c, n, true))) return r;
    if (n) {
        uint8_t* d_rows = nullptr;
        if ((r = dev_alloc(c, &d_rows, (size_t)n * 32))) return r;
        hipError_t e1 = hipMemcpyAsync(d_rows, rows, (size_t)n * 32, hipMemcpyHostToDevice, c->stream);
        launch_build_scene(d_rows, n, scene_dev(c), c->stream);
        hipError_t e2 = hipStreamSynchronize(c->stream);
        dev_free(&d_rows);
        for (hipError_t e : {e1, e2, hipGetLastError()})
            if (e != hipSuccess) return fail(c, GSR_ERR_HIP, "scene build failed: %s", hipGetErrorString(e));
    }
    c->n = n;
    c->have_rows = true;
    c->bin_capacity = 0;
    return alloc_bins(c);
}

int gsr_scene_translate(gsr_ctx* c, const double* t)
{
    if (!c || !t) return GSR_ERR_ARG;
    if (int r = need_rows(c)) return r;
    launch_scene_translate(c->n, scene_dev(c), t, c->stream);
    HIP_TRY(c, hipGetLastError());
    return GSR_OK;
}

int gsr_scene_rotate(gsr_ctx* c, const double* q)
{
    if (!c || !q) return GSR_ERR_ARG;
    if (int r = need_rows(c)) return r;
    launch_scene_rotate(c->n, scene_dev(c), q, c->stream);
    HIP_TRY(c, hipGetLastError());
    return GSR_OK;
}

int gsr_scene_scale(gsr_ctx* c, const double* sv)
{
    if (!c || !sv) return GSR_ERR_ARG;
    if (int r = need_rows(c)) return r;
    launch_scene_scale(c->n, scene_dev(c), sv, c->stream);
    HIP_TRY(c, hipGetLastError());
    return GSR_OK;
}

int gsr_scene_limit_box(gsr_ctx* c, const double* box, uint32_t* new_count)
{
    if (!c || !box) return GSR_ERR_ARG;
    if (box[0] >= box[1]) return fail(c, GSR_ERR_ARG, "xMin (%g) must be smaller than xMax (%g)", box[0], box[1]);   // Scene.ts:308-316
    if (box[2] >= box[3]) return fail(c, GSR_ERR_ARG, "yMin (%g) must be smaller than yMax (%g)", box[2], box[3]);
    if (box[4] >= box[5]) return fail(c, GSR_ERR_ARG, "zMin (%g) must be smaller than zMax (%g)", box[4], box[5]);
    if (int r = need_rows(c)) return r;
    const uint32_t n = c->n;
    uint32_t kept = 0;
    if (n) {
        gsr_ctx tmp_holder;  // only its pointer fields are used, as a second SoA
        gsr_ctx* d = &tmp_holder;
        uint32_t* block_count = nullptr;
        uint32_t* total = nullptr;
        auto free_tmp = [&]() {
            dev_free(&d->px); dev_free(&d->py); dev_free(&d->pz); dev_free(&d->cov0); dev_free(&d->cov1); dev_free(&d->cov2);
            dev_free(&d->rgba); dev_free(&d->rotv); dev_free(&d->sclv); dev_free(&block_count); dev_free(&total);
        };
        int r;
        if ((r = dev_alloc(c, &d->px, n)) || (r = dev_alloc(c, &d->py, n)) || (r = dev_alloc(c, &d->pz, n)) ||
            (r = dev_alloc(c, &d->cov0, n)) || (r = dev_alloc(c, &d->cov1, n)) || (r = dev_alloc(c, &d->cov2, n)) ||
            (r = dev_alloc(c, &d->rgba, n)) || (r = dev_alloc(c, &d->rotv, n)) || (r = dev_alloc(c, &d->sclv, n)) ||
            (r = dev_alloc(c, &block_count, (n + 1023) / 1024)) || (r = dev_alloc(c, &total, 1))) {
            free_tmp();
            return r;
        }
        launch_scene_limit_box(n, scene_dev(c), scene_dev(d), box, block_count, total, c->stream);
        hipError_t e1 = hipMemcpyAsync(&kept, total, 4, hipMemcpyDeviceToHost, c->stream);
        hipError_t e2 = hipStreamSynchronize(c->stream);
        const hipError_t e3 = hipGetLastError();
        if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) {
            std::swap(c->px, d->px); std::swap(c->py, d->py); std::swap(c->pz, d->pz);
            std::swap(c->cov0, d->cov0); std::swap(c->cov1, d->cov1); std::swap(c->cov2, d->cov2); std::swap(c->rgba, d->rgba);
            std::swap(c->rotv, d->rotv); std::swap(c->sclv, d->sclv);
        }
        free_tmp();
        for (hipError_t e : {e1, e2, e3})
            if (e != hipSuccess) return fail(c, GSR_ERR_HIP, "limitBox failed: %s", hipGetErrorString(e));
        c->n = kept;   // arrays keep their old capacity; per-frame buffers sized for the old count still fit
        c->sort_blocks = (kept + c->sort_kpb - 1) / c->sort_kpb;
        c->bin_blocks = (kept + 2048u * c->bin_rounds - 1u) / (2048u * c->bin_rounds);
        // The compaction renumbers the splats, so SH rows (indexed by splat - (bandsIndices[0] + 1)) and the band
        // thresholds no longer belong to them: the SH state is dropped and the scene falls back to its rgba8 colours
        // until gsr_set_scene_sh is called again.  (Scene.limitBox, Scene.ts:307-366, leaves shs_rgb / bandsIndices
        // untouched, i.e. stale; a host that wants SH after limitBox re-packs them for the kept splats.)
        c->sh_count = 0; c->band[0] = c->band[1] = c->band[2] = -1;
        dev_free(&c->sh_r); dev_free(&c->sh_g); dev_free(&c->sh_b); dev_free(&c->shcol);
    }
    if (new_count) *new_count = kept;
    return GSR_OK;
}

int gsr_read_scene(gsr_ctx* c, uint32_t* data, float* positions, float* rotations, float* scales, uint32_t* count)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t n = c->n;
    if (count) *count = n;
    if (!data && !positions && !rotations && !scales) return GSR_OK;  // count only: nothing to copy
    if ((rotations || scales) && !c->have_rows) return fail(c, GSR_ERR_ARG, "rotations/scales exist only for scenes built with gsr_set_scene_rows");
    std::vector<float> x(n), y(n), z(n);
    std::vector<uint32_t> c0, c1, c2, cw;
    std::vector<float4> rv, sv;
    HIP_TRY(c, hipMemcpyAsync(x.data(), c->px, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(y.data(), c->py, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(z.data(), c->pz, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (data) {
        c0.resize(n); c1.resize(n); c2.resize(n); cw.resize(n);
        HIP_TRY(c, hipMemcpyAsync(c0.data(), c->cov0, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c1.data(), c->cov1, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c2.data(), c->cov2, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(cw.data(), c->rgba, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (rotations) { rv.resize(n); HIP_TRY(c, hipMemcpyAsync(rv.data(), c->rotv, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream)); }
    if (scales) { sv.resize(n); HIP_TRY(c, hipMemcpyAsync(sv.data(), c->sclv, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream)); }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; i++) {
        if (positions) { positions[3 * (size_t)i] = x[i]; positions[3 * (size_t)i + 1] = y[i]; positions[3 * (size_t)i + 2] = z[i]; }
        if (data) {
            uint32_t* d = data + 8 * (size_t)i;
            memcpy(&d[0], &x[i], 4); memcpy(&d[1], &y[i], 4); memcpy(&d[2], &z[i], 4);
            d[3] = 0; d[4] = c0[i]; d[5] = c1[i]; d[6] = c2[i]; d[7] = cw[i];
        }
        if (rotations) { float* r = rotations + 4 * (size_t)i; r[0] = rv[i].x; r[1] = rv[i].y; r[2] = rv[i].z; r[3] = rv[i].w; }
        if (scales) { float* q = scales + 3 * (size_t)i; q[0] = sv[i].x; q[1] = sv[i].y; q[2] = sv[i].z; }
    }
    return GSR_OK;
}

int gsr_scene_count(gsr_ctx* c, uint32_t* count)
{
    if (!c || !count) return GSR_ERR_ARG;
    *count = c->n;
    return GSR_OK;
}

int gsr_set_scene_sh(gsr_ctx* c, const uint32_t* sh_r, const uint32_t* sh_g, const uint32_t* sh_b, uint32_t sh_count,
                     const int32_t* band_index)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sh_count = 0; c->band[0] = c->band[1] = c->band[2] = -1;
    c->have_frame = false;
    if (!sh_count) return GSR_OK;
    if (!sh_r || !sh_g || !sh_b || !band_index) return fail(c, GSR_ERR_ARG, "SH texture or band_index pointer is NULL");
    if (band_index[0] < -1 || (uint64_t)(band_index[0] + 1) + sh_count != c->n)
        return fail(c, GSR_ERR_SCENE, "sh_count (%u) must be vertexCount (%u) - (bandsIndices[0] + 1) (%d)", sh_count, c->n,
                    band_index[0] + 1);
    int r;
    if ((r = dev_alloc(c, &c->sh_r, (size_t)sh_count * 8)) || (r = dev_alloc(c, &c->sh_g, (size_t)sh_count * 8)) ||
        (r = dev_alloc(c, &c->sh_b, (size_t)sh_count * 8)) || (r = dev_alloc(c, &c->shcol, (size_t)c->n)))
        return r;
    HIP_TRY(c, hipMemcpyAsync(c->sh_r, sh_r, (size_t)sh_count * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sh_g, sh_g, (size_t)sh_count * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sh_b, sh_b, (size_t)sh_count * 32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->shcol, 0, (size_t)c->n * sizeof(float4), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sh_count = sh_count;
    c->band[0] = band_index[0]; c->band[1] = band_index[1]; c->band[2] = band_index[2];
    return GSR_OK;
}

int gsr_resize(gsr_ctx* c, int32_t w, int32_t h)
{
    if (!c) return GSR_ERR_ARG;
    if (w <= 0 || h <= 0 || w > 8192 || h > 8192) return fail(c, GSR_ERR_ARG, "bad framebuffer size %dx%d (1..8192)", w, h);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->comm && (w != c->W || h != c->H)) comm_release(c);   // slabs and band edges belong to the old size: join again
    c->W = w; c->H = h;
    c->band_x0 = c->band_x1 = 0;
    c->have_frame = false;
    if (int r = alloc_fb(c)) return r;
    return alloc_bins(c);
}

int gsr_set_band(gsr_ctx* c, int32_t x0, int32_t x1)
{
    if (!c) return GSR_ERR_ARG;
    if (x0 == 0 && x1 == 0) { c->band_x0 = c->band_x1 = 0; return alloc_bins(c); }
    if (x0 < 0 || x1 <= x0 || x0 % BIN_PX) return fail(c, GSR_ERR_ARG, "band [%d,%d) must start on a multiple of %d", x0, x1, BIN_PX);
    if (x1 > c->W) x1 = c->W;
    if (x0 >= c->W) return fail(c, GSR_ERR_ARG, "band starts outside the image");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->band_x0 = x0; c->band_x1 = x1;
    launch_clear_fb(c->fb, c->W, c->H, c->stream);
    return alloc_bins(c);
}

int gsr_set_camera(gsr_ctx* c, const float* view, const float* proj, const float* vp, float fx, float fy)
{
    if (!c) return GSR_ERR_ARG;
    if (!view || !proj || !vp) return fail(c, GSR_ERR_ARG, "matrix pointer is NULL");
    memcpy(c->cam.view, view, 64);
    memcpy(c->cam.proj, proj, 64);
    c->cam.vp2 = vp[2]; c->cam.vp6 = vp[6]; c->cam.vp10 = vp[10];
    c->cam.fx = fx; c->cam.fy = fy;
    c->have_cam = true;
    return GSR_OK;
}

int gsr_set_depth_fade(gsr_ctx* c, int32_t use_depth_fade, float depth_fade)
{
    if (!c) return GSR_ERR_ARG;
    c->cam.use_fade = use_depth_fade ? 1 : 0;
    c->cam.fade = depth_fade;
    return GSR_OK;
}

int gsr_render_async(gsr_ctx* c)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return enqueue_frame(c, true);
}

// wait for the context's stream; if frames overflowed, regrow and render the last frame again (when it was one of
// them); lost frames are added to dropped_unreported, which gsr_sync turns into one GSR_ERR_OVERFLOW
static int sync_and_repair(gsr_ctx* c)
{
    if (int r = finish_frame(c)) return r;
    bool last_ov = false;
    if (c->have_frame && c->ev_render) {
        if (int r = check_frame_words(c, &last_ov)) return r;
    }
    if (overflow_pending(c)) {
        uint64_t newly = 0;
        if (int r = handle_overflow(c, &newly)) return r;   // buffers regrown for the largest frame seen
        if (last_ov && newly) {  // the last frame is one of them and nothing has been enqueued behind it: render it again
            if (int r = enqueue_frame(c, true)) return r;
            if (int r = finish_frame(c)) return r;
            bool again = false;
            if (int r = check_frame_words(c, &again)) return r;
            if (again) return fail(c, GSR_ERR_OVERFLOW, "bin list overflow after regrowth");
            newly -= 1;
        }
        c->dropped_frames += newly;
        c->dropped_unreported += newly;
    }
    return GSR_OK;
}

int gsr_sync(gsr_ctx* c)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int r = sync_and_repair(c)) return r;
    if (c->comm_stream) HIP_TRY(c, hipStreamSynchronize(c->comm_stream));   // the frame exchange, if one is in flight
    if (c->dropped_unreported) {
        const unsigned long long k = c->dropped_unreported;
        c->dropped_unreported = 0;
        return fail(c, GSR_ERR_OVERFLOW,
                    "%llu asynchronous frame(s) were not composited: their bin lists did not fit and later frames had already been "
                    "enqueued (the framebuffer kept the preceding image for them); the lists have been regrown, the context stays usable",
                    k);
    }
    return GSR_OK;
}

int gsr_overflow_pending(gsr_ctx* c) { return c && overflow_pending(c) ? 1 : 0; }

int gsr_set_list_capacity(gsr_ctx* c, uint32_t entries)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->bin_capacity = std::max<uint32_t>(entries, 1024u);
    if (int r = dev_alloc(c, &c->bin_list, c->bin_capacity)) return r;
    c->max_items = 0;
    drop_graph(c);
    return alloc_bins(c);
}

int gsr_render(gsr_ctx* c)
{
    if (int r = gsr_render_async(c)) return r;
    c->ev_render = true;
    return gsr_sync(c);
}

int gsr_sort(gsr_ctx* c)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int r = enqueue_frame(c, false)) return r;
    return finish_frame(c);
}

int gsr_read_depth_index(gsr_ctx* c, uint32_t* out)
{
    if (!c || !out) return c ? fail(c, GSR_ERR_ARG, "out is NULL") : GSR_ERR_ARG;
    if (!c->have_sort) return fail(c, GSR_ERR_ARG, "no sort has run yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->sort_culled) {  // the band's frame sorted only its survivors: the caller wants the whole permutation
        if (int r = enqueue_frame(c, false)) return r;
        if (int r = finish_frame(c)) return r;
    }
    HIP_TRY(c, hipMemcpyAsync(out, c->depth_index, (size_t)c->n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GSR_OK;
}

int gsr_read_pixels_rgba32f(gsr_ctx* c, float* out)
{
    if (!c || !out) return c ? fail(c, GSR_ERR_ARG, "out is NULL") : GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, c->fb, (size_t)c->W * c->H * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GSR_OK;
}

int gsr_read_pixels_rgba8(gsr_ctx* c, uint8_t* out)
{
    if (!c || !out) return c ? fail(c, GSR_ERR_ARG, "out is NULL") : GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t np = (uint32_t)c->W * (uint32_t)c->H;
    launch_to_rgba8(c->fb, c->fb8, np, c->stream);
    HIP_TRY(c, hipMemcpyAsync(out, c->fb8, (size_t)np * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GSR_OK;
}

int gsr_get_timings(gsr_ctx* c, gsr_timings* out)
{
    if (!c || !out) return GSR_ERR_ARG;
    if (c->ev_recorded) { if (int r = finish_frame(c)) return r; }
    *out = c->tm;
    out->overflow_frames = c->overflow_frames;
    out->dropped_frames = c->dropped_frames;
    return GSR_OK;
}

int gsr_set_timing_interval(gsr_ctx* c, uint32_t every)
{
    if (!c || !every) return c ? fail(c, GSR_ERR_ARG, "timing interval must be >= 1") : GSR_ERR_ARG;
    c->timing_every = every;
    c->frame_no = 0;
    return GSR_OK;
}

int gsr_reset_timings(gsr_ctx* c)
{
    if (!c) return GSR_ERR_ARG;
    if (c->ev_recorded) { if (int r = finish_frame(c)) return r; }
    const uint64_t v = c->tm.visible, b = c->tm.bin_entries, d = c->tm.tile_entries;
    c->tm = gsr_timings{};
    HIP_TRY(c, hipMemsetAsync(c->accum, 0, 4 * sizeof(uint64_t), c->stream));
    c->tm.visible = v; c->tm.bin_entries = b; c->tm.tile_entries = d; c->tm.n = c->n;
    c->frame_no = 0;  // the sampling restarts: the next frame carries the stage events
    return GSR_OK;
}

int gsr_read_keys(gsr_ctx* c, uint32_t* keys, int32_t* minmax)
{
    if (!c) return GSR_ERR_ARG;
    if (!c->have_sort) return fail(c, GSR_ERR_ARG, "no sort has run yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->sort_culled) {
        if (int r = enqueue_frame(c, false)) return r;
        if (int r = finish_frame(c)) return r;
    }
    if (keys) HIP_TRY(c, hipMemcpyAsync(keys, c->keys, (size_t)c->n * 4, hipMemcpyDeviceToHost, c->stream));
    if (minmax) HIP_TRY(c, hipMemcpyAsync(minmax, c->fstate->minmax, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GSR_OK;
}

int gsr_read_records(gsr_ctx* c, float* rec, int32_t* bbox)
{
    if (!c) return GSR_ERR_ARG;
    if (!c->have_frame) return fail(c, GSR_ERR_ARG, "no frame has been rendered yet");
    HIP_TRY(c, hipSetDevice(c->device));
    if (rec) HIP_TRY(c, hipMemcpyAsync(rec, c->rec, (size_t)c->n * 32, hipMemcpyDeviceToHost, c->stream));
    std::vector<uint2> tmp;
    if (bbox) {
        // the pixel boxes are not part of a frame (no kernel reads them): project once more for the frame's camera, records and
        // boxes only (k_project_key, do_project == 2: the same arithmetic, so the same records)
        tmp.resize(c->n);
        if (c->n) {
            if (int r = dev_alloc(c, &c->bbox, c->n)) return r;
            SceneSoA sc{c->px, c->py, c->pz, c->cov0, c->cov1, c->cov2, c->rgba, c->sh_r, c->sh_g, c->sh_b, c->shcol};
            ProjectLaunch again{sc, c->n, c->cam_frame, 2, c->depth, c->slots, c->rec, c->bbox, c->rect_idx, &c->fstate->overflow, nullptr, nullptr, {}};
            launch_project_key(again, c->stream);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->bbox, (size_t)c->n * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            dev_free(&c->bbox);
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bbox)
        for (uint32_t i = 0; i < c->n; i++) {
            bbox[4 * (size_t)i + 0] = (int32_t)(tmp[i].x & 0xffff);
            bbox[4 * (size_t)i + 1] = (int32_t)(tmp[i].y & 0xffff);
            bbox[4 * (size_t)i + 2] = (int32_t)(tmp[i].x >> 16);
            bbox[4 * (size_t)i + 3] = (int32_t)(tmp[i].y >> 16);
        }
    return GSR_OK;
}

int gsr_read_sh_colors(gsr_ctx* c, float* rgba)
{
    if (!c || !rgba) return c ? fail(c, GSR_ERR_ARG, "out is NULL") : GSR_ERR_ARG;
    if (!c->have_frame || !c->sh_count) return fail(c, GSR_ERR_ARG, "no frame rendered with SH colours yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(rgba, c->shcol, (size_t)c->n * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GSR_OK;
}

int gsr_read_work_items(gsr_ctx* c, uint32_t* out)
{
    if (!c || !out) return c ? fail(c, GSR_ERR_ARG, "out is NULL") : GSR_ERR_ARG;
    if (!c->have_frame) return fail(c, GSR_ERR_ARG, "no frame has been rendered yet");
    HIP_TRY(c, hipSetDevice(c->device));
    static_assert(offsetof(FrameState, n_items) == offsetof(FrameState, seg_len) + 4 && offsetof(FrameState, spec) == offsetof(FrameState, seg_len) + 8,
                  "seg_len, n_items, spec are read through one pointer");
    HIP_TRY(c, hipMemcpyAsync(out, &c->fstate->seg_len, 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const BinGrid g = make_grid(c);
    out[3] = std::min(c->blend_sub, 2u);
    out[4] = (uint32_t)((g.bx_hi - g.bx_lo) * g.nby);
    return GSR_OK;
}

int gsr_read_bin_totals(gsr_ctx* c, uint32_t* out, int32_t* nbx, int32_t* nby)
{
    if (!c || !out) return c ? fail(c, GSR_ERR_ARG, "out is NULL") : GSR_ERR_ARG;
    if (!c->have_frame) return fail(c, GSR_ERR_ARG, "no frame has been rendered yet");
    const BinGrid g = make_grid(c);
    const int w = g.bx_hi - g.bx_lo;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, c->bin_total, (size_t)w * g.nby * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (nbx) *nbx = w;
    if (nby) *nby = g.nby;
    return GSR_OK;
}

int gsr_read_bin_lists(gsr_ctx* c, uint32_t* starts, uint32_t* list, uint64_t list_words)
{
    if (!c || !starts) return c ? fail(c, GSR_ERR_ARG, "starts is NULL") : GSR_ERR_ARG;
    if (!c->have_frame) return fail(c, GSR_ERR_ARG, "no frame has been rendered yet");
    const BinGrid g = make_grid(c);
    const size_t nbins = (size_t)(g.bx_hi - g.bx_lo) * g.nby;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(starts, c->bin_start, (nbins + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint64_t total = starts[nbins];
    if (list) {
        if (total > list_words || total > c->bin_capacity) return fail(c, GSR_ERR_ARG, "the frame's lists hold %llu entries", (unsigned long long)total);
        HIP_TRY(c, hipMemcpyAsync(list, c->bin_list, total * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return GSR_OK;
}

int gsr_convert_rgba8_async(gsr_ctx* c)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    launch_to_rgba8(c->fb, c->fb8, (uint32_t)c->W * (uint32_t)c->H, c->stream);
    HIP_TRY(c, hipGetLastError());
    return GSR_OK;
}

void* gsr_framebuffer8_device_ptr(gsr_ctx* c) { return c ? (void*)c->fb8 : nullptr; }

int gsr_pack_band_rgba8_async(gsr_ctx* c, void* slab, int32_t slab_width_px)
{
    if (!c) return GSR_ERR_ARG;
    if (!slab || !c->fb) return fail(c, GSR_ERR_ARG, "gsr_pack_band_rgba8_async: no slab / nothing rendered yet");
    const BinGrid g = make_grid(c);
    const int x0 = g.bx_lo * BIN_PX, x1 = std::min(g.bx_hi * BIN_PX, c->W);
    if (slab_width_px < x1 - x0) return fail(c, GSR_ERR_ARG, "gsr_pack_band_rgba8_async: slab narrower than the band");
    HIP_TRY(c, hipSetDevice(c->device));
    launch_pack_band_rgba8(c->fb, (uint32_t*)slab, c->W, c->H, x0, x1, slab_width_px, c->stream);
    HIP_TRY(c, hipGetLastError());
    return GSR_OK;
}

int gsr_unpack_slabs_rgba8_async(gsr_ctx* c, const void* gathered, void* image, int32_t slab_width_px, int32_t world,
                                 const int32_t* x0, const int32_t* x1, void* stream)
{
    if (!c) return GSR_ERR_ARG;
    if (!gathered || !image || !x0 || !x1 || world < 1 || world > MAX_SLABS)
        return fail(c, GSR_ERR_ARG, "gsr_unpack_slabs_rgba8_async: bad argument (1 <= world <= 16)");
    SlabEdges e{};
    for (int q = 0; q < world; q++) {
        if (x0[q] < 0 || x1[q] > c->W || x1[q] - x0[q] > slab_width_px)
            return fail(c, GSR_ERR_ARG, "gsr_unpack_slabs_rgba8_async: band outside the image or wider than the slab");
        e.x0[q] = x0[q]; e.x1[q] = x1[q];
    }
    HIP_TRY(c, hipSetDevice(c->device));
    launch_unpack_slabs_rgba8((const uint32_t*)gathered, (uint32_t*)image, c->W, c->H, slab_width_px, world, e, (hipStream_t)stream);
    HIP_TRY(c, hipGetLastError());
    return GSR_OK;
}

void* gsr_framebuffer_device_ptr(gsr_ctx* c) { return c ? (void*)c->fb : nullptr; }
void* gsr_stream_handle(gsr_ctx* c) { return c ? (void*)c->stream : nullptr; }

int gsr_stream_order(gsr_ctx* c, void* other_stream, int32_t ctx_waits)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipEvent_t& ev = c->link_ev[ctx_waits ? 1 : 0];
    if (!ev) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipStream_t from = ctx_waits ? (hipStream_t)other_stream : c->stream;
    hipStream_t to = ctx_waits ? c->stream : (hipStream_t)other_stream;
    HIP_TRY(c, hipEventRecord(ev, from));
    HIP_TRY(c, hipStreamWaitEvent(to, ev, 0));
    return GSR_OK;
}

int gsr_device_info(gsr_ctx* c, char* name, int32_t name_len, int32_t* cus, int32_t* clock_khz)
{
    if (!c) return GSR_ERR_ARG;
    hipDeviceProp_t p;
    HIP_TRY(c, hipGetDeviceProperties(&p, c->device));
    // (the marketing name comes from libdrm's amdgpu.ids and is empty where that file is missing)
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", p.name[0] ? p.name : "AMD GPU", p.gcnArchName);
    if (cus) *cus = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    return GSR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Multi-GPU: the framebuffer all-gather of SURVEY 8(e), issued by the library itself.  One process per GPU; every
// rank renders its band of tile columns and the RGBA8 slabs are exchanged with ONE ncclAllGather over xGMI (RCCL),
// so the per-frame path needs no Python and no torch: renderer.render(scene, camera) on a Node host returns the full
// frame.  RCCL is opened at run time (dlopen "librccl.so.1": in a process that already holds one -- a torch build
// bundles its own -- the loader hands back that copy, so a process never ends up with two), which also keeps
// libgsplat_hip.so loadable on single-GPU hosts without RCCL installed.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool ok = false;
};

RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { api.error = std::string("RCCL is not available: ") + dlerror(); return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
        if (!api.ok) api.error = "librccl.so lacks an expected entry point";
    });
    return api;
}

#define RCCL_TRY(c, expr)                                                                              \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return fail((c), GSR_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

void comm_release(gsr_ctx* c)
{
    // A leader that leaves the group (or is destroyed) before the contexts that borrowed its communicator and exchange stream:
    // they are detached first, while both still exist -- afterwards they are plain contexts that have to join again, instead of
    // holders of a destroyed stream (a garbage-collected host destroys contexts in any order).
    while (!c->comm_followers.empty()) comm_release(c->comm_followers.back());
    if (c->comm_leader) {
        auto& fl = c->comm_leader->comm_followers;
        fl.erase(std::remove(fl.begin(), fl.end(), c), fl.end());
        c->comm_leader = nullptr;
    }
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm && c->comm_owned && rccl().ok) (void)rccl().CommDestroy(c->comm);
    c->comm = nullptr;
    c->comm_fn = nullptr; c->comm_fn_user = nullptr;
    if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
    if (c->ev_slab_free) (void)hipEventDestroy(c->ev_slab_free);
    c->ev_packed = c->ev_slab_free = nullptr;
    if (c->comm_stream && c->comm_owned) (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
    c->comm_owned = true;
    dev_free(&c->slab); dev_free(&c->gathered); dev_free(&c->frame8);
    c->comm_world = 0; c->frame8_valid = false;
}

}  // namespace

extern "C" {

int gsr_comm_unique_id(uint8_t* id)
{
    if (!id) return GSR_ERR_ARG;
    static_assert(GSR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is passed through as bytes");
    if (!rccl().ok) return fail(nullptr, GSR_ERR_COMM, "%s", rccl().error.c_str());
    ncclUniqueId u;
    const ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, GSR_ERR_COMM, "ncclGetUniqueId failed: %s", rccl().GetErrorString(r));
    memcpy(id, u.internal, GSR_COMM_ID_BYTES);
    return GSR_OK;
}

// everything of gsr_comm_init but the communicator: argument checks, the context's band, slab / gathered / frame buffers,
// the exchange stream (its own, or `shared_stream`) and the two ordering events
static int comm_setup(gsr_ctx* c, const char* who, int32_t rank, int32_t world, const int32_t* x0, const int32_t* x1, hipStream_t shared_stream)
{
    if (!x0 || !x1 || world < 1 || world > MAX_SLABS || rank < 0 || rank >= world)
        return fail(c, GSR_ERR_ARG, "%s: bad argument (1 <= world <= %d, 0 <= rank < world)", who, MAX_SLABS);
    if (!c->W || !c->H) return fail(c, GSR_ERR_ARG, "%s: set the framebuffer size first", who);
    int sw = BIN_PX;
    for (int q = 0; q < world; q++) {
        // every rank must hold the same edges: whole 32-px bin columns, contiguous, covering the image
        const int want0 = q ? x1[q - 1] : 0;
        if (x0[q] != want0 || x1[q] <= x0[q] || x0[q] % BIN_PX || (x1[q] % BIN_PX && x1[q] != c->W) || x1[q] > c->W)
            return fail(c, GSR_ERR_ARG, "%s: band %d = [%d,%d) (bands are contiguous runs of whole %d-px columns)", who, q, x0[q], x1[q], BIN_PX);
        sw = std::max(sw, x1[q] - x0[q]);
    }
    if (x1[world - 1] != c->W) return fail(c, GSR_ERR_ARG, "%s: the bands end at %d, the image is %d wide", who, x1[world - 1], c->W);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    comm_release(c);
    if (int r = gsr_set_band(c, world == 1 ? 0 : x0[rank], world == 1 ? 0 : x1[rank])) return r;
    c->slab_w = sw;
    for (int q = 0; q < world; q++) { c->comm_edges.x0[q] = x0[q]; c->comm_edges.x1[q] = x1[q]; }
    // (a slab = the band's pixels + SLAB_FLAG_WORDS words "this band was not composited"; the assembled frame is followed by
    //  the word that collects those flags: k_pack_band_rgba8 / k_unpack_slabs_rgba8)
    const size_t slab_px = (size_t)sw * c->H + SLAB_FLAG_WORDS;
    int r;
    if ((r = dev_alloc(c, &c->slab, slab_px)) || (r = dev_alloc(c, &c->gathered, slab_px * world)) ||
        (r = dev_alloc(c, &c->frame8, (size_t)c->W * c->H + SLAB_FLAG_WORDS)))
        return r;
    HIP_TRY(c, hipMemsetAsync(c->slab, 0, slab_px * 4, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (shared_stream) { c->comm_stream = shared_stream; c->comm_owned = false; }
    else HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_slab_free, hipEventDisableTiming));
    c->comm_rank = rank; c->comm_world = world;
    return GSR_OK;
}

int gsr_comm_init(gsr_ctx* c, const uint8_t* id, int32_t rank, int32_t world, const int32_t* x0, const int32_t* x1)
{
    if (!c) return GSR_ERR_ARG;
    if (!id) return fail(c, GSR_ERR_ARG, "gsr_comm_init: id is NULL");
    if (!rccl().ok) return fail(c, GSR_ERR_COMM, "%s", rccl().error.c_str());
    if (int r = comm_setup(c, "gsr_comm_init", rank, world, x0, x1, nullptr)) return r;
    ncclUniqueId u;
    memcpy(u.internal, id, GSR_COMM_ID_BYTES);
    const ncclResult_t nr = rccl().CommInitRank(&c->comm, world, u, rank);   // collective: returns when every rank has joined
    if (nr != ncclSuccess) {
        c->comm = nullptr;
        comm_release(c);
        return fail(c, GSR_ERR_COMM, "ncclCommInitRank failed: %s", rccl().GetErrorString(nr));
    }
    return GSR_OK;
}

int gsr_comm_share(gsr_ctx* c, gsr_ctx* leader)
{
    if (!c || !leader) return GSR_ERR_ARG;
    if (c == leader || (!leader->comm && !leader->comm_fn) || !leader->comm_owned)
        return fail(c, GSR_ERR_ARG, "gsr_comm_share: the other context must have joined a group itself (gsr_comm_init)");
    if (c->device != leader->device || c->W != leader->W || c->H != leader->H)
        return fail(c, GSR_ERR_ARG, "gsr_comm_share: both contexts must be on one device and of one size");
    if (int r = comm_setup(c, "gsr_comm_share", leader->comm_rank, leader->comm_world, leader->comm_edges.x0, leader->comm_edges.x1, leader->comm_stream))
        return r;
    c->comm = leader->comm; c->comm_fn = leader->comm_fn; c->comm_fn_user = leader->comm_fn_user;
    c->comm_leader = leader;
    leader->comm_followers.push_back(c);
    return GSR_OK;
}

int gsr_comm_init_custom(gsr_ctx* c, int32_t rank, int32_t world, const int32_t* x0, const int32_t* x1, gsr_allgather_fn fn, void* user)
{
    if (!c) return GSR_ERR_ARG;
    if (!fn) return fail(c, GSR_ERR_ARG, "gsr_comm_init_custom: fn is NULL");
    if (int r = comm_setup(c, "gsr_comm_init_custom", rank, world, x0, x1, nullptr)) return r;
    c->comm_fn = fn; c->comm_fn_user = user;
    return GSR_OK;
}

int gsr_comm_destroy(gsr_ctx* c)
{
    if (!c) return GSR_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    comm_release(c);
    return GSR_OK;
}

int gsr_allgather_frame_async(gsr_ctx* c)
{
    if (!c) return GSR_ERR_ARG;
    if (!c->comm && !c->comm_fn) return fail(c, GSR_ERR_ARG, "gsr_allgather_frame_async: gsr_comm_init has not been called");
    if (!c->have_frame) return fail(c, GSR_ERR_ARG, "gsr_allgather_frame_async: nothing rendered yet");
    HIP_TRY(c, hipSetDevice(c->device));
    // never ship a band the compositor did not draw: if the device has reported a list overflow, regrow and render
    // the frame again first (lost earlier frames stay counted and are reported by the next gsr_sync)
    if (overflow_pending(c)) { if (int r = sync_and_repair(c)) return r; }
    const BinGrid g = make_grid(c);
    const int x0 = g.bx_lo * BIN_PX, x1 = std::min(g.bx_hi * BIN_PX, c->W);
    // render stream: the previous all-gather must have read the slab before it is overwritten; then pack the band
    if (c->frame8_valid) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_slab_free, 0));
    // (the pack also records, behind the pixels, whether the frame it packs was composited at all: the frame's overflow word,
    //  which the next frame's projection resets -- stream order puts this read in front of it)
    launch_pack_band_rgba8(c->fb, c->slab, c->W, c->H, x0, x1, c->slab_w, c->stream, &c->fstate->overflow);
    HIP_TRY(c, hipEventRecord(c->ev_packed, c->stream));
    // exchange stream: collective + de-slab, overlapping the next frame's kernels on the render stream
    HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0));
    const size_t slab_bytes = ((size_t)c->slab_w * c->H + SLAB_FLAG_WORDS) * 4;
    if (c->comm_fn) {
        if (const int r = c->comm_fn(c->comm_fn_user, c->slab, c->gathered, (uint64_t)slab_bytes, (void*)c->comm_stream))
            return fail(c, GSR_ERR_COMM, "the custom all-gather returned %d", r);
    } else {
        RCCL_TRY(c, rccl().AllGather(c->slab, c->gathered, slab_bytes, ncclUint8, c->comm, c->comm_stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev_slab_free, c->comm_stream));
    launch_unpack_slabs_rgba8(c->gathered, c->frame8, c->W, c->H, c->slab_w, c->comm_world, c->comm_edges, c->comm_stream,
                              c->frame8 + (size_t)c->W * c->H);
    HIP_TRY(c, hipGetLastError());
    c->frame8_valid = true;
    return GSR_OK;
}

int gsr_read_frame_rgba8(gsr_ctx* c, uint8_t* out)
{
    if (!c || !out) return c ? fail(c, GSR_ERR_ARG, "out is NULL") : GSR_ERR_ARG;
    if (!c->frame8_valid) return fail(c, GSR_ERR_ARG, "gsr_read_frame_rgba8: no gathered frame yet (gsr_allgather_frame_async)");
    HIP_TRY(c, hipSetDevice(c->device));
    // A band of the gathered frame may have been packed right behind a frame whose bin lists did not fit: that frame was not
    // composited and the band is the preceding image.  WHICH gathered frame that concerns is decided on the device and seen by
    // the whole group: every slab carries its frame's overflow flag through the all-gather and the de-slab kernel collects the
    // flags of all ranks behind the assembled frame.  So every rank refuses exactly the same frame (GSR_ERR_OVERFLOW) and the
    // group renders and gathers it again together -- no rank repeats a collective alone -- while frames dropped earlier and
    // not reported yet (gsr_sync's business) do not make a good frame unreadable.  The rank that overflowed regrows its lists
    // here, so that the repeated frame fits.
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (overflow_pending(c)) { if (int r = sync_and_repair(c)) return r; }
    uint32_t stale = 0;
    HIP_TRY(c, hipMemcpyAsync(out, c->frame8, (size_t)c->W * c->H * 4, hipMemcpyDeviceToHost, c->comm_stream));
    HIP_TRY(c, hipMemcpyAsync(&stale, c->frame8 + (size_t)c->W * c->H, 4, hipMemcpyDeviceToHost, c->comm_stream));
    HIP_TRY(c, hipStreamSynchronize(c->comm_stream));
    if (stale) {
        c->frame8_valid = false;
        return fail(c, GSR_ERR_OVERFLOW, "the gathered frame holds a band that was not composited (the bin lists of rank mask 0x%x did not fit; "
                                         "they have been regrown there): every rank of the group gets this error for this frame and all of them "
                                         "render and gather it again", stale);
    }
    return GSR_OK;
}

void* gsr_frame8_device_ptr(gsr_ctx* c) { return c ? (void*)c->frame8 : nullptr; }
void* gsr_comm_stream_handle(gsr_ctx* c) { return c ? (void*)c->comm_stream : nullptr; }

}  // extern "C"

extern "C" {

// ---- wasm `sort` drop-in (wasm/wasm.cpp:8-13; call site Worker.ts:39) ----
// Like the wasm export it keeps nothing of the caller's between calls: the positions are copied to the device on
// every call (12 bytes per splat over PCIe; a caller that sorts one scene many times uses gsr_set_scene + gsr_sort,
// which is what the renderer does).  An earlier version cached the upload by buffer address and size; JS hosts
// transform positions in place (Scene.translate/rotate/scale) and a collected Float32Array can come back at the same
// address, so that cache returned stale orders.  On failure depthIndex is zero-filled and the error goes to stderr
// (the reference's signature has no error channel).
void gsplat_sort_host(const float* viewProj, uint32_t vertexCount, const float* fBuffer, uint32_t* depthBuffer,
                      uint32_t* depthIndex, uint32_t* starts, uint32_t* counts)
{
    (void)starts; (void)counts;
    static std::mutex mu;
    static gsr_ctx* ctx = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto failed = [&](const char* what) {
        fprintf(stderr, "gsplat_sort_host: %s\n", what);
        if (depthIndex && vertexCount) memset(depthIndex, 0, (size_t)vertexCount * sizeof(uint32_t));
    };
    if (!viewProj || !depthIndex || (vertexCount && !fBuffer)) return failed("NULL argument");
    if (!ctx) {
        gsr_options o{};
        if (gsr_create(&ctx, &o) != GSR_OK) { ctx = nullptr; return failed(gsr_last_error(nullptr)); }
    }
    gsr_ctx* c = ctx;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return failed("device unavailable");
    if (vertexCount != c->n || !c->px) {
        if (vertexCount > 0x7fffffffu / 8) return failed("too many splats");
        if (alloc_scene(c, vertexCount, false) != GSR_OK) return failed(gsr_last_error(c));
        c->n = vertexCount;
    }
    c->have_sort = false; c->have_frame = false;
    // What the call keeps between calls is memory of its own, never the caller's data: the device staging buffer for the
    // xyz-interleaved positions lives in the process-wide context and grows with the largest scene seen (round 3 allocated and
    // freed it on every call: two driver round trips of ~0.1 ms each beside a 45 us sort).  The chain of a call: pageable H2D of
    // 12 N bytes -> repack to the SoA the key kernel reads -> key + min/max -> quantise + 17-bit radix sort -> D2H of 4 N bytes
    // (+ 4 N for the keys on request), all on the context's stream, one host wait at the end.
    static float* stage_pos = nullptr;      // (guarded by mu, like ctx)
    static size_t stage_cap = 0;
    if (vertexCount) {
        if ((size_t)vertexCount * 3 > stage_cap) {
            if (dev_alloc(c, &stage_pos, (size_t)vertexCount * 3) != GSR_OK) { stage_cap = 0; return failed(gsr_last_error(c)); }
            stage_cap = (size_t)vertexCount * 3;
        }
        hipError_t e1 = hipMemcpyAsync(stage_pos, fBuffer, (size_t)vertexCount * 12, hipMemcpyHostToDevice, c->stream);
        launch_repack_positions(stage_pos, vertexCount, c->px, c->py, c->pz, c->stream);
        for (hipError_t e : {e1, hipGetLastError()})
            if (e != hipSuccess) return failed(hipGetErrorString(e));
    }
    float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (gsr_set_camera(c, ident, ident, viewProj, 1.f, 1.f) != GSR_OK) return failed(gsr_last_error(c));
    if (hipSetDevice(c->device) != hipSuccess || enqueue_frame(c, false) != GSR_OK) return failed(gsr_last_error(c));
    if (vertexCount) {
        hipError_t e1 = hipMemcpyAsync(depthIndex, c->depth_index, (size_t)vertexCount * 4, hipMemcpyDeviceToHost, c->stream);
        hipError_t e2 = depthBuffer ? hipMemcpyAsync(depthBuffer, c->keys, (size_t)vertexCount * 4, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
        for (hipError_t e : {e1, e2})
            if (e != hipSuccess) return failed(hipGetErrorString(e));
    }
    if (finish_frame(c) != GSR_OK) return failed(gsr_last_error(c));
}

// Identifies the device code this library was built from (a hash of the kernel sources, set by the Makefile):
// measurements taken on one build (profiles/blend_traffic.json) are not attributed to another.
const char* gsr_build_id(void)
{
#ifdef GSR_BUILD_ID
    return GSR_BUILD_ID;
#else
    return "unknown";
#endif
}

}  // extern "C"
