// Order-preserving coarse binning: the depth-sorted splat stream is split into
// one list per BIN_PX x BIN_PX screen bin (a splat enters every bin its pixel
// bounding box touches), and every list keeps the global front-to-back order.
// No second sort of duplicated (tile, depth) pairs is needed: the split is a
// stable counting pass (count -> scan -> scatter).
//
// Stability in the scatter: a wave owns 8 steps of 64 consecutive ranks.  For
// step s, the set of lanes whose box covers bin (X, Y) is
//     colmask[s][X] & rowmask[s][Y]
// where colmask/rowmask are 64-bit lane sets built in LDS with ds_or_b64, one
// word per bin column / bin row (boxes are rectangles, so coverage separates).
// A splat's slot in a bin is the wave's first slot there, plus the sizes of the
// sets of the earlier steps, plus the population count of its own step's set
// below its lane: wavefront ballot arithmetic, no atomics with ordering
// requirements, and no running counter to update between steps.
#include "gsr_internal.h"

#include <algorithm>
#include <mutex>


namespace gsr {

#ifdef GSR_KSTAMPS
// Diagnostic build only: per-workgroup phase times of k_bin_scatter (s_memrealtime ticks, 10 ns), read by
// scripts/kernel_stamps.py through gsr_debug_bin_stamps.
__device__ unsigned int g_bin_stamps[4096 * 8];
#define KSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) g_bin_stamps[blockIdx.x * 8 + (slot)] = (unsigned int)__builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int gsr_debug_bin_stamps(unsigned int* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bin_stamps), sizeof g_bin_stamps) == hipSuccess ? 0 : -1; }
#else
#define KSTAMP(slot)
#endif

GSR_BOUNDS_DECL(bin)   // sites: 0 splat index of a rank, 1 rectangle inside the bin grid, 2 LDS cell of the scatter, 3 table row, 4 count cell
constexpr int BIN_THREADS = 256;
constexpr int BIN_STEPS = 8;                                  // 64-rank steps per wave
constexpr uint32_t BIN_RANKS_PER_BLOCK = BIN_THREADS * BIN_STEPS;  // 2048

struct BinRect { int x0, x1, y0, y1; };  // inclusive bin coordinates relative to the band; x0 > x1: none

// (the packed form, pack_bin_rect, is in gsr_internal.h: k_project_key writes it, the last radix pass sorts it)
__device__ __forceinline__ BinRect unpack_rect(uint32_t p)
{
    BinRect r;
    r.x0 = (int)(p & 0xffu); r.x1 = (int)((p >> 8) & 0xffu); r.y0 = (int)((p >> 16) & 0xffu); r.y1 = (int)(p >> 24);
    if (r.x0 > r.x1) { r.y0 = 1; r.y1 = 0; }
    return r;
}

// the rectangle in cells of 2^shift x 2^shift bins (two-level binning, below); shift 0: in bins
__device__ __forceinline__ BinRect unpack_rect(uint32_t p, int shift)
{
    BinRect r = unpack_rect(p);
    if (r.x0 <= r.x1) { r.x0 >>= shift; r.x1 >>= shift; r.y0 >>= shift; r.y1 >>= shift; }
    return r;
}

__device__ __forceinline__ uint32_t lanes_below64(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------------------
// count: table[block][bin] = entries this workgroup's 2048 ranks contribute to bin.  The ranks' packed bin rectangles
// (4 bytes per splat, k_project_key) are gathered through depth_index and left in depth order for the scatter pass.
// No global atomics: hundreds of workgroups hitting the same few words cost more than the whole pass.
// ---------------------------------------------------------------------------
constexpr int CNT_THREADS = 1024;
constexpr int CNT_STEPS = (int)BIN_RANKS_PER_BLOCK / CNT_THREADS;  // 2
constexpr int CNT_MAX_BINS = 12288;  // LDS counters per workgroup (48 KiB); larger grids are cut into row slices (blockIdx.y)

__global__ __launch_bounds__(CNT_THREADS) void k_bin_count(const uint32_t* __restrict__ depth_index, const uint32_t* __restrict__ rect_idx,
                                                           const uint32_t* __restrict__ count, BinGrid g, int slice_rows,
                                                           uint32_t rounds, uint32_t* __restrict__ table, uint32_t* __restrict__ rects, int shift, int sorted,
                                                           uint32_t n_max)
{
    // the first round's ranks (or their rectangles) do not depend on how many ranks the frame holds, only on how many the
    // buffers do (n_max): their loads go out in front of the count's, one round trip instead of two
    const uint32_t begin0 = blockIdx.x * rounds * BIN_RANKS_PER_BLOCK;
    uint32_t first[CNT_STEPS];
#pragma unroll
    for (int st = 0; st < CNT_STEPS; st++) {
        const uint32_t r = begin0 + st * CNT_THREADS + threadIdx.x;
        first[st] = r < n_max ? (sorted ? rects[r] : depth_index[r]) : 0u;
    }
    const uint32_t n = *count;  // ranks the sort produced (all splats, or the band's survivors)
    // (band mode: the grid is the scene's, the ranks are the band's: a workgroup past them leaves no row -- the scan and the
    //  scatter stop at the last row that has ranks, launch_column_scan's `live`)
    if ((unsigned long long)blockIdx.x * rounds * BIN_RANKS_PER_BLOCK >= n) return;
    extern __shared__ uint32_t s_cnt[];  // this slice's bins
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    const int y_lo = blockIdx.y * slice_rows, y_hi = min(y_lo + slice_rows, g.nby);  // bin rows of this slice
    const int nb_s = (y_hi - y_lo) * nbxb;
    // a cell pass (shift > 0: `g` is the grid of cells, the rectangles are in bins) also sums the rectangles' areas in BINS:
    // one more column of the table, whose total is the number of list entries the frame needs (launch_bin, two levels)
    const int stride = nbins + (shift ? 1 : 0);
    uint32_t area = 0;
    for (int b = threadIdx.x; b < nb_s + (shift ? 1 : 0); b += CNT_THREADS) s_cnt[b] = 0;
    __syncthreads();
    // a workgroup takes `rounds` rounds of 2048 consecutive ranks into the same counters: one table row per workgroup, so
    // that a large frame's [workgroup][bin] table stays small (launch_bin)
    for (uint32_t rd = 0; rd < rounds; rd++) {
    const uint32_t begin = (blockIdx.x * rounds + rd) * BIN_RANKS_PER_BLOCK;
    if (begin >= n) break;
    uint32_t idx[CNT_STEPS], rc[CNT_STEPS];
    if (sorted) {   // (uniform) the sort carried the rectangles into depth order: no gather
#pragma unroll
        for (int st = 0; st < CNT_STEPS; st++) {
            const uint32_t r = begin + st * CNT_THREADS + threadIdx.x;
            rc[st] = (r < n) ? (rd == 0 ? first[st] : rects[r]) : RECT_NONE;
        }
    } else {
#pragma unroll
        for (int st = 0; st < CNT_STEPS; st++) {
            const uint32_t r = begin + st * CNT_THREADS + threadIdx.x;
            idx[st] = (r < n) ? (rd == 0 ? first[st] : depth_index[r]) : 0xffffffffu;
        }
#pragma unroll
        for (int st = 0; st < CNT_STEPS; st++) rc[st] = (idx[st] != 0xffffffffu) ? rect_idx[idx[st]] : RECT_NONE;
    }
#ifdef GSR_BOUNDS
#pragma unroll
    for (int st = 0; st < CNT_STEPS; st++) {
        const BinRect bq = unpack_rect(rc[st], shift);
        if (bq.x0 <= bq.x1) { GSR_BOUND(bin, 1, bq.x1, nbxb); GSR_BOUND(bin, 1, bq.y1, g.nby); GSR_BOUND(bin, 1, bq.y0, bq.y1 + 1); }
    }
#endif
    if (blockIdx.y == 0 && !sorted) {   // slice 0 also leaves the rectangles in depth order
#pragma unroll
        for (int st = 0; st < CNT_STEPS; st++) {
            const uint32_t r = begin + st * CNT_THREADS + threadIdx.x;
            if (r < n) rects[r] = rc[st];
        }
    }
#pragma unroll
    for (int st = 0; st < CNT_STEPS; st++) {
        const BinRect br = unpack_rect(rc[st], shift);
        if (shift) {
            const BinRect fine = unpack_rect(rc[st]);
            if (fine.x0 <= fine.x1) area += (uint32_t)((fine.x1 - fine.x0 + 1) * (fine.y1 - fine.y0 + 1));
        }
        if (br.x0 <= br.x1)
            for (int y = max(br.y0, y_lo); y <= min(br.y1, y_hi - 1); y++)
                for (int x = br.x0; x <= br.x1; x++) {
                    GSR_BOUND(bin, 4, (y - y_lo) * nbxb + x, nb_s);
                    atomicAdd(&s_cnt[(y - y_lo) * nbxb + x], 1u);
                }
    }
    }   // rounds
    if (shift) {
#pragma unroll
        for (int off = WAVE / 2; off; off >>= 1) area += __shfl_xor(area, off);
        if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(&s_cnt[nb_s], area);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb_s + (shift ? 1 : 0); b += CNT_THREADS) table[(size_t)blockIdx.x * stride + y_lo * nbxb + b] = s_cnt[b];
}

// (the scan of table[block][bin] down the blocks, and bin_total[], is launch_column_scan of k_sort.hip)

// ---------------------------------------------------------------------------
// finalize (one workgroup): bin_start = exclusive scan of bin_total; the compositor's work items
// (bin, segment of seg_len list entries), one per segment, at least one per bin; frame counters.
// ---------------------------------------------------------------------------
constexpr int FIN_THREADS = 1024;
constexpr int FIN_WAVES = FIN_THREADS / WAVE;

// exclusive scan of N independent u32 streams over the workgroup's FIN_THREADS threads; totals in *tot
template <int N> struct UN { uint32_t v[N]; };

template <int N>
__device__ __forceinline__ UN<N> block_exclusive_scan(UN<N> x, uint32_t (*s_w)[FIN_WAVES], UN<N>* tot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    UN<N> inc = x;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t t = __shfl_up(inc.v[k], off);
            if (lane >= off) inc.v[k] += t;
        }
    }
    if (lane == WAVE - 1) {
#pragma unroll
        for (int k = 0; k < N; k++) s_w[k][wave] = inc.v[k];
    }
    __syncthreads();
    UN<N> base, total;
#pragma unroll
    for (int k = 0; k < N; k++) { base.v[k] = 0; total.v[k] = 0; }
#pragma unroll
    for (int w = 0; w < FIN_WAVES; w++) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const uint32_t t = s_w[k][w];
            if (w < wave) base.v[k] += t;
            total.v[k] += t;
        }
    }
    __syncthreads();
    *tot = total;
    UN<N> ex;
#pragma unroll
    for (int k = 0; k < N; k++) ex.v[k] = base.v[k] + inc.v[k] - x.v[k];
    return ex;
}

// Long lists raise the segment length: the compositor needs a few thousand work items to fill the chip, not one
// per 512 entries; every extra segment costs a 16 KiB partial written and read again by k_blend (the fold).
constexpr uint32_t SEG_LEN_MAX = 8192;

// Size class of a bin's last (partial) segment of r entries -- with long work items: of the whole bin -- for the order of
// the work items: 4 classes per power of two (r < 65536 -> 0 .. 63), larger = heavier.  The compositor draws the items of
// the heaviest class first.  (With several frames in flight the tail of one frame's compositor is filled by the other
// frames' kernels, and a tail made only of the lightest items was measured 2.5 % slower: then `by_size` is off and the
// bins' last segments stay in raster order.)
constexpr int FIN_CLASSES = 64;
__device__ __forceinline__ int partial_class(uint32_t r)
{
    if (r < 4u) return (int)r;
    const int e = 31 - __clz((int)r);                       // 2 .. 31
    return min(FIN_CLASSES - 1, (e - 1) * 4 + (int)((r >> (e - 2)) & 3u));
}

// Arguments of the finalize step (by value in the kernel arguments of whichever kernel runs it).
struct FinalizeArgs {
    const uint32_t* bin_total; int nbins;
    uint32_t seg_len_min, seg_target_items; int by_size;
    uint32_t* seg_len_out; uint32_t max_items, capacity;
    int32_t* slots; uint32_t have_counts;
    uint32_t *bin_start, *seg_start, *items, *overflow;
    uint64_t *visible, *tile_entries, *accum, *report;
    uint32_t* queue; uint32_t queue_start;
    uint64_t* mailbox;
    unsigned long long* bin_mask;
    int long_policy; uint32_t seg_len_long, long_tau, npix;   // see BinBuffers
    uint32_t long_tiles_x2, long_tau_bin, long_mass_min;
};
constexpr int FIN_LAYERS = 64;   // segments per bin at most (one bit each in the bin's arrival mask, k_blend)
constexpr int FIN_SCRATCH_WORDS = 64;   // LDS words the finalize step asks of its caller (none are used any more; kept so that every launch passes a non-zero size)

// One workgroup of FIN_THREADS threads.  It runs as an EXTRA workgroup of k_bin_scatter (the scatter workgroups
// compute the bin starts they need themselves), so its ~9 us no longer sit between the column scan and the scatter;
// k_bin_finalize is the stand-alone form for frames without splats.
__device__ __forceinline__ void bin_finalize_body(const FinalizeArgs& fa, uint32_t* __restrict__ scratch /* LDS, FIN_SCRATCH_WORDS */)
{
    const uint32_t* __restrict__ bin_total = fa.bin_total;
    const int nbins = fa.nbins;
    const uint32_t seg_len_min = fa.seg_len_min, seg_target_items = fa.seg_target_items, max_items = fa.max_items, capacity = fa.capacity;
    uint32_t* __restrict__ bin_start = fa.bin_start;
    uint32_t* __restrict__ seg_start = fa.seg_start;
    uint32_t* __restrict__ items = fa.items;
    int32_t* __restrict__ slots = fa.slots;
    const uint32_t have_counts = fa.have_counts;
    uint32_t* seg_len_out = fa.seg_len_out; uint32_t* overflow = fa.overflow; uint64_t* visible = fa.visible;
    uint64_t* tile_entries = fa.tile_entries; uint64_t* accum = fa.accum; uint64_t* report = fa.report;
    uint32_t* queue = fa.queue; const uint32_t queue_start = fa.queue_start; uint64_t* mailbox = fa.mailbox;
    __shared__ uint32_t s_w[5][FIN_WAVES];
    const int per = (nbins + FIN_THREADS - 1) / FIN_THREADS;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nbins);
    // The frame's totals, in ONE exchange through LDS (this workgroup runs beside the scatter workgroups and must not be the last
    // to finish; round 3 took three block-wide scans here -- six barriers -- where only sums were needed): list entries and the
    // longest list over the bins (all threads), and the projection's slot sums (wave 0, one slot per lane): optical depth over
    // the splats' boxes [4], optical mass [5], visible splats [2], tile overlaps [3].
    __shared__ unsigned long long s_tot[FIN_WAVES];
    __shared__ uint32_t s_mxb[FIN_WAVES];
    __shared__ unsigned long long s_slot[4];
    unsigned long long ent = 0;
    uint32_t mxbin = 0;   // the longest bin list of the frame
    for (int b = b0; b < b1; b++) { ent += bin_total[b]; mxbin = max(mxbin, bin_total[b]); }
    unsigned long long sv[4] = {0, 0, 0, 0};
    if (have_counts && threadIdx.x < FRAME_SLOTS) {
        const int32_t* sl = slots + (size_t)threadIdx.x * FRAME_SLOT_WORDS;
        sv[0] = (uint32_t)sl[4]; sv[1] = (uint32_t)sl[5]; sv[2] = (uint32_t)sl[2]; sv[3] = (uint32_t)sl[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ent += __shfl_xor(ent, off);
        mxbin = max(mxbin, (uint32_t)__shfl_xor((int)mxbin, off));
        if (threadIdx.x < WAVE) {   // (wave 0 only: uniform per wave)
#pragma unroll
            for (int k = 0; k < 4; k++) sv[k] += __shfl_xor(sv[k], off);
        }
    }
    if ((threadIdx.x & 63) == 0) { s_tot[threadIdx.x >> 6] = ent; s_mxb[threadIdx.x >> 6] = mxbin; }
    if (threadIdx.x == 0) { s_slot[0] = sv[0]; s_slot[1] = sv[1]; s_slot[2] = sv[2]; s_slot[3] = sv[3]; }
    __syncthreads();
    ent = 0;
#pragma unroll
    for (int w = 0; w < FIN_WAVES; w++) { ent += s_tot[w]; mxbin = max(mxbin, s_mxb[w]); }
    struct { uint32_t v[3]; } ent_tot = {{(uint32_t)min(ent, 0xffffffffull), 0u, 0u}};
    const uint64_t optical = s_slot[0];
    const uint64_t mass = s_slot[1] * 256u / 255u;   // sum of opacity x footprint, in pixels
    struct { uint32_t v[2]; } ctot = {{(uint32_t)s_slot[2], (uint32_t)s_slot[3]}};
    // dense enough to saturate: optical depth, and (one frame at a time, where the long items are the frame's tail) splats
    // that cover several tiles each -- small splats take many more entries to saturate a pixel (the C2 generator at
    // 1.6 M splats: tau 394, 3.6 tiles per splat, long items 8 % slower; C3: 5.3 tiles per splat, 23 % faster)
    const bool dense = optical * (16u * 256u) >= (uint64_t)fa.long_tau * 255u * (uint64_t)fa.npix &&
                       (uint64_t)ctot.v[1] * 2u >= (uint64_t)fa.long_tiles_x2 * ctot.v[0];
    // Which bins become ONE work item (gsr_api.cpp, "Work-item length").  A whole-bin item stops where the bin saturates, which
    // pays where the bin holds much more than it takes to saturate it -- and costs a serial pole as long as that takes; a bin cut
    // into segments is composited by several workgroups at once, every segment in full.  A frame can mix both kinds of bins (a
    // dense object in front of a sparse background), so the choice is made PER BIN, deterministically from figures of this
    // frame alone (no feedback from earlier frames):
    //  * the frame's prior, as before: `dense` (optical depth over the splats' boxes >= long_tau, and splats of several tiles
    //    each) -- calibrated on the frame-wide choice, which it reproduces on single-blob scenes;
    //  * a bin's own optical depth tau_b = c x (M / E) / 1024, c its entries, E the frame's entries, M the frame's optical mass
    //    (sum of opacity x footprint pixels, slot word [5]: what the splats really carry, independent of how coarse their tile
    //    boxes are) -- C3: up to 131 in the centre, 61 at the ninth decile; C2: 40 / 8; pixels stop changing around 20-25.
    // In a dense frame every bin is one item except those far from saturating (tau_b < LONG_TAU_B_LO: its sparse rim or
    // background, which is cut); in a frame that is not dense as a whole, bins that are several times past saturation
    // (tau_b >= LONG_TAU_B_HI) still become one item -- the dense object in a sparse scene -- provided the frame's entries are
    // heavy enough (M / E >= mass_min pixels: small splats saturate a pixel only after thousands of entries, a pole no skip
    // pays for; one frame at a time only).  long_policy 1 / 0 pin all / no bins.
    constexpr uint32_t LONG_TAU_B_LO = 8, LONG_TAU_B_HI = 60;
    uint32_t long_from = 0xffffffffu;                       // bins of at least this many entries are one work item
    if (seg_len_min < 0x40000000u) {
        if (fa.long_policy > 0) long_from = 0u;
        else if (fa.long_policy < 0 && mass > 0u) {
            const uint64_t E = ent_tot.v[0];
            const uint32_t tb = fa.long_tau_bin ? fa.long_tau_bin : dense ? LONG_TAU_B_LO : LONG_TAU_B_HI;
            const bool heavy = mass >= (uint64_t)fa.long_mass_min * E;
            if (dense || heavy || fa.long_tau_bin)
                long_from = (uint32_t)min((uint64_t)0xfffffff0u, ((uint64_t)tb * 1024u * E + mass - 1u) / mass);
        }
    }
    const uint32_t seg_min = seg_len_min;
    // segment length of the bins that ARE cut (a multiple of 256; the whole-bin sentinel of early termination passes through)
    // work items are ordered heaviest first whenever whole-bin items can occur: the few that run long must not start late (three
    // frames in flight, C3: 4970 -> 5310 frames/s); plain short segments in throughput contexts stay in raster order (C2: 11 550 vs 11 250)
    const bool by_size = fa.by_size != 0 || mxbin >= long_from;
    uint32_t seg_len = seg_min;
    if (seg_min < 0x40000000u)
        seg_len = min(max(ent_tot.v[0] / seg_target_items / 256u * 256u, seg_min), max(SEG_LEN_MAX, seg_min));
    // Items are emitted heaviest first -- by size class (whole bins, full segments, the bins' last segments alike), a counting
    // sort over FIN_CLASSES classes in LDS -- so the compositor's queue hands out the long items while the chip is still full
    // and only short ones are left for the tail.  The order inside a class is whatever the atomics give: the item list is a
    // work list, its order changes no pixel.  (by_size off: every full segment in raster order, then the last segments.)
    // Streams of the scan: entries, segments, full segments.
    __shared__ uint32_t s_cls[FIN_CLASSES];
    if (threadIdx.x < FIN_CLASSES) s_cls[threadIdx.x] = 0;
    (void)scratch;
    __syncthreads();
    // the cut of a bin of c entries: nf full segments, and whether a last segment of r entries follows (an empty bin is one
    // item: its pixels are cleared).  At most FIN_LAYERS segments per bin: the last one takes whatever is left.
    auto cut = [&](uint32_t c, uint32_t& nf, uint32_t& r, bool& part) {
        nf = c >= long_from ? 0u : min(c / seg_len, (uint32_t)FIN_LAYERS - 1u);   // (a whole-bin item is the bin's "last segment")
        r = c - nf * seg_len;
        part = r || !nf;
    };
    const int cls_full = partial_class(seg_len);
    UN<3> mine = {{0, 0, 0}};   // entries, segments, full segments
    for (int b = b0; b < b1; b++) {
        const uint32_t c = bin_total[b];
        uint32_t nf, r; bool part;
        cut(c, nf, r, part);
        mine.v[0] += c;
        mine.v[1] += nf + (part ? 1u : 0u);
        mine.v[2] += nf;
        if (by_size) {
            if (nf) atomicAdd(&s_cls[cls_full], nf);
            if (part) atomicAdd(&s_cls[partial_class(r)], 1u);
        }
    }
    UN<3> tot;
    const UN<3> ex3 = block_exclusive_scan<3>(mine, s_w, &tot);   // (its barriers also order the class counts)
    uint32_t ex = ex3.v[0], sx = ex3.v[1], fx = ex3.v[2];
    if (by_size && threadIdx.x < WAVE) {   // class counts -> first item index of each class, heaviest class first
        static_assert(FIN_CLASSES == WAVE, "one class per lane");
        const uint32_t cnt = s_cls[FIN_CLASSES - 1 - threadIdx.x];
        uint32_t inc = cnt;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off);
            if ((int)threadIdx.x >= off) inc += u;
        }
        s_cls[FIN_CLASSES - 1 - threadIdx.x] = inc - cnt;
    }
    __syncthreads();
    uint32_t p3 = tot.v[2] + (sx - fx);   // raster order of the last segments (by_size off)
    // A frame whose lists do not fit (entries > capacity or items > max_items) must not be composited:
    // it publishes no work items at all (every index the compositor derives stays in range), raises the
    // overflow word, and the host regrows the buffers and renders the frame again (gsr_sync).
    const uint32_t n_items = tot.v[1];
    const bool fits = tot.v[0] <= capacity && n_items <= max_items;
    for (int b = b0; b < b1; b++) {
        const uint32_t c = bin_total[b];
        uint32_t nf, r; bool part;
        cut(c, nf, r, part);
        bin_start[b] = fits ? ex : 0u;
        seg_start[b] = fits ? sx : 0u;
        if (fa.bin_mask) fa.bin_mask[b] = 0ull;
        if (fits) {
            // a work item says everything the compositor needs to start on it: (bin | segment << 16, first entry, end, the bin's
            // first partial slot | its segments << 25) -- one 16-byte load behind the queue's atomic instead of an item word and
            // then the bin's four table words
            uint4* items4 = reinterpret_cast<uint4*>(items);
            const uint32_t nseg = nf + (part ? 1u : 0u), w3 = sx | (nseg << 25);
            const uint32_t f0 = (by_size && nf) ? atomicAdd(&s_cls[cls_full], nf) : fx;
            for (uint32_t k = 0; k < nf; k++)
                items4[f0 + k] = make_uint4((uint32_t)b | (k << 16), ex + k * seg_len, k + 1u == nseg ? ex + c : ex + (k + 1u) * seg_len, w3);
            if (part) {
                const uint32_t pos = by_size ? atomicAdd(&s_cls[partial_class(r)], 1u) : p3++;
                items4[pos] = make_uint4((uint32_t)b | (nf << 16), ex + nf * seg_len, ex + c, w3);
            }
        }
        ex += c;
        sx += nf + (part ? 1u : 0u);
        fx += nf;
    }
    // This step is the last reader of the frame slots (k_quantise_hist took the depth range earlier in the frame): it leaves
    // them as the next frame's projection expects them -- min <- INT_MAX, max <- INT_MIN (the values wasm/wasm.cpp:14-15 starts
    // from), the counters <- 0 -- with its last stores, so that no kernel in front of a frame has to.
    if (have_counts && threadIdx.x < FRAME_SLOTS) {
        int32_t* sl = slots + (size_t)threadIdx.x * FRAME_SLOT_WORDS;
        sl[0] = 0x7fffffff; sl[1] = (int32_t)0x80000000; sl[2] = 0; sl[3] = 0; sl[4] = 0; sl[5] = 0;
    }
    if (threadIdx.x == 0) {
        *queue = queue_start;  // the compositor's workgroups take items 0..grid-1 by index, later ones from here
        seg_len_out[0] = seg_len;
        seg_len_out[1] = fits ? n_items : 0u;   // the compositor's queue length
        seg_len_out[2] = 0u;                    // (reserved)
        bin_start[nbins] = fits ? tot.v[0] : 0u;
        seg_start[nbins] = fits ? tot.v[1] : 0u;
        accum[4] = tot.v[0];  // entries this frame needs (the host sizes the regrowth from it)
        if (!fits) {
            atomicOr(overflow, tot.v[0] > capacity ? 1u : 2u);
            // sticky (never reset by k_begin_frame): frames that were not composited, and the most entries any
            // of them needed; the count also goes to a host-mapped word so that the host notices an overflow in
            // the middle of an asynchronous run without a copy or a sync (gsr_render_async polls it)
            accum[5] += 1;
            accum[6] = max(accum[6], (uint64_t)tot.v[0]);
            accum[7] = max(accum[7], (uint64_t)n_items);
            __hip_atomic_store(mailbox, accum[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        *visible = ctot.v[0];
        *tile_entries = ctot.v[1];
        accum[0] += ctot.v[0]; accum[1] += tot.v[0]; accum[2] += ctot.v[1]; accum[3] += 1;
#pragma unroll
        for (int k = 0; k < 5; k++) report[k] = accum[k];   // what the host reads at the next synchronisation: one copy
        report[5] = fits ? tot.v[0] : 0u;
    }
}

__global__ __launch_bounds__(FIN_THREADS) void k_bin_finalize(FinalizeArgs fa)
{
    extern __shared__ uint32_t s_fin[];   // FIN_SCRATCH_WORDS
    bin_finalize_body(fa, s_fin);
}

// ---------------------------------------------------------------------------
// scatter: list[...] = splat index, bins in raster order, depth order inside a bin.
// Workgroup = 16 waves over 2048 consecutive ranks.  The ranks form 4 groups of 512 (8 steps of 64);
// group g is shared by waves 4g..4g+3, each owning 2 of its steps, so input order is
// (workgroup, group, step, lane).  Sixteen waves instead of four do the same work with 4x the
// latency hiding: the pass is a chain of dependent LDS / global round trips, not arithmetic.
// ---------------------------------------------------------------------------
constexpr int SCAT_THREADS = 1024;
constexpr int SCAT_WAVES = SCAT_THREADS / WAVE;               // 16 waves, each owns 2 steps of 64 ranks
constexpr int SCAT_STEPS_PER_WAVE = 2;
constexpr int SCAT_STEPS = SCAT_WAVES * SCAT_STEPS_PER_WAVE;  // 32 steps = 2048 ranks
static_assert(SCAT_STEPS * WAVE == (int)BIN_RANKS_PER_BLOCK, "scatter and count must cut the ranks alike");
// The ranks of a workgroup form GROUPS groups of consecutive steps.  A splat's slot needs the number of entries the
// earlier steps of its group put into the bin -- one popcount of two LDS lane sets per earlier step -- plus the
// group's offset, which costs 16 bits of LDS per (group, bin).  More groups = shorter step loops (the kernel is bound
// by LDS instructions: ~11 reads per entry with 4 groups of 8 steps, ~6 with 8 groups of 4) but a larger table, so the
// group count follows what fits: 8 groups at 1080p (64 KiB), 4 where the bin grid is large (4K: 146 KiB).
//
// Framebuffers above 4K: the LDS tables outgrow a CU's 160 KiB, so the bin grid is cut into the fewest sub-grids that
// fit and blockIdx.y picks the sub-grid.  Every sub-grid workgroup reads the same 2048 rectangles and places the
// entries of its own bins.  Up to 4K (8160 bins) there is one sub-grid: cutting earlier was measured slower (C4:
// 0.50 -> 0.66 ms for the binning), the per-workgroup fixed work dominates.
constexpr size_t SCAT_LDS_BUDGET = 150 * 1024;
constexpr size_t SCAT_LDS_TWO_PER_CU = 72 * 1024;   // with at most this much, two workgroups share a CU

struct BinSlices { int32_t sx, sy, w, h; };  // sx x sy sub-grids of w x h bins (the last ones may be smaller)

inline size_t scatter_lds_bytes(int w, int h, int groups, int steps = SCAT_STEPS)
{
    return (size_t)(((1 + groups / 2) * w * h + 1) & ~1) * 4 + (size_t)steps * (w + h) * 8;
}

inline BinSlices make_slices(int nbxb, int nby)
{
    for (int t = 1;; t++)
        for (int sx = 1; sx <= t; sx++) {
            if (t % sx) continue;
            BinSlices sl;
            sl.sx = sx; sl.sy = t / sx;
            sl.w = (nbxb + sl.sx - 1) / sl.sx;
            sl.h = (nby + sl.sy - 1) / sl.sy;
            if (scatter_lds_bytes(sl.w, sl.h, 4) <= SCAT_LDS_BUDGET) return sl;
        }
}

// bin_start_pre = exclusive scan of the bin totals, for the scatter workgroups of large bin grids (the finalize step, which
// computes the same starts among much else, runs beside them as the scatter kernel's extra workgroup)
__global__ __launch_bounds__(FIN_THREADS) void k_bin_starts(const uint32_t* __restrict__ bin_total, int nbins, uint32_t* __restrict__ bin_start_pre)
{
    __shared__ uint32_t s_w[1][FIN_WAVES];
    const int per = (nbins + FIN_THREADS - 1) / FIN_THREADS;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nbins);
    UN<1> mine = {{0}}, tot;
    for (int b = b0; b < b1; b++) mine.v[0] += bin_total[b];
    uint32_t run = block_exclusive_scan<1>(mine, s_w, &tot).v[0];
    for (int b = b0; b < b1; b++) {
        bin_start_pre[b] = run;
        run += bin_total[b];
    }
    if (threadIdx.x == 0) bin_start_pre[nbins] = tot.v[0];
}


// ---------------------------------------------------------------------------
// Two-level binning (large bin grids: launch_bin).  Level one is the pass below over a grid of CELLS of 4 x 4 bins: a
// splat enters every cell its rectangle touches -- 1.7 cells instead of 5.2 bins per splat on C4, with tables of 510
// instead of 8160 columns -- and the cell lists keep the depth order; an entry carries the splat's rectangle in bins.
// Level two cuts every cell list into chunks of CELL_CHUNK entries and places each entry in the 16 bins of its cell:
// count (k_cell_count: per chunk and bin, and per wave of the chunk), scan (k_cell_scan: down the chunks of a cell),
// scatter (k_cell_scatter2: per 64 entries eight ballots -- columns and rows -- whose scalar ANDs are the sixteen bins'
// lane sets and, as they are, the exec masks of the stores) -- no LDS lane sets, no atomics, no loops of different
// lengths in one wave.  The lists are the ones the one-level pass builds, entry for entry.
// ---------------------------------------------------------------------------
constexpr int CELL_SHIFT = 2;
constexpr int CELL_SIDE = 1 << CELL_SHIFT;                 // bins along a cell's side
constexpr int CELL_FINE = CELL_SIDE * CELL_SIDE;           // bins of a cell
constexpr uint32_t CELL_CHUNK = 2048;                      // cell-list entries per level-two pass of a workgroup
struct CellArgs {
    uint32_t* cell_start;    // ncells + 1: first cell-list entry of every cell
    uint32_t* chunk_start;   // ncells + 2: first chunk of every cell; [ncells] = the frame's chunks; [ncells + 1] = list entries the frame needs
    uint4* chunk_info;       // per chunk: (cell, first entry, end entry, -) -- what a level-two workgroup needs to know about its chunk
    int shift;               // CELL_SHIFT in the cell pass, 0 otherwise
};

// The extra workgroup of the cell pass: starts of the cells and of their chunks.  cell_total[ncells] is the total of the
// count pass's area column = the list entries the frame needs: a frame that does not fit gets no chunks (level two then
// reports the need and the frame is rendered again after the regrowth, as in the one-level path).
__device__ __forceinline__ void cell_finalize_body(const uint32_t* __restrict__ cell_total, int ncells, uint32_t capacity, const CellArgs& ca)
{
    __shared__ uint32_t s_w[2][FIN_WAVES];
    const int per = (ncells + FIN_THREADS - 1) / FIN_THREADS;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, ncells);
    UN<2> mine = {{0, 0}}, tot;
    for (int b = b0; b < b1; b++) {
        const uint32_t c = cell_total[b];
        mine.v[0] += c;
        mine.v[1] += (c + CELL_CHUNK - 1u) / CELL_CHUNK;
    }
    const UN<2> ex = block_exclusive_scan<2>(mine, s_w, &tot);
    const uint32_t need = cell_total[ncells];
    uint32_t e = ex.v[0], k = ex.v[1];
    for (int b = b0; b < b1; b++) {
        const uint32_t c = cell_total[b];
        ca.cell_start[b] = e;
        ca.chunk_start[b] = k;
        if (need <= capacity)
            for (uint32_t o = 0; o < c; o += CELL_CHUNK) ca.chunk_info[k++] = make_uint4((uint32_t)b, e + o, e + min(o + CELL_CHUNK, c), 0u);
        else k += (c + CELL_CHUNK - 1u) / CELL_CHUNK;
        e += c;
    }
    if (threadIdx.x == 0) {
        ca.cell_start[ncells] = tot.v[0];
        ca.chunk_start[ncells] = need <= capacity ? tot.v[1] : 0u;
        ca.chunk_start[ncells + 1] = need;
    }
}

// The scatter body.  Two kernels wrap it (below):
//  * k_bin_scatter<GROUPS, FUSED>, 64 registers = 8 waves per SIMD, two of these 16-wave workgroups per CU.  FUSED (small
//    bin grids): the finalize step runs as an extra workgroup and every scatter workgroup scans the bin totals itself --
//    C3: binning 61.3 -> 56.5 us (the finalize step, built for one workgroup, wants 104 registers and spills a little in
//    its one workgroup instead of halving everybody's occupancy); !FUSED: the stand-alone k_bin_finalize has run and
//    published bin_start[].
//  * k_bin_scatter_big<GROUPS> (BIG), the large-grid form (4K: 8160 bins, 146 KiB of LDS, so one workgroup per CU whatever
//    its registers): 128 registers.  That budget is what lets it (a) run the finalize step as its FIRST workgroup (as the
//    last of 600+ it would start when the others end) with the bin starts coming from the one-workgroup k_bin_starts in
//    front -- k_bin_finalize's 28 us leave the chain -- and (b) take several ROUNDS of 2048 ranks per workgroup, base[]
//    carrying over, so that the [workgroup][bin] table all three binning kernels exchange is 20 MB instead of 80 MB at
//    5 M splats (more than the lists it helps to build).  Inside the 64-register kernel the same two things spilled 15 and
//    44 registers and cost more than they saved (C4 binning 477 -> 533 us, profiles/r03_experiments.txt).
template <int GROUPS, bool FUSED, bool BIG, int SPW /* 64-rank steps per wave and round */, bool CELLS = false>
__device__ __forceinline__ void bin_scatter_body(const uint32_t* __restrict__ depth_index,
                                                              const uint32_t* __restrict__ rects,
                                                              const uint32_t* __restrict__ count, BinGrid g, BinSlices sl,
                                                              const uint32_t* __restrict__ table,
                                                              const uint32_t* __restrict__ bin_total /* FUSED */,
                                                              const uint32_t* __restrict__ bin_start /* !FUSED */,
                                                              uint32_t* __restrict__ list, uint32_t capacity,
                                                              uint32_t* __restrict__ overflow, uint32_t rounds, const FinalizeArgs& fa,
                                                              const CellArgs& ca)
{
    static_assert(FIN_THREADS == SCAT_THREADS, "the finalize step runs as a workgroup of this kernel");
    static_assert(!(FUSED && BIG), "the large-grid form reads the bin starts from k_bin_starts");
    static_assert(!CELLS || FUSED, "the cell pass of the two-level binning is the fused form");
    constexpr bool EXTRA = FUSED || BIG;   // the finalize step is a workgroup of this launch
    if (EXTRA && blockIdx.x == (BIG ? 0u : gridDim.x - 1u)) {   // bin starts, work items and frame counters for the compositor
        extern __shared__ uint32_t s_fin[];   // this workgroup's share of the kernel's dynamic LDS (>= FIN_SCRATCH_WORDS, launch_bin)
        if (CELLS) cell_finalize_body(bin_total, (g.bx_hi - g.bx_lo) * g.nby, capacity, ca);   // (the cell pass: cell and chunk starts for level two)
        else if (blockIdx.y == 0) bin_finalize_body(fa, s_fin);
        return;
    }
    const int shift = CELLS ? ca.shift : 0;
    const int table_stride = (g.bx_hi - g.bx_lo) * g.nby + (CELLS ? 1 : 0);   // (the cell pass's table has the area column, k_bin_count)
    const uint32_t blk = xcd_group_remap(blockIdx.x - (BIG ? 1u : 0u), gridDim.x - (EXTRA ? 1u : 0u));   // neighbouring rank blocks on one XCD (gsr_internal.h)
    if ((unsigned long long)blk * (BIG ? rounds : 1u) * BIN_RANKS_PER_BLOCK >= *count) return;   // (band mode: no ranks here, no table row either)
    constexpr int WAVES_PER_GROUP = SCAT_WAVES / GROUPS;            // 4 or 2
    constexpr int STEPS = SCAT_WAVES * SPW;                         // steps of a round: 32 (2048 ranks) or 16 (1024)
    constexpr int GROUP_STEPS = STEPS / GROUPS;                     // 8 or 4
    constexpr int PAIR_WORDS = GROUPS / 2;                          // two 16-bit per-group counts / offsets per word
    extern __shared__ uint32_t s_mem[];
    const uint32_t n = *count;
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int group = wave / WAVES_PER_GROUP, sub = wave % WAVES_PER_GROUP;
    // this workgroup's sub-grid: bin columns [sx0, sx1), rows [sy0, sy1) of the band
    const int sx0 = (int)(blockIdx.y % sl.sx) * sl.w, sx1 = min(sx0 + sl.w, nbxb);
    const int sy0 = (int)(blockIdx.y / sl.sx) * sl.h, sy1 = min(sy0 + sl.h, g.nby);
    const int sw = sx1 - sx0, sh = sy1 - sy0, nb_s = sw * sh;
    // LDS: base[nb_s] (u32: the workgroup's first slot in each bin), pair[PAIR_WORDS][nb_s] (16 bits per group: a group
    // holds at most 512 ranks), then the lane sets: per step [sw] column words + [sh] row words.
    // Sized for a full sl.w x sl.h sub-grid.
    const int cap_s = sl.w * sl.h;
    uint32_t* base = s_mem;
    uint32_t* pair = s_mem + cap_s;
    const int nmask = sl.w + sl.h;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(s_mem + (((1 + PAIR_WORDS) * cap_s + 1) & ~1));
    unsigned long long* gmask = masks + (size_t)group * GROUP_STEPS * nmask;   // step s of my group: gmask + s*nmask
    uint32_t* mypair = pair + (size_t)(group >> 1) * cap_s;
    const int myshift = (group & 1) * 16;
    KSTAMP(0);
    // the workgroup's first slot in every bin = the bin's start (an exclusive scan of the bin totals, done here by
    // every workgroup: the finalize step that publishes bin_start[] runs beside this kernel, not before it) + the
    // entries earlier workgroups put into the bin (the scanned table).  base[] is not touched by phase 1.
    if (!FUSED) {
        for (int b = threadIdx.x; b < nb_s; b += SCAT_THREADS) {
            const int ly = b / sw, lx = b - ly * sw;
            const int gb = (sy0 + ly) * nbxb + sx0 + lx;  // the bin's index in the band
            GSR_BOUND(bin, 3, blk, gridDim.x);
            base[b] = bin_start[gb] + table[(size_t)blk * table_stride + gb];
        }
    } else {
        __shared__ uint32_t s_ws[SCAT_WAVES];
        const int per = (nbins + SCAT_THREADS - 1) / SCAT_THREADS;
        const int gb0 = threadIdx.x * per, gb1 = min(gb0 + per, nbins);
        uint32_t mine = 0;
        for (int gb = gb0; gb < gb1; gb++) mine += bin_total[gb];
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == WAVE - 1) s_ws[wave] = inc;
        __syncthreads();
        uint32_t run = inc - mine;
        for (int w = 0; w < wave; w++) run += s_ws[w];
        for (int gb = gb0; gb < gb1; gb++) {
            const int gy = gb / nbxb, gx = gb - gy * nbxb;
            const uint32_t c = bin_total[gb];
            if (gx >= sx0 && gx < sx1 && gy >= sy0 && gy < sy1) base[(gy - sy0) * sw + (gx - sx0)] = run + table[(size_t)blk * table_stride + gb];
            run += c;
        }
    }
    KSTAMP(2);
    // (a round of this kernel holds STEPS * 64 ranks; `rounds` counts the count kernel's 2048-rank rounds per workgroup)
    constexpr uint32_t ROUND_RANKS = (uint32_t)STEPS * WAVE;
    const uint32_t my_rounds = BIG ? rounds * (BIN_RANKS_PER_BLOCK / ROUND_RANKS) : 1u;
    for (uint32_t rd = 0; rd < my_rounds; rd++) {
    const uint32_t rbegin = blk * (BIG ? rounds : 1u) * BIN_RANKS_PER_BLOCK + rd * ROUND_RANKS;
    if (BIG && rbegin >= n) break;
    if (BIG && rd) __syncthreads();   // the previous round's slots are placed: its lane sets and group offsets can go (base[] carries on)
    for (int b = threadIdx.x; b < PAIR_WORDS * cap_s; b += SCAT_THREADS) pair[b] = 0;
    for (int b = threadIdx.x; b < STEPS * nmask; b += SCAT_THREADS) masks[b] = 0;
    __syncthreads();
    KSTAMP(1);

    // this wave's 2 steps of 64 consecutive ranks; rectangles clipped to the sub-grid, in sub-grid coordinates
    const uint32_t gbegin = rbegin + group * (GROUP_STEPS * WAVE);
    uint32_t idx[SPW];
    BinRect br[SPW];
    uint32_t raw[CELLS ? SPW : 1];   // the cell pass hands the rectangle (in bins) on to level two
#pragma unroll
    for (int k = 0; k < SPW; k++) {
        const uint32_t r = gbegin + (sub * SPW + k) * WAVE + lane;
        idx[k] = (r < n) ? depth_index[r] : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < SPW; k++) {
        const uint32_t r = gbegin + (sub * SPW + k) * WAVE + lane;
        const uint32_t packed = (r < n) ? rects[r] : RECT_NONE;
        if (CELLS) raw[k] = packed;
        BinRect b = unpack_rect(packed, shift);
        if (b.x0 <= b.x1) {
            b.x0 = max(b.x0, sx0) - sx0; b.x1 = min(b.x1, sx1 - 1) - sx0;
            b.y0 = max(b.y0, sy0) - sy0; b.y1 = min(b.y1, sy1 - 1) - sy0;
            if (b.x0 > b.x1 || b.y0 > b.y1) { b.x0 = 1; b.x1 = 0; b.y0 = 1; b.y1 = 0; }
        }
        br[k] = b;
    }
    // phase 1: per-group counts, and every lane ORs its bit into the column/row lane sets of its box
    const uint32_t one = 1u << myshift;
    const unsigned long long mybit = 1ull << lane;
#pragma unroll
    for (int k = 0; k < SPW; k++) {
        const BinRect b = br[k];
        unsigned long long* colm = gmask + (sub * SPW + k) * nmask;
        unsigned long long* rowm = colm + sl.w;
        for (int x = b.x0; x <= b.x1; x++) atomicOr(&colm[x], mybit);
        for (int y = b.y0; y <= b.y1; y++) {
            if (b.x0 <= b.x1) atomicOr(&rowm[y], mybit);
            for (int x = b.x0; x <= b.x1; x++) {
                GSR_BOUND(bin, 2, y * sw + x, nb_s);
                atomicAdd(&mypair[y * sw + x], one);
            }
        }
    }
    __syncthreads();
    KSTAMP(3);
    // phase 2: counts -> exclusive offsets of the groups inside the workgroup's run (16 bits each, in place).
    // BIG: base[] moves on to the next round's start here, and the offsets are stored relative to THAT (negative, 16-bit
    // two's complement: a round holds 2048 ranks), so that phase 3 needs no second per-bin word
    for (int b = threadIdx.x; b < nb_s; b += SCAT_THREADS) {
        uint32_t c[PAIR_WORDS], total = 0;
#pragma unroll
        for (int wd = 0; wd < PAIR_WORDS; wd++) {
            c[wd] = pair[wd * cap_s + b];
            total += (c[wd] & 0xffffu) + (c[wd] >> 16);
        }
        uint32_t run = BIG ? 0u - total : 0u;
#pragma unroll
        for (int wd = 0; wd < PAIR_WORDS; wd++) {
            const uint32_t lo = run, hi = run + (c[wd] & 0xffffu);
            run = hi + (c[wd] >> 16);
            pair[wd * cap_s + b] = (lo & 0xffffu) | (hi << 16);
        }
        if (BIG) base[b] += total;
    }
    __syncthreads();
    KSTAMP(4);
    // phase 3: slots.  The set of lanes of step s covering bin (X,Y) is col[s][X] & row[s][Y]; a splat's slot is
    // base[bin] + its group's offset + the sizes of the sets of the group's earlier steps + the number of
    // lower lanes in its own step's set: input order, from ballot-style arithmetic on LDS words that are
    // read-only by now (no ordered atomics, no running counter).  The row sets of a box row are read once per row.
#pragma unroll
    for (int k = 0; k < SPW; k++) {
        const BinRect b = br[k];
        const int st = sub * SPW + k;
        const uint32_t myidx = idx[k];
        for (int y = b.y0; y <= b.y1; y++) {
            unsigned long long rowm[GROUP_STEPS];
#pragma unroll
            for (int e = 0; e < GROUP_STEPS; e++) rowm[e] = (e <= st) ? gmask[e * nmask + sl.w + y] : 0ull;
            for (int x = b.x0; x <= b.x1; x++) {
                const uint32_t off16 = (mypair[y * sw + x] >> myshift) & 0xffffu;
                uint32_t dst = base[y * sw + x] + (BIG ? (uint32_t)(int32_t)(int16_t)off16 : off16);
                unsigned long long own = 0ull;
#pragma unroll
                for (int e = 0; e < GROUP_STEPS; e++) {
                    if (e > st) continue;                                 // wave-uniform
                    const unsigned long long m = gmask[e * nmask + x] & rowm[e];
                    if (e < st) dst += (uint32_t)__popcll(m);             // entries the earlier steps of this group put into the bin
                    else own = m;
                }
                dst += lanes_below64(own);
                GSR_BOUND(bin, 0, myidx, 0xfffffff0u);
                if (dst >= capacity) atomicOr(overflow, 1u);
                else if (CELLS) reinterpret_cast<uint2*>(list)[dst] = make_uint2(myidx, raw[k]);
                else list[dst] = myidx;
            }
        }
    }
    }   // rounds
#ifdef GSR_KSTAMPS
    __syncthreads();
    KSTAMP(5);
#endif
}

#define GSR_SCATTER_PARAMS                                                                                               \
    const uint32_t *__restrict__ depth_index, const uint32_t *__restrict__ rects, const uint32_t *__restrict__ count,   \
        BinGrid g, BinSlices sl, const uint32_t *__restrict__ table, const uint32_t *__restrict__ bin_total,            \
        const uint32_t *__restrict__ bin_start, uint32_t *__restrict__ list, uint32_t capacity,                         \
        uint32_t *__restrict__ overflow, uint32_t rounds, FinalizeArgs fa, CellArgs ca
#define GSR_SCATTER_ARGS depth_index, rects, count, g, sl, table, bin_total, bin_start, list, capacity, overflow, rounds, fa, ca
template <int GROUPS, bool FUSED>
__global__ __launch_bounds__(SCAT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bin_scatter(GSR_SCATTER_PARAMS)
{
    bin_scatter_body<GROUPS, FUSED, false, SCAT_STEPS_PER_WAVE>(GSR_SCATTER_ARGS);
}
// level one of the two-level binning: the same pass over a grid of CELLS, entries = (splat index, rectangle in bins)
template <int GROUPS>
__global__ __launch_bounds__(SCAT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_cell_scatter1(GSR_SCATTER_PARAMS)
{
    bin_scatter_body<GROUPS, true, false, SCAT_STEPS_PER_WAVE, true>(GSR_SCATTER_ARGS);
}
template <int GROUPS, int SPW>
__global__ __launch_bounds__(SCAT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_bin_scatter_big(GSR_SCATTER_PARAMS)
{
    bin_scatter_body<GROUPS, false, true, SPW>(GSR_SCATTER_ARGS);
}
#undef GSR_SCATTER_PARAMS
#undef GSR_SCATTER_ARGS


// ---- level two ----
struct CellGeom { int32_t ncx, ncells, nbxb, nby; };   // cells across, cells, bins across / down (of the band)

// chunk -> its cell and its entries [e0, e1) of the cell list (uniform; chunk < chunk_start[ncells])
__device__ __forceinline__ void chunk_span(uint32_t chunk, const uint4* __restrict__ chunk_info, uint32_t capacity, int& cell, uint32_t& e0, uint32_t& e1)
{
    const uint4 ci = chunk_info[chunk];
    cell = (int)ci.x;
    e0 = min(ci.y, capacity);
    e1 = min(ci.z, capacity);
}

// the columns (bits 0..3) and rows (bits 4..7) of the cell at (fx0, fy0) that a rectangle covers; 0: none
__device__ __forceinline__ uint32_t cell_cols_rows(uint32_t rect, int fx0, int fy0)
{
    const BinRect r = unpack_rect(rect);
    const int lx0 = max(r.x0 - fx0, 0), lx1 = min(r.x1 - fx0, CELL_SIDE - 1);
    const int ly0 = max(r.y0 - fy0, 0), ly1 = min(r.y1 - fy0, CELL_SIDE - 1);
    if (lx0 > lx1 || ly0 > ly1) return 0u;
    const uint32_t cols = ((2u << lx1) - 1u) & ~((1u << lx0) - 1u);
    const uint32_t rows = ((2u << ly1) - 1u) & ~((1u << ly0) - 1u);
    return cols | (rows << CELL_SIDE);
}
// ... as one bit per bin: bit ly * 4 + lx
__device__ __forceinline__ uint32_t cell_mask(uint32_t cr)
{
    const uint32_t cols = cr & 15u, rows = cr >> CELL_SIDE;
    const uint32_t rowexp = (rows & 1u) | ((rows & 2u) << 3) | ((rows & 4u) << 6) | ((rows & 8u) << 9);
    return cols * rowexp;
}

// entries of every bin of the cell among this wave's two steps: bin f's count in lane f.  Every lane spreads its mask's
// 16 bits over the bytes of four words, the words are summed over the wave (row sums with four DPP steps, the four rows
// through scalar registers), and lane f picks byte f: ~75 instructions instead of 32 ballots and population counts.
__device__ __forceinline__ uint32_t dpp_row_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);   // row_mirror
    return v;   // every lane: the sum over its row of 16
}
__device__ __forceinline__ uint32_t cell_wave_counts(uint32_t m0, uint32_t m1, int lane)
{
    uint32_t t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t w = ((((m0 >> (4 * j)) & 15u) * 0x00204081u) & 0x01010101u) + ((((m1 >> (4 * j)) & 15u) * 0x00204081u) & 0x01010101u);
        const uint32_t r = dpp_row_sum(w);   // <= 32 per byte
        t[j] = (uint32_t)__builtin_amdgcn_readlane((int)r, 0) + (uint32_t)__builtin_amdgcn_readlane((int)r, 16) +
               (uint32_t)__builtin_amdgcn_readlane((int)r, 32) + (uint32_t)__builtin_amdgcn_readlane((int)r, 48);   // <= 128 per byte
    }
    const int j = lane >> 2;
    const uint32_t w = j == 0 ? t[0] : j == 1 ? t[1] : j == 2 ? t[2] : t[3];
    return (w >> ((lane & 3) * 8)) & 0xffu;   // (lanes >= 16: of no use)
}

// table2[chunk][bin of the cell] = entries the chunk puts into the bin; wcnt[chunk][bin][wave] = those of each wave's 128
// entries (one byte each), which is what a wave of k_cell_scatter2 needs to know about the waves in front of it
__global__ __launch_bounds__(SCAT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_cell_count(const uint2* __restrict__ cell_list, const uint4* __restrict__ chunk_info, const uint32_t* __restrict__ chunk_start,
                  CellGeom cg, uint32_t capacity, uint32_t* __restrict__ table2, uint8_t* __restrict__ wcnt)
{
    __shared__ uint32_t s_cnt[CELL_FINE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t total_chunks = chunk_start[cg.ncells];
    for (uint32_t chunk = blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
        if (threadIdx.x < CELL_FINE) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        int cell; uint32_t e0, e1;
        chunk_span(chunk, chunk_info, capacity, cell, e0, e1);
        const int fx0 = (cell % cg.ncx) * CELL_SIDE, fy0 = (cell / cg.ncx) * CELL_SIDE;
        uint32_t m[SCAT_STEPS_PER_WAVE];
#pragma unroll
        for (int k = 0; k < SCAT_STEPS_PER_WAVE; k++) {
            const uint32_t e = e0 + (uint32_t)((wave * SCAT_STEPS_PER_WAVE + k) * WAVE + lane);
            m[k] = e < e1 ? cell_mask(cell_cols_rows(cell_list[e].y, fx0, fy0)) : 0u;
        }
        const uint32_t mine = cell_wave_counts(m[0], m[1], lane);
        if (lane < CELL_FINE) {
            wcnt[(size_t)chunk * (CELL_FINE * SCAT_WAVES) + lane * SCAT_WAVES + wave] = (uint8_t)mine;   // (<= 128)
            if (mine) atomicAdd(&s_cnt[lane], mine);
        }
        __syncthreads();
        if (threadIdx.x < CELL_FINE) table2[(size_t)chunk * CELL_FINE + threadIdx.x] = s_cnt[threadIdx.x];
    }
}

// per cell (one wave): table2 scanned down the cell's chunks for each of its 16 bins; the bins' totals.  Lane = (bin,
// one of four consecutive chunks).
__global__ __launch_bounds__(WAVE) void k_cell_scan(uint32_t* __restrict__ table2, const uint32_t* __restrict__ chunk_start, CellGeom cg,
                                                    uint32_t capacity, uint32_t* __restrict__ bin_total)
{
    const int cell = blockIdx.x, lane = threadIdx.x, f = lane & (CELL_FINE - 1), ph = lane >> 4;
    // a frame that does not fit has no chunks (cell_finalize_body): its need goes into the first bin's total, where the
    // finalize step finds it
    const uint32_t need = chunk_start[cg.ncells + 1];
    const uint32_t c0 = chunk_start[cell], c1 = need <= capacity ? chunk_start[cell + 1] : c0;
    uint32_t run = 0;
    for (uint32_t c = c0; c < c1; c += 4) {
        const bool ok = c + ph < c1;
        uint32_t* p = table2 + (size_t)(c + ph) * CELL_FINE + f;
        const uint32_t v = ok ? *p : 0u;
        uint32_t inc = v;
        const uint32_t u1 = __shfl_up(inc, 16);
        if (ph >= 1) inc += u1;
        const uint32_t u2 = __shfl_up(inc, 32);
        if (ph >= 2) inc += u2;
        if (ok) *p = run + inc - v;
        run += __shfl(inc, 48 + f);
    }
    if (cell == 0 && f == 0 && need > capacity) run = need;
    if (ph == 0) {
        const int bx = (cell % cg.ncx) * CELL_SIDE + (f & (CELL_SIDE - 1)), by = (cell / cg.ncx) * CELL_SIDE + (f >> CELL_SHIFT);
        if (bx < cg.nbxb && by < cg.nby) bin_total[by * cg.nbxb + bx] = run;
    }
}

// One store instruction of k_cell_scatter2: the lanes of set `b` write `val` to list[sb + (set lanes below)].  The set is
// the AND of two scalar lane sets, so it becomes the exec mask as it is -- no per-lane test, three vector instructions.
__device__ __forceinline__ void store_set(uint64_t b, uint32_t sb, uint32_t val, uint32_t* __restrict__ list)
{
    uint32_t t;
    uint64_t sv;
    asm volatile("s_and_saveexec_b64 %[sv], %[b]\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], %[blo], 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], %[bhi], %[t]\n\t"
                 "v_add_lshl_u32 %[t], %[t], %[sb], 2\n\t"
                 "global_store_dword %[t], %[val], %[base]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [t] "=&v"(t), [sv] "=&s"(sv)
                 : [b] "s"(b), [blo] "s"((uint32_t)b), [bhi] "s"((uint32_t)(b >> 32)), [sb] "s"(sb), [val] "v"(val), [base] "s"(list)
                 : "memory", "scc");
}

// the lists: entry e of a chunk goes to every bin of its mask, at the bin's start (k_bin_starts) + what earlier chunks of
// the cell put there (table2) + what earlier waves (wcnt), steps and lanes of this chunk do.  An entry's mask is columns x
// rows, so the set of lanes of a step that cover bin (x, y) is colset[x] & rowset[y]: eight ballots per step and scalar
// ANDs give all sixteen sets.  No LDS, no barrier.  Workgroup 0 is the finalize step.
__global__ __launch_bounds__(SCAT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_cell_scatter2(const uint2* __restrict__ cell_list, const uint4* __restrict__ chunk_info, const uint32_t* __restrict__ chunk_start,
                     CellGeom cg, const uint32_t* __restrict__ table2, const uint8_t* __restrict__ wcnt, const uint32_t* __restrict__ bin_start,
                     uint32_t* __restrict__ list, uint32_t capacity, FinalizeArgs fa)
{
    static_assert(FIN_THREADS == SCAT_THREADS, "the finalize step runs as a workgroup of this kernel");
    static_assert(SCAT_WAVES == 16, "a bin's per-wave counts are one 16-byte load");
    extern __shared__ uint32_t s_fin[];   // FIN_SCRATCH_WORDS
    if (blockIdx.x == 0) {
        bin_finalize_body(fa, s_fin);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t total_chunks = chunk_start[cg.ncells];
    for (uint32_t chunk = blockIdx.x - 1u; chunk < total_chunks; chunk += gridDim.x - 1u) {
        int cell; uint32_t e0, e1;
        chunk_span(chunk, chunk_info, capacity, cell, e0, e1);
        const int fx0 = (cell % cg.ncx) * CELL_SIDE, fy0 = (cell / cg.ncx) * CELL_SIDE;
        // lane f < 16: this wave's first slot in bin f
        uint32_t wb = 0;
        if (lane < CELL_FINE) {
            const int bx = fx0 + (lane & (CELL_SIDE - 1)), by = fy0 + (lane >> CELL_SHIFT);
            const uint4 wc = *reinterpret_cast<const uint4*>(wcnt + (size_t)chunk * (CELL_FINE * SCAT_WAVES) + lane * SCAT_WAVES);
            const uint32_t w4[4] = {wc.x, wc.y, wc.z, wc.w};
            if (bx < cg.nbxb && by < cg.nby) wb = bin_start[by * cg.nbxb + bx] + table2[(size_t)chunk * CELL_FINE + lane];
#pragma unroll
            for (int j = 0; j < 4; j++) {   // bytes of the waves in front of mine
                const int nb = min(max(wave - 4 * j, 0), 4);
                const uint32_t keep = nb == 4 ? 0xffffffffu : (1u << (8 * nb)) - 1u;
                wb = __builtin_amdgcn_sad_u8(w4[j] & keep, 0u, wb);
            }
        }
        uint32_t idx[SCAT_STEPS_PER_WAVE];
        uint64_t colset[SCAT_STEPS_PER_WAVE][CELL_SIDE], rowset[SCAT_STEPS_PER_WAVE][CELL_SIDE];
#pragma unroll
        for (int k = 0; k < SCAT_STEPS_PER_WAVE; k++) {
            const uint32_t e = e0 + (uint32_t)((wave * SCAT_STEPS_PER_WAVE + k) * WAVE + lane);
            const uint2 en = e < e1 ? cell_list[e] : make_uint2(0u, RECT_NONE);
            idx[k] = en.x;
            GSR_BOUND(bin, 0, idx[k], 0xfffffff0u);
            const uint32_t cr = e < e1 ? cell_cols_rows(en.y, fx0, fy0) : 0u;
#pragma unroll
            for (int x = 0; x < CELL_SIDE; x++) {
                colset[k][x] = __ballot((cr >> x) & 1u);
                rowset[k][x] = __ballot((cr >> (CELL_SIDE + x)) & 1u);
            }
        }
#pragma unroll
        for (int f = 0; f < CELL_FINE; f++) {
            uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)wb, f);
#pragma unroll
            for (int k = 0; k < SCAT_STEPS_PER_WAVE; k++) {
                const uint64_t b = colset[k][f & (CELL_SIDE - 1)] & rowset[k][f >> CELL_SHIFT];
                if (b == 0ull) continue;
                // (inside the list by construction: a frame whose need exceeds the capacity has no chunks)
                GSR_BOUND(bin, 2, sb + (uint32_t)__popcll(b) - 1u, capacity);
                store_set(b, sb, idx[k], list);
                sb += (uint32_t)__popcll(b);
            }
        }
    }
}

static void set_scatter_lds_attribute()
{
    // dynamic LDS above the default needs the attribute raised (1080p: 64 KiB, 4K: 8160 bins -> 146 KiB).  The attribute
    // belongs to the current device's copy of each instantiation: raised to the budget once per device for all of them.
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev >= 0 && dev < 64 ? dev : 0], [] {
        const int want = (int)(SCAT_LDS_BUDGET + 1024);
        for (const void* fn : {(const void*)k_bin_scatter<8, true>, (const void*)k_bin_scatter<8, false>,
                               (const void*)k_bin_scatter<4, true>, (const void*)k_bin_scatter<4, false>,
                               (const void*)k_bin_scatter_big<8, 2>, (const void*)k_bin_scatter_big<4, 2>,
                               (const void*)k_bin_scatter_big<4, 1>, (const void*)k_cell_scatter1<8>, (const void*)k_cell_scatter1<4>})
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want);
        (void)hipGetLastError();  // a failure shows up as the launch error
    });
}

static FinalizeArgs make_finalize_args(const BinBuffers& b, int nbins, uint32_t n)
{
    return FinalizeArgs{b.bin_total, nbins, b.seg_len, b.seg_target_items, b.items_by_size, b.seg_len_dev, b.max_items, b.capacity,
                        b.slots, n ? 1u : 0u, b.bin_start, b.seg_start, b.items, b.overflow, b.visible, b.tile_entries, b.accum,
                        b.report, b.queue, b.queue_start, b.mailbox, b.bin_mask, b.long_policy, b.seg_len_long, b.long_tau, b.npix, b.long_tiles_x2, b.long_tau_bin, b.long_mass_min};
}

// cells across / down a grid of bins
inline int cells_of(int bins) { return (bins + CELL_SIDE - 1) >> CELL_SHIFT; }

// Two levels (see "Two-level binning" above): count / scan / scatter over the cells, then count / scan / scatter of the
// cell lists' chunks into the bins; the finalize step is the first workgroup of the last kernel.
static void launch_bin_two_level(const BinBuffers& b, const BinGrid& g, hipStream_t s)
{
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    const int ncx = cells_of(nbxb), ncy = cells_of(g.nby), ncells = ncx * ncy;
    BinGrid gc = g;
    gc.nbx = ncx; gc.nby = ncy; gc.bx_lo = 0; gc.bx_hi = ncx;
    const BinSlices sl = make_slices(ncx, ncy);
    const bool eight = scatter_lds_bytes(sl.w, sl.h, 8) <= SCAT_LDS_TWO_PER_CU;
    const size_t lds1 = scatter_lds_bytes(sl.w, sl.h, eight ? 8 : 4);
    set_scatter_lds_attribute();
    const FinalizeArgs fa = make_finalize_args(b, nbins, 1u);
    const CellArgs ca{b.cell_start, b.chunk_start, reinterpret_cast<uint4*>(b.chunk_info), CELL_SHIFT};
    const uint4* ci = reinterpret_cast<const uint4*>(b.chunk_info);
    const CellGeom cg{ncx, ncells, nbxb, g.nby};
    // level one (b.nblocks workgroups of 2048 ranks; the table's last column sums the rectangles' areas in bins)
    hipLaunchKernelGGL(k_bin_count, dim3(b.nblocks), dim3(CNT_THREADS), (size_t)(ncells + 1) * sizeof(uint32_t), s, b.depth_index, b.rect_idx,
                       b.count, gc, ncy, 1u, b.table, b.rects, CELL_SHIFT, (int)b.rects_sorted, b.n_max);
    launch_column_scan(b.table, b.cell_total, ncells + 1, b.nblocks, s, b.band ? b.count : nullptr, BIN_RANKS_PER_BLOCK);
    {
        const dim3 grid(b.nblocks + 1), block(SCAT_THREADS);
#define GSR_LAUNCH_CELLS(K)                                                                                                          \
    hipLaunchKernelGGL((K), grid, block, lds1, s, b.depth_index, (const uint32_t*)b.rects, b.count, gc, sl, (const uint32_t*)b.table, \
                       (const uint32_t*)b.cell_total, (const uint32_t*)nullptr, b.cell_list, b.capacity, b.overflow, 1u, fa, ca)
        if (eight) GSR_LAUNCH_CELLS(k_cell_scatter1<8>);
        else GSR_LAUNCH_CELLS(k_cell_scatter1<4>);
#undef GSR_LAUNCH_CELLS
    }
    // level two
    const uint2* cl = reinterpret_cast<const uint2*>(b.cell_list);
    hipLaunchKernelGGL(k_cell_count, dim3(b.cell_grid), dim3(SCAT_THREADS), 0, s, cl, ci, (const uint32_t*)b.chunk_start, cg,
                       b.capacity, b.cell_table2, reinterpret_cast<uint8_t*>(b.cell_wcnt));
    hipLaunchKernelGGL(k_cell_scan, dim3(ncells), dim3(WAVE), 0, s, b.cell_table2, (const uint32_t*)b.chunk_start, cg, b.capacity, b.bin_total);
    hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(FIN_THREADS), 0, s, (const uint32_t*)b.bin_total, nbins, b.bin_start_pre);
    hipLaunchKernelGGL(k_cell_scatter2, dim3(b.cell_grid + 1), dim3(SCAT_THREADS), FIN_SCRATCH_WORDS * sizeof(uint32_t), s, cl, ci,
                       (const uint32_t*)b.chunk_start, cg, (const uint32_t*)b.cell_table2, reinterpret_cast<const uint8_t*>(b.cell_wcnt),
                       (const uint32_t*)b.bin_start_pre, b.list, b.capacity, fa);
}

void launch_bin(const BinBuffers& b, const BinGrid& g, uint32_t n, hipStream_t s)
{
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    if (nbins <= 0) return;
    const BinSlices sl = make_slices(nbxb, g.nby);
    // 8 groups of 4 steps where their table still lets two workgroups share a CU, else 4 groups of 8 steps
    const bool eight = scatter_lds_bytes(sl.w, sl.h, 8) <= SCAT_LDS_TWO_PER_CU;
    // (the finalize step, when it runs as this kernel's extra workgroup, uses FIN_SCRATCH_WORDS of the dynamic LDS)
    // the large-grid form with 1024-rank rounds (one step per wave): 4 groups of 4 steps -- ~6 instead of ~11 LDS reads per
    // list entry in the slot phase, the phase the kernel is bound by -- at twice the per-round fixed work (b.big == 2)
    // (at 1080p the large-grid form loses to the two-workgroups-per-CU kernel: C3 binning 47.3 -> 54.1 us, measured)
    const bool short_rounds = n && nbins > 4096 && b.big == 2 && !eight;
    const size_t lds = std::max(scatter_lds_bytes(sl.w, sl.h, eight ? 8 : 4, short_rounds ? SCAT_WAVES : SCAT_STEPS), FIN_SCRATCH_WORDS * sizeof(uint32_t));
    set_scatter_lds_attribute();
    if (n && b.two_level) {
        launch_bin_two_level(b, g, s);
        return;
    }
    // the count pass keeps one counter per bin in LDS and is cut into row slices only beyond 12288 bins (above 4K)
    const int cnt_slices = (nbins + CNT_MAX_BINS - 1) / CNT_MAX_BINS;
    const int cnt_rows = (g.nby + cnt_slices - 1) / cnt_slices;
    if (n) {
        hipLaunchKernelGGL(k_bin_count, dim3(b.nblocks, (g.nby + cnt_rows - 1) / cnt_rows), dim3(CNT_THREADS),
                           (size_t)cnt_rows * nbxb * sizeof(uint32_t), s, b.depth_index, b.rect_idx, b.count, g, cnt_rows, b.rounds, b.table, b.rects, 0, (int)b.rects_sorted, b.n_max);
        launch_column_scan(b.table, b.bin_total, nbins, b.nblocks, s, b.band ? b.count : nullptr, b.rounds * BIN_RANKS_PER_BLOCK);
    }
    const FinalizeArgs fa = make_finalize_args(b, nbins, n);
    const bool fused = n && nbins <= 4096;   // see bin_scatter_body
    const bool big = n && !fused && b.big;   // the large-grid form: finalize as the first workgroup, rounds
    if (!fused && !big) hipLaunchKernelGGL(k_bin_finalize, dim3(1), dim3(FIN_THREADS), FIN_SCRATCH_WORDS * sizeof(uint32_t), s, fa);
    if (n) {
        if (big) hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(FIN_THREADS), 0, s, (const uint32_t*)b.bin_total, nbins, b.bin_start_pre);
        const dim3 grid(b.nblocks + ((fused || big) ? 1 : 0), sl.sx * sl.sy), block(SCAT_THREADS);
#define GSR_LAUNCH_SCATTER(K, STARTS)                                                                                               \
    hipLaunchKernelGGL((K), grid, block, lds, s, b.depth_index, (const uint32_t*)b.rects, b.count, g, sl,                           \
                       (const uint32_t*)b.table, (const uint32_t*)b.bin_total, (const uint32_t*)(STARTS), b.list, b.capacity,       \
                       b.overflow, b.rounds, fa, CellArgs{nullptr, nullptr, nullptr, 0})
        if (big && eight) GSR_LAUNCH_SCATTER((k_bin_scatter_big<8, 2>), b.bin_start_pre);
        else if (big && short_rounds) GSR_LAUNCH_SCATTER((k_bin_scatter_big<4, 1>), b.bin_start_pre);
        else if (big) GSR_LAUNCH_SCATTER((k_bin_scatter_big<4, 2>), b.bin_start_pre);
        else if (eight && fused) GSR_LAUNCH_SCATTER((k_bin_scatter<8, true>), b.bin_start);
        else if (eight) GSR_LAUNCH_SCATTER((k_bin_scatter<8, false>), b.bin_start);
        else if (fused) GSR_LAUNCH_SCATTER((k_bin_scatter<4, true>), b.bin_start);
        else GSR_LAUNCH_SCATTER((k_bin_scatter<4, false>), b.bin_start);
#undef GSR_LAUNCH_SCATTER
    }
}

}  // namespace gsr
