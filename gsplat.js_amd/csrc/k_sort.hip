// Depth-key quantisation (wasm/wasm.cpp:33-40) and a stable 2-pass LSD radix
// sort of the 17-bit keys (8 low bits, then 9 high bits), which reproduces the
// reference's stable counting sort (wasm/wasm.cpp:42-51) exactly: ascending
// key, ties by ascending original index.
//
// Each pass is histogram -> per-digit scan -> stable scatter.  Stability inside
// a workgroup comes from wave64 ballot matching: for every key the set of lanes
// holding the same digit is built from one __ballot per digit bit, and a key's
// rank among equal digits is the population count of that set below its lane.
//
// Compiled with -ffp-contract=off (the quantisation is f32 arithmetic that
// must match the reference bit for bit).
#include "gsr_internal.h"

#include <mutex>

namespace gsr {

GSR_BOUNDS_DECL(sort)   // sites: 0 destination of a radix pass, 1 its LDS position, 2 destination of the bucket sort, 3 key above 65536, 4 band mode: a survivor's packed slot / original index
constexpr int SORT_THREADS = 256;

#ifdef GSR_KSTAMPS
// Diagnostic build only: per-workgroup phase times of the two k_scatter passes (s_memrealtime ticks, 10 ns).
__device__ unsigned int g_sort_stamps[2 * 4096 * 8];
#define KSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_sort_stamps[((FIRST ? 0 : 1) * 4096 + blockIdx.x) * 8 + (slot)] = (unsigned int)__builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int gsr_debug_sort_stamps(unsigned int* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_stamps), sizeof g_sort_stamps) == hipSuccess ? 0 : -1; }
#else
#define KSTAMP(slot)
#endif

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    // number of set bits of `mask` in lanes lower than this one (v_mbcnt)
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Lanes of this wave whose `digit` equals mine (lanes with valid == false match nobody).
template <int BITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t digit, bool valid)
{
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return valid ? m : 0ull;
}

// ---------------------------------------------------------------------------
// A2: q = (uint32)((float)(uint32)(depth - minDepth) * depthInv), plus the
// workgroup histogram of the pass-1 digit.
//
// The depth bounds arrive as the FRAME_SLOTS partial (min, max) pairs of k_project_key; every workgroup here folds
// them itself (64 loads that hit L2) instead of waiting for a one-workgroup reduction kernel, and workgroup 0 stores
// the result for read-backs.
//
// Band mode (multi-GPU, `live` set): a splat whose bin rectangle is empty for this context's band (culled, or outside
// the band) is absent from the band's frame.  k_project_key has packed every 256-splat block's survivors to the front
// of the block's depth slots and counted them; k_kept_scan turns the counts into offsets (koff[block]; the survivors S
// = the frame's *count), k_band_gather moves the survivors' depths and original indices to [0, S), in index order, and
// everything behind -- this pass, both radix passes, their scans, the binning -- runs on S keys: workgroups and table
// rows past S do nothing.  The survivors' sorted order is the restriction of the global order, so the band's pixels
// are unchanged.
// ---------------------------------------------------------------------------
constexpr int KS_THREADS = 1024;
constexpr int KS_WAVES = KS_THREADS / WAVE;
constexpr int KS_BATCH = 4;
// One workgroup.  A wave takes 64 consecutive counts per load (coalesced; one thread per run of counts made the single CU
// fetch 64 lines per load: 20 us for C4's 19 532 blocks): pass 1 leaves the total of every 64 blocks in LDS, wave 0 scans
// those, pass 2 adds the scan inside the 64.
__global__ __launch_bounds__(KS_THREADS) void k_kept_scan(const uint32_t* __restrict__ kept, uint32_t nb, uint32_t* __restrict__ koff,
                                                          uint32_t* __restrict__ count)
{
    extern __shared__ uint32_t s_sup[];   // (nb + 63) / 64 words
    const uint32_t nsup = (nb + WAVE - 1u) / WAVE;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t sb = wave; sb < nsup; sb += KS_WAVES * KS_BATCH) {
        uint32_t v[KS_BATCH];
#pragma unroll
        for (int k = 0; k < KS_BATCH; k++) {
            const uint32_t b = (sb + k * KS_WAVES) * WAVE + lane;
            v[k] = b < nb ? kept[b] : 0u;
        }
#pragma unroll
        for (int k = 0; k < KS_BATCH; k++) {
            uint32_t t = v[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if (lane == 0 && sb + k * KS_WAVES < nsup) s_sup[sb + k * KS_WAVES] = t;
        }
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nsup; base += WAVE) {
            const uint32_t v = base + lane < nsup ? s_sup[base + lane] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t u = __shfl_up(inc, off);
                if ((int)lane >= off) inc += u;
            }
            if (base + lane < nsup) s_sup[base + lane] = carry + inc - v;
            carry += __shfl(inc, WAVE - 1);
        }
        if (lane == 0) { koff[nb] = carry; *count = carry; }
    }
    __syncthreads();
    for (uint32_t sb = wave; sb < nsup; sb += KS_WAVES * KS_BATCH) {
        uint32_t v[KS_BATCH];
#pragma unroll
        for (int k = 0; k < KS_BATCH; k++) {
            const uint32_t b = (sb + k * KS_WAVES) * WAVE + lane;
            v[k] = b < nb ? kept[b] : 0u;
        }
#pragma unroll
        for (int k = 0; k < KS_BATCH; k++) {
            const uint32_t b = (sb + k * KS_WAVES) * WAVE + lane;
            uint32_t inc = v[k];
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t u = __shfl_up(inc, off);
                if ((int)lane >= off) inc += u;
            }
            if (b < nb) koff[b] = s_sup[sb + k * KS_WAVES] + inc - v[k];
        }
    }
}

// survivor t of block b (its packed slot b * 256 + t) -> position koff[b] + t: depth and original index, dense
constexpr int BG_BLOCKS = 4;   // blocks of k_project_key per workgroup: one wave each, four slots per lane
__global__ __launch_bounds__(BG_BLOCKS * WAVE) void k_band_gather(const int32_t* __restrict__ depth, const uint8_t* __restrict__ kept_lane,
                                                                 const uint32_t* __restrict__ kept, const uint32_t* __restrict__ koff,
                                                                 uint32_t n, uint32_t nb, int32_t* __restrict__ depth_out,
                                                                 uint32_t* __restrict__ idx_out)
{
    const uint32_t b = blockIdx.x * BG_BLOCKS + (threadIdx.x >> 6);
    if (b >= nb) return;
    const uint32_t c = kept[b], o = koff[b];
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t k = 0; k < PROJ_THREADS / WAVE; k++) {
        const uint32_t t = k * WAVE + lane;
        if (t < c) {
            const uint32_t slot = b * PROJ_THREADS + t;
            GSR_BOUND(sort, 4, slot, n);
            depth_out[o + t] = depth[slot];
            idx_out[o + t] = b * PROJ_THREADS + kept_lane[slot];
        }
    }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_quantise_hist(const int32_t* __restrict__ depth,
                                                                const int32_t* __restrict__ slots,
                                                                int32_t* __restrict__ minmax_out, uint32_t n,
                                                                uint32_t keys_per_block, const uint32_t* __restrict__ live,
                                                                uint32_t* __restrict__ keys,
                                                                uint32_t* __restrict__ block_hist, int hist_shift, int hist_bins)
{
    // the histogram is over the digit of the FIRST scatter pass: the low 8 bits (LSD order) or the high 9 (bucket order)
    __shared__ uint32_t h_lo[RADIX_HI_BINS];
    __shared__ int32_t s_mm[2][SORT_THREADS / WAVE];
    for (int d = threadIdx.x; d < hist_bins; d += SORT_THREADS) h_lo[d] = 0;
    // the workgroup's first depths do not depend on the bounds: their loads go out in front of the fold below, so that the
    // two round trips (slots, depths) overlap instead of following each other
    const uint32_t begin = blockIdx.x * keys_per_block;
    if (live) n = *live;   // (band mode: depth[] holds the survivors, dense: k_band_gather)
    const uint32_t end = min(begin + keys_per_block, n);
    constexpr int QH_PRE = 8;   // (2048 keys per workgroup = 8 per thread: small scenes load everything up front)
    int32_t pre[QH_PRE];
#pragma unroll
    for (int k = 0; k < QH_PRE; k++) {
        const uint32_t i = begin + threadIdx.x + (uint32_t)k * SORT_THREADS;
        pre[k] = i < end ? depth[i] : 0;
    }
    // wasm.cpp:14-31's running min / max over ALL splats: fold the projection's per-workgroup pairs
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
    if (threadIdx.x < FRAME_SLOTS) {
        mn = slots[(size_t)threadIdx.x * FRAME_SLOT_WORDS];
        mx = slots[(size_t)threadIdx.x * FRAME_SLOT_WORDS + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, __shfl_xor(mn, off));
        mx = max(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) { s_mm[0][threadIdx.x >> 6] = mn; s_mm[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    int32_t minDepth = s_mm[0][0], maxDepth = s_mm[1][0];
#pragma unroll
    for (int w = 1; w < SORT_THREADS / WAVE; w++) { minDepth = min(minDepth, s_mm[0][w]); maxDepth = max(maxDepth, s_mm[1][w]); }
    if (blockIdx.x == 0 && threadIdx.x == 0) { minmax_out[0] = minDepth; minmax_out[1] = maxDepth; }

    const bool degenerate = (maxDepth == minDepth);
    // wasm.cpp:34: (float)depthRange / (maxDepth - minDepth): int subtract, int->f32 RNE, f32 divide RNE
    const float depthInv = degenerate ? 0.0f : (float)DEPTH_RANGE / (float)(maxDepth - minDepth);

    auto quantise = [&](uint32_t i, int32_t d) {
        // wasm.cpp:38: u32 wrap-around subtract, u32->f32 RNE, f32 multiply, truncate
        const uint32_t rel = (uint32_t)d - (uint32_t)minDepth;
        uint32_t q = degenerate ? 0u : (uint32_t)((float)rel * depthInv);
        q = min(q, DEPTH_RANGE);
        keys[i] = q;
        atomicAdd(&h_lo[(q >> hist_shift) & (uint32_t)(hist_bins - 1)], 1u);
    };
#pragma unroll
    for (int k = 0; k < QH_PRE; k++) {
        const uint32_t i = begin + threadIdx.x + (uint32_t)k * SORT_THREADS;
        if (i < end) quantise(i, pre[k]);
    }
    for (uint32_t i = begin + threadIdx.x + QH_PRE * SORT_THREADS; i < end; i += SORT_THREADS) quantise(i, depth[i]);
    __syncthreads();
    for (int d = threadIdx.x; d < hist_bins; d += SORT_THREADS) block_hist[(size_t)blockIdx.x * hist_bins + d] = h_lo[d];
}

// Workgroup histogram of the pass-2 digit over the pass-1 output order.
// (*count = keys the first pass kept: n, or the survivors in band mode)
__global__ __launch_bounds__(SORT_THREADS) void k_hist_hi(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ count,
                                                          uint32_t keys_per_block, uint32_t* __restrict__ block_hist)
{
    const uint32_t n = *count;
    __shared__ uint32_t h[RADIX_HI_BINS];
    for (int d = threadIdx.x; d < RADIX_HI_BINS; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    const uint32_t begin = blockIdx.x * keys_per_block;
    const uint32_t end = min(begin + keys_per_block, n);
    for (uint32_t i = begin + threadIdx.x; i < end; i += SORT_THREADS) atomicAdd(&h[keys[i] >> RADIX_LO_BITS], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX_HI_BINS; d += SORT_THREADS) block_hist[(size_t)blockIdx.x * RADIX_HI_BINS + d] = h[d];
}

// ---------------------------------------------------------------------------
// Column scan shared by the radix sort (digit histograms) and the binning (bin counts):
// exclusive prefix down the rows of a row-major table[nrows][ncols], in place, per column;
// total[col] = column sum.
//
// A workgroup owns 16 adjacent columns and every row:
// thread = (row slot, column), 32 row slots x 16 columns, each slot a contiguous range of rows, so a
// wave instruction touches four 64-byte row segments instead of 64 scattered words (the one-wave-per-
// column form reads a whole cache line per word: at 4K / 5 M splats that was 80 MB of table read as
// ~1.3 GB).  Two sweeps: sum the slot's rows, exchange the slot sums through LDS, then rewrite the
// rows with running prefixes.  Loads are issued in independent batches.
// (4 columns per workgroup for the narrow digit tables -- 64 / 128 workgroups instead of 16 / 32 -- was measured
// slower: sort 41.2 -> 45.3 us on C3.)
constexpr int CS_THREADS = 512;   // 512 measured best alone and with frames in flight (256/512/1024 within 1.5 %)
constexpr int CS_COLS = 16;
constexpr int CS_SLOTS = CS_THREADS / CS_COLS;  // 32
constexpr int CS_BATCH = 8;

// KEEP: a slot holds at most CS_KEEP rows (tables of up to CS_SLOTS * CS_KEEP = 512 rows: every scene up to 1 M splats),
// which stay in registers between the two sweeps -- one global round trip less in a kernel that is nothing but round trips.
constexpr int CS_KEEP = 16;
constexpr int CS_KEEP_LONG = 40;   // ... and tables of up to 1280 rows (C4's radix tables: 1221) with 40 registers: round 4

template <int KEEP>
__global__ __launch_bounds__(CS_THREADS) void k_column_scan(uint32_t* __restrict__ table, uint32_t* __restrict__ total,
                                                            int ncols, uint32_t nrows, const uint32_t* __restrict__ live, uint32_t live_unit)
{
    // (band mode: the frame holds *live keys or ranks, live_unit per row: the rows behind them were not written and are not read)
    if (live) nrows = min(nrows, (*live + live_unit - 1u) / live_unit);
    __shared__ uint32_t s_sum[CS_SLOTS][CS_COLS];
    const int c = threadIdx.x & (CS_COLS - 1);
    const int slot = threadIdx.x / CS_COLS;
    const int col = blockIdx.x * CS_COLS + c;
    const bool ok = col < ncols;
    const uint32_t per = (nrows + CS_SLOTS - 1) / CS_SLOTS;
    const uint32_t r0 = min((uint32_t)slot * per, nrows), r1 = min(r0 + per, nrows);
    uint32_t* p = table + (size_t)r0 * ncols + (ok ? col : 0);

    uint32_t sum = 0;
    uint32_t kept[KEEP ? KEEP : 1];
    if (KEEP) {
#pragma unroll
        for (int k = 0; k < KEEP; k++) kept[k] = (ok && r0 + k < r1) ? p[(size_t)k * ncols] : 0u;
#pragma unroll
        for (int k = 0; k < KEEP; k++) sum += kept[k];
    } else if (ok) {
        for (uint32_t r = r0; r < r1; r += CS_BATCH) {
            uint32_t v[CS_BATCH];
#pragma unroll
            for (int k = 0; k < CS_BATCH; k++) v[k] = (r + k < r1) ? p[(size_t)(r - r0 + k) * ncols] : 0u;
#pragma unroll
            for (int k = 0; k < CS_BATCH; k++) sum += v[k];
        }
    }
    s_sum[slot][c] = sum;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll 8
    for (int k = 0; k < CS_SLOTS; k++) {
        const uint32_t v = s_sum[k][c];
        base += (k < slot) ? v : 0u;
        tot += v;
    }
    if (ok) {
        uint32_t run = base;
        if (KEEP) {
#pragma unroll
            for (int k = 0; k < KEEP; k++) {
                if (r0 + k < r1) p[(size_t)k * ncols] = run;
                run += kept[k];
            }
        } else {
            for (uint32_t r = r0; r < r1; r += CS_BATCH) {
                uint32_t v[CS_BATCH];
#pragma unroll
                for (int k = 0; k < CS_BATCH; k++) v[k] = (r + k < r1) ? p[(size_t)(r - r0 + k) * ncols] : 0u;
#pragma unroll
                for (int k = 0; k < CS_BATCH; k++) {
                    if (r + k < r1) p[(size_t)(r - r0 + k) * ncols] = run;
                    run += v[k];
                }
            }
        }
        if (slot == 0) total[col] = tot;
    }
}

void launch_column_scan(uint32_t* table, uint32_t* total, int ncols, uint32_t nrows, hipStream_t s, const uint32_t* live, uint32_t live_unit)
{
    const dim3 grid((ncols + CS_COLS - 1) / CS_COLS), block(CS_THREADS);
    if (nrows <= (uint32_t)CS_SLOTS * CS_KEEP) hipLaunchKernelGGL(k_column_scan<CS_KEEP>, grid, block, 0, s, table, total, ncols, nrows, live, live_unit);
    else if (nrows <= (uint32_t)CS_SLOTS * CS_KEEP_LONG) hipLaunchKernelGGL(k_column_scan<CS_KEEP_LONG>, grid, block, 0, s, table, total, ncols, nrows, live, live_unit);
    else hipLaunchKernelGGL(k_column_scan<0>, grid, block, 0, s, table, total, ncols, nrows, live, live_unit);
}

// ---------------------------------------------------------------------------
// Stable scatter of one pass.  Workgroup = 16 waves over keys_per_block keys (2048 .. 8192); wave w owns the w-th
// sixteenth (contiguous, 2 .. 8 steps of 64 keys), so the order of equal digits is: workgroup, then wave, then step,
// then lane = input order.
//   phase 1  count digits per wave (LDS)
//   phase 2  digit starts over all workgroups (scan of the column totals), this workgroup's run of every digit
//            (from the scanned table), and the LOCAL layout: the workgroup's keys ordered by digit, wave, input order
//   phase 3  rank with ballot matching and write (key, index) into LDS at the local position
//   phase 4  stream the LDS image out: consecutive lanes hold consecutive local positions, so within a digit they store
//            to consecutive addresses.
// Phase 4 is what the pass is about.  Storing straight from phase 3 gave every lane of a wave its own destination
// line: 64 cache lines per store instruction, and the address path -- not HBM -- set the kernel's time (in-kernel
// stamps on C3: 4.4 of a workgroup's 7.1 us in the rank-and-store phase; at 20 M splats the pass reached 15 % of the
// HBM roofline).  Staged, a store instruction covers one run per digit present among its 64 local positions:
// keys_per_block / 2^BITS keys per run on average (8 for 2048 keys and 8 bits, 32 for 8192).  Larger scenes take
// larger blocks (launch_sort) so that the runs are whole cache lines.
// ---------------------------------------------------------------------------
constexpr int SCAT_THREADS = 1024;
constexpr int SCAT_WAVES = SCAT_THREADS / WAVE;
constexpr int SCAT_MAX_STEPS = 8;  // keys_per_block <= SCAT_THREADS * SCAT_MAX_STEPS = 8192
// (the bucket order's second kernel, k_local_sort below: one workgroup per LOCAL_CHUNK keys of a bucket)
constexpr int LOCAL_THREADS = 1024;
constexpr int LOCAL_WAVES = LOCAL_THREADS / WAVE;
constexpr int LOCAL_STEPS = 4;                                    // 64-key steps per wave
constexpr uint32_t LOCAL_CHUNK = LOCAL_THREADS * LOCAL_STEPS;     // 4096 keys per workgroup

template <int BITS>
constexpr size_t scatter_lds_bytes(uint32_t keys_per_block)
{
    return (size_t)(SCAT_WAVES * (1 << BITS) + 2 * (1 << BITS) + 3 * ((1 << BITS) / WAVE) + 2 * keys_per_block) * sizeof(uint32_t);
}

// PAY: one more word per key travels with it (the splat's packed bin rectangle, from rect order to depth order in the two
// LSD passes: the binning then reads it in rank order instead of gathering 4 bytes per rank through depthIndex -- 5 M
// random reads, 60 of k_bin_count's 79 us on C4).  It takes the index image's place in LDS once that has been streamed out.
template <int BITS, int SHIFT, bool FIRST, bool PAY = false>
__global__ __launch_bounds__(SCAT_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ idx_in, uint32_t n_in,
                                                          uint32_t* __restrict__ count,
                                                          uint32_t keys_per_block, const uint32_t* __restrict__ base,
                                                          const uint32_t* __restrict__ total,
                                                          uint32_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                                                          uint32_t* __restrict__ max_bucket,
                                                          const uint32_t* __restrict__ pay_in = nullptr, uint32_t* __restrict__ pay_out = nullptr,
                                                          uint4* __restrict__ chunk_tab = nullptr)
{
    constexpr int BINS = 1 << BITS;
    static_assert(BINS <= SCAT_THREADS, "one digit per thread");
    // the first pass runs on all n_in keys (their indices are their positions) and publishes the number as *count -- or, in
    // band mode (idx_in set), on the *count survivors k_quantise_hist left dense, with their original indices in idx_in;
    // the second pass runs on *count
    const bool band = FIRST && idx_in != nullptr;   // (uniform)
    const uint32_t n = (FIRST && !band) ? n_in : *count;
    extern __shared__ uint32_t s_scat[];
    uint32_t (*cnt)[BINS] = reinterpret_cast<uint32_t (*)[BINS]>(s_scat);   // [wave][digit]: counts, then local positions
    uint32_t* gdelta = s_scat + SCAT_WAVES * BINS;   // [digit]: global destination of local position p is gdelta[digit] + p
    uint32_t* lstart = gdelta + BINS;                // [digit]: first local position of the digit (phase 2 scratch)
    uint32_t* wsum = lstart + BINS;                  // [3][BINS / WAVE]: per-wave sums of the two scans, per-wave maxima of the totals
    uint32_t* lkey = wsum + 3 * (BINS / WAVE);       // [keys_per_block]
    uint32_t* lidx = lkey + keys_per_block;          // [keys_per_block]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blk = xcd_group_remap(blockIdx.x, gridDim.x);   // neighbouring key blocks on one XCD (gsr_internal.h)
    // (band mode: the grid is the scene's, the keys are the band's -- a workgroup past them has nothing to place; workgroup 0
    //  stays for the frame-wide words it publishes)
    if ((unsigned long long)blk * keys_per_block >= n && blockIdx.x != 0) return;
    KSTAMP(0);
    // the two table reads of this workgroup do not depend on its keys: issued first
    uint32_t tot = 0, wg_base = 0;
    if (threadIdx.x < BINS) {
        tot = total[threadIdx.x];
        wg_base = base[(size_t)blk * BINS + threadIdx.x];
    }
    for (int d = threadIdx.x; d < SCAT_WAVES * BINS; d += SCAT_THREADS) (&cnt[0][0])[d] = 0;
    __syncthreads();
    KSTAMP(1);

    const uint32_t per_wave = keys_per_block / SCAT_WAVES;  // multiple of 64
    const uint32_t steps = per_wave / WAVE;                 // <= SCAT_MAX_STEPS
    const uint32_t wbegin = blk * keys_per_block + wave * per_wave;
    const uint32_t wend = min(wbegin + per_wave, n);

    // phase 1: load this wave's keys (and, in the last pass, their indices) into registers; count digits per wave
    uint32_t key[SCAT_MAX_STEPS], src[SCAT_MAX_STEPS];
    uint32_t pay[PAY ? SCAT_MAX_STEPS : 1], pos[PAY ? SCAT_MAX_STEPS : 1];
#pragma unroll
    for (int k = 0; k < SCAT_MAX_STEPS; k++) {
        const uint32_t i = wbegin + k * WAVE + lane;
        const bool in = (uint32_t)k < steps && i < wend;
        key[k] = in ? keys_in[i] : 0xffffffffu;
        src[k] = (FIRST && !band) ? i : (in ? idx_in[i] : 0u);
        if (band) GSR_BOUND(sort, 4, src[k], n_in);
        if (PAY) pay[k] = in ? pay_in[band ? src[k] : i] : 0u;   // (band mode: the rectangle stayed at the splat's own index)
    }
#pragma unroll
    for (int k = 0; k < SCAT_MAX_STEPS; k++)
        if (key[k] != 0xffffffffu) atomicAdd(&cnt[wave][(key[k] >> SHIFT) & (BINS - 1)], 1u);
    __syncthreads();
    KSTAMP(2);
    // phase 2, one digit per thread: (a) cnt[w][d] -> keys of digit d in earlier waves, H = the workgroup's count;
    // (b) two exclusive scans over the digits at once: the column totals (-> first destination of the digit over all
    // workgroups) and H (-> first local position of the digit)
    uint32_t H = 0, incl_t = 0, incl_h = 0;
    if (threadIdx.x < BINS) {
        const int d = threadIdx.x;
#pragma unroll
        for (int w = 0; w < SCAT_WAVES; w++) {
            const uint32_t c = cnt[w][d];
            cnt[w][d] = H;
            H += c;
        }
        incl_t = tot; incl_h = H;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(incl_t, off), v = __shfl_up(incl_h, off);
            if (lane >= off) { incl_t += u; incl_h += v; }
        }
        if (lane == WAVE - 1) { wsum[wave] = incl_t; wsum[BINS / WAVE + wave] = incl_h; }
        if (max_bucket && blockIdx.x == 0) {   // the largest high-digit bucket of the frame, for the host's choice of sort order
            uint32_t mx = tot;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
            if (lane == 0) wsum[2 * (BINS / WAVE) + wave] = mx;
        }
    }
    __syncthreads();
    if (max_bucket && blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t mx = 0;
        for (int w = 0; w < BINS / WAVE; w++) mx = max(mx, wsum[2 * (BINS / WAVE) + w]);
        *max_bucket = mx;   // plain store: the word is host-mapped (no atomics on it)
    }
    KSTAMP(3);
    if (threadIdx.x < BINS) {
        const int d = threadIdx.x;
        uint32_t run_t = incl_t - tot, run_h = incl_h - H;
        for (int w = 0; w < wave; w++) { run_t += wsum[w]; run_h += wsum[BINS / WAVE + w]; }
        if (FIRST && blockIdx.x == 0 && d == BINS - 1) *count = run_t + tot;
        lstart[d] = run_h;
        gdelta[d] = run_t + wg_base - run_h;   // may wrap: only gdelta[d] + p is used
    }
    if (chunk_tab && blockIdx.x == 0) {   // (uniform) bucket order: the work list of k_local_sort -- per LOCAL_CHUNK keys of a bucket
        // (bucket, chunk inside it, the bucket's first key, its size); entry [0] holds the number of chunks.  Every workgroup of
        // that kernel used to find its chunk itself: a load of the 512 totals, two scans and two barriers in front of its work.
        __shared__ uint32_t s_cw[BINS / WAVE];
        const uint32_t nch = threadIdx.x < BINS ? (tot + LOCAL_CHUNK - 1) / LOCAL_CHUNK : 0u;
        uint32_t inc = nch;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (threadIdx.x < BINS && lane == WAVE - 1) s_cw[wave] = inc;
        __syncthreads();
        if (threadIdx.x < BINS) {
            uint32_t first = inc - nch, run_t = incl_t - tot, all = 0;
            for (int w = 0; w < BINS / WAVE; w++) {
                if (w < wave) { first += s_cw[w]; run_t += wsum[w]; }
                all += s_cw[w];
            }
            for (uint32_t ch = 0; ch < nch; ch++) chunk_tab[1 + first + ch] = make_uint4((uint32_t)threadIdx.x, ch, run_t, tot);
            if (threadIdx.x == 0) chunk_tab[0] = make_uint4(all, 0u, 0u, 0u);
        }
    }
    __syncthreads();
    KSTAMP(4);
    // phase 3: local position = digit start + keys of the digit in earlier waves + rank among this wave's earlier keys
    uint32_t* wc = cnt[wave];   // private to this wave from here on
#pragma unroll
    for (int k = 0; k < SCAT_MAX_STEPS; k++) {
        if ((uint32_t)k >= steps) break;  // wave-uniform
        const bool valid = key[k] != 0xffffffffu;
        const uint32_t digit = (key[k] >> SHIFT) & (BINS - 1);
        const uint64_t m = match_digit<BITS>(digit, valid);
        const uint32_t rank = lanes_below(m);
        // All lanes read their digit's running count, THEN the lowest lane of every group of equal digits advances
        // it.  One wave, its own LDS words: LDS executes in order, and the wavefront-scope fences keep the compiler
        // from moving the read below the write.
        const uint32_t before = wc[valid ? digit : 0];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wc[digit] = before + (uint32_t)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t p = lstart[digit] + before + rank;
            GSR_BOUND(sort, 1, p, keys_per_block);
            lkey[p] = key[k];
            lidx[p] = src[k];
            if (PAY) pos[k] = p;
        }
    }
    __syncthreads();
    // phase 4: the LDS image, in order, to its runs in global memory
    const uint32_t nlocal = lstart[BINS - 1] + cnt[SCAT_WAVES - 1][BINS - 1];   // keys this workgroup holds (absent ones excluded)
    for (uint32_t p = threadIdx.x; p < nlocal; p += SCAT_THREADS) {
        const uint32_t kv = lkey[p];
        const uint32_t dst = gdelta[(kv >> SHIFT) & (BINS - 1)] + p;
        GSR_BOUND(sort, 0, dst, n);
        GSR_BOUND(sort, 3, kv, DEPTH_RANGE + 1u);
        if (keys_out) keys_out[dst] = kv;
        idx_out[dst] = lidx[p];
    }
    if (PAY) {
        __syncthreads();   // the index image has been streamed out
#pragma unroll
        for (int k = 0; k < SCAT_MAX_STEPS; k++) {
            if ((uint32_t)k >= steps) break;
            if (key[k] != 0xffffffffu) lidx[pos[k]] = pay[k];
        }
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < nlocal; p += SCAT_THREADS) pay_out[gdelta[(lkey[p] >> SHIFT) & (BINS - 1)] + p] = lidx[p];
    }
#ifdef GSR_KSTAMPS
    __syncthreads();
    KSTAMP(5);
#endif
}

// ---------------------------------------------------------------------------
// Bucket order (small scenes): the first pass partitions by the HIGH 9 bits (k_scatter<9, 8, true>, histogram taken in
// k_quantise_hist), which leaves 257 contiguous buckets in index order; k_local_sort then sorts every bucket stably by
// the low 8 bits.  Same permutation as the LSD order (ascending key, ties by index), but four launches instead of
// six: no second histogram, no second column scan.
//
// One workgroup per LOCAL_CHUNK keys of a bucket, and no communication between workgroups: a workgroup counts the
// digits of its whole bucket itself (the bucket's digit starts) and of the chunks in front of its own (its offset
// inside every digit) -- a redundant read of the bucket, which sits in L2 -- then ranks its chunk exactly like
// k_scatter does and places it.  The redundant counting is quadratic in the bucket size, so a bucket holding most of
// the scene (depth outliers stretch the key range) is slow this way: every frame reports its largest bucket to the
// host, which falls back to the LSD order while that exceeds LOCAL_BUCKET_LIMIT (gsr_api.cpp).
// ---------------------------------------------------------------------------
inline uint32_t local_sort_grid(uint32_t n) { return (n + LOCAL_CHUNK - 1) / LOCAL_CHUNK + RADIX_HI_BINS / 2 + 1; }   // >= sum over buckets of ceil(size / chunk)
// (a 2-D grid -- chunk x bucket, bucket starts handed over by the partition pass, no search -- was measured slower:
//  3084 mostly empty 1024-thread workgroups cost more to dispatch than the search saves: 13.7 -> 21.0 us on C3)

__global__ __launch_bounds__(LOCAL_THREADS) void k_local_sort(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                                                              const uint4* __restrict__ chunk_tab,
                                                              uint32_t* __restrict__ depth_index,
                                                              const uint32_t* __restrict__ pay_in, uint32_t* __restrict__ pay_out)
{
    constexpr int BINS = RADIX_LO_BINS;
    __shared__ uint32_t cnt[LOCAL_WAVES][BINS];   // per wave: counts of my chunk, then keys of the digit in earlier waves
    __shared__ uint32_t tot[BINS];                // digit counts of the bucket behind my chunk, then the digit's first position
    __shared__ uint32_t bef[BINS];                // digit counts of the chunks in front of mine
    __shared__ uint32_t wsum[BINS / WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // which (bucket, chunk) am I?  The partition pass's first workgroup has listed them (k_scatter, chunk_tab)
    if (blockIdx.x >= chunk_tab[0].x) return;                 // more workgroups than chunks
    const uint4 me = chunk_tab[1 + blockIdx.x];
    for (int d = threadIdx.x; d < LOCAL_WAVES * BINS; d += LOCAL_THREADS) (&cnt[0][0])[d] = 0;
    if (threadIdx.x < BINS) { tot[threadIdx.x] = 0; bef[threadIdx.x] = 0; }
    __syncthreads();
    const uint32_t chunk = me.y, s0 = me.z, sz = me.w;
    const uint32_t cbeg = chunk * LOCAL_CHUNK, cend = min(cbeg + LOCAL_CHUNK, sz);

    // my keys (registers), counted per wave
    const uint32_t wbegin = cbeg + wave * (LOCAL_STEPS * WAVE), wend = min(wbegin + LOCAL_STEPS * WAVE, cend);
    uint32_t key[LOCAL_STEPS], src[LOCAL_STEPS], pay[LOCAL_STEPS];   // (pay: the packed bin rectangles travel along, k_scatter PAY)
#pragma unroll
    for (int k = 0; k < LOCAL_STEPS; k++) {
        const uint32_t i = wbegin + k * WAVE + lane;
        const bool in = i < wend;
        key[k] = in ? (keys[s0 + i] & (BINS - 1)) : 0xffffffffu;
        src[k] = in ? idx[s0 + i] : 0u;
        pay[k] = (in && pay_out) ? pay_in[s0 + i] : 0u;
    }
#pragma unroll
    for (int k = 0; k < LOCAL_STEPS; k++)
        if (key[k] != 0xffffffffu) atomicAdd(&cnt[wave][key[k]], 1u);
    // the rest of the bucket: digits of the chunks in front of mine (-> bef) and behind it (-> tot); loads in batches
    // of eight so that their round trips overlap (a load -> LDS-add chain per key made this the slowest part)
    constexpr int CB = 8;
    for (uint32_t i0 = threadIdx.x; i0 < sz; i0 += CB * LOCAL_THREADS) {
        uint32_t v[CB];
#pragma unroll
        for (int j = 0; j < CB; j++) {
            const uint32_t i = i0 + j * LOCAL_THREADS;
            v[j] = (i < sz && (i < cbeg || i >= cend)) ? keys[s0 + i] : 0xffffffffu;
        }
#pragma unroll
        for (int j = 0; j < CB; j++) {
            const uint32_t i = i0 + j * LOCAL_THREADS;
            if (v[j] != 0xffffffffu) atomicAdd(i < cbeg ? &bef[v[j] & (BINS - 1)] : &tot[v[j] & (BINS - 1)], 1u);
        }
    }
    __syncthreads();
    // one digit per thread: prefix over my waves; bucket-wide digit starts
    uint32_t H = 0, incl = 0, td = 0;
    if (threadIdx.x < BINS) {
        const int d = threadIdx.x;
#pragma unroll
        for (int w = 0; w < LOCAL_WAVES; w++) {
            const uint32_t cv = cnt[w][d];
            cnt[w][d] = H;
            H += cv;
        }
        td = tot[d] + bef[d] + H;     // the digit's keys in the whole bucket
        incl = td;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == WAVE - 1) wsum[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x < BINS) {
        uint32_t run = incl - td;
        for (int w = 0; w < wave; w++) run += wsum[w];
        tot[threadIdx.x] = run + bef[threadIdx.x];   // first position, inside the bucket, of my chunk's keys of this digit
    }
    __syncthreads();
    // rank and place
    uint32_t* wc = cnt[wave];
#pragma unroll
    for (int k = 0; k < LOCAL_STEPS; k++) {
        const bool valid = key[k] != 0xffffffffu;
        const uint32_t digit = valid ? key[k] : 0u;
        const uint64_t m = match_digit<RADIX_LO_BITS>(digit, valid);
        const uint32_t rank = lanes_below(m);
        const uint32_t earlier = wc[digit];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wc[digit] = earlier + (uint32_t)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            GSR_BOUND(sort, 2, tot[digit] + earlier + rank, sz);
            depth_index[s0 + tot[digit] + earlier + rank] = src[k];
            if (pay_out) pay_out[s0 + tot[digit] + earlier + rank] = pay[k];
        }
    }
}

void launch_sort(const SortBuffers& b, uint32_t n, hipStream_t s)
{
    if (!n) return;
    const dim3 grid(b.nblocks), block(SORT_THREADS);
    uint32_t* total_lo = b.digit_total;
    uint32_t* total_hi = b.digit_total + RADIX_LO_BINS;
    const size_t lds_lo = scatter_lds_bytes<RADIX_LO_BITS>(b.keys_per_block), lds_hi = scatter_lds_bytes<RADIX_HI_BITS>(b.keys_per_block);
    {   // the last pass needs more than the default 48 KiB of dynamic LDS even at 2048 keys: raise the limit once per
        // device (the attribute belongs to the device's copy of the kernel), to what the largest block size needs
        // (contexts on different host threads may arrive here together: one flag per device, set exactly once)
        static std::once_flag once[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(once[dev >= 0 && dev < 64 ? dev : 0], [] {
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_LO_BITS, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_LO_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_HI_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_HI_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_HI_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_LO_BITS, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_LO_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_HI_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipGetLastError();   // a failure shows up as the launch error
        });
    }
    // band mode: the survivors' offsets from the projection's per-block counts, then their depths and original indices dense
    // in keys_tmp[] / depth_index[] (both free: the first radix pass writes the one, the last kernel of the sort the other)
    uint32_t* band_idx = b.koff ? b.depth_index : nullptr;
    const uint32_t* live = b.koff ? b.count : nullptr;
    const int32_t* depth_in = b.koff ? reinterpret_cast<const int32_t*>(b.keys_tmp) : b.depth;
    if (b.koff) {
        const uint32_t nb = (n + PROJ_THREADS - 1u) / PROJ_THREADS;
        hipLaunchKernelGGL(k_kept_scan, dim3(1), dim3(KS_THREADS), (size_t)((nb + WAVE - 1u) / WAVE) * sizeof(uint32_t), s, b.kept, nb, b.koff, b.count);
        hipLaunchKernelGGL(k_band_gather, dim3((nb + BG_BLOCKS - 1) / BG_BLOCKS), dim3(BG_BLOCKS * WAVE), 0, s, b.depth, b.kept_lane, b.kept,
                           (const uint32_t*)b.koff, n, nb, reinterpret_cast<int32_t*>(b.keys_tmp), band_idx);
    }
    if (b.bucket_order) {
        // bucket order: partition by the high 9 bits, then one workgroup per bucket sorts by the low 8 (k_local_sort)
        hipLaunchKernelGGL(k_quantise_hist, grid, block, 0, s, depth_in, b.slots, b.minmax, n, b.keys_per_block,
                           live, b.keys, b.block_hist, RADIX_LO_BITS, RADIX_HI_BINS);
        launch_column_scan(b.block_hist, total_hi, RADIX_HI_BINS, b.nblocks, s, live, b.keys_per_block);
        uint4* tab = reinterpret_cast<uint4*>(b.chunk_tab);   // k_local_sort's work list, written by the partition pass's first workgroup
        if (b.rects_out)
            hipLaunchKernelGGL((k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, true, true>), grid, dim3(SCAT_THREADS), lds_hi, s, (const uint32_t*)b.keys,
                               (const uint32_t*)band_idx, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                               (const uint32_t*)total_hi, b.keys_tmp, b.idx_tmp, b.max_bucket, b.rect, b.rect_tmp, tab);
        else
            hipLaunchKernelGGL((k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, true>), grid, dim3(SCAT_THREADS), lds_hi, s, (const uint32_t*)b.keys,
                               (const uint32_t*)band_idx, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                               (const uint32_t*)total_hi, b.keys_tmp, b.idx_tmp, b.max_bucket, (const uint32_t*)nullptr, (uint32_t*)nullptr, tab);
        hipLaunchKernelGGL(k_local_sort, dim3(local_sort_grid(n)), dim3(LOCAL_THREADS), 0, s, (const uint32_t*)b.keys_tmp,
                           (const uint32_t*)b.idx_tmp, (const uint4*)tab, b.depth_index, (const uint32_t*)b.rect_tmp, b.rects_out);
        return;
    }
    hipLaunchKernelGGL(k_quantise_hist, grid, block, 0, s, depth_in, b.slots, b.minmax, n, b.keys_per_block,
                       live, b.keys, b.block_hist, 0, RADIX_LO_BINS);
    launch_column_scan(b.block_hist, total_lo, RADIX_LO_BINS, b.nblocks, s, live, b.keys_per_block);
    const bool carry = b.rects_out != nullptr;   // the packed rectangles travel with the keys (k_scatter, PAY)
    if (carry)
        hipLaunchKernelGGL((k_scatter<RADIX_LO_BITS, 0, true, true>), grid, dim3(SCAT_THREADS), lds_lo, s, (const uint32_t*)b.keys,
                           (const uint32_t*)band_idx, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                           (const uint32_t*)total_lo, b.keys_tmp, b.idx_tmp, (uint32_t*)nullptr, b.rect, b.rect_tmp);
    else
        hipLaunchKernelGGL((k_scatter<RADIX_LO_BITS, 0, true>), grid, dim3(SCAT_THREADS), lds_lo, s, (const uint32_t*)b.keys,
                           (const uint32_t*)band_idx, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                           (const uint32_t*)total_lo, b.keys_tmp, b.idx_tmp, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_hist_hi, grid, block, 0, s, (const uint32_t*)b.keys_tmp, (const uint32_t*)b.count, b.keys_per_block,
                       b.block_hist);
    launch_column_scan(b.block_hist, total_hi, RADIX_HI_BINS, b.nblocks, s, live, b.keys_per_block);
    if (carry)
        hipLaunchKernelGGL((k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, false, true>), grid, dim3(SCAT_THREADS), lds_hi, s, (const uint32_t*)b.keys_tmp,
                           (const uint32_t*)b.idx_tmp, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                           (const uint32_t*)total_hi, (uint32_t*)nullptr, b.depth_index, b.max_bucket, (const uint32_t*)b.rect_tmp, b.rects_out);
    else
        hipLaunchKernelGGL((k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, false>), grid, dim3(SCAT_THREADS), lds_hi, s, (const uint32_t*)b.keys_tmp,
                           (const uint32_t*)b.idx_tmp, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                           (const uint32_t*)total_hi, (uint32_t*)nullptr, b.depth_index, b.max_bucket, (const uint32_t*)nullptr, (uint32_t*)nullptr);
}

}  // namespace gsr
