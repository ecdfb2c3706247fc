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

namespace gsr {

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
// Band mode (multi-GPU, `cull`): a splat whose bin rectangle is empty for this context's band (culled, or
// outside the band) gets the key 0xffffffff = "absent".  The first scatter drops absent keys, so everything after
// it -- second radix pass, binning -- runs on the survivors only; their order is the restriction of the global
// order, so the band's pixels are unchanged.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_quantise_hist(const int32_t* __restrict__ depth,
                                                                const int32_t* __restrict__ slots,
                                                                int32_t* __restrict__ minmax_out, uint32_t n,
                                                                uint32_t keys_per_block, const uint32_t* __restrict__ rect, int cull,
                                                                uint32_t* __restrict__ keys,
                                                                uint32_t* __restrict__ block_hist)
{
    __shared__ uint32_t h_lo[RADIX_LO_BINS];
    __shared__ int32_t s_mm[2][SORT_THREADS / WAVE];
    for (int d = threadIdx.x; d < RADIX_LO_BINS; d += SORT_THREADS) h_lo[d] = 0;
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

    const uint32_t begin = blockIdx.x * keys_per_block;
    const uint32_t end = min(begin + keys_per_block, n);
    for (uint32_t i = begin + threadIdx.x; i < end; i += SORT_THREADS) {
        // wasm.cpp:38: u32 wrap-around subtract, u32->f32 RNE, f32 multiply, truncate
        const uint32_t rel = (uint32_t)depth[i] - (uint32_t)minDepth;
        uint32_t q = degenerate ? 0u : (uint32_t)((float)rel * depthInv);
        q = min(q, DEPTH_RANGE);
        if (cull && rect[i] == RECT_NONE) q = 0xffffffffu;  // nothing to draw in this band: absent from its frame
        keys[i] = q;
        if (q != 0xffffffffu) atomicAdd(&h_lo[q & (RADIX_LO_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX_LO_BINS; d += SORT_THREADS) block_hist[(size_t)blockIdx.x * RADIX_LO_BINS + d] = h_lo[d];
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

__global__ __launch_bounds__(CS_THREADS) void k_column_scan(uint32_t* __restrict__ table, uint32_t* __restrict__ total,
                                                            int ncols, uint32_t nrows)
{
    __shared__ uint32_t s_sum[CS_SLOTS][CS_COLS];
    const int c = threadIdx.x & (CS_COLS - 1);
    const int slot = threadIdx.x / CS_COLS;
    const int col = blockIdx.x * CS_COLS + c;
    const bool ok = col < ncols;
    const uint32_t per = (nrows + CS_SLOTS - 1) / CS_SLOTS;
    const uint32_t r0 = min((uint32_t)slot * per, nrows), r1 = min(r0 + per, nrows);
    uint32_t* p = table + (size_t)r0 * ncols + (ok ? col : 0);

    uint32_t sum = 0;
    if (ok) {
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
        if (slot == 0) total[col] = tot;
    }
}

void launch_column_scan(uint32_t* table, uint32_t* total, int ncols, uint32_t nrows, hipStream_t s)
{
    hipLaunchKernelGGL(k_column_scan, dim3((ncols + CS_COLS - 1) / CS_COLS), dim3(CS_THREADS), 0, s, table, total, ncols, nrows);
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

template <int BITS>
constexpr size_t scatter_lds_bytes(uint32_t keys_per_block)
{
    return (size_t)(SCAT_WAVES * (1 << BITS) + 2 * (1 << BITS) + 2 * ((1 << BITS) / WAVE) + 2 * keys_per_block) * sizeof(uint32_t);
}

template <int BITS, int SHIFT, bool FIRST>
__global__ __launch_bounds__(SCAT_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ idx_in, uint32_t n_in,
                                                          uint32_t* __restrict__ count,
                                                          uint32_t keys_per_block, const uint32_t* __restrict__ base,
                                                          const uint32_t* __restrict__ total,
                                                          uint32_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out)
{
    constexpr int BINS = 1 << BITS;
    static_assert(BINS <= SCAT_THREADS, "one digit per thread");
    // the first pass reads all n_in keys (absent ones are skipped) and publishes how many it kept;
    // the second pass runs on that many
    const uint32_t n = FIRST ? n_in : *count;
    extern __shared__ uint32_t s_scat[];
    uint32_t (*cnt)[BINS] = reinterpret_cast<uint32_t (*)[BINS]>(s_scat);   // [wave][digit]: counts, then local positions
    uint32_t* gdelta = s_scat + SCAT_WAVES * BINS;   // [digit]: global destination of local position p is gdelta[digit] + p
    uint32_t* lstart = gdelta + BINS;                // [digit]: first local position of the digit (phase 2 scratch)
    uint32_t* wsum = lstart + BINS;                  // [2][BINS / WAVE]: per-wave sums of the two scans
    uint32_t* lkey = wsum + 2 * (BINS / WAVE);       // [keys_per_block]
    uint32_t* lidx = lkey + keys_per_block;          // [keys_per_block]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    KSTAMP(0);
    // the two table reads of this workgroup do not depend on its keys: issued first
    uint32_t tot = 0, wg_base = 0;
    if (threadIdx.x < BINS) {
        tot = total[threadIdx.x];
        wg_base = base[(size_t)blockIdx.x * BINS + threadIdx.x];
    }
    for (int d = threadIdx.x; d < SCAT_WAVES * BINS; d += SCAT_THREADS) (&cnt[0][0])[d] = 0;
    __syncthreads();
    KSTAMP(1);

    const uint32_t per_wave = keys_per_block / SCAT_WAVES;  // multiple of 64
    const uint32_t steps = per_wave / WAVE;                 // <= SCAT_MAX_STEPS
    const uint32_t wbegin = blockIdx.x * keys_per_block + wave * per_wave;
    const uint32_t wend = min(wbegin + per_wave, n);

    // phase 1: load this wave's keys (and, in the last pass, their indices) into registers; count digits per wave
    uint32_t key[SCAT_MAX_STEPS], src[SCAT_MAX_STEPS];
#pragma unroll
    for (int k = 0; k < SCAT_MAX_STEPS; k++) {
        const uint32_t i = wbegin + k * WAVE + lane;
        const bool in = (uint32_t)k < steps && i < wend;
        key[k] = in ? keys_in[i] : 0xffffffffu;
        src[k] = FIRST ? i : (in ? idx_in[i] : 0u);
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
    }
    __syncthreads();
    KSTAMP(3);
    if (threadIdx.x < BINS) {
        const int d = threadIdx.x;
        uint32_t run_t = incl_t - tot, run_h = incl_h - H;
        for (int w = 0; w < wave; w++) { run_t += wsum[w]; run_h += wsum[BINS / WAVE + w]; }
        if (FIRST && blockIdx.x == 0 && d == BINS - 1) *count = run_t + tot;
        lstart[d] = run_h;
        gdelta[d] = run_t + wg_base - run_h;   // may wrap: only gdelta[d] + p is used
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
            lkey[p] = key[k];
            lidx[p] = src[k];
        }
    }
    __syncthreads();
    // phase 4: the LDS image, in order, to its runs in global memory
    const uint32_t nlocal = lstart[BINS - 1] + cnt[SCAT_WAVES - 1][BINS - 1];   // keys this workgroup holds (absent ones excluded)
    for (uint32_t p = threadIdx.x; p < nlocal; p += SCAT_THREADS) {
        const uint32_t kv = lkey[p];
        const uint32_t dst = gdelta[(kv >> SHIFT) & (BINS - 1)] + p;
        if (keys_out) keys_out[dst] = kv;
        idx_out[dst] = lidx[p];
    }
#ifdef GSR_KSTAMPS
    __syncthreads();
    KSTAMP(5);
#endif
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
        static unsigned done_mask = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 32 && !(done_mask & (1u << dev))) {
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_LO_BITS, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_LO_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipFuncSetAttribute((const void*)k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)scatter_lds_bytes<RADIX_HI_BITS>(SCAT_THREADS * SCAT_MAX_STEPS));
            (void)hipGetLastError();   // a failure shows up as the launch error
            done_mask |= 1u << dev;
        }
    }
    hipLaunchKernelGGL(k_quantise_hist, grid, block, 0, s, b.depth, b.slots, b.minmax, n, b.keys_per_block,
                       b.rect, b.cull, b.keys, b.block_hist);
    launch_column_scan(b.block_hist, total_lo, RADIX_LO_BINS, b.nblocks, s);
    hipLaunchKernelGGL((k_scatter<RADIX_LO_BITS, 0, true>), grid, dim3(SCAT_THREADS), lds_lo, s, (const uint32_t*)b.keys,
                       (const uint32_t*)nullptr, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                       (const uint32_t*)total_lo, b.keys_tmp, b.idx_tmp);
    hipLaunchKernelGGL(k_hist_hi, grid, block, 0, s, (const uint32_t*)b.keys_tmp, (const uint32_t*)b.count, b.keys_per_block,
                       b.block_hist);
    launch_column_scan(b.block_hist, total_hi, RADIX_HI_BINS, b.nblocks, s);
    hipLaunchKernelGGL((k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, false>), grid, dim3(SCAT_THREADS), lds_hi, s, (const uint32_t*)b.keys_tmp,
                       (const uint32_t*)b.idx_tmp, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                       (const uint32_t*)total_hi, (uint32_t*)nullptr, b.depth_index);
}

}  // namespace gsr
