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
// Band mode (multi-GPU, `bbox` != null): a splat whose projected box is empty for this context's band (culled, or
// outside the band) gets the key 0xffffffff = "absent".  The first scatter drops absent keys, so everything after
// it -- second radix pass, binning -- runs on the survivors only; their order is the restriction of the global
// order, so the band's pixels are unchanged.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_quantise_hist(const int32_t* __restrict__ depth,
                                                                const int32_t* __restrict__ minmax, uint32_t n,
                                                                uint32_t keys_per_block, const uint2* __restrict__ bbox,
                                                                uint32_t* __restrict__ keys,
                                                                uint32_t* __restrict__ block_hist)
{
    __shared__ uint32_t h_lo[RADIX_LO_BINS];
    for (int d = threadIdx.x; d < RADIX_LO_BINS; d += SORT_THREADS) h_lo[d] = 0;
    __syncthreads();

    const int32_t minDepth = minmax[0], maxDepth = minmax[1];
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
        if (bbox) {
            const uint32_t bx = bbox[i].x;
            if ((bx & 0xffffu) > (bx >> 16)) q = 0xffffffffu;  // empty box: absent from this band's frame
        }
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
// Stable scatter of one pass.  Workgroup = 16 waves over keys_per_block (2048) keys; wave w owns the w-th
// sixteenth (contiguous, 2 steps of 64 keys), so the order of equal digits is: workgroup, then wave, then
// step, then lane = input order.  Phase 1 counts per wave, phase 2 turns the counts into running destinations,
// phase 3 ranks with ballot matching and scatters.  Sixteen waves (not four) because the pass is a chain of
// dependent round trips: more waves in flight hide them.
// ---------------------------------------------------------------------------
constexpr int SCAT_THREADS = 1024;
constexpr int SCAT_WAVES = SCAT_THREADS / WAVE;
constexpr int SCAT_MAX_STEPS = 4;  // keys_per_block <= SCAT_THREADS * SCAT_MAX_STEPS

template <int BITS, int SHIFT, bool FIRST>
__global__ __launch_bounds__(SCAT_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ idx_in, uint32_t n_in,
                                                          uint32_t* __restrict__ count,
                                                          uint32_t keys_per_block, const uint32_t* __restrict__ base,
                                                          const uint32_t* __restrict__ total,
                                                          uint32_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out)
{
    constexpr int BINS = 1 << BITS;
    // the first pass reads all n_in keys (absent ones are skipped) and publishes how many it kept;
    // the second pass runs on that many
    const uint32_t n = FIRST ? n_in : *count;
    __shared__ uint32_t cnt[SCAT_WAVES][BINS];
    __shared__ uint32_t dstart[BINS];          // keys with a smaller digit, all workgroups
    __shared__ uint32_t wsum[BINS / WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < SCAT_WAVES * BINS; d += SCAT_THREADS) (&cnt[0][0])[d] = 0;
    __syncthreads();

    const uint32_t per_wave = keys_per_block / SCAT_WAVES;  // multiple of 64
    const uint32_t steps = per_wave / WAVE;                 // <= SCAT_MAX_STEPS
    const uint32_t wbegin = blockIdx.x * keys_per_block + wave * per_wave;
    const uint32_t wend = min(wbegin + per_wave, n);

    // phase 1: load this wave's keys (kept in registers) and count digits per wave
    uint32_t key[SCAT_MAX_STEPS];
#pragma unroll
    for (int k = 0; k < SCAT_MAX_STEPS; k++) {
        const uint32_t i = wbegin + k * WAVE + lane;
        key[k] = ((uint32_t)k < steps && i < wend) ? keys_in[i] : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < SCAT_MAX_STEPS; k++)
        if (key[k] != 0xffffffffu) atomicAdd(&cnt[wave][(key[k] >> SHIFT) & (BINS - 1)], 1u);
    // digit starts: exclusive scan of total[0..BINS), one digit per thread on the first BINS threads
    uint32_t tot = 0, incl = 0;
    if (threadIdx.x < BINS) {
        tot = total[threadIdx.x];
        incl = tot;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == WAVE - 1) wsum[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x < BINS) {
        uint32_t run = incl - tot;
        for (int w = 0; w < wave; w++) run += wsum[w];
        dstart[threadIdx.x] = run;
        if (FIRST && blockIdx.x == 0 && threadIdx.x == BINS - 1) *count = run + tot;
    }
    __syncthreads();
    // phase 2: cnt[w][d] <- first destination of digit d for this workgroup + counts of earlier waves
    for (int d = threadIdx.x; d < BINS; d += SCAT_THREADS) {
        uint32_t run = dstart[d] + base[(size_t)blockIdx.x * BINS + d];
#pragma unroll
        for (int w = 0; w < SCAT_WAVES; w++) {
            const uint32_t c = cnt[w][d];
            cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
    // phase 3: rank + scatter (cnt[wave][*] is private to this wave from here on)
    uint32_t* wc = cnt[wave];
#pragma unroll
    for (int k = 0; k < SCAT_MAX_STEPS; k++) {
        if ((uint32_t)k >= steps) break;  // wave-uniform
        const uint32_t i = wbegin + k * WAVE + lane;
        const bool valid = key[k] != 0xffffffffu;
        const uint32_t digit = (key[k] >> SHIFT) & (BINS - 1);
        const uint64_t m = match_digit<BITS>(digit, valid);
        const uint32_t rank = lanes_below(m);
        // All lanes read their digit's running destination, THEN the lowest lane of every group of
        // equal digits advances it.  One wave, its own LDS words: LDS executes in order, and the
        // wavefront-scope fences keep the compiler from moving the read below the write.
        const uint32_t start = wc[valid ? digit : 0];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wc[digit] = start + (uint32_t)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t dst = start + rank;
            if (keys_out) keys_out[dst] = key[k];
            idx_out[dst] = FIRST ? i : idx_in[i];
        }
    }
}

void launch_sort(const SortBuffers& b, uint32_t n, hipStream_t s)
{
    if (!n) return;
    const dim3 grid(b.nblocks), block(SORT_THREADS);
    uint32_t* total_lo = b.digit_total;
    uint32_t* total_hi = b.digit_total + RADIX_LO_BINS;
    hipLaunchKernelGGL(k_quantise_hist, grid, block, 0, s, b.depth, b.minmax, n, b.keys_per_block, b.cull_bbox, b.keys,
                       b.block_hist);
    launch_column_scan(b.block_hist, total_lo, RADIX_LO_BINS, b.nblocks, s);
    hipLaunchKernelGGL((k_scatter<RADIX_LO_BITS, 0, true>), grid, dim3(SCAT_THREADS), 0, s, (const uint32_t*)b.keys,
                       (const uint32_t*)nullptr, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                       (const uint32_t*)total_lo, b.keys_tmp, b.idx_tmp);
    hipLaunchKernelGGL(k_hist_hi, grid, block, 0, s, (const uint32_t*)b.keys_tmp, (const uint32_t*)b.count, b.keys_per_block,
                       b.block_hist);
    launch_column_scan(b.block_hist, total_hi, RADIX_HI_BINS, b.nblocks, s);
    hipLaunchKernelGGL((k_scatter<RADIX_HI_BITS, RADIX_LO_BITS, false>), grid, dim3(SCAT_THREADS), 0, s, (const uint32_t*)b.keys_tmp,
                       (const uint32_t*)b.idx_tmp, n, b.count, b.keys_per_block, (const uint32_t*)b.block_hist,
                       (const uint32_t*)total_hi, (uint32_t*)nullptr, b.depth_index);
}

}  // namespace gsr
