// Order-preserving coarse binning: the depth-sorted splat stream is split into
// one list per BIN_PX x BIN_PX screen bin (a splat enters every bin its pixel
// bounding box touches), and every list keeps the global front-to-back order.
// No second sort of duplicated (tile, depth) pairs is needed: the split is a
// stable counting pass (count -> scan -> scatter).
//
// Stability in the scatter: a wave walks 64 consecutive ranks at a time.  For
// one step, the set of lanes whose box covers bin (X, Y) is
//     colmask[X] & rowmask[Y]
// where colmask/rowmask are 64-bit lane sets built in LDS with ds_or_b64, one
// word per bin column / bin row (boxes are rectangles, so coverage separates).
// A splat's slot inside the step is the population count of that set below its
// lane: wavefront ballot arithmetic, no atomics with ordering requirements.
#include "gsr_internal.h"

namespace gsr {

constexpr int BIN_THREADS = 256;
constexpr int BIN_WAVES = BIN_THREADS / WAVE;
constexpr int BIN_STEPS = 8;                                  // 64-rank steps per wave
constexpr uint32_t BIN_RANKS_PER_BLOCK = BIN_THREADS * BIN_STEPS;  // 2048

struct BinRect { int x0, x1, y0, y1; };  // inclusive bin coordinates relative to the band; x0 > x1: none

__device__ __forceinline__ BinRect bin_rect(uint2 bb, const BinGrid& g)
{
    BinRect r;
    const int px0 = bb.x & 0xffff, px1 = bb.x >> 16, py0 = bb.y & 0xffff, py1 = bb.y >> 16;
    if (px0 > px1) { r.x0 = 1; r.x1 = 0; r.y0 = 1; r.y1 = 0; return r; }
    r.x0 = max(px0 / BIN_PX, g.bx_lo) - g.bx_lo;
    r.x1 = min(px1 / BIN_PX, g.bx_hi - 1) - g.bx_lo;
    r.y0 = py0 / BIN_PX;
    r.y1 = py1 / BIN_PX;
    return r;
}

__device__ __forceinline__ uint32_t lanes_below64(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------------------
// count: table[block][bin] = entries block contributes to bin; bin_total[bin] += same
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BIN_THREADS) void k_bin_count(const uint32_t* __restrict__ depth_index,
                                                           const uint2* __restrict__ bbox, uint32_t n, BinGrid g,
                                                           uint32_t* __restrict__ table, uint32_t* __restrict__ bin_total,
                                                           uint64_t* __restrict__ visible, uint64_t* __restrict__ tile_entries,
                                                           uint64_t* __restrict__ accum)
{
    extern __shared__ uint32_t s_cnt[];  // nbins
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    for (int b = threadIdx.x; b < nbins; b += BIN_THREADS) s_cnt[b] = 0;
    __syncthreads();
    const uint32_t begin = blockIdx.x * BIN_RANKS_PER_BLOCK;
    uint32_t vis = 0, tiles = 0;
#pragma unroll
    for (int st = 0; st < BIN_STEPS; st++) {
        const uint32_t r = begin + st * BIN_THREADS + threadIdx.x;
        if (r < n) {
            const uint2 bb = bbox[depth_index[r]];
            const BinRect br = bin_rect(bb, g);
            if (br.x0 <= br.x1) {
                vis++;
                const int tx0 = max((int)(bb.x & 0xffff) / TILE, g.bx_lo * BIN_TILES);
                const int tx1 = min((int)(bb.x >> 16) / TILE, g.bx_hi * BIN_TILES - 1);
                tiles += (uint32_t)((tx1 - tx0 + 1) * ((int)(bb.y >> 16) / TILE - (int)(bb.y & 0xffff) / TILE + 1));
            }
            for (int y = br.y0; y <= br.y1; y++)
                for (int x = br.x0; x <= br.x1; x++) atomicAdd(&s_cnt[y * nbxb + x], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += BIN_THREADS) {
        const uint32_t c = s_cnt[b];
        table[(size_t)blockIdx.x * nbins + b] = c;
        if (c) atomicAdd(&bin_total[b], c);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vis += __shfl_xor(vis, off);
        tiles += __shfl_xor(tiles, off);
    }
    if ((threadIdx.x & 63) == 0 && vis) {
        atomicAdd((unsigned long long*)visible, (unsigned long long)vis);
        atomicAdd((unsigned long long*)tile_entries, (unsigned long long)tiles);
        atomicAdd((unsigned long long*)&accum[0], (unsigned long long)vis);
        atomicAdd((unsigned long long*)&accum[2], (unsigned long long)tiles);
    }
}

// ---------------------------------------------------------------------------
// scan: one wave per bin.  bin_start[bin] = entries of all earlier bins;
// table[block][bin] <- bin_start[bin] + entries of earlier blocks in this bin.
// The wave handling the last bin also writes bin_start[nbins].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BIN_THREADS) void k_bin_scan(uint32_t* __restrict__ table,
                                                          const uint32_t* __restrict__ bin_total, int nbins,
                                                          uint32_t nblocks, uint32_t* __restrict__ bin_start,
                                                          uint64_t* __restrict__ accum)
{
    const int lane = threadIdx.x & 63;
    const int bin = blockIdx.x * BIN_WAVES + (threadIdx.x >> 6);
    if (bin >= nbins) return;
    uint32_t acc = 0;
    for (int j = lane; j < bin; j += WAVE) acc += bin_total[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        bin_start[bin] = acc;
        if (bin == nbins - 1) {
            const uint32_t total = acc + bin_total[bin];
            bin_start[nbins] = total;
            atomicAdd((unsigned long long*)&accum[1], (unsigned long long)total);
            atomicAdd((unsigned long long*)&accum[3], 1ull);
        }
    }
    uint32_t run = acc;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += WAVE) {
        const uint32_t b = b0 + lane;
        const uint32_t v = (b < nblocks) ? table[(size_t)b * nbins + bin] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (b < nblocks) table[(size_t)b * nbins + bin] = run + incl - v;
        run += __shfl(incl, WAVE - 1);
    }
}

// ---------------------------------------------------------------------------
// scatter: list[...] = splat index, bins in raster order, depth order inside a bin.
// Workgroup = 4 waves over 2048 consecutive ranks; wave w owns steps
// [w*8, w*8+8) of 64 ranks, so input order = (workgroup, wave, step, lane).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BIN_THREADS) void k_bin_scatter(const uint32_t* __restrict__ depth_index,
                                                             const uint2* __restrict__ bbox, uint32_t n, BinGrid g,
                                                             const uint32_t* __restrict__ table,
                                                             uint32_t* __restrict__ list, uint32_t capacity,
                                                             uint32_t* __restrict__ overflow)
{
    extern __shared__ uint32_t s_mem[];
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* cnt = s_mem;                                   // [BIN_WAVES][nbins]
    // per-wave lane sets: [nbxb] column words then [nby] row words (8-byte aligned region)
    const int nmask = nbxb + g.nby;
    uint64_t* masks = reinterpret_cast<uint64_t*>(s_mem + ((BIN_WAVES * nbins + 1) & ~1));
    unsigned long long* colm = reinterpret_cast<unsigned long long*>(masks + (size_t)wave * nmask);
    unsigned long long* rowm = colm + nbxb;

    for (int b = threadIdx.x; b < BIN_WAVES * nbins; b += BIN_THREADS) cnt[b] = 0;
    for (int b = threadIdx.x; b < BIN_WAVES * nmask; b += BIN_THREADS) masks[b] = 0;
    __syncthreads();

    // this wave's 8 steps of 64 consecutive ranks: indices and bin rectangles stay in registers
    const uint32_t wbegin = blockIdx.x * BIN_RANKS_PER_BLOCK + wave * (BIN_STEPS * WAVE);
    uint32_t idx[BIN_STEPS];
    BinRect br[BIN_STEPS];
#pragma unroll
    for (int st = 0; st < BIN_STEPS; st++) {
        const uint32_t r = wbegin + st * WAVE + lane;
        idx[st] = (r < n) ? depth_index[r] : 0u;
    }
#pragma unroll
    for (int st = 0; st < BIN_STEPS; st++) {
        const uint32_t r = wbegin + st * WAVE + lane;
        if (r < n) br[st] = bin_rect(bbox[idx[st]], g);
        else { br[st].x0 = 1; br[st].x1 = 0; br[st].y0 = 1; br[st].y1 = 0; }
    }
    // phase 1: per-wave counts
    uint32_t* mycnt = cnt + (size_t)wave * nbins;
#pragma unroll
    for (int st = 0; st < BIN_STEPS; st++)
        for (int y = br[st].y0; y <= br[st].y1; y++)
            for (int x = br[st].x0; x <= br[st].x1; x++) atomicAdd(&mycnt[y * nbxb + x], 1u);
    __syncthreads();
    // phase 2: running destinations: workgroup base + earlier waves
    for (int b = threadIdx.x; b < nbins; b += BIN_THREADS) {
        uint32_t run = table[(size_t)blockIdx.x * nbins + b];
#pragma unroll
        for (int w = 0; w < BIN_WAVES; w++) {
            const uint32_t c = cnt[(size_t)w * nbins + b];
            cnt[(size_t)w * nbins + b] = run;
            run += c;
        }
    }
    __syncthreads();
    // phase 3: per step, build the lane sets, rank, write, advance the running destinations
    volatile uint32_t* vcnt = mycnt;
    const unsigned long long mybit = 1ull << lane;
#pragma unroll
    for (int st = 0; st < BIN_STEPS; st++) {
        const BinRect b = br[st];
        const bool any = __ballot(b.x0 <= b.x1) != 0ull;
        if (!any) continue;  // wave-uniform
        for (int x = b.x0; x <= b.x1; x++) atomicOr(&colm[x], mybit);
        for (int y = b.y0; y <= b.y1 && b.x0 <= b.x1; y++) atomicOr(&rowm[y], mybit);
        __builtin_amdgcn_wave_barrier();
        // sweep 1: read running destination + rank for every covered bin, write the entries
        for (int y = b.y0; y <= b.y1; y++) {
            const uint64_t rm = ((volatile unsigned long long*)rowm)[y];
            for (int x = b.x0; x <= b.x1; x++) {
                const uint64_t m = rm & ((volatile unsigned long long*)colm)[x];
                const uint32_t dst = vcnt[y * nbxb + x] + lanes_below64(m);
                if (dst < capacity) list[dst] = idx[st];
                else atomicOr(overflow, 1u);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // sweep 2: the lowest lane of every bin's set advances that bin's running destination
        for (int y = b.y0; y <= b.y1; y++) {
            const uint64_t rm = ((volatile unsigned long long*)rowm)[y];
            for (int x = b.x0; x <= b.x1; x++) {
                const uint64_t m = rm & ((volatile unsigned long long*)colm)[x];
                if ((m & (mybit - 1)) == 0ull) vcnt[y * nbxb + x] += (uint32_t)__popcll(m);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // clear the words this lane set
        for (int x = b.x0; x <= b.x1; x++) ((volatile unsigned long long*)colm)[x] = 0ull;
        for (int y = b.y0; y <= b.y1 && b.x0 <= b.x1; y++) ((volatile unsigned long long*)rowm)[y] = 0ull;
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_bin(const BinBuffers& b, const BinGrid& g, uint32_t n, hipStream_t s)
{
    if (!n) return;
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    const dim3 grid(b.nblocks), block(BIN_THREADS);
    hipLaunchKernelGGL(k_bin_count, grid, block, nbins * sizeof(uint32_t), s, b.depth_index, b.bbox, n, g, b.table,
                       b.bin_total, b.visible, b.tile_entries, b.accum);
    hipLaunchKernelGGL(k_bin_scan, dim3((nbins + BIN_WAVES - 1) / BIN_WAVES), block, 0, s, b.table,
                       (const uint32_t*)b.bin_total, nbins, b.nblocks, b.bin_start, b.accum);
    const size_t lds = (size_t)((BIN_WAVES * nbins + 1) & ~1) * 4 + (size_t)BIN_WAVES * (nbxb + g.nby) * 8;
    static bool lds_raised = false;  // allow up to the CU's full 160 KiB of dynamic LDS (4K: 8160 bins)
    if (!lds_raised) {
        (void)hipFuncSetAttribute((const void*)k_bin_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_bin_count, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        lds_raised = true;
    }
    hipLaunchKernelGGL(k_bin_scatter, grid, block, lds, s, b.depth_index, b.bbox, n, g, (const uint32_t*)b.table, b.list,
                       b.capacity, b.overflow);
}

}  // namespace gsr
