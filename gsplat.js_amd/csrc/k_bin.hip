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

// bin rectangle of one rank in 4 bytes (x0 | x1<<8 | y0<<16 | y1<<24; bins per axis <= 256): the count pass
// gathers the boxes once and leaves the rectangles in rank order, the scatter pass reads them coalesced
__device__ __forceinline__ uint32_t pack_rect(const BinRect& r)
{
    return (r.x0 <= r.x1) ? ((uint32_t)r.x0 | ((uint32_t)r.x1 << 8) | ((uint32_t)r.y0 << 16) | ((uint32_t)r.y1 << 24)) : 1u;
}
__device__ __forceinline__ BinRect unpack_rect(uint32_t p)
{
    BinRect r;
    r.x0 = (int)(p & 0xffu); r.x1 = (int)((p >> 8) & 0xffu); r.y0 = (int)((p >> 16) & 0xffu); r.y1 = (int)(p >> 24);
    if (r.x0 > r.x1) { r.y0 = 1; r.y1 = 0; }
    return r;
}

__device__ __forceinline__ uint32_t lanes_below64(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------------------
// count: table[block][bin] = entries this workgroup contributes to bin;
// blk_counts[block] = (visible splats, 16x16 tiles their boxes overlap).
// No global atomics: hundreds of workgroups hitting the same few words cost
// more than the whole pass.
// ---------------------------------------------------------------------------
constexpr int CNT_THREADS = 1024;
constexpr int CNT_STEPS = (int)BIN_RANKS_PER_BLOCK / CNT_THREADS;  // 2

__global__ __launch_bounds__(CNT_THREADS) void k_bin_count(const uint32_t* __restrict__ depth_index,
                                                           const uint2* __restrict__ bbox, const uint32_t* __restrict__ count,
                                                           BinGrid g, uint32_t* __restrict__ table,
                                                           uint2* __restrict__ blk_counts, uint32_t* __restrict__ rects)
{
    const uint32_t n = *count;  // ranks the sort produced (all splats, or the band's survivors)
    extern __shared__ uint32_t s_cnt[];  // nbins
    __shared__ uint32_t s_red[2 * (CNT_THREADS / WAVE)];
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    for (int b = threadIdx.x; b < nbins; b += CNT_THREADS) s_cnt[b] = 0;
    __syncthreads();
    const uint32_t begin = blockIdx.x * BIN_RANKS_PER_BLOCK;
    uint32_t vis = 0, tiles = 0;
    uint32_t idx[CNT_STEPS];
#pragma unroll
    for (int st = 0; st < CNT_STEPS; st++) {
        const uint32_t r = begin + st * CNT_THREADS + threadIdx.x;
        idx[st] = (r < n) ? depth_index[r] : 0xffffffffu;
    }
    uint2 bbs[CNT_STEPS];
#pragma unroll
    for (int st = 0; st < CNT_STEPS; st++)
        bbs[st] = (idx[st] != 0xffffffffu) ? bbox[idx[st]] : make_uint2(BBOX_INVISIBLE_X, BBOX_INVISIBLE_Y);
#pragma unroll
    for (int st = 0; st < CNT_STEPS; st++) {
        const uint2 bb = bbs[st];
        const BinRect br = bin_rect(bb, g);
        {
            const uint32_t r = begin + st * CNT_THREADS + threadIdx.x;
            if (r < n) rects[r] = pack_rect(br);
        }
        if (br.x0 <= br.x1) {
            vis++;
            const int tx0 = max((int)(bb.x & 0xffff) / TILE, g.bx_lo * BIN_TILES);
            const int tx1 = min((int)(bb.x >> 16) / TILE, g.bx_hi * BIN_TILES - 1);
            tiles += (uint32_t)((tx1 - tx0 + 1) * ((int)(bb.y >> 16) / TILE - (int)(bb.y & 0xffff) / TILE + 1));
            for (int y = br.y0; y <= br.y1; y++)
                for (int x = br.x0; x <= br.x1; x++) atomicAdd(&s_cnt[y * nbxb + x], 1u);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vis += __shfl_xor(vis, off);
        tiles += __shfl_xor(tiles, off);
    }
    constexpr int NW = CNT_THREADS / WAVE;
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = vis; s_red[NW + (threadIdx.x >> 6)] = tiles; }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += CNT_THREADS) table[(size_t)blockIdx.x * nbins + b] = s_cnt[b];
    if (threadIdx.x == 0) {
        uint32_t v = 0, t = 0;
        for (int w = 0; w < NW; w++) { v += s_red[w]; t += s_red[NW + w]; }
        blk_counts[blockIdx.x] = make_uint2(v, t);
    }
}

// ---------------------------------------------------------------------------
// scan: one wave per bin: table[block][bin] <- entries of earlier workgroups in this bin;
// bin_total[bin] = the bin's entry count.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BIN_THREADS) void k_bin_scan(uint32_t* __restrict__ table, uint32_t* __restrict__ bin_total,
                                                          int nbins, uint32_t nblocks)
{
    const int lane = threadIdx.x & 63;
    const int bin = blockIdx.x * BIN_WAVES + (threadIdx.x >> 6);
    if (bin >= nbins) return;
    // One wave per column: lane l takes rows l, l+64, ...  The strided loads of a batch are issued together
    // (independent), then scanned; doing them one at a time made this kernel a chain of L2 round trips.
    constexpr int BATCH = 8;
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += BATCH * WAVE) {
        uint32_t v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; k++) {
            const uint32_t b = b0 + k * WAVE + lane;
            v[k] = (b < nblocks) ? table[(size_t)b * nbins + bin] : 0u;
        }
#pragma unroll
        for (int k = 0; k < BATCH; k++) {
            const uint32_t b = b0 + k * WAVE + lane;
            uint32_t incl = v[k];
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            if (b < nblocks) table[(size_t)b * nbins + bin] = run + incl - v[k];
            run += __shfl(incl, WAVE - 1);
        }
    }
    if (lane == 0) bin_total[bin] = run;
}

// ---------------------------------------------------------------------------
// finalize (one workgroup): bin_start = exclusive scan of bin_total; the compositor's work items
// (bin, segment of seg_len list entries), one per segment, at least one per bin; frame counters.
// ---------------------------------------------------------------------------
constexpr int FIN_THREADS = 1024;
constexpr int FIN_WAVES = FIN_THREADS / WAVE;

struct U3 { uint32_t a, b, c; };

// exclusive scan of three independent u32 streams over the workgroup's FIN_THREADS threads; totals in *tot
__device__ __forceinline__ U3 block_exclusive_scan3(U3 v, uint32_t (*s_w)[FIN_WAVES], U3* tot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    U3 inc = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t ta = __shfl_up(inc.a, off), tb = __shfl_up(inc.b, off), tc = __shfl_up(inc.c, off);
        if (lane >= off) { inc.a += ta; inc.b += tb; inc.c += tc; }
    }
    if (lane == WAVE - 1) { s_w[0][wave] = inc.a; s_w[1][wave] = inc.b; s_w[2][wave] = inc.c; }
    __syncthreads();
    U3 base = {0, 0, 0}, total = {0, 0, 0};
#pragma unroll
    for (int w = 0; w < FIN_WAVES; w++) {
        const uint32_t xa = s_w[0][w], xb = s_w[1][w], xc = s_w[2][w];
        if (w < wave) { base.a += xa; base.b += xb; base.c += xc; }
        total.a += xa; total.b += xb; total.c += xc;
    }
    __syncthreads();
    *tot = total;
    return U3{base.a + inc.a - v.a, base.b + inc.b - v.b, base.c + inc.c - v.c};
}

__global__ __launch_bounds__(FIN_THREADS) void k_bin_finalize(const uint32_t* __restrict__ bin_total, int nbins,
                                                              uint32_t seg_len, uint32_t max_items, uint32_t capacity,
                                                              const uint2* __restrict__ blk_counts, uint32_t nblocks,
                                                              uint32_t* __restrict__ bin_start,
                                                              uint32_t* __restrict__ seg_start, uint32_t* __restrict__ items,
                                                              uint32_t* __restrict__ overflow, uint64_t* __restrict__ visible,
                                                              uint64_t* __restrict__ tile_entries, uint64_t* __restrict__ accum)
{
    __shared__ uint32_t s_w[3][FIN_WAVES];
    const int per = (nbins + FIN_THREADS - 1) / FIN_THREADS;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nbins);
    // Items are emitted heaviest first: every full segment (seg_len entries) before every partial or
    // empty one, so the compositor's queue hands out the long items while the chip is still full.
    U3 mine = {0, 0, 0};  // entries, segments, full segments of this thread's bins
    for (int b = b0; b < b1; b++) {
        const uint32_t c = bin_total[b];
        mine.a += c;
        mine.b += max(1u, (c + seg_len - 1) / seg_len);
        mine.c += c / seg_len;
    }
    U3 tot;
    const U3 ex3 = block_exclusive_scan3(mine, s_w, &tot);
    uint32_t ex = ex3.a, sx = ex3.b, fx = ex3.c;
    uint32_t px = tot.c + (sx - fx);  // partial/empty items follow all full ones
    // A frame whose lists do not fit (entries > capacity or items > max_items) must not be composited:
    // it publishes no work items at all (every index the compositor derives stays in range), raises the
    // overflow word, and the host regrows the buffers and renders the frame again (gsr_sync).
    const bool fits = tot.a <= capacity && tot.b <= max_items;
    for (int b = b0; b < b1; b++) {
        const uint32_t c = bin_total[b];
        const uint32_t ns = max(1u, (c + seg_len - 1) / seg_len);
        const uint32_t nf = c / seg_len;
        bin_start[b] = fits ? ex : 0u;
        seg_start[b] = fits ? sx : 0u;
        if (fits)
            for (uint32_t k = 0; k < ns; k++) items[(k < nf) ? fx + k : px + (k - nf)] = (uint32_t)b | (k << 16);
        ex += c;
        sx += ns;
        fx += nf;
        px += ns - nf;
    }
    // frame counters
    U3 cnt = {0, 0, 0};
    for (uint32_t b = threadIdx.x; b < nblocks; b += FIN_THREADS) { cnt.a += blk_counts[b].x; cnt.b += blk_counts[b].y; }
    U3 ctot;
    block_exclusive_scan3(cnt, s_w, &ctot);
    if (threadIdx.x == 0) {
        bin_start[nbins] = fits ? tot.a : 0u;
        seg_start[nbins] = fits ? tot.b : 0u;
        if (!fits) atomicOr(overflow, tot.a > capacity ? 1u : 2u);
        accum[4] = tot.a;  // entries this frame needs (the host sizes the regrowth from it)
        *visible = ctot.a;
        *tile_entries = ctot.b;
        accum[0] += ctot.a; accum[1] += tot.a; accum[2] += ctot.b; accum[3] += 1;
    }
}

// ---------------------------------------------------------------------------
// scatter: list[...] = splat index, bins in raster order, depth order inside a bin.
// Workgroup = 16 waves over 2048 consecutive ranks.  The ranks form 4 groups of 512 (8 steps of 64);
// group g is shared by waves 4g..4g+3, each owning 2 of its steps, so input order is
// (workgroup, group, step, lane).  Sixteen waves instead of four do the same work with 4x the
// latency hiding: the pass is a chain of dependent LDS / global round trips, not arithmetic.
// ---------------------------------------------------------------------------
constexpr int SCAT_THREADS = 1024;
constexpr int SCAT_GROUPS = 4;                       // rank groups per workgroup (512 ranks each)
constexpr int SCAT_WAVES_PER_GROUP = 4;
constexpr int SCAT_STEPS_PER_WAVE = BIN_STEPS / SCAT_WAVES_PER_GROUP;  // 2
static_assert(SCAT_GROUPS * BIN_STEPS * WAVE == (int)BIN_RANKS_PER_BLOCK, "scatter and count must cut the ranks alike");

__global__ __launch_bounds__(SCAT_THREADS) void k_bin_scatter(const uint32_t* __restrict__ depth_index,
                                                              const uint32_t* __restrict__ rects,
                                                              const uint32_t* __restrict__ count, BinGrid g,
                                                              const uint32_t* __restrict__ table,
                                                              const uint32_t* __restrict__ bin_start,
                                                              uint32_t* __restrict__ list, uint32_t capacity,
                                                              uint32_t* __restrict__ overflow)
{
    extern __shared__ uint32_t s_mem[];
    const uint32_t n = *count;
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int group = wave / SCAT_WAVES_PER_GROUP, sub = wave % SCAT_WAVES_PER_GROUP;
    // LDS: base[nbins] (u32: the workgroup's first slot in each bin), pair[2][nbins] (two 16-bit per-group
    // counts/offsets per word: groups 0|1 and 2|3; a group holds 512 ranks, so 16 bits suffice), then the
    // lane sets: per group, per step: [nbxb] column words + [nby] row words.
    uint32_t* base = s_mem;
    uint32_t* pair = s_mem + nbins;
    const int nmask = nbxb + g.nby;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(s_mem + ((3 * nbins + 1) & ~1));
    unsigned long long* gmask = masks + (size_t)group * BIN_STEPS * nmask;   // step s of my group: gmask + s*nmask
    uint32_t* mypair = pair + (size_t)(group >> 1) * nbins;
    const int myshift = (group & 1) * 16;

    for (int b = threadIdx.x; b < 2 * nbins; b += SCAT_THREADS) pair[b] = 0;
    for (int b = threadIdx.x; b < SCAT_GROUPS * BIN_STEPS * nmask; b += SCAT_THREADS) masks[b] = 0;
    __syncthreads();

    // this wave's 2 steps of 64 consecutive ranks
    const uint32_t gbegin = blockIdx.x * BIN_RANKS_PER_BLOCK + group * (BIN_STEPS * WAVE);
    uint32_t idx[SCAT_STEPS_PER_WAVE];
    BinRect br[SCAT_STEPS_PER_WAVE];
#pragma unroll
    for (int k = 0; k < SCAT_STEPS_PER_WAVE; k++) {
        const uint32_t r = gbegin + (sub * SCAT_STEPS_PER_WAVE + k) * WAVE + lane;
        idx[k] = (r < n) ? depth_index[r] : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < SCAT_STEPS_PER_WAVE; k++) {
        const uint32_t r = gbegin + (sub * SCAT_STEPS_PER_WAVE + k) * WAVE + lane;
        br[k] = unpack_rect((r < n) ? rects[r] : 1u);
    }
    // phase 1: per-group counts, and every lane ORs its bit into the column/row lane sets of its box
    const uint32_t one = 1u << myshift;
    const unsigned long long mybit = 1ull << lane;
#pragma unroll
    for (int k = 0; k < SCAT_STEPS_PER_WAVE; k++) {
        const BinRect b = br[k];
        unsigned long long* colm = gmask + (sub * SCAT_STEPS_PER_WAVE + k) * nmask;
        unsigned long long* rowm = colm + nbxb;
        for (int x = b.x0; x <= b.x1; x++) atomicOr(&colm[x], mybit);
        for (int y = b.y0; y <= b.y1; y++) {
            if (b.x0 <= b.x1) atomicOr(&rowm[y], mybit);
            for (int x = b.x0; x <= b.x1; x++) atomicAdd(&mypair[y * nbxb + x], one);
        }
    }
    __syncthreads();
    // phase 2: counts -> offsets of each group inside the workgroup's run; workgroup base from the table
    for (int b = threadIdx.x; b < nbins; b += SCAT_THREADS) {
        base[b] = bin_start[b] + table[(size_t)blockIdx.x * nbins + b];
        const uint32_t c01 = pair[b], c23 = pair[nbins + b];
        const uint32_t o1 = c01 & 0xffffu, o2 = o1 + (c01 >> 16), o3 = o2 + (c23 & 0xffffu);
        pair[b] = o1 << 16;                // group 0: 0, group 1: o1
        pair[nbins + b] = o2 | (o3 << 16); // group 2: o2, group 3: o3
    }
    __syncthreads();
    // phase 3: slots.  The set of lanes of step s covering bin (X,Y) is col[s][X] & row[s][Y]; a splat's slot is
    // base[bin] + its group's offset + the sizes of the sets of the group's earlier steps + the number of
    // lower lanes in its own step's set: input order, from ballot-style arithmetic on LDS words that are
    // read-only by now (no ordered atomics, no running counter).
#pragma unroll
    for (int k = 0; k < SCAT_STEPS_PER_WAVE; k++) {
        const BinRect b = br[k];
        const int st = sub * SCAT_STEPS_PER_WAVE + k;
        const uint32_t myidx = idx[k];
        for (int y = b.y0; y <= b.y1; y++) {
            for (int x = b.x0; x <= b.x1; x++) {
                uint32_t dst = base[y * nbxb + x] + ((mypair[y * nbxb + x] >> myshift) & 0xffffu);
                for (int e = 0; e < st; e++)  // entries the earlier steps of this group put into the bin
                    dst += (uint32_t)__popcll(gmask[e * nmask + x] & gmask[e * nmask + nbxb + y]);
                dst += lanes_below64(gmask[st * nmask + x] & gmask[st * nmask + nbxb + y]);
                if (dst < capacity) list[dst] = myidx;
                else atomicOr(overflow, 1u);
            }
        }
    }
}

void launch_bin(const BinBuffers& b, const BinGrid& g, uint32_t n, hipStream_t s)
{
    const int nbxb = g.bx_hi - g.bx_lo, nbins = nbxb * g.nby;
    if (nbins <= 0) return;
    const dim3 grid(b.nblocks), block(BIN_THREADS);
    const size_t lds = (size_t)((3 * nbins + 1) & ~1) * 4 + (size_t)SCAT_GROUPS * BIN_STEPS * (nbxb + g.nby) * 8;
    // dynamic LDS above the 64 KiB default needs the attribute raised (4K: 8160 bins -> ~146 KiB).  Set per call:
    // the attribute belongs to the current device's copy of the kernel, and this is off the per-frame fast path
    // (1080p needs 49 KiB).
    if (lds > 60 * 1024) {
        const int want = (int)std::min<size_t>(lds + 1024, 160 * 1024 - 256);
        if (hipFuncSetAttribute((const void*)k_bin_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess)
            (void)hipGetLastError();  // the launch below then reports the real failure
    }
    if (n) {
        hipLaunchKernelGGL(k_bin_count, grid, dim3(CNT_THREADS), nbins * sizeof(uint32_t), s, b.depth_index, b.bbox, b.count, g, b.table,
                           b.blk_counts, b.rects);
        hipLaunchKernelGGL(k_bin_scan, dim3((nbins + BIN_WAVES - 1) / BIN_WAVES), block, 0, s, b.table, b.bin_total, nbins,
                           b.nblocks);
    }
    hipLaunchKernelGGL(k_bin_finalize, dim3(1), dim3(FIN_THREADS), 0, s, (const uint32_t*)b.bin_total, nbins, b.seg_len,
                       b.max_items, b.capacity, (const uint2*)b.blk_counts, n ? b.nblocks : 0u, b.bin_start, b.seg_start, b.items,
                       b.overflow, b.visible, b.tile_entries, b.accum);
    if (n)
        hipLaunchKernelGGL(k_bin_scatter, grid, dim3(SCAT_THREADS), lds, s, b.depth_index, (const uint32_t*)b.rects, b.count, g, (const uint32_t*)b.table,
                           (const uint32_t*)b.bin_start, b.list, b.capacity, b.overflow);
}

}  // namespace gsr
