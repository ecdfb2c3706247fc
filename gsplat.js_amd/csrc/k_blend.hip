// Front-to-back tile compositor: frag.glsl.ts:13-21 plus the blend state of
// WebGLRenderer.ts:139-142,279-285, evaluated per pixel centre in f32.
//
// One workgroup (4 waves) per work item = (32x32 screen bin, segment of the
// bin's depth-ordered splat list); one wave per 16x16 tile, four pixels per
// lane.  The segment is streamed through LDS in chunks of 256 entries: every
// thread stages one record plus a 16-bit mask of the bin's 8x8-pixel quadrants
// its splat can touch; then every wave picks its tile's entries out of the chunk
// with a wave64 ballot on its mask nibble (order preserving) and walks the set
// bits, reading each record as an LDS broadcast and visiting only the flagged
// quadrants.  No per-tile list is ever materialised in HBM and nothing spills:
// the LDS footprint is fixed (13 KiB) however long the list is.
//
// Coordinates are bin-relative: staging folds the splat centre into the two
// constants -dot(u, c - o), -dot(w, c - o) (o = centre of the bin's first pixel),
// so a pixel costs vx = ux*px + (uy*py + ncu) with small exact integers px, py,
// and the row terms are shared by the two quadrants of a row.
//
// Long lists (screen centre) are cut into segments that are composited
// concurrently by different workgroups, each from (colour 0, transmittance 1);
// "under" compositing is associative, so the partial (colour, transmittance)
// pairs of a bin are folded front to back:
//     C = C0 + T0*C1 + T0*T1*C2 + ...,  T = T0*T1*T2*...
// by the workgroup that delivers the bin's last segment (see the end of k_blend;
// k_combine is the same fold as a separate launch, GSR_FUSE_COMBINE=0).
// This removes the serial critical path of the heaviest tiles where nothing else
// does.  Where the frame saturates, something else does: a quadrant whose pixels
// can no longer change a bit (transmittance under half an ulp of the colour) is
// not visited any more, a bin then needs only the few thousand entries in front,
// and k_bin_finalize makes the work items whole bins (a segment would start from
// transmittance 1 and never saturate on its own).  With early termination
// enabled a bin is one item too, processed until every pixel is below the
// caller's threshold.
//
// Compiled with -ffp-contract=off; the fused multiply-adds below are explicit,
// so the coverage test (|vPosition|^2 <= 4) is bit-identical to the oracle's.
#include "gsr_internal.h"

namespace gsr {

constexpr int BLEND_THREADS = 256;
constexpr int CHUNK = BLEND_THREADS;
constexpr int BIN_PIXELS = BIN_PX * BIN_PX;
constexpr float LOG2E = 1.4426950408889634f;
constexpr uint32_t SAT_FROM = CHUNK;   // entries of a long work item before its first saturation test

#ifdef GSR_BLEND_STAMPS
// Diagnostic build only (scripts/build_exp.sh stamps "-DGSR_BLEND_STAMPS", read by scripts/blend_stamps.py): where a
// compositor wave spends its cycles and when it starts and ends.  Per wave: [0] lifetime, [1] barrier at chunk start,
// [2] staging, [3] barrier after staging, [4] composite loop (shader cycles), [5] start, [6] end (s_memrealtime,
// 100 MHz, common to all XCDs), [7] entry visits, [8] items, [9] start of the last item, [10] longest item (ticks),
// [11] its entries, [12] its entry visits, [13] its bin.  Never compiled into the shipped library.
__device__ unsigned int g_blend_stamps[4096 * 4 * 16];
#define STAMP(v) const unsigned int v = (unsigned int)__builtin_readcyclecounter()
#ifdef GSR_BLEND_COUNT_QUADS   // slow: counts visited quadrants, those without a covered pixel, and covered pixels
#define GSR_COUNT_QUAD(q) { const unsigned long long cb_ = __ballot((q) <= 4.0f); a_quads++; a_empty += cb_ == 0ull; a_cov += __popcll(cb_); }
#else
#define GSR_COUNT_QUAD(q)
#endif
#else
#define STAMP(v)
#define GSR_COUNT_QUAD(q)
#endif

// 7 waves per SIMD: the kernel needs 74 VGPRs unconstrained (6 waves); capped at 72 it spills one register pair that is
// stored once per workgroup and reloaded once per work item, and the seventh wave hides more of the LDS/barrier
// waits (measured: k_blend -5.7 % alone; 8 waves = 64 VGPRs spills inside the loops and is no better).
// (Fetching the next entry's record before the current entry's arithmetic -- two register sets, loop unrolled by two --
//  was measured 24 % SLOWER at 7 and at 6 waves: the compiler's s_waitcnt placement in the rotated loop waits for the
//  new reads as well, and the extra scalar control costs more than the hidden LDS latency; the other waves of the
//  SIMD already cover that latency.)
__global__ __launch_bounds__(BLEND_THREADS) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_blend(const uint32_t* __restrict__ items,
                                                         const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ bin_start,
                                                         const uint32_t* __restrict__ list,
                                                         const Record* __restrict__ rec,
                                                         const float4* __restrict__ shcol, float4* __restrict__ fb,
                                                         float4* __restrict__ partial, uint32_t* __restrict__ queue,
                                                         BinGrid g, float eps, const uint32_t* __restrict__ seg_len_dev, uint32_t capacity,
                                                         uint32_t nsplats, uint32_t* __restrict__ bin_done, uint32_t saturate,
                                                         uint32_t prio_a, uint32_t prio_b, uint32_t prio_c)
{
    const uint32_t seg_len = *seg_len_dev;  // this frame's list entries per work item (k_bin_finalize)
    // [0]: ux, uy, -dot(u, c - bin origin), wx   [1]: wy, -dot(w, c - bin origin), log2(opacity), blue   [2]: red, green
    __shared__ float4 s_rec[3][CHUNK];   // one address register serves the three reads of an entry
    __shared__ uint32_t s_mask[CHUNK];
    __shared__ uint32_t s_done;
    __shared__ uint32_t s_item;
    __shared__ uint32_t s_last;

    const int nbxb = g.bx_hi - g.bx_lo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    const uint32_t total_items = seg_len_dev[1];   // the frame's work items (k_bin_finalize; heavy bins count four)
    const uint32_t wlen = seg_len_dev[2];          // entries per window segment of a multi-segment bin, or 0 (k_bin_finalize, "front window")

    // Work items come from one device-wide queue (items are ordered heaviest first), so a workgroup
    // that drew light items simply draws more: no static assignment, no long pole.
    // The first item of a workgroup is its own index (the queue starts at gridDim.x, k_bin_finalize sets it): a
    // kernel start with ~2000 workgroups drawing from one counter serialises ~2000 same-address atomics.
#ifdef GSR_BLEND_STAMPS
    unsigned int a_waitA = 0, a_stage = 0, a_waitB = 0, a_comp = 0, a_entries = 0;
    unsigned int a_quads = 0, a_empty = 0, a_cov = 0;
    unsigned int a_items = 0, a_last_start = 0, a_max_dur = 0, a_max_len = 0, a_max_vis = 0, a_max_bin = 0, a_item_t0 = 0, a_item_vis0 = 0, a_item_len = 0, a_item_bin = 0;
    const unsigned int t_kernel0 = (unsigned int)__builtin_readcyclecounter();
    const unsigned int t_real0 = (unsigned int)__builtin_amdgcn_s_memrealtime();
#endif
    bool first = true;
    for (;;) {
        __syncthreads();  // the previous item no longer uses the LDS words
        if (threadIdx.x == 0) {
            s_item = first ? blockIdx.x : atomicAdd(queue, 1u);
            s_done = 0;
        }
        first = false;
        __syncthreads();
        const uint32_t qi = s_item;
        if (qi >= total_items) break;
        const uint32_t it = items[qi];
        const int bin = (int)(it & 0xffffu);
        // Wave priority by the item's place in the queue (items are ordered heaviest first): the longest items are the
        // kernel's pole when seven waves share a SIMD and every one of them gets a seventh of its issue slots -- the
        // waves of the first prio_a items issue ahead of the others on their SIMDs, then those below prio_b, prio_c
        // (longest job first; the lighter items fill the slots the heavy ones leave while they wait on LDS or barriers).
        {
            const int pr = (qi < prio_a ? 1 : 0) + (qi < prio_b ? 1 : 0) + (qi < prio_c ? 1 : 0);
            if (pr == 3) __builtin_amdgcn_s_setprio(3);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#ifdef GSR_BLEND_STAMPS
        a_items++; a_item_t0 = (unsigned int)__builtin_amdgcn_s_memrealtime(); a_last_start = a_item_t0; a_item_vis0 = a_entries; a_item_bin = (unsigned int)bin;
#endif
        // A tile item (ITEM_TILE0 + t: a heavy single-segment bin, handed out as four items): this workgroup composites
        // tile t alone and every wave takes ONE of its 8x8 quadrants -- one pixel per lane, the (.)00 accumulators --
        // instead of a whole tile: the same arithmetic per pixel, a third of the entries and quadrants per wave.
        const bool tile_item = (it >> 16) >= ITEM_TILE0;
        const uint32_t seg = tile_item ? 0u : it >> 16;
        const int tile = tile_item ? (int)((it >> 16) - ITEM_TILE0) : wave;
        const uint32_t nseg = seg_start[bin + 1] - seg_start[bin];
        const int by = bin / nbxb, bxl = bin - by * nbxb;
        const int binX0 = (g.bx_lo + bxl) * BIN_PX, binY0 = by * BIN_PX;
        const int ox = (tile & 1) * TILE + (tile_item ? (wave & 1) * 8 : 0), oy = (tile >> 1) * TILE + (tile_item ? (wave >> 1) * 8 : 0);
        const int X0 = binX0 + ox, Y0 = binY0 + oy;
        // which bits of an entry's 16-bit quadrant mask are mine: my tile's nibble, or my one quadrant of it
        const int mask_shift = tile_item ? tile * 4 + wave : wave * 4;
        const uint32_t mask_sel = tile_item ? 1u : 15u;
        // pixel centres relative to the centre of the bin's first pixel: small exact integers
        const float pxf0 = (float)(ox + lx), pxf1 = pxf0 + 8.0f;
        const float pyf0 = (float)(oy + ly), pyf1 = pyf0 + 8.0f;
        const float bx0c = (float)binX0 + 0.5f, by0c = (float)binY0 + 0.5f;

        float T00 = 1.f, T10 = 1.f, T01 = 1.f, T11 = 1.f;  // Tij: pixel (x+8i, y+8j); 1 - alpha
        float r00 = 0.f, r10 = 0.f, r01 = 0.f, r11 = 0.f;
        float g00 = 0.f, g10 = 0.f, g01 = 0.f, g11 = 0.f;
        float b00 = 0.f, b10 = 0.f, b01 = 0.f, b11 = 0.f;

        // A bin with a front window (k_bin_finalize): its nseg items are the window's segments of wlen entries, and the
        // entries behind the window belong to whichever workgroup folds the window (below).
        const bool windowed = wlen != 0u && nseg > 1u;
        const uint32_t step = windowed ? wlen : seg_len;
        const uint32_t bin_begin = bin_start[bin];
        const uint32_t bin_end = min(bin_start[bin + 1], capacity);
        uint32_t begin = min(bin_begin + seg * step, bin_end);
        uint32_t end = min(begin + step, bin_end);
        const uint32_t tail_begin = windowed ? min(bin_begin + nseg * wlen, bin_end) : bin_end;
        // only long items test for saturation (a window segment starts from transmittance 1 like any segment: it cannot)
        bool sat_item = saturate != 0u && !windowed && end - begin > 2u * CHUNK;
        uint32_t sat_from = SAT_FROM;
        // quadrants of mine that can still change (wave-uniform); see the saturation test below.  Quadrants that lie outside
        // the image (the last bin row of 1080p: rows 1080..1087) never could: their pixels are not stored.
        uint32_t alive0 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (X0 + 8 * (q & 1) < g.W && Y0 + 8 * (q >> 1) < g.H) alive0 |= 1u << q;
        alive0 &= mask_sel;
        uint32_t alive = alive0;
        bool done = alive0 == 0u;
        if (done && lane == 0) atomicAdd(&s_done, 1u);
        bool final_pass = false, write_fb = false;
#ifdef GSR_BLEND_STAMPS
        a_item_len = end - begin;
#endif

        for (;;) {   // pass 1: the item's own entries; pass 2, only in the workgroup that folds a window: the entries behind it
        for (uint32_t base = begin; base < end; base += CHUNK) {
            STAMP(t_c0);
            __syncthreads();  // previous chunk fully consumed (and s_done visible)
            STAMP(t_cA);
            if (s_done == BLEND_THREADS / WAVE) break;  // every tile of the bin is saturated
            // ---- stage one entry per thread: record, unpacked colour, and a 16-bit mask of the bin's
            //      8x8-pixel quadrants the splat can touch (bit = tile*4 + quadrant).  The mask is a
            //      conservative cull only: a quadrant is dropped when it lies outside the oriented box
            //      |vPosition.x|,|vPosition.y| <= 2 (separating axes u, w) or farther from the centre than the
            //      longer semi-axis.  Pixels that pass are still tested exactly (q <= 4) below. ----
            const uint32_t e = base + threadIdx.x;
            uint32_t mask = 0;
            if (e < end) {
                const uint32_t i = min(list[e], nsplats - 1u);
                const float4* rp = reinterpret_cast<const float4*>(rec + i);
                const float4 ra = rp[0], rb = rp[1];
                const float eu = 3.5f * (fabsf(ra.z) + fabsf(ra.w)) + 2.0005f;
                const float ew = 3.5f * (fabsf(rb.x) + fabsf(rb.y)) + 2.0005f;
                const float minlen2 = fminf(ra.z * ra.z + ra.w * ra.w, rb.x * rb.x + rb.y * rb.y);
                float ucol[4], wcol[4], dcol[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float dc = (float)(binX0 + 8 * k + 4) - ra.x;
                    ucol[k] = ra.z * dc; wcol[k] = rb.x * dc;
                    const float dd = fmaxf(fabsf(dc) - 3.5f, 0.0f);
                    dcol[k] = dd * dd;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float dc = (float)(binY0 + 8 * r + 4) - ra.y;
                    const float ur = ra.w * dc, wr = rb.y * dc;
                    const float dd = fmaxf(fabsf(dc) - 3.5f, 0.0f);
                    const float d2 = dd * dd;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const bool hit = fabsf(ucol[k] + ur) <= eu && fabsf(wcol[k] + wr) <= ew && (dcol[k] + d2) * minlen2 <= 4.002f;
                        // quadrant (k, r) of the bin -> tile (k>>1, r>>1), quadrant (k&1, r&1)
                        if (hit) mask |= 1u << ((((r >> 1) * 2 + (k >> 1)) << 2) + ((r & 1) * 2 + (k & 1)));
                    }
                }
                const uint32_t rgb8 = __float_as_uint(rb.w);
                float cr = (float)(rgb8 & 0xffu) * (1.0f / 255.0f), cg = (float)((rgb8 >> 8) & 0xffu) * (1.0f / 255.0f),
                      cb = (float)((rgb8 >> 16) & 0xffu) * (1.0f / 255.0f);
                if (rgb8 & RGB8_IN_SHCOL) {  // SH-coloured splat: the projection kernel evaluated its colour for this view
                    const float4 sc = shcol[i];
                    cr = sc.x; cg = sc.y; cb = sc.z;
                }
                const float cxr = ra.x - bx0c, cyr = ra.y - by0c;
                const float ncu = -__builtin_fmaf(ra.w, cyr, ra.z * cxr), ncw = -__builtin_fmaf(rb.y, cyr, rb.x * cxr);
                s_rec[0][threadIdx.x] = make_float4(ra.z, ra.w, ncu, rb.x);
                s_rec[1][threadIdx.x] = make_float4(rb.y, ncw, rb.z, cb);
                *reinterpret_cast<float2*>(&s_rec[2][threadIdx.x]) = make_float2(cr, cg);
            }
            s_mask[threadIdx.x] = mask;
            STAMP(t_cS);
            __syncthreads();
            STAMP(t_cB);

            if (!done) {
                const uint32_t cnt = min((uint32_t)CHUNK, end - base);
                for (uint32_t c0 = 0; c0 < cnt; c0 += WAVE) {
                    const uint32_t mine = (s_mask[c0 + lane] >> mask_shift) & alive;  // entry (c0+lane) vs my live quadrants
                    uint64_t bal = __ballot(mine != 0u);
#ifdef GSR_BLEND_STAMPS
                    a_entries += __popcll(bal);
#endif
#define GSR_QUAD(BIT, PX, UR, WR, T, R, G, B_)                                                     \
    if (qm & (BIT)) {                                                                              \
        const float vx_ = __builtin_fmaf(ux, (PX), (UR)), vy_ = __builtin_fmaf(wx, (PX), (WR));    \
        const float q_ = __builtin_fmaf(vy_, vy_, vx_ * vx_);                                      \
        GSR_COUNT_QUAD(q_)                                                                         \
        if (q_ <= 4.0f) {                                                                          \
            const float w_ = (T) * __builtin_amdgcn_exp2f(__builtin_fmaf(q_, -LOG2E, la));         \
            (T) = (T) - w_;                                                                        \
            (R) = __builtin_fmaf(w_, cr, (R));                                                     \
            (G) = __builtin_fmaf(w_, cg, (G));                                                     \
            (B_) = __builtin_fmaf(w_, cb, (B_));                                                   \
        }                                                                                          \
    }
                    // vPosition = (ux*px + (uy*py - dot(u,c)), wx*px + (wy*py - dot(w,c))), all bin-relative:
                    // the row terms are shared by the two quadrants of a row
                    // frag.glsl.ts:15  if (A < -4.0) discard;   (A = -q)
                    // frag.glsl.ts:16-20  B = clamp(exp(A) * opacity, 0, 1)  (never clamps: exp(A) <= 1, opacity <= 1)
                    // blend: dst += (1 - dst.a) * (B*rgb, B)
#define GSR_ENTRY(RA, RB, RC, QM)                                                                              \
    {                                                                                                          \
        const uint32_t qm = (QM);                                                                              \
        const float ux = RA.x, uy = RA.y, ncu = RA.z, wx = RA.w, wy = RB.x, ncw = RB.y, la = RB.z;             \
        const float cr = RC.x, cg = RC.y, cb = RB.w;                                                           \
        const float ur0 = __builtin_fmaf(uy, pyf0, ncu), ur1 = __builtin_fmaf(uy, pyf1, ncu);                  \
        const float wr0 = __builtin_fmaf(wy, pyf0, ncw), wr1 = __builtin_fmaf(wy, pyf1, ncw);                  \
        GSR_QUAD(1u, pxf0, ur0, wr0, T00, r00, g00, b00)                                                       \
        GSR_QUAD(2u, pxf1, ur0, wr0, T10, r10, g10, b10)                                                       \
        GSR_QUAD(4u, pxf0, ur1, wr1, T01, r01, g01, b01)                                                       \
        GSR_QUAD(8u, pxf1, ur1, wr1, T11, r11, g11, b11)                                                       \
    }
#define GSR_FETCH(RA, RB, RC, QM, J)                                                 \
    RA = s_rec[0][c0 + (J)];                                                          \
    RB = s_rec[1][c0 + (J)];                                                          \
    RC = *reinterpret_cast<const float2*>(&s_rec[2][c0 + (J)]);                       \
    QM = __builtin_amdgcn_readlane(mine, (J));  /* wave-uniform quadrant mask */
                    while (bal) {
                        const int j = __builtin_ctzll(bal);
                        bal &= bal - 1;
                        float4 ra, rb;
                        float2 rc;
                        uint32_t qm1;
                        GSR_FETCH(ra, rb, rc, qm1, j)
                        GSR_ENTRY(ra, rb, rc, qm1)
                    }
#undef GSR_FETCH
#undef GSR_ENTRY
#undef GSR_QUAD
                    if (eps > 0.0f) {
                        const float tmax = fmaxf(fmaxf(T00, T10), fmaxf(T01, T11));
                        if (__ballot(tmax >= eps) == 0ull) {
                            done = true;
                            if (lane == 0) atomicAdd(&s_done, 1u);
                            break;
                        }
                    }
                    // ---- saturation that changes no bit (after every 64 entries of a long item) ----
                    // A pixel whose transmittance is below 2^-27 of its smallest colour channel is finished: every later
                    // weight is w <= T (opacity <= 1, exp <= 1), colours are <= 1, so w*c is under half an ulp of each channel
                    // and fma(w, c, C) returns C; its alpha is 1 - T = 1.0f for any T that small.  The segment's own T
                    // would still shrink, but only ever multiplies later segments' colours (the fold), whose terms are then
                    // under half an ulp of the folded colour as well: the image is bit-identical to compositing every entry
                    // (test_saturated_quadrants_are_skipped_without_changing_a_bit).  A quadrant whose 64 pixels are all
                    // finished drops out of `alive`; a tile with no live quadrant is done, a bin with no live tile ends its
                    // work item (s_done).  Pixels with a zero channel finish only at T == 0.
                    // (Items of up to two chunks -- all a frame that does not saturate has -- never run the test, which
                    //  cost 3 % on C2, and nothing saturates within an item's first SAT_FROM entries.)
                    if (sat_item && base - begin + c0 >= sat_from) {
                        constexpr float K = 0x1p-27f, NEAR = 1e-6f;   // nothing above NEAR can pass the test: a cheap filter first
#define GSR_FINISHED(BIT, T, R, G, B_)                                                                          \
    if ((alive & (BIT)) && __ballot((T) >= NEAR) == 0ull &&                                                      \
        __ballot(!((T) < K * fminf((R), fminf((G), (B_))) || (T) == 0.0f)) == 0ull)                              \
        alive &= ~(BIT);
                        GSR_FINISHED(1u, T00, r00, g00, b00)
                        GSR_FINISHED(2u, T10, r10, g10, b10)
                        GSR_FINISHED(4u, T01, r01, g01, b01)
                        GSR_FINISHED(8u, T11, r11, g11, b11)
#undef GSR_FINISHED
                        if (alive == 0u) {
                            done = true;
                            if (lane == 0) atomicAdd(&s_done, 1u);
                            break;
                        }
                    }
                }
            }
#ifdef GSR_BLEND_STAMPS
            {
                const unsigned int t_cC = (unsigned int)__builtin_readcyclecounter();
                a_waitA += t_cA - t_c0; a_stage += t_cS - t_cA; a_waitB += t_cB - t_cS; a_comp += t_cC - t_cB;
            }
#endif
        }

        if (nseg == 1u || final_pass) { write_fb = true; break; }
        {
            // ---- partial (colour, transmittance) of this segment, slot-major: fully coalesced ----
            float4* p0 = partial + (size_t)seg_start[bin] * BIN_PIXELS + wave * (TILE * TILE) + lane;
            float4* p = p0 + (size_t)seg * BIN_PIXELS;
            if (!bin_done) {   // k_combine folds the bin after this kernel
                p[0] = make_float4(r00, g00, b00, T00);
                p[64] = make_float4(r10, g10, b10, T10);
                p[128] = make_float4(r01, g01, b01, T01);
                p[192] = make_float4(r11, g11, b11, T11);
                break;
            }
            // The workgroup that delivers a bin's LAST segment folds the bin itself (front to back, k_combine's fixed
            // order: which workgroup does it changes nothing in the result), so the fold runs beside the other
            // workgroups' arithmetic and the k_combine launch goes away.
            // Visibility between workgroups on different XCDs (one L2 each) WITHOUT a release fence: an agent-scope
            // release is buffer_wbl2, a write-back of the XCD's whole L2, and one per work item made the kernel 3.5x
            // slower.  Instead the partials are the only data exchanged and they move with agent-scope accesses on
            // both sides (sc1: the stores write through, the loads bypass the CU's L1): every storing wave drains its
            // stores (vmcnt 0), the workgroup's barrier, then ONE lane counts the arrival with an agent-scope atomic
            // add; the workgroup whose add came last takes one agent-scope acquire and loads behind a barrier
            // (MI355X_MICROARCH.md, Workgroup dispatch ... inter-workgroup visibility, Valid forms).
            typedef float v4f __attribute__((ext_vector_type(4)));
            {
                const v4f o0 = {r00, g00, b00, T00}, o1 = {r10, g10, b10, T10}, o2 = {r01, g01, b01, T01}, o3 = {r11, g11, b11, T11};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1\n\t"
                             "global_store_dwordx4 %0, %3, off offset:2048 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:3072 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             :: "v"(p), "v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory");
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                const bool last = __hip_atomic_fetch_add(&bin_done[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nseg - 1u;
                if (last) {
                    // one agent-scope acquire on the folding CU (buffer_inv sc1: drops this CU's L1, about 1.7 us, once
                    // per multi-segment bin) in front of the barrier the other waves load behind.  The loads below are
                    // sc1 and bypass the L1 by themselves; the acquire makes the hand-off the documented form
                    // (MI355X_MICROARCH.md, Valid forms) rather than one that rests on that alone.
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    s_done = 0;   // (a second pass starts with every tile live again)
                }
                s_last = last ? 1u : 0u;
            }
            __syncthreads();
            if (!s_last) break;
            // fold, front to back, into the accumulators: C = C0 + T0*C1 + T0*T1*C2 + ..., T = T0*T1*...
            r00 = r10 = r01 = r11 = 0.f; g00 = g10 = g01 = g11 = 0.f; b00 = b10 = b01 = b11 = 0.f;
            T00 = T10 = T01 = T11 = 1.f;
            for (uint32_t k = 0; k < nseg; k++) {
                v4f v0, v1, v2, v3;
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0 + (size_t)k * BIN_PIXELS) : "memory");
                r00 = __builtin_fmaf(T00, v0.x, r00); g00 = __builtin_fmaf(T00, v0.y, g00); b00 = __builtin_fmaf(T00, v0.z, b00); T00 = T00 * v0.w;
                r10 = __builtin_fmaf(T10, v1.x, r10); g10 = __builtin_fmaf(T10, v1.y, g10); b10 = __builtin_fmaf(T10, v1.z, b10); T10 = T10 * v1.w;
                r01 = __builtin_fmaf(T01, v2.x, r01); g01 = __builtin_fmaf(T01, v2.y, g01); b01 = __builtin_fmaf(T01, v2.z, b01); T01 = T01 * v2.w;
                r11 = __builtin_fmaf(T11, v3.x, r11); g11 = __builtin_fmaf(T11, v3.y, g11); b11 = __builtin_fmaf(T11, v3.z, b11); T11 = T11 * v3.w;
            }
            if (tail_begin >= bin_end) { write_fb = true; break; }
            // ---- behind the window: this workgroup goes on with the bin's remaining entries under the folded
            //      transmittance, so the saturation test sees the true state of every pixel (from the first 64 entries on:
            //      a window that already saturated the bin costs one staged chunk) ----
            begin = tail_begin; end = bin_end;
            final_pass = true;
            sat_item = saturate != 0u;
            sat_from = 0u;
            alive = alive0;
            done = alive0 == 0u;
            if (done && lane == 0) atomicAdd(&s_done, 1u);
        }
        }   // passes
#ifdef GSR_BLEND_STAMPS
        {
            const unsigned int d = (unsigned int)__builtin_amdgcn_s_memrealtime() - a_item_t0;
            if (d > a_max_dur) { a_max_dur = d; a_max_len = a_item_len + (final_pass ? end - begin : 0u); a_max_vis = a_entries - a_item_vis0; a_max_bin = a_item_bin; }
        }
#endif
        if (write_fb) {
            // ---- write the tile, premultiplied RGBA, alpha = 1 - T ----
            const int x0 = X0 + lx, x1 = x0 + 8, y0 = Y0 + ly, y1 = y0 + 8;
            if (y0 < g.H) {
                if (x0 < g.W) fb[(size_t)y0 * g.W + x0] = make_float4(r00, g00, b00, 1.0f - T00);
                if (x1 < g.W && !tile_item) fb[(size_t)y0 * g.W + x1] = make_float4(r10, g10, b10, 1.0f - T10);
            }
            if (y1 < g.H && !tile_item) {
                if (x0 < g.W) fb[(size_t)y1 * g.W + x0] = make_float4(r01, g01, b01, 1.0f - T01);
                if (x1 < g.W) fb[(size_t)y1 * g.W + x1] = make_float4(r11, g11, b11, 1.0f - T11);
            }
        }
    }
#ifdef GSR_BLEND_STAMPS
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned int* o = g_blend_stamps + ((size_t)blockIdx.x * 4 + wave) * 16;
        o[0] = (unsigned int)__builtin_readcyclecounter() - t_kernel0; o[1] = a_waitA; o[2] = a_stage; o[3] = a_waitB; o[4] = a_comp;
        o[5] = t_real0; o[6] = (unsigned int)__builtin_amdgcn_s_memrealtime(); o[7] = a_entries;
        o[8] = a_items; o[9] = a_last_start; o[10] = a_max_dur; o[11] = a_max_len; o[12] = a_max_vis; o[13] = a_max_bin;
#ifdef GSR_BLEND_COUNT_QUADS
        o[1] = a_quads; o[2] = a_empty; o[3] = a_cov;
#else
        (void)a_quads; (void)a_empty; (void)a_cov;
#endif
    }
#endif
}

#ifdef GSR_BLEND_STAMPS
extern "C" int gsr_debug_blend_stamps(unsigned int* out /* 4096*4*16 */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_stamps), sizeof g_blend_stamps) == hipSuccess ? 0 : -1;
}
#endif

// Fold the per-segment partials of every multi-segment bin, front to back -- the stand-alone form (BlendBuffers::bin_done
// null); by default the fold runs inside k_blend and this kernel is not launched.  A thread folds its four
// pixels as four independent chains and the segment loop is unrolled, so 16 loads are in flight per
// thread: the kernel is a latency-bound read of the partials (85 MB on C3 with 512-entry segments).
__global__ __launch_bounds__(BLEND_THREADS) void k_combine(const uint32_t* __restrict__ seg_start,
                                                           const float4* __restrict__ partial, float4* __restrict__ fb,
                                                           BinGrid g)
{
    const int nbxb = g.bx_hi - g.bx_lo;
    const int bin = blockIdx.x;
    const uint32_t s0 = seg_start[bin], nseg = seg_start[bin + 1] - s0;
    if (nseg <= 1) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int by = bin / nbxb, bxl = bin - by * nbxb;
    const int X0 = (g.bx_lo + bxl) * BIN_PX + (wave & 1) * TILE, Y0 = by * BIN_PX + (wave >> 1) * TILE;
    const int lx = lane & 7, ly = lane >> 3;
    const float4* p = partial + (size_t)s0 * BIN_PIXELS + wave * (TILE * TILE) + lane;
    float r[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, T[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll 4
    for (uint32_t k = 0; k < nseg; k++) {
        float4 v[4];
#pragma unroll
        for (int slot = 0; slot < 4; slot++) v[slot] = p[(size_t)k * BIN_PIXELS + slot * 64];
#pragma unroll
        for (int slot = 0; slot < 4; slot++) {
            r[slot] = __builtin_fmaf(T[slot], v[slot].x, r[slot]);
            gg[slot] = __builtin_fmaf(T[slot], v[slot].y, gg[slot]);
            b[slot] = __builtin_fmaf(T[slot], v[slot].z, b[slot]);
            T[slot] = T[slot] * v[slot].w;
        }
    }
#pragma unroll
    for (int slot = 0; slot < 4; slot++) {
        const int x = X0 + lx + 8 * (slot & 1), y = Y0 + ly + 8 * (slot >> 1);
        if (x < g.W && y < g.H) fb[(size_t)y * g.W + x] = make_float4(r[slot], gg[slot], b[slot], 1.0f - T[slot]);
    }
}

void launch_blend(const BlendBuffers& b, const BinGrid& g, float early_out_eps, hipStream_t s, hipEvent_t between)
{
    const int nbins = (g.bx_hi - g.bx_lo) * g.nby;
    if (nbins <= 0) return;
    hipLaunchKernelGGL(k_blend, dim3(b.grid), dim3(BLEND_THREADS), 0, s, b.items, b.seg_start, b.bin_start, b.list, b.rec,
                       b.shcol, b.fb, b.partial, b.queue, g, early_out_eps, b.seg_len_dev, b.capacity, b.nsplats, b.bin_done, b.saturate,
                       b.prio[0], b.prio[1], b.prio[2]);
    if (between) (void)hipEventRecord(between, s);
    if (b.seg_len < 0x40000000u && !b.bin_done)
        hipLaunchKernelGGL(k_combine, dim3(nbins), dim3(BLEND_THREADS), 0, s, b.seg_start, (const float4*)b.partial, b.fb, g);
}

__global__ void k_clear_fb(float4* fb, uint32_t npix)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npix) fb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

void launch_clear_fb(float4* fb, int32_t W, int32_t H, hipStream_t s)
{
    const uint32_t npix = (uint32_t)W * (uint32_t)H;
    if (!npix) return;
    hipLaunchKernelGGL(k_clear_fb, dim3((npix + 255) / 256), dim3(256), 0, s, fb, npix);
}

__device__ __forceinline__ uint32_t to_rgba8(float4 v)
{
    auto q = [](float x) -> uint32_t {
        x = fminf(fmaxf(x, 0.0f), 1.0f);
        return (uint32_t)(x * 255.0f + 0.5f);
    };
    return q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
}

// round(clamp(x, 0, 1) * 255) per channel (Appendix A of SURVEY.md: output framebuffer contract)
__global__ void k_to_rgba8(const float4* __restrict__ fb, uint32_t* __restrict__ out, uint32_t npix)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    out[i] = to_rgba8(fb[i]);
}

// Multi-GPU exchange, sender side: the band's columns [x0, x1) as RGBA8 straight into the all-gather slab
// ([H][slab_w] pixels), one pass over the band instead of a whole-frame conversion plus a strided copy.
__global__ void k_pack_band_rgba8(const float4* __restrict__ fb, uint32_t* __restrict__ slab, int W, int H, int x0, int x1,
                                  int slab_w)
{
    const int x = x0 + blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= x1) return;
    slab[(size_t)y * slab_w + (x - x0)] = to_rgba8(fb[(size_t)y * W + x]);
}

void launch_pack_band_rgba8(const float4* fb, uint32_t* slab, int W, int H, int x0, int x1, int slab_w, hipStream_t s)
{
    if (x1 <= x0 || H <= 0) return;
    hipLaunchKernelGGL(k_pack_band_rgba8, dim3((x1 - x0 + 255) / 256, H), dim3(256), 0, s, fb, slab, W, H, x0, x1, slab_w);
}

// Receiver side: the gathered slabs [world][H][slab_w] -> one row-major [H][W] image.
__global__ void k_unpack_slabs_rgba8(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ image, int W, int H,
                                     int slab_w, int world, SlabEdges e)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    uint32_t v = 0;
    for (int q = 0; q < world; q++)
        if (x >= e.x0[q] && x < e.x1[q]) v = gathered[((size_t)q * H + y) * slab_w + (x - e.x0[q])];
    image[(size_t)y * W + x] = v;
}

void launch_unpack_slabs_rgba8(const uint32_t* gathered, uint32_t* image, int W, int H, int slab_w, int world,
                               const SlabEdges& e, hipStream_t s)
{
    if (W <= 0 || H <= 0) return;
    hipLaunchKernelGGL(k_unpack_slabs_rgba8, dim3((W + 255) / 256, H), dim3(256), 0, s, gathered, image, W, H, slab_w, world, e);
}

void launch_to_rgba8(const float4* fb, uint32_t* out, uint32_t npix, hipStream_t s)
{
    if (!npix) return;
    hipLaunchKernelGGL(k_to_rgba8, dim3((npix + 255) / 256), dim3(256), 0, s, fb, out, npix);
}

}  // namespace gsr
