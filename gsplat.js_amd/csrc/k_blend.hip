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

GSR_BOUNDS_DECL(blend)   // sites: 0 work item's bin, 1 its segment, 2 list position, 3 splat index in the list, 4 item range inside the bin
constexpr int BLEND_THREADS = 256;
constexpr int CHUNK = BLEND_THREADS;
constexpr int BIN_PIXELS = BIN_PX * BIN_PX;
[[maybe_unused]] constexpr float LOG2E = 1.4426950408889634f;   // (0x3fb8aa3b; the assembly walk carries -log2(e) as the literal 0xbfb8aa3b)
constexpr uint32_t SAT_FROM = CHUNK;   // entries of a long work item before its first saturation test

#ifdef GSR_BLEND_STAMPS
// Diagnostic build only (scripts/build_exp.sh stamps "-DGSR_BLEND_STAMPS", read by scripts/blend_stamps.py): where a
// compositor wave spends its cycles and when it starts and ends.  Per wave: [0] lifetime, [1] barrier at chunk start,
// [2] staging, [3] barrier after staging, [4] composite loop (shader cycles), [5] start, [6] end (s_memrealtime,
// 100 MHz, common to all XCDs), [7] entry visits, [8] items, [9] start of the last item, [10] longest item (ticks),
// [11] its entries, [12] its entry visits, [13] its bin.  Never compiled into the shipped library.
__device__ unsigned int g_blend_stamps[4096 * 4 * 16];
// per bin (whole-bin work items): [0] entries, [1] entries staged before the item ended (its saturation depth, or
// all), [2..5] entry visits of the four waves, [6] the item's duration (s_memrealtime ticks, 10 ns), [7] its start tick
__device__ unsigned int g_bin_info[16384 * 8];
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

// (Fetching the next entry's record before the current entry's arithmetic -- two register sets, loop unrolled by two --
//  was measured 24 % SLOWER at 7 and at 6 waves: the compiler's s_waitcnt placement in the rotated loop waits for the
//  new reads as well, and the extra scalar control costs more than the hidden LDS latency; the other waves of the
//  SIMD already cover that latency.)
//
// SUB = 2 ("two waves per tile", one frame at a time): the pole of the kernel is one wave's serial walk over its tile's
// entries -- a wave ALONE on its SIMD needs ~560 cycles per entry visit (a 256-workgroup grid, one wave per SIMD: 503 us for
// the 1.93 M visits of a C3 frame), so a bin whose tiles take 600 visits runs 160-200 us whatever else the chip does (wave
// priorities change nothing, measured).  With 8 waves per workgroup, waves 0-3 (part 0) walk the first half of every
// 256-entry chunk under the running transmittance and waves 4-7 (part 1) the second half from (colour 0, transmittance
// 1); "under" is associative, so part 1's chunk partial goes through LDS and part 0 folds it at the chunk boundary
// (C += T*C', T *= T'), where it also runs the saturation test and hands the live quadrants to part 1.
// FUSED: the workgroup that delivers a bin's last segment folds the bin (the shipped form); !FUSED: the partials are left for
// the separate k_combine launch (GSR_FUSE_COMBINE=0: the reference the fused fold is tested against bit for bit).
typedef float v2f __attribute__((ext_vector_type(2)));
#define r00 rg00.x
#define g00 rg00.y
#define r10 rg10.x
#define g10 rg10.y
#define r01 rg01.x
#define g01 rg01.y
#define r11 rg11.x
#define g11 rg11.y

#if defined(GSR_BLEND_COUNT_QUADS) && !defined(GSR_CPP_WALK)
#define GSR_CPP_WALK   // (the quadrant counters live in the C++ form of the walk)
#endif
#ifndef GSR_CPP_WALK
// The walk over a step's entries -- the innermost loop of the renderer -- as one block of assembly (round 4).
// The C++ form of the same loop is below (GSR_CPP_WALK: the readable statement of what this does, and the build the assembly is
// tested against bit for bit: tests/test_gpu_parity.py::test_assembly_walk_equals_the_cpp_loop).  What the assembly changes is
// the scalar stream around the arithmetic, not the arithmetic: per entry s_ff1 / s_bitset0 for the walk instead of ctz + shift +
// andn2; per quadrant s_bitcmp1 + ONE branch instead of a 64-bit and + compare + branch and, behind the coverage compare,
// s_and_saveexec + s_cbranch_execz -- v_cmpx writes the coverage straight into exec (12.9 % of the visited quadrants have no
// covered pixel: their seven masked instructions cost less than a branch in every quadrant) and one s_mov restores it.  ~51
// instead of ~61 instructions per entry visit, the ten fewer all scalar: the compositor shares a CU's one scalar unit among up to
// 28 waves (SQ_INSTS_SALU / SQ_INSTS_VALU was 0.66).  Same fused multiply-adds in the same operand order: no bit of any image
// changes.  Measured: C3 three frames in flight 5480 -> 5690 frames/s, C4 k_blend 405 -> 393 us, C3 k_blend2 144.1 -> 141.4 us.
// Temporaries live in fixed registers, because inline asm cannot name one register of the tuple a ds_read_b128 fills:
//   v40 LDS address | v41 v42 row terms | v[44:47] ux uy ncu wx | v[48:51] wy ncw la blue | v[52:53] red green
//   v54 v55 per-pixel temporaries (v[54:55] also the packed colour FMA's broadcast source)
#define GSR_ASM_QUAD(N, BQ, PX, T, RG, B)                                                           \
    "s_bitcmp1_b64 %[" BQ "], %[j]\n\t"                                                             \
    "s_cbranch_scc0 " N "f\n\t"                                                                     \
    "v_fma_f32 v54, v44, %[" PX "], v41\n\t"       /* vPosition.x = ux*px + (uy*py - dot(u,c)) */     \
    "v_fma_f32 v55, v47, %[" PX "], v42\n\t"       /* vPosition.y */                                  \
    "v_mul_f32 v54, v54, v54\n\t"                                                                   \
    "v_fmac_f32 v54, v55, v55\n\t"                 /* q = |vPosition|^2 */                            \
    "v_cmpx_ge_f32 4.0, v54\n\t"                   /* frag.glsl.ts:15: discard q > 4 */              \
    "v_fmamk_f32 v54, v54, 0xbfb8aa3b, v50\n\t"    /* -q log2(e) + log2(opacity) */                  \
    "v_exp_f32 v54, v54\n\t"                                                                        \
    "s_nop 0\n\t"                                  /* (a transcendental's result: one wait state) */ \
    "v_mul_f32 v54, %[" T "], v54\n\t"             /* w = T * B */                                   \
    "v_sub_f32 %[" T "], %[" T "], v54\n\t"        /* T -= w */                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                      \
    "v_pk_fma_f32 %[" RG "], v[54:55], v[52:53], %[" RG "] op_sel_hi:[0,1,1]\n\t"   /* (r, g) += w * (cr, cg) */ \
    "v_fmac_f32 %[" B "], v54, v51\n\t"            /* b += w * cb */                                 \
    "s_mov_b64 exec, %[ex]\n"                                                                       \
    N ":\n\t"
#define GSR_ASM_WALK_STEP()                                                                                              \
    if (bal) {                                                                                                           \
        uint32_t j_, a_;                                                                                                 \
        uint64_t ex_;                                                                                                    \
        const uint32_t base_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)&s_rec[0][c0]);        \
        asm volatile(                                                                                                    \
            "s_mov_b64 %[ex], exec\n"                                                                                     \
            "1:\n\t"                                                                                                      \
            "s_ff1_i32_b64 %[j], %[bal]\n\t"       /* the next entry of the tile: lowest set bit */                       \
            "s_bitset0_b64 %[bal], %[j]\n\t"                                                                              \
            "s_lshl_b32 %[a], %[j], 4\n\t"                                                                                \
            "s_add_u32 %[a], %[a], %[base]\n\t"                                                                           \
            "v_mov_b32 v40, %[a]\n\t"                                                                                     \
            "ds_read_b128 v[44:47], v40\n\t"                                                                              \
            "ds_read_b128 v[48:51], v40 offset:4096\n\t"                                                                  \
            "ds_read_b64 v[52:53], v40 offset:8192\n\t"                                                                   \
            "s_waitcnt lgkmcnt(1)\n\t"                                                                                    \
            "v_fma_f32 v41, v45, %[py0], v46\n\t"  /* the row terms, shared by the two quadrants of a row */              \
            "v_fma_f32 v42, v48, %[py0], v49\n\t"                                                                         \
            GSR_ASM_QUAD("2", "bq0", "px0", "T00", "rg00", "b00")                                                        \
            GSR_ASM_QUAD("3", "bq1", "px1", "T10", "rg10", "b10")                                                        \
            "v_fma_f32 v41, v45, %[py1], v46\n\t"                                                                         \
            "v_fma_f32 v42, v48, %[py1], v49\n\t"                                                                         \
            GSR_ASM_QUAD("4", "bq2", "px0", "T01", "rg01", "b01")                                                        \
            GSR_ASM_QUAD("5", "bq3", "px1", "T11", "rg11", "b11")                                                        \
            "s_cmp_lg_u64 %[bal], 0\n\t"                                                                                  \
            "s_cbranch_scc1 1b\n\t"                                                                                       \
            "s_waitcnt lgkmcnt(0)"                                                                                       \
            : [bal] "+s"(bal), [j] "=&s"(j_), [a] "=&s"(a_), [ex] "=&s"(ex_), [T00] "+v"(T00), [T10] "+v"(T10), [T01] "+v"(T01), \
              [T11] "+v"(T11), [rg00] "+v"(rg00), [rg10] "+v"(rg10), [rg01] "+v"(rg01), [rg11] "+v"(rg11), [b00] "+v"(b00),      \
              [b10] "+v"(b10), [b01] "+v"(b01), [b11] "+v"(b11)                                                          \
            : [bq0] "s"(bq0), [bq1] "s"(bq1), [bq2] "s"(bq2), [bq3] "s"(bq3), [base] "s"(base_), [px0] "v"(pxf0),        \
              [px1] "v"(pxf1), [py0] "v"(pyf0), [py1] "v"(pyf1)                                                          \
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",   \
              "v55", "vcc", "scc", "memory");                                                                            \
    }
#endif

template <int SUB, bool FUSED>
__device__ __forceinline__ void blend_body(const uint32_t* __restrict__ items,
                                                         const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ bin_start,
                                                         const uint32_t* __restrict__ list,
                                                         const Record* __restrict__ rec,
                                                         const float4* __restrict__ shcol, float4* __restrict__ fb,
                                                         float4* __restrict__ partial, uint32_t* __restrict__ queue,
                                                         BinGrid g, float eps, const uint32_t* __restrict__ seg_len_dev, uint32_t capacity,
                                                         uint32_t nsplats, unsigned long long* __restrict__ bin_mask, uint32_t saturate)
{
    // [0]: ux, uy, -dot(u, c - bin origin), wx   [1]: wy, -dot(w, c - bin origin), log2(opacity), blue   [2]: red, green
    __shared__ float4 s_rec[3][CHUNK];   // one address register serves the three reads of an entry
    __shared__ uint32_t s_mask[CHUNK];
    __shared__ uint32_t s_done;
    __shared__ uint32_t s_item;
    __shared__ uint32_t s_last;
    __shared__ float s_part[SUB == 2 ? 4 * 16 * WAVE : 1];   // part 1's chunk partial: [tile][4 pixels x (r, g, b, T)][lane]
    __shared__ uint32_t s_alive[4];                           // part 0's live quadrants of each tile, for part 1

    const int nbxb = g.bx_hi - g.bx_lo;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, part = SUB == 2 ? (int)(threadIdx.x >> 8) : 0;   // (tile, half of a chunk)
    const int lx = lane & 7, ly = lane >> 3;
    const uint32_t total_items = seg_len_dev[1];   // the frame's work items (k_bin_finalize)

    // Work items come from one device-wide queue (items are ordered heaviest first), so a workgroup
    // that drew light items simply draws more: no static assignment, no long pole.
    // The first item of a workgroup is its own index (the queue starts at gridDim.x, k_bin_finalize sets it): a
    // kernel start with ~2000 workgroups drawing from one counter serialises ~2000 same-address atomics.
#ifdef GSR_BLEND_STAMPS
    unsigned int a_waitA = 0, a_stage = 0, a_waitB = 0, a_comp = 0, a_entries = 0, a_staged = 0;
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
        const uint4 it4 = reinterpret_cast<const uint4*>(items)[qi];   // (bin | segment << 16, first entry, end, partial slot | segments << 25)
        const uint32_t it = it4.x;
        const int bin = (int)(it & 0xffffu);
#ifdef GSR_BLEND_STAMPS
        a_items++; a_item_t0 = (unsigned int)__builtin_amdgcn_s_memrealtime(); a_last_start = a_item_t0; a_item_vis0 = a_entries; a_item_bin = (unsigned int)bin; a_staged = 0;
#endif
        const uint32_t seg = it >> 16;
        const int tile = wave;
        const uint32_t nseg = it4.w >> 25, slot0 = it4.w & 0x1ffffffu;
        GSR_BOUND(blend, 0, bin, nbxb * g.nby);
        GSR_BOUND(blend, 1, seg, nseg);
        const int by = bin / nbxb, bxl = bin - by * nbxb;
        const int binX0 = (g.bx_lo + bxl) * BIN_PX, binY0 = by * BIN_PX;
        const int ox = (tile & 1) * TILE, oy = (tile >> 1) * TILE;
        const int X0 = binX0 + ox, Y0 = binY0 + oy;
        const int mask_shift = wave * 4;   // which bits of an entry's 16-bit quadrant mask are mine: my tile's nibble
        // pixel centres relative to the centre of the bin's first pixel: small exact integers
        const float pxf0 = (float)(ox + lx), pxf1 = pxf0 + 8.0f;
        const float pyf0 = (float)(oy + ly), pyf1 = pyf0 + 8.0f;
        const float bx0c = (float)binX0 + 0.5f, by0c = (float)binY0 + 0.5f;

        float T00 = 1.f, T10 = 1.f, T01 = 1.f, T11 = 1.f;  // Tij: pixel (x+8i, y+8j); 1 - alpha
        v2f rg00 = {0.f, 0.f}, rg10 = {0.f, 0.f}, rg01 = {0.f, 0.f}, rg11 = {0.f, 0.f};   // (red, green): one register pair per pixel
        float b00 = 0.f, b10 = 0.f, b01 = 0.f, b11 = 0.f;

        const uint32_t end = min(it4.z, capacity), begin = min(it4.y, end);   // (a bin has at most 64 segments: the last takes the rest)
        // only long items test for saturation while they run, and only where the transmittance is the true one: whole bins
        // and a bin's first segment (later segments start from 1)
        const bool sat_item = saturate != 0u && seg == 0u && end - begin > 2u * CHUNK;
        GSR_BOUND(blend, 4, it4.z, (unsigned long long)capacity + 1ull);
        GSR_BOUND(blend, 4, it4.y, (unsigned long long)it4.z + 1ull);
        // quadrants of mine that can still change (wave-uniform); see the saturation test below.  Quadrants that lie outside
        // the image (the last bin row of 1080p: rows 1080..1087) never could: their pixels are not stored.
        uint32_t alive0 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (X0 + 8 * (q & 1) < g.W && Y0 + 8 * (q >> 1) < g.H) alive0 |= 1u << q;
        uint32_t alive = alive0;
        bool done = alive0 == 0u;
        if (done && lane == 0 && part == 0) atomicAdd(&s_done, 1u);
        if (SUB == 2 && part == 0 && lane == 0) s_alive[wave] = alive0;   // (read by part 1 behind the first chunk's two barriers)
#ifdef GSR_BLEND_STAMPS
        a_item_len = end - begin;
#endif
        // fold part 1's chunk partial (LDS) behind part 0's state: C += T*C', T *= T'
#define GSR_FOLD_PART()                                                                                                  \
    {                                                                                                                    \
        const float* sp_ = s_part + wave * (16 * WAVE) + lane;                                                           \
        r00 = __builtin_fmaf(T00, sp_[0 * WAVE], r00); g00 = __builtin_fmaf(T00, sp_[1 * WAVE], g00); b00 = __builtin_fmaf(T00, sp_[2 * WAVE], b00); T00 = T00 * sp_[3 * WAVE];       \
        r10 = __builtin_fmaf(T10, sp_[4 * WAVE], r10); g10 = __builtin_fmaf(T10, sp_[5 * WAVE], g10); b10 = __builtin_fmaf(T10, sp_[6 * WAVE], b10); T10 = T10 * sp_[7 * WAVE];       \
        r01 = __builtin_fmaf(T01, sp_[8 * WAVE], r01); g01 = __builtin_fmaf(T01, sp_[9 * WAVE], g01); b01 = __builtin_fmaf(T01, sp_[10 * WAVE], b01); T01 = T01 * sp_[11 * WAVE];    \
        r11 = __builtin_fmaf(T11, sp_[12 * WAVE], r11); g11 = __builtin_fmaf(T11, sp_[13 * WAVE], g11); b11 = __builtin_fmaf(T11, sp_[14 * WAVE], b11); T11 = T11 * sp_[15 * WAVE];  \
    }
        // ---- saturation that changes no bit ----
        // A pixel whose transmittance is below 2^-27 of its smallest colour channel is finished: every later
        // weight is w <= T (opacity <= 1, exp <= 1), colours are <= 1, so w*c is under half an ulp of each channel
        // and fma(w, c, C) returns C; its alpha is 1 - T = 1.0f for any T that small.  The segment's own T
        // would still shrink, but only ever multiplies later segments' colours (the fold), whose terms are then
        // under half an ulp of the folded colour as well: the image is bit-identical to compositing every entry
        // (test_saturated_quadrants_are_skipped_without_changing_a_bit).  A quadrant whose 64 pixels are all
        // finished drops out of `alive`; a tile with no live quadrant is done, a bin with no live tile ends its
        // work item (s_done).  Pixels with a zero channel finish only at T == 0.
#define GSR_FINISHED(BIT, T, R, G, B_)                                                                          \
    if ((alive & (BIT)) && __ballot((T) >= NEAR) == 0ull &&                                                      \
        __ballot(!((T) < K * fminf((R), fminf((G), (B_))) || (T) == 0.0f)) == 0ull)                              \
        alive &= ~(BIT);
#define GSR_SAT_TEST()                                                                                           \
    {                                                                                                            \
        constexpr float K = 0x1p-27f, NEAR = 1e-6f;   /* nothing above NEAR can pass the test: a cheap filter first */ \
        GSR_FINISHED(1u, T00, r00, g00, b00)                                                                     \
        GSR_FINISHED(2u, T10, r10, g10, b10)                                                                     \
        GSR_FINISHED(4u, T01, r01, g01, b01)                                                                     \
        GSR_FINISHED(8u, T11, r11, g11, b11)                                                                     \
    }
        bool pending = false;   // SUB == 2: part 1's partial of the last composited chunk is in LDS, not folded yet (uniform)
        for (uint32_t base = begin; base < end; base += CHUNK) {
            STAMP(t_c0);
            __syncthreads();  // previous chunk fully consumed (and s_done visible)
            STAMP(t_cA);
            if (SUB == 2 && pending) {
                // part 0: the chunk just composited ends with part 1's half; then the tests that need the true state
                if (part == 0 && !done) {
                    GSR_FOLD_PART()
                    if (eps > 0.0f) {
                        const float tmax = fmaxf(fmaxf(T00, T10), fmaxf(T01, T11));
                        if (__ballot(tmax >= eps) == 0ull) alive = 0u;
                    }
                    if (sat_item && base - begin >= SAT_FROM) GSR_SAT_TEST()
                    if (alive == 0u) {
                        done = true;
                        if (lane == 0) atomicAdd(&s_done, 1u);
                    }
                    if (lane == 0) s_alive[wave] = alive;
                }
                pending = false;
            }
            if (SUB == 1 && s_done == 4u) break;  // every tile of the bin is saturated
#ifdef GSR_BLEND_STAMPS
            a_staged = min(base + CHUNK, end) - begin;
#endif
            // ---- stage one entry per thread: record, unpacked colour, and a 16-bit mask of the bin's
            //      8x8-pixel quadrants the splat can touch (bit = tile*4 + quadrant).  The mask is a
            //      conservative cull only: a quadrant is dropped when it lies outside the oriented box
            //      |vPosition.x|,|vPosition.y| <= 2 (separating axes u, w) or farther from the centre than the
            //      longer semi-axis.  Pixels that pass are still tested exactly (q <= 4) below. ----
            // (two waves per tile: two threads per entry, thread t and t + 256, each testing two of the four quadrant rows
            //  -- one byte of the mask -- and writing its share of the record)
            const uint32_t slot = threadIdx.x & (CHUNK - 1), half = SUB == 2 ? threadIdx.x >> 8 : 0u;
            const uint32_t e = base + slot;
            uint32_t mask = 0;
            if (e < end) {
                GSR_BOUND(blend, 2, e, capacity);
                GSR_BOUND(blend, 3, list[e], nsplats);
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
                for (int rr = 0; rr < 4 / SUB; rr++) {
                    // quadrant row r of the bin: all four (one wave per tile), or rows 2*half, 2*half + 1 (two threads per entry)
                    const int r = SUB == 2 ? (int)(2u * half) + rr : rr;
                    const float dc = (float)(binY0 + 8 * r + 4) - ra.y;
                    const float ur = ra.w * dc, wr = rb.y * dc;
                    const float dd = fmaxf(fabsf(dc) - 3.5f, 0.0f);
                    const float d2 = dd * dd;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const bool hit = fabsf(ucol[k] + ur) <= eu && fabsf(wcol[k] + wr) <= ew && (dcol[k] + d2) * minlen2 <= 4.002f;
                        // quadrant (k, r) of the bin -> tile (k>>1, r>>1), quadrant (k&1, r&1): bit tile*4 + quadrant (SUB == 2: of
                        // this thread's byte, which holds the tile row r>>1 = half)
                        if (hit) mask |= 1u << (((((SUB == 2 ? 0 : rr >> 1) * 2) + (k >> 1)) << 2) + ((rr & 1) * 2 + (k & 1)));
                    }
                }
                const float cxr = ra.x - bx0c, cyr = ra.y - by0c;
                if (half == 0u) {
                    const uint32_t rgb8 = __float_as_uint(rb.w);
                    float cr = (float)(rgb8 & 0xffu) * (1.0f / 255.0f), cg = (float)((rgb8 >> 8) & 0xffu) * (1.0f / 255.0f),
                          cb = (float)((rgb8 >> 16) & 0xffu) * (1.0f / 255.0f);
                    if (rgb8 & RGB8_IN_SHCOL) {  // SH-coloured splat: the projection kernel evaluated its colour for this view
                        const float4 sc = shcol[i];
                        cr = sc.x; cg = sc.y; cb = sc.z;
                    }
                    const float ncw = -__builtin_fmaf(rb.y, cyr, rb.x * cxr);
                    s_rec[1][slot] = make_float4(rb.y, ncw, rb.z, cb);
                    *reinterpret_cast<float2*>(&s_rec[2][slot]) = make_float2(cr, cg);
                }
                if (SUB == 1 || half == 1u) {
                    const float ncu = -__builtin_fmaf(ra.w, cyr, ra.z * cxr);
                    s_rec[0][slot] = make_float4(ra.z, ra.w, ncu, rb.x);
                }
            }
            if (SUB == 2) reinterpret_cast<uint8_t*>(s_mask)[slot * 4u + half] = (uint8_t)mask;   // (bytes 2, 3 are never read)
            else s_mask[slot] = mask;
            STAMP(t_cS);
            __syncthreads();
            STAMP(t_cB);
            if (SUB == 2) {
                if (s_done == 4u) break;   // every tile of the bin is saturated (the chunk just staged is not needed)
                if (part == 1) alive = s_alive[wave];
            }

            if (!done) {
                const uint32_t cnt = min((uint32_t)CHUNK, end - base);
                // two waves per tile: part 0 takes the first half of the chunk's entries that touch the tile, part 1 the second
                // half -- by count, so that the two waves walk about equally long, and by the entries' masks ALONE: were the
                // live quadrants taken into account, the cut -- and with it the f32 association of pixels that are still live
                // -- would depend on what the saturation skip has dropped, which must not change a bit
                uint32_t hits_half = 0, hits_before = 0;
                if (SUB == 2) {
                    uint32_t h = 0;
#pragma unroll
                    for (uint32_t c0 = 0; c0 < CHUNK; c0 += WAVE) h += (uint32_t)__popcll(__ballot(((s_mask[c0 + lane] >> mask_shift) & alive0) != 0u));
                    hits_half = (h + 1u) >> 1;
                }
                for (uint32_t c0 = 0; c0 < cnt; c0 += WAVE) {
                    const uint32_t touch = (s_mask[c0 + lane] >> mask_shift) & alive0;
                    const uint32_t mine = touch & alive;  // entry (c0+lane) vs my live quadrants
                    uint64_t bal = __ballot(mine != 0u);
                    if (SUB == 2) {
                        const uint64_t all = __ballot(touch != 0u);
                        const uint32_t rank = hits_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(all >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)all, 0u));
                        hits_before += (uint32_t)__popcll(all);
                        bal = __ballot(mine != 0u && (rank < hits_half) == (part == 0));
                    }
#ifdef GSR_BLEND_STAMPS
                    a_entries += __popcll(bal);
#endif
#define GSR_QUAD(BIT, PX, UR, WR, T, R, G, B_)                                                     \
    if (GSR_QUAD_HIT(BIT)) {                                                                       \
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
#define GSR_ENTRY(RA, RB, RC)                                                                                  \
    {                                                                                                          \
        const float ux = RA.x, uy = RA.y, ncu = RA.z, wx = RA.w, wy = RB.x, ncw = RB.y, la = RB.z;             \
        const float cr = RC.x, cg = RC.y, cb = RB.w;                                                           \
        const float ur0 = __builtin_fmaf(uy, pyf0, ncu), ur1 = __builtin_fmaf(uy, pyf1, ncu);                  \
        const float wr0 = __builtin_fmaf(wy, pyf0, ncw), wr1 = __builtin_fmaf(wy, pyf1, ncw);                  \
        GSR_QUAD(1u, pxf0, ur0, wr0, T00, r00, g00, b00)                                                       \
        GSR_QUAD(2u, pxf1, ur0, wr0, T10, r10, g10, b10)                                                       \
        GSR_QUAD(4u, pxf0, ur1, wr1, T01, r01, g01, b01)                                                       \
        GSR_QUAD(8u, pxf1, ur1, wr1, T11, r11, g11, b11)                                                       \
    }
#define GSR_FETCH(RA, RB, RC, J)                                                     \
    RA = s_rec[0][c0 + (J)];                                                          \
    RB = s_rec[1][c0 + (J)];                                                          \
    RC = *reinterpret_cast<const float2*>(&s_rec[2][c0 + (J)]);
                    // Which of my quadrants an entry touches: four lane sets in scalar registers, tested against the entry's one
                    // bit (1 << j, which also clears it from the walk: s_andn2) -- instead of a v_readlane of the entry's mask,
                    // its hazard towards the scalar tests, and a three-instruction `bal &= bal - 1` (bit-identical; C4 k_blend
                    // 411 -> 395 us, three frames in flight on C3 +1.4 %).
                    const uint64_t bq0 = __ballot((mine & 1u) != 0u), bq1 = __ballot((mine & 2u) != 0u),
                                   bq2 = __ballot((mine & 4u) != 0u), bq3 = __ballot((mine & 8u) != 0u);
#define GSR_QUAD_HIT(BIT) ((((BIT) == 1u ? bq0 : (BIT) == 2u ? bq1 : (BIT) == 4u ? bq2 : bq3) >> j) & 1ull)
#ifndef GSR_CPP_WALK
                    GSR_ASM_WALK_STEP()
#else
                    while (bal) {
                        const int j = __builtin_ctzll(bal);
                        bal &= ~(1ull << j);
                        float4 ra, rb;
                        float2 rc;
                        GSR_FETCH(ra, rb, rc, j)
                        GSR_ENTRY(ra, rb, rc)
                    }
#endif
#undef GSR_QUAD_HIT
#undef GSR_FETCH
#undef GSR_ENTRY
#undef GSR_QUAD
                    if (SUB == 1 && eps > 0.0f) {
                        const float tmax = fmaxf(fmaxf(T00, T10), fmaxf(T01, T11));
                        if (__ballot(tmax >= eps) == 0ull) {
                            done = true;
                            if (lane == 0) atomicAdd(&s_done, 1u);
                            break;
                        }
                    }
                    // (one wave per tile: after every 64 entries of a long item.  Items of up to two chunks -- all a frame
                    //  that does not saturate has -- never run the test, which cost 3 % on C2, and nothing saturates within
                    //  an item's first SAT_FROM entries.  Two waves per tile: at the chunk boundaries, above.)
                    if (SUB == 1 && sat_item && base - begin + c0 >= SAT_FROM) {
                        GSR_SAT_TEST()
                        if (alive == 0u) {
                            done = true;
                            if (lane == 0) atomicAdd(&s_done, 1u);
                            break;
                        }
                    }
                }
            }
            if (SUB == 2) {
                if (part == 1) {   // my half of the chunk, from (0, 1): to LDS for part 0, and start over
                    float* sp = s_part + wave * (16 * WAVE) + lane;
                    sp[0 * WAVE] = r00; sp[1 * WAVE] = g00; sp[2 * WAVE] = b00; sp[3 * WAVE] = T00;
                    sp[4 * WAVE] = r10; sp[5 * WAVE] = g10; sp[6 * WAVE] = b10; sp[7 * WAVE] = T10;
                    sp[8 * WAVE] = r01; sp[9 * WAVE] = g01; sp[10 * WAVE] = b01; sp[11 * WAVE] = T01;
                    sp[12 * WAVE] = r11; sp[13 * WAVE] = g11; sp[14 * WAVE] = b11; sp[15 * WAVE] = T11;
                    r00 = r10 = r01 = r11 = 0.f; g00 = g10 = g01 = g11 = 0.f; b00 = b10 = b01 = b11 = 0.f;
                    T00 = T10 = T01 = T11 = 1.f;
                }
                pending = true;
            }
#ifdef GSR_BLEND_STAMPS
            {
                const unsigned int t_cC = (unsigned int)__builtin_readcyclecounter();
                a_waitA += t_cA - t_c0; a_stage += t_cS - t_cA; a_waitB += t_cB - t_cS; a_comp += t_cC - t_cB;
            }
#endif
        }

        if (SUB == 2) {
            __syncthreads();
            if (pending && part == 0 && !done) GSR_FOLD_PART()
        }
#undef GSR_SAT_TEST
#undef GSR_FINISHED
#undef GSR_FOLD_PART
#ifdef GSR_BLEND_STAMPS
        {
            const unsigned int d = (unsigned int)__builtin_amdgcn_s_memrealtime() - a_item_t0;
            if (d > a_max_dur) { a_max_dur = d; a_max_len = a_item_len; a_max_vis = a_entries - a_item_vis0; a_max_bin = a_item_bin; }
            if (nseg == 1u && bin < 16384 && lane == 0 && part == 0) {
                unsigned int* bi = g_bin_info + (size_t)bin * 8;
                bi[2 + wave] = a_entries - a_item_vis0;
                if (wave == 0) { bi[0] = end - begin; bi[1] = a_staged; bi[6] = d; bi[7] = a_item_t0; }
            }
        }
#endif
        bool write_fb = nseg == 1u;
        if (nseg > 1u) {
            // ---- partial (colour, transmittance) of this segment, slot-major: fully coalesced ----
            float4* p0 = partial + (size_t)slot0 * BIN_PIXELS + wave * (TILE * TILE) + lane;
            float4* p = p0 + (size_t)seg * BIN_PIXELS;
            if (!FUSED) {   // k_combine folds the bin after this kernel
                if (part == 0) {
                    p[0] = make_float4(r00, g00, b00, T00);
                    p[64] = make_float4(r10, g10, b10, T10);
                    p[128] = make_float4(r01, g01, b01, T01);
                    p[192] = make_float4(r11, g11, b11, T11);
                }
                continue;
            }
            // The workgroup that delivers a bin's LAST segment folds the bin itself (front to back, k_combine's fixed
            // order: which workgroup does it changes nothing in the result), so the fold runs beside the other
            // workgroups' arithmetic and the k_combine launch goes away.
            // Visibility between workgroups on different XCDs (one L2 each) WITHOUT a release fence: an agent-scope
            // release is buffer_wbl2, a write-back of the XCD's whole L2, and one per work item made the kernel 3.5x
            // slower.  Instead the partials are the only data exchanged and they move with agent-scope accesses on
            // both sides (sc1: the stores write through, the loads bypass the CU's L1): every storing wave drains its
            // stores (vmcnt 0), the workgroup's barrier, then ONE lane counts the arrival with an agent-scope atomic
            // (its segment's bit in the bin's mask); the workgroup whose arrival completes the mask takes one agent-scope
            // acquire and loads behind a barrier (MI355X_MICROARCH.md, Workgroup dispatch ... inter-workgroup visibility,
            // Valid forms).
            typedef float v4f __attribute__((ext_vector_type(4)));
            if (part == 0) {   // (the state is part 0's: part 1's halves have been folded into it)
                const v4f o0 = {r00, g00, b00, T00}, o1 = {r10, g10, b10, T10}, o2 = {r01, g01, b01, T01}, o3 = {r11, g11, b11, T11};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1\n\t"
                             "global_store_dwordx4 %0, %3, off offset:2048 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:3072 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             :: "v"(p), "v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory");
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned long long bit = 1ull << seg, all = nseg >= 64u ? ~0ull : (1ull << nseg) - 1ull;
                const unsigned long long now = __hip_atomic_fetch_or(&bin_mask[bin], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | bit;
                if (now == all) {
                    // one agent-scope acquire on the folding CU (buffer_inv sc1: drops this CU's L1, about 1.7 us, once
                    // per fold) in front of the barrier the other waves load behind.  The loads below are
                    // sc1 and bypass the L1 by themselves; the acquire makes the hand-off the documented form
                    // (MI355X_MICROARCH.md, Valid forms) rather than one that rests on that alone.
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                s_last = now == all ? 1u : 0u;
            }
            __syncthreads();
            if (!s_last) continue;
            // fold, front to back, into the accumulators: C = C0 + T0*C1 + T0*T1*C2 + ..., T = T0*T1*...
            r00 = r10 = r01 = r11 = 0.f; g00 = g10 = g01 = g11 = 0.f; b00 = b10 = b01 = b11 = 0.f;
            T00 = T10 = T01 = T11 = 1.f;
            for (uint32_t k = 0; k < (part == 0 ? nseg : 0u); k++) {
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
            write_fb = true;
        }
        if (write_fb && part == 0) {
            // ---- write the tile, premultiplied RGBA, alpha = 1 - T ----
            const int x0 = X0 + lx, x1 = x0 + 8, y0 = Y0 + ly, y1 = y0 + 8;
            if (y0 < g.H) {
                if (x0 < g.W) fb[(size_t)y0 * g.W + x0] = make_float4(r00, g00, b00, 1.0f - T00);
                if (x1 < g.W) fb[(size_t)y0 * g.W + x1] = make_float4(r10, g10, b10, 1.0f - T10);
            }
            if (y1 < g.H) {
                if (x0 < g.W) fb[(size_t)y1 * g.W + x0] = make_float4(r01, g01, b01, 1.0f - T01);
                if (x1 < g.W) fb[(size_t)y1 * g.W + x1] = make_float4(r11, g11, b11, 1.0f - T11);
            }
        }
    }
#ifdef GSR_BLEND_STAMPS
    if (lane == 0 && blockIdx.x < 4096 && part == 0) {
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

// 7 waves per SIMD: the kernel needs 74 VGPRs unconstrained (6 waves); capped at 72 it spills one register pair that is
// stored once per workgroup and reloaded once per work item, and the seventh wave hides more of the LDS/barrier
// waits (measured: k_blend -5.7 % alone; 8 waves = 64 VGPRs spills inside the loops and is no better).
#define GSR_BLEND_PARAMS                                                                                                  \
    const uint32_t *__restrict__ items, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ bin_start,   \
        const uint32_t *__restrict__ list, const Record *__restrict__ rec, const float4 *__restrict__ shcol,              \
        float4 *__restrict__ fb, float4 *__restrict__ partial, uint32_t *__restrict__ queue, BinGrid g, float eps,        \
        const uint32_t *__restrict__ seg_len_dev, uint32_t capacity, uint32_t nsplats,                                    \
        unsigned long long *__restrict__ bin_mask, uint32_t saturate
#define GSR_BLEND_ARGS items, seg_start, bin_start, list, rec, shcol, fb, partial, queue, g, eps, seg_len_dev, capacity, nsplats, bin_mask, saturate
__global__ __launch_bounds__(BLEND_THREADS) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_blend(GSR_BLEND_PARAMS)
{
    blend_body<1, true>(GSR_BLEND_ARGS);
}
// two waves per tile: 512-thread workgroups, three per CU (6 waves per SIMD, 80 registers)
__global__ __launch_bounds__(2 * BLEND_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_blend2(GSR_BLEND_PARAMS)
{
    blend_body<2, true>(GSR_BLEND_ARGS);
}
// the same two with the partials left to k_combine (GSR_FUSE_COMBINE=0; test reference)
__global__ __launch_bounds__(BLEND_THREADS) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_blend_unfused(GSR_BLEND_PARAMS)
{
    blend_body<1, false>(GSR_BLEND_ARGS);
}
__global__ __launch_bounds__(2 * BLEND_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_blend2_unfused(GSR_BLEND_PARAMS)
{
    blend_body<2, false>(GSR_BLEND_ARGS);
}
#undef GSR_BLEND_PARAMS
#undef GSR_BLEND_ARGS
#undef r00
#undef g00
#undef r10
#undef g10
#undef r01
#undef g01
#undef r11
#undef g11

#ifdef GSR_BLEND_STAMPS
extern "C" int gsr_debug_blend_stamps(unsigned int* out /* 4096*4*16 */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_stamps), sizeof g_blend_stamps) == hipSuccess ? 0 : -1;
}
extern "C" int gsr_debug_bin_info(unsigned int* out /* 16384*8 */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bin_info), sizeof g_bin_info) == hipSuccess ? 0 : -1;
}
#endif

// Fold the per-segment partials of every multi-segment bin, front to back -- the stand-alone form (BlendBuffers::bin_mask
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
#define GSR_LAUNCH_BLEND(K, THREADS)                                                                                      \
    hipLaunchKernelGGL(K, dim3(b.grid), dim3(THREADS), 0, s, b.items, b.seg_start, b.bin_start, b.list, b.rec, b.shcol, b.fb, \
                       b.partial, b.queue, g, early_out_eps, b.seg_len_dev, b.capacity, b.nsplats, b.bin_mask, b.saturate)
    if (b.sub >= 2 && b.bin_mask) GSR_LAUNCH_BLEND(k_blend2, 2 * BLEND_THREADS);
    else if (b.sub >= 2) GSR_LAUNCH_BLEND(k_blend2_unfused, 2 * BLEND_THREADS);
    else if (b.bin_mask) GSR_LAUNCH_BLEND(k_blend, BLEND_THREADS);
    else GSR_LAUNCH_BLEND(k_blend_unfused, BLEND_THREADS);
#undef GSR_LAUNCH_BLEND
    if (between) (void)hipEventRecord(between, s);
    if (b.seg_len < 0x40000000u && !b.bin_mask)
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
// `overflow` (may be null: the harness's own slabs carry no flag): the frame's overflow word.  A frame whose bin lists did not
// fit was not composited -- the band about to be packed is the PRECEDING image -- and the slab says so in SLAB_FLAG_WORDS
// words behind its pixels, which travel with it through the all-gather: every rank of the group then sees, for the same
// gathered frame, that one of its bands is stale (k_unpack_slabs_rgba8) and refuses it -- a group-wide decision, so that no
// rank redoes a collective alone.
__global__ void k_pack_band_rgba8(const float4* __restrict__ fb, uint32_t* __restrict__ slab, int W, int H, int x0, int x1,
                                  int slab_w, const uint32_t* __restrict__ overflow)
{
    if (overflow && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < SLAB_FLAG_WORDS)
        slab[(size_t)H * slab_w + threadIdx.x] = *overflow != 0u ? 1u : 0u;
    const int x = x0 + blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= x1) return;
    slab[(size_t)y * slab_w + (x - x0)] = to_rgba8(fb[(size_t)y * W + x]);
}

void launch_pack_band_rgba8(const float4* fb, uint32_t* slab, int W, int H, int x0, int x1, int slab_w, hipStream_t s, const uint32_t* overflow)
{
    if (x1 <= x0 || H <= 0) return;
    hipLaunchKernelGGL(k_pack_band_rgba8, dim3((x1 - x0 + 255) / 256, H), dim3(256), 0, s, fb, slab, W, H, x0, x1, slab_w, overflow);
}

// Receiver side: the gathered slabs [world][H][slab_w] -> one row-major [H][W] image.
// `stale` (may be null; then the slabs carry no flag words): out, one bit per rank whose band was not composited (see
// k_pack_band_rgba8) -- the same word on every rank of the group.
__global__ void k_unpack_slabs_rgba8(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ image, int W, int H,
                                     int slab_w, int world, SlabEdges e, uint32_t* __restrict__ stale)
{
    const size_t slab_words = (size_t)H * slab_w + (stale ? SLAB_FLAG_WORDS : 0);
    if (stale && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        uint32_t bits = 0;
        for (int q = 0; q < world; q++)
            if (gathered[(size_t)q * slab_words + (size_t)H * slab_w] != 0u) bits |= 1u << q;
        *stale = bits;
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    uint32_t v = 0;
    for (int q = 0; q < world; q++)
        if (x >= e.x0[q] && x < e.x1[q]) v = gathered[(size_t)q * slab_words + (size_t)y * slab_w + (x - e.x0[q])];
    image[(size_t)y * W + x] = v;
}

void launch_unpack_slabs_rgba8(const uint32_t* gathered, uint32_t* image, int W, int H, int slab_w, int world,
                               const SlabEdges& e, hipStream_t s, uint32_t* stale)
{
    if (W <= 0 || H <= 0) return;
    hipLaunchKernelGGL(k_unpack_slabs_rgba8, dim3((W + 255) / 256, H), dim3(256), 0, s, gathered, image, W, H, slab_w, world, e, stale);
}

void launch_to_rgba8(const float4* fb, uint32_t* out, uint32_t npix, hipStream_t s)
{
    if (!npix) return;
    hipLaunchKernelGGL(k_to_rgba8, dim3((npix + 255) / 256), dim3(256), 0, s, fb, out, npix);
}

}  // namespace gsr
