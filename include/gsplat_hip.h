/*
 * gsplat_hip.h -- C ABI of libgsplat_hip.so, the MI355X-native drop-in for the
 * per-frame hot path of Lanv1/gsplat.js (sort + project + composite).
 *
 * Plain C: opaque context, plain pointers and sizes, int return codes
 * (0 = ok, <0 = error, text via gsr_last_error).  No function throws, aborts
 * or keeps a caller's pointer after it returns (inputs are copied to the
 * device during the call).  A context is bound to one host thread at a time;
 * distinct contexts are independent.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   wasm `sort(...)`                 wasm/wasm.cpp:8-13, called at
 *                                    src/renderers/webgl/utils/Worker.ts:39
 *                                      -> gsplat_sort_host (same 7 arguments) / gsr_sort
 *   worker scene init                Worker.ts:23-34 (positions copied once per scene)
 *   + texImage2D(scene.data)         src/renderers/WebGLRenderer.ts:185-195
 *                                      -> gsr_set_scene
 *   Scene.setData / translate / rotate / scale / limitBox   src/core/Scene.ts:58-366
 *                                      -> gsr_set_scene_rows, gsr_scene_* (optional device-side versions)
 *   setShTextures + u_bandIndex      WebGLRenderer.ts:202-211,321-366
 *                                      -> gsr_set_scene_sh
 *   uniforms projection/view/focal/viewport + postMessage({viewProj})
 *                                    WebGLRenderer.ts:144-159,268-269,275
 *                                      -> gsr_set_camera, gsr_resize
 *   drawArraysInstanced + blend state WebGLRenderer.ts:279-290 with
 *   vertex.glsl.ts:130-231, frag.glsl.ts:13-21
 *                                      -> gsr_render
 *   worker.onmessage depthIndex      WebGLRenderer.ts:223-229
 *                                      -> gsr_read_depth_index
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_ERR_ARG (-1)      /* bad argument / call order */
#define GSR_ERR_HIP (-2)      /* a HIP runtime call failed */
#define GSR_ERR_NO_DEVICE (-3)/* no usable AMD GPU */
#define GSR_ERR_SCENE (-4)    /* scene buffers inconsistent */
#define GSR_ERR_OVERFLOW (-5) /* internal list capacity exceeded even after regrowth; or: asynchronous frames were lost (gsr_sync) */
#define GSR_ERR_COMM (-6)     /* RCCL is unavailable or a collective call failed */

typedef struct gsr_ctx gsr_ctx;

typedef struct gsr_options {
    int32_t device;         /* HIP device ordinal */
    int32_t width, height;  /* framebuffer size in pixels (canvas.width/height) */
    float early_out_eps;    /* 0: composite every splat like the reference (no early
                               termination); >0: a 16x16 tile stops once every pixel's
                               remaining transmittance 1-alpha is below this value */
    int32_t band_x0, band_x1; /* multi-GPU: this context composites only pixel columns
                               [band_x0, band_x1): band_x0 a multiple of 32 (whole compositor bins), band_x1 a
                               multiple of 32 or the image width; 0,0 = whole image */
    int32_t flags;          /* GSR_FLAG_* */
} gsr_options;

#define GSR_FLAG_TIMING 1   /* record HIP events around every stage (gsr_get_timings) */
#define GSR_FLAG_THROUGHPUT 2 /* the caller keeps several frames in flight on this device (one
                               context per frame): the other contexts' kernels, not extra pieces of
                               this frame, fill the GPU.  The compositor then cuts a frame into ~1300
                               work items instead of ~5000 and gives every 16x16 tile one wave
                               (k_blend, 7 four-wave workgroups per CU); without the flag a context
                               renders one frame at a time and, up to 4096 bins, gives every tile two
                               waves (k_blend2, 3 eight-wave workgroups per CU), which halves the
                               serial walk that bounds a lone frame.  Same pixels within float
                               rounding (pieces are combined associatively). */

/* Per-stage device times of the last completed gsr_render / gsr_sort, measured
 * with HIP events on the context's stream, plus the frame's list sizes. */
typedef struct gsr_timings {
    float ms_project_key; /* projection + depth key + min/max            */
    float ms_sort;        /* quantise + 2 radix passes -> depthIndex      */
    float ms_bin;         /* coarse bin count/scan/scatter                */
    float ms_blend;       /* k_blend: tile composite (and the fold of multi-segment bins), the dominant kernel */
    float ms_combine;     /* k_combine: fold of the per-segment partials; 0 when the fold runs inside k_blend (default) */
    float ms_total;       /* first event -> last event                    */
    uint64_t visible;     /* V: splats with a non-empty screen bbox (within the band) */
    uint64_t bin_entries; /* entries in the coarse (32x32 px) bin lists   */
    uint64_t tile_entries;/* D: sum over visible splats of 16x16 tiles their bbox overlaps */
    uint32_t n;           /* splats                                       */
    uint32_t frames;      /* frames accumulated in the sums below         */
    double sum_ms_project_key, sum_ms_sort, sum_ms_bin, sum_ms_blend, sum_ms_combine, sum_ms_total;
    /* device-side sums over every frame rendered since gsr_reset_timings (valid after gsr_sync) */
    uint64_t sum_visible, sum_bin_entries, sum_tile_entries, sum_frames;
    /* sticky since gsr_create (gsr_reset_timings does not clear them): */
    uint64_t overflow_frames; /* frames whose bin lists did not fit the list capacity (each made the lists regrow) */
    uint64_t dropped_frames;  /* of those, frames that were never composited: asynchronous frames behind which later
                                 frames had been enqueued before the host noticed (gsr_sync reports them once with
                                 GSR_ERR_OVERFLOW); 0 for callers of the blocking gsr_render */
} gsr_timings;

/* ---- lifetime ---- */
int gsr_create(gsr_ctx **out, const gsr_options *opt);
int gsr_destroy(gsr_ctx *ctx);
const char *gsr_last_error(gsr_ctx *ctx); /* ctx may be NULL: error of the failed gsr_create */

/* ---- per scene ---- */
/* data: Scene.data layout, 8 u32 per splat (src/core/Scene.ts:141-148,174-176);
 * positions: Scene.positions, 3 f32 per splat, must equal data words 0..2.
 * Repacked once into SoA on the device. */
int gsr_set_scene(gsr_ctx *ctx, const uint32_t *data, const float *positions, uint32_t n);

/* On-device scene build: rows = .splat bytes, 32 per splat (src/core/Scene.ts:9,126-148).  The
 * device does what Scene.setData does (covariance in f64, truncated halves) and keeps rotations/scales, so the
 * transforms below run as kernels instead of JavaScript loops + full re-upload.  Results are bit-identical to the
 * JavaScript ones (Scene.ts:126-366).  q = (x, y, z, w); box = xMin, xMax, yMin, yMax, zMin, zMax. */
int gsr_set_scene_rows(gsr_ctx *ctx, const uint8_t *rows, uint32_t n);
int gsr_scene_translate(gsr_ctx *ctx, const double *t /* 3 */);
int gsr_scene_rotate(gsr_ctx *ctx, const double *q /* 4 */);
int gsr_scene_scale(gsr_ctx *ctx, const double *s /* 3 */);
int gsr_scene_limit_box(gsr_ctx *ctx, const double *box /* 6 */, uint32_t *new_count);
/* Any of the outputs may be NULL.  data: 8 u32 per splat; rotations (w,x,y,z) / scales only for scenes built from rows. */
int gsr_read_scene(gsr_ctx *ctx, uint32_t *data, float *positions, float *rotations, float *scales, uint32_t *count);
int gsr_scene_count(gsr_ctx *ctx, uint32_t *count); /* splats in the device scene (changes with gsr_scene_limit_box); no copy */

/* Spherical-harmonics colour (the fork's SH textures): sh_r/g/b = Scene.shs_rgb (8 u32 = 16 truncated halves per
 * SH-carrying splat and channel, src/core/Scene.ts:108-124; uploaded by setShTextures, WebGLRenderer.ts:321-366),
 * band_index = Scene.bandsIndices (uniform u_bandIndex, WebGLRenderer.ts:209-211): splat i > band_index[0] takes its
 * colour from eval_sh_rgb (vertex.glsl.ts:57-104,180-204) with degree 1/2/3 by band_index[1], band_index[2].
 * sh_count must be n - (band_index[0] + 1).  Call after gsr_set_scene (which clears any SH state); sh_count 0 clears.
 * gsr_scene_limit_box renumbers the splats and therefore also clears the SH state. */
int gsr_set_scene_sh(gsr_ctx *ctx, const uint32_t *sh_r, const uint32_t *sh_g, const uint32_t *sh_b, uint32_t sh_count,
                     const int32_t *band_index /* 3 */);

/* ---- per resize / per frame ---- */
int gsr_resize(gsr_ctx *ctx, int32_t width, int32_t height);
/* Multi-GPU: restrict the context to the pixel columns [x0, x1) (same rule as gsr_options.band_x0/x1; 0,0 = whole
 * image).  A band context projects every splat (the depth key's min/max needs them all) but sorts, bins and composites
 * only the splats whose box touches the band. */
int gsr_set_band(gsr_ctx *ctx, int32_t x0, int32_t x1);
/* view, proj, view_proj: column-major f32[16] exactly as `new Float32Array(m.buffer)`
 * of Camera.viewMatrix / projectionMatrix / viewProj (src/cameras/Camera.ts:81-92). */
int gsr_set_camera(gsr_ctx *ctx, const float *view, const float *proj, const float *view_proj, float fx, float fy);

/* FadeInPass uniforms u_useDepthFade / u_depthFade (src/renderers/webgl/passes/FadeInPass.ts:8-37, consumed at
 * vertex.glsl.ts:214-229): while enabled every splat's axes are scaled by the depth-dependent factor. Off by default. */
int gsr_set_depth_fade(gsr_ctx *ctx, int32_t use_depth_fade, float depth_fade);

int gsr_sort(gsr_ctx *ctx);         /* depth key + sort only (blocking)                    */
int gsr_render(gsr_ctx *ctx);       /* sort + project + bin + composite (blocking)         */
int gsr_render_async(gsr_ctx *ctx); /* enqueue one frame on the context's stream           */
int gsr_sync(gsr_ctx *ctx);         /* wait for the stream; reports deferred errors        */
/* List overflow.  The per-bin splat lists live in one device buffer sized from the scene (6 entries per splat + 1 M);
 * a frame that needs more publishes no compositor work (its framebuffer keeps the preceding image), bumps a sticky
 * device counter and stores it in a host-mapped word.  The blocking gsr_render regrows and renders the frame again, so
 * its caller never sees this.  With gsr_render_async the host learns of it later: gsr_render_async polls the word
 * (a host memory read) and regrows before enqueuing the next frame; gsr_sync regrows, renders the LAST frame again if
 * it was among them, and returns GSR_ERR_OVERFLOW once if earlier frames were lost (gsr_timings.dropped_frames counts
 * them; the context stays usable).  gsr_overflow_pending: 1 while the device has reported an overflow that the host
 * has not handled yet -- a caller about to ship the frame it just enqueued (multi-GPU exchange) calls gsr_sync first. */
int gsr_overflow_pending(gsr_ctx *ctx);
/* Tuning/test hook: set the list capacity in entries (>= 1024).  Call after gsr_set_scene* (which sizes it anew). */
int gsr_set_list_capacity(gsr_ctx *ctx, uint32_t entries);

/* ---- results ---- */
/* The whole permutation (wasm's depthIndex).  On a band context the frame sorted only the band's survivors; this
 * call then runs the full sort first. */
int gsr_read_depth_index(gsr_ctx *ctx, uint32_t *out /* n */);
int gsr_read_pixels_rgba32f(gsr_ctx *ctx, float *out /* w*h*4, premultiplied, row 0 = top */);
int gsr_read_pixels_rgba8(gsr_ctx *ctx, uint8_t *out /* w*h*4, round(clamp(x,0,1)*255)   */);
int gsr_get_timings(gsr_ctx *ctx, gsr_timings *out);
int gsr_reset_timings(gsr_ctx *ctx);
/* With GSR_FLAG_TIMING: record the stage events only on every `every`-th frame (default 1).  The six events of a
 * frame are packets the GPU's command processor has to retire; on short frames (small scenes, one band of a
 * multi-GPU frame) timing every frame costs up to 15 % of the frame rate.  gsr_timings averages the sampled frames;
 * the first frame after this call or after gsr_reset_timings is always sampled.  every = 0xffffffff: no frame is
 * sampled (a context created with GSR_FLAG_TIMING then issues frames exactly like one created without). */
int gsr_set_timing_interval(gsr_ctx *ctx, uint32_t every);

/* ---- parity/debug read-backs (intermediate device buffers of the last frame) ---- */
int gsr_read_keys(gsr_ctx *ctx, uint32_t *keys /* n, 17-bit */, int32_t *minmax /* 2 */);
int gsr_read_records(gsr_ctx *ctx, float *rec /* 8n */, int32_t *bbox /* 4n: x0,y0,x1,y1 */);
int gsr_read_sh_colors(gsr_ctx *ctx, float *rgba /* 4n: evaluated SH colour of every splat that has one */);
/* How the last rendered frame's bin lists were handed to the compositor (the choices change no depth order, only f32
 * association): out[0] = list entries per segment, out[1] = work items, out[2] = 0 (reserved), out[3] = waves
 * per 16x16 tile (1: k_blend, 2: k_blend2), out[4] = bins of the context's band. */
int gsr_read_work_items(gsr_ctx *ctx, uint32_t *out /* 5 */);

/* ---- multi-GPU helpers ---- */
/* Entries per 32x32 bin of the last rendered frame, row-major over the context's band (cost model for balanced bands). */
int gsr_read_bin_totals(gsr_ctx *ctx, uint32_t *out /* nbx*nby */, int32_t *nbx, int32_t *nby);
/* The last rendered frame's bin lists (diagnostic; what the compositor walks: the reference has no counterpart, its GPU
 * rasteriser visits every splat for every pixel, WebGLRenderer.ts:282-296): starts[b] .. starts[b + 1] delimit bin b's
 * entries in `list` (splat indices, front to back), bins row-major over the context's band, starts[nbx * nby] = entries of
 * the frame.  `list` may be NULL (starts only); GSR_ERR_ARG when it holds fewer than that many words. */
int gsr_read_bin_lists(gsr_ctx *ctx, uint32_t *starts /* nbx*nby + 1 */, uint32_t *list, uint64_t list_words);
/* Enqueue the f32 -> RGBA8 conversion of the framebuffer on the context's stream (result: gsr_framebuffer8_device_ptr). */
int gsr_convert_rgba8_async(gsr_ctx *ctx);
void *gsr_framebuffer8_device_ptr(gsr_ctx *ctx); /* uint8[h][w][4] on the device */
/* Sender side of the framebuffer all-gather: convert this context's band columns to RGBA8 and
 * write them into `slab` (device memory, `height` rows of `slab_width_px` pixels, >= band width), on the
 * context's stream.  One pass over the band; replaces gsr_convert_rgba8_async + a strided copy. */
int gsr_pack_band_rgba8_async(gsr_ctx *ctx, void *slab, int32_t slab_width_px);
/* Receiver side: de-slab the gathered buffer [world][height][slab_width_px] (RGBA8) into the row-major
 * [height][width] `image`; rank q's columns are [x0[q], x1[q]).  Runs on `stream` (a hipStream_t: the stream
 * the collective was issued on, e.g. torch's current stream), device = the context's.  world <= 16. */
int gsr_unpack_slabs_rgba8_async(gsr_ctx *ctx, const void *gathered, void *image, int32_t slab_width_px, int32_t world,
                                 const int32_t *x0, const int32_t *x1, void *stream);

/* ---- multi-GPU frame exchange inside the library: RCCL all-gather over xGMI ----
 * One process per GPU, one context per process (or per frame in flight).  Rank 0 makes an id and hands its 128 bytes
 * to the other ranks by whatever the host has (a file, a socket, a torch/gloo broadcast); then EVERY rank calls
 * gsr_comm_init with the same id, world and band edges (x0[q], x1[q]) = pixel columns of rank q: contiguous, whole
 * 32-px bin columns, covering [0, width).  gsr_comm_init is collective (returns when all ranks have joined), sets this
 * context's band to its own columns and allocates the exchange buffers.  Per frame:
 *     gsr_render_async(ctx);            band: project all, sort/bin/composite the band's splats
 *     gsr_allgather_frame_async(ctx);   band -> RGBA8 slab (render stream), ONE ncclAllGather of equal slabs + one
 *                                       de-slab kernel on the context's exchange stream; device-side ordering only,
 *                                       so the next frame's kernels overlap the collective
 * and every rank holds the whole RGBA8 frame: gsr_read_frame_rgba8 (waits for the exchange, copies to the host) or
 * gsr_frame8_device_ptr.  The replaced reference entry is still renderer.render(scene, camera)
 * (src/renderers/WebGLRenderer.ts:241-296): the JS HIPRenderer calls exactly this sequence when it has joined a group.
 * world == 1 is allowed (self test; the band is the whole image). */
#define GSR_COMM_ID_BYTES 128
int gsr_comm_unique_id(uint8_t *id /* GSR_COMM_ID_BYTES */);
int gsr_comm_init(gsr_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world, const int32_t *x0, const int32_t *x1);
/* A second (third, ...) context of the SAME rank -- frames in flight -- joins the group `leader` has joined: it uses
 * leader's communicator and exchange stream (so a rank's collectives are issued on ONE stream, in the order of the
 * gsr_allgather_frame_async calls, which must be the same on every rank) and gets its own slab and frame buffers.
 * Several communicators per device with collectives in flight on different streams are the RCCL/NCCL case that can
 * deadlock when the ranks' collectives are scheduled in different orders; this avoids it.  A leader that is destroyed
 * (or leaves with gsr_comm_destroy) first detaches its sharers: they become plain contexts again and fail
 * gsr_allgather_frame_async with GSR_ERR_ARG until they join a group anew.  A custom collective's `user` pointer must
 * outlive every context that uses it. */
int gsr_comm_share(gsr_ctx *ctx, gsr_ctx *leader);
/* Test hook: gsr_comm_init with the caller's collective in place of ncclAllGather, for hosts that cannot form an RCCL
 * communicator of more than one rank (RCCL refuses two ranks on one device) but want to run the exchange's choreography
 * -- band pack, slab padding to the widest band, event ordering against the render stream, de-slab with unequal edges --
 * with world > 1.  gsr_allgather_frame_async calls fn(user, send, recv, bytes_per_rank, stream) on the host in place of
 * ncclAllGather: send = this rank's slab, recv = [world][bytes_per_rank], both device memory; work enqueued on `stream`
 * (a hipStream_t, the exchange stream) before the call has packed the slab, work enqueued on it afterwards reads recv.
 * fn may block.  Returns non-zero on failure.  No product path uses it. */
typedef int (*gsr_allgather_fn)(void *user, const void *send, void *recv, uint64_t bytes_per_rank, void *stream);
int gsr_comm_init_custom(gsr_ctx *ctx, int32_t rank, int32_t world, const int32_t *x0, const int32_t *x1, gsr_allgather_fn fn,
                         void *user);
int gsr_comm_destroy(gsr_ctx *ctx);
int gsr_allgather_frame_async(gsr_ctx *ctx);
/* Waits for the exchange AND for the render stream.  A band packed behind a frame whose bin lists did not fit is the
 * preceding image: every slab carries its frame's overflow flag through the all-gather, so EVERY rank of the group sees
 * which gathered frame holds a stale band and gets GSR_ERR_OVERFLOW for that frame (the rank concerned has regrown its
 * lists by then): all ranks render and gather it again -- the collective is repeated by the whole group, never by one
 * rank alone.  Frames dropped earlier on this rank (reported once by gsr_sync) do not make a good frame unreadable. */
int gsr_read_frame_rgba8(gsr_ctx *ctx, uint8_t *out /* w*h*4: the gathered frame */);
void *gsr_frame8_device_ptr(gsr_ctx *ctx);    /* uint8[h][w][4], the gathered frame on the device */
void *gsr_comm_stream_handle(gsr_ctx *ctx);   /* hipStream_t the exchange runs on */

/* ---- device interop (torch / RCCL plumbing in the harness) ---- */
void *gsr_framebuffer_device_ptr(gsr_ctx *ctx); /* float4[h][w] on the device */
void *gsr_stream_handle(gsr_ctx *ctx);          /* hipStream_t */
/* Device-side ordering between the context's stream and another stream of the same device (no host wait):
 * ctx_waits = 0: work submitted to `other_stream` after this call waits for everything enqueued on the context so far;
 * ctx_waits = 1: the context's later work waits for everything enqueued on `other_stream` so far. */
int gsr_stream_order(gsr_ctx *ctx, void *other_stream, int32_t ctx_waits);
int gsr_device_info(gsr_ctx *ctx, char *name, int32_t name_len, int32_t *compute_units, int32_t *clock_khz);
/* Hash of the kernel sources this library was built from (hex string; "unknown" for an ad-hoc build): profiler
 * measurements are stamped with it so that they are never attributed to a different build. */
const char *gsr_build_id(void);

/* ---- drop-in for the wasm export, same argument list as wasm/wasm.cpp:8-13.
 * Host pointers; depthBuffer/starts/counts may be NULL (depthBuffer, when given,
 * receives the 17-bit keys like the reference leaves them).  Uses a process-wide
 * context on device 0; returns nothing, like the reference.  The positions are copied
 * to the device on every call (nothing of the caller's is remembered between calls:
 * Worker.ts:23-27 re-copies them on every scene message, and JS hosts edit them in place);
 * on failure depthIndex is zero-filled and the reason is printed to stderr. */
void gsplat_sort_host(const float *viewProj, uint32_t vertexCount, const float *fBuffer, uint32_t *depthBuffer,
                      uint32_t *depthIndex, uint32_t *starts, uint32_t *counts);

#ifdef __cplusplus
}
#endif
#endif
