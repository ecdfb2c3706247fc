// gsplat_hip.node -- N-API (raw node_api.h, N-API <= 8, Node >= 12) binding of the C ABI in
// include/gsplat_hip.h.  TypedArray backing stores are handed to the library zero-copy for the duration of
// each call; nothing is retained.  Every failure becomes a JavaScript exception carrying gsr_last_error().
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../../include/gsplat_hip.h"

namespace {

#define NAPI_OK_OR_NULL(env, call)                                    \
    do {                                                              \
        if ((call) != napi_ok) {                                      \
            napi_throw_error((env), nullptr, "N-API call failed: " #call); \
            return nullptr;                                           \
        }                                                             \
    } while (0)

napi_value throw_gsr(napi_env env, gsr_ctx* ctx, int rc, const char* what)
{
    char buf[768];
    snprintf(buf, sizeof buf, "%s failed (%d): %s", what, rc, gsr_last_error(ctx));
    napi_throw_error(env, "GSPLAT_HIP", buf);
    return nullptr;
}

void finalize_ctx(napi_env, void* data, void*)
{
    gsr_ctx** slot = static_cast<gsr_ctx**>(data);
    if (*slot) gsr_destroy(*slot);
    delete slot;
}

bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
        napi_throw_type_error(env, nullptr, "too few arguments");
        return false;
    }
    return true;
}

gsr_ctx* get_ctx(napi_env env, napi_value v)
{
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !*static_cast<gsr_ctx**>(p)) {
        napi_throw_type_error(env, nullptr, "expected a live renderer handle");
        return nullptr;
    }
    return *static_cast<gsr_ctx**>(p);
}

// data pointer + element count of a TypedArray of the given type (nullptr allowed when `optional`)
bool get_typed(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len, bool optional = false)
{
    napi_valuetype t;
    napi_typeof(env, v, &t);
    if (optional && (t == napi_undefined || t == napi_null)) { *data = nullptr; *len = 0; return true; }
    bool is_ta = false;
    napi_is_typedarray(env, v, &is_ta);
    napi_typedarray_type type;
    napi_value ab;
    size_t off;
    if (!is_ta || napi_get_typedarray_info(env, v, &type, len, data, &ab, &off) != napi_ok || type != want) {
        napi_throw_type_error(env, nullptr, "wrong TypedArray type");
        return false;
    }
    return true;
}

bool get_i32(napi_env env, napi_value v, int32_t* out) { return napi_get_value_int32(env, v, out) == napi_ok; }
bool get_f64(napi_env env, napi_value v, double* out) { return napi_get_value_double(env, v, out) == napi_ok; }

napi_value undefined(napi_env env) { napi_value u; napi_get_undefined(env, &u); return u; }

// create({device,width,height,earlyOutEps,bandX0,bandX1,timing,throughput}) -> handle
napi_value Create(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    gsr_options o;
    memset(&o, 0, sizeof o);
    auto num = [&](const char* key, double dflt) {
        napi_value v;
        bool has = false;
        double d = dflt;
        if (napi_has_named_property(env, argv[0], key, &has) == napi_ok && has &&
            napi_get_named_property(env, argv[0], key, &v) == napi_ok)
            napi_get_value_double(env, v, &d);
        return d;
    };
    o.device = (int32_t)num("device", 0);
    o.width = (int32_t)num("width", 0);
    o.height = (int32_t)num("height", 0);
    o.early_out_eps = (float)num("earlyOutEps", 0);
    o.band_x0 = (int32_t)num("bandX0", 0);
    o.band_x1 = (int32_t)num("bandX1", 0);
    o.flags = (num("timing", 0) != 0 ? GSR_FLAG_TIMING : 0) | (num("throughput", 0) != 0 ? GSR_FLAG_THROUGHPUT : 0);
    gsr_ctx* ctx = nullptr;
    const int rc = gsr_create(&ctx, &o);
    if (rc != GSR_OK) return throw_gsr(env, nullptr, rc, "gsr_create");
    gsr_ctx** slot = new gsr_ctx*(ctx);
    napi_value ext;
    if (napi_create_external(env, slot, finalize_ctx, nullptr, &ext) != napi_ok) {
        gsr_destroy(ctx);
        delete slot;
        napi_throw_error(env, nullptr, "napi_create_external failed");
        return nullptr;
    }
    return ext;
}

napi_value Destroy(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    void* p = nullptr;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        gsr_ctx** slot = static_cast<gsr_ctx**>(p);
        if (*slot) { gsr_destroy(*slot); *slot = nullptr; }
    }
    return undefined(env);
}

napi_value SetScene(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    void *data, *pos;
    size_t nd, np;
    int32_t n;
    if (!get_typed(env, argv[1], napi_uint32_array, &data, &nd) || !get_typed(env, argv[2], napi_float32_array, &pos, &np) ||
        !get_i32(env, argv[3], &n))
        return nullptr;
    if (n < 0 || nd < (size_t)n * 8 || np < (size_t)n * 3) {
        napi_throw_range_error(env, nullptr, "scene buffers are smaller than vertexCount requires");
        return nullptr;
    }
    const int rc = gsr_set_scene(c, (const uint32_t*)data, (const float*)pos, (uint32_t)n);
    return rc ? throw_gsr(env, c, rc, "gsr_set_scene") : undefined(env);
}

// setSceneSh(handle, Uint32Array r, Uint32Array g, Uint32Array b, shCount, Int32Array bandsIndices)
napi_value SetSceneSh(napi_env env, napi_callback_info info)
{
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    void *t[3], *band;
    size_t len[3], nb;
    int32_t count;
    for (int k = 0; k < 3; k++)
        if (!get_typed(env, argv[1 + k], napi_uint32_array, &t[k], &len[k])) return nullptr;
    if (!get_i32(env, argv[4], &count) || !get_typed(env, argv[5], napi_int32_array, &band, &nb)) return nullptr;
    if (count < 0 || nb < 3 || len[0] < (size_t)count * 8 || len[1] < (size_t)count * 8 || len[2] < (size_t)count * 8) {
        napi_throw_range_error(env, nullptr, "SH buffers are smaller than shCount requires");
        return nullptr;
    }
    const int rc = gsr_set_scene_sh(c, (const uint32_t*)t[0], (const uint32_t*)t[1], (const uint32_t*)t[2], (uint32_t)count,
                                    (const int32_t*)band);
    return rc ? throw_gsr(env, c, rc, "gsr_set_scene_sh") : undefined(env);
}

// setSceneRows(handle, Uint8Array rows)
napi_value SetSceneRows(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    void* rows;
    size_t len;
    if (!c || !get_typed(env, argv[1], napi_uint8_array, &rows, &len)) return nullptr;
    if (len % 32) { napi_throw_range_error(env, nullptr, "rows length must be a multiple of 32"); return nullptr; }
    const int rc = gsr_set_scene_rows(c, (const uint8_t*)rows, (uint32_t)(len / 32));
    return rc ? throw_gsr(env, c, rc, "gsr_set_scene_rows") : undefined(env);
}

// sceneTransform(handle, kind, Float64Array args): kind 0 translate(3) 1 rotate(4: x,y,z,w) 2 scale(3) 3 limitBox(6) -> new count
napi_value SceneTransform(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    int32_t kind;
    void* a;
    size_t len;
    if (!c || !get_i32(env, argv[1], &kind) || !get_typed(env, argv[2], napi_float64_array, &a, &len)) return nullptr;
    static const size_t need[4] = {3, 4, 3, 6};
    if (kind < 0 || kind > 3 || len < need[kind]) { napi_throw_range_error(env, nullptr, "bad transform arguments"); return nullptr; }
    uint32_t count = 0;
    int rc;
    const double* d = (const double*)a;
    if (kind == 0) rc = gsr_scene_translate(c, d);
    else if (kind == 1) rc = gsr_scene_rotate(c, d);
    else if (kind == 2) rc = gsr_scene_scale(c, d);
    else rc = gsr_scene_limit_box(c, d, &count);
    if (rc) return throw_gsr(env, c, rc, "gsr_scene transform");
    if (kind != 3) gsr_scene_count(c, &count);   // no copy: the transforms exist to keep the scene on the device
    napi_value n;
    napi_create_uint32(env, count, &n);
    return n;
}

// readScene(handle, Uint32Array data | null, Float32Array positions | null) -> count
napi_value ReadScene(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    void *data, *pos;
    size_t nd, np;
    if (!c || !get_typed(env, argv[1], napi_uint32_array, &data, &nd, true) || !get_typed(env, argv[2], napi_float32_array, &pos, &np, true))
        return nullptr;
    uint32_t count = 0;
    int rc = gsr_scene_count(c, &count);
    if (!rc) {
        if ((data && nd < (size_t)count * 8) || (pos && np < (size_t)count * 3)) {
            napi_throw_range_error(env, nullptr, "output arrays are smaller than the scene");
            return nullptr;
        }
        rc = gsr_read_scene(c, (uint32_t*)data, (float*)pos, nullptr, nullptr, &count);
    }
    if (rc) return throw_gsr(env, c, rc, "gsr_read_scene");
    napi_value n;
    napi_create_uint32(env, count, &n);
    return n;
}

napi_value SetDepthFade(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    int32_t use;
    double v;
    if (!c || !get_i32(env, argv[1], &use) || !get_f64(env, argv[2], &v)) return nullptr;
    const int rc = gsr_set_depth_fade(c, use, (float)v);
    return rc ? throw_gsr(env, c, rc, "gsr_set_depth_fade") : undefined(env);
}

napi_value Resize(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    int32_t w, h;
    if (!c || !get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h)) return nullptr;
    const int rc = gsr_resize(c, w, h);
    return rc ? throw_gsr(env, c, rc, "gsr_resize") : undefined(env);
}

napi_value SetBand(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    int32_t x0, x1;
    if (!c || !get_i32(env, argv[1], &x0) || !get_i32(env, argv[2], &x1)) return nullptr;
    const int rc = gsr_set_band(c, x0, x1);
    return rc ? throw_gsr(env, c, rc, "gsr_set_band") : undefined(env);
}

// setCamera(handle, Float32Array view, Float32Array proj, Float32Array viewProj, fx, fy)
napi_value SetCamera(napi_env env, napi_callback_info info)
{
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    void* m[3];
    size_t len;
    for (int k = 0; k < 3; k++) {
        if (!get_typed(env, argv[1 + k], napi_float32_array, &m[k], &len)) return nullptr;
        if (len < 16) { napi_throw_range_error(env, nullptr, "matrix needs 16 elements"); return nullptr; }
    }
    double fx, fy;
    if (!get_f64(env, argv[4], &fx) || !get_f64(env, argv[5], &fy)) return nullptr;
    const int rc = gsr_set_camera(c, (const float*)m[0], (const float*)m[1], (const float*)m[2], (float)fx, (float)fy);
    return rc ? throw_gsr(env, c, rc, "gsr_set_camera") : undefined(env);
}

template <int (*FN)(gsr_ctx*)>
napi_value Call0(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    const int rc = FN(c);
    return rc ? throw_gsr(env, c, rc, "libgsplat_hip call") : undefined(env);
}

napi_value ReadDepthIndex(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    void* out;
    size_t len;
    if (!c || !get_typed(env, argv[1], napi_uint32_array, &out, &len)) return nullptr;
    uint32_t n = 0;
    gsr_scene_count(c, &n);
    if (len < n) { napi_throw_range_error(env, nullptr, "output array is smaller than vertexCount"); return nullptr; }
    const int rc = gsr_read_depth_index(c, (uint32_t*)out);
    return rc ? throw_gsr(env, c, rc, "gsr_read_depth_index") : undefined(env);
}

// readPixels(handle, out, width, height): out is Float32Array (RGBA f32) or Uint8Array (RGBA8)
napi_value ReadPixels(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    int32_t w, h;
    if (!get_i32(env, argv[2], &w) || !get_i32(env, argv[3], &h)) return nullptr;
    napi_typedarray_type type;
    size_t len, off;
    void* data;
    napi_value ab;
    if (napi_get_typedarray_info(env, argv[1], &type, &len, &data, &ab, &off) != napi_ok) {
        napi_throw_type_error(env, nullptr, "expected a TypedArray");
        return nullptr;
    }
    if (len < (size_t)w * h * 4) { napi_throw_range_error(env, nullptr, "output array is smaller than width*height*4"); return nullptr; }
    int rc;
    if (type == napi_float32_array) rc = gsr_read_pixels_rgba32f(c, (float*)data);
    else if (type == napi_uint8_array || type == napi_uint8_clamped_array) rc = gsr_read_pixels_rgba8(c, (uint8_t*)data);
    else { napi_throw_type_error(env, nullptr, "expected Float32Array or Uint8Array"); return nullptr; }
    return rc ? throw_gsr(env, c, rc, "gsr_read_pixels") : undefined(env);
}

napi_value GetTimings(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    gsr_timings t;
    const int rc = gsr_get_timings(c, &t);
    if (rc) return throw_gsr(env, c, rc, "gsr_get_timings");
    napi_value o;
    NAPI_OK_OR_NULL(env, napi_create_object(env, &o));
    auto put = [&](const char* k, double v) {
        napi_value n;
        napi_create_double(env, v, &n);
        napi_set_named_property(env, o, k, n);
    };
    put("msProjectKey", t.ms_project_key); put("msSort", t.ms_sort); put("msBin", t.ms_bin); put("msBlend", t.ms_blend); put("msCombine", t.ms_combine);
    put("msTotal", t.ms_total); put("visible", (double)t.visible); put("binEntries", (double)t.bin_entries);
    put("tileEntries", (double)t.tile_entries); put("n", t.n); put("frames", t.frames);
    put("sumMsProjectKey", t.sum_ms_project_key); put("sumMsSort", t.sum_ms_sort); put("sumMsBin", t.sum_ms_bin);
    put("sumMsBlend", t.sum_ms_blend); put("sumMsCombine", t.sum_ms_combine); put("sumMsTotal", t.sum_ms_total);
    put("overflowFrames", (double)t.overflow_frames); put("droppedFrames", (double)t.dropped_frames);
    return o;
}

// overflowPending(handle) -> boolean
napi_value OverflowPending(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    napi_value b;
    napi_get_boolean(env, gsr_overflow_pending(c) != 0, &b);
    return b;
}

// setListCapacity(handle, entries): tuning/test hook
napi_value SetListCapacity(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    double v;
    if (!c || !get_f64(env, argv[1], &v)) return nullptr;
    const int rc = gsr_set_list_capacity(c, v < 0 ? 0u : (uint32_t)v);
    return rc ? throw_gsr(env, c, rc, "gsr_set_list_capacity") : undefined(env);
}

// commUniqueId() -> Uint8Array(128): rank 0 makes it, the host hands it to the other ranks
napi_value CommUniqueId(napi_env env, napi_callback_info)
{
    uint8_t id[GSR_COMM_ID_BYTES];
    const int rc = gsr_comm_unique_id(id);
    if (rc) return throw_gsr(env, nullptr, rc, "gsr_comm_unique_id");
    napi_value ab, out;
    void* data = nullptr;
    NAPI_OK_OR_NULL(env, napi_create_arraybuffer(env, GSR_COMM_ID_BYTES, &data, &ab));
    memcpy(data, id, GSR_COMM_ID_BYTES);
    NAPI_OK_OR_NULL(env, napi_create_typedarray(env, napi_uint8_array, GSR_COMM_ID_BYTES, ab, 0, &out));
    return out;
}

// commInit(handle, Uint8Array id, rank, world, Int32Array x0, Int32Array x1): collective
napi_value CommInit(napi_env env, napi_callback_info info)
{
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    void *id, *x0, *x1;
    size_t nid, n0, n1;
    int32_t rank, world;
    if (!get_typed(env, argv[1], napi_uint8_array, &id, &nid) || !get_i32(env, argv[2], &rank) || !get_i32(env, argv[3], &world) ||
        !get_typed(env, argv[4], napi_int32_array, &x0, &n0) || !get_typed(env, argv[5], napi_int32_array, &x1, &n1))
        return nullptr;
    if (nid < GSR_COMM_ID_BYTES || world < 1 || n0 < (size_t)world || n1 < (size_t)world) {
        napi_throw_range_error(env, nullptr, "commInit: id needs 128 bytes, x0/x1 one entry per rank");
        return nullptr;
    }
    const int rc = gsr_comm_init(c, (const uint8_t*)id, rank, world, (const int32_t*)x0, (const int32_t*)x1);
    return rc ? throw_gsr(env, c, rc, "gsr_comm_init") : undefined(env);
}

// commShare(handle, leaderHandle): this context joins the group the leader has joined (same rank: another frame in flight)
napi_value CommShare(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    gsr_ctx* leader = get_ctx(env, argv[1]);
    if (!c || !leader) return nullptr;
    const int rc = gsr_comm_share(c, leader);
    return rc ? throw_gsr(env, c, rc, "gsr_comm_share") : undefined(env);
}

// readFrame(handle, Uint8Array out, width, height): the gathered RGBA8 frame
napi_value ReadFrame(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    void* out;
    size_t len;
    int32_t w, h;
    if (!c || !get_typed(env, argv[1], napi_uint8_array, &out, &len) || !get_i32(env, argv[2], &w) || !get_i32(env, argv[3], &h)) return nullptr;
    if (len < (size_t)w * h * 4) { napi_throw_range_error(env, nullptr, "output array is smaller than width*height*4"); return nullptr; }
    const int rc = gsr_read_frame_rgba8(c, (uint8_t*)out);
    return rc ? throw_gsr(env, c, rc, "gsr_read_frame_rgba8") : undefined(env);
}

napi_value BuildId(napi_env env, napi_callback_info)
{
    napi_value s;
    napi_create_string_utf8(env, gsr_build_id(), NAPI_AUTO_LENGTH, &s);
    return s;
}

napi_value DeviceInfo(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    gsr_ctx* c = get_ctx(env, argv[0]);
    if (!c) return nullptr;
    char name[256] = {0};
    int32_t cus = 0, khz = 0;
    const int rc = gsr_device_info(c, name, sizeof name, &cus, &khz);
    if (rc) return throw_gsr(env, c, rc, "gsr_device_info");
    napi_value o, s, a, b;
    NAPI_OK_OR_NULL(env, napi_create_object(env, &o));
    napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &s);
    napi_create_int32(env, cus, &a);
    napi_create_int32(env, khz, &b);
    napi_set_named_property(env, o, "name", s);
    napi_set_named_property(env, o, "computeUnits", a);
    napi_set_named_property(env, o, "clockKhz", b);
    return o;
}

// sortHost(viewProj f32[16], vertexCount, fBuffer f32, depthBuffer u32|null, depthIndex u32): the 7-argument wasm
// export of wasm/wasm.cpp:8-13 (starts/counts are scratch the device path does not need)
napi_value SortHost(napi_env env, napi_callback_info info)
{
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return nullptr;
    void *vp, *fb, *db, *di;
    size_t lvp, lfb, ldb, ldi;
    int32_t n;
    if (!get_typed(env, argv[0], napi_float32_array, &vp, &lvp) || !get_i32(env, argv[1], &n) ||
        !get_typed(env, argv[2], napi_float32_array, &fb, &lfb) || !get_typed(env, argv[3], napi_uint32_array, &db, &ldb, true) ||
        !get_typed(env, argv[4], napi_uint32_array, &di, &ldi))
        return nullptr;
    if (n < 0 || lvp < 16 || lfb < (size_t)n * 3 || ldi < (size_t)n || (db && ldb < (size_t)n)) {
        napi_throw_range_error(env, nullptr, "buffers are smaller than vertexCount requires");
        return nullptr;
    }
    gsplat_sort_host((const float*)vp, (uint32_t)n, (const float*)fb, (uint32_t*)db, (uint32_t*)di, nullptr, nullptr);
    return undefined(env);
}

napi_value Init(napi_env env, napi_value exports)
{
    struct { const char* name; napi_callback fn; } fns[] = {
        {"create", Create}, {"destroy", Destroy}, {"setScene", SetScene}, {"setSceneSh", SetSceneSh}, {"setDepthFade", SetDepthFade}, {"setSceneRows", SetSceneRows},
        {"sceneTransform", SceneTransform}, {"readScene", ReadScene}, {"resize", Resize}, {"setBand", SetBand},
        {"setCamera", SetCamera}, {"sort", Call0<gsr_sort>}, {"render", Call0<gsr_render>},
        {"renderAsync", Call0<gsr_render_async>}, {"sync", Call0<gsr_sync>}, {"resetTimings", Call0<gsr_reset_timings>},
        {"readDepthIndex", ReadDepthIndex}, {"readPixels", ReadPixels}, {"getTimings", GetTimings},
        {"deviceInfo", DeviceInfo}, {"sortHost", SortHost}, {"overflowPending", OverflowPending},
        {"setListCapacity", SetListCapacity}, {"buildId", BuildId}, {"commUniqueId", CommUniqueId}, {"commInit", CommInit}, {"commShare", CommShare},
        {"commDestroy", Call0<gsr_comm_destroy>}, {"allgatherFrameAsync", Call0<gsr_allgather_frame_async>}, {"readFrame", ReadFrame},
    };
    for (auto& f : fns) {
        napi_value fn;
        if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn) != napi_ok ||
            napi_set_named_property(env, exports, f.name, fn) != napi_ok)
            return nullptr;
    }
    return exports;
}

}  // namespace

NAPI_MODULE(gsplat_hip, Init)
