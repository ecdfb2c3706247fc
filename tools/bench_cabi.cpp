// Stand-alone C++ caller of the C ABI (include/gsplat_hip.h): no Python, no Node, no torch.
// SURVEY.md 8(b) lists three callers of libgsplat_hip.so -- the N-API addon, the ctypes harness and a C++ program;
// this is the third.  It renders the bench's 120-pose orbit with F frames in flight and prints one
// JSON line; with --rows / --dump it doubles as a cross-check of the other two hosts (same bytes in, same hashes out).
//
//   bench_cabi [--config C1|C2|C3|C4] [--rows file.splat] [--frames K] [--warmup W] [--in-flight F] [--dump prefix]
//
// Scene: the seeded synthetic generator of gsplat_hip/synth.py (mulberry32 counter PRNG, 24 draws per splat) written
// out again in C++; log/exp/cos come from libm here and from numpy there, so a byte may differ in a rare rounding --
// pass --rows to feed the exact bytes another host used.  Scene.setData itself runs on the device (gsr_set_scene_rows).
// Camera: Camera.update (src/cameras/Camera.ts:81-92) and the OrbitControls pose formula (OrbitControls.ts:275-283)
// in double precision, rounded to float like `new Float32Array(m.buffer)`.
// Build: gsplat.js_amd/csrc/Makefile (target bench_cabi), plain g++ against the header and the shared library.
#include "../include/gsplat_hip.h"

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Config { const char* name; uint32_t seed, n; int w, h; double sigma, s_lo, s_hi, fx; };
const Config CONFIGS[] = {
    {"C1", 1, 10000, 640, 480, 1.5, 0.004, 0.06, 1132.0},
    {"C2", 2, 300000, 1920, 1080, 1.0, 0.003, 0.04, 1132.0},
    {"C3", 3, 1000000, 1920, 1080, 1.5, 0.004, 0.06, 1132.0},
    {"C4", 4, 5000000, 3840, 2160, 1.5, 0.004, 0.06, 2264.0},
};

uint32_t mulberry32(uint32_t seed, uint64_t call)   // call is 1-based, like synth.mulberry32
{
    uint32_t t = (uint32_t)(seed + call * 0x6D2B79F5ull);
    t = (t ^ (t >> 15)) * (t | 1u);
    t = t ^ (t + (t ^ (t >> 7)) * (t | 61u));
    return t ^ (t >> 14);
}

std::vector<uint8_t> synth_rows(const Config& c)
{
    const int DRAWS = 24;
    std::vector<uint8_t> out((size_t)c.n * 32);
    const double ls = std::log(c.s_lo), lr = std::log(c.s_hi) - std::log(c.s_lo);
    for (uint32_t i = 0; i < c.n; i++) {
        double u[DRAWS];
        for (int d = 0; d < DRAWS; d++) u[d] = (double)mulberry32(c.seed, (uint64_t)i * DRAWS + d + 1) / 4294967296.0;
        auto normal = [&](int a, int b) { return std::sqrt(-2.0 * std::log(1.0 - u[a])) * std::cos(2.0 * M_PI * u[b]); };
        uint8_t* row = &out[(size_t)i * 32];
        for (int k = 0; k < 3; k++) {
            double p = normal(2 * k, 2 * k + 1) * c.sigma;
            p = p < -6.0 ? -6.0 : p > 6.0 ? 6.0 : p;
            const float pf = (float)p, sf = (float)std::exp(ls + u[6 + k] * lr);
            std::memcpy(row + 4 * k, &pf, 4);
            std::memcpy(row + 12 + 4 * k, &sf, 4);
            row[24 + k] = (uint8_t)std::floor(256.0 * u[9 + k]);
        }
        row[27] = (uint8_t)(32 + std::floor(224.0 * u[12]));
        double q[4], len = 0;
        for (int k = 0; k < 4; k++) { q[k] = normal(13 + 2 * k, 14 + 2 * k); len += q[k] * q[k]; }
        len = std::sqrt(len);
        if (len < 1e-12) len = 1e-12;
        for (int k = 0; k < 4; k++) {
            double v = std::nearbyint(q[k] / len * 128.0 + 128.0);   // numpy.round: half to even
            row[28 + k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    return out;
}

struct Cam { float view[16], proj[16], vp[16]; };

Cam orbit_camera(int k, int frames, int W, int H, double fx)
{
    const double alpha = 2.0 * M_PI * k / frames, beta = 0.3, radius = 8.0, near = 0.01, far = 1000.0;
    const double x = radius * std::sin(alpha) * std::cos(beta), y = -radius * std::sin(beta), z = -radius * std::cos(alpha) * std::cos(beta);
    double dx = -x, dy = -y, dz = -z;
    const double ln = std::sqrt(dx * dx + dy * dy + dz * dz);
    dx /= ln; dy /= ln; dz /= ln;
    const double rx = std::asin(-dy), ry = std::atan2(dx, dz);
    // Quaternion.FromEuler (src/math/Quaternion.ts:65-83) with ez = 0
    const double hx = rx / 2, hy = ry / 2, cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hx), sp = std::sin(hx), cz = 1.0, sz = 0.0;
    const double qx = cy * sp * cz + sy * cp * sz, qy = sy * cp * cz - cy * sp * sz, qz = cy * cp * sz - sy * sp * cz,
                 qw = cy * cp * cz + sy * sp * sz;
    // Matrix3.RotationFromQuaternion (src/math/Matrix3.ts:67-80)
    const double R[9] = {1 - 2 * qy * qy - 2 * qz * qz, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw,
                         2 * qx * qy + 2 * qz * qw, 1 - 2 * qx * qx - 2 * qz * qz, 2 * qy * qz - 2 * qx * qw,
                         2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx * qx - 2 * qy * qy};
    const double P[16] = {2 * fx / W, 0, 0, 0, 0, -2 * fx / H, 0, 0, 0, 0, far / (far - near), 1, 0, 0, -(far * near) / (far - near), 0};
    const double V[16] = {R[0], R[1], R[2], 0, R[3], R[4], R[5], 0, R[6], R[7], R[8], 0,
                          -x * R[0] - y * R[3] - z * R[6], -x * R[1] - y * R[4] - z * R[7], -x * R[2] - y * R[5] - z * R[8], 1};
    Cam c;
    for (int i = 0; i < 4; i++)      // Matrix4.multiply (src/math/Matrix4.ts:32-53): viewProj = projection.multiply(view)
        for (int j = 0; j < 4; j++)
            c.vp[4 * i + j] = (float)(V[4 * i + 0] * P[j] + V[4 * i + 1] * P[4 + j] + V[4 * i + 2] * P[8 + j] + V[4 * i + 3] * P[12 + j]);
    for (int i = 0; i < 16; i++) { c.view[i] = (float)V[i]; c.proj[i] = (float)P[i]; }
    return c;
}

uint64_t fnv1a(const void* p, size_t bytes)
{
    uint64_t h = 1469598103934665603ull;
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        const int rc_ = (call);                                                                         \
        if (rc_) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, gsr_last_error(ctx0)); return 1; } \
    } while (0)

}  // namespace

int main(int argc, char** argv)
{
    std::string config = "C1", rows_path, dump;
    int frames = 240, warmup = 20, in_flight = 3;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--config") config = next();
        else if (a == "--rows") rows_path = next();
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--warmup") warmup = std::atoi(next());
        else if (a == "--in-flight") in_flight = std::atoi(next());
        else if (a == "--dump") dump = next();
        else { std::fprintf(stderr, "usage: bench_cabi [--config C1..C4] [--rows f.splat] [--frames K] [--warmup W] [--in-flight F] [--dump prefix]\n"); return 2; }
    }
    const Config* cfg = nullptr;
    for (const Config& c : CONFIGS) if (config == c.name) cfg = &c;
    if (!cfg || in_flight < 1 || frames < 1) { std::fprintf(stderr, "bad arguments\n"); return 2; }

    std::vector<uint8_t> rows;
    if (!rows_path.empty()) {
        FILE* f = std::fopen(rows_path.c_str(), "rb");
        if (!f) { std::fprintf(stderr, "cannot open %s\n", rows_path.c_str()); return 2; }
        std::fseek(f, 0, SEEK_END);
        rows.resize((size_t)std::ftell(f));
        std::fseek(f, 0, SEEK_SET);
        if (std::fread(rows.data(), 1, rows.size(), f) != rows.size()) { std::fclose(f); return 2; }
        std::fclose(f);
    } else {
        rows = synth_rows(*cfg);
    }
    const uint32_t n = (uint32_t)(rows.size() / 32);

    gsr_ctx* ctx0 = nullptr;
    std::vector<gsr_ctx*> ctx(in_flight, nullptr);
    for (int c = 0; c < in_flight; c++) {
        gsr_options o{};
        o.device = 0; o.width = cfg->w; o.height = cfg->h;
        o.flags = in_flight > 1 ? GSR_FLAG_THROUGHPUT : 0;
        const int rc = gsr_create(&ctx[c], &o);
        if (rc) { std::fprintf(stderr, "gsr_create failed (%d): %s\n", rc, gsr_last_error(nullptr)); return 1; }
        ctx0 = ctx[c];
        CHECK(gsr_set_scene_rows(ctx[c], rows.data(), n));
    }
    std::vector<Cam> poses(120);
    for (int k = 0; k < 120; k++) poses[k] = orbit_camera(k, 120, cfg->w, cfg->h, cfg->fx);

    auto step = [&](int k) -> int {
        gsr_ctx* c = ctx[k % in_flight];
        const Cam& p = poses[k % 120];
        if (int rc = gsr_set_camera(c, p.view, p.proj, p.vp, (float)cfg->fx, (float)cfg->fx)) return rc;
        return gsr_render_async(c);
    };
    for (int k = 0; k < warmup; k++) { ctx0 = ctx[k % in_flight]; CHECK(step(k)); }
    for (gsr_ctx* c : ctx) { ctx0 = c; CHECK(gsr_sync(c)); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < frames; k++) { ctx0 = ctx[(warmup + k) % in_flight]; CHECK(step(warmup + k)); }
    for (gsr_ctx* c : ctx) { ctx0 = c; CHECK(gsr_sync(c)); }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // one more frame at pose 0 on context 0 for the cross-check hashes
    ctx0 = ctx[0];
    CHECK(gsr_set_camera(ctx[0], poses[0].view, poses[0].proj, poses[0].vp, (float)cfg->fx, (float)cfg->fx));
    CHECK(gsr_render(ctx[0]));
    std::vector<uint32_t> di(n);
    std::vector<uint8_t> px((size_t)cfg->w * cfg->h * 4);
    CHECK(gsr_read_depth_index(ctx[0], di.data()));
    CHECK(gsr_read_pixels_rgba8(ctx[0], px.data()));
    if (!dump.empty()) {
        FILE* f = std::fopen((dump + ".depth_index.bin").c_str(), "wb");
        if (f) { std::fwrite(di.data(), 4, di.size(), f); std::fclose(f); }
        f = std::fopen((dump + ".rgba8.bin").c_str(), "wb");
        if (f) { std::fwrite(px.data(), 1, px.size(), f); std::fclose(f); }
    }
    char name[128] = "";
    int32_t cus = 0, khz = 0;
    (void)gsr_device_info(ctx[0], name, (int32_t)sizeof name, &cus, &khz);
    std::printf("{\"caller\": \"C++ (tools/bench_cabi.cpp)\", \"config\": \"%s\", \"n\": %u, \"width\": %d, \"height\": %d, "
                "\"frames\": %d, \"warmup\": %d, \"frames_in_flight\": %d, \"frames_per_sec\": %.1f, \"ms_per_frame\": %.4f, "
                "\"rows_fnv1a\": \"%016llx\", \"depth_index_fnv1a\": \"%016llx\", \"rgba8_fnv1a\": \"%016llx\", \"device\": \"%s\", \"compute_units\": %d}\n",
                cfg->name, n, cfg->w, cfg->h, frames, warmup, in_flight, frames / sec, sec / frames * 1e3,
                (unsigned long long)fnv1a(rows.data(), rows.size()), (unsigned long long)fnv1a(di.data(), di.size() * 4),
                (unsigned long long)fnv1a(px.data(), px.size()), name, cus);
    for (gsr_ctx* c : ctx) gsr_destroy(c);
    return 0;
}
