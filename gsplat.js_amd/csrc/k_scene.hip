// On-device scene build and transforms: what Scene.setData / translate / rotate / scale /
// limitBox do in JavaScript loops (src/core/Scene.ts:126-366), as kernels over the SoA scene.  The arithmetic is
// f64 in the reference's order of operations with f32 stores where the reference stores into Float32Arrays, and the
// file is compiled with -ffp-contract=off, so every word matches the JavaScript result bit for bit.
#include "gsr_internal.h"

namespace gsr {

// src/utils.ts:16-43 floatToHalf: truncating; JS `>>` takes the shift count modulo 32
__device__ __forceinline__ uint32_t float_to_half_trunc(double value)
{
    const float fv = (float)value;  // _floatView[0] = float
    const int32_t f = __float_as_int(fv);
    const int32_t sign = (f >> 31) & 1;
    const int32_t exp = (f >> 23) & 0xff;
    int32_t frac = f & 0x007fffff;
    int32_t out_exp;
    if (exp == 0) {
        out_exp = 0;
    } else if (exp < 113) {
        out_exp = 0;
        frac = (frac | 0x00800000) >> ((113 - exp) & 31);
        if (frac & 0x01000000) { out_exp = 1; frac = 0; }
    } else if (exp < 142) {
        out_exp = exp - 112;
    } else {
        out_exp = 31;
        frac = 0;
    }
    return (uint32_t)((sign << 15) | (out_exp << 10) | (frac >> 13));
}

__device__ __forceinline__ uint32_t pack_half2(double x, double y)
{
    return float_to_half_trunc(x) | (float_to_half_trunc(y) << 16);
}

// Scene.ts:150-176: rot = (w, x, y, z) as stored in _rotations, scl = _scales
__device__ __forceinline__ void pack_cov(float4 rot, float4 scl, uint32_t& c0, uint32_t& c1, uint32_t& c2)
{
    const double qx = rot.y, qy = rot.z, qz = rot.w, qw = -(double)rot.x;  // Quaternion(r1, r2, r3, -r0)
    const double R[9] = {1 - 2 * qy * qy - 2 * qz * qz, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw,
                         2 * qx * qy + 2 * qz * qw, 1 - 2 * qx * qx - 2 * qz * qz, 2 * qy * qz - 2 * qx * qw,
                         2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx * qx - 2 * qy * qy};
    const double a[9] = {(double)scl.x, 0, 0, 0, (double)scl.y, 0, 0, 0, (double)scl.z};
    const double* b = R;
    // Matrix3.multiply (Matrix3.ts:33-47), this = Diagonal(scale), m = rot
    const double M[9] = {
        b[0] * a[0] + b[3] * a[1] + b[6] * a[2], b[1] * a[0] + b[4] * a[1] + b[7] * a[2], b[2] * a[0] + b[5] * a[1] + b[8] * a[2],
        b[0] * a[3] + b[3] * a[4] + b[6] * a[5], b[1] * a[3] + b[4] * a[4] + b[7] * a[5], b[2] * a[3] + b[5] * a[4] + b[8] * a[5],
        b[0] * a[6] + b[3] * a[7] + b[6] * a[8], b[1] * a[6] + b[4] * a[7] + b[7] * a[8], b[2] * a[6] + b[5] * a[7] + b[8] * a[8]};
    const double s0 = M[0] * M[0] + M[3] * M[3] + M[6] * M[6];
    const double s1 = M[0] * M[1] + M[3] * M[4] + M[6] * M[7];
    const double s2 = M[0] * M[2] + M[3] * M[5] + M[6] * M[8];
    const double s3 = M[1] * M[1] + M[4] * M[4] + M[7] * M[7];
    const double s4 = M[1] * M[2] + M[4] * M[5] + M[7] * M[8];
    const double s5 = M[2] * M[2] + M[5] * M[5] + M[8] * M[8];
    c0 = pack_half2(4 * s0, 4 * s1);
    c1 = pack_half2(4 * s2, 4 * s3);
    c2 = pack_half2(4 * s4, 4 * s5);
}

// Scene.setData (Scene.ts:126-177): one 32-byte .splat row per thread
__global__ __launch_bounds__(256) void k_build_scene(const uint4* __restrict__ rows, uint32_t n, SceneDev sc)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 a = rows[2 * (size_t)i], b = rows[2 * (size_t)i + 1];  // pos xyz, scale x | scale yz, rgba, rot
    sc.px[i] = __uint_as_float(a.x); sc.py[i] = __uint_as_float(a.y); sc.pz[i] = __uint_as_float(a.z);
    const float4 scl = make_float4(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y), 0.f);
    const uint32_t rb = b.w;
    const float4 rot = make_float4((float)(((double)(rb & 0xffu) - 128) / 128), (float)(((double)((rb >> 8) & 0xffu) - 128) / 128),
                                   (float)(((double)((rb >> 16) & 0xffu) - 128) / 128), (float)(((double)(rb >> 24) - 128) / 128));
    sc.rgba[i] = b.z;
    sc.rot[i] = rot;
    sc.scl[i] = scl;
    uint32_t c0, c1, c2;
    pack_cov(rot, scl, c0, c1, c2);
    sc.cov0[i] = c0; sc.cov1[i] = c1; sc.cov2[i] = c2;
}

// Scene.translate (Scene.ts:182-195)
__global__ __launch_bounds__(256) void k_scene_translate(uint32_t n, SceneDev sc, double tx, double ty, double tz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sc.px[i] = (float)((double)sc.px[i] + tx);
    sc.py[i] = (float)((double)sc.py[i] + ty);
    sc.pz[i] = (float)((double)sc.pz[i] + tz);
}

// Scene.rotate (Scene.ts:197-257), q = (x, y, z, w)
__global__ __launch_bounds__(256) void k_scene_rotate(uint32_t n, SceneDev sc, double x, double y, double z, double w)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double R[9] = {1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                         2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                         2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y};
    const double px = sc.px[i], py = sc.py[i], pz = sc.pz[i];
    sc.px[i] = (float)(R[0] * px + R[1] * py + R[2] * pz);
    sc.py[i] = (float)(R[3] * px + R[4] * py + R[5] * pz);
    sc.pz[i] = (float)(R[6] * px + R[7] * py + R[8] * pz);
    const float4 r = sc.rot[i];  // (w, x, y, z)
    // rotation.multiply(Quaternion(r1, r2, r3, r0)), Quaternion.ts:39-55
    const double w1 = w, x1 = x, y1 = y, z1 = z, w2 = r.x, x2 = r.y, y2 = r.z, z2 = r.w;
    float4 nr;
    nr.y = (float)(w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2);
    nr.z = (float)(w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2);
    nr.w = (float)(w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2);
    nr.x = (float)(w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2);
    sc.rot[i] = nr;
    uint32_t c0, c1, c2;
    pack_cov(nr, sc.scl[i], c0, c1, c2);
    sc.cov0[i] = c0; sc.cov1[i] = c1; sc.cov2[i] = c2;
}

// Scene.scale (Scene.ts:259-305)
__global__ __launch_bounds__(256) void k_scene_scale(uint32_t n, SceneDev sc, double sx, double sy, double sz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sc.px[i] = (float)((double)sc.px[i] * sx);
    sc.py[i] = (float)((double)sc.py[i] * sy);
    sc.pz[i] = (float)((double)sc.pz[i] * sz);
    float4 s = sc.scl[i];
    s.x = (float)((double)s.x * sx); s.y = (float)((double)s.y * sy); s.z = (float)((double)s.z * sz);
    sc.scl[i] = s;
    uint32_t c0, c1, c2;
    pack_cov(sc.rot[i], s, c0, c1, c2);
    sc.cov0[i] = c0; sc.cov1[i] = c1; sc.cov2[i] = c2;
}

// ---- Scene.limitBox (Scene.ts:307-366): order-preserving compaction ----
constexpr int BOX_THREADS = 1024;

__device__ __forceinline__ bool in_box(const SceneDev& sc, uint32_t i, const double* box)
{
    const double x = sc.px[i], y = sc.py[i], z = sc.pz[i];
    return x >= box[0] && x <= box[1] && y >= box[2] && y <= box[3] && z >= box[4] && z <= box[5];
}

struct Box { double v[6]; };

__global__ __launch_bounds__(BOX_THREADS) void k_box_count(uint32_t n, SceneDev sc, Box box, uint32_t* __restrict__ block_count)
{
    __shared__ uint32_t s_w[BOX_THREADS / WAVE];
    const uint32_t i = blockIdx.x * BOX_THREADS + threadIdx.x;
    const bool keep = i < n && in_box(sc, i, box.v);
    const uint64_t m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < BOX_THREADS / WAVE; w++) t += s_w[w];
        block_count[blockIdx.x] = t;
    }
}

// one workgroup: block_count -> exclusive offsets in place, total to *total
__global__ __launch_bounds__(BOX_THREADS) void k_box_scan(uint32_t* __restrict__ block_count, uint32_t nblocks, uint32_t* __restrict__ total)
{
    __shared__ uint32_t s_part[BOX_THREADS];
    const uint32_t per = (nblocks + BOX_THREADS - 1) / BOX_THREADS;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, nblocks);
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) sum += block_count[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < BOX_THREADS; t++) { const uint32_t v = s_part[t]; s_part[t] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t b = b0; b < b1; b++) { const uint32_t v = block_count[b]; block_count[b] = run; run += v; }
}

__global__ __launch_bounds__(BOX_THREADS) void k_box_compact(uint32_t n, SceneDev src, SceneDev dst, Box box,
                                                             const uint32_t* __restrict__ block_off)
{
    __shared__ uint32_t s_w[BOX_THREADS / WAVE];
    const uint32_t i = blockIdx.x * BOX_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool keep = i < n && in_box(src, i, box.v);
    const uint64_t m = __ballot(keep);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = block_off[blockIdx.x];
    for (int w = 0; w < wave; w++) off += s_w[w];
    if (!keep) return;
    const uint32_t o = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    dst.px[o] = src.px[i]; dst.py[o] = src.py[i]; dst.pz[o] = src.pz[i];
    dst.cov0[o] = src.cov0[i]; dst.cov1[o] = src.cov1[i]; dst.cov2[o] = src.cov2[i]; dst.rgba[o] = src.rgba[i];
    dst.rot[o] = src.rot[i]; dst.scl[o] = src.scl[i];
}

// ---- launchers ----
void launch_build_scene(const uint8_t* rows, uint32_t n, const SceneDev& sc, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_build_scene, dim3((n + 255) / 256), dim3(256), 0, s, (const uint4*)rows, n, sc);
}
void launch_scene_translate(uint32_t n, const SceneDev& sc, const double* t, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_scene_translate, dim3((n + 255) / 256), dim3(256), 0, s, n, sc, t[0], t[1], t[2]);
}
void launch_scene_rotate(uint32_t n, const SceneDev& sc, const double* q, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_scene_rotate, dim3((n + 255) / 256), dim3(256), 0, s, n, sc, q[0], q[1], q[2], q[3]);
}
void launch_scene_scale(uint32_t n, const SceneDev& sc, const double* sv, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_scene_scale, dim3((n + 255) / 256), dim3(256), 0, s, n, sc, sv[0], sv[1], sv[2]);
}
void launch_scene_limit_box(uint32_t n, const SceneDev& src, const SceneDev& dst, const double* box, uint32_t* block_count,
                            uint32_t* total, hipStream_t s)
{
    if (!n) return;
    Box b;
    for (int k = 0; k < 6; k++) b.v[k] = box[k];
    const uint32_t nblocks = (n + BOX_THREADS - 1) / BOX_THREADS;
    hipLaunchKernelGGL(k_box_count, dim3(nblocks), dim3(BOX_THREADS), 0, s, n, src, b, block_count);
    hipLaunchKernelGGL(k_box_scan, dim3(1), dim3(BOX_THREADS), 0, s, block_count, nblocks, total);
    hipLaunchKernelGGL(k_box_compact, dim3(nblocks), dim3(BOX_THREADS), 0, s, n, src, dst, b, (const uint32_t*)block_count);
}

}  // namespace gsr
