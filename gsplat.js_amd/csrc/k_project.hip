// Per-splat stage: scene repack (once per scene) and the fused
// depth-key + min/max + projection kernel (once per frame).
//
// Compiled with -ffp-contract=off: every f32 operation below is a single
// correctly rounded IEEE operation in the written order, so the depth key is
// bit-identical to wasm/wasm.cpp:14-31 and every projected field is
// bit-identical to the CPU restatement of vertex.glsl.ts:130-231.
#include "gsr_internal.h"

namespace gsr {

// ---------------------------------------------------------------------------
// Scene.data (AoS, 8 u32 per splat: the RGBA32UI texel pair of
// WebGLRenderer.ts:185-195) -> SoA arrays, so the per-frame kernels read
// 28 B/splat fully coalesced instead of 32-B gathers.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_repack_scene(const uint4* __restrict__ data, const float* __restrict__ positions,
                                                      uint32_t n, float* __restrict__ px, float* __restrict__ py,
                                                      float* __restrict__ pz, uint32_t* __restrict__ cov0,
                                                      uint32_t* __restrict__ cov1, uint32_t* __restrict__ cov2,
                                                      uint32_t* __restrict__ rgba, uint32_t* __restrict__ mismatch)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a = data[2 * (size_t)i], b = data[2 * (size_t)i + 1];
    float x = positions[3 * (size_t)i], y = positions[3 * (size_t)i + 1], z = positions[3 * (size_t)i + 2];
    if (__float_as_uint(x) != a.x || __float_as_uint(y) != a.y || __float_as_uint(z) != a.z) atomicOr(mismatch, 1u);
    px[i] = x; py[i] = y; pz[i] = z;
    cov0[i] = b.x; cov1[i] = b.y; cov2[i] = b.z; rgba[i] = b.w;
}

void launch_repack_scene(const uint32_t* data, const float* positions, uint32_t n, float* px, float* py, float* pz,
                         uint32_t* cov0, uint32_t* cov1, uint32_t* cov2, uint32_t* rgba, uint32_t* mismatch, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_repack_scene, dim3((n + 255) / 256), dim3(256), 0, s, (const uint4*)data, positions, n, px, py,
                       pz, cov0, cov1, cov2, rgba, mismatch);
}

// positions only (the wasm drop-in gsplat_sort_host: Worker.ts:26-27 copies nothing else): AoS xyz -> SoA
__global__ __launch_bounds__(256) void k_repack_positions(const float* __restrict__ positions, uint32_t n, float* __restrict__ px,
                                                          float* __restrict__ py, float* __restrict__ pz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    px[i] = positions[3 * (size_t)i]; py[i] = positions[3 * (size_t)i + 1]; pz[i] = positions[3 * (size_t)i + 2];
}

void launch_repack_positions(const float* positions, uint32_t n, float* px, float* py, float* pz, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_repack_positions, dim3((n + 255) / 256), dim3(256), 0, s, positions, n, px, py, pz);
}

// ---------------------------------------------------------------------------
// helpers (column-major 3x3, GLSL conventions: m[c*3+r])
// ---------------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* r)
{
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int ro = 0; ro < 3; ro++) {
            float s = a[0 * 3 + ro] * b[c * 3 + 0];
            s = s + a[1 * 3 + ro] * b[c * 3 + 1];
            s = s + a[2 * 3 + ro] * b[c * 3 + 2];
            r[c * 3 + ro] = s;
        }
}

__device__ __forceinline__ float half_bits_to_float(uint32_t h)
{
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu));  // v_cvt_f32_f16: exact
}

__device__ __forceinline__ bool finite4(float a, float b, float c, float d)
{
    return isfinite(a) && isfinite(b) && isfinite(c) && isfinite(d);
}

// ---------------------------------------------------------------------------
// SH colour, vertex.glsl.ts:57-104 (eval_sh_rgb) in f32, products and sums in the written order.
// k[0..15]: the 16 coefficients of one colour channel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float eval_sh_channel(const float* k, uint32_t deg, float x, float y, float z)
{
    constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                    C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
    constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                    C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                    C3_6 = -0.5900435899266435f;
    float r = C0 * k[0];
    if (deg > 0) {
        r = r - ((((C1 * y) * k[1]) + ((C1 * z) * k[2])) - ((C1 * x) * k[3]));
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            float s2 = (C2_0 * xy) * k[4];
            s2 = s2 + (C2_1 * yz) * k[5];
            s2 = s2 + (C2_2 * ((2.0f * zz - xx) - yy)) * k[6];
            s2 = s2 + (C2_3 * xz) * k[7];
            s2 = s2 + (C2_4 * (xx - yy)) * k[8];
            r = r + s2;
            if (deg > 2) {
                float s3 = ((C3_0 * y) * (3.0f * xx - yy)) * k[9];
                s3 = s3 + ((C3_1 * xy) * z) * k[10];
                s3 = s3 + ((C3_2 * y) * ((4.0f * zz - xx) - yy)) * k[11];
                s3 = s3 + ((C3_3 * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy)) * k[12];
                s3 = s3 + ((C3_4 * x) * ((4.0f * zz - xx) - yy)) * k[13];
                s3 = s3 + ((C3_5 * z) * (xx - yy)) * k[14];
                s3 = s3 + ((C3_6 * x) * (xx - 3.0f * yy)) * k[15];
                r = r + s3;
            }
        }
    }
    r = r + 0.5f;
    r = (r > 0.0f) ? r : 0.0f;  // :103 max(result, 0)
    return (r < 1.0f) ? r : 1.0f;  // :200 min(rgb, 1)
}

__device__ __forceinline__ float eval_sh_texture(const uint32_t* __restrict__ tex, uint32_t t, uint32_t deg, float x, float y, float z)
{
    const uint4 a = reinterpret_cast<const uint4*>(tex)[2 * (size_t)t], b = reinterpret_cast<const uint4*>(tex)[2 * (size_t)t + 1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float k[16];
#pragma unroll
    for (int j = 0; j < 8; j++) { k[2 * j] = half_bits_to_float(w[j]); k[2 * j + 1] = half_bits_to_float(w[j] >> 16); }
    return eval_sh_channel(k, deg, x, y, z);
}

// ---------------------------------------------------------------------------
// One thread per splat, original order (fully coalesced SoA reads).
//   A1  depth key + min/max        wasm/wasm.cpp:14-31
//   B1-B6 projection               vertex.glsl.ts:130-231 (non-SH colour branch, scalingFactor 1)
// Writes depth[i] and the packed bin rectangle rect[i] always; rec[i] for splats that survive the culls.  The inclusive pixel
// box bbox[i] (x0 > x1 marks "nothing to draw") is a parity read-back, not an input of any kernel: it is written only when
// asked for (bbox != null).  gsr_read_records asks by running this kernel once more for the frame's camera with
// do_project == 2, which writes records and boxes only -- no key, no rectangle, no frame counters -- so a frame does not pay
// 8 bytes per splat for a buffer nothing on the path reads (C4: 40 MB of the projection's ~400).
// ---------------------------------------------------------------------------
// The camera is a kernel argument (scalar loads from the kernarg segment).  It is the one argument of the frame's chain that
// changes from frame to frame: when the chain is replayed as a HIP graph, the host rewrites this kernel node's parameters
// (hipGraphExecKernelNodeSetParams, gsr_api.cpp) -- there is no kernel in front of the frame that would carry the camera to
// device memory and reset the frame's words.  What such a kernel did is done where it costs nothing: the frame slots are
// reset by their last reader (the finalize step, k_bin.hip), the overflow word here.
__global__ __launch_bounds__(PROJ_THREADS) void k_project_key(SceneSoA sc, uint32_t n, CamParams cam,
                                                              int do_project, int32_t* __restrict__ depth,
                                                              int32_t* __restrict__ slots,
                                                              Record* __restrict__ rec, uint2* __restrict__ bbox,
                                                              uint32_t* __restrict__ rect, uint32_t* __restrict__ overflow,
                                                              uint32_t* __restrict__ kept, uint8_t* __restrict__ kept_lane)
{
    const bool frame = do_project != 2;   // (uniform) false: the read-back re-run
    // Band mode (a context that composites a column band of the screen: multi-GPU) with `kept` set: most splats
    // are somebody else's, and everything after this kernel wants the band's survivors only.  The workgroup packs its
    // survivors' depths to the front of ITS OWN 256 slots of depth[], in index order, with the lane each came from in
    // kept_lane[] and their number in kept[workgroup]; the key pass and the first radix pass read kept[] and touch only those
    // slots (k_sort.hip).  Nothing is written for the others: 12 bytes read per splat that is not the band's, instead of
    // 12 read + 8 written here and 12 + 4 + 4 in the two passes behind.  The rectangle stays at the splat's own index.
    const bool pack = frame && kept != nullptr;   // (uniform)
    if (frame && blockIdx.x == 0 && threadIdx.x == 0) *overflow = 0u;   // (raised by the binning kernels later in this frame)
    __shared__ int32_t s_min[4], s_max[4];
    __shared__ uint32_t s_vis[4], s_til[4], s_oti[4], s_oma[4], s_keep[4];
    int32_t mydepth = 0;
    bool keep = false;
    int32_t dmin = 0x7fffffff, dmax = (int32_t)0x80000000;
    uint32_t vis = 0, tiles = 0;   // this thread's visible splats and the 16x16 tiles their boxes overlap (V and D of the byte model)
    uint32_t otiles = 0;           // sum of opacity byte x tiles / 16 over them
    uint32_t omass = 0;            // sum of opacity byte x footprint in pixels / 256 over them (below)
    const int bx_lo = cam.band_px0 / BIN_PX, bx_hi = (cam.band_px1 + BIN_PX - 1) / BIN_PX;   // this context's band of bin columns

    const uint32_t i = blockIdx.x * PROJ_THREADS + threadIdx.x;
    if (i < n) {
        const float x = sc.px[i], y = sc.py[i], z = sc.pz[i];
        // ---- A1: three separate f32 multiplies, left-to-right adds, *4096 in f32, truncate ----
        const float f0 = cam.vp2 * x;
        const float f1 = cam.vp6 * y;
        const float f2 = cam.vp10 * z;
        const int32_t d = (int32_t)(((f0 + f1) + f2) * 4096.0f);
        if (frame && !pack) depth[i] = d;
        mydepth = d;
        dmin = d; dmax = d;

        if (do_project) {
            uint2 bb = make_uint2(BBOX_INVISIBLE_X, BBOX_INVISIBLE_Y);
            uint32_t op8 = 0;   // the splat's opacity byte (for the frame's optical-depth figure below)
            float area = 0.0f;  // its footprint: the integral of exp(-|vPosition|^2) over the |vPosition| <= 2 ellipse, in pixels
            do {
                // :133-136  cam = view * vec4(p,1); pos2d = projection * cam
                float camv[4], pos2d[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float s = cam.view[0 * 4 + r] * x;
                    s = s + cam.view[1 * 4 + r] * y;
                    s = s + cam.view[2 * 4 + r] * z;
                    s = s + cam.view[3 * 4 + r];
                    camv[r] = s;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float s = cam.proj[0 * 4 + r] * camv[0];
                    s = s + cam.proj[1 * 4 + r] * camv[1];
                    s = s + cam.proj[2 * 4 + r] * camv[2];
                    s = s + cam.proj[3 * 4 + r] * camv[3];
                    pos2d[r] = s;
                }
                // :138-142
                const float clip = 1.2f * pos2d[3];
                if (pos2d[2] < -pos2d[3] || pos2d[0] < -clip || pos2d[0] > clip || pos2d[1] < -clip || pos2d[1] > clip) break;

                // :144-146
                const uint32_t c0 = sc.cov0[i], c1 = sc.cov1[i], c2 = sc.cov2[i], cw = sc.rgba[i];
                const float u1x = half_bits_to_float(c0), u1y = half_bits_to_float(c0 >> 16);
                const float u2x = half_bits_to_float(c1), u2y = half_bits_to_float(c1 >> 16);
                const float u3x = half_bits_to_float(c2), u3y = half_bits_to_float(c2 >> 16);
                const float Vrk[9] = {u1x, u1y, u2x, u1y, u2y, u3x, u2x, u3x, u3y};
                // :148-152
                const float zz = camv[2] * camv[2];
                const float J[9] = {cam.fx / camv[2], 0.f, -(cam.fx * camv[0]) / zz,
                                    0.f, -cam.fy / camv[2], (cam.fy * camv[1]) / zz,
                                    0.f, 0.f, 0.f};
                // :154-155  T = transpose(mat3(view)) * J; cov2d = transpose(T) * Vrk * T
                const float V3t[9] = {cam.view[0], cam.view[4], cam.view[8], cam.view[1], cam.view[5],
                                      cam.view[9], cam.view[2], cam.view[6], cam.view[10]};
                float T[9], Tt[9], TtV[9], cov2d[9];
                mat3_mul(V3t, J, T);
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int r = 0; r < 3; r++) Tt[c * 3 + r] = T[r * 3 + c];
                mat3_mul(Tt, Vrk, TtV);
                mat3_mul(TtV, T, cov2d);
                // :158-163
                const float a = cov2d[0] + 0.3f;
                const float b = cov2d[1];
                const float c = cov2d[4] + 0.3f;
                const float det = a * c - b * b;
                if (det == 0.0f) break;
                // :166-171
                const float mid = (a + c) / 2.0f;
                const float rad = mid * mid - det;
                const float sq = sqrtf((0.1f < rad) ? rad : 0.1f);
                const float lambda1 = mid + sq;
                const float lambda2 = mid - sq;
                if (lambda2 < 0.0f) break;
                // :173-175
                const float dvx = b, dvy = lambda1 - a;
                const float len = sqrtf(dvx * dvx + dvy * dvy);
                const float dgx = dvx / len, dgy = dvy / len;
                float smaj = sqrtf(2.0f * lambda1), smin = sqrtf(2.0f * lambda2);
                smaj = (1024.0f < smaj) ? 1024.0f : smaj;
                smin = (1024.0f < smin) ? 1024.0f : smin;
                float majx = smaj * dgx, majy = smaj * dgy;
                float minx = smin * dgy, miny = smin * -dgx;
                if (cam.use_fade) {  // :216-223, FadeInPass: axes scaled by the depth-dependent factor
                    const float depthNorm = (pos2d[2] / pos2d[3] + 1.0f) / 2.0f;
                    const float nearv = 0.1f, farv = 100.0f;
                    const float normalizedDepth = (2.0f * nearv) / (farv + nearv - depthNorm * (farv - nearv));
                    float st = normalizedDepth - 0.1f; st = (st > 0.0f) ? st : 0.0f;
                    float en = normalizedDepth + 0.1f; en = (en < 1.0f) ? en : 1.0f;
                    float sf = (cam.fade - st) / (en - st);
                    sf = (sf < 0.0f) ? 0.0f : ((sf > 1.0f) ? 1.0f : sf);
                    if (!(sf > 0.0f)) break;  // zero-area quad: nothing drawn
                    majx = majx * sf; majy = majy * sf; minx = minx * sf; miny = miny * sf;
                }
                if (!finite4(majx, majy, minx, miny)) break;  // normalize(0,0) -> NaN: splat dropped
                // :177-178: opacity; colour stays packed (:207 divides by 255 at composite time)
                const float opacity = (float)((cw >> 24) & 0xffu) / 255.0f;
                op8 = (cw >> 24) & 0xffu;
                // :226-229 + viewport transform, image rows top-down
                const float vcx = pos2d[0] / pos2d[3], vcy = pos2d[1] / pos2d[3];
                const float cx = ((vcx + 1.0f) * 0.5f) * (float)cam.W;
                const float cy = ((1.0f - vcy) * 0.5f) * (float)cam.H;
                const float m2 = majx * majx + majy * majy;
                const float n2 = minx * minx + miny * miny;
                const float im = 2.0f / m2, in = 2.0f / n2;
                // pi (1 - e^-4) / 4 x |major| x |minor|: what the splat adds to the optical depth of the pixels it covers, summed over
                // them, per unit of opacity (an estimate for the work-item policy, k_bin_finalize: not a parity quantity)
                area = fminf(0.77f * __builtin_amdgcn_sqrtf(m2 * n2), 65535.0f);
                Record r;
                r.cx = cx; r.cy = cy;
                r.ux = majx * im; r.uy = -majy * im;
                r.wx = minx * in; r.wy = -miny * in;
                if (!(finite4(r.ux, r.uy, r.wx, r.wy) && isfinite(cx) && isfinite(cy))) break;
                r.la = __log2f(opacity);
                r.rgb8 = cw & 0x00ffffffu;
                if (cam.sh_on && (int32_t)i > cam.band[0]) {  // vertex.glsl.ts:180-204
                    const uint32_t deg = (int32_t)i > cam.band[1] ? ((int32_t)i > cam.band[2] ? 3u : 2u) : 1u;
                    // inverse(view)[3].xyz for the rigid view matrix [A | b]: -A^T b
                    float cp[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        float t = cam.view[q * 4 + 0] * cam.view[12];
                        t = t + cam.view[q * 4 + 1] * cam.view[13];
                        t = t + cam.view[q * 4 + 2] * cam.view[14];
                        cp[q] = -t;
                    }
                    const float dvx = x - cp[0], dvy = y - cp[1], dvz = z - cp[2];
                    const float dl = sqrtf((dvx * dvx + dvy * dvy) + dvz * dvz);
                    const float dxn = dvx / dl, dyn = dvy / dl, dzn = dvz / dl;
                    const uint32_t t = i - (uint32_t)(cam.band[0] + 1);
                    sc.shcol[i] = make_float4(eval_sh_texture(sc.sh_r, t, deg, dxn, dyn, dzn), eval_sh_texture(sc.sh_g, t, deg, dxn, dyn, dzn),
                                              eval_sh_texture(sc.sh_b, t, deg, dxn, dyn, dzn), 0.0f);
                    r.rgb8 = RGB8_IN_SHCOL;
                }
                // bounding box of the |vPosition| <= 2 ellipse, pixel centres at +0.5
                const float ex = sqrtf(majx * majx + minx * minx);
                const float ey = sqrtf(majy * majy + miny * miny);
                float fx0 = floorf(cx - ex - 0.5f), fx1 = ceilf(cx + ex - 0.5f);
                float fy0 = floorf(cy - ey - 0.5f), fy1 = ceilf(cy + ey - 0.5f);
                fx0 = fmaxf(fx0, 0.0f); fy0 = fmaxf(fy0, 0.0f);
                fx1 = fminf(fx1, (float)(cam.W - 1)); fy1 = fminf(fy1, (float)(cam.H - 1));
                // a splat whose box misses this context's column band (multi-GPU split) is somebody else's: no record
                if (fx0 <= fx1 && (fx1 < (float)cam.band_px0 || fx0 >= (float)cam.band_px1)) break;
                // the record is written even when the bbox is empty (parity read-back)
                float4* rp = reinterpret_cast<float4*>(rec + i);
                rp[0] = make_float4(r.cx, r.cy, r.ux, r.uy);
                rp[1] = make_float4(r.wx, r.wy, r.la, __uint_as_float(r.rgb8));
                if (fx0 > fx1 || fy0 > fy1) break;
                bb.x = (uint32_t)(int32_t)fx0 | ((uint32_t)(int32_t)fx1 << 16);
                bb.y = (uint32_t)(int32_t)fy0 | ((uint32_t)(int32_t)fy1 << 16);
            } while (0);
            if (bbox) bbox[i] = bb;
            if (frame) {
                const uint32_t rv = pack_bin_rect(bb.x, bb.y, bx_lo, bx_hi);
                keep = rv != RECT_NONE;
                if (!pack || keep) rect[i] = rv;
            }
            if ((bb.x & 0xffffu) <= (bb.x >> 16)) {
                vis++;
                const int tx0 = max((int)(bb.x & 0xffffu) / TILE, bx_lo * BIN_TILES), tx1 = min((int)(bb.x >> 16) / TILE, bx_hi * BIN_TILES - 1);
                const uint32_t nt = (uint32_t)((tx1 - tx0 + 1) * ((int)(bb.y >> 16) / TILE - (int)(bb.y & 0xffffu) / TILE + 1));
                tiles += nt;
                // opacity x tiles, /16 so that a slot's sum stays inside 32 bits (<= 255 * 4096 / 16 per splat): the frame's
                // optical depth in the unit k_bin_finalize compares (FinalizeArgs::long_tau)
                otiles += (op8 * min(nt, 4096u)) >> 4;
                omass += (op8 * (uint32_t)area) >> 8;   // (<= 65280 per splat: a slot's sum stays inside 32 bits)
            }
        }
    }

    // ---- workgroup totals, folded into this workgroup's frame slot (see FRAME_SLOTS in gsr_internal.h) ----
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dmin = min(dmin, __shfl_xor(dmin, off));
        dmax = max(dmax, __shfl_xor(dmax, off));
        vis += __shfl_xor(vis, off);
        tiles += __shfl_xor(tiles, off);
        otiles += __shfl_xor(otiles, off);
        omass += __shfl_xor(omass, off);
    }
    const int wave = threadIdx.x >> 6;
    uint32_t keep_rank = 0;
    if (pack) {
        const unsigned long long kb = __ballot(keep);
        keep_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(kb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kb, 0u));
        if ((threadIdx.x & 63) == 0) s_keep[wave] = (uint32_t)__popcll(kb);
    }
    if ((threadIdx.x & 63) == 0) { s_min[wave] = dmin; s_max[wave] = dmax; s_vis[wave] = vis; s_til[wave] = tiles; s_oti[wave] = otiles; s_oma[wave] = omass; }
    __syncthreads();
    if (pack) {
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += s_keep[w];
        if (keep) {
            const uint32_t slot = blockIdx.x * PROJ_THREADS + before + keep_rank;   // <= my own index: inside the scene
            depth[slot] = mydepth;
            kept_lane[slot] = (uint8_t)threadIdx.x;
        }
        if (threadIdx.x == 0) kept[blockIdx.x] = s_keep[0] + s_keep[1] + s_keep[2] + s_keep[3];
    }
    if (threadIdx.x == 0 && frame) {
        int32_t* slot = slots + (size_t)(blockIdx.x & (FRAME_SLOTS - 1)) * FRAME_SLOT_WORDS;
        atomicMin(&slot[0], min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3])));
        atomicMax(&slot[1], max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
        if (do_project) {
            atomicAdd(reinterpret_cast<uint32_t*>(&slot[2]), s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3]);
            atomicAdd(reinterpret_cast<uint32_t*>(&slot[3]), s_til[0] + s_til[1] + s_til[2] + s_til[3]);
            atomicAdd(reinterpret_cast<uint32_t*>(&slot[4]), s_oti[0] + s_oti[1] + s_oti[2] + s_oti[3]);
            atomicAdd(reinterpret_cast<uint32_t*>(&slot[5]), s_oma[0] + s_oma[1] + s_oma[2] + s_oma[3]);
        }
    }
}

// The depth key alone (A1, wasm/wasm.cpp:14-31): what gsr_sort -- the reference worker's whole job -- needs of the pass above.
// KEY_PER_THREAD splats per thread, so that a 1 M-splat scene is 977 workgroups instead of 3907 one-splat-per-thread ones
// whose dispatch, not their 16 bytes per splat, set the pass's time (12.1 us at 1 M).
constexpr int KEY_PER_THREAD = 4;
// It takes the three matrix entries it needs by value and owns its frame slots: sort-only frames alternate between two sets,
// and the first workgroup of one frame resets the set of the next (its last readers -- the frame before -- are done: one
// stream), so that no k_begin_frame launch stands in front of the reference worker's job.
__global__ __launch_bounds__(PROJ_THREADS) void k_depth_key(SceneSoA sc, uint32_t n, float vp2, float vp6, float vp10,
                                                            int32_t* __restrict__ depth, int32_t* __restrict__ slots,
                                                            int32_t* __restrict__ slots_next)
{
    struct { float vp2, vp6, vp10; } cam = {vp2, vp6, vp10};
    if (blockIdx.x == 0 && threadIdx.x < FRAME_SLOTS) {
        slots_next[(size_t)threadIdx.x * FRAME_SLOT_WORDS + 0] = 0x7fffffff;
        slots_next[(size_t)threadIdx.x * FRAME_SLOT_WORDS + 1] = (int32_t)0x80000000;
    }
    __shared__ int32_t s_min[PROJ_THREADS / WAVE], s_max[PROJ_THREADS / WAVE];
    int32_t dmin = 0x7fffffff, dmax = (int32_t)0x80000000;
    float x[KEY_PER_THREAD], y[KEY_PER_THREAD], z[KEY_PER_THREAD];
#pragma unroll
    for (int k = 0; k < KEY_PER_THREAD; k++) {
        const uint32_t i = (blockIdx.x * KEY_PER_THREAD + k) * PROJ_THREADS + threadIdx.x;
        const bool in = i < n;
        x[k] = in ? sc.px[i] : 0.0f; y[k] = in ? sc.py[i] : 0.0f; z[k] = in ? sc.pz[i] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < KEY_PER_THREAD; k++) {
        const uint32_t i = (blockIdx.x * KEY_PER_THREAD + k) * PROJ_THREADS + threadIdx.x;
        if (i < n) {
            // three separate f32 multiplies, left-to-right adds, *4096 in f32, truncate (as in k_project_key)
            const float f0 = cam.vp2 * x[k];
            const float f1 = cam.vp6 * y[k];
            const float f2 = cam.vp10 * z[k];
            const int32_t d = (int32_t)(((f0 + f1) + f2) * 4096.0f);
            depth[i] = d;
            dmin = min(dmin, d); dmax = max(dmax, d);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dmin = min(dmin, __shfl_xor(dmin, off));
        dmax = max(dmax, __shfl_xor(dmax, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_min[wave] = dmin; s_max[wave] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t* slot = slots + (size_t)(blockIdx.x & (FRAME_SLOTS - 1)) * FRAME_SLOT_WORDS;
        int32_t mn = s_min[0], mx = s_max[0];
        for (int w = 1; w < PROJ_THREADS / WAVE; w++) { mn = min(mn, s_min[w]); mx = max(mx, s_max[w]); }
        atomicMin(&slot[0], mn);
        atomicMax(&slot[1], mx);
    }
}

void launch_depth_key(const SceneSoA& sc, uint32_t n, const CamParams& cam, int32_t* depth, int32_t* slots, int32_t* slots_next, hipStream_t s)
{
    if (!n) return;
    const uint32_t per = PROJ_THREADS * KEY_PER_THREAD;
    hipLaunchKernelGGL(k_depth_key, dim3((n + per - 1) / per), dim3(PROJ_THREADS), 0, s, sc, n, cam.vp2, cam.vp6, cam.vp10, depth, slots, slots_next);
}

void ProjectLaunch::bind()
{
    void* p[] = {&sc, &n, &cam, &do_project, &depth, &slots, &rec, &bbox, &rect, &overflow, &kept, &kept_lane};
    static_assert(sizeof p == sizeof ptrs, "one pointer per kernel argument");
    for (size_t k = 0; k < sizeof p / sizeof p[0]; k++) ptrs[k] = p[k];
}
const void* project_key_kernel() { return (const void*)k_project_key; }
dim3 project_key_grid(uint32_t n) { return dim3((n + PROJ_THREADS - 1) / PROJ_THREADS); }

void launch_project_key(ProjectLaunch& a, hipStream_t s)
{
    if (!a.n) return;
    a.bind();
    (void)hipLaunchKernel(project_key_kernel(), project_key_grid(a.n), dim3(PROJ_THREADS), a.ptrs, 0, s);
}

// Initialisation of a context's frame words and of one set of frame slots (minDepth / maxDepth to the values
// wasm/wasm.cpp:14-15 starts from, everything else to zero); the by-value camera goes to a device slot.  Once per context
// and slot set: the frames keep their words clean themselves (k_project_key, k_depth_key, the finalize step).
__global__ void k_begin_frame(CamParams cam, CamParams* __restrict__ dst, uint32_t* __restrict__ frame_words, uint32_t nwords,
                              int32_t* __restrict__ slots)
{
    // the frame slots: min <- INT_MAX, max <- INT_MIN (the values wasm/wasm.cpp:14-15 starts from), the counters <- 0
    for (uint32_t t = threadIdx.x; t < (uint32_t)FRAME_SLOTS * 8; t += blockDim.x) {
        const uint32_t k = t & 7u;
        slots[(size_t)(t >> 3) * FRAME_SLOT_WORDS + k] = k == 0 ? 0x7fffffff : k == 1 ? (int32_t)0x80000000 : 0;
    }
    constexpr uint32_t WORDS = sizeof(CamParams) / 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&cam);
    for (uint32_t w = threadIdx.x; w < WORDS; w += blockDim.x) reinterpret_cast<uint32_t*>(dst)[w] = src[w];
    for (uint32_t w = threadIdx.x; w < nwords; w += blockDim.x) frame_words[w] = w == 0 ? 0x7fffffffu : w == 1 ? 0x80000000u : 0u;
}

void launch_begin_frame(const CamParams& cam, CamParams* dst, uint32_t* frame_words, uint32_t nwords, int32_t* slots, hipStream_t s)
{
    static_assert(sizeof(CamParams) % 4 == 0, "camera is copied word by word");
    hipLaunchKernelGGL(k_begin_frame, dim3(1), dim3(256), 0, s, cam, dst, frame_words, nwords, slots);
}

}  // namespace gsr
