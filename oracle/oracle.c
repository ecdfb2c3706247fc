/*
 * oracle.c -- CPU restatement of the gsplat.js per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gsplat.js_amd/,
 * include/) may link, load or call this file.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() use it, as the checker.
 *
 * Parity status
 *   sort path   (orc_sort)        PINNED: checked bit-for-bit against the
 *                                 reference's own wasm/wasm.cpp compiled from
 *                                 source (oracle/_ref, tests/test_oracle_ref.py)
 *                                 and against tests/golden/ fixtures generated
 *                                 from it (tests/golden/make_golden.py).
 *   scene pack / build / transforms (orc_scene_*)
 *                                 PINNED by fixtures produced by running the
 *                                 reference's own TypeScript under Node
 *                                 (tests/golden/make_golden_half.js,
 *                                 make_golden_host.js; tests/test_host_golden.py).
 *   render path (orc_project*, the fragment weight of orc_render, orc_eval_sh)
 *                                 PINNED BY EXECUTION OF THE SHADER TEXT, under
 *                                 stated float semantics: the reference renders
 *                                 with GLSL on a WebGL2 context, which cannot run
 *                                 in this container (no GL, no GPU), and its repo
 *                                 holds no test images.  tests/golden/
 *                                 make_golden_shader.py reads vertex.glsl.ts /
 *                                 frag.glsl.ts where they lie, translates the GLSL
 *                                 statement by statement (glsl_eval.py) and RUNS
 *                                 it in IEEE binary32 (left-to-right evaluation,
 *                                 no contraction, the rules DESIGN.md 4 states);
 *                                 tests/test_shader_golden.py: the quad's axes,
 *                                 centre, colour, opacity and gl_Position of every
 *                                 sample splat, every exit of the shader and
 *                                 eval_sh_rgb are reproduced BIT FOR BIT, the
 *                                 fragment colour to f32 rounding.  What remains
 *                                 unpinned, structurally: a GPU's own rounding of
 *                                 GLSL's divisions, square roots and exp, and the
 *                                 rasteriser / RGBA8 blend (orc_render's
 *                                 accumulation).  Also kept: tests/
 *                                 independent_math.py, a float64 statement of the
 *                                 MATHEMATICS (numpy linear algebra, no shared
 *                                 operation sequence with this file).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off is REQUIRED: FMA contraction changes the sort result
 * (SURVEY.md section 8(c)).
 *
 * All citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_DEPTH_RANGE 65536u /* wasm/wasm.cpp:33  (256*256) */

/* ------------------------------------------------------------------------
 * A1-A4: depth key, min/max, 16-bit quantise, stable counting sort.
 * Follows wasm/wasm.cpp:8-52 statement by statement.
 *
 * Defined semantics for the reference's max-bucket overflow:
 * the key domain is [0, 65536] (17 bits); splats whose key is 65536 go LAST,
 * in ascending original index.  The unmodified reference produces exactly
 * this when starts[65536] is preset to N - #{q==65536} (oracle/ref_driver.c
 * does that); here the counting arrays simply have 65537 entries.
 * Degenerate maxDepth==minDepth (reference: 0*inf = NaN, undefined cast):
 * every key is 0, identity permutation.
 *
 * keys_out (nullable): the 17-bit key of every splat.
 * minmax   (nullable): {minDepth, maxDepth}.
 * ---------------------------------------------------------------------- */
int orc_sort(const float *vp, uint32_t n, const float *pos,
             uint32_t *depth_index, uint32_t *keys_out, int32_t *minmax)
{
    uint32_t *depth = (uint32_t *)malloc((size_t)(n ? n : 1) * 4);
    uint32_t *counts = (uint32_t *)calloc(ORC_DEPTH_RANGE + 1, 4);
    uint32_t *starts = (uint32_t *)malloc((ORC_DEPTH_RANGE + 1) * 4);
    if (!depth || !counts || !starts) { free(depth); free(counts); free(starts); return -1; }

    /* wasm.cpp:14-31 */
    int32_t minDepth = 0x7fffffff;
    int32_t maxDepth = (int32_t)0x80000000;
    for (uint32_t i = 0; i < n; i++) {
        float f0 = vp[2] * pos[3 * i + 0];
        float f1 = vp[6] * pos[3 * i + 1];
        float f2 = vp[10] * pos[3 * i + 2];
        int32_t d = (int32_t)((f0 + f1 + f2) * 4096);
        depth[i] = (uint32_t)d;
        if (d > maxDepth) maxDepth = d;
        if (d < minDepth) minDepth = d;
    }
    if (minmax) { minmax[0] = minDepth; minmax[1] = maxDepth; }

    /* wasm.cpp:33-40 */
    if (n && maxDepth != minDepth) {
        const float depthInv = (float)ORC_DEPTH_RANGE / (maxDepth - minDepth);
        for (uint32_t i = 0; i < n; i++) {
            depth[i] = (uint32_t)((depth[i] - (uint32_t)minDepth) * depthInv);
            if (depth[i] > ORC_DEPTH_RANGE) depth[i] = ORC_DEPTH_RANGE; /* never taken for in-range data */
            counts[depth[i]]++;
        }
    } else {
        for (uint32_t i = 0; i < n; i++) depth[i] = 0;
        counts[0] = n;
    }

    /* wasm.cpp:42-46, extended by the one extra bucket */
    starts[0] = 0;
    for (uint32_t i = 1; i <= ORC_DEPTH_RANGE; i++) starts[i] = starts[i - 1] + counts[i - 1];

    /* wasm.cpp:48-51 */
    for (uint32_t i = 0; i < n; i++) depth_index[starts[depth[i]]++] = i;

    if (keys_out) memcpy(keys_out, depth, (size_t)n * 4);
    free(depth); free(counts); free(starts);
    return 0;
}

/* ------------------------------------------------------------------------
 * D2: Scene.setData -- .splat rows -> scene.data (8 u32 per splat) + positions.
 * Follows src/core/Scene.ts:126-177 in f64 (JS number) arithmetic, with
 * src/math/Matrix3.ts:33-47,63-80 and src/utils.ts:16-48.
 * ---------------------------------------------------------------------- */

/* src/utils.ts:16-43 floatToHalf: truncating, with JS's ">>" taking the shift
 * count modulo 32 for f32 exponents below 81. */
uint32_t orc_float_to_half(double value)
{
    float fv = (float)value; /* _floatView[0] = float (RNE to f32) */
    int32_t f;
    memcpy(&f, &fv, 4);
    int32_t sign = (f >> 31) & 0x0001;
    int32_t exp = (f >> 23) & 0x00ff;
    int32_t frac = f & 0x007fffff;
    int32_t newExp;
    if (exp == 0) {
        newExp = 0;
    } else if (exp < 113) {
        newExp = 0;
        frac |= 0x00800000;
        frac = frac >> ((113 - exp) & 31);
        if (frac & 0x01000000) { newExp = 1; frac = 0; }
    } else if (exp < 142) {
        newExp = exp - 112;
    } else {
        newExp = 31;
        frac = 0;
    }
    return (uint32_t)((sign << 15) | (newExp << 10) | (frac >> 13));
}

/* src/utils.ts:46-48 */
uint32_t orc_pack_half2x16(double x, double y)
{
    return (orc_float_to_half(x) | (orc_float_to_half(y) << 16));
}

/* Scene.ts:108-124: shs holds 48 floats per SH-carrying splat, laid out (coefficient k, channel c) at 3k+c; texture c,
 * word j of a splat packs coefficients 2j (low half) and 2j+1 (high half) of channel c. */
void orc_scene_pack_sh(const float *shs, uint32_t count, uint32_t *sh_r, uint32_t *sh_g, uint32_t *sh_b)
{
    uint32_t *out[3] = { sh_r, sh_g, sh_b };
    for (uint32_t i = 0; i < count; i++) {
        uint32_t ind = i * 48;
        for (int j = 0; j < 8; j++) {
            for (int c = 0; c < 3; c++) out[c][8 * (size_t)i + j] = orc_pack_half2x16(shs[ind + c], shs[ind + 3 + c]);
            ind += 6;
        }
    }
}

/* 4*Sigma of one splat from its stored rotation (w,x,y,z as f32) and scale (f32), packed as six truncated halves
 * into d[4..6].  Scene.ts:150-176 with Matrix3.ts:33-47,63-80, all in f64 in the written order. */
static void orc_pack_cov(const float *rot, const float *scl, uint32_t *d)
{
    /* Scene.ts:150-157: Quaternion(x=r1, y=r2, z=r3, w=-r0); Matrix3.ts:67-80 */
    double qx = rot[1], qy = rot[2], qz = rot[3], qw = -(double)rot[0];
    double R[9] = {
        1 - 2 * qy * qy - 2 * qz * qz, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw,
        2 * qx * qy + 2 * qz * qw, 1 - 2 * qx * qx - 2 * qz * qz, 2 * qy * qz - 2 * qx * qw,
        2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx * qx - 2 * qy * qy,
    };
    /* Scene.ts:159-163: Diagonal(scale).multiply(rot); Matrix3.ts:33-47 with a = diag, b = R */
    double a[9] = { (double)scl[0], 0, 0, 0, (double)scl[1], 0, 0, 0, (double)scl[2] };
    const double *b = R;
    double M[9] = {
        b[0] * a[0] + b[3] * a[1] + b[6] * a[2], b[1] * a[0] + b[4] * a[1] + b[7] * a[2], b[2] * a[0] + b[5] * a[1] + b[8] * a[2],
        b[0] * a[3] + b[3] * a[4] + b[6] * a[5], b[1] * a[3] + b[4] * a[4] + b[7] * a[5], b[2] * a[3] + b[5] * a[4] + b[8] * a[5],
        b[0] * a[6] + b[3] * a[7] + b[6] * a[8], b[1] * a[6] + b[4] * a[7] + b[7] * a[8], b[2] * a[6] + b[5] * a[7] + b[8] * a[8],
    };
    /* Scene.ts:165-172 */
    double s0 = M[0] * M[0] + M[3] * M[3] + M[6] * M[6];
    double s1 = M[0] * M[1] + M[3] * M[4] + M[6] * M[7];
    double s2 = M[0] * M[2] + M[3] * M[5] + M[6] * M[8];
    double s3 = M[1] * M[1] + M[4] * M[4] + M[7] * M[7];
    double s4 = M[1] * M[2] + M[4] * M[5] + M[7] * M[8];
    double s5 = M[2] * M[2] + M[5] * M[5] + M[8] * M[8];
    /* Scene.ts:174-176 */
    d[4] = orc_pack_half2x16(4 * s0, 4 * s1);
    d[5] = orc_pack_half2x16(4 * s2, 4 * s3);
    d[6] = orc_pack_half2x16(4 * s4, 4 * s5);
}

/* Scene.setData: rows -> data (8 u32 per splat), positions, and the rotations (w,x,y,z) / scales the transforms use.
 * rotations / scales may be NULL. */
void orc_scene_build(const uint8_t *rows, uint32_t n, uint32_t *data, float *positions, float *rotations, float *scales)
{
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t *row = rows + (size_t)32 * i;
        float f[6], rot[4];
        memcpy(f, row, 24);
        uint32_t *d = data + (size_t)8 * i;
        /* Scene.ts:128-143 */
        positions[3 * i + 0] = f[0]; positions[3 * i + 1] = f[1]; positions[3 * i + 2] = f[2];
        memcpy(&d[0], &f[0], 4); memcpy(&d[1], &f[1], 4); memcpy(&d[2], &f[2], 4);
        d[3] = 0;
        /* Scene.ts:145-148 */
        d[7] = (uint32_t)row[24] | ((uint32_t)row[25] << 8) | ((uint32_t)row[26] << 16) | ((uint32_t)row[27] << 24);
        /* Scene.ts:132-135 (stored in a Float32Array: exact) */
        for (int k = 0; k < 4; k++) rot[k] = (float)(((double)row[28 + k] - 128) / 128);
        orc_pack_cov(rot, f + 3, d);
        if (rotations) memcpy(rotations + (size_t)4 * i, rot, 16);
        if (scales) memcpy(scales + (size_t)3 * i, f + 3, 12);
    }
}

void orc_scene_pack(const uint8_t *rows, uint32_t n, uint32_t *data, float *positions)
{
    orc_scene_build(rows, n, data, positions, NULL, NULL);
}

static void orc_write_pos(uint32_t i, uint32_t *data, const float *positions)
{
    memcpy(&data[(size_t)8 * i], &positions[(size_t)3 * i], 12);
}

/* Scene.translate, Scene.ts:182-195 (Float32Array += number: f64 add, rounded to f32 on store) */
void orc_scene_translate(uint32_t n, uint32_t *data, float *positions, const double *t)
{
    for (uint32_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) positions[3 * i + k] = (float)((double)positions[3 * i + k] + t[k]);
        orc_write_pos(i, data, positions);
    }
}

/* Scene.rotate, Scene.ts:197-257.  q = (x, y, z, w). */
void orc_scene_rotate(uint32_t n, uint32_t *data, float *positions, float *rotations, const float *scales, const double *q)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double R[9] = {   /* Matrix3.ts:67-80 */
        1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
        2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
        2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y,
    };
    for (uint32_t i = 0; i < n; i++) {
        const double px = positions[3 * i], py = positions[3 * i + 1], pz = positions[3 * i + 2];
        positions[3 * i + 0] = (float)(R[0] * px + R[1] * py + R[2] * pz);
        positions[3 * i + 1] = (float)(R[3] * px + R[4] * py + R[5] * pz);
        positions[3 * i + 2] = (float)(R[6] * px + R[7] * py + R[8] * pz);
        orc_write_pos(i, data, positions);
        float *r = rotations + (size_t)4 * i;
        /* currentRotation = Quaternion(x=r1, y=r2, z=r3, w=r0); newRot = rotation.multiply(current), Quaternion.ts:39-55 */
        const double w1 = w, x1 = x, y1 = y, z1 = z, w2 = r[0], x2 = r[1], y2 = r[2], z2 = r[3];
        const double nx = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
        const double ny = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
        const double nz = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
        const double nw = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
        r[1] = (float)nx; r[2] = (float)ny; r[3] = (float)nz; r[0] = (float)nw;
        orc_pack_cov(r, scales + (size_t)3 * i, data + (size_t)8 * i);
    }
}

/* Scene.scale, Scene.ts:259-305 */
void orc_scene_scale(uint32_t n, uint32_t *data, float *positions, const float *rotations, float *scales, const double *s)
{
    for (uint32_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            positions[3 * i + k] = (float)((double)positions[3 * i + k] * s[k]);
            scales[3 * i + k] = (float)((double)scales[3 * i + k] * s[k]);
        }
        orc_write_pos(i, data, positions);
        orc_pack_cov(rotations + (size_t)4 * i, scales + (size_t)3 * i, data + (size_t)8 * i);
    }
}

/* Scene.limitBox, Scene.ts:307-366: keeps splats with position inside the closed box, order preserved. Returns the new count. */
uint32_t orc_scene_limit_box(uint32_t n, uint32_t *data, float *positions, float *rotations, float *scales, const double *box)
{
    uint32_t kept = 0;
    for (uint32_t i = 0; i < n; i++) {
        const double x = positions[3 * i], y = positions[3 * i + 1], z = positions[3 * i + 2];
        if (!(x >= box[0] && x <= box[1] && y >= box[2] && y <= box[3] && z >= box[4] && z <= box[5])) continue;
        memmove(data + (size_t)8 * kept, data + (size_t)8 * i, 32);
        memmove(positions + (size_t)3 * kept, positions + (size_t)3 * i, 12);
        memmove(rotations + (size_t)4 * kept, rotations + (size_t)4 * i, 16);
        memmove(scales + (size_t)3 * kept, scales + (size_t)3 * i, 12);
        kept++;
    }
    return kept;
}

/* ------------------------------------------------------------------------
 * B1-B6: per-splat vertex stage, src/renderers/webgl/shaders/vertex.glsl.ts:130-231
 * (non-SH colour branch :207, scalingFactor = 1), in f32.  GLSL leaves the
 * evaluation order of matrix products open; this restatement fixes it as
 * left-to-right sums of products with no fused multiply-add, division and
 * square root correctly rounded, normalize(v) = v / sqrt(dot(v,v)).  The HIP
 * kernel uses the identical sequence, so every field below is compared
 * bit-for-bit.
 *
 * Outputs per splat
 *   raw[12]  the shader's own varyings, GL conventions (window y up):
 *            0,1 centre in window px  2,3 majorAxis  4,5 minorAxis (px, y up)
 *            6 opacity  7,8,9 rgb  10 pos2d.w  11 visible flag (1/0)
 *   rec[8]   the 32-byte record the tile blender consumes, IMAGE conventions
 *            (row 0 = top): cx, cy, ux, uy, wx, wy, log2(opacity), rgb8
 *            where u = 2*major/|major|^2, w = 2*minor/|minor|^2 (y flipped),
 *            so that vPosition at pixel p is (dot(p-c,u), dot(p-c,w)).
 *   bbox[4]  x0,y0,x1,y1 inclusive pixel bounds (clamped to the image) of the
 *            |vPosition| <= 2 ellipse's bounding box; x0 > x1 when invisible.
 * ---------------------------------------------------------------------- */

static float half_to_float(uint32_t h)
{
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1f, m = h & 0x3ff;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 0x400), (int)e - 25);
    return s ? -v : v;
}

/* r = a*b for 3x3 matrices held column-major (GLSL): m[c*3+r] */
static void mat3_mul(const float *a, const float *b, float *r)
{
    for (int c = 0; c < 3; c++)
        for (int ro = 0; ro < 3; ro++) {
            float s = a[0 * 3 + ro] * b[c * 3 + 0];
            s = s + a[1 * 3 + ro] * b[c * 3 + 1];
            s = s + a[2 * 3 + ro] * b[c * 3 + 2];
            r[c * 3 + ro] = s;
        }
}

static void mat3_transpose(const float *a, float *r)
{
    for (int c = 0; c < 3; c++)
        for (int ro = 0; ro < 3; ro++) r[c * 3 + ro] = a[ro * 3 + c];
}


/* ------------------------------------------------------------------------
 * SH colour: vertex.glsl.ts:9-104 (constants, fill_sh_from_packed, eval_sh_rgb) and :180-204.
 * Three RGBA32UI textures (one per colour channel), 8 u32 = 16 truncated halves per splat
 * (Scene.ts:108-124); coefficient k of channel c is half k of texture c.  f32, products and sums in the
 * written order (left to right), no fused multiply-add; normalize(v) = v / sqrt(dot(v,v)).
 * ---------------------------------------------------------------------- */
static const float ORC_SH_C0 = 0.28209479177387814f;
static const float ORC_SH_C1 = 0.4886025119029199f;
static const float ORC_SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
static const float ORC_SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                    -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

static void orc_eval_sh(const uint32_t *const sh[3], uint32_t t, uint32_t deg, const float dir[3], float rgb[3])
{
    float c[3][16];
    for (int ch = 0; ch < 3; ch++)
        for (int j = 0; j < 8; j++) {
            uint32_t w = sh[ch][(size_t)8 * t + j];
            c[ch][2 * j] = half_to_float(w & 0xffff);
            c[ch][2 * j + 1] = half_to_float(w >> 16);
        }
    const float x = dir[0], y = dir[1], z = dir[2];
    for (int ch = 0; ch < 3; ch++) {
        const float *k = c[ch];
        float r = ORC_SH_C0 * k[0];
        if (deg > 0) {
            r = r - ((((ORC_SH_C1 * y) * k[1]) + ((ORC_SH_C1 * z) * k[2])) - ((ORC_SH_C1 * x) * k[3]));
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                float s2 = (ORC_SH_C2[0] * xy) * k[4];
                s2 = s2 + (ORC_SH_C2[1] * yz) * k[5];
                s2 = s2 + (ORC_SH_C2[2] * ((2.0f * zz - xx) - yy)) * k[6];
                s2 = s2 + (ORC_SH_C2[3] * xz) * k[7];
                s2 = s2 + (ORC_SH_C2[4] * (xx - yy)) * k[8];
                r = r + s2;
                if (deg > 2) {
                    float s3 = ((ORC_SH_C3[0] * y) * (3.0f * xx - yy)) * k[9];
                    s3 = s3 + ((ORC_SH_C3[1] * xy) * z) * k[10];
                    s3 = s3 + ((ORC_SH_C3[2] * y) * ((4.0f * zz - xx) - yy)) * k[11];
                    s3 = s3 + ((ORC_SH_C3[3] * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy)) * k[12];
                    s3 = s3 + ((ORC_SH_C3[4] * x) * ((4.0f * zz - xx) - yy)) * k[13];
                    s3 = s3 + ((ORC_SH_C3[5] * z) * (xx - yy)) * k[14];
                    s3 = s3 + ((ORC_SH_C3[6] * x) * (xx - 3.0f * yy)) * k[15];
                    r = r + s3;
                }
            }
        }
        r = r + 0.5f;
        r = (r > 0.0f) ? r : 0.0f;   /* :103 max(result, 0) */
        r = (r < 1.0f) ? r : 1.0f;   /* :200 min(rgb, 1)    */
        rgb[ch] = r;
    }
}

/* eval_sh_rgb alone (vertex.glsl.ts:57-104 + the min(rgb, 1) of :200), for the fixture produced by running the reference's
 * own shader text (tests/golden/shader_golden.json, tests/test_shader_golden.py) */
void orc_eval_sh_rgb(const uint32_t *sh_r, const uint32_t *sh_g, const uint32_t *sh_b, uint32_t t, uint32_t deg,
                     const float *dir, float *rgb)
{
    const uint32_t *const sh[3] = { sh_r, sh_g, sh_b };
    orc_eval_sh(sh, t, deg, dir, rgb);
}

#define ORC_INVISIBLE(bb) do { (bb)[0] = 1; (bb)[1] = 1; (bb)[2] = 0; (bb)[3] = 0; } while (0)

/* sh_r/g/b (nullable together): the three SH textures; band[3] = Scene.bandsIndices (index of the last splat with
 * 0 / <=1 / <=2 SH bands).  With SH, splat i > band[0] takes its colour from eval_sh_rgb (vertex.glsl.ts:180-204) and its
 * record's rgb8 word is 0x01000000 ("colour is in raw[7..9]").
 * use_fade / fade: u_useDepthFade / u_depthFade (FadeInPass.ts:8-37): the quad's axes are scaled by
 * scalingFactor (vertex.glsl.ts:214-229); raw[2..5] hold the scaled axes; factor 0 = nothing drawn. */
void orc_project_full(const uint32_t *data, uint32_t n, const float *view, const float *proj,
                      float fx, float fy, int W, int H,
                      const uint32_t *sh_r, const uint32_t *sh_g, const uint32_t *sh_b, const int32_t *band,
                      int use_fade, float fade,
                      float *rec_out /*8n, nullable*/, int32_t *bbox_out /*4n, nullable*/, float *raw_out /*12n, nullable*/)
{
    const uint32_t *const sh[3] = { sh_r, sh_g, sh_b };
    /* inverse(view)[3].xyz (vertex.glsl.ts:197) for a rigid view matrix [A | b]: -A^T b */
    float campos[3];
    for (int r = 0; r < 3; r++) {
        float t = view[r * 4 + 0] * view[12];
        t = t + view[r * 4 + 1] * view[13];
        t = t + view[r * 4 + 2] * view[14];
        campos[r] = -t;
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *d = data + (size_t)8 * i;
        float raw[12];
        float rec[8];
        int32_t bb[4];
        uint32_t recu7 = 0;
        memset(raw, 0, sizeof raw);
        memset(rec, 0, sizeof rec);
        ORC_INVISIBLE(bb);
        do {
            /* :133-136 */
            float p[3];
            memcpy(p, d, 12);
            float cam[4], pos2d[4];
            for (int r = 0; r < 4; r++) {
                float s = view[0 * 4 + r] * p[0];
                s = s + view[1 * 4 + r] * p[1];
                s = s + view[2 * 4 + r] * p[2];
                s = s + view[3 * 4 + r];
                cam[r] = s;
            }
            for (int r = 0; r < 4; r++) {
                float s = proj[0 * 4 + r] * cam[0];
                s = s + proj[1 * 4 + r] * cam[1];
                s = s + proj[2 * 4 + r] * cam[2];
                s = s + proj[3 * 4 + r] * cam[3];
                pos2d[r] = s;
            }
            raw[10] = pos2d[3];
            /* :138-142 */
            float clip = 1.2f * pos2d[3];
            if (pos2d[2] < -pos2d[3] || pos2d[0] < -clip || pos2d[0] > clip || pos2d[1] < -clip || pos2d[1] > clip) break;
            /* :144-146 */
            float u1x = half_to_float(d[4] & 0xffff), u1y = half_to_float(d[4] >> 16);
            float u2x = half_to_float(d[5] & 0xffff), u2y = half_to_float(d[5] >> 16);
            float u3x = half_to_float(d[6] & 0xffff), u3y = half_to_float(d[6] >> 16);
            float Vrk[9] = { u1x, u1y, u2x, u1y, u2y, u3x, u2x, u3x, u3y };
            /* :148-152 */
            float zz = cam[2] * cam[2];
            float J[9] = {
                fx / cam[2], 0.f, -(fx * cam[0]) / zz,
                0.f, -fy / cam[2], (fy * cam[1]) / zz,
                0.f, 0.f, 0.f,
            };
            /* :154-155 */
            float V3[9] = { view[0], view[1], view[2], view[4], view[5], view[6], view[8], view[9], view[10] };
            float V3t[9], T[9], Tt[9], TtV[9], cov2d[9];
            mat3_transpose(V3, V3t);
            mat3_mul(V3t, J, T);
            mat3_transpose(T, Tt);
            mat3_mul(Tt, Vrk, TtV);
            mat3_mul(TtV, T, cov2d);
            /* :158-159 */
            float a = cov2d[0] + 0.3f;       /* [0][0] */
            float b = cov2d[1];              /* [0][1] */
            float c = cov2d[4] + 0.3f;       /* [1][1] */
            /* :161-163 */
            float det = a * c - b * b;
            if (det == 0.0f) break;
            /* :166-171 */
            float mid = (a + c) / 2.0f;
            float rad = mid * mid - det;
            float sq = sqrtf((0.1f < rad) ? rad : 0.1f); /* GLSL max(0.1, x) */
            float lambda1 = mid + sq;
            float lambda2 = mid - sq;
            if (lambda2 < 0.0f) break;
            /* :173-175 */
            float dvx = b, dvy = lambda1 - a;
            float len = sqrtf(dvx * dvx + dvy * dvy);
            float dgx = dvx / len, dgy = dvy / len;
            float smaj = sqrtf(2.0f * lambda1), smin = sqrtf(2.0f * lambda2);
            smaj = (1024.0f < smaj) ? 1024.0f : smaj; /* GLSL min(x, 1024) */
            smin = (1024.0f < smin) ? 1024.0f : smin;
            float majx = smaj * dgx, majy = smaj * dgy;
            float minx = smin * dgy, miny = smin * -dgx;
            if (use_fade) {   /* :216-223 */
                float depthNorm = (pos2d[2] / pos2d[3] + 1.0f) / 2.0f;
                float nearv = 0.1f, farv = 100.0f;
                float normalizedDepth = (2.0f * nearv) / (farv + nearv - depthNorm * (farv - nearv));
                float st = normalizedDepth - 0.1f; st = (st > 0.0f) ? st : 0.0f;
                float en = normalizedDepth + 0.1f; en = (en < 1.0f) ? en : 1.0f;
                float sf = (fade - st) / (en - st);
                sf = (sf < 0.0f) ? 0.0f : ((sf > 1.0f) ? 1.0f : sf);   /* clamp(x, 0, 1) = min(max(x, 0), 1) */
                if (!(sf > 0.0f)) break;   /* zero-area quad: no fragments (NaN factor: position undefined, dropped) */
                majx = majx * sf; majy = majy * sf; minx = minx * sf; miny = miny * sf;   /* :226-229 */
            }
            if (!(isfinite(majx) && isfinite(majy) && isfinite(minx) && isfinite(miny))) break; /* normalize(0,0): dropped */
            /* :177-178, :207 */
            uint32_t cw = d[7];
            float opacity = (float)((cw >> 24) & 0xff) / 255.0f;
            float cr = (float)(cw & 0xff) / 255.0f, cg = (float)((cw >> 8) & 0xff) / 255.0f, cb = (float)((cw >> 16) & 0xff) / 255.0f;
            int use_sh = sh_r && (int32_t)i > band[0];   /* :180 */
            if (use_sh) {
                uint32_t deg = (int32_t)i > band[1] ? ((int32_t)i > band[2] ? 3u : 2u) : 1u;   /* :189 */
                float dv[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
                float dl = sqrtf((dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2]);
                float dir[3] = { dv[0] / dl, dv[1] / dl, dv[2] / dl };
                float rgb[3];
                orc_eval_sh(sh, i - (uint32_t)(band[0] + 1), deg, dir, rgb);
                cr = rgb[0]; cg = rgb[1]; cb = rgb[2];
            }
            /* :226-229 + the GL viewport transform */
            float vcx = pos2d[0] / pos2d[3], vcy = pos2d[1] / pos2d[3];
            float xw = ((vcx + 1.0f) * 0.5f) * (float)W;
            float yw = ((vcy + 1.0f) * 0.5f) * (float)H;
            raw[0] = xw; raw[1] = yw; raw[2] = majx; raw[3] = majy; raw[4] = minx; raw[5] = miny;
            raw[6] = opacity; raw[7] = cr; raw[8] = cg; raw[9] = cb; raw[11] = 1.0f;

            /* ---- derived record (this build's own, image conventions) ---- */
            float cx = xw;
            float cy = ((1.0f - vcy) * 0.5f) * (float)H;
            float m2 = majx * majx + majy * majy;
            float n2 = minx * minx + miny * miny;
            float im = 2.0f / m2, in = 2.0f / n2;
            float ux = majx * im, uy = -majy * im;
            float wx = minx * in, wy = -miny * in;
            if (!(isfinite(ux) && isfinite(uy) && isfinite(wx) && isfinite(wy) && isfinite(cx) && isfinite(cy))) { raw[11] = 0.f; break; }
            float ex = sqrtf(majx * majx + minx * minx);
            float ey = sqrtf(majy * majy + miny * miny);
            float fx0 = floorf(cx - ex - 0.5f), fx1 = ceilf(cx + ex - 0.5f);
            float fy0 = floorf(cy - ey - 0.5f), fy1 = ceilf(cy + ey - 0.5f);
            fx0 = fmaxf(fx0, 0.0f); fy0 = fmaxf(fy0, 0.0f);
            fx1 = fminf(fx1, (float)(W - 1)); fy1 = fminf(fy1, (float)(H - 1));
            rec[0] = cx; rec[1] = cy; rec[2] = ux; rec[3] = uy; rec[4] = wx; rec[5] = wy;
            rec[6] = log2f(opacity); /* compared with a 2-ulp tolerance, not bitwise: v_log_f32 */
            recu7 = use_sh ? 0x01000000u : (cw & 0x00ffffffu);
            if (fx0 > fx1 || fy0 > fy1) break; /* off-screen: keeps raw visible flag, empty bbox */
            bb[0] = (int32_t)fx0; bb[1] = (int32_t)fy0; bb[2] = (int32_t)fx1; bb[3] = (int32_t)fy1;
        } while (0);
        if (raw_out) memcpy(raw_out + (size_t)12 * i, raw, sizeof raw);
        if (rec_out) {
            memcpy(rec_out + (size_t)8 * i, rec, sizeof rec);
            memcpy(rec_out + (size_t)8 * i + 7, &recu7, 4);
        }
        if (bbox_out) memcpy(bbox_out + (size_t)4 * i, bb, sizeof bb);
    }
}

void orc_project_sh(const uint32_t *data, uint32_t n, const float *view, const float *proj, float fx, float fy, int W, int H,
                    const uint32_t *sh_r, const uint32_t *sh_g, const uint32_t *sh_b, const int32_t *band,
                    float *rec_out, int32_t *bbox_out, float *raw_out)
{
    orc_project_full(data, n, view, proj, fx, fy, W, H, sh_r, sh_g, sh_b, band, 0, 1.0f, rec_out, bbox_out, raw_out);
}

void orc_project(const uint32_t *data, uint32_t n, const float *view, const float *proj,
                 float fx, float fy, int W, int H, float *rec_out, int32_t *bbox_out, float *raw_out)
{
    orc_project_full(data, n, view, proj, fx, fy, W, H, NULL, NULL, NULL, NULL, 0, 1.0f, rec_out, bbox_out, raw_out);
}

/* Counts used by the bench's byte model: V = splats with a
 * non-empty bbox, D = sum over them of TILE x TILE tiles their bbox overlaps. */
void orc_tile_stats(const int32_t *bbox, uint32_t n, int tile, uint64_t *V, uint64_t *D)
{
    uint64_t v = 0, d = 0;
    for (uint32_t i = 0; i < n; i++) {
        const int32_t *b = bbox + (size_t)4 * i;
        if (b[0] > b[2] || b[1] > b[3]) continue;
        v++;
        d += (uint64_t)(b[2] / tile - b[0] / tile + 1) * (uint64_t)(b[3] / tile - b[1] / tile + 1);
    }
    *V = v; *D = d;
}

/* ------------------------------------------------------------------------
 * C1-C2: fragment stage + blend, frag.glsl.ts:13-21 with the blend state of
 * WebGLRenderer.ts:139-142,279-285: clear (0,0,0,0); per fragment, in
 * depth_index order:  dst.rgb += (1-dst.a)*B*rgb ; dst.a += (1-dst.a)*B.
 * Accumulation is f64 (the browser's RGBA8 ROP rounding is NOT reproduced).
 *
 * mode 0 "ideal":    vPosition solved in f64 from the f32 varyings (raw),
 *                    window coordinates, coverage |vPosition|^2 <= 4.
 * mode 1 "restated": vPosition from the f32 record with the exact f32
 *                    expression the HIP kernel uses (explicit fmaf = v_fma_f32,
 *                    everything else unfused), so the discard decision
 *                    (A < -4) is bit-identical; weight and sums in f64.
 * mode 2 "RGBA8 ROP": geometry of mode 0, but the destination is re-quantised to
 *                    8-bit UNORM after EVERY fragment, like the reference's default
 *                    drawing buffer (RGBA8, premultipliedAlpha; WebGLRenderer.ts:38,
 *                    139-142,282-285): dst = round(clamp(dst + (1-dst.a)*src, 0, 1)*255)/255.
 *                    A model of a fixed-function ROP (real hardware differs in the last
 *                    bit of the blend arithmetic); used only to QUANTIFY the gap between
 *                    the browser's canvas and the fp32 image this build defines parity on.
 * out: W*H*4 floats, premultiplied RGBA, row 0 = top.
 * y_begin/y_end restrict rows (lets callers thread over row bands).
 * ---------------------------------------------------------------------- */
/* frag.glsl.ts:13-20 for one fragment: A = -dot(vPosition, vPosition); discard below -4; B = clamp(exp(A) * opacity, 0, 1).
 * Returns 0 when the fragment is discarded.  (f64: the modes of orc_render that follow the shader's own geometry use it.) */
static inline int orc_frag_weight(double vx, double vy, double opacity, double *B)
{
    double A = -(vx * vx + vy * vy);
    if (A < -4.0) return 0;
    double b = exp(A) * opacity;
    if (b > 1.0) b = 1.0;
    if (b < 0.0) b = 0.0;
    *B = b;
    return 1;
}

/* WebGLRenderer.ts:139-142,282-285: blendFuncSeparate(ONE_MINUS_DST_ALPHA, ONE, ONE_MINUS_DST_ALPHA, ONE), FUNC_ADD on the
 * fragment (B rgb, B): dst.rgb += (1 - dst.a) * B rgb; dst.a += (1 - dst.a) * B  -- front-to-back "under" */
static inline void orc_blend_under(double *px, double B, double cr, double cg, double cb)
{
    double t = 1.0 - px[3];
    px[0] += t * B * cr; px[1] += t * B * cg; px[2] += t * B * cb; px[3] += t * B;
}

/* k fragments (vPosition xy, colour rgba: 6 floats each) onto one cleared pixel, in order: the fixture of the reference's
 * blend state applied to its executed fragment shader's outputs (tests/test_shader_golden.py) */
void orc_composite(const float *frags, uint32_t k, float *out)
{
    double px[4] = { 0.0, 0.0, 0.0, 0.0 };
    for (uint32_t j = 0; j < k; j++) {
        const float *f = frags + (size_t)6 * j;
        double B;
        if (!orc_frag_weight((double)f[0], (double)f[1], (double)f[5], &B)) continue;
        orc_blend_under(px, B, f[2], f[3], f[4]);
    }
    for (int ch = 0; ch < 4; ch++) out[ch] = (float)px[ch];
}

/* the same for the fixture of the reference's executed fragment shader: out = (B * rgb, B); returns 0 on discard */
int orc_fragment(const float *vpos, const float *color, float *out)
{
    double B;
    if (!orc_frag_weight((double)vpos[0], (double)vpos[1], (double)color[3], &B)) return 0;
    out[0] = (float)(B * color[0]); out[1] = (float)(B * color[1]); out[2] = (float)(B * color[2]); out[3] = (float)B;
    return 1;
}

void orc_render(uint32_t n, const uint32_t *depth_index, const float *raw, const float *rec,
                const int32_t *bbox, int W, int H, int mode, int y_begin, int y_end, float *out)
{
    size_t np = (size_t)W * (size_t)(y_end - y_begin);
    double *acc = (double *)calloc(np * 4, sizeof(double));
    for (uint32_t k = 0; k < n; k++) {
        uint32_t i = depth_index[k];
        const int32_t *b = bbox + (size_t)4 * i;
        if (b[0] > b[2] || b[1] > b[3]) continue;
        int y0 = b[1] < y_begin ? y_begin : b[1];
        int y1 = b[3] >= y_end ? y_end - 1 : b[3];
        const float *rw = raw + (size_t)12 * i;
        const float *rc = rec + (size_t)8 * i;
        uint32_t rgb8;
        memcpy(&rgb8, rc + 7, 4);
        double opacity = rw[6];
        double cr = rw[7], cg = rw[8], cb = rw[9];
        double M2 = (double)rw[2] * rw[2] + (double)rw[3] * rw[3];
        double N2 = (double)rw[4] * rw[4] + (double)rw[5] * rw[5];
        for (int y = y0; y <= y1; y++) {
            for (int x = b[0]; x <= b[2]; x++) {
                double B;
                if (mode == 0 || mode == 2) {
                    double dx = (x + 0.5) - (double)rw[0];
                    double dy = ((double)H - (y + 0.5)) - (double)rw[1];
                    double vx = 2.0 * (dx * rw[2] + dy * rw[3]) / M2;
                    double vy = 2.0 * (dx * rw[4] + dy * rw[5]) / N2;
                    if (!orc_frag_weight(vx, vy, opacity, &B)) continue;
                } else {
                    /* k_blend's expression: coordinates relative to the centre of the first pixel of
                     * the 32x32 bin that holds (x, y); the centre is folded into ncu / ncw */
                    int bx0 = (x / 32) * 32, by0 = (y / 32) * 32;
                    float cxr = rc[0] - ((float)bx0 + 0.5f), cyr = rc[1] - ((float)by0 + 0.5f);
                    float ncu = -fmaf(rc[3], cyr, rc[2] * cxr);
                    float ncw = -fmaf(rc[5], cyr, rc[4] * cxr);
                    float pxl = (float)(x - bx0), pyl = (float)(y - by0);
                    float vx = fmaf(rc[2], pxl, fmaf(rc[3], pyl, ncu));
                    float vy = fmaf(rc[4], pxl, fmaf(rc[5], pyl, ncw));
                    float q = fmaf(vy, vy, vx * vx);
                    if (q > 4.0f) continue;
                    B = exp(-(double)q) * opacity;
                }
                if (B > 1.0) B = 1.0;
                if (B < 0.0) B = 0.0;
                double *px = acc + ((size_t)(y - y_begin) * W + x) * 4;
                orc_blend_under(px, B, cr, cg, cb);
                if (mode == 2)
                    for (int ch = 0; ch < 4; ch++) {
                        double v = px[ch] < 0.0 ? 0.0 : (px[ch] > 1.0 ? 1.0 : px[ch]);
                        px[ch] = floor(v * 255.0 + 0.5) / 255.0;
                    }
            }
        }
        (void)rgb8;
    }
    float *o = out + (size_t)y_begin * W * 4;
    for (size_t j = 0; j < np * 4; j++) o[j] = (float)acc[j];
    free(acc);
}
