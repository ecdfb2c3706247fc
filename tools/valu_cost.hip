// Per-instruction issue cost on one SIMD of gfx950, for the instruction forms the compositor can be written with.
// Each kernel issues one instruction form in 8 independent copies per unrolled step, 8 waves per SIMD; cost =
// shader cycles one SIMD spends per wave-instruction, using the in-kernel clock (s_memtime / s_memrealtime), not the
// nominal one.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_cost.hip -o tools/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define REP8(S) S S S S S S S S

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamp, int iters, float a, float b, const float* lds_src)
{
    __shared__ v4f s_buf[256];
    s_buf[threadIdx.x] = (v4f){a, b, a, b};
    __syncthreads();
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v2f pa = {a, a}, pb = {b, b};
    float va = a, vb = b, vc = a + b;
    float sa = a;   // scalar operand
    asm volatile("" : "+v"(va), "+v"(vb), "+v"(vc), "+v"(pa), "+v"(pb), "+s"(sa));
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    unsigned lds_addr = (threadIdx.x & 192) * 16;   // wave-uniform LDS byte address (broadcast read)
    v4f l0, l1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va));
            else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
            else if (KIND == 2) asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
            else if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(sa), "v"(vb));
            else if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            else if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            else if (KIND == 6) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            else if (KIND == 7) asm volatile("v_cmp_ge_f32 s[20:21], 4.0, %0\n v_cmp_ge_f32 s[20:21], 4.0, %1\n v_cmp_ge_f32 s[20:21], 4.0, %2\n v_cmp_ge_f32 s[20:21], 4.0, %3\n v_cmp_ge_f32 s[20:21], 4.0, %4\n v_cmp_ge_f32 s[20:21], 4.0, %5\n v_cmp_ge_f32 s[20:21], 4.0, %6\n v_cmp_ge_f32 s[20:21], 4.0, %7"
                                             : : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7) : "s20", "s21");
            else if (KIND == 8) asm volatile("v_fmamk_f32 %0, %0, 0xbfb8aa3b, %8\n v_fmamk_f32 %1, %1, 0xbfb8aa3b, %8\n v_fmamk_f32 %2, %2, 0xbfb8aa3b, %8\n v_fmamk_f32 %3, %3, 0xbfb8aa3b, %8\n v_fmamk_f32 %4, %4, 0xbfb8aa3b, %8\n v_fmamk_f32 %5, %5, 0xbfb8aa3b, %8\n v_fmamk_f32 %6, %6, 0xbfb8aa3b, %8\n v_fmamk_f32 %7, %7, 0xbfb8aa3b, %8"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va));
            else if (KIND == 9) asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8"
                                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va));
            else if (KIND == 10) {   // LDS broadcast reads: 2 x b128 + 1 x b64 per "entry", 4 entries, drained once
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n ds_read_b64 %3, %2 offset:8192\n"
                             "ds_read_b128 %0, %2 offset:16\n ds_read_b128 %1, %2 offset:4112\n ds_read_b64 %3, %2 offset:8208\n"
                             "ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:4128\n ds_read_b64 %3, %2 offset:8224\n"
                             "ds_read_b128 %0, %2 offset:48\n ds_read_b128 %1, %2 offset:4144\n ds_read_b64 %3, %2 offset:8240\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(l0), "=&v"(l1) : "v"(lds_addr & 0xfff), "v"(p0));
            } else if (KIND == 11) {   // the pk variant's 9-VALU quadrant
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n s_nop 0\n v_mul_f32 %1, %8, %8\n v_fmac_f32 %1, %9, %9\n v_cmp_ge_f32 vcc, 4.0, %1\n v_fmamk_f32 %1, %1, 0xbfb8aa3b, %6\n v_exp_f32 %1, %1\n s_nop 0\n"
                             "v_mul_f32 %1, %7, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                             : "+v"(p0), "+v"(x4), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb), "v"(va), "v"(vb), "v"(x5), "v"(x6) : "vcc");
            } else if (KIND == 13) {   // shipped geometry (2 fma), colour + T as two pk_fma (no sub, no scalar fma): 10 VALU
                asm volatile("v_fma_f32 %0, %6, %7, %0\n v_fma_f32 %1, %6, %7, %1\n v_mul_f32 %2, %0, %0\n v_fmac_f32 %2, %1, %1\n"
                             "v_cmp_ge_f32 vcc, 4.0, %2\n v_fmamk_f32 %2, %2, 0xbfb8aa3b, %7\n v_exp_f32 %2, %2\n s_nop 0\n v_mul_f32 %3, %4, %2\n"
                             "v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %10, %8, %9, %10"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(p0) : "v"(va), "v"(vb), "v"(pa), "v"(pb), "v"(p1) : "vcc");
            } else if (KIND == 14) {   // all scalar: 2 fma, mul, fmac, cmp, fmamk, exp, mul, 3 fmac, sub: 12 VALU
                asm volatile("v_fma_f32 %0, %6, %7, %0\n v_fma_f32 %1, %6, %7, %1\n v_mul_f32 %2, %0, %0\n v_fmac_f32 %2, %1, %1\n"
                             "v_cmp_ge_f32 vcc, 4.0, %2\n v_fmamk_f32 %2, %2, 0xbfb8aa3b, %7\n v_exp_f32 %2, %2\n s_nop 0\n v_mul_f32 %3, %4, %2\n"
                             "v_sub_f32 %4, %4, %3\n v_fmac_f32 %5, %3, %6\n v_fmac_f32 %8, %3, %6\n v_fmac_f32 %9, %3, %7"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(va), "+v"(vb), "+v"(x6), "+v"(x7) : : "vcc");
            } else if (KIND == 15) {   // coverage test only (what an uncovered quadrant costs): 2 fma, mul, fmac, cmp
                asm volatile("v_fma_f32 %0, %3, %4, %0\n v_fma_f32 %1, %3, %4, %1\n v_mul_f32 %2, %0, %0\n v_fmac_f32 %2, %1, %1\n v_cmp_ge_f32 vcc, 4.0, %2"
                             : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(va), "v"(vb) : "vcc");
            } else if (KIND == 16) {   // coverage test with pk geometry: pk_fma, nop, mul, fmac, cmp
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n s_nop 0\n v_mul_f32 %1, %4, %4\n v_fmac_f32 %1, %5, %5\n v_cmp_ge_f32 vcc, 4.0, %1"
                             : "+v"(p0), "+v"(x2) : "v"(pa), "v"(pb), "v"(x0), "v"(x1) : "vcc");
            } else if (KIND == 17) {   // s_nop 0 alone
                asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
            } else if (KIND == 18) {   // v_pk_fma with a broadcast source (op_sel_hi:[0,1,1]) like the colour update
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[0,1,1]\n"
                             "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[0,1,1]"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (KIND == 19) {   // two independent shipped quadrant chains interleaved (22 VALU): is one chain latency-bound?
                asm volatile("v_fma_f32 %0, %12, %13, %0\n v_fma_f32 %6, %12, %13, %6\n v_fma_f32 %1, %12, %13, %1\n v_fma_f32 %7, %12, %13, %7\n"
                             "v_mul_f32 %2, %0, %0\n v_mul_f32 %8, %6, %6\n v_fmac_f32 %2, %1, %1\n v_fmac_f32 %8, %7, %7\n"
                             "v_cmp_ge_f32 vcc, 4.0, %2\n v_cmp_ge_f32 s[20:21], 4.0, %8\n v_fmamk_f32 %2, %2, 0xbfb8aa3b, %13\n v_fmamk_f32 %8, %8, 0xbfb8aa3b, %13\n"
                             "v_exp_f32 %2, %2\n v_exp_f32 %8, %8\n s_nop 0\n v_mul_f32 %3, %4, %2\n v_mul_f32 %9, %10, %8\n"
                             "v_sub_f32 %4, %4, %3\n v_sub_f32 %10, %10, %9\n v_pk_fma_f32 %5, %5, %14, %15\n v_pk_fma_f32 %11, %11, %14, %15\n v_fmac_f32 %0, %3, %12\n v_fmac_f32 %6, %9, %12"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(p0), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(vc), "+v"(p1.x), "+v"(p2)
                             : "v"(va), "v"(vb), "v"(pa), "v"(pb) : "vcc", "s20", "s21");
            } else if (KIND == 20) {   // short chain: scaled axes, arg = la - vx'^2 - vy'^2 by two chained fma, cover test on arg: 9 VALU, depth 6
                asm volatile("v_fma_f32 %0, %6, %7, %0\n v_fma_f32 %1, %6, %7, %1\n v_fma_f32 %2, -%0, %0, %7\n v_fma_f32 %2, -%1, %1, %2\n"
                             "v_cmp_ge_f32 vcc, %2, %6\n v_exp_f32 %2, %2\n s_nop 0\n v_mul_f32 %3, %4, %2\n"
                             "v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %10, %8, %9, %10"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(p0) : "v"(va), "v"(vb), "v"(pa), "v"(pb), "v"(p1) : "vcc");
            } else if (KIND == 21) {   // two short chains interleaved (18 VALU)
                asm volatile("v_fma_f32 %0, %12, %13, %0\n v_fma_f32 %6, %12, %13, %6\n v_fma_f32 %1, %12, %13, %1\n v_fma_f32 %7, %12, %13, %7\n"
                             "v_fma_f32 %2, -%0, %0, %13\n v_fma_f32 %8, -%6, %6, %13\n v_fma_f32 %2, -%1, %1, %2\n v_fma_f32 %8, -%7, %7, %8\n"
                             "v_cmp_ge_f32 vcc, %2, %12\n v_cmp_ge_f32 s[20:21], %8, %12\n v_exp_f32 %2, %2\n v_exp_f32 %8, %8\n s_nop 0\n"
                             "v_mul_f32 %3, %4, %2\n v_mul_f32 %9, %10, %8\n v_pk_fma_f32 %5, %14, %15, %5\n v_pk_fma_f32 %11, %14, %15, %11\n"
                             "v_pk_fma_f32 %16, %14, %15, %16\n v_pk_fma_f32 %17, %14, %15, %17"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(p0), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(vc), "+v"(p1.x), "+v"(p2)
                             : "v"(va), "v"(vb), "v"(pa), "v"(pb), "v"(p3), "v"(p1) : "vcc", "s20", "s21");
            } else {   // KIND 12: the shipped 11-VALU quadrant
                asm volatile("v_fma_f32 %0, %6, %7, %0\n v_fma_f32 %1, %6, %7, %1\n v_mul_f32 %2, %0, %0\n v_fmac_f32 %2, %1, %1\n"
                             "v_cmp_ge_f32 vcc, 4.0, %2\n v_fmamk_f32 %2, %2, 0xbfb8aa3b, %7\n v_exp_f32 %2, %2\n s_nop 0\n v_mul_f32 %3, %4, %2\n"
                             "v_sub_f32 %4, %4, %3\n v_pk_fma_f32 %5, %5, %8, %9\n v_fmac_f32 %0, %3, %6"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(p0) : "v"(va), "v"(vb), "v"(pa), "v"(pb) : "vcc");
            }
        }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (blockIdx.x * 256 + threadIdx.x) >> 6;
        stamp[2 * w] = t1 - t0;
        stamp[2 * w + 1] = r1 - r0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + l0.x + l1.y;
}

template <int KIND>
void run(const char* name, int instr_per_u, int blocks_per_cu, float* d, unsigned long long* dc)
{
    const int blocks = 256 * blocks_per_cu, iters = 2048;   // a block = 4 waves = one wave per SIMD of its CU
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, 16, 1.0001f, 0.5f, (const float*)d);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0001f, 0.5f, (const float*)d);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float wall_ms = 0;
    (void)hipEventElapsedTime(&wall_ms, e0, e1);
    std::vector<unsigned long long> c(blocks * 4 * 2);
    (void)hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (size_t w = 0; w < c.size() / 2; w++) { cyc += (double)c[2 * w]; real += (double)c[2 * w + 1]; }
    const double n = c.size() / 2.0;
    cyc /= n; real /= n;
    const double per_wave = (double)iters * 4 * instr_per_u;
    // a SIMD hosts blocks_per_cu waves: cycles it spends per wave-instruction = wave lifetime / (instr per wave * waves per SIMD)
    printf("%-44s %d waves/SIMD  %6.2f SIMD cycles per wave-instr   (clock %.2f GHz)   wall %.3f ms -> %.3f T wave-instr/s, wave lifetime %.3f ms\n", name, blocks_per_cu,
           cyc / (per_wave * blocks_per_cu), cyc / real * 0.1, wall_ms, (double)blocks * 4 * per_wave / (wall_ms * 1e-3) / 1e12, real * 1e-5);
}

int main()
{
    float* d;
    unsigned long long* dc;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    (void)hipMalloc(&dc, 256 * 8 * 4 * 8 * 2);
    for (int w : {7}) {
        run<0>("v_mul_f32 (2 VGPR)", 8, w, d, dc);
        run<9>("v_sub_f32 (2 VGPR)", 8, w, d, dc);
        run<1>("v_fma_f32 (3 VGPR)", 8, w, d, dc);
        run<2>("v_fmac_f32 (VOP2, 2 VGPR + acc)", 8, w, d, dc);
        run<3>("v_fma_f32 (SGPR, VGPR, VGPR)", 8, w, d, dc);
        run<8>("v_fmamk_f32 (literal)", 8, w, d, dc);
        run<4>("v_pk_fma_f32 (3 pairs)", 8, w, d, dc);
        run<5>("v_pk_mul_f32 (2 pairs)", 8, w, d, dc);
        run<6>("v_exp_f32", 8, w, d, dc);
        run<7>("v_cmp_ge_f32 -> sgpr pair", 8, w, d, dc);
        run<10>("LDS broadcast entry (2 b128 + b64), per entry", 4, w, d, dc);
        run<12>("shipped quadrant (11 VALU), per instr", 11, w, d, dc);
        run<11>("pk quadrant (9 VALU), per instr", 9, w, d, dc);
        run<19>("2 shipped quadrants interleaved, per quadrant pair /22", 22, w, d, dc);
        run<20>("short-chain quadrant (9 VALU), per instr", 9, w, d, dc);
        run<21>("2 short-chain quadrants interleaved /18", 18, w, d, dc);
        run<13>("quadrant B: 2 fma + 2 pk colour (10 VALU), per instr", 10, w, d, dc);
        run<14>("quadrant all scalar (13 VALU), per instr", 13, w, d, dc);
        run<15>("coverage test scalar (5 VALU), per instr", 5, w, d, dc);
        run<16>("coverage test pk (4 VALU + nop), per instr", 4, w, d, dc);
        run<17>("s_nop 0", 8, w, d, dc);
        run<18>("v_pk_fma_f32 op_sel_hi:[0,1,1]", 8, w, d, dc);
    }
    return 0;
}
