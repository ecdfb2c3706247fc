// Calibration microbenchmark: sustained VALU issue rate of one SIMD for the instruction kinds the compositor
// uses, with 1..8 waves per SIMD and independent accumulators.  The instructions are written as inline asm:
// plain C++ FMAs on adjacent scalars are SLP-packed into v_pk_fma_f32 by -O3, which made an earlier version
// of this tool report a "v_fma_f32" rate that was really the packed one.
// Rates are wall-clock (HIP events) over the whole chip; cycles are quoted at the nominal 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    float2_t p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    float2_t pa = {a, a}, pb = {b, b};
    float va = a, vb = b;
    asm volatile("" : "+v"(va), "+v"(vb), "+v"(pa), "+v"(pb));   // operands in VGPRs, like the compositor's
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (KIND == 0) {          // 8 x v_fma_f32, independent
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
            } else if (KIND == 1) {   // 4 x v_pk_fma_f32 (8 FMAs)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (KIND == 2) {   // 8 x v_exp_f32
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 3) {   // 8 x v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va));
            } else {                  // one covered quadrant of the compositor: 11 VALU incl. 1 exp and 1 pk_fma
                asm volatile("v_fma_f32 %0, %6, %7, %0\n v_fma_f32 %1, %6, %7, %1\n v_mul_f32 %2, %0, %0\n v_fmac_f32 %2, %1, %1\n"
                             "v_cmp_ge_f32 vcc, 4.0, %2\n v_fma_f32 %2, %2, %6, %7\n v_exp_f32 %2, %2\n s_nop 0\n v_mul_f32 %3, %4, %2\n"
                             "v_sub_f32 %4, %4, %3\n v_pk_fma_f32 %5, %5, %8, %9\n v_fmac_f32 %0, %3, %6"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(p0) : "v"(va), "v"(vb), "v"(pa), "v"(pb) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
void run(const char* name, int instr_per_u, int blocks_per_cu, float* d, unsigned long long* dc)
{
    const int blocks = 256 * blocks_per_cu, iters = 4096;   // a block = 4 waves = one wave per SIMD of its CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, 16, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks * 4);
    (void)hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= c.size();
    const double per_wave = (double)iters * 4 * instr_per_u;            // instructions one wave issued
    const double wave_instr = (double)blocks * 4 * per_wave;
    (void)mean;  // s_memtime deltas are kept for inspection; the wall-clock rate is the robust figure
    const double rate = wave_instr / (ms * 1e-3);
    printf("%-34s %d waves/SIMD  %6.3f T wave-instr/s  = %5.2f SIMD cycles per wave-instr at 2.4 GHz\n", name, blocks_per_cu,
           rate / 1e12, 1024.0 * 2.4e9 / rate);
}

int main()
{
    float* d;
    unsigned long long* dc;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    (void)hipMalloc(&dc, 256 * 8 * 4 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 8, w, d, dc);
        run<1>("v_pk_fma_f32 (2 FMA each)", 4, w, d, dc);
        run<2>("v_exp_f32", 8, w, d, dc);
        run<3>("v_mul_f32", 8, w, d, dc);
        run<4>("compositor quadrant (11 VALU)", 11, w, d, dc);
    }
    return 0;
}
