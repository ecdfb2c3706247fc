// Calibration microbenchmark: sustained VALU issue rate on this GPU for the instruction kinds the compositor
// uses (v_fma_f32, v_pk_fma_f32, v_exp_f32, v_mul_f32), 8 waves per SIMD, independent accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    float2_t p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const float2_t pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) {   // 8 x v_fma_f32
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            } else if (KIND == 1) {   // 8 x v_pk_fma_f32 (16 FMAs), 8 independent chains like the scalar case
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
                p4 = __builtin_elementwise_fma(p4, pa, pb); p5 = __builtin_elementwise_fma(p5, pa, pb);
                p6 = __builtin_elementwise_fma(p6, pa, pb); p7 = __builtin_elementwise_fma(p7, pa, pb);
            } else if (KIND == 2) {   // 8 x v_exp_f32
                x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
                x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
            } else {   // compositor-like mix per pixel: 4 fma, 1 mul, 1 sub, 1 exp, 1 cndmask
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, x0); x2 = x1 * x1; x3 = __builtin_fmaf(x0, x0, x2);
                x4 = __builtin_amdgcn_exp2f(x3); x5 = (x3 <= 4.0f) ? x4 : 0.0f; x6 = x6 - x5 * x6; x7 = __builtin_fmaf(x5, a, x7);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int KIND>
void run(const char* name, int instr_per_u, float* d)
{
    const int blocks = 256 * 8, iters = 4096;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * 8 * instr_per_u;
    printf("%-28s %8.3f ms  %7.3f T wave-instr/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms,
           wave_instr / (ms * 1e-3) / 1e12, 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)));
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 8, d);
    run<1>("v_pk_fma_f32 (2 FMA each)", 8, d);
    run<2>("v_exp_f32", 8, d);
    run<3>("compositor mix (9 instr)", 9, d);
    return 0;
}
