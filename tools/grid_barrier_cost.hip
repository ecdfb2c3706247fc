// What does a grid-wide barrier cost on MI355X, beside the kernel boundary it would replace?  (the cooperative
// single-launch sort of VERDICT r3 item 5 trades four launch boundaries of the sort chain for four grid barriers.)
//   flat:         every workgroup adds 1 to ONE counter (agent scope) and its first wave spins until the counter reaches
//                 round x workgroups;
//   hierarchical: a counter per XCD (workgroup % 8, the dispatcher's round-robin), the last arrival of an XCD adds to the
//                 global one: 8 + workgroups / 8 same-address atomics instead of `workgroups`;
//   boundary:     the same grid as a chain of (nearly) empty kernels replayed as one HIP graph: one launch boundary per node.
// Every spin is bounded (SPIN_LIMIT polls, then an error flag and on): a lost arrival cannot hang the box.  The grids are
// the sort chain's: 489 workgroups of 1024 threads (C3, two per CU: all resident) and 256 (one per CU).
// Build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_cost.hip -o tools/grid_barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ bool spin_until(const unsigned* word, unsigned target, unsigned* err)
{
    unsigned polls = 0;
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++polls >= SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicOr(err, 1u); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

template <bool HIER>
__global__ __launch_bounds__(1024) void k_rounds(unsigned* counters /* [0] global, [16 * (1 + x)] per XCD */, unsigned* err, int rounds, unsigned* sink)
{
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u;
    const unsigned per_xcd = (nwg >> 3) + ((nwg & 7u) > xcd ? 1u : 0u);
    unsigned acc = threadIdx.x;
    __shared__ int s_abort;
    if (threadIdx.x == 0) s_abort = 0;
    for (int r = 0; r < rounds; r++) {
        acc = acc * 1664525u + 1013904223u;   // (something between the barriers)
        __syncthreads();
        if (threadIdx.x == 0) {
            if (HIER) {
                const unsigned before = __hip_atomic_fetch_add(&counters[16 * (1 + xcd)], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1u == (unsigned)(r + 1) * per_xcd) __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (!spin_until(&counters[0], (unsigned)(r + 1) * 8u, err)) s_abort = 1;
            } else {
                __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (!spin_until(&counters[0], (unsigned)(r + 1) * nwg, err)) s_abort = 1;
            }
        }
        __syncthreads();
        if (s_abort) break;   // (a barrier that did not complete: every workgroup leaves within one more limit)
    }
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void k_empty(unsigned* sink, unsigned v)
{
    unsigned acc = threadIdx.x * 1664525u + v;
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

int main()
{
    unsigned *counters, *err, *sink;
    CHECK(hipMalloc(&counters, 4 * 16 * 9));
    CHECK(hipMalloc(&err, 4));
    CHECK(hipMalloc(&sink, 4));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int rounds = 200;
    for (int nwg : {256, 489}) {
        if (nwg > 512) continue;   // two 1024-thread workgroups per CU at most: everything must be resident
        for (int hier = 0; hier < 2; hier++) {
            float best = 1e30f;
            unsigned herr = 0;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipMemsetAsync(counters, 0, 4 * 16 * 9, s));
                CHECK(hipMemsetAsync(err, 0, 4, s));
                CHECK(hipEventRecord(e0, s));
                if (hier) hipLaunchKernelGGL(k_rounds<true>, dim3(nwg), dim3(1024), 0, s, counters, err, rounds, sink);
                else hipLaunchKernelGGL(k_rounds<false>, dim3(nwg), dim3(1024), 0, s, counters, err, rounds, sink);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                unsigned h = 0;
                CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                herr |= h;
            }
            printf("%3d workgroups x 1024 threads, %-12s barrier: %6.2f us per round (%d rounds in one launch, best of 5)%s\n", nwg,
                   hier ? "hierarchical" : "flat", best * 1e3f / rounds, rounds, herr ? "   ** a spin ran into its limit **" : "");
        }
        // the same grid as a chain of kernels in one graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_empty, dim3(nwg), dim3(1024), 0, s, sink, (unsigned)r);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0, s));
            CHECK(hipGraphLaunch(ge, s));
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%3d workgroups x 1024 threads, kernel boundary:      %6.2f us per node (%d empty kernels in one graph, best of 5)\n", nwg, best * 1e3f / rounds, rounds);
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
    }
    return 0;
}
