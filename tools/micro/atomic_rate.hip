// Microbenchmark: what the per-wave queue appends of a level cost.  Every wave of a launch does `per_wave`
// returning atomic adds (lane 0, as the trace kernels' compaction does) on one counter, on two counters
// in one 128-byte line, or on two counters in different lines; or the same adds without using the result.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_rate.hip -o build_micro/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>  // 0: one counter, 1: two in a line, 2: two in different lines, 3: one counter, result unused
__global__ void __launch_bounds__(128) k_atomic(uint32_t* ctr, uint32_t* out, uint32_t spin) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t v = 0;
    // (some arithmetic first so that the waves do not all arrive in the first microsecond)
    float x = (float)threadIdx.x;
    for (uint32_t s = 0; s < spin; ++s) x = x * 1.0001f + 0.5f;
    if (lane == 0) {
        if (MODE == 3) { atomicAdd(&ctr[0], 3u); }
        else {
            v = atomicAdd(&ctr[0], 3u);
            if (MODE == 1) v += atomicAdd(&ctr[1], 5u);
            if (MODE == 2) v += atomicAdd(&ctr[1024], 5u);
        }
    }
    v = __builtin_amdgcn_readfirstlane(v);
    out[blockIdx.x * 128u + threadIdx.x] = v + (uint32_t)x;
}

int main() {
    const uint32_t blocks = 64800;  // 3840 x 2160 rays in blocks of 128
    uint32_t *ctr, *out;
    (void)hipMalloc(&ctr, 8192); (void)hipMemset(ctr, 0, 8192);
    (void)hipMalloc(&out, (size_t)blocks * 128 * 4);
    for (uint32_t spin : {0u, 2000u}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(128), 0, 0, ctr, out, spin);
                else if (mode == 1) hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(128), 0, 0, ctr, out, spin);
                else if (mode == 2) hipLaunchKernelGGL(k_atomic<2>, dim3(blocks), dim3(128), 0, 0, ctr, out, spin);
                else hipLaunchKernelGGL(k_atomic<3>, dim3(blocks), dim3(128), 0, 0, ctr, out, spin);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                (void)hipEventElapsedTime(&ms, a, b);
            }
            printf("spin %5u  %-28s %8.3f ms for %u waves (%6.2f ns per wave)\n", spin,
                   mode == 0 ? "one counter" : mode == 1 ? "two counters, one line" : mode == 2 ? "two counters, two lines" : "one counter, no result",
                   ms, blocks * 2, ms * 1e6 / (blocks * 2));
        }
    }
    return 0;
}
