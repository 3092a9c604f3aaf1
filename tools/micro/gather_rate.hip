// Microbenchmark: what a wave pays for eight 16-byte loads per lane from a lane-specific 128-byte
// record group ("divergent": 64 different lines per load instruction, the leaf walk's access
// pattern) against the same bytes fetched co-operatively, eight lanes per line ("octet": lane i of
// an octet loads bytes 16 i .. 16 i + 15 of the group that lane k of the octet wants, k = 0..7: 8
// lines per load instruction).  The next group depends on what was loaded, as in the walk.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/gather_rate.hip -o tools/micro/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64, 5) k_divergent(const float4* __restrict__ table, uint32_t n_groups, uint32_t steps, uint32_t* out) {
    uint32_t g = (blockIdx.x * 64u + threadIdx.x) * 2654435761u % n_groups;
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s) {
        const float4* p = table + (size_t)g * 8u;
        float4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = p[k];
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m += r[k].x * r[k].y + r[k].z * r[k].w;
        acc += m;
        g = (g * 1664525u + 1013904223u + (uint32_t)(m > 1e30f)) % n_groups;
    }
    out[blockIdx.x * 64u + threadIdx.x] = __float_as_uint(acc);
}

__global__ void __launch_bounds__(64, 5) k_octet(const float4* __restrict__ table, uint32_t n_groups, uint32_t steps, uint32_t* out) {
    const uint32_t lane = threadIdx.x, sub = lane & 7u, base = lane & ~7u;
    uint32_t g = (blockIdx.x * 64u + threadIdx.x) * 2654435761u % n_groups;
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s) {
        float m = 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t gk = (uint32_t)__shfl((int)g, (int)(base + k), 64);  // the group lane k of my octet wants
            const float4 r = table[(size_t)gk * 8u + sub];
            const float v = r.x * r.y + r.z * r.w;
            // (each lane holds one record's value for owner k; the owner needs the sum over its octet)
            float t = v;
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
            m = (sub == k) ? t : m;
        }
        acc += m;
        g = (g * 1664525u + 1013904223u + (uint32_t)(m > 1e30f)) % n_groups;
    }
    out[blockIdx.x * 64u + threadIdx.x] = __float_as_uint(acc);
}

// All lanes of a wave want the same group (coherent rays): the same eight vector loads (COH 64), or with
// a group per 8 lanes (COH 8), or only every fourth lane active (QUARTER); and the group fetched through
// the scalar cache instead (k_scalar), the arithmetic taking its operands from scalar registers.
template <uint32_t COH, bool QUARTER>
__global__ void __launch_bounds__(64, 5) k_coherent(const float4* __restrict__ table, uint32_t n_groups, uint32_t steps, uint32_t* out) {
    uint32_t g = (blockIdx.x * 64u + (threadIdx.x / COH) * COH) * 2654435761u % n_groups;
    float acc = 0.0f;
    if (QUARTER && (threadIdx.x & 3u) != 0u) return;
    for (uint32_t s = 0; s < steps; ++s) {
        const float4* p = table + (size_t)g * 8u;
        float4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = p[k];
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m += r[k].x * r[k].y + r[k].z * r[k].w;
        acc += m;
        g = (g * 1664525u + 1013904223u + (uint32_t)(m > 1e30f)) % n_groups;
    }
    out[blockIdx.x * 64u + threadIdx.x] = __float_as_uint(acc);
}

__global__ void __launch_bounds__(64, 5) k_scalar(const float4* __restrict__ table, uint32_t n_groups, uint32_t steps, uint32_t* out) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef const __attribute__((address_space(4))) f32x16* cf16_ptr;
    uint32_t g = (blockIdx.x * 64u) * 2654435761u % n_groups;
    float acc = 0.0f;
    const float lanef = (float)threadIdx.x;
    for (uint32_t s = 0; s < steps; ++s) {
        const cf16_ptr p = (cf16_ptr)(table + (size_t)__builtin_amdgcn_readfirstlane(g) * 8u);
        const f32x16 a = p[0], b = p[1];
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) m += (a[4 * k] + lanef) * a[4 * k + 1] + a[4 * k + 2] * a[4 * k + 3];
#pragma unroll
        for (int k = 0; k < 4; ++k) m += (b[4 * k] + lanef) * b[4 * k + 1] + b[4 * k + 2] * b[4 * k + 3];
        acc += m;
        g = (g * 1664525u + 1013904223u + (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m > 1e30f))) % n_groups;
    }
    out[blockIdx.x * 64u + threadIdx.x] = __float_as_uint(acc);
}

// The table staged in LDS (it must fit), each lane reading its group's eight records; ROT: in an
// order rotated by the lane's number, so that the lanes of a 16-lane LDS group do not all ask for
// the same two 16-byte slots of the 256-byte bank row at the same time.
template <bool ROT>
__global__ void __launch_bounds__(256, 4) k_lds(const float4* __restrict__ table, uint32_t n_groups, uint32_t steps, uint32_t* out) {
    extern __shared__ float4 lds[];
    for (uint32_t i = threadIdx.x; i < n_groups * 8u; i += 256u) lds[i] = table[i];
    __syncthreads();
    uint32_t g = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_groups;
    const uint32_t rot = ROT ? (threadIdx.x & 7u) : 0u;
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s) {
        float4 r[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) r[k] = lds[g * 8u + ((k + rot) & 7u)];
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m += r[k].x * r[k].y + r[k].z * r[k].w;
        acc += m;
        g = (g * 1664525u + 1013904223u + (uint32_t)(m > 1e30f)) % n_groups;
    }
    out[blockIdx.x * 256u + threadIdx.x] = __float_as_uint(acc);
}

int main(int argc, char** argv) {
    const uint32_t steps = 2000;
    for (uint32_t kb : {16u, 128u, 1024u}) {
        const uint32_t n_groups = kb * 1024u / 128u;
        std::vector<float> h((size_t)n_groups * 32, 1.0f);
        float4* d; uint32_t* o;
        hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const uint32_t blocks = 256 * 4 * 5;  // 5 waves per SIMD on every CU
        hipMalloc(&o, blocks * 64 * 4);
        for (int variant = 0; variant < 8; ++variant) {
            if ((variant == 2 || variant == 3) && kb > 16) continue;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (variant == 0) hipLaunchKernelGGL(k_divergent, dim3(blocks), dim3(64), 0, 0, d, n_groups, steps, o);
                else if (variant == 1) hipLaunchKernelGGL(k_octet, dim3(blocks), dim3(64), 0, 0, d, n_groups, steps, o);
                else if (variant == 2) hipLaunchKernelGGL(k_lds<false>, dim3(blocks / 4), dim3(256), kb * 1024, 0, d, n_groups, steps, o);
                else if (variant == 3) hipLaunchKernelGGL(k_lds<true>, dim3(blocks / 4), dim3(256), kb * 1024, 0, d, n_groups, steps, o);
                else if (variant == 4) hipLaunchKernelGGL((k_coherent<64, false>), dim3(blocks), dim3(64), 0, 0, d, n_groups, steps, o);
                else if (variant == 5) hipLaunchKernelGGL((k_coherent<8, false>), dim3(blocks), dim3(64), 0, 0, d, n_groups, steps, o);
                else if (variant == 6) hipLaunchKernelGGL((k_coherent<1, true>), dim3(blocks), dim3(64), 0, 0, d, n_groups, steps, o);
                else hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(64), 0, 0, d, n_groups, steps, o);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            // cycles per step per SIMD round (5 waves per SIMD take turns): ms * 2.4e6 / steps
            printf("table %5u KiB  %-9s %8.3f ms  %7.1f cycles per wave-step (5 waves/SIMD sharing), %6.1f GB/s/CU of records\n", kb,
                   variant == 0 ? "divergent" : variant == 1 ? "octet" : variant == 2 ? "lds" : variant == 3 ? "lds rot" : variant == 4 ? "same grp" : variant == 5 ? "grp per 8" : variant == 6 ? "1 in 4 on" : "scalar", ms, ms * 2.4e6 / steps / 5.0,
                   (double)blocks * 64 * steps * 128 / (ms * 1e-3) / 256 / 1e9);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
