// Issue cost of the vector instruction classes the trace kernels are made of, on gfx950: eight independent
// chains per lane, 4 and 8 waves per SIMD, cycles per wave instruction and SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o build_micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(a) a a a a a a a a
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const uint32_t ia = __float_as_uint(a), ib = __float_as_uint(b);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // plain f32 multiply-add
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (MODE == 1) {  // mul / add (VOP2)
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (MODE == 2) {  // compare into a scalar pair + select (two instructions per step)
            REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_gt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                              "v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_gt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 3) {  // bitwise and shifts
            REP8(asm volatile("v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %9\n v_xor_b32 %2, %2, %8\n v_lshlrev_b32 %3, 1, %3\n"
                              "v_and_b32 %4, %4, %8\n v_or_b32 %5, %5, %9\n v_xor_b32 %6, %6, %8\n v_lshrrev_b32 %7, 1, %7\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ia), "v"(ib));)
        } else if (MODE == 4) {  // integer add
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_sub_u32 %4, %4, %9\n v_sub_u32 %5, %5, %9\n v_sub_u32 %6, %6, %9\n v_sub_u32 %7, %7, %9\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ia), "v"(ib));)
        } else if (MODE == 5) {  // three-operand integer / bit ops (VOP3)
            REP8(asm volatile("v_or3_b32 %0, %0, %8, %9\n v_lshl_or_b32 %1, %1, 1, %9\n v_and_or_b32 %2, %2, %8, %9\n v_bfe_u32 %3, %3, 1, 31\n"
                              "v_or3_b32 %4, %4, %8, %9\n v_lshl_or_b32 %5, %5, 1, %9\n v_and_or_b32 %6, %6, %8, %9\n v_alignbit_b32 %7, %7, %8, 31\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ia), "v"(ib));)
        } else if (MODE == 6) {  // reciprocal, square root, reciprocal square root
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
        } else if (MODE == 7) {  // f32 ops in the three-operand encoding with modifiers (|x|, -x), max/min
            REP8(asm volatile("v_mul_f32 %0, %0, |%8|\n v_sub_f32 %1, %1, |%9|\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %9\n"
                              "v_mul_f32 %4, %4, |%8|\n v_sub_f32 %5, %5, |%9|\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (MODE == 8) {  // move
            REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n"
                              "v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));)
        } else if (MODE == 9) {  // f32 with a scalar-register operand
            REP8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                              "v_add_f32 %4, %9, %4\n v_add_f32 %5, %9, %5\n v_add_f32 %6, %9, %6\n v_add_f32 %7, %9, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "s"(b));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}

template <int MODE>
float run(float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    // blocks of 256 threads = one wave per SIMD of a CU; k blocks per CU = k waves per SIMD
    const int iters = 2048;
    float* out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const char* names[] = {"v_fma_f32", "v_mul/v_add", "v_cmp+v_cndmask", "and/or/xor/shift", "v_add_u32/v_sub_u32", "or3/lshl_or/bfe/alignbit",
                           "rcp/sqrt/rsq", "f32 VOP3 |x|, max/min", "v_mov_b32", "f32 with SGPR operand"};
    for (int w : {1, 4, 8}) {
        const int blocks = 256 * w;
        float ms[10] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters),
                        run<4>(out, blocks, iters), run<5>(out, blocks, iters), run<6>(out, blocks, iters), run<7>(out, blocks, iters),
                        run<8>(out, blocks, iters), run<9>(out, blocks, iters)};
        for (int m = 0; m < 10; ++m)
            printf("%d waves/SIMD  %-26s %8.3f ms  %.2f cycles per wave instruction and SIMD at 2.4 GHz\n", w, names[m], ms[m],
                   ms[m] * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
    }
    return 0;
}
