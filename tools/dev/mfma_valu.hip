// Does v_mfma_f32_16x16x4_f32 share the SIMD's fp32 ALUs with VALU work?  Waves 0-3 of a block (one per SIMD)
// issue dependent-free MFMAs, waves 4-7 (their SIMD partners) issue v_fma_f32 chains.  Compare t(mfma only),
// t(valu only), t(both): overlap => t(both) ~ max, shared => t(both) ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0 = f32 16x16x4, 1 = bf16 16x16x32, 2 = f32 32x32x2
__global__ __launch_bounds__(512) void k(float* out, int iters, int do_mfma, int do_valu) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (do_mfma) {
            f32x4 acc[8]; f32x16 acc32[4];
            for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
            float a = threadIdx.x * 1e-3f, b = 1.0f;
            bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
            for (int it = 0; it < iters; ++it) {
                if (MODE == 0) { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0); }
                else if (MODE == 1) { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc[i], 0, 0, 0); }
                else { for (int i = 0; i < 4; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[i], 0, 0, 0); }
            }
            for (int i = 0; i < 8; ++i) r += acc[i][0];
            for (int i = 0; i < 4; ++i) r += acc32[i][0];
        }
    } else if (do_valu) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
        for (int it = 0; it < iters; ++it)
            for (int q = 0; q < 8; ++q)        // 64 independent-ish fmas per iteration
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
        for (int i = 0; i < 8; ++i) r += x[i];
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}
template <int MODE> float run(int m, int v, int iters) {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, m, v);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, m, v);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(d); return ms;
}
int main() {
    const int it = 20000;
    printf("f32 16x16x4 : mfma %.3f ms  valu %.3f ms  both %.3f ms\n", run<0>(1, 0, it), run<0>(0, 1, it), run<0>(1, 1, it));
    printf("f32 32x32x2 : mfma %.3f ms  valu %.3f ms  both %.3f ms\n", run<2>(1, 0, it), run<2>(0, 1, it), run<2>(1, 1, it));
    printf("bf16 16x16x32: mfma %.3f ms  valu %.3f ms  both %.3f ms\n", run<1>(1, 0, it), run<1>(0, 1, it), run<1>(1, 1, it));
    return 0;
}
