// Do fp32 MFMAs and global-memory streaming overlap on one CU?  Waves 0-3 of a block (one per SIMD) issue
// dependency-free v_mfma_f32_16x16x4_f32 (MODE 0) or v_mfma_f32_16x16x32_bf16 (MODE 1); waves 4-7 (their SIMD
// partners) stream-copy a large buffer with 16-byte accesses.  overlap => t(both) ~ max, shared => t(both) ~ sum.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dev/mfma_mem.hip -o /tmp/mfma_mem
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float4* __restrict__ src, float4* __restrict__ dst, long n4,
                                         int iters, int do_mfma, int do_mem) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (do_mfma) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
            float a = threadIdx.x * 1e-3f, b = 1.0f;
            bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
            for (int it = 0; it < iters; ++it) {
                if (MODE == 0) { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0); }
                else { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc[i], 0, 0, 0); }
            }
            for (int i = 0; i < 8; ++i) r += acc[i][0];
        }
    } else if (do_mem) {
        if (do_mem == 2) __builtin_amdgcn_s_setprio(3);      // streaming waves above the MFMA waves
        const long t = (long)blockIdx.x * 256 + (threadIdx.x - 256), nt = (long)gridDim.x * 256;
        for (long i = t; i < n4; i += 4 * nt) {          // 4 loads in flight per lane
            float4 v0 = src[i], v1, v2, v3;
            const bool b1 = i + nt < n4, b2 = i + 2 * nt < n4, b3 = i + 3 * nt < n4;
            if (b1) v1 = src[i + nt]; if (b2) v2 = src[i + 2 * nt]; if (b3) v3 = src[i + 3 * nt];
            dst[i] = v0; if (b1) dst[i + nt] = v1; if (b2) dst[i + 2 * nt] = v2; if (b3) dst[i + 3 * nt] = v3;
        }
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}
template <int MODE> float run(int m, int v, int iters, float4* src, float4* dst, long n4) {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(512), 0, 0, d, src, dst, n4, iters, m, v);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(512), 0, 0, d, src, dst, n4, iters, m, v);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    hipFree(d); return best;
}
int main() {
    const long bytes = 1L << 30;                       // 1 GiB read + 1 GiB written per launch
    float4 *src, *dst; hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    const long n4 = bytes / 16;
    // 512 blocks of 8 waves = 2 blocks per CU: 2 MFMA waves + 2 streaming waves per SIMD (the dense kernels' occupancy)
    for (int it : {2000, 4000}) {
        printf("f32 16x16x4  iters %d: mfma %.3f ms  mem %.3f ms (%.2f TB/s)  both %.3f ms\n", it, run<0>(1, 0, it, src, dst, n4),
               run<0>(0, 1, it, src, dst, n4), 2.0 * bytes / run<0>(0, 1, it, src, dst, n4) * 1e-9, run<0>(1, 1, it, src, dst, n4));
        printf("f32 16x16x4  iters %d, streaming waves at s_setprio 3: both %.3f ms\n", it, run<0>(1, 2, it, src, dst, n4));
        printf("bf16 16x16x32 iters %d: mfma %.3f ms  mem %.3f ms  both %.3f ms\n", it, run<1>(1, 0, it, src, dst, n4),
               run<1>(0, 1, it, src, dst, n4), run<1>(1, 1, it, src, dst, n4));
    }
    return 0;
}
