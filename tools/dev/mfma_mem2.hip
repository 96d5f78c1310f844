// Which memory operations make progress on a SIMD that is executing fp32 MFMAs (v_mfma_f32_16x16x4_f32)?
// Waves 0-3 of a block: dependency-free MFMAs.  Waves 4-7: MEM 1 = copy, 3 = loads only (data returned to VGPRs),
// 4 = stores only, 5 = loads only straight into LDS (global_load_lds_dwordx4, no VGPR write-back).
// Build: hipcc -O3 --offload-arch=gfx950 tools/dev/mfma_mem2.hip -o /tmp/mfma_mem2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float4* __restrict__ src, float4* __restrict__ dst, long n4,
                                         int iters, int do_mfma, int mem) {
    __shared__ float4 stage[4][4][64];                 // [streaming wave][slot][lane]
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
    } else if (mem) {
        const long t = (long)blockIdx.x * 256 + (threadIdx.x - 256), nt = (long)gridDim.x * 256;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long i = t; i + 3 * nt < n4; i += 4 * nt) {
            if (mem == 1) {
                float4 v0 = src[i], v1 = src[i + nt], v2 = src[i + 2 * nt], v3 = src[i + 3 * nt];
                dst[i] = v0; dst[i + nt] = v1; dst[i + 2 * nt] = v2; dst[i + 3 * nt] = v3;
            } else if (mem == 3) {
                float4 v0 = src[i], v1 = src[i + nt], v2 = src[i + 2 * nt], v3 = src[i + 3 * nt];
                s.x += v0.x + v1.x + v2.x + v3.x; s.y += v0.y + v1.y + v2.y + v3.y;
            } else if (mem == 4) {
                dst[i] = s; dst[i + nt] = s; dst[i + 2 * nt] = s; dst[i + 3 * nt] = s;
            } else {
                typedef __attribute__((address_space(1))) const void* gptr;
                typedef __attribute__((address_space(3))) void* lptr;
                __builtin_amdgcn_global_load_lds((gptr)(src + i), (lptr)&stage[wave - 4][0][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(src + i + nt), (lptr)&stage[wave - 4][1][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(src + i + 2 * nt), (lptr)&stage[wave - 4][2][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(src + i + 3 * nt), (lptr)&stage[wave - 4][3][0], 16, 0, 0);
                __builtin_amdgcn_s_waitcnt(0);
            }
        }
        if (mem == 5) s = stage[wave - 4][threadIdx.x & 3][threadIdx.x & 63];
        r += s.x + s.y;
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}
template <int MODE> float run(int m, int v, int iters, float4* src, float4* dst, long n4) {
    float* d; (void)hipMalloc(&d, 4096);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(512), 0, 0, d, src, dst, n4, iters, m, v);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(512), 0, 0, d, src, dst, n4, iters, m, v);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    (void)hipFree(d); return best;
}
int main() {
    const long bytes = 1L << 30;
    float4 *src, *dst; (void)hipMalloc(&src, bytes); (void)hipMalloc(&dst, bytes);
    (void)hipMemset(src, 1, bytes); (void)hipMemset(dst, 0, bytes);
    const long n4 = bytes / 16;
    const int it = 2000;
    const char* names[6] = {"", "copy", "", "loads->VGPR", "stores", "loads->LDS"};
    printf("f32 mfma alone %.3f ms   bf16 mfma alone %.3f ms\n", run<0>(1, 0, it, src, dst, n4), run<1>(1, 0, it, src, dst, n4));
    for (int mem : {1, 3, 4, 5})
        printf("%-12s alone %.3f ms   with f32 mfma %.3f ms   with bf16 mfma %.3f ms\n", names[mem], run<0>(0, mem, it, src, dst, n4),
               run<0>(1, mem, it, src, dst, n4), run<1>(1, mem, it, src, dst, n4));
    return 0;
}
