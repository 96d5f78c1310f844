// launch_probe.hip - what a launch-bound kernel costs on this box, by ingredient (development aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/launch_probe.hip -o /tmp/launch_probe && /tmp/launch_probe
// Each line: a chain of REP dependent launches of one kernel on one stream, replayed from a HIP graph; time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { int n; float coef[8]; const float* ptr[8]; };

__global__ void k_empty() {}
__global__ void k_store(float* out, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = 1.f; }
__global__ void k_copy(float* out, const float* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = a[i] + 1.f; }
__global__ void k_copy4(float4* out, const float4* a, int n4) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n4) { float4 v = a[i]; v.x += 1.f; out[i] = v; } }
__global__ void k_big(float* out, Big b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float r = 0.f; for (int j = 0; j < 8; ++j) if (j < b.n) r = fmaf(b.coef[j], b.ptr[j][i], r); out[i] = r; }
}
__global__ void k_gather1(float* out, const int* idx, const float* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = a[idx[i]]; }
__global__ void k_gather2(float* out, const int* idx, const int* idx2, const float* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = a[idx2[idx[i]]]; }
__global__ void k_gather3(float* out, const int* idx, const int* idx2, const int* idx3, const float* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = a[idx3[idx2[idx[i]]]]; }
__global__ void k_lds(float* out, const float* a, int n) {
    __shared__ float w[17 * 36];
    for (int i = threadIdx.x; i < 17 * 36; i += blockDim.x) w[i] = a[i];
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = w[i % (17 * 36)] + a[i];
}

template <typename F>
int timed(const char* name, hipStream_t s, F launch, int rep = 512, int replays = 20) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < rep; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // the same launches issued eagerly
    CK(hipEventRecord(a, s));
    for (int i = 0; i < rep * 4; ++i) launch();
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms2; CK(hipEventElapsedTime(&ms2, a, b));
    printf("%-46s graph %6.2f us/launch   eager %6.2f us/launch\n", name, 1e3 * ms / (rep * replays), 1e3 * ms2 / (rep * 4));
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return 0;
}

int main() {
    const int n = 3327 * 16;          // Citeseer x hidden 16
    hipStream_t s; CK(hipStreamCreate(&s));
    float *a, *b; int *i1, *i2, *i3;
    CK(hipMalloc(&a, n * 4 * 8)); CK(hipMalloc(&b, n * 4 * 8)); CK(hipMalloc(&i1, n * 4)); CK(hipMalloc(&i2, n * 4)); CK(hipMalloc(&i3, n * 4));
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919) % n);
    CK(hipMemcpy(i1, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(i2, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(i3, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(a, 0, n * 4 * 8)); CK(hipMemset(b, 0, n * 4 * 8));
    const int blk = (n + 255) / 256;
    int flip = 0;
    // ping-pong buffers: every launch reads what the previous one wrote (as the stages of a solve do)
    auto pp = [&](float*& src, float*& dst) { src = flip ? b : a; dst = flip ? a : b; flip ^= 1; };
    timed("empty, 1 block", s, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    timed("empty, 832 blocks x 256", s, [&] { hipLaunchKernelGGL(k_empty, dim3(832), dim3(256), 0, s); });
    timed("store only, n = 53 232 (208 blocks)", s, [&] { hipLaunchKernelGGL(k_store, dim3(blk), dim3(256), 0, s, b, n); });
    timed("copy previous output (1 load level)", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(blk), dim3(256), 0, s, y, x, n); });
    timed("copy, 16-byte accesses (52 blocks)", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy4, dim3((n / 4 + 255) / 256), dim3(256), 0, s, (float4*)y, (const float4*)x, n / 4); });
    timed("copy, constant input (L2-resident)", s, [&] { hipLaunchKernelGGL(k_copy, dim3(blk), dim3(256), 0, s, b, a, n); });
    timed("104-byte struct argument, 1 term", s, [&] { float *x, *y; pp(x, y); Big g = {}; g.n = 1; g.coef[0] = 1.f; g.ptr[0] = x; hipLaunchKernelGGL(k_big, dim3(blk), dim3(256), 0, s, y, g, n); });
    timed("104-byte struct argument, 4 terms", s, [&] { float *x, *y; pp(x, y); Big g = {}; g.n = 4; for (int j = 0; j < 4; ++j) { g.coef[j] = 0.25f; g.ptr[j] = x + (size_t)j * n; } hipLaunchKernelGGL(k_big, dim3(blk), dim3(256), 0, s, y, g, n); });
    timed("gather: 2 dependent load levels", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_gather1, dim3(blk), dim3(256), 0, s, y, i1, x, n); });
    timed("gather: 3 dependent load levels", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_gather2, dim3(blk), dim3(256), 0, s, y, i1, i2, x, n); });
    timed("gather: 4 dependent load levels", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_gather3, dim3(blk), dim3(256), 0, s, y, i1, i2, i3, x, n); });
    timed("stage 2.4 KB in LDS + barrier, then copy", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_lds, dim3(blk), dim3(256), 0, s, y, x, n); });
    timed("copy on 832 blocks x 64 threads", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(832), dim3(64), 0, s, y, x, n); });
    timed("copy on 13 blocks x 1024 threads... n/4", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy4, dim3(13), dim3(1024), 0, s, (float4*)y, (const float4*)x, n / 4); });
    // which ingredient made the 16-byte copy slow?
    timed("copy4, 208 blocks x 64 threads", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy4, dim3(208), dim3(64), 0, s, (float4*)y, (const float4*)x, n / 4); });
    timed("copy4, 104 blocks x 128 threads", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy4, dim3(104), dim3(128), 0, s, (float4*)y, (const float4*)x, n / 4); });
    timed("copy4, constant input (no ping-pong)", s, [&] { hipLaunchKernelGGL(k_copy4, dim3(52), dim3(256), 0, s, (float4*)b, (const float4*)a, n / 4); });
    timed("scalar copy, 52 blocks x 1024 threads", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(52), dim3(1024), 0, s, y, x, n); });
    timed("scalar copy, 26 blocks x 256 (n/8 elements)", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(26), dim3(256), 0, s, y, x, n / 8); });
    timed("scalar copy, 1 block x 256", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, s, y, x, 256); });
    timed("scalar copy, 8 blocks x 256", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(8), dim3(256), 0, s, y, x, 2048); });
    timed("scalar copy, 64 blocks x 256", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s, y, x, 64 * 256); });
    timed("scalar copy, 832 blocks x 256 (4 n)", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy, dim3(832), dim3(256), 0, s, y, x, 4 * n); });
    timed("copy4, 832 blocks x 256 (16 n)", s, [&] { float *x, *y; pp(x, y); hipLaunchKernelGGL(k_copy4, dim3(832), dim3(256), 0, s, (float4*)y, (const float4*)x, 2 * n); });
    return 0;
}
