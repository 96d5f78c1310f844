// segment.hip — per-graph (segment) attention readout of Set2Set, forward and backward.
//
// Reference: QC/set2set.py:59-75.  One processing step of Set2Set is, per graph b with query q_b,
//     e_i = <x_i, q_b>,  a = softmax(e over the nodes of b),  r_b = sum_i a_i x_i
// which the reference evaluates with a Python loop over the graphs of the batch (masked_select + softmax +
// masked assignment per graph) followed by a scatter_add.  Here one workgroup owns one graph: the four waves
// stride over its nodes, the 64 lanes over the feature columns, logits / maxima / sums are wave and block
// reductions, and nothing of size N x h is written in the forward pass.
#include "common.h"
#include "prof.h"

namespace {

constexpr int kWaves = 4;

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {   // v wave-uniform
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kWaves; ++j) s += red[j];
    return s;
}

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int j = 1; j < kWaves; ++j) s = fmaxf(s, red[j]);
    return s;
}

// MC = feature chunks of 64 per lane (h <= 64*MC).
template <int MC>
__global__ __launch_bounds__(256) void seg_attn_fwd_kernel(const int32_t* __restrict__ segptr,
                                                           const int32_t* __restrict__ perm,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ q, int h,
                                                           float* __restrict__ a, float* __restrict__ r) {
    __shared__ float red[kWaves];
    __shared__ float racc[kWaves][MC * 64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int beg = segptr[b], end = segptr[b + 1];
    if (end <= beg) {                                       // graph without nodes: r = 0
        for (int f = threadIdx.x; f < h; f += 256) r[(int64_t)b * h + f] = 0.f;
        return;
    }
    float qv[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) { const int f = lane + 64 * c; qv[c] = f < h ? q[(int64_t)b * h + f] : 0.f; }

    // pass 1: logits (kept in a[]) and their maximum
    float m = -INFINITY;
    for (int k = beg + w; k < end; k += kWaves) {
        const int node = perm ? perm[k] : k;
        const float* xr = x + (int64_t)node * ldx;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c) { const int f = lane + 64 * c; if (f < h) dot = fmaf(xr[f], qv[c], dot); }
        dot = wave_sum(dot);
        if (lane == 0) a[node] = dot;
        m = fmaxf(m, dot);
    }
    m = block_reduce_max(m, red);

    // pass 2: weights, their sum and the weighted feature sum (every wave revisits the nodes it wrote)
    float s = 0.f, acc[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[c] = 0.f;
    for (int k = beg + w; k < end; k += kWaves) {
        const int node = perm ? perm[k] : k;
        const float* xr = x + (int64_t)node * ldx;
        float e = (lane == 0) ? a[node] : 0.f;
        e = __shfl(e, 0, 64);
        const float wgt = expf(e - m);
        s += wgt;
#pragma unroll
        for (int c = 0; c < MC; ++c) { const int f = lane + 64 * c; if (f < h) acc[c] = fmaf(wgt, xr[f], acc[c]); }
        if (lane == 0) a[node] = wgt;
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) racc[w][lane + 64 * c] = acc[c];
    const float S = block_reduce_sum(s, red);               // also orders racc[] and a[] for the block
    const float inv = 1.f / S;
    for (int f = threadIdx.x; f < h; f += 256) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < kWaves; ++j) t += racc[j][f];
        r[(int64_t)b * h + f] = t * inv;
    }
    for (int k = beg + threadIdx.x; k < end; k += 256) {
        const int node = perm ? perm[k] : k;
        a[node] *= inv;
    }
}

// Backward of (x, q) -> r with a saved:  da_i = <x_i, dr>,  S = sum_i a_i da_i,  de_i = a_i (da_i - S),
//   dx_i = a_i dr + de_i q,   dq = sum_i de_i x_i.
template <int MC>
__global__ __launch_bounds__(256) void seg_attn_bwd_kernel(const int32_t* __restrict__ segptr,
                                                           const int32_t* __restrict__ perm,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ q, const float* __restrict__ a,
                                                           const float* __restrict__ dr, int h,
                                                           float* __restrict__ dx, float* __restrict__ dq) {
    __shared__ float red[kWaves];
    __shared__ float qacc[kWaves][MC * 64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int beg = segptr[b], end = segptr[b + 1];
    if (end <= beg) {
        for (int f = threadIdx.x; f < h; f += 256) dq[(int64_t)b * h + f] = 0.f;
        return;
    }
    float qv[MC], gv[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        const int f = lane + 64 * c;
        qv[c] = f < h ? q[(int64_t)b * h + f] : 0.f;
        gv[c] = f < h ? dr[(int64_t)b * h + f] : 0.f;
    }
    float t = 0.f;
    for (int k = beg + w; k < end; k += kWaves) {
        const int node = perm ? perm[k] : k;
        const float* xr = x + (int64_t)node * ldx;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c) { const int f = lane + 64 * c; if (f < h) dot = fmaf(xr[f], gv[c], dot); }
        t = fmaf(a[node], wave_sum(dot), t);
    }
    const float S = block_reduce_sum(t, red);

    float acc[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[c] = 0.f;
    for (int k = beg + w; k < end; k += kWaves) {
        const int node = perm ? perm[k] : k;
        const float* xr = x + (int64_t)node * ldx;
        float xv[MC], dot = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c) { const int f = lane + 64 * c; xv[c] = f < h ? xr[f] : 0.f; dot = fmaf(xv[c], gv[c], dot); }
        const float ai = a[node];
        const float de = ai * (wave_sum(dot) - S);
        float* dxr = dx + (int64_t)node * h;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            const int f = lane + 64 * c;
            if (f < h) dxr[f] = fmaf(ai, gv[c], de * qv[c]);
            acc[c] = fmaf(de, xv[c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) qacc[w][lane + 64 * c] = acc[c];
    __syncthreads();
    for (int f = threadIdx.x; f < h; f += 256) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < kWaves; ++j) v += qacc[j][f];
        dq[(int64_t)b * h + f] = v;
    }
}

}  // namespace

#define GODE_SEG_DISPATCH(KERNEL, ...)                                                                   \
    do {                                                                                                 \
        const int mc = (int)((h + 63) / 64);                                                             \
        if (mc <= 1) hipLaunchKernelGGL(KERNEL<1>, dim3((unsigned)n_seg), dim3(256), 0, s, __VA_ARGS__); \
        else if (mc <= 2) hipLaunchKernelGGL(KERNEL<2>, dim3((unsigned)n_seg), dim3(256), 0, s, __VA_ARGS__); \
        else if (mc <= 4) hipLaunchKernelGGL(KERNEL<4>, dim3((unsigned)n_seg), dim3(256), 0, s, __VA_ARGS__); \
        else if (mc <= 8) hipLaunchKernelGGL(KERNEL<8>, dim3((unsigned)n_seg), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<16>, dim3((unsigned)n_seg), dim3(256), 0, s, __VA_ARGS__);        \
    } while (0)

extern "C" int gode_segment_attention_f32_fwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx,
                                              const float* q, int64_t n_seg, int64_t h, float* a, float* r,
                                              void* stream) {
    if (n_seg < 0 || h <= 0 || ldx < h) return GODE_E_SHAPE;
    if (n_seg == 0) return 0;
    if (!segptr || !x || !q || !a || !r) return GODE_E_NULLPTR;
    if (h > 1024) return GODE_E_UNSUPPORTED;
    if (n_seg > INT32_MAX) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    GODE_SEG_DISPATCH(seg_attn_fwd_kernel, segptr, perm, x, ldx, q, (int)h, a, r);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_segment_attention_f32_bwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx,
                                              const float* q, const float* a, const float* dr, int64_t n_seg,
                                              int64_t h, float* dx, float* dq, void* stream) {
    if (n_seg < 0 || h <= 0 || ldx < h) return GODE_E_SHAPE;
    if (n_seg == 0) return 0;
    if (!segptr || !x || !q || !a || !dr || !dx || !dq) return GODE_E_NULLPTR;
    if (h > 1024) return GODE_E_UNSUPPORTED;
    if (n_seg > INT32_MAX) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    GODE_SEG_DISPATCH(seg_attn_bwd_kernel, segptr, perm, x, ldx, q, a, dr, (int)h, dx, dq);
    GODE_LAUNCH_CHECK();
    return 0;
}
