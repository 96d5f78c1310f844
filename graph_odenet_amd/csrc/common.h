// common.h — shared device helpers for libgraphode (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/graphode.h"

#define GODE_WAVE 64

// Device-side copy of gode_lincomb_t (passed by value as a kernel argument).
struct LinComb {
    int n;
    float coef[GODE_MAX_TERMS];
    const float* ptr[GODE_MAX_TERMS];
};

static inline LinComb make_lincomb(const gode_lincomb_t* h) {
    LinComb lc;
    lc.n = 0;
    for (int j = 0; j < GODE_MAX_TERMS; ++j) { lc.coef[j] = 0.f; lc.ptr[j] = nullptr; }
    if (h) {
        lc.n = h->n;
        for (int j = 0; j < h->n && j < GODE_MAX_TERMS; ++j) { lc.coef[j] = h->coef[j]; lc.ptr[j] = h->ptr[j]; }
    }
    return lc;
}

static inline int check_lincomb(const gode_lincomb_t* h, bool required) {
    if (!h) return required ? GODE_E_NULLPTR : 0;
    if (h->n < 0 || h->n > GODE_MAX_TERMS) return GODE_E_RANGE;
    if (required && h->n == 0) return GODE_E_RANGE;
    for (int j = 0; j < h->n; ++j) if (!h->ptr[j]) return GODE_E_NULLPTR;
    return 0;
}

static inline bool lincomb_aligned16(const gode_lincomb_t* h) {
    if (!h) return true;
    for (int j = 0; j < h->n; ++j) if (((uintptr_t)h->ptr[j]) & 15) return false;
    return true;
}

// Straight-line combination of NT terms: all NT loads are issued before the first use, so their
// latencies overlap (a loop over a run-time term count makes hipcc wait for each load in turn).
template <int NT>
__device__ __forceinline__ float4 lc_load4_n(const LinComb& lc, int64_t idx) {
    float4 v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) v[j] = *reinterpret_cast<const float4*>(lc.ptr[j] + idx);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float c = lc.coef[j];
        r.x = fmaf(c, v[j].x, r.x); r.y = fmaf(c, v[j].y, r.y);
        r.z = fmaf(c, v[j].z, r.z); r.w = fmaf(c, v[j].w, r.w);
    }
    return r;
}

__device__ __forceinline__ float4 lc_load4(const LinComb& lc, int64_t idx) {
    switch (lc.n) {          // wave-uniform
        case 1: return lc_load4_n<1>(lc, idx);
        case 2: return lc_load4_n<2>(lc, idx);
        case 3: return lc_load4_n<3>(lc, idx);
        case 4: return lc_load4_n<4>(lc, idx);
        case 5: return lc_load4_n<5>(lc, idx);
        case 6: return lc_load4_n<6>(lc, idx);
        case 7: return lc_load4_n<7>(lc, idx);
        case 8: return lc_load4_n<8>(lc, idx);
        default: return make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ float lc_load1(const LinComb& lc, int64_t idx) {
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < GODE_MAX_TERMS; ++j)
        if (j < lc.n) r = fmaf(lc.coef[j], lc.ptr[j][idx], r);
    return r;
}

// When the first coefficient is exactly 1 the first term is taken as-is so that
// a one-term {1.0, y} combination reproduces y bit for bit.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// gemm.hip: raise a kernel's dynamic-LDS limit once per (device, kernel); thread-safe, a no-op up to 64 KB
int gode_set_lds_once(const void* fn, size_t bytes);
// rk.hip: out[0..n) = 0 by a kernel (never a memset node)
int gode_zero_f32(float* out, int64_t n, void* stream);

#define GODE_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
