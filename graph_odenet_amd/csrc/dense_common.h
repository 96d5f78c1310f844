// dense_common.h - device helpers shared by the dense kernels of the ODE function (gemm.hip, gemm_pc.hip): GroupNorm
// arithmetic in ATen's CPU form, the exact three-way bf16 cut of an fp32 value, the LDS-only block barrier.  gfx950.
#pragma once
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// GroupNorm statistics of the 4 values a lane holds.  CG = channels per group.
// CG = 1,2,4: the group is inside the float4.  CG = 8,16: spread over lane groups
// g (xor 16 / xor 32).  mean/rstd are returned per component.
// 1/sqrt(v): v_rsq_f32 (1 ulp) + one Newton step -> within 1 ulp of the correctly rounded value; the
// IEEE sqrtf + divide sequence it replaces costs ~40 dependent VALU instructions per group.
__device__ __forceinline__ float rsqrt_nr(float v) {
    float r = __builtin_amdgcn_rsqf(v);
    r = r * fmaf(-0.5f * v * r, r, 1.5f);
    return r;
}

template <int CG>
__device__ __forceinline__ void gn_stats(const float4 x, float eps, float4& mean, float4& rstd) {
    if (CG == 1) {
        mean = x;
        const float r = rsqrt_nr(eps);
        rstd = make_float4(r, r, r, r);
    } else if (CG == 2) {
        const float m0 = (x.x + x.y) * 0.5f, m1 = (x.z + x.w) * 0.5f;
        const float v0 = ((x.x - m0) * (x.x - m0) + (x.y - m0) * (x.y - m0)) * 0.5f;
        const float v1 = ((x.z - m1) * (x.z - m1) + (x.w - m1) * (x.w - m1)) * 0.5f;
        const float r0 = rsqrt_nr(v0 + eps), r1 = rsqrt_nr(v1 + eps);
        mean = make_float4(m0, m0, m1, m1);
        rstd = make_float4(r0, r0, r1, r1);
    } else {
        float s = (x.x + x.y) + (x.z + x.w);
        if (CG >= 8) s += __shfl_xor(s, 16, 64);
        if (CG >= 16) s += __shfl_xor(s, 32, 64);
        const float m = s * (1.0f / CG);
        const float dx = x.x - m, dy = x.y - m, dz = x.z - m, dw = x.w - m;
        float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
        if (CG >= 8) q += __shfl_xor(q, 16, 64);
        if (CG >= 16) q += __shfl_xor(q, 32, 64);
        const float r = rsqrt_nr(q * (1.0f / CG) + eps);
        mean = make_float4(m, m, m, m);
        rstd = make_float4(r, r, r, r);
    }
}

// y = x*scale + shift with scale = rstd*gamma, shift = beta - mean*scale (ATen's CPU form,
// aten/src/ATen/native/cpu/group_norm_kernel.cpp), rounded step by step (no contraction).
__device__ __forceinline__ float gn_apply1(float x, float mean, float rstd, float gam, float bet) {
    const float scale = __fmul_rn(rstd, gam);
    const float shift = __fsub_rn(bet, __fmul_rn(mean, scale));
    return __fadd_rn(__fmul_rn(x, scale), shift);
}

template <int CG>
__device__ __forceinline__ float4 gn_forward(const float4 x, float eps, const float* gamma, const float* beta, int c0) {
    if (CG == 0) return x;
    float4 mean, rstd;
    gn_stats<CG>(x, eps, mean, rstd);
    float4 gm = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma) gm = ld4(gamma + c0);
    if (beta) bt = ld4(beta + c0);
    return make_float4(gn_apply1(x.x, mean.x, rstd.x, gm.x, bt.x), gn_apply1(x.y, mean.y, rstd.y, gm.y, bt.y),
                       gn_apply1(x.z, mean.z, rstd.z, gm.z, bt.z), gn_apply1(x.w, mean.w, rstd.w, gm.w, bt.w));
}

template <int CG>
__device__ __forceinline__ float4 gn_forward_v(const float4 x, float eps, const float4 gm, const float4 bt) {
    if (CG == 0) return x;
    float4 mean, rstd;
    gn_stats<CG>(x, eps, mean, rstd);
    return make_float4(gn_apply1(x.x, mean.x, rstd.x, gm.x, bt.x), gn_apply1(x.y, mean.y, rstd.y, gm.y, bt.y),
                       gn_apply1(x.z, mean.z, rstd.z, gm.z, bt.z), gn_apply1(x.w, mean.w, rstd.w, gm.w, bt.w));
}

__device__ __forceinline__ void split3_trunc(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);            // exact: the low 16 bits of x's significand
    m = __float_as_uint(r1) & 0xffff0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));       // exact, at most 8 significant bits: its low half is zero
}
// the bf16 halves (upper 16 bits) of four fp32 words -> 8 bytes
__device__ __forceinline__ uint2 pack_hi16x4(unsigned u0, unsigned u1, unsigned u2, unsigned u3) {
    return make_uint2(__builtin_amdgcn_perm(u1, u0, 0x07060302u), __builtin_amdgcn_perm(u3, u2, 0x07060302u));
}
// Block barrier that orders LDS traffic only: __syncthreads() also drains the wave's outstanding GLOBAL loads
// (s_waitcnt vmcnt(0)), which would turn every barrier into a wait for the tile being prefetched.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace
