// gemm.hip — dense part of the ODE function: S = [t | GroupNorm(x)] * W, its VJP and
// the weight gradient, on fp32-input MFMA (v_mfma_f32_16x16x4_f32, exact fp32) — gfx950.
//
// Replaces GCN/models.py:175-177 (GroupNorm, time column, concat) + GCN/layers.py:70
// (torch.mm) of the reference, and their autograd.
//
// Layout of the fast path (d_in = d_out = d in {16,32,64,128}, GroupNorm(min(32,d), d)):
//   * W (without its time row) lives in LDS for the whole launch, row stride d+4 floats
//     so that the two k-rows a 32-lane group reads fall on disjoint banks;
//   * the x operand never goes through LDS: lane (r = lane&15, g = lane>>4) loads
//     float4 x[row0+r][16j+4g .. +3]; these are exactly the B-operand values of the
//     MFMA whose k-slot g carries k = 16j+4g+c, so the stage combination
//     (y + h*sum a_j k_j), GroupNorm (a group of 4 channels = one float4) and the MFMA
//     feed all happen in registers;
//   * the product is formed transposed, D[n][row], so each lane ends with 4 consecutive
//     output columns of one row -> one 16-byte store, and (in the VJP) one GroupNorm
//     group per accumulator register quad.
// Bounds: fwd/bwd-data read + write one N x d operand each (HBM) against
// 2*N*d*d flop on the fp32 MFMA pipe (157 TFLOP/s peak) - near the ridge at d=128.
#include <map>
#include <mutex>
#include <utility>
#include "common.h"
#include "dense_common.h"
#include "dense_pc.h"
#include "options.h"
#include "prof.h"

namespace {

constexpr int kMaxBlocks = 512;   // 2 blocks per CU
constexpr int64_t kWgradSplitMinRows = 65536;   // below: the fp32-MFMA weight-gradient kernel (2 blocks per CU, 32 KB of LDS each)

// Block prologue: W1 (d x d, without the time row) -> LDS with row stride d+4, as 16-byte loads that are all
// in flight together (TRANSPOSE: Wlds[n][i] = W1[i][n] for the VJP); gamma / beta / time row -> LDS vectors.
template <int D, int NTHREADS, bool TRANSPOSE>
__device__ __forceinline__ void fill_w_lds(float* Wl, const float* __restrict__ W, int has_time) {
    constexpr int LDW = D + 4;
    constexpr int N4 = D * D / 4;
    constexpr int IT = (N4 + NTHREADS - 1) / NTHREADS;
    float4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < N4) v[it] = ld4(W + (int64_t)has_time * D + (int64_t)idx * 4);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = threadIdx.x + it * NTHREADS;
        if (idx < N4) {
            const int k = idx / (D / 4), n = (idx % (D / 4)) * 4;
            if (!TRANSPOSE) {
                *reinterpret_cast<float4*>(Wl + k * LDW + n) = v[it];
            } else {
                Wl[(n + 0) * LDW + k] = v[it].x; Wl[(n + 1) * LDW + k] = v[it].y;
                Wl[(n + 2) * LDW + k] = v[it].z; Wl[(n + 3) * LDW + k] = v[it].w;
            }
        }
    }
}
template <int D, int NTHREADS>
__device__ __forceinline__ void fill_vec_lds(float* dst, const float* __restrict__ src, float fill) {
    for (int c = threadIdx.x; c < D; c += NTHREADS) dst[c] = src ? src[c] : fill;
}

template <int CG>
__device__ __forceinline__ float group_mean4(float4 v) {   // mean over the lane's group (CG >= 4)
    float s = (v.x + v.y) + (v.z + v.w);
    if (CG >= 8) s += __shfl_xor(s, 16, 64);
    if (CG >= 16) s += __shfl_xor(s, 32, 64);
    return s * (1.0f / CG);
}


// Loads the lane's NJ float4 of one 16-row tile, combining the stage terms.  The switch on the
// (wave-uniform) term count sits OUTSIDE the j loop so that all NT*NJ loads are in flight together.
template <int NJ, int NT, int TC = 2>
__device__ __forceinline__ void load_tile_n(const LinComb& lc, int64_t base, float4 (&xv)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TC) {     // TC terms (TC*NJ loads) in flight at a time bounds the registers
        float4 v[TC][NJ];
#pragma unroll
        for (int tt = 0; tt < TC; ++tt)
            if (t0 + tt < NT) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) v[tt][j] = ld4(lc.ptr[t0 + tt] + base + 16 * j);
            }
#pragma unroll
        for (int tt = 0; tt < TC; ++tt)
            if (t0 + tt < NT) {
                const float c = lc.coef[t0 + tt];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    xv[j].x = fmaf(c, v[tt][j].x, xv[j].x); xv[j].y = fmaf(c, v[tt][j].y, xv[j].y);
                    xv[j].z = fmaf(c, v[tt][j].z, xv[j].z); xv[j].w = fmaf(c, v[tt][j].w, xv[j].w);
                }
            }
    }
}

template <int NJ, int TC = 2>
__device__ __forceinline__ void load_tile(const LinComb& lc, int64_t base, bool valid, float4 (&xv)[NJ]) {
    if (!valid) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    switch (lc.n) {
        case 1: load_tile_n<NJ, 1, TC>(lc, base, xv); break;
        case 2: load_tile_n<NJ, 2, TC>(lc, base, xv); break;
        case 3: load_tile_n<NJ, 3, TC>(lc, base, xv); break;
        case 4: load_tile_n<NJ, 4, TC>(lc, base, xv); break;
        case 5: load_tile_n<NJ, 5, TC>(lc, base, xv); break;
        case 6: load_tile_n<NJ, 6, TC>(lc, base, xv); break;
        case 7: load_tile_n<NJ, 7, TC>(lc, base, xv); break;
        default: load_tile_n<NJ, 8, TC>(lc, base, xv); break;
    }
}

// One 16-row panel: acc[tt] += sum_k Wlds[k][16tt + r] * xb[k]   (k-slot g of step (j,c) carries k = 16j+4g+c).
// The A operands of step s+1 are read from LDS before the MFMAs of step s are issued, and a
// scheduling barrier per step keeps hipcc from hoisting all 4*NJ*NJ LDS reads to the top (spills).
template <int NJ>
__device__ __forceinline__ void mfma_panel(const float* wl /* Wlds + 4g*LDW + r */, const float4 (&xv)[NJ], f32x4 (&acc)[NJ]) {
    constexpr int LDW = 16 * NJ + 4;
    // ping-pong operand registers: step s multiplies out of a[s&1] while a[(s+1)&1] is being read from LDS
    // (indices are compile-time after unrolling, so no register moves are generated)
    float a[2][NJ];
#pragma unroll
    for (int tt = 0; tt < NJ; ++tt) a[0][tt] = wl[16 * tt];
#pragma unroll
    for (int s = 0; s < 4 * NJ; ++s) {
        const int j = s / 4, c = s % 4;
        const float xb = c == 0 ? xv[j].x : (c == 1 ? xv[j].y : (c == 2 ? xv[j].z : xv[j].w));
        if (s + 1 < 4 * NJ) {
            const int jn = (s + 1) / 4, cn = (s + 1) % 4;
#pragma unroll
            for (int tt = 0; tt < NJ; ++tt) a[(s + 1) & 1][tt] = wl[(16 * jn + cn) * LDW + 16 * tt];
        }
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][tt], xb, acc[tt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Steps [S0, S1) of mfma_panel (same operand order, hence the same sums): lets the caller put global-memory
// work between two parts of a panel.
template <int NJ, int S0, int S1>
__device__ __forceinline__ void mfma_panel_part(const float* wl, const float4 (&xv)[NJ], f32x4 (&acc)[NJ]) {
    constexpr int LDW = 16 * NJ + 4;
    float a[2][NJ];
#pragma unroll
    for (int tt = 0; tt < NJ; ++tt) a[S0 & 1][tt] = wl[(16 * (S0 / 4) + (S0 % 4)) * LDW + 16 * tt];
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        const int j = s / 4, c = s % 4;
        const float xb = c == 0 ? xv[j].x : (c == 1 ? xv[j].y : (c == 2 ? xv[j].z : xv[j].w));
        if (s + 1 < S1) {
            const int jn = (s + 1) / 4, cn = (s + 1) % 4;
#pragma unroll
            for (int tt = 0; tt < NJ; ++tt) a[(s + 1) & 1][tt] = wl[(16 * jn + cn) * LDW + 16 * tt];
        }
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][tt], xb, acc[tt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Raw loads of terms [T0, T0+2) of one tile (issued, not yet used) and their accumulation, in the term order of
// load_tile_n so that prefetched and direct loads give the same bits.
template <int NJ>
__device__ __forceinline__ void issue_terms2(const LinComb& lc, int t0, int64_t base, bool valid, float4 (&v)[2][NJ]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
        if (valid && t0 + tt < lc.n) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[tt][j] = ld4(lc.ptr[t0 + tt] + base + 16 * j);
        }
}
template <int NJ>
__device__ __forceinline__ void fold_terms2(const LinComb& lc, int t0, bool valid, const float4 (&v)[2][NJ], float4 (&xv)[NJ]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
        if (valid && t0 + tt < lc.n) {
            const float c = lc.coef[t0 + tt];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                xv[j].x = fmaf(c, v[tt][j].x, xv[j].x); xv[j].y = fmaf(c, v[tt][j].y, xv[j].y);
                xv[j].z = fmaf(c, v[tt][j].z, xv[j].z); xv[j].w = fmaf(c, v[tt][j].w, xv[j].w);
            }
        }
}

// ---------------------------------------------------------------------------------
// forward:  S[row, :] = t*W[0,:] + GN(x[row,:]) * W[1:, :]
// ---------------------------------------------------------------------------------
// Lane layouts of a 16-row tile (wave64):
//   memory layout  ("M"): lane m -> row m>>2, 16-byte chunk m&3 of each 64-byte column block j.  Adjacent
//                         lanes touch adjacent 16 B, so one wave instruction is 16 runs of 64 contiguous
//                         bytes (loads AND stores are issued in this layout);
//   MFMA layout    ("F"): lane f -> row f&15, k-slot / column quad f>>4 (fixed by v_mfma_f32_16x16x4_f32).
// A direct global access in layout F puts adjacent lanes 512 B apart: 64 separate 16-B requests per
// instruction (measured: 1 TB/s).  The two layouts hold the same (row, chunk) set, so one ds_bpermute per
// register converts between them: F-lane (g, r) takes from M-lane 4r+g; M-lane m takes from F-lane
// (m&3)*16 + (m>>2).  Term combination and GroupNorm act per float4 and are done in layout M.
__device__ __forceinline__ float4 to_mfma_layout(const float4 v, int src_m_lane_x4) {
    return make_float4(__int_as_float(__builtin_amdgcn_ds_bpermute(src_m_lane_x4, __float_as_int(v.x))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_m_lane_x4, __float_as_int(v.y))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_m_lane_x4, __float_as_int(v.z))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_m_lane_x4, __float_as_int(v.w))));
}
__device__ __forceinline__ float4 acc_to_mem_layout(const f32x4 a, int src_f_lane_x4) {
    return make_float4(__int_as_float(__builtin_amdgcn_ds_bpermute(src_f_lane_x4, __float_as_int(a[0]))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_f_lane_x4, __float_as_int(a[1]))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_f_lane_x4, __float_as_int(a[2]))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_f_lane_x4, __float_as_int(a[3]))));
}

template <int NJ, int CG>   // d = 16*NJ
__global__ __launch_bounds__(256, 2) void gn_gemm_fwd_kernel(LinComb xin, int n_rows, float eps,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ W, int has_time, float t,
                                                             float* __restrict__ S, float* __restrict__ xout,
                                                             const float* __restrict__ W2, float* __restrict__ S2)
{
    constexpr int D = 16 * NJ;
    constexpr int LDW = D + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (blockIdx.y == 1) { W = W2; S = S2; xout = nullptr; }      // second (W, S) pair over the same input (grid.y = 2)
    float* Ws = smem;
    float* Gs = smem + D * LDW;          // gamma | beta | t * W[0,:]
    float* Bs = Gs + D;
    float* T0 = Bs + D;
    fill_w_lds<D, 256, false>(Ws, W, has_time);
    fill_vec_lds<D, 256>(Gs, gamma, 1.f);
    fill_vec_lds<D, 256>(Bs, beta, 0.f);
    for (int c = threadIdx.x; c < D; c += 256) T0[c] = has_time ? t * W[c] : 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int r = l & 15, g = l >> 4;          // MFMA layout
    const int mr = l >> 2, mg = l & 3;         // memory layout
    const int to_f = (4 * r + g) * 4, to_m = (mg * 16 + mr) * 4;
    const int n_tiles = (n_rows + 15) / 16;
    const int stride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    // The combined input rows of the NEXT tile are fetched while the matrix pipe works on the current one: terms 0-1
    // are issued before the first half of the panel, terms 2-3 before the second half (more than 4 terms - adaptive
    // solvers only - are completed after the panel).  With two waves per SIMD (LDS-bound occupancy) the loads of a
    // 3- or 4-term stage input are otherwise exposed.
    float4 nx[NJ];
    load_tile<NJ, 2>(xin, (int64_t)(tile * 16 + mr) * D + 4 * mg, tile * 16 + mr < n_rows, nx);
    for (; tile < n_tiles; tile += stride) {
        const int row = tile * 16 + mr;
        const bool valid = row < n_rows;
        const int nrow = row + stride * 16;
        const bool nvalid = (tile + stride < n_tiles) && nrow < n_rows;
        const int64_t nbase = (int64_t)nrow * D + 4 * mg;
        float4 xv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (xout != nullptr && valid) *reinterpret_cast<float4*>(xout + (int64_t)row * D + 16 * j + 4 * mg) = nx[j];
            xv[j] = gn_forward_v<CG>(nx[j], eps, ld4(Gs + 16 * j + 4 * mg), ld4(Bs + 16 * j + 4 * mg));
            xv[j] = to_mfma_layout(xv[j], to_f);
            nx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x4 acc[NJ];
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            const float4 w0 = ld4(T0 + 16 * tt + 4 * g);
            acc[tt] = (f32x4){w0.x, w0.y, w0.z, w0.w};
        }
        float4 v[2][NJ];
        issue_terms2<NJ>(xin, 0, nbase, nvalid, v);
        mfma_panel_part<NJ, 0, 2 * NJ>(Ws + 4 * g * LDW + r, xv, acc);
        fold_terms2<NJ>(xin, 0, nvalid, v, nx);
        issue_terms2<NJ>(xin, 2, nbase, nvalid, v);
        mfma_panel_part<NJ, 2 * NJ, 4 * NJ>(Ws + 4 * g * LDW + r, xv, acc);
        fold_terms2<NJ>(xin, 2, nvalid, v, nx);
        if (xin.n > 4) { issue_terms2<NJ>(xin, 4, nbase, nvalid, v); fold_terms2<NJ>(xin, 4, nvalid, v, nx); }
        if (xin.n > 6) { issue_terms2<NJ>(xin, 6, nbase, nvalid, v); fold_terms2<NJ>(xin, 6, nvalid, v, nx); }
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            const float4 o = acc_to_mem_layout(acc[tt], to_m);
            if (valid) *reinterpret_cast<float4*>(S + (int64_t)row * D + 16 * tt + 4 * mg) = o;
        }
    }
}

// ---------------------------------------------------------------------------------
// Split-bf16 variant of the forward product (d = 128).
// v_mfma_f32_16x16x4_f32 runs on the SIMD's fp32 vector ALUs (64 FLOP/clk/SIMD, measured: an fp32-MFMA wave and
// a VALU wave on one SIMD take the SUM of their times, tools/dev/mfma_valu.hip), so the exact-fp32 kernel above
// pays MFMA + VALU serially.  Here every fp32 operand is decomposed EXACTLY into three bf16 pieces
// (x = x_hi + x_mid + x_lo, 8+8+8 significant bits, each piece the round-to-nearest bf16 of the running residual)
// and the product is accumulated in fp32, smallest first, from the eight piece products whose weight is >= 2^-24 of
// the result (lo*mid, mid*lo, lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; only lo*lo, < 2^-32 of the product, is
// left out: every product enters the accumulator more exactly than an fp32 FMA forms it) on
// v_mfma_f32_16x16x32_bf16 - 16 cycles per instruction at 8x the K depth: 256 MFMAs x 16 cycles per 16-row tile
// instead of 256 x 32, and - unlike v_mfma_f32_16x16x4_f32 - the other waves of the SIMD keep issuing their memory
// instructions meanwhile (profiles/r02_mfma_mem.txt).  Parity is the same 1e-5 bar (tests/test_gpu_kernels.py runs both variants).
// ---------------------------------------------------------------------------------

__device__ __forceinline__ unsigned short bf16_bits(float x) {
    const __bf16 h = (__bf16)x;                               // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_val(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_bits(x);
    const float r1 = x - bf16_val(h);
    m = bf16_bits(r1);
    l = bf16_bits(r1 - bf16_val(m));
}

template <int CG, int NX>   // d = 128; NX = 1, 2: that many terms, the NEXT tile's raw loads in flight during the matrix phase; 0: any count
__global__ __launch_bounds__(512, 2) void gn_gemm_fwd_split_kernel(LinComb xin, int n_rows, float eps,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ W, int has_time, float t,
                                                                   float* __restrict__ S)
{
    constexpr int D = 128, NJ = 8, NK = 4;          // NK k-blocks of 32
    constexpr int LDK = D + 8;                      // bf16 elements per Wt row (272 B: conflict-free ds_read_b128)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* Wp = reinterpret_cast<unsigned short*>(smem);          // [3][D][LDK]: Wt pieces, Wt[n][k] = W1[k][n]
    float* Gs = smem + (3 * D * LDK) / 2;
    float* Bs = Gs + D;
    float* T0 = Bs + D;
    for (int idx = threadIdx.x; idx < D * D / 4; idx += 512) {
        const int k = idx / (D / 4), n = (idx % (D / 4)) * 4;
        const float4 w = ld4(W + (int64_t)(k + has_time) * D + n);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short h, m, l;
            split3(wv[q], h, m, l);
            Wp[(0 * D + n + q) * LDK + k] = h; Wp[(1 * D + n + q) * LDK + k] = m; Wp[(2 * D + n + q) * LDK + k] = l;
        }
    }
    fill_vec_lds<D, 512>(Gs, gamma, 1.f);
    fill_vec_lds<D, 512>(Bs, beta, 0.f);
    for (int c = threadIdx.x; c < D; c += 512) T0[c] = has_time ? t * W[c] : 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int r = l & 15, g = l >> 4;          // MFMA layout
    const int mr = l >> 2, mg = l & 3;         // memory layout
    const int to_m = (mg * 16 + mr) * 4;
    const int n_tiles = (n_rows + 15) / 16;
    constexpr int NXR = NX > 0 ? NX : 1;
    float4 raw[NXR][NJ];                       // NX > 0: the raw terms of the tile about to be processed
    auto request = [&](int tile) {             // unconditional loads from a clamped row (no wait at a branch join)
        const int rr = tile * 16 + mr;
        const int64_t base = (int64_t)(rr < n_rows ? rr : n_rows - 1) * D + 4 * mg;
#pragma unroll
        for (int t = 0; t < NXR; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) raw[t][j] = ld4(xin.ptr[t] + base + 16 * j);
    };
    if (NX > 0 && blockIdx.x * 8 + wave < n_tiles) request(blockIdx.x * 8 + wave);
    for (int tile = blockIdx.x * 8 + wave; tile < n_tiles; tile += gridDim.x * 8) {
        const int row = tile * 16 + mr;
        const bool valid = row < n_rows;
        float4 xv[NJ];
        if (NX > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {     // the term order and arithmetic of load_tile_n
                xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < NXR; ++t) {
                    const float c = xin.coef[t];
                    xv[j].x = fmaf(c, raw[t][j].x, xv[j].x); xv[j].y = fmaf(c, raw[t][j].y, xv[j].y);
                    xv[j].z = fmaf(c, raw[t][j].z, xv[j].z); xv[j].w = fmaf(c, raw[t][j].w, xv[j].w);
                }
                if (!valid) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const int nt_ = tile + gridDim.x * 8;
            request(nt_ < n_tiles ? nt_ : tile);
            __builtin_amdgcn_sched_barrier(0);         // issued HERE, ahead of the cut and the matrix phase, not sunk to the loop end
        } else {
            load_tile<NJ, 2>(xin, (int64_t)row * D + 4 * mg, valid, xv);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            xv[j] = gn_forward_v<CG>(xv[j], eps, ld4(Gs + 16 * j + 4 * mg), ld4(Bs + 16 * j + 4 * mg));
        // memory layout -> bf16 MFMA layout: F-lane (r, g) needs x[row r][32kb + 8g + 4h + c]; that float4 sits in
        // M-lane 4r + 2(g&1) + h, register 2kb + (g>>1)
        bf16x8 xp[NK][3];
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) {
            float xf[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int src = (4 * r + 2 * (g & 1) + h) * 4;
                const float4 a = to_mfma_layout(xv[2 * kb], src), b = to_mfma_layout(xv[2 * kb + 1], src);
                const float4 v = (g >> 1) ? b : a;
                xf[4 * h] = v.x; xf[4 * h + 1] = v.y; xf[4 * h + 2] = v.z; xf[4 * h + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned short h, m, lo;
                split3(xf[e], h, m, lo);
                xp[kb][0][e] = (short)h; xp[kb][1][e] = (short)m; xp[kb][2][e] = (short)lo;
            }
        }
        f32x4 acc[NJ];
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            const float4 w0 = ld4(T0 + 16 * tt + 4 * g);
            acc[tt] = (f32x4){w0.x, w0.y, w0.z, w0.w};
        }
        // ping-pong A pieces: group s = kb*NJ + tt multiplies out of a[s&1] while a[(s+1)&1] is read from LDS;
        // the per-group scheduling barrier keeps hipcc from hoisting all 96 16-byte LDS reads (spills)
        bf16x8 a[2][3];
        {
            const unsigned short* wr = Wp + r * LDK + 8 * g;
            a[0][0] = *reinterpret_cast<const bf16x8*>(wr);
            a[0][1] = *reinterpret_cast<const bf16x8*>(wr + D * LDK);
            a[0][2] = *reinterpret_cast<const bf16x8*>(wr + 2 * D * LDK);
        }
#pragma unroll
        for (int sidx = 0; sidx < NK * NJ; ++sidx) {
            const int kb = sidx / NJ, tt = sidx % NJ;
            if (sidx + 1 < NK * NJ) {
                const int kb2 = (sidx + 1) / NJ, tt2 = (sidx + 1) % NJ;
                const unsigned short* wr = Wp + (16 * tt2 + r) * LDK + 32 * kb2 + 8 * g;
                a[(sidx + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(wr);
                a[(sidx + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(wr + D * LDK);
                a[(sidx + 1) & 1][2] = *reinterpret_cast<const bf16x8*>(wr + 2 * D * LDK);
            }
            const bf16x8 ah = a[sidx & 1][0], am = a[sidx & 1][1], al = a[sidx & 1][2];
            f32x4 c = acc[tt];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xp[kb][1], c, 0, 0, 0);      // lo*mid, mid*lo: 2^-24 of the product each
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xp[kb][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xp[kb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xp[kb][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xp[kb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, xp[kb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xp[kb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xp[kb][0], c, 0, 0, 0);
            acc[tt] = c;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            const float4 o = acc_to_mem_layout(acc[tt], to_m);
            if (valid) *reinterpret_cast<float4*>(S + (int64_t)row * D + 16 * tt + 4 * mg) = o;
        }
    }
}

// ---------------------------------------------------------------------------------
// VJP w.r.t. x:  dxn = dS * W1^T ; dx = GN'(x)^T dxn ; out = out_scale*dx
// one partial sum of dgamma / dbeta per block.
// ---------------------------------------------------------------------------------
template <int NJ, int CG>
__global__ __launch_bounds__(256, 2) void gn_gemm_bwd_kernel(LinComb xin, int n_rows, float eps,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ W, int has_time,
                                                             const float* __restrict__ dS, float out_scale,
                                                             LinComb pre, float* __restrict__ dx,
                                                             float* __restrict__ dgamma_part,
                                                             float* __restrict__ dbeta_part, int n_part)
{
    constexpr int D = 16 * NJ;
    constexpr int LDW = D + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wt = smem;   // Wt[n][i] = W1[i][n]
    float* Gs = smem + D * LDW;
    fill_w_lds<D, 256, true>(Wt, W, has_time);
    fill_vec_lds<D, 256>(Gs, gamma, 1.f);
    __syncthreads();
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int r = l & 15, g = l >> 4;          // MFMA layout
    const int mr = l >> 2, mg = l & 3;         // memory layout
    const int to_f = (4 * r + g) * 4, to_m = (mg * 16 + mr) * 4;
    const int n_tiles = (n_rows + 15) / 16;
    float4 dgs[NJ], dbs[NJ];                   // memory layout: channels 16tt + 4mg .. +3 of row mr
#pragma unroll
    for (int j = 0; j < NJ; ++j) { dgs[j] = make_float4(0.f, 0.f, 0.f, 0.f); dbs[j] = make_float4(0.f, 0.f, 0.f, 0.f); }

    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int row = tile * 16 + mr;
        const bool valid = row < n_rows;
        float4 gv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            gv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) gv[j] = ld4(dS + (int64_t)row * D + 16 * j + 4 * mg);
        }
        float4 xt[NJ];      // x tile (memory layout), fetched before the MFMA phase so that its latency hides under it
        if (CG != 0) load_tile<NJ>(xin, (int64_t)row * D + 4 * mg, valid, xt);
#pragma unroll
        for (int j = 0; j < NJ; ++j) gv[j] = to_mfma_layout(gv[j], to_f);
        f32x4 acc[NJ];
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_panel<NJ>(Wt + 4 * g * LDW + r, gv, acc);
        // acc[tt] (MFMA layout) = dxn[row r][16tt+4g .. +3]  ->  memory layout for GroupNorm backward + store
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            const int c0 = 16 * tt + 4 * mg;
            const float4 dy = acc_to_mem_layout(acc[tt], to_m);
            float4 out = dy;
            if (CG != 0) {
                const float4 x = xt[tt];
                float4 mean, rstd;
                gn_stats<CG>(x, eps, mean, rstd);
                const float4 xh = make_float4((x.x - mean.x) * rstd.x, (x.y - mean.y) * rstd.y,
                                              (x.z - mean.z) * rstd.z, (x.w - mean.w) * rstd.w);
                const float4 gm = ld4(Gs + c0);
                const float4 dh = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
                if (valid) {
                    dgs[tt].x += dy.x * xh.x; dgs[tt].y += dy.y * xh.y; dgs[tt].z += dy.z * xh.z; dgs[tt].w += dy.w * xh.w;
                    dbs[tt].x += dy.x; dbs[tt].y += dy.y; dbs[tt].z += dy.z; dbs[tt].w += dy.w;
                }
                if (CG == 4) {
                    // four channels per group (the group is this lane's float4): the normalised form
                    // dx = rstd * (dh - mean(dh) - xh * mean(dh * xh)) - a third fewer vector instructions than
                    // ATen's algebraic form below, which is kept for the ill-conditioned 1- and 2-channel groups
                    //  where its rounding is the reference's.  The fp32 matrix instruction shares
                    // the vector pipe on gfx950, so these instructions are paid in full.
                    const float m1 = ((dh.x + dh.y) + (dh.z + dh.w)) * 0.25f;
                    const float m2 = ((dh.x * xh.x + dh.y * xh.y) + (dh.z * xh.z + dh.w * xh.w)) * 0.25f;
                    const float rs = rstd.x;
                    out = make_float4(rs * (dh.x - m1 - xh.x * m2), rs * (dh.y - m1 - xh.y * m2),
                                      rs * (dh.z - m1 - xh.z * m2), rs * (dh.w - m1 - xh.w * m2));
                } else {
                // ATen's CPU form (group_norm_kernel.cpp, GroupNormBackward): with ds = sum dy*gamma*x and
                // db = sum dy*gamma over the group,  c2 = (db*mean - ds)*rstd^3/CG,  c3 = -c2*mean - db*rstd/CG,
                // dx = rstd*gamma*dy + c2*x + c3.  Same algebra as rstd*(dh - mean(dh) - xh*mean(dh*xh));
                // kept in this form so that the rounding behaves like the reference's on ill-conditioned
                // groups (1 or 2 channels per group).
                const float4 px = make_float4(dh.x * x.x, dh.y * x.y, dh.z * x.z, dh.w * x.w);
                float4 ds, db;
                if (CG == 1) {
                    ds = px; db = dh;
                } else if (CG == 2) {
                    ds = make_float4(px.x + px.y, px.x + px.y, px.z + px.w, px.z + px.w);
                    db = make_float4(dh.x + dh.y, dh.x + dh.y, dh.z + dh.w, dh.z + dh.w);
                } else {
                    const float a = (px.x + px.y) + (px.z + px.w);
                    const float b = (dh.x + dh.y) + (dh.z + dh.w);
                    ds = make_float4(a, a, a, a); db = make_float4(b, b, b, b);
                }
                constexpr float sc = 1.0f / (CG > 0 ? CG : 1);
                const float4 r3 = make_float4(rstd.x * rstd.x * rstd.x * sc, rstd.y * rstd.y * rstd.y * sc,
                                              rstd.z * rstd.z * rstd.z * sc, rstd.w * rstd.w * rstd.w * sc);
                const float4 c2 = make_float4((db.x * mean.x - ds.x) * r3.x, (db.y * mean.y - ds.y) * r3.y,
                                              (db.z * mean.z - ds.z) * r3.z, (db.w * mean.w - ds.w) * r3.w);
                const float4 c3 = make_float4(-c2.x * mean.x - db.x * rstd.x * sc, -c2.y * mean.y - db.y * rstd.y * sc,
                                              -c2.z * mean.z - db.z * rstd.z * sc, -c2.w * mean.w - db.w * rstd.w * sc);
                out = make_float4(rstd.x * gm.x * dy.x + c2.x * x.x + c3.x, rstd.y * gm.y * dy.y + c2.y * x.y + c3.y,
                                  rstd.z * gm.z * dy.z + c2.z * x.z + c3.z, rstd.w * gm.w * dy.w + c2.w * x.w + c3.w);
                }
            }
            if (valid) {
                float4 o = make_float4(out_scale * out.x, out_scale * out.y, out_scale * out.z, out_scale * out.w);
                if (pre.n > 0) {     // fused RK solution combine of the adjoint component
                    const float4 pv = lc_load4(pre, (int64_t)row * D + c0);
                    o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
                }
                *reinterpret_cast<float4*>(dx + (int64_t)row * D + c0) = o;
            }
        }
    }
    if (CG != 0 && dgamma_part) {
        __syncthreads();                       // every wave is done with Wt: reuse LDS for the reduction
        float* red = smem;                     // [2][4 waves][D]  (Gs lies behind Wt and is not touched)
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            float4 a = dgs[tt], b = dbs[tt];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {  // lanes with equal mg (l & 3) hold the same channels
                a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
                b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); b.z += __shfl_xor(b.z, o, 64); b.w += __shfl_xor(b.w, o, 64);
            }
            if (mr == 0) {
                *reinterpret_cast<float4*>(red + wave * D + 16 * tt + 4 * mg) = a;
                *reinterpret_cast<float4*>(red + (4 + wave) * D + 16 * tt + 4 * mg) = b;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            dgamma_part[(int64_t)blockIdx.x * D + c] = (red[c] + red[D + c]) + (red[2 * D + c] + red[3 * D + c]);
            dbeta_part[(int64_t)blockIdx.x * D + c] = (red[4 * D + c] + red[5 * D + c]) + (red[6 * D + c] + red[7 * D + c]);
        }
        // the caller's buffers hold gode_gemm_bwd_parts() rows: zero the ones no block owns
        for (int p = gridDim.x + blockIdx.x; p < n_part; p += gridDim.x)
            for (int c = threadIdx.x; c < D; c += 256) { dgamma_part[(int64_t)p * D + c] = 0.f; dbeta_part[(int64_t)p * D + c] = 0.f; }
    }
}

// ---------------------------------------------------------------------------------
// weight gradient: dW[has_time + i][n] = sum_rows xn[row][i] * dS[row][n]; row 0 (has_time) = sum_rows dS[row][n]
// (the gradient w.r.t. a unit time column: the caller scales it by t and uses it for dL/dt)
// one block partial per block.
// ---------------------------------------------------------------------------------
template <int NJ, int CG>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(LinComb xin, int n_rows, float eps,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ dS, int has_time,
                                                       float* __restrict__ dW_part)
{
    constexpr int D = 16 * NJ;
    constexpr int LD = D + 16;
    constexpr int R = 32;                       // rows per staged tile
    constexpr int TPR = D / 4;                  // threads per row (float4 each)
    constexpr int RPP = 256 / TPR;              // rows per pass
    constexpr int ITW = (NJ >= 4) ? NJ / 4 : 1; // i-tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;            // [R][LD]
    float* Gs = smem + R * LD;   // [R][LD]
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const int trow = threadIdx.x / TPR, tcol = (threadIdx.x % TPR) * 4;
    const bool wave_active = wave * ITW < NJ;

    f32x4 acc[ITW][NJ];
#pragma unroll
    for (int a = 0; a < ITW; ++a)
#pragma unroll
        for (int b = 0; b < NJ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

    const float4 gmv = (gamma && tcol < D) ? ld4(gamma + tcol) : make_float4(1.f, 1.f, 1.f, 1.f);   // this thread's columns
    const float4 btv = (beta && tcol < D) ? ld4(beta + tcol) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_tiles = (n_rows + R - 1) / R;
    constexpr int NP = (R + RPP - 1) / RPP;    // staging passes per tile (4 at d = 128)
    float4 xr[NP], gr[NP];
    // register prefetch of the next tile: its global loads are in flight while the MFMAs of the
    // current tile run out of LDS.
    auto prefetch = [&](int tile) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int rr = trow + p * RPP;
            const int row = tile * R + rr;
            xr[p] = make_float4(0.f, 0.f, 0.f, 0.f); gr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rr < R && row < n_rows) gr[p] = ld4(dS + (int64_t)row * D + tcol);
        }
        switch (xin.n) {
#define GODE_PF(NTV) case NTV: _Pragma("unroll") for (int p = 0; p < NP; ++p) { \
                const int rr = trow + p * RPP; const int row = tile * R + rr; \
                if (rr < R && row < n_rows) xr[p] = lc_load4_n<NTV>(xin, (int64_t)row * D + tcol); } break;
            GODE_PF(1) GODE_PF(2) GODE_PF(3) GODE_PF(4) GODE_PF(5) GODE_PF(6) GODE_PF(7)
            default: _Pragma("unroll") for (int p = 0; p < NP; ++p) {
                const int rr = trow + p * RPP; const int row = tile * R + rr;
                if (rr < R && row < n_rows) xr[p] = lc_load4_n<8>(xin, (int64_t)row * D + tcol); } break;
#undef GODE_PF
        }
    };
    int tile = blockIdx.x;
    if (tile < n_tiles) prefetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                       // previous tile's MFMAs are done reading LDS
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int rr = trow + p * RPP;
            if (rr < R) {
                const int row = tile * R + rr;
                float4 x = gn_forward_v<CG>(xr[p], eps, gmv, btv);            // CG in {0,1,2,4}: group inside the float4
                if (row >= n_rows) x = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(Xs + rr * LD + tcol) = x;
                *reinterpret_cast<float4*>(Gs + rr * LD + tcol) = gr[p];
                csum.x += gr[p].x; csum.y += gr[p].y; csum.z += gr[p].z; csum.w += gr[p].w;
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) prefetch(tile + gridDim.x);
        if (wave_active) {
#pragma unroll
            for (int kb = 0; kb < R; kb += 4) {
                float av[ITW], bv[NJ];
#pragma unroll
                for (int a = 0; a < ITW; ++a) av[a] = Xs[(kb + g) * LD + 16 * (wave * ITW + a) + r];
#pragma unroll
                for (int b = 0; b < NJ; ++b) bv[b] = Gs[(kb + g) * LD + 16 * b + r];
#pragma unroll
                for (int a = 0; a < ITW; ++a)
#pragma unroll
                    for (int b = 0; b < NJ; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
        }
    }
    float* out = dW_part + (int64_t)blockIdx.x * (D + has_time) * D;
    if (wave_active) {
#pragma unroll
        for (int a = 0; a < ITW; ++a)
#pragma unroll
            for (int b = 0; b < NJ; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * (wave * ITW + a) + 4 * g + q;
                    out[(int64_t)(i + has_time) * D + 16 * b + r] = acc[a][b][q];
                }
    }
    if (has_time) {
        __syncthreads();
        float* red = smem;   // [RPP][D]
        *reinterpret_cast<float4*>(red + trow * D + tcol) = csum;
        __syncthreads();
        if (threadIdx.x < D) {
            float s = 0.f;
            for (int p = 0; p < RPP; ++p) s += red[p * D + threadIdx.x];
            out[threadIdx.x] = s;
        }
    }
}

// ---------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores with fp32-exact operands (d = 128).
//
// Measured on gfx950 (tools/dev/mfma_mem2.hip, profiles/r02_mfma_mem.txt): v_mfma_f32_16x16x4_f32 does not overlap
// with ANYTHING another wave of the SIMD does - VALU work, global loads, stores, loads into LDS all take the SUM of the
// two times - whereas v_mfma_f32_16x16x32_bf16 overlaps with all of them.  The exact-fp32 kernel above therefore pays
// matrix time + memory time + GroupNorm time in series (0.43 ms alone, 0.96 ms next to the SpMM stream).
//
// Here every fp32 operand is cut EXACTLY into three bf16 pieces by truncation (x = hi + mid + lo: 8 + 8 + 8 significant
// bits, every piece representable, no rounding anywhere) and a product x*g is accumulated in fp32 from the piece
// products in increasing order of magnitude.  NT = 8 keeps every product down to 2^-24 of x*g (what is dropped,
// lo*lo, is below 2^-32 of it: closer to the exact product than an fp32 FMA chain gets); NT = 6 also drops mid*lo
// and lo*mid (<= 2^-23 of the product).
//
// LDS holds the six piece arrays TRANSPOSED, as 16-byte chunks = 8 consecutive rows of one column (the 8 k-values one
// lane feeds to a 16x16x32 MFMA): chunk(piece, g, c) at ((piece*4 + g)*144 + (c&3)*36 + (c>>2)) - the staging threads
// (4 rows x 4 columns each) write 8-byte halves at consecutive chunks, the operand reads of 16 lanes cover all 64 banks.
// ---------------------------------------------------------------------------------
// one column of a staging thread: its 4 consecutive rows -> the three piece arrays
__device__ __forceinline__ void stage_col4(char* chunk0 /* piece 0 */, int piece_bytes, float v0, float v1, float v2, float v3) {
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3_trunc(v0, h0, m0, l0); split3_trunc(v1, h1, m1, l1); split3_trunc(v2, h2, m2, l2); split3_trunc(v3, h3, m3, l3);
    *reinterpret_cast<uint2*>(chunk0) = pack_hi16x4(h0, h1, h2, h3);
    *reinterpret_cast<uint2*>(chunk0 + piece_bytes) = pack_hi16x4(m0, m1, m2, m3);
    *reinterpret_cast<uint2*>(chunk0 + 2 * piece_bytes) = pack_hi16x4(l0, l1, l2, l3);
}

// Wave-specialised, 16 waves per block (one block per CU): waves 0-7 CONSUME (operand reads + MFMAs, 4 x 2 output
// tiles each) while waves 8-11 and 12-15 PRODUCE alternate tiles (global loads, GroupNorm, the three-way cut, LDS
// writes) on two LDS buffers, one block barrier per tile.  A producer group requests its next tile as soon as it has
// staged one, so that request is in flight for two tile periods while the OTHER group stages: two tiles (64 KB) per CU
// are always on their way, held by different waves.  (Two tiles in flight in ONE wave's registers do not work: hipcc
// takes the wait count at the loop head as the minimum over all paths and ends up draining the tile just requested;
// the same happens behind a run-time switch on the term count, hence the template parameter NX.)  The block barrier
// is lds_barrier() (dense_common.h): LDS traffic only, outstanding global loads stay in flight across it.
template <int CG, int NT, int NX>   // NX = number of terms of x held raw in the prefetch registers (1..4); 0 = any count, combined at load
__global__ __launch_bounds__(1024, 1) void wgrad_split_kernel(LinComb xin, int n_rows, float eps,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ dS, int has_time,
                                                             float* __restrict__ dW_part)
{
    constexpr int D = 128, R = 32, NP = 4;
    constexpr int QS = 36, GRP = 4 * QS, PIECE = 4 * GRP;          // in 16-byte chunks
    constexpr int PIECE_B = PIECE * 16, BUF_B = 6 * PIECE_B;       // one buffer: X pieces, then dS pieces
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const bool producer = threadIdx.x >= 512;
    const int n_tiles = (n_rows + R - 1) / R;
    const int stride = gridDim.x;
    const int my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / stride + 1 : 0;   // tiles blockIdx.x + k*stride
    float* out = dW_part + (int64_t)blockIdx.x * (D + has_time) * D;

    if (producer) {
        const int grp = (threadIdx.x - 512) >> 8;                  // this group stages the block's tiles k = grp, grp+2, ...
        const int pt = (threadIdx.x - 512) & 255;
        const int trow = pt >> 5, tc4 = pt & 31, tcol = 4 * tc4;   // rows 4*trow + p, columns tcol..tcol+3
        const int st_off = ((trow >> 1) * GRP + tc4) * 16 + (trow & 1) * 8;
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gmv = gamma ? ld4(gamma + tcol) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 btv = beta ? ld4(beta + tcol) : make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int NXR = NX > 0 ? NX : 1;
        float4 xr[NXR][NP], gr[NP];
        // Unconditional loads from clamped rows (a load under a branch would be waited for at the join; rows past the
        // end and tiles past the block's last one are discarded when the tile is staged).
        auto prefetch = [&](int k) {
            const int tile = blockIdx.x + (k < my_tiles ? k : (my_tiles > 0 ? my_tiles - 1 : 0)) * stride;
            int64_t off[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int row = tile * R + 4 * trow + p;
                off[p] = (int64_t)(row < n_rows ? row : n_rows - 1) * D + tcol;
                gr[p] = ld4(dS + off[p]);
            }
            if (NX > 0) {
#pragma unroll
                for (int j = 0; j < NXR; ++j)
#pragma unroll
                    for (int p = 0; p < NP; ++p) xr[j][p] = ld4(xin.ptr[j] + off[p]);
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) xr[0][p] = lc_load4(xin, off[p]);
            }
        };
        auto stage = [&](int k) {                                  // tile k of this block -> buffer k & 1
            char* Xp = lds + (k & 1) * BUF_B;
            char* Gp = Xp + 3 * PIECE_B;
            const int tile = blockIdx.x + k * stride;
            float4 xn[NP], gn[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const bool valid = tile * R + 4 * trow + p < n_rows;
                float4 x = xr[0][p];
                if (NX > 0) {                                   // the term order and arithmetic of lc_load4_n
                    x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < NXR; ++j) {
                        const float c = xin.coef[j];
                        x.x = fmaf(c, xr[j][p].x, x.x); x.y = fmaf(c, xr[j][p].y, x.y);
                        x.z = fmaf(c, xr[j][p].z, x.z); x.w = fmaf(c, xr[j][p].w, x.w);
                    }
                }
                xn[p] = gn_forward_v<CG>(x, eps, gmv, btv);
                gn[p] = gr[p];
                if (!valid) { xn[p] = make_float4(0.f, 0.f, 0.f, 0.f); gn[p] = make_float4(0.f, 0.f, 0.f, 0.f); }
                csum.x += gn[p].x; csum.y += gn[p].y; csum.z += gn[p].z; csum.w += gn[p].w;
            }
            stage_col4(Xp + st_off + 0 * QS * 16, PIECE_B, xn[0].x, xn[1].x, xn[2].x, xn[3].x);
            stage_col4(Xp + st_off + 1 * QS * 16, PIECE_B, xn[0].y, xn[1].y, xn[2].y, xn[3].y);
            stage_col4(Xp + st_off + 2 * QS * 16, PIECE_B, xn[0].z, xn[1].z, xn[2].z, xn[3].z);
            stage_col4(Xp + st_off + 3 * QS * 16, PIECE_B, xn[0].w, xn[1].w, xn[2].w, xn[3].w);
            stage_col4(Gp + st_off + 0 * QS * 16, PIECE_B, gn[0].x, gn[1].x, gn[2].x, gn[3].x);
            stage_col4(Gp + st_off + 1 * QS * 16, PIECE_B, gn[0].y, gn[1].y, gn[2].y, gn[3].y);
            stage_col4(Gp + st_off + 2 * QS * 16, PIECE_B, gn[0].z, gn[1].z, gn[2].z, gn[3].z);
            stage_col4(Gp + st_off + 3 * QS * 16, PIECE_B, gn[0].w, gn[1].w, gn[2].w, gn[3].w);
        };
        prefetch(grp);
        if (grp == 0 && my_tiles > 0) { stage(0); prefetch(2); }
        lds_barrier();                                             // tile 0 staged
        for (int k = 0; k < my_tiles; ++k) {                       // while the consumers multiply tile k: stage tile k+1
            if (((k + 1) & 1) == grp && k + 1 < my_tiles) { stage(k + 1); prefetch(k + 3); }
            lds_barrier();
        }
        if (has_time) {
            float* red = smem;   // [16][D]; every consumer is past its last operand read (final barrier above)
            *reinterpret_cast<float4*>(red + (grp * 8 + trow) * D + tcol) = csum;
            __syncthreads();                                    // the consumers wait in the matching barrier below
            if (grp == 0 && pt < D) {
                float sacc = 0.f;
                for (int p = 0; p < 16; ++p) sacc += red[p * D + pt];
                out[pt] = sacc;
            }
        }
        return;
    }

    // ---- consumers: wave w owns output tiles a = 4(w>>2)..+3 (rows of dW) x b = 2(w&3), 2(w&3)+1 (columns)
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const int a0 = 4 * (wave >> 2), b0 = 2 * (wave & 3);
    const int rd_off = (g * GRP + (r & 3) * QS + (r >> 2)) * 16;   // lane (r, g), 16-column tile a -> + 64 a bytes
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    lds_barrier();                                     // tile 0 staged
    for (int k = 0; k < my_tiles; ++k) {
        const char* Xp = lds + (k & 1) * BUF_B;
        const char* Gp = Xp + 3 * PIECE_B;
        bf16x8 B[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) B[b][pc] = *reinterpret_cast<const bf16x8*>(Gp + pc * PIECE_B + rd_off + (b0 + b) * 64);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16x8 A[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) A[pc] = *reinterpret_cast<const bf16x8*>(Xp + pc * PIECE_B + rd_off + (a0 + a) * 64);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 c = acc[a][b];
                if (NT == 8) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[b][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[b][2], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[b][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[b][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[b][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[b][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[b][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[b][0], c, 0, 0, 0);
                acc[a][b] = c;
            }
        }
        lds_barrier();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (a0 + a) + 4 * g + q;
                out[(int64_t)(i + has_time) * D + 16 * (b0 + b) + r] = acc[a][b][q];
            }
    if (has_time) __syncthreads();                     // pairs with the producers' barrier before their column-sum reduction
}

// ---------------------------------------------------------------------------------
// Narrow products: d_out = NC in {1, 2, 4} columns (the two logit columns of the GAT ODE function), d_in = 16*NJ.
// One memory-layout tile of 16 rows per wave as above, but no matrix pipe: the NC weight columns live in registers,
// every lane multiplies its 4*NJ normalised values and the four lanes of a row are combined with two xor-shuffles.
// HBM-bound (reads x once); the generic kernels below take 1.6 - 2.1 ms for these shapes at 2^20 rows.
// ---------------------------------------------------------------------------------
template <int NJ, int NC>
__device__ __forceinline__ void narrow_load_w(float (&wk)[NJ][4][NC], const float* __restrict__ W, int has_time, int mg) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) wk[j][q][c] = W[(int64_t)(has_time + 16 * j + 4 * mg + q) * NC + c];
}

template <int NJ, int CG, int NC>
__global__ __launch_bounds__(256) void gn_narrow_fwd_kernel(LinComb xin, int n_rows, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ W, int has_time, float t,
                                                            float* __restrict__ S, float* __restrict__ xout)
{
    constexpr int D = 16 * NJ;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int mr = l >> 2, mg = l & 3;
    float wk[NJ][4][NC];
    narrow_load_w<NJ, NC>(wk, W, has_time, mg);
    float t0[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) t0[c] = has_time ? t * W[c] : 0.f;
    const int n_tiles = (n_rows + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int row = tile * 16 + mr;
        const bool valid = row < n_rows;
        float4 xv[NJ];
        load_tile<NJ, 2>(xin, (int64_t)row * D + 4 * mg, valid, xv);
        float p[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) p[c] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (xout != nullptr && valid) *reinterpret_cast<float4*>(xout + (int64_t)row * D + 16 * j + 4 * mg) = xv[j];
            const float4 gm = gamma ? ld4(gamma + 16 * j + 4 * mg) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 bt = beta ? ld4(beta + 16 * j + 4 * mg) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 xn = gn_forward_v<CG>(xv[j], eps, gm, bt);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                p[c] += (xn.x * wk[j][0][c] + xn.y * wk[j][1][c]) + (xn.z * wk[j][2][c] + xn.w * wk[j][3][c]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) { p[c] += __shfl_xor(p[c], 1, 64); p[c] += __shfl_xor(p[c], 2, 64); }
        if (valid && mg == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) S[(int64_t)row * NC + c] = t0[c] + p[c];
        }
    }
}

template <int NJ, int CG, int NC>
__global__ __launch_bounds__(256) void gn_narrow_bwd_kernel(LinComb xin, int n_rows, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ W,
                                                            int has_time, const float* __restrict__ dS, float out_scale,
                                                            LinComb pre, float* __restrict__ dx,
                                                            float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                            int n_part)
{
    constexpr int D = 16 * NJ;
    __shared__ float red[2 * 4 * D];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int mr = l >> 2, mg = l & 3;
    float wk[NJ][4][NC];
    narrow_load_w<NJ, NC>(wk, W, has_time, mg);
    float4 dgs[NJ], dbs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { dgs[j] = make_float4(0.f, 0.f, 0.f, 0.f); dbs[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
    const int n_tiles = (n_rows + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int row = tile * 16 + mr;
        const bool valid = row < n_rows;
        float ds[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) ds[c] = valid ? dS[(int64_t)row * NC + c] : 0.f;
        float4 xt[NJ];
        if (CG != 0) load_tile<NJ>(xin, (int64_t)row * D + 4 * mg, valid, xt);
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            const int c0 = 16 * tt + 4 * mg;
            float4 dy = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                dy.x = fmaf(ds[c], wk[tt][0][c], dy.x); dy.y = fmaf(ds[c], wk[tt][1][c], dy.y);
                dy.z = fmaf(ds[c], wk[tt][2][c], dy.z); dy.w = fmaf(ds[c], wk[tt][3][c], dy.w);
            }
            float4 out = dy;
            if (CG != 0) {          // GroupNorm backward, ATen's form - the same algebra as in gn_gemm_bwd_kernel
                const float4 x = xt[tt];
                float4 mean, rstd;
                gn_stats<CG>(x, eps, mean, rstd);
                const float4 xh = make_float4((x.x - mean.x) * rstd.x, (x.y - mean.y) * rstd.y,
                                              (x.z - mean.z) * rstd.z, (x.w - mean.w) * rstd.w);
                const float4 gm = gamma ? ld4(gamma + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 dh = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
                if (valid) {
                    dgs[tt].x += dy.x * xh.x; dgs[tt].y += dy.y * xh.y; dgs[tt].z += dy.z * xh.z; dgs[tt].w += dy.w * xh.w;
                    dbs[tt].x += dy.x; dbs[tt].y += dy.y; dbs[tt].z += dy.z; dbs[tt].w += dy.w;
                }
                const float4 px = make_float4(dh.x * x.x, dh.y * x.y, dh.z * x.z, dh.w * x.w);
                float4 dsg, dbg;
                if (CG == 1) {
                    dsg = px; dbg = dh;
                } else if (CG == 2) {
                    dsg = make_float4(px.x + px.y, px.x + px.y, px.z + px.w, px.z + px.w);
                    dbg = make_float4(dh.x + dh.y, dh.x + dh.y, dh.z + dh.w, dh.z + dh.w);
                } else {
                    const float a = (px.x + px.y) + (px.z + px.w);
                    const float b = (dh.x + dh.y) + (dh.z + dh.w);
                    dsg = make_float4(a, a, a, a); dbg = make_float4(b, b, b, b);
                }
                constexpr float sc = 1.0f / (CG > 0 ? CG : 1);
                const float4 r3 = make_float4(rstd.x * rstd.x * rstd.x * sc, rstd.y * rstd.y * rstd.y * sc,
                                              rstd.z * rstd.z * rstd.z * sc, rstd.w * rstd.w * rstd.w * sc);
                const float4 c2 = make_float4((dbg.x * mean.x - dsg.x) * r3.x, (dbg.y * mean.y - dsg.y) * r3.y,
                                              (dbg.z * mean.z - dsg.z) * r3.z, (dbg.w * mean.w - dsg.w) * r3.w);
                const float4 c3 = make_float4(-c2.x * mean.x - dbg.x * rstd.x * sc, -c2.y * mean.y - dbg.y * rstd.y * sc,
                                              -c2.z * mean.z - dbg.z * rstd.z * sc, -c2.w * mean.w - dbg.w * rstd.w * sc);
                out = make_float4(rstd.x * gm.x * dy.x + c2.x * x.x + c3.x, rstd.y * gm.y * dy.y + c2.y * x.y + c3.y,
                                  rstd.z * gm.z * dy.z + c2.z * x.z + c3.z, rstd.w * gm.w * dy.w + c2.w * x.w + c3.w);
            }
            if (valid) {
                float4 o = make_float4(out_scale * out.x, out_scale * out.y, out_scale * out.z, out_scale * out.w);
                if (pre.n > 0) {
                    const float4 pv = lc_load4(pre, (int64_t)row * D + c0);
                    o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
                }
                *reinterpret_cast<float4*>(dx + (int64_t)row * D + c0) = o;
            }
        }
    }
    if (CG != 0 && dgamma_part) {
#pragma unroll
        for (int tt = 0; tt < NJ; ++tt) {
            float4 a = dgs[tt], b = dbs[tt];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {
                a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
                b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); b.z += __shfl_xor(b.z, o, 64); b.w += __shfl_xor(b.w, o, 64);
            }
            if (mr == 0) {
                *reinterpret_cast<float4*>(red + wave * D + 16 * tt + 4 * mg) = a;
                *reinterpret_cast<float4*>(red + (4 + wave) * D + 16 * tt + 4 * mg) = b;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) {
            dgamma_part[(int64_t)blockIdx.x * D + c] = (red[c] + red[D + c]) + (red[2 * D + c] + red[3 * D + c]);
            dbeta_part[(int64_t)blockIdx.x * D + c] = (red[4 * D + c] + red[5 * D + c]) + (red[6 * D + c] + red[7 * D + c]);
        }
        for (int p = gridDim.x + blockIdx.x; p < n_part; p += gridDim.x)
            for (int c = threadIdx.x; c < D; c += 256) { dgamma_part[(int64_t)p * D + c] = 0.f; dbeta_part[(int64_t)p * D + c] = 0.f; }
    }
}

template <int NJ, int CG, int NC>
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(LinComb xin, int n_rows, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dS, int has_time,
                                                           float* __restrict__ dW_part)
{
    constexpr int D = 16 * NJ;
    __shared__ float red[4][(D + 1) * NC];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int mr = l >> 2, mg = l & 3;
    float acc[NJ][4][NC], acc0[NC];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[j][q][c] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc0[c] = 0.f;
    const int n_tiles = (n_rows + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int row = tile * 16 + mr;
        const bool valid = row < n_rows;
        float ds[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { ds[c] = valid ? dS[(int64_t)row * NC + c] : 0.f; if (mg == 0) acc0[c] += ds[c]; }
        float4 xv[NJ];
        load_tile<NJ, 2>(xin, (int64_t)row * D + 4 * mg, valid, xv);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float4 gm = gamma ? ld4(gamma + 16 * j + 4 * mg) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 bt = beta ? ld4(beta + 16 * j + 4 * mg) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 xn = gn_forward_v<CG>(xv[j], eps, gm, bt);
#pragma unroll
            for (int c = 0; c < NC; ++c) {      // ds is 0 on invalid rows
                acc[j][0][c] = fmaf(xn.x, ds[c], acc[j][0][c]); acc[j][1][c] = fmaf(xn.y, ds[c], acc[j][1][c]);
                acc[j][2][c] = fmaf(xn.z, ds[c], acc[j][2][c]); acc[j][3][c] = fmaf(xn.w, ds[c], acc[j][3][c]);
            }
        }
    }
    // lanes with equal mg hold the same channels: xor 4 .. 32; then the four waves through LDS
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[j][q][c] += __shfl_xor(acc[j][q][c], o, 64);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc0[c] += __shfl_xor(acc0[c], o, 64);
    }
    if (mr == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < NC; ++c) red[wave][(1 + 16 * j + 4 * mg + q) * NC + c] = acc[j][q][c];
        if (mg == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) red[wave][c] = acc0[c];
        }
    }
    __syncthreads();
    float* out = dW_part + (int64_t)blockIdx.x * (D + has_time) * NC;
    for (int idx = threadIdx.x; idx < (D + has_time) * NC; idx += 256) {
        const int src_idx = idx + (has_time ? 0 : NC);           // without a time row the colsum(dS) row is dropped
        out[idx] = (red[0][src_idx] + red[1][src_idx]) + (red[2][src_idx] + red[3][src_idx]);
    }
}


// ---------------------------------------------------------------------------------
// generic fallbacks (any d_in, d_out, groups): correct, not tuned.
// block = 256 threads, RB rows per block pass, rows staged (normalised) in LDS.
// ---------------------------------------------------------------------------------
constexpr int RB = 8;

__device__ __forceinline__ void stage_rows_gn(float* xs, float* stat, const LinComb& xin, int row0, int n_rows,
                                              int d_in, int groups, float eps, const float* gamma,
                                              const float* beta, bool apply)
{
    // xs[RB][d_in] raw x; stat[RB][groups][2] = mean, rstd
    for (int idx = threadIdx.x; idx < RB * d_in; idx += 256) {
        const int rr = idx / d_in, c = idx % d_in;
        const int row = row0 + rr;
        xs[idx] = row < n_rows ? lc_load1(xin, (int64_t)row * d_in + c) : 0.f;
    }
    __syncthreads();
    if (groups > 0) {
        const int cg = d_in / groups;
        for (int idx = threadIdx.x; idx < RB * groups; idx += 256) {
            const int rr = idx / groups, gi = idx % groups;
            const float* p = xs + rr * d_in + gi * cg;
            float m = 0.f;
            for (int c = 0; c < cg; ++c) m += p[c];
            m /= cg;
            float v = 0.f;
            for (int c = 0; c < cg; ++c) v += (p[c] - m) * (p[c] - m);
            v /= cg;
            stat[idx * 2] = m;
            stat[idx * 2 + 1] = 1.0f / sqrtf(v + eps);
        }
        __syncthreads();
        if (apply) {
            for (int idx = threadIdx.x; idx < RB * d_in; idx += 256) {
                const int rr = idx / d_in, c = idx % d_in;
                const int gi = c / cg;
                const float m = stat[(rr * groups + gi) * 2], rs = stat[(rr * groups + gi) * 2 + 1];
                xs[idx] = gn_apply1(xs[idx], m, rs, gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f);
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void gn_gemm_fwd_generic(LinComb xin, int n_rows, int d_in, int groups, float eps,
                                                           const float* gamma, const float* beta, const float* W,
                                                           int d_out, int has_time, float t, float* S)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* stat = smem + RB * d_in;
    const int n_blk = (n_rows + RB - 1) / RB;
    for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
        __syncthreads();
        stage_rows_gn(xs, stat, xin, blk * RB, n_rows, d_in, groups, eps, gamma, beta, true);
        for (int idx = threadIdx.x; idx < RB * d_out; idx += 256) {
            const int rr = idx / d_out, n = idx % d_out;
            const int row = blk * RB + rr;
            if (row >= n_rows) continue;
            if (!W) { S[(int64_t)row * d_out + n] = xs[rr * d_in + n]; continue; }      // stand-alone GroupNorm
            float acc = has_time ? t * W[n] : 0.f;
            for (int k = 0; k < d_in; ++k) acc = fmaf(xs[rr * d_in + k], W[(int64_t)(k + has_time) * d_out + n], acc);
            S[(int64_t)row * d_out + n] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void gn_gemm_bwd_generic(LinComb xin, int n_rows, int d_in, int groups, float eps,
                                                           const float* gamma, const float* W, int d_out,
                                                           int has_time, const float* dS, float out_scale, LinComb pre,
                                                           float* dx, float* dgamma_part, float* dbeta_part, int n_part)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                       // raw x  [RB][d_in]
    float* stat = xs + RB * d_in;           // [RB][groups][2]
    float* dy = stat + RB * (groups > 0 ? groups : 1) * 2;   // dxn [RB][d_in]
    float* red = dy + RB * d_in;            // [RB][groups][2] group means of dh, dh*xh
    const int n_blk = (n_rows + RB - 1) / RB;
    for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
        __syncthreads();
        stage_rows_gn(xs, stat, xin, blk * RB, n_rows, d_in, groups, eps, nullptr, nullptr, false);
        for (int idx = threadIdx.x; idx < RB * d_in; idx += 256) {
            const int rr = idx / d_in, i = idx % d_in;
            const int row = blk * RB + rr;
            float acc = 0.f;
            if (row < n_rows) {
                if (!W) acc = dS[(int64_t)row * d_out + i];                               // stand-alone GroupNorm: dy given
                else
                    for (int n = 0; n < d_out; ++n)
                        acc = fmaf(dS[(int64_t)row * d_out + n], W[(int64_t)(i + has_time) * d_out + n], acc);
            }
            dy[idx] = acc;
        }
        __syncthreads();
        if (groups > 0) {
            const int cg = d_in / groups;
            for (int idx = threadIdx.x; idx < RB * groups; idx += 256) {
                const int rr = idx / groups, gi = idx % groups;
                float dsum = 0.f, bsum = 0.f;      // ds, db of ATen's GroupNorm backward
                for (int c = 0; c < cg; ++c) {
                    const int cc = gi * cg + c;
                    const float dh = dy[rr * d_in + cc] * (gamma ? gamma[cc] : 1.f);
                    dsum += dh * xs[rr * d_in + cc]; bsum += dh;
                }
                red[idx * 2] = dsum; red[idx * 2 + 1] = bsum;
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < RB * d_in; idx += 256) {
                const int rr = idx / d_in, c = idx % d_in;
                const int row = blk * RB + rr;
                if (row >= n_rows) continue;
                const int gi = c / cg;
                const float m = stat[(rr * groups + gi) * 2], rs = stat[(rr * groups + gi) * 2 + 1];
                const float dsum = red[(rr * groups + gi) * 2], bsum = red[(rr * groups + gi) * 2 + 1];
                const float sc = 1.0f / cg;
                const float c2 = (bsum * m - dsum) * rs * rs * rs * sc;
                const float c3 = -c2 * m - bsum * rs * sc;
                dx[(int64_t)row * d_in + c] = out_scale * (rs * (gamma ? gamma[c] : 1.f) * dy[idx] + c2 * xs[idx] + c3) +
                                              (pre.n > 0 ? lc_load1(pre, (int64_t)row * d_in + c) : 0.f);
            }
            if (dgamma_part) {
                // per-block partial over the block's rows: written by the block's first pass of the grid-stride loop,
                // accumulated by its later ones (same thread, same address), so the buffer needs no zeroing beforehand
                const bool first = blk == (int)blockIdx.x;
                for (int c = threadIdx.x; c < d_in; c += 256) {
                    const int gi = c / cg;
                    float sg = 0.f, sb = 0.f;
                    for (int rr = 0; rr < RB; ++rr) {
                        if (blk * RB + rr >= n_rows) break;
                        const float m = stat[(rr * groups + gi) * 2], rs = stat[(rr * groups + gi) * 2 + 1];
                        sg += dy[rr * d_in + c] * (xs[rr * d_in + c] - m) * rs;
                        sb += dy[rr * d_in + c];
                    }
                    float* pg = dgamma_part + (int64_t)blockIdx.x * d_in + c;
                    float* pb = dbeta_part + (int64_t)blockIdx.x * d_in + c;
                    *pg = first ? sg : *pg + sg;
                    *pb = first ? sb : *pb + sb;
                }
            }
        } else {
            for (int idx = threadIdx.x; idx < RB * d_in; idx += 256) {
                const int rr = idx / d_in, c = idx % d_in;
                const int row = blk * RB + rr;
                if (row < n_rows) dx[(int64_t)row * d_in + c] = out_scale * dy[idx] + (pre.n > 0 ? lc_load1(pre, (int64_t)row * d_in + c) : 0.f);
            }
        }
    }
    if (dgamma_part && groups > 0) {
        // rows of the partial buffers that belong to no block (the caller sizes them for the largest grid of any path;
        // a block beyond the last row block never enters the loop above)
        const int owned = n_blk < (int)gridDim.x ? n_blk : (int)gridDim.x;
        for (int p = owned + blockIdx.x; p < n_part; p += gridDim.x)
            for (int c = threadIdx.x; c < d_in; c += 256) { dgamma_part[(int64_t)p * d_in + c] = 0.f; dbeta_part[(int64_t)p * d_in + c] = 0.f; }
    }
}

__global__ __launch_bounds__(256) void wgrad_generic(LinComb xin, int n_rows, int d_in, int groups, float eps,
                                                     const float* gamma, const float* beta, const float* dS,
                                                     int d_out, int has_time, float* dW_part)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* stat = smem + RB * d_in;
    const int n_blk = (n_rows + RB - 1) / RB;
    const int K = d_in + has_time;
    float* out = dW_part + (int64_t)blockIdx.x * K * d_out;
    for (int idx = threadIdx.x; idx < K * d_out; idx += 256) out[idx] = 0.f;
    for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
        __syncthreads();
        stage_rows_gn(xs, stat, xin, blk * RB, n_rows, d_in, groups, eps, gamma, beta, true);
        const int nr = (n_rows - blk * RB) < RB ? (n_rows - blk * RB) : RB;
        for (int idx = threadIdx.x; idx < K * d_out; idx += 256) {
            const int i = idx / d_out, n = idx % d_out;
            float acc = 0.f;
            for (int rr = 0; rr < nr; ++rr) {
                const float a = (has_time && i == 0) ? 1.0f : xs[rr * d_in + (i - has_time)];
                acc = fmaf(a, dS[(int64_t)(blk * RB + rr) * d_out + n], acc);
            }
            out[idx] += acc;
        }
    }
}

template <typename K>
int set_lds(K kernel, size_t bytes) { return gode_set_lds_once(reinterpret_cast<const void*>(kernel), bytes); }

int fast_cg(int64_t d_in, int64_t d_out, int32_t groups) {
    // returns CG (0,1,2,4) when the MFMA fast path applies, else -1
    if (d_in != d_out) return -1;
    if (d_in != 16 && d_in != 32 && d_in != 64 && d_in != 128) return -1;
    if (groups == 0) return 0;
    if (d_in % groups) return -1;
    const int64_t cg = d_in / groups;
    if (cg == 1 || cg == 2 || cg == 4) return (int)cg;
    return -1;
}

int narrow_cg(int64_t d_in, int64_t d_out, int32_t groups) {
    // CG when the narrow (2-column) kernels apply, else -1
    if (d_out != 2) return -1;
    if (d_in != 16 && d_in != 32 && d_in != 64 && d_in != 128) return -1;
    if (groups == 0) return 0;
    if (d_in % groups) return -1;
    const int64_t cg = d_in / groups;
    return (cg == 1 || cg == 2 || cg == 4) ? (int)cg : -1;
}

int64_t fwd_blocks(int64_t n_rows) {
    int64_t b = ((n_rows + 15) / 16 + 3) / 4;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return b;
}
int64_t wgrad_blocks(int64_t n_rows) {
    int64_t b = (n_rows + 31) / 32;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return b;
}

}  // namespace

// Dynamic LDS above 64 KB needs the function attribute once per (device, kernel); remembering it keeps the call out
// of the launch path (and out of HIP-graph captures, where only stream operations should occur).
int gode_set_lds_once(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(dev, fn);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    done[key] = bytes;
    return 0;
}

#define GODE_DISPATCH_NJ_CG(NJV, CGV, MACRO)                                        \
    if (nj == NJV && cg == CGV) { MACRO(NJV, CGV) }

#define GODE_DISPATCH_ALL(MACRO)                                                    \
    GODE_DISPATCH_NJ_CG(1, 0, MACRO) GODE_DISPATCH_NJ_CG(1, 1, MACRO)               \
    GODE_DISPATCH_NJ_CG(2, 0, MACRO) GODE_DISPATCH_NJ_CG(2, 1, MACRO)               \
    GODE_DISPATCH_NJ_CG(4, 0, MACRO) GODE_DISPATCH_NJ_CG(4, 2, MACRO)               \
    GODE_DISPATCH_NJ_CG(8, 0, MACRO) GODE_DISPATCH_NJ_CG(8, 4, MACRO)               \
    GODE_DISPATCH_NJ_CG(1, 2, MACRO) GODE_DISPATCH_NJ_CG(1, 4, MACRO)               \
    GODE_DISPATCH_NJ_CG(2, 2, MACRO) GODE_DISPATCH_NJ_CG(2, 4, MACRO)               \
    GODE_DISPATCH_NJ_CG(4, 1, MACRO) GODE_DISPATCH_NJ_CG(4, 4, MACRO)               \
    GODE_DISPATCH_NJ_CG(8, 1, MACRO) GODE_DISPATCH_NJ_CG(8, 2, MACRO)

static int check_common(const gode_lincomb_t* xin, int64_t n_rows, int64_t d_in, int32_t groups, int64_t d_out) {
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || groups < 0) return GODE_E_SHAPE;
    if (groups > 0 && d_in % groups) return GODE_E_SHAPE;
    if (n_rows > INT32_MAX || d_in > 2048 || d_out > 65536) return GODE_E_RANGE;
    return check_lincomb(xin, true);
}

extern "C" int gode_gn_time_gemm_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d_in, int32_t groups,
                                     float eps, const float* gamma, const float* beta, const float* W,
                                     int64_t d_out, int has_time, float t, float* S, void* stream)
{
    return gode_gn_time_gemm_xout_f32(xin, n_rows, d_in, groups, eps, gamma, beta, W, d_out, has_time, t, S, nullptr,
                                      stream);
}

extern "C" int gode_gn_time_gemm_xout_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d_in, int32_t groups,
                                          float eps, const float* gamma, const float* beta, const float* W,
                                          int64_t d_out, int has_time, float t, float* S, float* x_out, void* stream)
{
    int rc = check_common(xin, n_rows, d_in, groups, d_out); if (rc) return rc;
    if (n_rows == 0) return 0;
    if (!W || !S) return GODE_E_NULLPTR;
    if (x_out && (((uintptr_t)x_out) & 15)) return GODE_E_ALIGN;
    has_time = has_time ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    LinComb lc = make_lincomb(xin);
    const int cg = fast_cg(d_in, d_out, groups);
    const bool al = lincomb_aligned16(xin) && !(((uintptr_t)S) & 15) && !(((uintptr_t)W) & 15) &&
                    (!gamma || !(((uintptr_t)gamma) & 15)) && (!beta || !(((uintptr_t)beta) & 15));
    // producer / consumer form on the bf16 matrix cores (gemm_pc.hip): fwd_pc bit 0 = launches of <= 2 terms, bit 1 = 3
    // and more terms (with or without x_out)
    if (cg >= 0 && al && d_in == 128 && (n_rows >= kWgradSplitMinRows || gode_opt_wgrad_split_small()) &&
        ((lc.n <= 2 && !x_out) ? (gode_opt_fwd_pc() & 1) : (gode_opt_fwd_pc() & 2))) {
        rc = gode_pc_fwd_launch(lc, n_rows, eps, gamma, beta, W, has_time, t, S, x_out, cg, s);
        if (rc != GODE_E_UNSUPPORTED) return rc;
    }
    if (cg >= 0 && al && d_in == 128 && !x_out && (gode_opt_gemm_split() == 1 || (gode_opt_gemm_split() == 2 && lc.n <= 2 && n_rows >= 65536))) {
        const size_t lds = (size_t)3 * 128 * (128 + 8) * sizeof(unsigned short) + 3 * 128 * sizeof(float);
        int64_t blocks = ((n_rows + 15) / 16 + 7) / 8; if (blocks < 1) blocks = 1; if (blocks > 256) blocks = 256;
#define GODE_FWDS2(CGV, NXV) { rc = set_lds(gn_gemm_fwd_split_kernel<CGV, NXV>, lds); if (rc) return rc;    \
          const int slot = gode_prof_begin(s, d_in, n_rows, (int64_t)lc.n - 1, GODE_PROF_GEMM_FWD | GODE_PROF_FORM_SPLIT); \
          hipLaunchKernelGGL((gn_gemm_fwd_split_kernel<CGV, NXV>), dim3((unsigned)blocks), dim3(512), lds, s, \
                             lc, (int)n_rows, eps, gamma, beta, W, has_time, t, S);                           \
          gode_prof_end(s, slot);                                                                             \
          GODE_LAUNCH_CHECK(); return 0; }
#define GODE_FWDS(CGV) { if (lc.n == 1) GODE_FWDS2(CGV, 1) else if (lc.n == 2) GODE_FWDS2(CGV, 2) else GODE_FWDS2(CGV, 0) }
        if (cg == 0) GODE_FWDS(0) else if (cg == 1) GODE_FWDS(1) else if (cg == 2) GODE_FWDS(2) else GODE_FWDS(4)
#undef GODE_FWDS
#undef GODE_FWDS2
    }
    if (cg >= 0 && al) {
        const int nj = (int)(d_in / 16);
        const size_t lds = ((size_t)d_in * (d_in + 4) + 3 * d_in) * sizeof(float);
        const int64_t blocks = fwd_blocks(n_rows);
#define GODE_FWD(NJV, CGV)                                                                          \
        { rc = set_lds(gn_gemm_fwd_kernel<NJV, CGV>, lds); if (rc) return rc;                       \
          const int slot = gode_prof_begin(s, d_in, n_rows, (int64_t)lc.n - 1 + (x_out ? 1 : 0), GODE_PROF_GEMM_FWD); \
          hipLaunchKernelGGL((gn_gemm_fwd_kernel<NJV, CGV>), dim3((unsigned)blocks), dim3(256), lds, s, \
                             lc, (int)n_rows, eps, gamma, beta, W, has_time, t, S, x_out,           \
                             (const float*)nullptr, (float*)nullptr);                               \
          gode_prof_end(s, slot);                                                                   \
          GODE_LAUNCH_CHECK(); return 0; }
        GODE_DISPATCH_ALL(GODE_FWD)
#undef GODE_FWD
    }
    {
        const int cg = narrow_cg(d_in, d_out, groups);
        if (cg >= 0 && al) {
            const int nj = (int)(d_in / 16);
            const int64_t blocks = fwd_blocks(n_rows);
#define GODE_NFWD(NJV, CGV)                                                                                 \
            { hipLaunchKernelGGL((gn_narrow_fwd_kernel<NJV, CGV, 2>), dim3((unsigned)blocks), dim3(256), 0, s, \
                                 lc, (int)n_rows, eps, gamma, beta, W, has_time, t, S, x_out);              \
              GODE_LAUNCH_CHECK(); return 0; }
            GODE_DISPATCH_ALL(GODE_NFWD)
#undef GODE_NFWD
        }
    }
    if (x_out) { rc = gode_lincomb_f32(x_out, xin, n_rows * d_in, stream); if (rc) return rc; }
    const size_t lds = ((size_t)RB * d_in + (size_t)RB * (groups > 0 ? groups : 1) * 2) * sizeof(float);
    rc = set_lds(gn_gemm_fwd_generic, lds); if (rc) return rc;
    int64_t blocks = (n_rows + RB - 1) / RB; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gn_gemm_fwd_generic, dim3((unsigned)blocks), dim3(256), lds, s, lc, (int)n_rows, (int)d_in,
                       (int)groups, eps, gamma, beta, W, (int)d_out, has_time, t, S);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gn_time_gemm_pair_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d, int32_t groups, float eps,
                                          const float* gamma, const float* beta, const float* Wa, const float* Wb,
                                          int has_time, float t, float* Sa, float* Sb, float* x_out, void* stream)
{
    // two square products over the same normalised input in one launch (grid.y selects the pair); shapes outside the
    // MFMA path run as two calls
    int rc = check_common(xin, n_rows, d, groups, d); if (rc) return rc;
    if (n_rows == 0) return 0;
    if (!Wa || !Wb || !Sa || !Sb) return GODE_E_NULLPTR;
    has_time = has_time ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    LinComb lc = make_lincomb(xin);
    const int cg = fast_cg(d, d, groups);
    const bool al = lincomb_aligned16(xin) && !((((uintptr_t)Sa) | ((uintptr_t)Sb) | ((uintptr_t)Wa) | ((uintptr_t)Wb)) & 15) &&
                    (!gamma || !(((uintptr_t)gamma) & 15)) && (!beta || !(((uintptr_t)beta) & 15)) &&
                    (!x_out || !(((uintptr_t)x_out) & 15));
    if (cg >= 0 && al) {
        const int nj = (int)(d / 16);
        const size_t lds = ((size_t)d * (d + 4) + 3 * d) * sizeof(float);
        const int64_t blocks = fwd_blocks(n_rows);
#define GODE_FWDP(NJV, CGV)                                                                         \
        { rc = set_lds(gn_gemm_fwd_kernel<NJV, CGV>, lds); if (rc) return rc;                       \
          hipLaunchKernelGGL((gn_gemm_fwd_kernel<NJV, CGV>), dim3((unsigned)blocks, 2), dim3(256), lds, s, \
                             lc, (int)n_rows, eps, gamma, beta, Wa, has_time, t, Sa, x_out, Wb, Sb);  \
          GODE_LAUNCH_CHECK(); return 0; }
        GODE_DISPATCH_ALL(GODE_FWDP)
#undef GODE_FWDP
    }
    rc = gode_gn_time_gemm_xout_f32(xin, n_rows, d, groups, eps, gamma, beta, Wa, d, has_time, t, Sa, x_out, stream);
    if (rc) return rc;
    return gode_gn_time_gemm_f32(xin, n_rows, d, groups, eps, gamma, beta, Wb, d, has_time, t, Sb, stream);
}

extern "C" int64_t gode_gemm_bwd_parts(int64_t n_rows) {
    // upper bound valid for both the MFMA path (one partial per block) and the generic path
    int64_t a = fwd_blocks(n_rows);
    int64_t b = (n_rows + RB - 1) / RB; if (b > 2048) b = 2048; if (b < 1) b = 1;
    return a > b ? a : b;
}

extern "C" int gode_gn_time_gemm_bwd_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d_in, int32_t groups,
                                         float eps, const float* gamma, const float* W, int64_t d_out,
                                         int has_time, const float* dS, float out_scale,
                                         const gode_lincomb_t* pre, float* dx,
                                         float* dgamma_part, float* dbeta_part, void* stream)
{
    int rc = check_common(xin, n_rows, d_in, groups, d_out); if (rc) return rc;
    if (n_rows == 0) return 0;
    if (!W || !dS || !dx) return GODE_E_NULLPTR;
    if ((dgamma_part == nullptr) != (dbeta_part == nullptr)) return GODE_E_NULLPTR;
    if (pre && pre->n > 0) { rc = check_lincomb(pre, true); if (rc) return rc; } else pre = nullptr;
    has_time = has_time ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    LinComb lc = make_lincomb(xin);
    LinComb lpre = make_lincomb(pre);
    const int64_t n_part = gode_gemm_bwd_parts(n_rows);
    const int cg = fast_cg(d_in, d_out, groups);
    const bool al = lincomb_aligned16(xin) && lincomb_aligned16(pre) && !(((uintptr_t)dS) & 15) && !(((uintptr_t)dx) & 15) &&
                    (!gamma || !(((uintptr_t)gamma) & 15)) &&
                    (!dgamma_part || (!(((uintptr_t)dgamma_part) & 15) && !(((uintptr_t)dbeta_part) & 15)));
    if ((cg == 0 || cg == 4) && al && d_in == 128 && gode_opt_bwd_pc() &&
        (n_rows >= kWgradSplitMinRows || gode_opt_wgrad_split_small())) {
        rc = gode_pc_bwd_launch(lc, n_rows, eps, gamma, W, has_time, dS, out_scale, lpre, dx, dgamma_part, dbeta_part,
                                n_part, cg, s);
        if (rc != GODE_E_UNSUPPORTED) return rc;
    }
    if (cg >= 0 && al) {
        const int nj = (int)(d_in / 16);
        size_t lds = ((size_t)d_in * (d_in + 4) + d_in) * sizeof(float);
        if (lds < (size_t)8 * d_in * sizeof(float)) lds = (size_t)8 * d_in * sizeof(float);
        const int64_t blocks = fwd_blocks(n_rows);
#define GODE_BWD(NJV, CGV)                                                                          \
        { rc = set_lds(gn_gemm_bwd_kernel<NJV, CGV>, lds); if (rc) return rc;                       \
          const int slot = gode_prof_begin(s, d_in, n_rows, (int64_t)lc.n - 1 + lpre.n, GODE_PROF_GEMM_BWD); \
          hipLaunchKernelGGL((gn_gemm_bwd_kernel<NJV, CGV>), dim3((unsigned)blocks), dim3(256), lds, s, \
                             lc, (int)n_rows, eps, gamma, W, has_time, dS, out_scale, lpre, dx, dgamma_part, dbeta_part, (int)n_part); \
          gode_prof_end(s, slot);                                                                   \
          GODE_LAUNCH_CHECK(); return 0; }
        GODE_DISPATCH_ALL(GODE_BWD)
#undef GODE_BWD
    }
    {
        const int ncg = narrow_cg(d_in, d_out, groups);
        if (ncg >= 0 && al) {
            const int nj = (int)(d_in / 16);
            const int cg = ncg;
            const int64_t blocks = fwd_blocks(n_rows);
#define GODE_NBWD(NJV, CGV)                                                                                 \
            { hipLaunchKernelGGL((gn_narrow_bwd_kernel<NJV, CGV, 2>), dim3((unsigned)blocks), dim3(256), 0, s, \
                                 lc, (int)n_rows, eps, gamma, W, has_time, dS, out_scale, lpre, dx, dgamma_part, \
                                 dbeta_part, (int)n_part);                                                  \
              GODE_LAUNCH_CHECK(); return 0; }
            GODE_DISPATCH_ALL(GODE_NBWD)
#undef GODE_NBWD
        }
    }
    const size_t g2 = (size_t)RB * (groups > 0 ? groups : 1) * 2;
    const size_t lds = ((size_t)RB * d_in * 2 + g2 * 2) * sizeof(float);
    rc = set_lds(gn_gemm_bwd_generic, lds); if (rc) return rc;
    int64_t blocks = (n_rows + RB - 1) / RB; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gn_gemm_bwd_generic, dim3((unsigned)blocks), dim3(256), lds, s, lc, (int)n_rows, (int)d_in,
                       (int)groups, eps, gamma, W, (int)d_out, has_time, dS, out_scale, lpre, dx, dgamma_part, dbeta_part,
                       (int)n_part);
    GODE_LAUNCH_CHECK();
    return 0;
}

// VJP w.r.t. x AND the weight gradient from one read of x and dS (gemm_pc.hip: gn_gemm_bwd_wgrad_pc_kernel); the same
// outputs as gode_gn_time_gemm_bwd_f32 + gode_wgrad_f32, except that dW_part holds gode_bwd_wgrad_parts(n_rows) partials.
extern "C" int gode_bwd_wgrad_supported(int64_t n_rows, int64_t d_in, int64_t d_out, int32_t groups) {
    const int cg = fast_cg(d_in, d_out, groups);
    return (cg == 0 || cg == 4) && d_in == 128 && gode_opt_bwd_wgrad() && gode_opt_bwd_pc() && gode_opt_wgrad_split() == 8 &&
           (n_rows >= kWgradSplitMinRows || gode_opt_wgrad_split_small());
}
extern "C" int64_t gode_bwd_wgrad_parts(int64_t n_rows) { return gode_pc_bwd_wgrad_parts(n_rows); }

extern "C" int gode_gn_time_gemm_bwd_wgrad_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d_in, int32_t groups,
                                               float eps, const float* gamma, const float* beta, const float* W,
                                               int64_t d_out, int has_time, const float* dS, float out_scale,
                                               const gode_lincomb_t* pre, float* dx, float* dgamma_part,
                                               float* dbeta_part, float* dW_part, void* stream)
{
    int rc = check_common(xin, n_rows, d_in, groups, d_out); if (rc) return rc;
    if (n_rows == 0) return 0;
    if (!W || !dS || !dx || !dW_part) return GODE_E_NULLPTR;
    if ((dgamma_part == nullptr) != (dbeta_part == nullptr)) return GODE_E_NULLPTR;
    if (pre && pre->n > 0) { rc = check_lincomb(pre, true); if (rc) return rc; } else pre = nullptr;
    if (!gode_bwd_wgrad_supported(n_rows, d_in, d_out, groups)) return GODE_E_UNSUPPORTED;
    const bool al = lincomb_aligned16(xin) && lincomb_aligned16(pre) && !(((uintptr_t)dS) & 15) && !(((uintptr_t)dx) & 15) &&
                    (!gamma || !(((uintptr_t)gamma) & 15)) && (!beta || !(((uintptr_t)beta) & 15)) &&
                    (!dgamma_part || (!(((uintptr_t)dgamma_part) & 15) && !(((uintptr_t)dbeta_part) & 15)));
    if (!al) return GODE_E_ALIGN;
    return gode_pc_bwd_wgrad_launch(make_lincomb(xin), n_rows, eps, gamma, beta, W, has_time ? 1 : 0, dS, out_scale,
                                    make_lincomb(pre), dx, dgamma_part, dbeta_part, gode_gemm_bwd_parts(n_rows), dW_part,
                                    fast_cg(d_in, d_out, groups), (hipStream_t)stream);
}

extern "C" int64_t gode_wgrad_parts(int64_t n_rows) {
    return wgrad_blocks(n_rows);
}

extern "C" int gode_wgrad_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d_in, int32_t groups, float eps,
                              const float* gamma, const float* beta, const float* dS, int64_t d_out, int has_time,
                              float* dW_part, void* stream)
{
    int rc = check_common(xin, n_rows, d_in, groups, d_out); if (rc) return rc;
    if (!dW_part || (n_rows > 0 && !dS)) return GODE_E_NULLPTR;
    has_time = has_time ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    LinComb lc = make_lincomb(xin);
    const int64_t blocks = wgrad_blocks(n_rows);
    const int cg = fast_cg(d_in, d_out, groups);
    const bool al = lincomb_aligned16(xin) && !(((uintptr_t)dS) & 15) &&
                    (!gamma || !(((uintptr_t)gamma) & 15)) && (!beta || !(((uintptr_t)beta) & 15));
    if ((cg == 0 || cg == 4) && al && d_in == 128 && n_rows > 0 && gode_opt_wgrad_split() &&
        (n_rows >= kWgradSplitMinRows || gode_opt_wgrad_split_small())) {
        const size_t lds = (size_t)2 * 6 * 576 * 16;                    // two buffers of six piece arrays: 108 KB, one block per CU
        const int nt = gode_opt_wgrad_split();
        const int nx = lc.n <= 2 ? lc.n : 0;            // 128 registers per wave: two raw terms at most
#define GODE_WGS(CGV, NTV, NXV)                                                                     \
        { rc = set_lds(wgrad_split_kernel<CGV, NTV, NXV>, lds); if (rc) return rc;                  \
          const int slot = gode_prof_begin(s, d_in, n_rows, (int64_t)lc.n - 1, GODE_PROF_WGRAD | GODE_PROF_FORM_PC);   \
          hipLaunchKernelGGL((wgrad_split_kernel<CGV, NTV, NXV>), dim3((unsigned)blocks), dim3(1024), lds, s, \
                             lc, (int)n_rows, eps, gamma, beta, dS, has_time, dW_part);             \
          gode_prof_end(s, slot);                                                                   \
          GODE_LAUNCH_CHECK(); return 0; }
#define GODE_WGS_NX(CGV, NTV)                                                                       \
        { if (nx == 1) GODE_WGS(CGV, NTV, 1) else if (nx == 2) GODE_WGS(CGV, NTV, 2) else GODE_WGS(CGV, NTV, 0) }
        if (nt == 6) { if (cg == 0) GODE_WGS_NX(0, 6) else GODE_WGS_NX(4, 6) }
        else { if (cg == 0) GODE_WGS_NX(0, 8) else GODE_WGS_NX(4, 8) }
#undef GODE_WGS_NX
#undef GODE_WGS
    }
    if (cg >= 0 && al) {
        const int nj = (int)(d_in / 16);
        const size_t lds = (size_t)2 * 32 * (d_in + 16) * sizeof(float);
#define GODE_WG(NJV, CGV)                                                                           \
        { rc = set_lds(wgrad_kernel<NJV, CGV>, lds); if (rc) return rc;                             \
          const int slot = gode_prof_begin(s, d_in, n_rows, (int64_t)lc.n - 1, GODE_PROF_WGRAD);   \
          hipLaunchKernelGGL((wgrad_kernel<NJV, CGV>), dim3((unsigned)blocks), dim3(256), lds, s,   \
                             lc, (int)n_rows, eps, gamma, beta, dS, has_time, dW_part);             \
          gode_prof_end(s, slot);                                                                   \
          GODE_LAUNCH_CHECK(); return 0; }
        GODE_DISPATCH_ALL(GODE_WG)
#undef GODE_WG
    }
    {
        const int ncg = narrow_cg(d_in, d_out, groups);
        if (ncg >= 0 && al && !(((uintptr_t)dW_part) & 3)) {
            const int nj = (int)(d_in / 16);
            const int cg = ncg;
#define GODE_NWG(NJV, CGV)                                                                                  \
            { hipLaunchKernelGGL((narrow_wgrad_kernel<NJV, CGV, 2>), dim3((unsigned)blocks), dim3(256), 0, s, \
                                 lc, (int)n_rows, eps, gamma, beta, dS, has_time, dW_part);                 \
              GODE_LAUNCH_CHECK(); return 0; }
            GODE_DISPATCH_ALL(GODE_NWG)
#undef GODE_NWG
        }
    }
    const size_t lds = ((size_t)RB * d_in + (size_t)RB * (groups > 0 ? groups : 1) * 2) * sizeof(float);
    rc = set_lds(wgrad_generic, lds); if (rc) return rc;
    hipLaunchKernelGGL(wgrad_generic, dim3((unsigned)blocks), dim3(256), lds, s, lc, (int)n_rows, (int)d_in,
                       (int)groups, eps, gamma, beta, dS, (int)d_out, has_time, dW_part);
    GODE_LAUNCH_CHECK();
    return 0;
}


// ---- stand-alone GroupNorm on an (n_rows x d) matrix (nn.GroupNorm applied to a 2-D tensor, GCN/models.py:88,133-156)
// The same generic kernels with the dense product switched off.
extern "C" int64_t gode_group_norm_parts(int64_t n_rows) {
    int64_t b = (n_rows + RB - 1) / RB; if (b > 2048) b = 2048; if (b < 1) b = 1;
    return b;
}

extern "C" int gode_group_norm_f32_fwd(const float* x, int64_t n_rows, int64_t d, int32_t groups, float eps,
                                       const float* gamma, const float* beta, float* y, void* stream)
{
    if (n_rows < 0 || d <= 0 || groups <= 0 || d % groups) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!x || !y) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || d > 2048) return GODE_E_RANGE;
    gode_lincomb_t lc; lc.n = 1; lc.coef[0] = 1.f; lc.ptr[0] = x;
    const size_t lds = ((size_t)RB * d + (size_t)RB * groups * 2) * sizeof(float);
    int rc = set_lds(gn_gemm_fwd_generic, lds); if (rc) return rc;
    hipLaunchKernelGGL(gn_gemm_fwd_generic, dim3((unsigned)gode_group_norm_parts(n_rows)), dim3(256), lds, (hipStream_t)stream,
                       make_lincomb(&lc), (int)n_rows, (int)d, (int)groups, eps, gamma, beta, (const float*)nullptr,
                       (int)d, 0, 0.f, y);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_group_norm_f32_bwd(const float* x, int64_t n_rows, int64_t d, int32_t groups, float eps,
                                       const float* gamma, const float* dy, float* dx,
                                       float* dgamma_part, float* dbeta_part, void* stream)
{
    if (n_rows < 0 || d <= 0 || groups <= 0 || d % groups) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!x || !dy || !dx) return GODE_E_NULLPTR;
    if ((dgamma_part == nullptr) != (dbeta_part == nullptr)) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || d > 2048) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_part = gode_group_norm_parts(n_rows);
    gode_lincomb_t lc; lc.n = 1; lc.coef[0] = 1.f; lc.ptr[0] = x;
    LinComb none = make_lincomb(nullptr);
    const size_t g2 = (size_t)RB * groups * 2;
    const size_t lds = ((size_t)RB * d * 2 + g2 * 2) * sizeof(float);
    int rc = set_lds(gn_gemm_bwd_generic, lds); if (rc) return rc;
    hipLaunchKernelGGL(gn_gemm_bwd_generic, dim3((unsigned)n_part), dim3(256), lds, s, make_lincomb(&lc), (int)n_rows, (int)d,
                       (int)groups, eps, gamma, (const float*)nullptr, (int)d, 0, dy, 1.f, none, dx, dgamma_part, dbeta_part,
                       (int)n_part);
    GODE_LAUNCH_CHECK();
    return 0;
}
