// small.hip — the GCN ODE function and its vector-Jacobian products on LAUNCH-BOUND graphs (citation-graph sizes), as
// one kernel per evaluation and one per VJP, gfx950.
//
// Replaces, for graphs of at most 65 536 nodes and widths 16 / 32 (the reference's `--hidden 16` default; its
// GroupNorm(min(32, d), d) has one channel per group there), the launch sequences of
//   ODEfunc.forward (GCN/models.py:172-179): GroupNorm, time column, FixedGraphConvolution (GCN/layers.py:69-75), relu
//   and its autograd (what an adjoint stage needs)
// that the large-graph path issues as 2 + 8 launches (gn_time_gemm, spmm; spmm^T, VJP, weight gradient, column sums,
// reductions).  On Cora (2 708 x 16) every one of those kernels runs for 2-4 us and the training step is bound by the
// NUMBER of kernels (768 per step, ~4.4 us each once captured).  No grid barrier and no persistent kernel is needed to
// fuse them: the products are re-associated so that everything after the gather is ROW-LOCAL.
//
//   forward   z_i = ( sum_j a_ij [t | GN(x_j)] ) W + b      = (A [t|GN(x)]) W  instead of  A ([t|GN(x)] W): the gather
//             collects normalised neighbour rows (GroupNorm is recomputed per gathered row: a few flops), the dense
//             product with the (d+1) x d weight block in LDS follows in the same wave, then bias, relu, the RK combine
//             and the masked cotangent dZ = cot * [z > 0] - ONE launch per f-eval;
//   VJP       dS_i = sum_j a^T_ij dZ_j (gather), dxn_i = dS_i W1^T, GroupNorm backward, k_a row; weight gradient
//             sum_i [1|xn_i]^T dS_i, bias gradient sum_i dZ_i, dgamma / dbeta: per-wave register accumulators, one
//             block partial row per block - ONE launch; a third (gode_reduce_segments_f32) closes the stage.
// Same mathematics as the multi-launch path; the forward sums in another order ((A xn) W vs A (xn W): differences of a
// few ulp, inside the 1e-5 parity bar - tests/test_gpu_gcn.py runs both against the oracle).
//
// Work decomposition: a wave owns a row at a time; a lane holds 4 consecutive columns (float4), so d/4 lanes span a row
// and the wave's 256/d sub-groups share the row's non-zeros (whose indices are fetched 64 at a time, so that the gathers of a
// chunk are in flight together: a 168-neighbour hub of Cora costs three chunks, not 168 dependent trips) and the k-range
// of the dense product; sub-group sums are combined with xor-shuffles (fixed
// order: deterministic).  Bound: launch latency (the graphs live in L2).
#include "common.h"
#include "dense_common.h"
#include "options.h"

namespace {

constexpr int kSmallPartBlocks = 256;

// sum over the sub-groups of a wave (lanes `from` apart and beyond), every lane ends with the total
__device__ __forceinline__ void xor_combine4(float4& v, int from, int upto = 64) {
#pragma unroll
    for (int off = from; off < upto; off <<= 1) {
        v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
        v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
}

// lc_load4 (same order of the multiply-adds: bit for bit what a later launch would combine from memory) with the term
// that names `self` taken from a register
__device__ __forceinline__ float4 lc_load4_self(const LinComb& lc, int64_t idx, const float* self, const float4 self_v) {
    float4 v[GODE_MAX_TERMS];
#pragma unroll
    for (int j = 0; j < GODE_MAX_TERMS; ++j)
        if (j < lc.n && lc.ptr[j] != self) v[j] = *reinterpret_cast<const float4*>(lc.ptr[j] + idx);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < GODE_MAX_TERMS; ++j)
        if (j < lc.n) {
            const float4 x = lc.ptr[j] == self ? self_v : v[j];
            const float c = lc.coef[j];
            r.x = fmaf(c, x.x, r.x); r.y = fmaf(c, x.y, r.y); r.z = fmaf(c, x.z, r.z); r.w = fmaf(c, x.w, r.w);
        }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// forward: out[i] = (sum pre)[i] + alpha * relu(z_i),  z_i = (sum_j a_ij [t | GN(x_j)]) W + b;  Y2[i] = (sum cot)[i] * [z_i > 0]
// ---------------------------------------------------------------------------------------------------------------
// GW = lanes that share a row: 64 (a wave per row: citation graphs of a few thousand rows, where a 168-neighbour hub wants the
// whole wave) or 16 (FOUR rows per wave: from 8 192 rows on the launch is bound by rows in flight - 97 registers allow
// four waves per SIMD, 4 096 waves on the chip, each a ~4.5 us chain of dependent latencies - Pubmed's 19 717 rows took
// 26.7 us per launch with a wave per row).
// (the next-stage arguments travel only with the launches that use them: an rk4 evaluation on Cora is ~8 us of launch-bound
// work, and 112 more bytes of kernel arguments in each of its 128 launches per step showed in the step time)
template <bool NEXT> struct NextArgs { LinComb nxt; float* x_next; };
template <> struct NextArgs<false> {};

template <int D, int CG, int GW, bool NEXT>
__global__ __launch_bounds__(256) void gcn_feval_small_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                             const float* __restrict__ val, LinComb xin, int n_rows,
                                                             float eps, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float t, float alpha,
                                                             LinComb pre, LinComb cot, float* __restrict__ Y2,
                                                             float* __restrict__ out, NextArgs<NEXT> nx)
{
    constexpr int LPR = D / 4, SG = GW / LPR, RPW = 64 / GW, PU = 4;   // lanes per row, sub-groups per row, rows per wave, gathers in flight per lane
    __shared__ __attribute__((aligned(16))) float Ws[(D + 1) * D];
    __shared__ __attribute__((aligned(16))) float mrow[4][RPW][D + 4];       // [0] = t * rowsum, [4 ..] = aggregated GN rows
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, gl = l & (GW - 1), q = gl & (LPR - 1), s = gl / LPR;
    // the kernel is a chain of dependent memory latencies (row pointer -> column -> operand row): the first row's
    // pointers are requested before the weight block is staged, so that the two round trips overlap
    const int row0 = (blockIdx.x * 4 + wave) * RPW + l / GW;
    int row = row0;
    int b = 0, e = 0;
    if (row < n_rows) { b = rowptr[row]; e = rowptr[row + 1]; }
    for (int i = threadIdx.x; i < (D + 1) * D; i += 256) Ws[i] = W[i];
    __syncthreads();
    const float4 gm = gamma ? ld4(gamma + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bt = beta ? ld4(beta + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bi = bias ? ld4(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* mr = mrow[wave][l / GW];
    for (; row < n_rows; row += gridDim.x * 4 * RPW) {                              // uniform over the row's lanes
        if (row != row0) { b = rowptr[row]; e = rowptr[row + 1]; }
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        float r = 0.f;
        // indices and values of GW non-zeros at once, then the neighbour rows with the indices taken from registers: the
        // gathers of a chunk do not depend on each other and issue back to back (a loop of col[j] -> x[col[j]] trips made
        // Cora's 168-neighbour hub set the kernel's duration: 8.2 us)
        for (int base = b; base < e; base += GW) {
            int cj = 0; float av = 0.f;
            if (base + gl < e) { cj = col[base + gl]; av = val ? val[base + gl] : 1.f; }
            const int cnt = e - base;
            for (int u0 = 0; u0 < LPR; u0 += PU) {
                if (u0 * SG >= cnt) break;
                float4 xv[PU]; float aj[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int j = (u0 + u) * SG + s;           // < GW; slots beyond the row carry weight 0 and row 0
                    aj[u] = __shfl(av, j, GW);
                    xv[u] = lc_load4(xin, (int64_t)__shfl(cj, j, GW) * D + 4 * q);
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const float4 xn = gn_forward_v<CG>(xv[u], eps, gm, bt);
                    m.x = fmaf(aj[u], xn.x, m.x); m.y = fmaf(aj[u], xn.y, m.y); m.z = fmaf(aj[u], xn.z, m.z); m.w = fmaf(aj[u], xn.w, m.w);
                    r += aj[u];
                }
            }
        }
        xor_combine4(m, LPR, GW);
#pragma unroll
        for (int off = LPR; off < GW; off <<= 1) r += __shfl_xor(r, off, 64);
        if (s == 0) {
            *reinterpret_cast<float4*>(mr + 4 + 4 * q) = m;
            if (q == 0) mr[0] = t * r;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dense: this sub-group's share of the k range (k = 0: the time row)
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = s; k <= D; k += SG) {
            const float mk = k == 0 ? mr[0] : mr[3 + k];
            const float4 w = *reinterpret_cast<const float4*>(Ws + k * D + 4 * q);
            z.x = fmaf(mk, w.x, z.x); z.y = fmaf(mk, w.y, z.y); z.z = fmaf(mk, w.z, z.z); z.w = fmaf(mk, w.w, z.w);
        }
        xor_combine4(z, LPR, GW);
        if (s == 0) {
            z.x += bi.x; z.y += bi.y; z.z += bi.z; z.w += bi.w;
            float4 y = make_float4(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f));
            const int64_t o = (int64_t)row * D + 4 * q;
            if (pre.n > 0) {
                const float4 p = lc_load4(pre, o);
                y.x = fmaf(alpha, y.x, p.x); y.y = fmaf(alpha, y.y, p.y); y.z = fmaf(alpha, y.z, p.z); y.w = fmaf(alpha, y.w, p.w);
            } else if (alpha != 1.f) {
                y.x *= alpha; y.y *= alpha; y.z *= alpha; y.w *= alpha;
            }
            *reinterpret_cast<float4*>(out + o) = y;
            if (Y2) {
                float4 g = lc_load4(cot, o);
                g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f; g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
                *reinterpret_cast<float4*>(Y2 + o) = g;
            }
            // the NEXT stage's combined input, row by row (a term that names `out` takes this row's value from the
            // register): the next evaluation then gathers ONE array per neighbour instead of one per term of its stage
            // input - 4.3 terms on average over the six stages of a dopri5 step, and the gather is what this kernel waits for
            if constexpr (NEXT) *reinterpret_cast<float4*>(nx.x_next + o) = lc_load4_self(nx.nxt, o, out, y);
        }
        __builtin_amdgcn_wave_barrier();                         // the next row's mrow stores follow these reads
    }
}

// GroupNorm backward of one float4 (CG = 1, 2 or 4 channels per group, all inside the float4), the arithmetic of
// gn_gemm_bwd_kernel: ATen's algebraic form for the ill-conditioned 1- and 2-channel groups, the normalised form for 4.
template <int CG>
__device__ __forceinline__ float4 gn_backward4(const float4 x, const float4 dy, const float4 gm, float eps, float4& xh_out) {
    float4 mean, rstd;
    gn_stats<CG>(x, eps, mean, rstd);
    const float4 xh = make_float4((x.x - mean.x) * rstd.x, (x.y - mean.y) * rstd.y, (x.z - mean.z) * rstd.z, (x.w - mean.w) * rstd.w);
    xh_out = xh;
    const float4 dh = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
    if (CG == 4) {
        const float m1 = ((dh.x + dh.y) + (dh.z + dh.w)) * 0.25f;
        const float m2 = ((dh.x * xh.x + dh.y * xh.y) + (dh.z * xh.z + dh.w * xh.w)) * 0.25f;
        const float rs = rstd.x;
        return make_float4(rs * (dh.x - m1 - xh.x * m2), rs * (dh.y - m1 - xh.y * m2), rs * (dh.z - m1 - xh.z * m2), rs * (dh.w - m1 - xh.w * m2));
    }
    const float4 px = make_float4(dh.x * x.x, dh.y * x.y, dh.z * x.z, dh.w * x.w);
    float4 ds, db;
    if (CG == 1) { ds = px; db = dh; }
    else {
        ds = make_float4(px.x + px.y, px.x + px.y, px.z + px.w, px.z + px.w);
        db = make_float4(dh.x + dh.y, dh.x + dh.y, dh.z + dh.w, dh.z + dh.w);
    }
    constexpr float sc = 1.0f / CG;
    const float4 r3 = make_float4(rstd.x * rstd.x * rstd.x * sc, rstd.y * rstd.y * rstd.y * sc, rstd.z * rstd.z * rstd.z * sc, rstd.w * rstd.w * rstd.w * sc);
    const float4 c2 = make_float4((db.x * mean.x - ds.x) * r3.x, (db.y * mean.y - ds.y) * r3.y, (db.z * mean.z - ds.z) * r3.z, (db.w * mean.w - ds.w) * r3.w);
    const float4 c3 = make_float4(-c2.x * mean.x - db.x * rstd.x * sc, -c2.y * mean.y - db.y * rstd.y * sc,
                                  -c2.z * mean.z - db.z * rstd.z * sc, -c2.w * mean.w - db.w * rstd.w * sc);
    return make_float4(rstd.x * gm.x * dy.x + c2.x * x.x + c3.x, rstd.y * gm.y * dy.y + c2.y * x.y + c3.y,
                       rstd.z * gm.z * dy.z + c2.z * x.z + c3.z, rstd.w * gm.w * dy.w + c2.w * x.w + c3.w);
}

// ---------------------------------------------------------------------------------------------------------------
// VJP: dS_i = sum_j aT_ij dZ_j;  ka_i = (sum pre)_i + out_scale * GN'(x_i)^T (dS_i W1^T);  block partial row
// part[block] = [ sum_i [1|xn_i]^T dS_i  ((D+1) x D, row 0 = colsum(dS): the time row) | colsum(dZ) | dgamma | dbeta | a_t' share ]
// ---------------------------------------------------------------------------------------------------------------
template <int D, int CG, int GW>
__global__ __launch_bounds__(256) void gcn_vjp_small_kernel(const int* __restrict__ rowptrT, const int* __restrict__ colT,
                                                           const float* __restrict__ valT, LinComb xin, int n_rows,
                                                           float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ W,
                                                           const float* __restrict__ dZ, float out_scale, LinComb pre,
                                                           float* __restrict__ ka, float* __restrict__ part)
{
    constexpr int LPR = D / 4, SG = GW / LPR, RPW = 64 / GW, NS = D / SG, PU = 4;     // NS = columns of dW per lane
    constexpr int PLEN = (D + 1) * D + 3 * D + 1;                 // ... | the block's share of a_t' = colsum(dS) . W[0, :]
    __shared__ __attribute__((aligned(16))) float Wt[D * (D + 4)];           // Wt[n][k] = W1[k][n], row stride D + 4
    __shared__ __attribute__((aligned(16))) float dsrow[4][RPW][D];
    __shared__ float red[PLEN];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, gl = l & (GW - 1), q = gl & (LPR - 1), s = gl / LPR;
    const int row0 = (blockIdx.x * 4 + wave) * RPW + l / GW;
    int row = row0;
    int b = 0, e = 0;
    if (row < n_rows) { b = rowptrT[row]; e = rowptrT[row + 1]; }        // requested before the weight block is staged
    for (int i = threadIdx.x; i < D * D; i += 256) {
        const int k = i / D, n = i % D;
        Wt[n * (D + 4) + k] = W[(int64_t)(k + 1) * D + n];
    }
    __syncthreads();
    const float4 gm = gamma ? ld4(gamma + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bt = beta ? ld4(beta + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* dr = dsrow[wave][l / GW];
    float acc[4][NS];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < NS; ++i) acc[a][i] = 0.f;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cz = cs, dg = cs, db = cs;
    for (; row < n_rows; row += gridDim.x * 4 * RPW) {
        if (row != row0) { b = rowptrT[row]; e = rowptrT[row + 1]; }
        float4 dS = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = b; base < e; base += GW) {             // prefetched indices, as in the forward kernel
            int cj = 0; float av = 0.f;
            if (base + gl < e) { cj = colT[base + gl]; av = valT ? valT[base + gl] : 1.f; }
            const int cnt = e - base;
            for (int u0 = 0; u0 < LPR; u0 += PU) {
                if (u0 * SG >= cnt) break;
                float4 gv[PU]; float aj[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int j = (u0 + u) * SG + s;
                    aj[u] = __shfl(av, j, GW);
                    gv[u] = ld4(dZ + (int64_t)__shfl(cj, j, GW) * D + 4 * q);
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    dS.x = fmaf(aj[u], gv[u].x, dS.x); dS.y = fmaf(aj[u], gv[u].y, dS.y);
                    dS.z = fmaf(aj[u], gv[u].z, dS.z); dS.w = fmaf(aj[u], gv[u].w, dS.w);
                }
            }
        }
        xor_combine4(dS, LPR, GW);
        if (s == 0) *reinterpret_cast<float4*>(dr + 4 * q) = dS;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dxn[4q ..] = sum_n dS_n W1[k][n]: this sub-group's share of n
        float4 dy = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = s; n < D; n += SG) {
            const float dn = dr[n];
            const float4 w = *reinterpret_cast<const float4*>(Wt + n * (D + 4) + 4 * q);
            dy.x = fmaf(dn, w.x, dy.x); dy.y = fmaf(dn, w.y, dy.y); dy.z = fmaf(dn, w.z, dy.z); dy.w = fmaf(dn, w.w, dy.w);
        }
        xor_combine4(dy, LPR, GW);
        const int64_t o = (int64_t)row * D + 4 * q;
        const float4 x = lc_load4(xin, o);
        float4 xh;
        const float4 dx = gn_backward4<CG>(x, dy, gm, eps, xh);
        const float4 xn = gn_forward_v<CG>(x, eps, gm, bt);
        if (s == 0) {
            float4 out = make_float4(out_scale * dx.x, out_scale * dx.y, out_scale * dx.z, out_scale * dx.w);
            if (pre.n > 0) { const float4 p = lc_load4(pre, o); out.x += p.x; out.y += p.y; out.z += p.z; out.w += p.w; }
            *reinterpret_cast<float4*>(ka + o) = out;
            dg.x += dy.x * xh.x; dg.y += dy.y * xh.y; dg.z += dy.z * xh.z; dg.w += dy.w * xh.w;
            db.x += dy.x; db.y += dy.y; db.z += dy.z; db.w += dy.w;
            cs.x += dS.x; cs.y += dS.y; cs.z += dS.z; cs.w += dS.w;
            const float4 zz = ld4(dZ + o);
            cz.x += zz.x; cz.y += zz.y; cz.z += zz.z; cz.w += zz.w;
        }
        // weight gradient: lane (q, s) owns rows k = 4q .. 4q+3 of dW1 and columns n = s NS .. s NS + NS - 1
        const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float dn = dr[s * NS + i];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][i] = fmaf(xv[a], dn, acc[a][i]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // rows of the same wave first (lanes GW apart hold the same entries of the partial row: fixed xor order) ...
    if (RPW > 1) {
#pragma unroll
        for (int off = GW; off < 64; off <<= 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[a][i] += __shfl_xor(acc[a][i], off, 64);
            cs.x += __shfl_xor(cs.x, off, 64); cs.y += __shfl_xor(cs.y, off, 64); cs.z += __shfl_xor(cs.z, off, 64); cs.w += __shfl_xor(cs.w, off, 64);
            cz.x += __shfl_xor(cz.x, off, 64); cz.y += __shfl_xor(cz.y, off, 64); cz.z += __shfl_xor(cz.z, off, 64); cz.w += __shfl_xor(cz.w, off, 64);
            dg.x += __shfl_xor(dg.x, off, 64); dg.y += __shfl_xor(dg.y, off, 64); dg.z += __shfl_xor(dg.z, off, 64); dg.w += __shfl_xor(dg.w, off, 64);
            db.x += __shfl_xor(db.x, off, 64); db.y += __shfl_xor(db.y, off, 64); db.z += __shfl_xor(db.z, off, 64); db.w += __shfl_xor(db.w, off, 64);
        }
    }
    // ... then the four waves add through LDS in wave order (fixed order: deterministic); with several rows per wave the
    // first row group writes for the wave
    const bool writer = l < GW;
    for (int w = 0; w < 4; ++w) {
        if (wave == w && writer) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int idx = (1 + 4 * q + a) * D + s * NS + i;
                    red[idx] = (w == 0 ? 0.f : red[idx]) + acc[a][i];
                }
            if (s == 0) {
                const float c4[4][4] = {{cs.x, cs.y, cs.z, cs.w}, {cz.x, cz.y, cz.z, cz.w}, {dg.x, dg.y, dg.z, dg.w}, {db.x, db.y, db.z, db.w}};
                const int base[4] = {0, (D + 1) * D, (D + 1) * D + D, (D + 1) * D + 2 * D};
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int idx = base[v] + 4 * q + a;
                        red[idx] = (w == 0 ? 0.f : red[idx]) + c4[v][a];
                    }
            }
        }
        __syncthreads();
    }
    if (wave == 0) {                                             // a_t' share: the closing launch then only adds columns
        float v = 0.f;
        for (int c = l; c < D; c += 64) v = fmaf(red[c], W[c], v);
        v = wave_sum(v);
        if (l == 0) red[PLEN - 1] = v;
    }
    __syncthreads();
    float* out = part + (int64_t)blockIdx.x * PLEN;
    for (int i = threadIdx.x; i < PLEN; i += 256) out[i] = red[i];
}

// One launch per RK STEP for the small components: theta[j] += sum_s wb[s] * scale_s(j) * sum_p part_s[p][j] over the
// four stages' block partials (the adjoint ODE is linear in a_theta and a_t and a fixed grid never looks at them, so
// their stage derivatives need not exist as vectors).  Outputs: the (d+1) d + 3 d entries of [W | b | gamma | beta] (row
// 0 of W - the time row - scaled by the stage time) and a_t = sum_s wb[s] colsum(dS_s) . W[0, :]  (last block).
struct Finish4 { const float* part[4]; float wb[4]; float ts[4]; };
// 1 024 threads: 32 part-groups x 32 outputs; a thread has 4 x 8 loads per round of 256 partial rows, all independent and
// in flight together - the launch is one memory round trip, not a loop of them (8 part-groups took 34 us).  Output
// out_len is a_t: the sum of the blocks' shares (last column of a partial row).
__global__ __launch_bounds__(1024) void small_finish4_kernel(Finish4 g, int n_part, int plen, int d, float* __restrict__ theta, int out_len)
{
    __shared__ float sm[32][33];
    const int jj = threadIdx.x & 31, qq = threadIdx.x >> 5;
    const int j = (int)blockIdx.x * 32 + jj;
    const int src = j < out_len ? j : (j == out_len ? plen - 1 : -1);
    float v = 0.f;
    if (src >= 0) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p0 = qq; p0 < n_part; p0 += 256) {
            float x[4][8];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + 32 * u;
                    x[s][u] = p < n_part ? g.part[s][(int64_t)p * plen + src] : 0.f;
                }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < 8; ++u) a[s] += x[s][u];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float sc = j < d ? g.wb[s] * g.ts[s] : g.wb[s];       // j < d: the time row of W, scaled by the stage time
            v = fmaf(sc, a[s], v);
        }
    }
    sm[qq][jj] = v;
    __syncthreads();
    if (qq == 0 && src >= 0) {
        float tsum = sm[0][jj];
#pragma unroll
        for (int k = 1; k < 32; ++k) tsum += sm[k][jj];
        theta[j] += tsum;
    }
}

// one stage: ktheta[j] = (j < d ? t : 1) * sum_p part[p][j], ktheta[out_len] = a_t' = sum_p part[p][plen - 1]; the same
// shape of launch (an adaptive step closes every stage by itself; the segment-reduction kernel it used walked the partial
// rows 8 at a time: 34 us for Pubmed's 1 024 rows)
__global__ __launch_bounds__(1024) void small_finish1_kernel(const float* __restrict__ part, int n_part, int plen, int d, float t,
                                                            float* __restrict__ ktheta, int out_len)
{
    __shared__ float sm[32][33];
    const int jj = threadIdx.x & 31, qq = threadIdx.x >> 5;
    const int j = (int)blockIdx.x * 32 + jj;
    const int src = j < out_len ? j : (j == out_len ? plen - 1 : -1);
    float v = 0.f;
    if (src >= 0) {
        for (int p0 = qq; p0 < n_part; p0 += 32 * 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = p0 + 32 * u;
                x[u] = p < n_part ? part[(int64_t)p * plen + src] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) v += x[u];
        }
    }
    sm[qq][jj] = v;
    __syncthreads();
    if (qq == 0 && src >= 0) {
        float tsum = sm[0][jj];
#pragma unroll
        for (int k = 1; k < 32; ++k) tsum += sm[k][jj];
        ktheta[j] = j < d ? t * tsum : tsum;
    }
}

// the same for up to eight stages in one launch (blockIdx.y = stage): an adaptive step closes all its stages at once -
// nothing inside the step reads the small components' stage derivatives (csrc/ode_driver.hip).  Per stage the arithmetic is
// small_finish1_kernel's.
struct FinishN { const float* part[8]; float* ktheta[8]; float t[8]; };
__global__ __launch_bounds__(1024) void small_finish_multi_kernel(FinishN g, int n_part, int plen, int d, int out_len)
{
    __shared__ float sm[32][33];
    const float* __restrict__ part = g.part[blockIdx.y];
    const int jj = threadIdx.x & 31, qq = threadIdx.x >> 5;
    const int j = (int)blockIdx.x * 32 + jj;
    const int src = j < out_len ? j : (j == out_len ? plen - 1 : -1);
    float v = 0.f;
    if (src >= 0) {
        for (int p0 = qq; p0 < n_part; p0 += 32 * 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = p0 + 32 * u;
                x[u] = p < n_part ? part[(int64_t)p * plen + src] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) v += x[u];
        }
    }
    sm[qq][jj] = v;
    __syncthreads();
    if (qq == 0 && src >= 0) {
        float tsum = sm[0][jj];
#pragma unroll
        for (int k = 1; k < 32; ++k) tsum += sm[k][jj];
        g.ktheta[blockIdx.y][j] = j < d ? g.t[blockIdx.y] * tsum : tsum;
    }
}

int small_cg(int64_t d, int32_t groups) {
    // channels per group when the fused small-graph kernels are instantiated for (d, groups), else -1
    // widths 16 and 32 only: at 64 the MFMA kernels of the multi-launch path are faster (measured on Cora, hidden 64,
    // rk4: 5.9 ms per step against 6.2 ms with these kernels instantiated for 64)
    if (d != 16 && d != 32) return -1;
    if (groups <= 0 || d % groups) return -1;
    const int64_t cg = d / groups;
    return (cg == 1 || cg == 2 || cg == 4) ? (int)cg : -1;
}

// four rows per wave from 8 192 rows on (see gcn_feval_small_kernel)
bool rows4(int64_t n) { return n >= 8192; }
int64_t feval_blocks(int64_t n) { const int64_t per = rows4(n) ? 16 : 4; int64_t b = (n + per - 1) / per; if (b < 1) b = 1; if (b > 2048) b = 2048; return b; }

}  // namespace

// one partial row per block: 256 blocks up to 4 096 rows (Cora, Citeseer: 2-3 rows per wave), then rows / 16 up to 1 024
// (Pubmed's 19 717 rows: 5 rows per wave instead of 19; 40 -> 17 us per VJP launch)
extern "C" int64_t gode_gcn_small_parts(int64_t n_rows) {
    int64_t b = (n_rows + 3) / 4;
    int64_t cap = n_rows / 16;
    if (cap < kSmallPartBlocks) cap = kSmallPartBlocks;
    if (cap > 4 * kSmallPartBlocks) cap = 4 * kSmallPartBlocks;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return b;
}
extern "C" int64_t gode_gcn_small_part_len(int64_t d) { return (d + 1) * d + 3 * d + 1; }

extern "C" int gode_gcn_small_supported(int64_t n_rows, int64_t d, int32_t groups) {
    return n_rows > 0 && n_rows <= 65536 && small_cg(d, groups) > 0;
}

#define GODE_SMALL_DISPATCH(MACRO)                                                                           \
    if (d == 16 && cg == 1) { MACRO(16, 1) } else if (d == 16 && cg == 2) { MACRO(16, 2) } else if (d == 16 && cg == 4) { MACRO(16, 4) } \
    else if (d == 32 && cg == 1) { MACRO(32, 1) } else if (d == 32 && cg == 2) { MACRO(32, 2) } else if (d == 32 && cg == 4) { MACRO(32, 4) }

extern "C" int gode_gcn_feval_small_f32(const gode_gcn_odefunc_t* f, const gode_lincomb_t* xin, float t,
                                        float alpha, const gode_lincomb_t* pre, const gode_lincomb_t* cot, float* Y2,
                                        float* out, void* stream)
{
    return gode_gcn_feval_small_next_f32(f, xin, t, alpha, pre, cot, Y2, out, nullptr, nullptr, stream);
}

extern "C" int gode_gcn_feval_small_next_f32(const gode_gcn_odefunc_t* f, const gode_lincomb_t* xin, float t,
                                             float alpha, const gode_lincomb_t* pre, const gode_lincomb_t* cot, float* Y2,
                                             float* out, const gode_lincomb_t* next, float* x_next, void* stream)
{
    if (!f || !xin || !out) return GODE_E_NULLPTR;
    if (x_next) {
        if (!next) return GODE_E_NULLPTR;
        int rcn = check_lincomb(next, true); if (rcn) return rcn;
        if (!lincomb_aligned16(next) || (((uintptr_t)x_next) & 15)) return GODE_E_ALIGN;
        if (x_next == out) return GODE_E_SHAPE;
        for (int j = 0; j < xin->n; ++j) if (xin->ptr[j] == x_next) return GODE_E_SHAPE;      // rows of xin are gathered by other blocks
    } else next = nullptr;
    if (!gode_gcn_small_supported(f->n, f->d, f->groups)) return GODE_E_UNSUPPORTED;
    if (!f->A.rowptr || !f->A.col || !f->W) return GODE_E_NULLPTR;
    int rc = check_lincomb(xin, true); if (rc) return rc;
    if (pre && pre->n > 0) { rc = check_lincomb(pre, true); if (rc) return rc; } else pre = nullptr;
    if (Y2) { rc = check_lincomb(cot, true); if (rc) return rc; } else cot = nullptr;
    if (!lincomb_aligned16(xin) || !lincomb_aligned16(pre) || !lincomb_aligned16(cot) ||
        ((((uintptr_t)out) | ((uintptr_t)Y2) | ((uintptr_t)f->gamma) | ((uintptr_t)f->beta) | ((uintptr_t)f->b)) & 15)) return GODE_E_ALIGN;
    const LinComb lx = make_lincomb(xin), lp = make_lincomb(pre), lcot = make_lincomb(cot), lnext = make_lincomb(next);
    const int64_t d = f->d;
    const int cg = small_cg(d, f->groups);
    const dim3 grid((unsigned)feval_blocks(f->n));
    const bool r4 = rows4(f->n);
    NextArgs<true> nxa; nxa.nxt = lnext; nxa.x_next = x_next;
    const NextArgs<false> nx0;
#define GODE_FEV(DV, CGV) if (r4) GODE_FEV_(DV, CGV, 16) else GODE_FEV_(DV, CGV, 64)
#define GODE_FEV_(DV, CGV, GWV) if (x_next) GODE_FEV__(DV, CGV, GWV, true, nxa) else GODE_FEV__(DV, CGV, GWV, false, nx0)
#define GODE_FEV__(DV, CGV, GWV, NXV, NXA) hipLaunchKernelGGL((gcn_feval_small_kernel<DV, CGV, GWV, NXV>), grid, dim3(256), 0, (hipStream_t)stream, \
                                             f->A.rowptr, f->A.col, f->A.val, lx, (int)f->n, f->eps, f->gamma, f->beta, f->W,    \
                                             f->b, t, alpha, lp, lcot, Y2, out, NXA);
    GODE_SMALL_DISPATCH(GODE_FEV)
#undef GODE_FEV
#undef GODE_FEV_
#undef GODE_FEV__
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gcn_vjp_small_f32(const gode_gcn_odefunc_t* f, const gode_lincomb_t* xin, const float* dZ,
                                      float out_scale, const gode_lincomb_t* pre, float* ka, float* part, void* stream)
{
    if (!f || !xin || !dZ || !ka || !part) return GODE_E_NULLPTR;
    if (!gode_gcn_small_supported(f->n, f->d, f->groups)) return GODE_E_UNSUPPORTED;
    if (!f->AT.rowptr || !f->AT.col || !f->W) return GODE_E_NULLPTR;
    int rc = check_lincomb(xin, true); if (rc) return rc;
    if (pre && pre->n > 0) { rc = check_lincomb(pre, true); if (rc) return rc; } else pre = nullptr;
    if (!lincomb_aligned16(xin) || !lincomb_aligned16(pre) ||
        ((((uintptr_t)dZ) | ((uintptr_t)ka) | ((uintptr_t)f->gamma) | ((uintptr_t)f->beta)) & 15)) return GODE_E_ALIGN;
    const LinComb lx = make_lincomb(xin), lp = make_lincomb(pre);
    const int64_t d = f->d;
    const int cg = small_cg(d, f->groups);
    const dim3 grid((unsigned)gode_gcn_small_parts(f->n));
    const bool r4 = rows4(f->n);
#define GODE_VJS(DV, CGV) if (r4) GODE_VJS_(DV, CGV, 16) else GODE_VJS_(DV, CGV, 64)
#define GODE_VJS_(DV, CGV, GWV) hipLaunchKernelGGL((gcn_vjp_small_kernel<DV, CGV, GWV>), grid, dim3(256), 0, (hipStream_t)stream,           \
                                             f->AT.rowptr, f->AT.col, f->AT.val, lx, (int)f->n, f->eps, f->gamma, f->beta, f->W, \
                                             dZ, out_scale, lp, ka, part);
    GODE_SMALL_DISPATCH(GODE_VJS)
#undef GODE_VJS
#undef GODE_VJS_
    GODE_LAUNCH_CHECK();
    return 0;
}

// theta-k = [ W ((d+1) d, row 0 = t * colsum(dS)) | b | gamma | beta | a_t ] from the block partials of
// gode_gcn_vjp_small_f32, in one launch
extern "C" int gode_gcn_small_finish_f32(const gode_gcn_odefunc_t* f, const float* part, float* ktheta, float t, void* stream)
{
    if (!f || !part || !ktheta) return GODE_E_NULLPTR;
    const int64_t d = f->d, out_len = (d + 1) * d + 3 * d, plen = gode_gcn_small_part_len(d);
    const int64_t parts = gode_gcn_small_parts(f->n);
    const int64_t blocks = (out_len + 1 + 31) / 32;
    hipLaunchKernelGGL(small_finish1_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, part, (int)parts, (int)plen,
                       (int)d, t, ktheta, (int)out_len);
    GODE_LAUNCH_CHECK();
    return 0;
}

// n_stages <= 8 stages at once: stage s reads the partial buffer part + s * parts * part_len and writes ktheta[s] (evaluated
// at ts[s]); every ktheta[s] is bit for bit what gode_gcn_small_finish_f32 writes
extern "C" int gode_gcn_small_finish_multi_f32(const gode_gcn_odefunc_t* f, const float* part, int32_t n_stages,
                                               float* const* ktheta /* host[n_stages] */, const float* ts /* host[n_stages] */,
                                               void* stream)
{
    if (!f || !part || !ktheta || !ts) return GODE_E_NULLPTR;
    if (n_stages < 1 || n_stages > 8) return GODE_E_SHAPE;
    const int64_t d = f->d, out_len = (d + 1) * d + 3 * d, plen = gode_gcn_small_part_len(d);
    const int64_t parts = gode_gcn_small_parts(f->n);
    FinishN g;
    for (int s = 0; s < 8; ++s) {
        const bool on = s < n_stages;
        if (on && !ktheta[s]) return GODE_E_NULLPTR;
        g.part[s] = part + (int64_t)(on ? s : 0) * parts * plen; g.ktheta[s] = on ? ktheta[s] : nullptr; g.t[s] = on ? ts[s] : 0.f;
    }
    const int64_t blocks = (out_len + 1 + 31) / 32;
    hipLaunchKernelGGL(small_finish_multi_kernel, dim3((unsigned)blocks, (unsigned)n_stages), dim3(1024), 0, (hipStream_t)stream, g,
                       (int)parts, (int)plen, (int)d, (int)out_len);
    GODE_LAUNCH_CHECK();
    return 0;
}

// theta += sum_s wb[s] * (stage derivative of the small components from the block partials of stage s), s < 4:
// `part` holds the four stages' partial buffers back to back (stage s at part + s * parts * part_len).
extern "C" int gode_gcn_small_finish4_f32(const gode_gcn_odefunc_t* f, const float* part, float* theta, const float* wb /* host[4] */,
                                          const float* ts /* host[4] */, void* stream)
{
    if (!f || !part || !theta || !wb || !ts) return GODE_E_NULLPTR;
    const int64_t d = f->d, out_len = (d + 1) * d + 3 * d, plen = gode_gcn_small_part_len(d);
    const int64_t parts = gode_gcn_small_parts(f->n);
    Finish4 g;
    for (int s = 0; s < 4; ++s) { g.part[s] = part + (int64_t)s * parts * plen; g.wb[s] = wb[s]; g.ts[s] = ts[s]; }
    const int64_t blocks = (out_len + 1 + 31) / 32;
    hipLaunchKernelGGL(small_finish4_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, g, (int)parts, (int)plen,
                       (int)d, theta, (int)out_len);
    GODE_LAUNCH_CHECK();
    return 0;
}
