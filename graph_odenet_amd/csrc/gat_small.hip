// gat_small.hip — the ROW-LOCAL (dense) half of the GAT ODE function and of its vector-Jacobian product on LAUNCH-BOUND
// graphs (Citeseer: 3 327 nodes), as one kernel per direction, gfx950.
//
// f(t, x) = relu(EdgeAttention([t | GroupNorm(x)]))  (GAT/models.py:172-179 -> GAT/layers.py:95-122).  The two Linear
// layers of the reference (f: o x 2i on [h_src | h_tgt], w: 1 x 2i; GAT/layers.py:43,45) are applied at node level and
// split by role (graph_odenet_amd/gat_ode.py): Ps = [t|xn] Wsrc, Pt = [t|xn] Wtgt, A2 = [t|xn] Wlog (n x 2H logit
// columns).  On a large graph these are three products on the MFMA kernels of gemm.hip, and their autograd is three VJP
// launches, three weight-gradient launches, two column sums and the reductions; on a citation graph every one of them
// runs for 4-27 us (the 2H-column block on the generic kernels) and an adjoint stage is ~18 launches.  Here:
//
//   gode_gat_project_small_f32     Ps, Pt (+ per-head bias), A2 (and the combined stage input) from ONE read of x:
//                                  replaces gode_gn_time_gemm_pair_f32 + gode_gn_time_gemm_f32 (+ the bias add)
//   gode_gat_dense_vjp_small_f32   k_a = GN'(x)^T (dPs Wsrc^T + dPt Wtgt^T + dA2 Wlog^T) (+ pre), and block partials of
//                                  ALL parameter gradients of the stage - dWsrc, dWtgt, dWlog (row 0 = column sums of
//                                  dPs / dPt / dA2: the time rows, and, read again, the bias gradients bf = colsum(dPt),
//                                  bw = odd columns of colsum(dA2)), dgamma, dbeta, and the block's share of
//                                  a_t = colsums . W[0, :]: replaces 3 + 3 + 2 launches
//   gode_gat_small_finish_f32      k_theta and k_a_t from the partials: one launch, every load in flight at once
//
// Work decomposition as in small.hip: a wave owns a row at a time, a lane holds 4 consecutive columns, the wave's
// 256/d sub-groups stride over the reduction index of the dense products and are combined with xor-shuffles (fixed
// order: deterministic).  Weights live in LDS ((d+1) x (2d + 2H) floats: 37 KB at d = 64, H = 8).
#include "common.h"
#include "dense_common.h"
#include "options.h"

namespace {

constexpr int kGatPartBlocks = 512;        // at d = 16: one or two rows per wave on a citation graph (the closing launch reads 16 partial rows per thread at once)

__device__ __forceinline__ void xor_combine4(float4& v, int from) {
#pragma unroll
    for (int off = from; off < 64; off <<= 1) {
        v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
        v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
}
__device__ __forceinline__ void fma4(float4& acc, float a, const float4 w) {
    acc.x = fmaf(a, w.x, acc.x); acc.y = fmaf(a, w.y, acc.y); acc.z = fmaf(a, w.z, acc.z); acc.w = fmaf(a, w.w, acc.w);
}
__device__ __forceinline__ void add4(float4& acc, const float4 v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }

// GroupNorm backward of one float4: small.hip's gn_backward4 (the arithmetic of gn_gemm_bwd_kernel)
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

// four logit-row entries 4 q .. 4 q + 3 of a row of nl floats (nl = 2 H: rows are 8-byte aligned only)
__device__ __forceinline__ float4 load_logit4(const float* __restrict__ p, int c0, int nl) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 < nl) v.x = p[c0];
    if (c0 + 1 < nl) v.y = p[c0 + 1];
    if (c0 + 2 < nl) v.z = p[c0 + 2];
    if (c0 + 3 < nl) v.w = p[c0 + 3];
    return v;
}

// LEN floats (a multiple of 4, 16-byte aligned on both sides) from global memory into LDS with every load of a thread in
// flight before its first store.  (Staging the d = 64 blocks element by element, transposing on the way, was a chain of
// ~40 dependent round trips: 8 of the 33 us of the dense VJP launch, measured with s_memrealtime stamps.)
template <int LEN>
__device__ __forceinline__ void stage_copy(float* __restrict__ dst, const float* __restrict__ src) {
    constexpr int N4 = LEN / 4, NIT = (N4 + 255) / 256;
    float4 tmp[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) { const int i = threadIdx.x + 256 * j; if (i < N4) tmp[j] = ld4(src + 4 * i); }
#pragma unroll
    for (int j = 0; j < NIT; ++j) { const int i = threadIdx.x + 256 * j; if (i < N4) *reinterpret_cast<float4*>(dst + 4 * i) = tmp[j]; }
}

// the two LDS images of the weights, formed ONCE per solve (the weights do not change inside one): [ Wall | WtAll ],
//   Wall  (d+1) x (2d + NLP):  row k = [Wsrc[k] | Wtgt[k] | Wlog[k] | 0 ..]                      (projection kernel)
//   WtAll (2d + NLP) x (d+4):  row c = column c of [Wsrc | Wtgt | Wlog] without the time row, 4 pad floats   (VJP kernel)
__global__ __launch_bounds__(256) void gat_small_pack_kernel(const float* __restrict__ Wsrc, const float* __restrict__ Wtgt,
                                                            const float* __restrict__ Wlog, int d, int nl, int nlp,
                                                            float* __restrict__ out)
{
    const int ws = 2 * d + nlp, n_all = (d + 1) * ws, n_t = ws * (d + 4);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_all + n_t; i += gridDim.x * 256) {
        float v = 0.f;
        if (i < n_all) {
            const int k = i / ws, c = i % ws;
            v = c < d ? Wsrc[k * d + c] : (c < 2 * d ? Wtgt[k * d + c - d] : (c - 2 * d < nl ? Wlog[k * nl + c - 2 * d] : 0.f));
        } else {
            const int j = i - n_all, c = j / (d + 4), k = j % (d + 4);
            if (k < d) v = c < d ? Wsrc[(k + 1) * d + c] : (c < 2 * d ? Wtgt[(k + 1) * d + c - d] : (c - 2 * d < nl ? Wlog[(k + 1) * nl + c - 2 * d] : 0.f));
        }
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// projections:  [Ps | Pt | A2][i] = [t | GN(x_i)] [Wsrc | Wtgt | Wlog]     (NLP = logit columns padded to 4 or 16)
// ---------------------------------------------------------------------------------------------------------------
template <int D, int CG, int NLP>
__global__ __launch_bounds__(256) void gat_project_small_kernel(LinComb xin, int n_rows, float eps,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ Wsrc, const float* __restrict__ Wtgt,
                                                               const float* __restrict__ Wlog, int nl,
                                                               const float* __restrict__ packed,
                                                               const float* __restrict__ pt_bias, float t,
                                                               float* __restrict__ Ps, float* __restrict__ Pt,
                                                               float* __restrict__ A2, float* __restrict__ xout)
{
    constexpr int LPR = D / 4, SG = 64 / LPR, WS = 2 * D + NLP;
    __shared__ __attribute__((aligned(16))) float Wall[(D + 1) * WS];
    __shared__ __attribute__((aligned(16))) float mrow[4][D + 4];            // [0] = t, [4 ..] = the normalised row
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, q = l & (LPR - 1), s = l / LPR;
    if (packed) stage_copy<(D + 1) * WS>(Wall, packed);           // the block as gode_gat_small_pack_f32 laid it out
    else {
        for (int i = threadIdx.x; i < (D + 1) * D; i += 256) {
            const int k = i / D, c = i % D;
            Wall[k * WS + c] = Wsrc[i];
            Wall[k * WS + D + c] = Wtgt[i];
        }
        for (int i = threadIdx.x; i < (D + 1) * NLP; i += 256) {
            const int k = i / NLP, c = i % NLP;
            Wall[k * WS + 2 * D + c] = c < nl ? Wlog[k * nl + c] : 0.f;
        }
    }
    __syncthreads();
    const float4 gm = gamma ? ld4(gamma + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bt = beta ? ld4(beta + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 pb = pt_bias ? ld4(pt_bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_l = 4 * q < NLP;
    float* mr = mrow[wave];
    for (int row = blockIdx.x * 4 + wave; row < n_rows; row += gridDim.x * 4) {            // wave-uniform
        const int64_t o = (int64_t)row * D + 4 * q;
        const float4 x = lc_load4(xin, o);
        const float4 xn = gn_forward_v<CG>(x, eps, gm, bt);
        if (s == 0) {
            if (xout) *reinterpret_cast<float4*>(xout + o) = x;
            *reinterpret_cast<float4*>(mr + 4 + 4 * q) = xn;
            if (q == 0) mr[0] = t;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float4 ps = make_float4(0.f, 0.f, 0.f, 0.f), pt = ps, pl = ps;
        for (int k = s; k <= D; k += SG) {
            const float mk = k == 0 ? mr[0] : mr[3 + k];
            const float* wk = Wall + k * WS + 4 * q;
            fma4(ps, mk, *reinterpret_cast<const float4*>(wk));
            fma4(pt, mk, *reinterpret_cast<const float4*>(wk + D));
            if (has_l) fma4(pl, mk, *reinterpret_cast<const float4*>(wk + 2 * D));
        }
        xor_combine4(ps, LPR);
        xor_combine4(pt, LPR);
        xor_combine4(pl, LPR);                                   // lanes without logit columns carry zeros
        if (s == 0) {
            *reinterpret_cast<float4*>(Ps + o) = ps;
            add4(pt, pb);
            *reinterpret_cast<float4*>(Pt + o) = pt;
            if (has_l) {
                float* ap = A2 + (int64_t)row * nl;
                const int c0 = 4 * q;
                if (c0 < nl) ap[c0] = pl.x;
                if (c0 + 1 < nl) ap[c0 + 1] = pl.y;
                if (c0 + 2 < nl) ap[c0 + 2] = pl.z;
                if (c0 + 3 < nl) ap[c0 + 3] = pl.w;
            }
        }
        __builtin_amdgcn_wave_barrier();                         // the next row's mrow stores follow these reads
    }
}

// d = 64 on the fp32 matrix instruction: a block owns tiles of 16 rows, [Ps | Pt | A2] (16 x NIN) = [t | xn] (16 x 65) . Wall
// (65 x NIN) as nine 16 x 16 output tiles of 17 k-steps (v_mfma_f32_16x16x4_f32: exact fp32 multiply-adds, fixed order); the
// weights' packed image goes straight from L2 into the B operand, once per block.  (The kernel above gives a wave a row:
// 65 LDS-fed k-steps and a shuffle reduction per row - 8.5 us per evaluation on Citeseer, 128 evaluations per step.)
template <int CG, int NLP>
__global__ __launch_bounds__(256) void gat_project_d64_kernel(LinComb xin, int n_rows, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int nl, const float* __restrict__ packed,
                                                             const float* __restrict__ pt_bias, float t,
                                                             float* __restrict__ Ps, float* __restrict__ Pt,
                                                             float* __restrict__ A2, float* __restrict__ xout)
{
    constexpr int D = 64, R = 16, WS = 2 * D + NLP, KS = 17, XS = 69;        // k = 0 .. 67 (k = 0: the time column, 65 .. 67: zero)
    constexpr int NT = (WS + 15) / 16;                                       // 9 output tiles (NLP = 4: the last one has 4 live columns)
    __shared__ float Xs[R * XS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
    const int r = threadIdx.x >> 4, q = threadIdx.x & 15;
    // B operand: Wall[k][c] for k = 4 ks + g, c = 16 nt + rl, the wave's tiles nt = wave, wave + 4, wave + 8
    float wb[3][KS];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int nt = wave + 4 * u, c = 16 * nt + rl;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 4 * ks + g;
            const float w = packed[(k <= D ? k : D) * WS + (c < WS ? c : WS - 1)];           // unconditional, clamped
            wb[u][ks] = (nt < NT && k <= D && c < WS) ? w : 0.f;
        }
    }
    const float4 gm = gamma ? ld4(gamma + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bt = beta ? ld4(beta + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float pbv = (pt_bias && wave < 4) ? pt_bias[16 * wave + rl] : 0.f;   // the Pt tile of wave w is nt = w + 4: columns 16 w ..
    if (threadIdx.x < R) { Xs[threadIdx.x * XS + 65] = 0.f; Xs[threadIdx.x * XS + 66] = 0.f; Xs[threadIdx.x * XS + 67] = 0.f; }
    const int n_tiles = (n_rows + R - 1) / R;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row = tile * R + r;
        const bool ok = row < n_rows;
        const int64_t o = (int64_t)(ok ? row : 0) * D + 4 * q;
        const float4 x = lc_load4(xin, o);
        const float4 xn = gn_forward_v<CG>(x, eps, gm, bt);
        __syncthreads();                                           // the previous tile's operand reads are done
        if (ok && xout) *reinterpret_cast<float4*>(xout + o) = x;
        {
            float* xr = Xs + r * XS + 1 + 4 * q;                   // odd stride: scalar stores
            xr[0] = ok ? xn.x : 0.f; xr[1] = ok ? xn.y : 0.f; xr[2] = ok ? xn.z : 0.f; xr[3] = ok ? xn.w : 0.f;
            if (q == 0) Xs[r * XS] = ok ? t : 0.f;
        }
        __syncthreads();
        float av[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[ks] = Xs[rl * XS + 4 * ks + g];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int nt = wave + 4 * u;
            if (nt >= NT) continue;                                // wave-uniform
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wb[u][ks], acc, 0, 0, 0);
            // D[i = 4 g + e][j = rl]: row tile * 16 + 4 g + e, column 16 nt + rl of [Ps | Pt | A2]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int orow = tile * R + 4 * g + e;
                if (orow >= n_rows) continue;
                if (nt < 4) Ps[(int64_t)orow * D + 16 * nt + rl] = acc[e];
                else if (nt < 8) Pt[(int64_t)orow * D + 16 * (nt - 4) + rl] = acc[e] + pbv;
                else if (rl < nl) A2[(int64_t)orow * nl + rl] = acc[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// VJP of the projections + every parameter-gradient partial of the stage.
// part[block] = [ dWsrc ((D+1) x D, row 0 = colsum dPs) | dWtgt (row 0 = colsum dPt) | dWlog ((D+1) x nl, row 0 = colsum dA2)
//                 | dgamma | dbeta | colsums . [Wsrc[0] | Wtgt[0] | Wlog[0]] ]
// ---------------------------------------------------------------------------------------------------------------
// The path of the gradient through each head's maximum (GAT/layers.py:47), closed here instead of in a launch of its own:
// the max-path step left per-block sums of da (psum) and arg-max candidates (pidx) per head; head h's sum T_h is taken off
// the two dA2 entries its arg-max edge e* feeds - (src(e*), column 2 h) and (tgt(e*), column 2 h + 1) - when those rows are
// loaded.  On the H-fold graph src / tgt are virtual indices (node * H + head).  psum == NULL: nothing to do.
struct MaxFix { const float* psum; const int* pidx; int n_part; const int* esrc; const int* etgt; int n_edges; int H; };

template <int D, int NLP> struct GatVjpShape {
    static constexpr int LPR = D / 4, SG = 64 / LPR, NS = D / SG, NSL = (NLP + SG - 1) / SG;
    static constexpr int NIN = 2 * D + NLP;                      // columns of [dPs | dPt | dA2]
    static constexpr int WT = NIN * (D + 4);                     // transposed weights, row stride D + 4
    static constexpr int RS = D + 1, RSL = NLP + 1;              // odd row strides of the block reduction (LDS banks)
    static constexpr int PMAX = 2 * (D + 1) * RS + (D + 1) * RSL + 2 * D + 1;
    static constexpr int BUF = WT > PMAX ? WT : PMAX;
};

template <int D, int CG, int NLP>
__global__ __launch_bounds__(256) void gat_dense_vjp_small_kernel(LinComb xin, int n_rows, float eps,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ Wsrc, const float* __restrict__ Wtgt,
                                                                 const float* __restrict__ Wlog, int nl,
                                                                 const float* __restrict__ dPs, const float* __restrict__ dPt,
                                                                 const float* __restrict__ dA2, float out_scale, LinComb pre,
                                                                 float* __restrict__ ka, float* __restrict__ part, MaxFix mf,
                                                                 const float* __restrict__ packed_t)
{
    using S = GatVjpShape<D, NLP>;
    constexpr int LPR = S::LPR, SG = S::SG, NS = S::NS, NSL = S::NSL, NIN = S::NIN;
    // Wt[c][k] = W(k + 1, c) over the 2 D + NLP input columns c; the same storage holds the block partial afterwards
    __shared__ __attribute__((aligned(16))) float buf[S::BUF];
    __shared__ __attribute__((aligned(16))) float dsrow[4][NIN];
    __shared__ float mpT[8];
    __shared__ int mpS[8], mpD[8];
    float* Wt = buf;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, q = l & (LPR - 1), s = l / LPR;
    if (threadIdx.x < 8) {
        float tsum = 0.f; int f = INT32_MAX;
        if (mf.psum && (int)threadIdx.x < mf.H)
            for (int b = 0; b < mf.n_part; ++b) { tsum += mf.psum[b * mf.H + threadIdx.x]; f = min(f, mf.pidx[b * mf.H + threadIdx.x]); }
        const bool hit = mf.psum && f < mf.n_edges;
        mpT[threadIdx.x] = hit ? tsum : 0.f;
        mpS[threadIdx.x] = hit ? mf.esrc[f] / mf.H : -1;
        mpD[threadIdx.x] = hit ? mf.etgt[f] / mf.H : -1;
    }
    if (packed_t) stage_copy<S::WT>(Wt, packed_t);               // the transposed image of gode_gat_small_pack_f32
    else {
        for (int i = threadIdx.x; i < D * D; i += 256) {
            const int k = i / D, c = i % D;
            Wt[c * (D + 4) + k] = Wsrc[(int64_t)(k + 1) * D + c];
            Wt[(D + c) * (D + 4) + k] = Wtgt[(int64_t)(k + 1) * D + c];
        }
        for (int i = threadIdx.x; i < D * NLP; i += 256) {
            const int k = i / NLP, c = i % NLP;
            Wt[(2 * D + c) * (D + 4) + k] = c < nl ? Wlog[(int64_t)(k + 1) * nl + c] : 0.f;
        }
    }
    __syncthreads();
    const float4 gm = gamma ? ld4(gamma + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bt = beta ? ld4(beta + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_l = 4 * q < NLP;
    float* dr = dsrow[wave];
    float acc_s[4][NS], acc_t[4][NS], acc_l[4][NSL];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int i = 0; i < NS; ++i) { acc_s[a][i] = 0.f; acc_t[a][i] = 0.f; }
#pragma unroll
        for (int i = 0; i < NSL; ++i) acc_l[a][i] = 0.f;
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cs_s = zero4, cs_t = zero4, cs_l = zero4, dg = zero4, db = zero4;
    // the next row's operands are requested before the current row is worked on: a row is a chain of ~4 us of dependent
    // latency (global -> LDS -> products -> shuffles) and a wave owns several rows at d = 64
    int row = blockIdx.x * 4 + wave;
    float4 ngs = zero4, ngt = zero4, ngl = zero4, nx = zero4;
    if (row < n_rows) {
        const int64_t o0 = (int64_t)row * D + 4 * q;
        ngs = ld4(dPs + o0); ngt = ld4(dPt + o0);
        if (has_l) ngl = load_logit4(dA2 + (int64_t)row * nl, 4 * q, nl);
        nx = lc_load4(xin, o0);
    }
    for (; row < n_rows; row += gridDim.x * 4) {
        const int64_t o = (int64_t)row * D + 4 * q;
        const float4 gs = ngs, gt = ngt, x = nx;
        float4 gl = ngl;
        if (mf.psum && has_l) {                                  // max-path fix of this row's logit cotangents (see MaxFix)
            float gv[4] = {gl.x, gl.y, gl.z, gl.w};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = 4 * q + a, h = (c >> 1) & 7;
                if (c < nl && row == ((c & 1) ? mpD[h] : mpS[h])) gv[a] -= mpT[h];
            }
            gl = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
        {
            const int nrow = row + gridDim.x * 4;
            if (nrow < n_rows) {
                const int64_t o1 = (int64_t)nrow * D + 4 * q;
                ngs = ld4(dPs + o1); ngt = ld4(dPt + o1);
                if (has_l) ngl = load_logit4(dA2 + (int64_t)nrow * nl, 4 * q, nl);
                nx = lc_load4(xin, o1);
            }
        }
        if (s == 0) {
            *reinterpret_cast<float4*>(dr + 4 * q) = gs;
            *reinterpret_cast<float4*>(dr + D + 4 * q) = gt;
            if (has_l) *reinterpret_cast<float4*>(dr + 2 * D + 4 * q) = gl;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dxn[4q ..] = sum_c g_c W(1 + 4q .., c): this sub-group's share of the 2 D + NLP columns
        float4 dy = zero4;
        for (int c = s; c < NIN; c += SG) fma4(dy, dr[c], *reinterpret_cast<const float4*>(Wt + c * (D + 4) + 4 * q));
        xor_combine4(dy, LPR);
        float4 xh;
        const float4 dx = gn_backward4<CG>(x, dy, gm, eps, xh);
        const float4 xn = gn_forward_v<CG>(x, eps, gm, bt);
        if (s == 0) {
            float4 out = make_float4(out_scale * dx.x, out_scale * dx.y, out_scale * dx.z, out_scale * dx.w);
            if (pre.n > 0) add4(out, lc_load4(pre, o));
            *reinterpret_cast<float4*>(ka + o) = out;
            dg.x += dy.x * xh.x; dg.y += dy.y * xh.y; dg.z += dy.z * xh.z; dg.w += dy.w * xh.w;
            add4(db, dy);
            add4(cs_s, gs); add4(cs_t, gt); add4(cs_l, gl);
        }
        // weight gradients: lane (q, s) owns rows 1 + 4q .. 4q + 4 of the three blocks and the columns c = s (mod SG):
        // interleaved, so that the block reduction below touches 32 different LDS banks per instruction
        const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float a_s = dr[s + SG * i], a_t = dr[D + s + SG * i];
#pragma unroll
            for (int a = 0; a < 4; ++a) { acc_s[a][i] = fmaf(xv[a], a_s, acc_s[a][i]); acc_t[a][i] = fmaf(xv[a], a_t, acc_t[a][i]); }
        }
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int c = s + SG * i;
            const float a_l = c < NLP ? dr[2 * D + c] : 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) acc_l[a][i] = fmaf(xv[a], a_l, acc_l[a][i]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                             // every wave is done with Wt: the storage becomes the partial
    // Block reduction in LDS, the four waves in wave order (deterministic).  Layout: rows of D + 1 (NLP + 1) floats - an
    // ODD stride: for one (a, i) the 64 lanes of a wave write rows 4 q + a + 1 and columns s + SG i, i.e. banks
    // (4 q (D+1) + s) mod 32 = 4 q + s: all 32 banks, two lanes each.  (With rows of D floats every lane of the wave hit one
    // of TWO banks: 288 fully serialised LDS instructions per wave and phase, 35 of the kernel's 52 us at d = 64.)
    float* red = buf;
    constexpr int RS = S::RS, RSL = S::RSL, rT = (D + 1) * RS, rL = 2 * rT, rG = rL + (D + 1) * RSL, rB = rG + D, rA = rB + D;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int idx = (1 + 4 * q + a) * RS + s + SG * i;
                    red[idx] = (w == 0 ? 0.f : red[idx]) + acc_s[a][i];
                    red[rT + idx] = (w == 0 ? 0.f : red[rT + idx]) + acc_t[a][i];
                }
#pragma unroll
                for (int i = 0; i < NSL; ++i) {
                    const int c = s + SG * i;
                    if (c < NLP) {
                        const int idx = rL + (1 + 4 * q + a) * RSL + c;
                        red[idx] = (w == 0 ? 0.f : red[idx]) + acc_l[a][i];
                    }
                }
            }
            if (s == 0) {
                const float c4[4][4] = {{cs_s.x, cs_s.y, cs_s.z, cs_s.w}, {cs_t.x, cs_t.y, cs_t.z, cs_t.w},
                                        {dg.x, dg.y, dg.z, dg.w}, {db.x, db.y, db.z, db.w}};
                const int base[4] = {0, rT, rG, rB};
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int idx = base[v] + 4 * q + a;
                        red[idx] = (w == 0 ? 0.f : red[idx]) + c4[v][a];
                    }
                if (has_l) {
                    const float cl[4] = {cs_l.x, cs_l.y, cs_l.z, cs_l.w};
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int idx = rL + 4 * q + a;
                        red[idx] = (w == 0 ? 0.f : red[idx]) + cl[a];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) {                                             // the block's share of a_t' = colsums . time rows of the weights
        float v = 0.f;
        for (int c = l; c < 2 * D + nl; c += 64) {
            const float cs = c < D ? red[c] : (c < 2 * D ? red[rT + c - D] : red[rL + c - 2 * D]);
            const float w0 = c < D ? Wsrc[c] : (c < 2 * D ? Wtgt[c - D] : Wlog[c - 2 * D]);
            v = fmaf(cs, w0, v);
        }
        v = wave_sum(v);
        if (l == 0) red[rA] = v;
    }
    __syncthreads();
    // the partial row leaves in the LDS layout (rows of D + 1 / NLP + 1 floats): a straight copy here, the index arithmetic
    // once per OUTPUT in the closing launch instead of once per element and block (two runtime divisions each: 5 of the
    // 33 us of this launch at d = 64)
    constexpr int PLENP = S::PMAX;
    float* out = part + (int64_t)blockIdx.x * PLENP;
    for (int i = threadIdx.x; i < PLENP; i += 256) out[i] = red[i];
}

// ---------------------------------------------------------------------------------------------------------------
// d = 64: the same launch on the fp32 matrix instruction.  The kernel above gives a wave a ROW at a time - a chain of
// global load -> LDS -> 144 x 64 multiply-adds -> shuffles per row, three rows per wave on Citeseer, then a four-phase
// block reduction of 9 700 partial sums: 29 us per adjoint stage, 23 % of the 8-head training step's kernel time
// (profiles/r04_gat_citeseer_kernel_stats.txt).  Here a block owns TILES of 16 rows and both products are MFMA tiles
// (v_mfma_f32_16x16x4_f32: exact fp32 multiply-adds, fixed order):
//     dY (16 x 64)   = G (16 x NIN) . W^T (NIN x 64)            4 waves x 1 column tile, NIN / 4 k-steps
//     dW (65 x NIN)  = [xn | 1]^T (65 x 16) . G (16 x NIN)      45 tiles of 16 x 16 (the ones row gives the column sums), 4 k-steps
// with G = [dPs | dPt | dA2] staged in LDS (odd row stride: conflict-free as either operand), the weights' transposed
// image read straight from L2 into the B operand (once per block), GroupNorm backward row-wise between the two.  The
// partial row has the layout of the kernel above (gat_small_finish_kernel is shared); a block writes it from the
// accumulators, no reduction phase.
template <int CG, int NLP>
__global__ __launch_bounds__(256) void gat_dense_vjp_d64_kernel(LinComb xin, int n_rows, float eps,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ Wsrc, const float* __restrict__ Wtgt,
                                                               const float* __restrict__ Wlog, int nl,
                                                               const float* __restrict__ dPs, const float* __restrict__ dPt,
                                                               const float* __restrict__ dA2, float out_scale, LinComb pre,
                                                               float* __restrict__ ka, float* __restrict__ part, MaxFix mf,
                                                               const float* __restrict__ packed_t)
{
    constexpr int D = 64, R = 16, NIN = 2 * D + NLP, KS = NIN / 4, NT = 9, MT = 5;
    constexpr int GS = 145, XS = 80, YS = 68;                    // LDS row strides (floats): G, xn, dY
    using S = GatVjpShape<D, NLP>;
    __shared__ __attribute__((aligned(16))) float Gs[R * GS];
    __shared__ __attribute__((aligned(16))) float Xs[R * XS];
    __shared__ __attribute__((aligned(16))) float Ys[R * YS];
    __shared__ __attribute__((aligned(16))) float Pg[2][R][D];   // dy * xhat, dy: summed over the tile's rows below
    __shared__ float valid[R], at_w[4];
    __shared__ float mpT[8];
    __shared__ int mpS[8], mpD[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, rl = lane & 15, g = lane >> 4;
    const int r = threadIdx.x >> 4, q = threadIdx.x & 15;        // row-wise phases: thread (r, q) holds columns 4 q .. 4 q + 3 of row r
    if (threadIdx.x < 8) {
        float tsum = 0.f; int f = INT32_MAX;
        if (mf.psum && (int)threadIdx.x < mf.H)
            for (int b = 0; b < mf.n_part; ++b) { tsum += mf.psum[b * mf.H + threadIdx.x]; f = min(f, mf.pidx[b * mf.H + threadIdx.x]); }
        const bool hit = mf.psum && f < mf.n_edges;
        mpT[threadIdx.x] = hit ? tsum : 0.f;
        mpS[threadIdx.x] = hit ? mf.esrc[f] / mf.H : -1;
        mpD[threadIdx.x] = hit ? mf.etgt[f] / mf.H : -1;
    }
    for (int i = threadIdx.x; i < R * GS; i += 256) Gs[i] = 0.f;             // columns NIN .. 143 stay zero
    // B operand of dY, once per block: W^T[c][k] for the wave's 16 output columns k = 16 wave + rl, rows c = 4 ks + g
    float wt[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wt[ks] = packed_t[(4 * ks + g) * (D + 4) + 16 * wave + rl];
    const float4 gm = gamma ? ld4(gamma + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bt = beta ? ld4(beta + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_l = 4 * q < NLP;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[3][MT];                                            // dW tiles (mt, nt = wave + 4 u)
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[u][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dgb = 0.f;                                             // threads 0-63: dgamma[tid], 64-127: dbeta[tid - 64]
    const int n_tiles = (n_rows + R - 1) / R;
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row = tile * R + r;
        const bool ok = row < n_rows;
        const int64_t o = (int64_t)(ok ? row : 0) * D + 4 * q;
        float4 gs = ld4(dPs + o), gt = ld4(dPt + o), x = lc_load4(xin, o);
        float4 gl = has_l ? load_logit4(dA2 + (int64_t)(ok ? row : 0) * nl, 4 * q, nl) : zero4;
        if (!ok) { gs = zero4; gt = zero4; gl = zero4; x = zero4; }
        if (mf.psum && has_l && ok) {                            // max-path fix of this row's logit cotangents (see MaxFix)
            float gv[4] = {gl.x, gl.y, gl.z, gl.w};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = 4 * q + a, h = (c >> 1) & 7;
                if (c < nl && row == ((c & 1) ? mpD[h] : mpS[h])) gv[a] -= mpT[h];
            }
            gl = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
        float4 xn = gn_forward_v<CG>(x, eps, gm, bt);
        if (!ok) xn = zero4;
        {
            float* gr = Gs + r * GS + 4 * q;                     // odd stride: scalar stores
            gr[0] = gs.x; gr[1] = gs.y; gr[2] = gs.z; gr[3] = gs.w;
            gr[D] = gt.x; gr[D + 1] = gt.y; gr[D + 2] = gt.z; gr[D + 3] = gt.w;
            if (has_l) { gr[2 * D] = gl.x; gr[2 * D + 1] = gl.y; gr[2 * D + 2] = gl.z; gr[2 * D + 3] = gl.w; }
            *reinterpret_cast<float4*>(Xs + r * XS + 4 * q) = xn;
            if (q == 0) valid[r] = ok ? 1.f : 0.f;
        }
        __syncthreads();
        // dY: A[i = rl][k = g] = G[rl][4 ks + g]
        {
            f32x4 dyacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dyacc = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[rl * GS + 4 * ks + g], wt[ks], dyacc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) Ys[(4 * g + e) * YS + 16 * wave + rl] = dyacc[e];
        }
        __syncthreads();
        {
            const float4 dy = *reinterpret_cast<const float4*>(Ys + r * YS + 4 * q);
            float4 xh;
            const float4 dx = gn_backward4<CG>(x, dy, gm, eps, xh);
            if (ok) {
                float4 out = make_float4(out_scale * dx.x, out_scale * dx.y, out_scale * dx.z, out_scale * dx.w);
                if (pre.n > 0) add4(out, lc_load4(pre, o));
                *reinterpret_cast<float4*>(ka + o) = out;
            }
            *reinterpret_cast<float4*>(&Pg[0][r][4 * q]) = make_float4(dy.x * xh.x, dy.y * xh.y, dy.z * xh.z, dy.w * xh.w);
            *reinterpret_cast<float4*>(&Pg[1][r][4 * q]) = dy;
        }
        // dW: A[i = rl][k = g] = [xn | 1][4 ks + g][16 mt + rl],  B[k = g][j = rl] = G[4 ks + g][16 nt + rl]
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int nt = wave + 4 * u;
            if (nt < NT) {                                       // wave-uniform
                float bv[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bv[ks] = Gs[(4 * ks + g) * GS + 16 * nt + rl];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const float av = mt < 4 ? Xs[(4 * ks + g) * XS + 16 * mt + rl] : (rl == 0 ? valid[4 * ks + g] : 0.f);
                        acc[u][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ks], acc[u][mt], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * D) {
            const int which = threadIdx.x >> 6, m = threadIdx.x & 63;
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) sum += Pg[which][rr][m];
            dgb += sum;
        }
        // (the next tile's LDS stores follow a barrier: every read of this tile is above the last one)
    }
    // the partial row, in the layout of gat_dense_vjp_small_kernel's block reduction
    constexpr int RS = S::RS, RSL = S::RSL, rT = (D + 1) * RS, rL = 2 * rT, rG = rL + (D + 1) * RSL, rA = rG + 2 * D;
    float* out = part + (int64_t)blockIdx.x * S::PMAX;
    float at_share = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int nt = wave + 4 * u;
        if (nt >= NT) continue;
        const int c = 16 * nt + rl;                              // column of [dPs | dPt | dA2]
        float* blk = nt < 4 ? out : (nt < 8 ? out + rT : out + rL);
        const int cc = nt < 4 ? c : (nt < 8 ? c - D : c - 2 * D), stride = nt < 8 ? RS : RSL;
        const bool col_ok = nt < 8 || cc < NLP;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col_ok) blk[(1 + 16 * mt + 4 * g + e) * stride + cc] = acc[u][mt][e];
        if (g == 0) {                                            // row 64 of the A operand was the ones row: column sums
            const float cs = acc[u][4][0];
            if (col_ok) blk[cc] = cs;
            const float w0 = nt < 4 ? Wsrc[cc] : (nt < 8 ? Wtgt[cc] : (cc < nl ? Wlog[cc] : 0.f));
            at_share = fmaf(cs, w0, at_share);
        }
    }
    if (threadIdx.x < 2 * D) out[rG + threadIdx.x] = dgb;        // dgamma | dbeta are contiguous
    at_share = wave_sum(at_share);
    if (lane == 0) at_w[wave] = at_share;
    __syncthreads();
    if (threadIdx.x == 0) out[rA] = (at_w[0] + at_w[1]) + (at_w[2] + at_w[3]);
}

// k_theta = [Wsrc | Wtgt | Wlog | bf | bw | gamma | beta] (time rows scaled by t) and k_a_t from the block partials:
// 1 024 threads = 32 part-groups x 32 outputs, 16 loads per thread in flight together (512 partial rows: one round).
struct GatFinish {
    const float* part; int n_part, plen, d, nl, nlp, heads, out_len;
    float t; float* ktheta; float* kat;
    // n_slots > 0: the partials of n_slots stages (slot s at part + s * n_part * plen, evaluated at ts[s]) are closed in ONE
    // launch that ADDS sum_s w[s] * (stage derivative) to ktheta / kat - the solution combine of the small components of a
    // fixed-grid Runge-Kutta step (the adjoint ODE is linear in them and a fixed grid never looks at them in between)
    int n_slots; float ts[4]; float w[4];
};
__global__ __launch_bounds__(1024) void gat_small_finish_kernel(GatFinish g)
{
    __shared__ float sm[32][33];
    const int jj = threadIdx.x & 31, qq = threadIdx.x >> 5;
    const int j = (int)blockIdx.x * 32 + jj;
    const int d = g.d, nW = (d + 1) * d, nL = (d + 1) * g.nl;
    const int oBf = 2 * nW + nL, oBw = oBf + d, oGm = oBw + g.heads;     // theta offsets; gamma, beta are contiguous in both
    // partial rows are in the VJP kernel's LDS layout: two (d+1) x (d+1) blocks, one (d+1) x (NLP+1) block, dgamma, dbeta, a_t
    const int RS = d + 1, RSL = g.nlp + 1, rT = (d + 1) * RS, rL = 2 * rT, rG = rL + (d + 1) * RSL;
    int src = -1; float scale = 1.f; bool is_time_row = false;
    if (j < g.out_len) {
        if (j < 2 * nW) { const int b = j >= nW, jj = j - b * nW, r = jj / d; src = b * rT + r * RS + (jj - r * d); if (r == 0) { scale = g.t; is_time_row = true; } }
        else if (j < oBf) { const int jj = j - 2 * nW, r = jj / g.nl; src = rL + r * RSL + (jj - r * g.nl); if (r == 0) { scale = g.t; is_time_row = true; } }
        else if (j < oBw) src = rT + (j - oBf);                          // bf = colsum(dPt): the time row of the Wtgt block
        else if (j < oGm) src = rL + 2 * (j - oBw) + 1;                  // bw_h = colsum(dA2)[2h + 1]
        else src = rG + (j - oGm);                                       // dgamma | dbeta
    } else if (j == g.out_len) src = g.plen - 1;                         // a_t'
    const int n_rounds = g.n_slots > 0 ? g.n_slots : 1;
    float total = 0.f;
    for (int sl = 0; sl < n_rounds; ++sl) {
        const float* part = g.part + (int64_t)sl * g.n_part * g.plen;
        float v = 0.f;
        if (src >= 0) {
            for (int p0 = qq; p0 < g.n_part; p0 += 32 * 16) {       // 16 independent loads in flight per thread
                float x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int p = p0 + 32 * u;
                    x[u] = p < g.n_part ? part[(int64_t)p * g.plen + src] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) v += x[u];
            }
        }
        __syncthreads();                                             // the previous round's sums have been read
        sm[qq][jj] = v;
        __syncthreads();
        if (qq == 0 && src >= 0) {
            float tsum = sm[0][jj];
#pragma unroll
            for (int k = 1; k < 32; ++k) tsum += sm[k][jj];
            if (g.n_slots > 0) {
                const float sc = (j < g.out_len && is_time_row) ? g.ts[sl] : 1.f;
                total = fmaf(g.w[sl], sc * tsum, total);
            } else {
                total = (j < g.out_len ? scale : 1.f) * tsum;
            }
        }
    }
    if (qq == 0 && src >= 0) {
        if (g.n_slots > 0) {
            if (j < g.out_len) g.ktheta[j] += total; else *g.kat += total;
        } else {
            if (j < g.out_len) g.ktheta[j] = total; else *g.kat = total;
        }
    }
}

int gat_small_cg(int64_t d, int32_t groups) {
    if (d != 16 && d != 32 && d != 64) return -1;
    if (groups <= 0 || d % groups) return -1;
    const int64_t cg = d / groups;
    return (cg == 1 || cg == 2 || cg == 4) ? (int)cg : -1;
}

int64_t project_blocks(int64_t n) { int64_t b = (n + 3) / 4; if (b < 1) b = 1; if (b > 1024) b = 1024; return b; }

}  // namespace

extern "C" int gode_gat_small_supported(int64_t n_rows, int64_t d, int32_t groups, int64_t heads) {
    if (heads < 1) heads = 1;
    return n_rows > 0 && n_rows <= 65536 && heads <= 8 && d % heads == 0 && gat_small_cg(d, groups) > 0;
}
// a partial row is (2 d + 2 H + 2)(d + 1) floats: 2.4 KB at d = 16, 38 KB at d = 64, H = 8 - the wider the function, the
// fewer blocks.  Measured at d = 64 (Citeseer, 8 heads, training step): 64 / 128 / 256 / 512 blocks 11.2 / 10.2 / 9.8 /
// 11.0 ms (512 blocks write and re-read 19 MB per stage).
extern "C" int64_t gode_gat_small_parts(int64_t n_rows, int64_t d) {
    int64_t b = d == 64 ? (n_rows + 15) / 16 : (n_rows + 3) / 4;          // d = 64: a block per tile of 16 rows
    const int64_t cap = d <= 16 ? kGatPartBlocks : kGatPartBlocks / 2;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return b;
}
extern "C" int64_t gode_gat_small_part_len(int64_t d, int64_t heads) {          // the padded layout of the VJP kernel
    if (heads < 1) heads = 1;
    const int64_t nlp = 2 * heads <= 4 ? 4 : 16;
    return 2 * (d + 1) * (d + 1) + (d + 1) * (nlp + 1) + 2 * d + 1;
}
extern "C" int64_t gode_gat_small_pack_len(int64_t d, int64_t heads) {
    if (heads < 1) heads = 1;
    const int64_t nlp = 2 * heads <= 4 ? 4 : 16, ws = 2 * d + nlp;
    return (d + 1) * ws + ws * (d + 4);
}
extern "C" int gode_gat_small_pack_f32(const float* Wsrc, const float* Wtgt, const float* Wlog, int64_t d, int64_t heads,
                                       float* packed, void* stream) {
    if (heads < 1) heads = 1;
    if (!Wsrc || !Wtgt || !Wlog || !packed) return GODE_E_NULLPTR;
    if (d <= 0 || d % 16 || d > 64 || heads > 8) return GODE_E_UNSUPPORTED;
    const int nl = (int)(2 * heads), nlp = nl <= 4 ? 4 : 16;
    const int64_t total = gode_gat_small_pack_len(d, heads);
    hipLaunchKernelGGL(gat_small_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wsrc, Wtgt, Wlog,
                       (int)d, nl, nlp, packed);
    GODE_LAUNCH_CHECK();
    return 0;
}

#define GODE_GATS_DISPATCH(MACRO)                                                                                   \
    if (nlp == 4) {                                                                                                  \
        if (d == 16 && cg == 1) { MACRO(16, 1, 4) } else if (d == 16 && cg == 2) { MACRO(16, 2, 4) } else if (d == 16 && cg == 4) { MACRO(16, 4, 4) } \
        else if (d == 32 && cg == 1) { MACRO(32, 1, 4) } else if (d == 32 && cg == 2) { MACRO(32, 2, 4) } else if (d == 32 && cg == 4) { MACRO(32, 4, 4) } \
        else if (d == 64 && cg == 1) { MACRO(64, 1, 4) } else if (d == 64 && cg == 2) { MACRO(64, 2, 4) } else if (d == 64 && cg == 4) { MACRO(64, 4, 4) } \
    } else {                                                                                                         \
        if (d == 16 && cg == 1) { MACRO(16, 1, 16) } else if (d == 16 && cg == 2) { MACRO(16, 2, 16) } else if (d == 16 && cg == 4) { MACRO(16, 4, 16) } \
        else if (d == 32 && cg == 1) { MACRO(32, 1, 16) } else if (d == 32 && cg == 2) { MACRO(32, 2, 16) } else if (d == 32 && cg == 4) { MACRO(32, 4, 16) } \
        else if (d == 64 && cg == 1) { MACRO(64, 1, 16) } else if (d == 64 && cg == 2) { MACRO(64, 2, 16) } else if (d == 64 && cg == 4) { MACRO(64, 4, 16) } \
    }

extern "C" int gode_gat_project_small_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d, int32_t groups, float eps,
                                          const float* gamma, const float* beta, const float* Wsrc, const float* Wtgt,
                                          const float* Wlog, int64_t heads, const float* pt_bias, float t, float* Ps,
                                          float* Pt, float* A2, float* x_out, const float* packed, void* stream)
{
    if (heads < 1) heads = 1;
    if (!xin || !Wsrc || !Wtgt || !Wlog || !Ps || !Pt || !A2) return GODE_E_NULLPTR;
    if (!gode_gat_small_supported(n_rows, d, groups, heads)) return GODE_E_UNSUPPORTED;
    int rc = check_lincomb(xin, true); if (rc) return rc;
    if (!lincomb_aligned16(xin) || ((((uintptr_t)Ps) | ((uintptr_t)Pt) | ((uintptr_t)x_out) | ((uintptr_t)gamma) | ((uintptr_t)beta) |
                                     ((uintptr_t)pt_bias) | ((uintptr_t)packed)) & 15)) return GODE_E_ALIGN;
    const LinComb lx = make_lincomb(xin);
    const int cg = gat_small_cg(d, groups);
    const int nl = (int)(2 * heads), nlp = nl <= 4 ? 4 : 16;
    if (d == 64 && packed) {                                     // tiles of 16 rows on the matrix instruction
        int64_t tb = (n_rows + 15) / 16; if (tb > 1024) tb = 1024;
        const dim3 grid64((unsigned)tb);
#define GODE_GP64(CGV, NLV) hipLaunchKernelGGL((gat_project_d64_kernel<CGV, NLV>), grid64, dim3(256), 0, (hipStream_t)stream, \
                                               lx, (int)n_rows, eps, gamma, beta, nl, packed, pt_bias, t, Ps, Pt, A2, x_out);
        if (nlp == 4) { if (cg == 1) { GODE_GP64(1, 4) } else if (cg == 2) { GODE_GP64(2, 4) } else { GODE_GP64(4, 4) } }
        else { if (cg == 1) { GODE_GP64(1, 16) } else if (cg == 2) { GODE_GP64(2, 16) } else { GODE_GP64(4, 16) } }
#undef GODE_GP64
        GODE_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((unsigned)project_blocks(n_rows));
#define GODE_GPJ(DV, CGV, NLV) hipLaunchKernelGGL((gat_project_small_kernel<DV, CGV, NLV>), grid, dim3(256), 0, (hipStream_t)stream, \
                                                  lx, (int)n_rows, eps, gamma, beta, Wsrc, Wtgt, Wlog, nl, packed, pt_bias, t, Ps, Pt, A2, x_out);
    GODE_GATS_DISPATCH(GODE_GPJ)
#undef GODE_GPJ
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gat_dense_vjp_small_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d, int32_t groups, float eps,
                                            const float* gamma, const float* beta, const float* Wsrc, const float* Wtgt,
                                            const float* Wlog, int64_t heads, const float* dPs, const float* dPt,
                                            const float* dA2, float out_scale, const gode_lincomb_t* pre, float* ka,
                                            float* part, const void* maxpath_scratch, const int32_t* esrc,
                                            const int32_t* etgt, int64_t n_edges, const float* packed, void* stream)
{
    if (heads < 1) heads = 1;
    if (!xin || !Wsrc || !Wtgt || !Wlog || !dPs || !dPt || !dA2 || !ka || !part) return GODE_E_NULLPTR;
    if (!gode_gat_small_supported(n_rows, d, groups, heads)) return GODE_E_UNSUPPORTED;
    int rc = check_lincomb(xin, true); if (rc) return rc;
    if (pre && pre->n > 0) { rc = check_lincomb(pre, true); if (rc) return rc; } else pre = nullptr;
    if (!lincomb_aligned16(xin) || !lincomb_aligned16(pre) ||
        ((((uintptr_t)dPs) | ((uintptr_t)dPt) | ((uintptr_t)ka) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15)) return GODE_E_ALIGN;
    const LinComb lx = make_lincomb(xin), lp = make_lincomb(pre);
    const int cg = gat_small_cg(d, groups);
    const int nl = (int)(2 * heads), nlp = nl <= 4 ? 4 : 16;
    MaxFix mf = {nullptr, nullptr, 0, nullptr, nullptr, 0, (int)heads};
    if (maxpath_scratch && n_edges > 0) {
        if (!esrc || !etgt) return GODE_E_NULLPTR;
        if (n_edges > INT32_MAX) return GODE_E_RANGE;
        const int64_t cap = gode_gat_heads_block_cap();
        mf.psum = (const float*)maxpath_scratch; mf.pidx = (const int*)(mf.psum + cap * heads);
        mf.n_part = (int)gode_gat_heads_parts(n_edges); mf.esrc = esrc; mf.etgt = etgt; mf.n_edges = (int)n_edges;
    }
    if (((uintptr_t)packed) & 15) return GODE_E_ALIGN;
    const float* packed_t = packed ? packed + (d + 1) * (2 * d + nlp) : nullptr;      // the transposed image follows Wall
    const dim3 grid((unsigned)gode_gat_small_parts(n_rows, d));
    if (d == 64 && packed_t) {                                   // tiles of 16 rows on the matrix instruction
#define GODE_GV64(CGV, NLV) hipLaunchKernelGGL((gat_dense_vjp_d64_kernel<CGV, NLV>), grid, dim3(256), 0, (hipStream_t)stream, \
                                               lx, (int)n_rows, eps, gamma, beta, Wsrc, Wtgt, Wlog, nl, dPs, dPt, dA2, out_scale, lp, ka, part, mf, packed_t);
        if (nlp == 4) { if (cg == 1) { GODE_GV64(1, 4) } else if (cg == 2) { GODE_GV64(2, 4) } else { GODE_GV64(4, 4) } }
        else { if (cg == 1) { GODE_GV64(1, 16) } else if (cg == 2) { GODE_GV64(2, 16) } else { GODE_GV64(4, 16) } }
#undef GODE_GV64
        GODE_LAUNCH_CHECK();
        return 0;
    }
#define GODE_GVJ(DV, CGV, NLV) hipLaunchKernelGGL((gat_dense_vjp_small_kernel<DV, CGV, NLV>), grid, dim3(256), 0, (hipStream_t)stream, \
                                                  lx, (int)n_rows, eps, gamma, beta, Wsrc, Wtgt, Wlog, nl, dPs, dPt, dA2, out_scale, lp, ka, part, mf, packed_t);
    GODE_GATS_DISPATCH(GODE_GVJ)
#undef GODE_GVJ
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gat_small_finish_f32(const float* part, int64_t n_rows, int64_t d, int64_t heads, float t, float* ktheta,
                                         float* kat, void* stream)
{
    if (heads < 1) heads = 1;
    if (!part || !ktheta || !kat) return GODE_E_NULLPTR;
    if (n_rows <= 0 || n_rows > 65536 || d <= 0 || heads > 8) return GODE_E_SHAPE;
    GatFinish g;
    g.part = part; g.n_part = (int)gode_gat_small_parts(n_rows, d); g.plen = (int)gode_gat_small_part_len(d, heads);
    g.d = (int)d; g.nl = (int)(2 * heads); g.nlp = g.nl <= 4 ? 4 : 16; g.heads = (int)heads;
    g.out_len = (int)(2 * (d + 1) * d + (d + 1) * 2 * heads + d + heads + 2 * d);
    g.t = t; g.ktheta = ktheta; g.kat = kat; g.n_slots = 0;
    for (int q = 0; q < 4; ++q) { g.ts[q] = 0.f; g.w[q] = 0.f; }
    const int blocks = (g.out_len + 1 + 31) / 32;
    hipLaunchKernelGGL(gat_small_finish_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, g);
    GODE_LAUNCH_CHECK();
    return 0;
}

// The same for the stages of one fixed-grid Runge-Kutta step at once: theta += sum_s w[s] k_theta(stage s), a_t likewise,
// from n_slots <= 4 partial buffers laid out one after the other (each gode_gat_small_parts x gode_gat_small_part_len floats).
extern "C" int gode_gat_small_finish_step_f32(const float* part, int64_t n_rows, int64_t d, int64_t heads, int32_t n_slots,
                                              const float* ts /* host */, const float* w /* host */, float* theta, float* at,
                                              void* stream)
{
    if (heads < 1) heads = 1;
    if (!part || !theta || !at || !ts || !w) return GODE_E_NULLPTR;
    if (n_rows <= 0 || n_rows > 65536 || d <= 0 || heads > 8 || n_slots < 1 || n_slots > 4) return GODE_E_SHAPE;
    GatFinish g;
    g.part = part; g.n_part = (int)gode_gat_small_parts(n_rows, d); g.plen = (int)gode_gat_small_part_len(d, heads);
    g.d = (int)d; g.nl = (int)(2 * heads); g.nlp = g.nl <= 4 ? 4 : 16; g.heads = (int)heads;
    g.out_len = (int)(2 * (d + 1) * d + (d + 1) * 2 * heads + d + heads + 2 * d);
    g.t = 0.f; g.ktheta = theta; g.kat = at; g.n_slots = n_slots;
    for (int q = 0; q < 4; ++q) { g.ts[q] = q < n_slots ? ts[q] : 0.f; g.w[q] = q < n_slots ? w[q] : 0.f; }
    const int blocks = (g.out_len + 1 + 31) / 32;
    hipLaunchKernelGGL(gat_small_finish_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, g);
    GODE_LAUNCH_CHECK();
    return 0;
}
