// edge.hip — per-edge message kernels (gfx950):
//   * GAT-style edge attention (GAT/layers.py:40-55 of the reference): logits, global max,
//     exp, per-target normalised aggregation, and the matching backward;
//   * QC edge-conditioned messages (QC/mpnn.py:27-29, QC/layers.py:143-145): per-edge h x h
//     matvec gathered by source, summed per target, and the matching backward.
//
// The reference materialises h = [x[src] | x[tgt]] (E x 2i) and runs two Linear layers on it.
// Here the Linear layers are applied at NODE level by the caller (P = x * [Wf_src | Wf_tgt |
// ww_src | ww_tgt], one dense GEMM) and the kernels gather N x o rows per edge:
//   z_e = P[src_e, 0:o] + P[tgt_e, o:2o] + bf,  y_e = relu(z_e)
//   a_e = P[src_e, 2o] + P[tgt_e, 2o+1] + bw,   w_e = exp(a_e - max_e a_e)
//   out_v = sum_{e in row v of Mtgt} val * w_e * y_e / (sum val * w_e + eps)
// Edges of a target are visited in edge-id order (the order torch.spmm sums a coalesced Mtgt).
// Bound: HBM (launch-bound at Citeseer/QM9 sizes).  Algorithmic bytes per layer:
//   GAT: E*(2*4 idx + 4*o gather + 4 logit) + 2*N*o*4;  QC: E*h*h*4 + E*h*8 + N*h*8.
#include "common.h"

namespace {

// Node-level projections as the edge kernels see them (device copy of gode_gat_proj_t).
struct Proj {
    const float* ps; int64_t lds;     // message part gathered by source : ps[v*lds + c], c < o
    const float* pt; int64_t ldt;     // message part gathered by target
    const float* as; const float* at; int64_t lda;   // logit parts per node
};
Proj proj_of(const gode_gat_proj_t* p) { return Proj{p->ps, p->ld_s, p->pt, p->ld_t, p->as, p->at, p->ld_a}; }

// H-fold graphs (gat_heads.py): the maximum a virtual row's logits are shifted by is its HEAD's maximum (row % H).  The
// logits launch leaves per-block partial maxima (pmax[block * H + head]); the consumers reduce their head's partials
// themselves - one more load level inside a kernel instead of a launch that rewrites every logit (7.3 us x 129 per
// Citeseer step).  H == 0: one global maximum at amax[0].
struct HeadMax { const float* pmax; int n_part; int H; };
// `width` lanes (a power of two: the whole wave, or the lanes that share a row when a wave works on several rows) reduce
// together; `lane` is the lane's index inside its group.
__device__ __forceinline__ float shift_of(const HeadMax& hm, const float* amax, int row, int lane, int width = 64) {
    if (hm.H <= 0) return amax[0];
    const int h = row % hm.H;
    float m = -INFINITY;
    for (int b = lane; b < hm.n_part; b += width) m = fmaxf(m, hm.pmax[b * hm.H + h]);
    for (int off = width >> 1; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    return m;
}

__device__ __forceinline__ float block_max(float v) {
    __shared__ float sm[4];
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// a[e] = P[src,2o] + P[tgt,2o+1] + bw ; block maxima -> pmax[block]
__global__ __launch_bounds__(256) void gat_logits_kernel(Proj pv, const float* __restrict__ bw,
                                                         const int* __restrict__ src, const int* __restrict__ tgt,
                                                         int n_edges, float* __restrict__ a, float* __restrict__ pmax) {
    float m = -INFINITY;
    const float b = bw ? bw[0] : 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n_edges; e += gridDim.x * 256) {
        const float v = pv.as[(int64_t)src[e] * pv.lda] + pv.at[(int64_t)tgt[e] * pv.lda] + b;
        a[e] = v;
        m = fmaxf(m, v);
    }
    m = block_max(m);
    if (threadIdx.x == 0) pmax[blockIdx.x] = m;
}

__global__ __launch_bounds__(256) void final_max_kernel(const float* pmax, int n, float* out) {
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, pmax[i]);
    m = block_max(m);
    if (threadIdx.x == 0) out[0] = m;
}

// One group of G lanes (power of two <= 64) per target row; lane handles columns c, c+G, ...
// up to MAXC columns per lane (o <= G*MAXC).
template <int MAXC>
__global__ __launch_bounds__(256) void gat_agg_fwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                          const float* __restrict__ val,
                                                          const int* __restrict__ src, const int* __restrict__ tgt,
                                                          Proj pv, int o,
                                                          const float* __restrict__ bf, const float* __restrict__ a,
                                                          const float* __restrict__ amax, float eps, int n_rows, int G,
                                                          float* __restrict__ out, float* __restrict__ w_out,
                                                          float* __restrict__ s_out) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int v = (int)(tid / G);
    const int lane = threadIdx.x & (G - 1);
    if (v >= n_rows) return;
    const float m = amax[0];
    float acc[MAXC], bias[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) { acc[q] = 0.f; const int c = lane + q * G; bias[q] = (bf && c < o) ? bf[c] : 0.f; }
    float s = 0.f;
    for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
        const int e = eid ? eid[k] : k;
        const float w = expf(a[e] - m);
        const float we = (val ? val[k] : 1.f) * w;
        if (lane == 0) w_out[e] = w;
        s += we;
        const float* ps = pv.ps + (int64_t)src[e] * pv.lds;
        const float* pt = pv.pt + (int64_t)tgt[e] * pv.ldt;
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            const int c = lane + q * G;
            if (c < o) {
                const float z = ps[c] + pt[c] + bias[q];
                acc[q] = fmaf(we, fmaxf(z, 0.f), acc[q]);
            }
        }
    }
    const float den = s + eps;
    if (lane == 0) s_out[v] = den;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
        const int c = lane + q * G;
        if (c < o) out[(int64_t)v * o + c] = acc[q] / den;
    }
}

// backward: per target row v (group of G lanes)
//   dA = dout_v / den_v ; dsum = -(dout_v . out_v) / den_v
//   per edge: dy = val*w*dA ; dz = dy*(z>0) -> dz_out[e,:] ; dw = val*(dA.y + dsum) ; da[e] = dw*w
//   dPtgt[v] (+)= sum_e dz  is NOT formed here (tgt[e] may be any node): dz is scattered by the caller
//   with two SpMM launches over the src / tgt incidence matrices.
template <int MAXC>
__global__ __launch_bounds__(256) void gat_agg_bwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                          const float* __restrict__ val,
                                                          const int* __restrict__ src, const int* __restrict__ tgt,
                                                          Proj pv, int o,
                                                          const float* __restrict__ bf, const float* __restrict__ w,
                                                          const float* __restrict__ den, const float* __restrict__ out,
                                                          const float* __restrict__ dout, int n_rows, int G,
                                                          float* __restrict__ dz, float* __restrict__ da) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int v = (int)(tid / G);
    const int lane = threadIdx.x & (G - 1);
    if (v >= n_rows) return;
    const float dn = den[v];
    float dA[MAXC], bias[MAXC];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
        const int c = lane + q * G;
        dA[q] = 0.f; bias[q] = 0.f;
        if (c < o) {
            const float g = dout[(int64_t)v * o + c];
            dA[q] = g / dn;
            dot += g * out[(int64_t)v * o + c];
            bias[q] = bf ? bf[c] : 0.f;
        }
    }
    for (int off = G >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    const float dsum = -dot / dn;
    for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
        const int e = eid ? eid[k] : k;
        const float we = w[e];
        const float vv = val ? val[k] : 1.f;
        const float* ps = pv.ps + (int64_t)src[e] * pv.lds;
        const float* pt = pv.pt + (int64_t)tgt[e] * pv.ldt;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            const int c = lane + q * G;
            if (c < o) {
                const float z = ps[c] + pt[c] + bias[q];
                const float y = fmaxf(z, 0.f);
                part += dA[q] * y;
                dz[(int64_t)e * o + c] = z > 0.f ? vv * we * dA[q] : 0.f;
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) da[e] = vv * (part + dsum) * we;
    }
}


// ---- small-graph variants: a whole wave per target row ------------------------------------------
// 64/G sub-groups of G lanes stride through the row's edges (a 99-edge hub of Citeseer takes 25 trips instead of
// 99) and are combined with xor-shuffles; used below 65 536 rows, where one lane group per row leaves the chip idle
// behind the longest row.
template <int MAXC>
__global__ __launch_bounds__(256) void gat_agg_fwd_wave_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                               const float* __restrict__ val,
                                                               const int* __restrict__ src, const int* __restrict__ tgt,
                                                               Proj pv, int o,
                                                               const float* __restrict__ bf, const float* __restrict__ a,
                                                               const float* __restrict__ amax, HeadMax hm, float eps, int n_rows, int G,
                                                               float* __restrict__ out, float* __restrict__ w_out,
                                                               float* __restrict__ s_out) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_rows) return;
    const int l = threadIdx.x & 63, lane = l & (G - 1), sg = l / G, ns = 64 / G;
    const float m = shift_of(hm, amax, v, l);
    float acc[MAXC], bias[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) { acc[q] = 0.f; const int c = lane + q * G; bias[q] = (bf && c < o) ? bf[c] : 0.f; }
    float s = 0.f;
    const int end = rowptr[v + 1];
    for (int k = rowptr[v] + sg; k < end; k += ns) {
        const int e = eid ? eid[k] : k;
        const float w = expf(a[e] - m);
        const float we = (val ? val[k] : 1.f) * w;
        if (lane == 0) w_out[e] = w;
        s += we;
        const float* ps = pv.ps + (int64_t)src[e] * pv.lds;
        const float* pt = pv.pt + (int64_t)tgt[e] * pv.ldt;
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            const int c = lane + q * G;
            if (c < o) acc[q] = fmaf(we, fmaxf(ps[c] + pt[c] + bias[q], 0.f), acc[q]);
        }
    }
    for (int off = G; off < 64; off <<= 1) {
        s += __shfl_xor(s, off, 64);
#pragma unroll
        for (int q = 0; q < MAXC; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
    }
    const float den = s + eps;
    if (l == 0) s_out[v] = den;
    if (sg == 0) {
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            const int c = lane + q * G;
            if (c < o) out[(int64_t)v * o + c] = acc[q] / den;
        }
    }
}

// Backward, wave per row.  The cotangent is either dout[v, c] or, when cot.n > 0, cot_scale * (sum_j cot_j[v, c])
// masked by out[v, c] > 0 (the relu that follows the layer inside the ODE function): the solver's stage cotangent
// is combined and masked here instead of in three elementwise launches.
template <int MAXC>
__global__ __launch_bounds__(256) void gat_agg_bwd_wave_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                               const float* __restrict__ val,
                                                               const int* __restrict__ src, const int* __restrict__ tgt,
                                                               Proj pv, int o,
                                                               const float* __restrict__ bf, const float* __restrict__ w,
                                                               const float* __restrict__ den, const float* __restrict__ out,
                                                               const float* __restrict__ dout, LinComb cot, float cot_scale,
                                                               int n_rows, int G,
                                                               float* __restrict__ dz, float* __restrict__ da) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_rows) return;
    const int l = threadIdx.x & 63, lane = l & (G - 1), sg = l / G, ns = 64 / G;
    const float dn = den[v];
    float dA[MAXC], bias[MAXC];
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
        const int c = lane + q * G;
        dA[q] = 0.f; bias[q] = 0.f;
        if (c < o) {
            const float ov = out[(int64_t)v * o + c];
            float g;
            if (cot.n > 0) g = ov > 0.f ? cot_scale * lc_load1(cot, (int64_t)v * o + c) : 0.f;
            else g = dout[(int64_t)v * o + c];
            dA[q] = g / dn;
            dot += g * ov;
            bias[q] = bf ? bf[c] : 0.f;
        }
    }
    for (int off = G >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    const float dsum = -dot / dn;
    const int end = rowptr[v + 1];
    for (int k = rowptr[v] + sg; k < end; k += ns) {
        const int e = eid ? eid[k] : k;
        const float we = w[e];
        const float vv = val ? val[k] : 1.f;
        const float* ps = pv.ps + (int64_t)src[e] * pv.lds;
        const float* pt = pv.pt + (int64_t)tgt[e] * pv.ldt;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            const int c = lane + q * G;
            if (c < o) {
                const float z = ps[c] + pt[c] + bias[q];
                part += dA[q] * fmaxf(z, 0.f);
                dz[(int64_t)e * o + c] = z > 0.f ? vv * we * dA[q] : 0.f;
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) da[e] = vv * (part + dsum) * we;
    }
}

// ---- small-graph variants with PREFETCHED INDICES ----------------------------------------------------------------
// (o in {16, 32, 64}.)  The wave kernels above walk a row's edges a few at a time, and every trip is a chain of dependent loads
// (eid[k] -> src[e], a[e] -> Ps[src]): Citeseer's 99-edge hub takes 25 trips of ~0.3 us and sets the kernel's duration
// (8-10 us for 12 k edges).  Here the 64 lanes first fetch the indices and per-edge scalars of 64 CSR slots at once
// (two load levels for the whole chunk), then LPR = o/4 lanes per edge (16 B each) gather the chunk's projection rows
// with the indices taken from registers (__shfl): the gathers of a chunk are independent of each other and issue back
// to back, so a hub costs about as many round trips as a leaf.
__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// GW = lanes that share a row: 64 (a wave per row) or 16 (four rows per wave, for narrow rows: eight heads of 8 columns on
// the H-fold graph are 26 616 virtual rows of ~4 edges - a wave per row left three quarters of the prefetch lanes idle).  A
// chunk is GW CSR slots; NS = GW / LPR edges are gathered per trip.
template <int LPR, int GW>
__global__ __launch_bounds__(256) void gat_agg_fwd_pf_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                             const float* __restrict__ val,
                                                             const int* __restrict__ src, const int* __restrict__ tgt,
                                                             Proj pv, const float* __restrict__ bf, const float* __restrict__ a,
                                                             const float* __restrict__ amax, HeadMax hm, float eps, int n_rows,
                                                             float* __restrict__ out, float* __restrict__ w_out,
                                                             float* __restrict__ s_out) {
    constexpr int O = 4 * LPR, NS = GW / LPR, U = LPR < 4 ? LPR : 4, RPW = 64 / GW;
    const int l = threadIdx.x & 63, gl = l & (GW - 1), q = gl & (LPR - 1), sg = gl / LPR;
    const int v = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + l / GW;
    if (v >= n_rows) return;                                   // whole groups leave together
    const int b = rowptr[v], end = rowptr[v + 1];
    const float m = shift_of(hm, amax, v, gl, GW);
    const float4 bias = bf ? ldf4(bf + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cv = ldf4(pv.pt + (int64_t)v * pv.ldt + 4 * q);
    cv.x += bias.x; cv.y += bias.y; cv.z += bias.z; cv.w += bias.w;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    for (int base = b; base < end; base += GW) {
        const int k = base + gl;
        int sc = 0, tg = v; float we = 0.f;
        if (k < end) {
            const int e = eid ? eid[k] : k;
            sc = src[e]; tg = tgt[e];
            const float w = expf(a[e] - m);
            w_out[e] = w;
            we = (val ? val[k] : 1.f) * w;
        }
        const int cnt = end - base;
        for (int u0 = 0; u0 < LPR; u0 += U) {
            if (u0 * NS >= cnt) break;
            float4 xv[U], cc[U]; float wj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = (u0 + u) * NS + sg;              // < GW; slots beyond the row carry weight 0 and row 0
                const int sj = __shfl(sc, j, GW), tj = __shfl(tg, j, GW);
                wj[u] = __shfl(we, j, GW);
                xv[u] = ldf4(pv.ps + (int64_t)sj * pv.lds + 4 * q);
                cc[u] = cv;
                if (tj != v) {                                 // an aggregation matrix that disagrees with tgt (never from the reference's loaders)
                    cc[u] = ldf4(pv.pt + (int64_t)tj * pv.ldt + 4 * q);
                    cc[u].x += bias.x; cc[u].y += bias.y; cc[u].z += bias.z; cc[u].w += bias.w;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s += wj[u];
                acc.x = fmaf(wj[u], fmaxf(xv[u].x + cc[u].x, 0.f), acc.x); acc.y = fmaf(wj[u], fmaxf(xv[u].y + cc[u].y, 0.f), acc.y);
                acc.z = fmaf(wj[u], fmaxf(xv[u].z + cc[u].z, 0.f), acc.z); acc.w = fmaf(wj[u], fmaxf(xv[u].w + cc[u].w, 0.f), acc.w);
            }
        }
    }
#pragma unroll
    for (int off = LPR; off < GW; off <<= 1) {
        s += __shfl_xor(s, off, 64);
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    const float den = s + eps;
    if (gl == 0) s_out[v] = den;
    if (sg == 0) *reinterpret_cast<float4*>(out + (int64_t)v * O + 4 * q) = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
}

template <int LPR, int GW>
__global__ __launch_bounds__(256) void gat_agg_bwd_pf_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                             const float* __restrict__ val,
                                                             const int* __restrict__ src, const int* __restrict__ tgt,
                                                             Proj pv, const float* __restrict__ bf, const float* __restrict__ w,
                                                             const float* __restrict__ den, const float* __restrict__ out,
                                                             const float* __restrict__ dout, LinComb cot, float cot_scale,
                                                             int n_rows, float* __restrict__ dz, float* __restrict__ da) {
    constexpr int O = 4 * LPR, NS = GW / LPR, U = LPR < 4 ? LPR : 4, RPW = 64 / GW;
    const int l = threadIdx.x & 63, gl = l & (GW - 1), q = gl & (LPR - 1), sg = gl / LPR;
    const int v = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + l / GW;
    if (v >= n_rows) return;
    const int b = rowptr[v], end = rowptr[v + 1];
    const float dn = den[v];
    const int64_t ro = (int64_t)v * O + 4 * q;
    const float4 ov = ldf4(out + ro);
    float4 g;
    if (cot.n > 0) {
        const float4 c = lc_load4(cot, ro);
        g = make_float4(ov.x > 0.f ? cot_scale * c.x : 0.f, ov.y > 0.f ? cot_scale * c.y : 0.f,
                        ov.z > 0.f ? cot_scale * c.z : 0.f, ov.w > 0.f ? cot_scale * c.w : 0.f);
    } else g = ldf4(dout + ro);
    const float4 dA = make_float4(g.x / dn, g.y / dn, g.z / dn, g.w / dn);
    float dot = (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    const float dsum = -dot / dn;
    const float4 bias = bf ? ldf4(bf + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cv = ldf4(pv.pt + (int64_t)v * pv.ldt + 4 * q);
    cv.x += bias.x; cv.y += bias.y; cv.z += bias.z; cv.w += bias.w;
    for (int base = b; base < end; base += GW) {
        const int k = base + gl;
        int e = 0, sc = 0, tg = v; float we = 0.f, vv = 0.f;
        if (k < end) { e = eid ? eid[k] : k; sc = src[e]; tg = tgt[e]; we = w[e]; vv = val ? val[k] : 1.f; }
        const int cnt = end - base;
        for (int u0 = 0; u0 < LPR; u0 += U) {
            if (u0 * NS >= cnt) break;
            float4 xv[U], cc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = (u0 + u) * NS + sg;
                const int sj = __shfl(sc, j, GW), tj = __shfl(tg, j, GW);
                xv[u] = ldf4(pv.ps + (int64_t)sj * pv.lds + 4 * q);
                cc[u] = cv;
                if (tj != v) {
                    cc[u] = ldf4(pv.pt + (int64_t)tj * pv.ldt + 4 * q);
                    cc[u].x += bias.x; cc[u].y += bias.y; cc[u].z += bias.z; cc[u].w += bias.w;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = (u0 + u) * NS + sg;
                const int ej = __shfl(e, j, GW);
                const float wej = __shfl(we, j, GW), vj = __shfl(vv, j, GW);
                const float4 z = make_float4(xv[u].x + cc[u].x, xv[u].y + cc[u].y, xv[u].z + cc[u].z, xv[u].w + cc[u].w);
                float part = (dA.x * fmaxf(z.x, 0.f) + dA.y * fmaxf(z.y, 0.f)) + (dA.z * fmaxf(z.z, 0.f) + dA.w * fmaxf(z.w, 0.f));
#pragma unroll
                for (int off = LPR >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
                if (j < cnt) {
                    const float sc2 = vj * wej;
                    *reinterpret_cast<float4*>(dz + (int64_t)ej * O + 4 * q) =
                        make_float4(z.x > 0.f ? sc2 * dA.x : 0.f, z.y > 0.f ? sc2 * dA.y : 0.f,
                                    z.z > 0.f ? sc2 * dA.z : 0.f, z.w > 0.f ? sc2 * dA.w : 0.f);
                    if (q == 0) da[ej] = vj * (part + dsum) * wej;
                }
            }
        }
    }
}

// both incidence sums of the VJP (over the edges leaving a node and over those entering it), prefetched the same way
template <int LPR, int GW>
__global__ __launch_bounds__(256) void gat_scatter_pf_kernel(const int* __restrict__ rp_s, const int* __restrict__ e_s,
                                                             const int* __restrict__ rp_t, const int* __restrict__ e_t,
                                                             const float* __restrict__ dz, const float* __restrict__ da,
                                                             int n_rows, float* __restrict__ dps, int64_t lds,
                                                             float* __restrict__ dpt, int64_t ldt, float* __restrict__ das,
                                                             float* __restrict__ dat, int64_t lda) {
    constexpr int O = 4 * LPR, NS = GW / LPR, U = LPR < 4 ? LPR : 4, RPW = 64 / GW;
    const int l = threadIdx.x & 63, gl = l & (GW - 1), q = gl & (LPR - 1), sg = gl / LPR;
    const int v = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + l / GW;
    if (v >= n_rows) return;
    const int bs = rp_s[v], es = rp_s[v + 1], bt = rp_t[v], et = rp_t[v + 1];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int* ee = side == 0 ? e_s : e_t;
        const int b = side == 0 ? bs : bt, end = side == 0 ? es : et;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float sa = 0.f;
        for (int base = b; base < end; base += GW) {
            const int k = base + gl;
            int e = -1;
            if (k < end) { e = ee[k]; sa += da[e]; }
            const int cnt = end - base;
            for (int u0 = 0; u0 < LPR; u0 += U) {
                if (u0 * NS >= cnt) break;
                float4 xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ej = __shfl(e, (u0 + u) * NS + sg, GW);
                    xv[u] = ej >= 0 ? ldf4(dz + (int64_t)ej * O + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { acc.x += xv[u].x; acc.y += xv[u].y; acc.z += xv[u].z; acc.w += xv[u].w; }
            }
        }
#pragma unroll
        for (int off = 1; off < GW; off <<= 1) sa += __shfl_xor(sa, off, 64);
#pragma unroll
        for (int off = LPR; off < GW; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
        }
        float* dp = side == 0 ? dps + (int64_t)v * lds : dpt + (int64_t)v * ldt;
        if (sg == 0) *reinterpret_cast<float4*>(dp + 4 * q) = acc;
        if (gl == 0) (side == 0 ? das : dat)[(int64_t)v * lda] = sa;
    }
}

// ---- large-graph variants: nnz-balanced records, as in the SpMM (csrc/spmm.hip) ---------------------------------
// Edges are in target-sorted order (the caller passes eid == NULL: edge k IS position k of the CSR), so every
// per-edge array (a, w, dz, da, src) is read and written as a stream and the only gather is Ps[src_k]: one
// coalesced o*4-byte row per edge with 16 B per lane.  A group of LPR = o/4 lanes owns one record {row, begin, end,
// slot}; rows longer than the graph's split length are cut into records that write raw partial sums into the slab
// (row stride o+4: o numerator columns + the denominator) and are finished in slot order by a second small kernel.
template <int LPR>
__global__ __launch_bounds__(256) void gat_agg_fwd_rec_kernel(const int4* __restrict__ items, int n_items,
                                                              const int* __restrict__ src, const float* __restrict__ val,
                                                              Proj pv, const float* __restrict__ bf,
                                                              const float* __restrict__ a, const float* __restrict__ amax,
                                                              float eps, float* __restrict__ out, float* __restrict__ w_out,
                                                              float* __restrict__ den_out, float* __restrict__ partial) {
    constexpr int U = 4, O = LPR * 4;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / LPR);
    const int lane = threadIdx.x & (LPR - 1);
    const bool active = gid < n_items;
    int row = 0, b = 0, e = 0, slot = -1;
    if (active) { const int4 it = items[gid]; row = it.x; b = it.y; e = it.z; slot = it.w; }
    const int len = e - b;
    const float m = amax[0];
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        c = *reinterpret_cast<const float4*>(pv.pt + (int64_t)row * pv.ldt + lane * 4);
        if (bf) { const float4 q = *reinterpret_cast<const float4*>(bf + lane * 4); c.x += q.x; c.y += q.y; c.z += q.z; c.w += q.w; }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    const float* Xl = pv.ps + lane * 4;
    for (int base = 0; __any(base < len); base += LPR) {
        int sc = 0; float we = 0.f;
        if (base + lane < len) {
            const int k = b + base + lane;
            sc = src[k];
            const float w = expf(a[k] - m);
            w_out[k] = w;
            we = (val ? val[k] : 1.f) * w;
        }
        const int cnt = len - base;
        for (int k = 0; k < LPR; k += U) {
            if (!__any(k < cnt)) break;
            int cc[U]; float vv[U]; float4 xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { cc[u] = __shfl(sc, k + u, LPR); vv[u] = __shfl(we, k + u, LPR); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + u < cnt && k + u < LPR) xv[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)cc[u] * pv.lds);
                else vv[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s += vv[u];
                acc.x = fmaf(vv[u], fmaxf(xv[u].x + c.x, 0.f), acc.x); acc.y = fmaf(vv[u], fmaxf(xv[u].y + c.y, 0.f), acc.y);
                acc.z = fmaf(vv[u], fmaxf(xv[u].z + c.z, 0.f), acc.z); acc.w = fmaf(vv[u], fmaxf(xv[u].w + c.w, 0.f), acc.w);
            }
        }
    }
    if (!active) return;
    if (slot < 0) {
        const float den = s + eps;
        if (lane == 0) den_out[row] = den;
        *reinterpret_cast<float4*>(out + (int64_t)row * O + lane * 4) = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
    } else {
        float* p = partial + (int64_t)slot * (O + 4);
        *reinterpret_cast<float4*>(p + lane * 4) = acc;
        if (lane == 0) *reinterpret_cast<float4*>(p + O) = make_float4(s, 0.f, 0.f, 0.f);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void gat_agg_fwd_finish_kernel(const int4* __restrict__ long_rows, int n_long,
                                                                 const float* __restrict__ partial, float eps,
                                                                 float* __restrict__ out, float* __restrict__ den_out) {
    constexpr int O = LPR * 4;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / LPR);
    const int lane = threadIdx.x & (LPR - 1);
    if (gid >= n_long) return;
    const int4 lr = long_rows[gid];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    for (int q = lr.y; q < lr.z; ++q) {
        const float* p = partial + (int64_t)q * (O + 4);
        const float4 v = *reinterpret_cast<const float4*>(p + lane * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        s += p[O];
    }
    const float den = s + eps;
    if (lane == 0) den_out[lr.x] = den;
    *reinterpret_cast<float4*>(out + (int64_t)lr.x * O + lane * 4) = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
}

// Backward over records.  Besides dz[k,:] and da[k] it accumulates the target-side sums dpt[v,:] = sum_k dz[k,:]
// and dat[v] = sum_k da[k] of its record (rows of edge cotangents are contiguous per target in this order), so the
// caller needs incidence products only for the source side.
template <int LPR>
__global__ __launch_bounds__(256) void gat_agg_bwd_rec_kernel(const int4* __restrict__ items, int n_items,
                                                              const int* __restrict__ src, const float* __restrict__ val,
                                                              Proj pv, const float* __restrict__ bf,
                                                              const float* __restrict__ w, const float* __restrict__ den,
                                                              const float* __restrict__ out, const float* __restrict__ dout,
                                                              LinComb cot, float cot_scale,
                                                              float* __restrict__ dz, float* __restrict__ da,
                                                              float* __restrict__ dpt, int64_t ld_dpt, float* __restrict__ dat,
                                                              int64_t ld_dat, float* __restrict__ partial) {
    constexpr int U = 4, O = LPR * 4;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / LPR);
    const int lane = threadIdx.x & (LPR - 1);
    const bool active = gid < n_items;
    int row = 0, b = 0, e = 0, slot = -1;
    if (active) { const int4 it = items[gid]; row = it.x; b = it.y; e = it.z; slot = it.w; }
    const int len = e - b;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f), dA = c;
    float dsum = 0.f;
    if (active) {
        c = *reinterpret_cast<const float4*>(pv.pt + (int64_t)row * pv.ldt + lane * 4);
        if (bf) { const float4 q = *reinterpret_cast<const float4*>(bf + lane * 4); c.x += q.x; c.y += q.y; c.z += q.z; c.w += q.w; }
        const float dn = den[row];
        const float4 ov = *reinterpret_cast<const float4*>(out + (int64_t)row * O + lane * 4);
        float4 g;
        if (cot.n > 0) {
            g = lc_load4(cot, (int64_t)row * O + lane * 4);
            g.x = ov.x > 0.f ? cot_scale * g.x : 0.f; g.y = ov.y > 0.f ? cot_scale * g.y : 0.f;
            g.z = ov.z > 0.f ? cot_scale * g.z : 0.f; g.w = ov.w > 0.f ? cot_scale * g.w : 0.f;
        } else {
            g = *reinterpret_cast<const float4*>(dout + (int64_t)row * O + lane * 4);
        }
        float dot = (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
        for (int off = LPR >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, LPR);
        dA = make_float4(g.x / dn, g.y / dn, g.z / dn, g.w / dn);
        dsum = -dot / dn;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float sda = 0.f;
    const float* Xl = pv.ps + lane * 4;
    for (int base = 0; __any(base < len); base += LPR) {
        int sc = 0; float we = 0.f, vk = 0.f;
        if (base + lane < len) {
            const int k = b + base + lane;
            sc = src[k];
            vk = val ? val[k] : 1.f;
            we = w[k];
        }
        const int cnt = len - base;
        float da_lane = 0.f;                       // da of edge base + lane, filled below
        for (int k = 0; k < LPR; k += U) {
            if (!__any(k < cnt)) break;
            int cc[U]; float ww[U], vv[U]; float4 xv[U]; float part[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { cc[u] = __shfl(sc, k + u, LPR); ww[u] = __shfl(we, k + u, LPR); vv[u] = __shfl(vk, k + u, LPR); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + u < cnt && k + u < LPR) xv[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)cc[u] * pv.lds);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = k + u < cnt && k + u < LPR;
                const float4 z = make_float4(xv[u].x + c.x, xv[u].y + c.y, xv[u].z + c.z, xv[u].w + c.w);
                part[u] = (dA.x * fmaxf(z.x, 0.f) + dA.y * fmaxf(z.y, 0.f)) + (dA.z * fmaxf(z.z, 0.f) + dA.w * fmaxf(z.w, 0.f));
                const float f = vv[u] * ww[u];
                const float4 t = make_float4(z.x > 0.f ? f * dA.x : 0.f, z.y > 0.f ? f * dA.y : 0.f,
                                             z.z > 0.f ? f * dA.z : 0.f, z.w > 0.f ? f * dA.w : 0.f);
                if (live) {
                    *reinterpret_cast<float4*>(dz + (int64_t)(b + base + k + u) * O + lane * 4) = t;
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                }
            }
            for (int off = LPR >> 1; off > 0; off >>= 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) part[u] += __shfl_xor(part[u], off, LPR);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (lane == k + u) da_lane = vv[u] * (part[u] + dsum) * ww[u];
        }
        if (base + lane < len) { da[b + base + lane] = da_lane; sda += da_lane; }
    }
    for (int off = LPR >> 1; off > 0; off >>= 1) sda += __shfl_xor(sda, off, LPR);
    if (!active) return;
    if (slot < 0) {
        *reinterpret_cast<float4*>(dpt + (int64_t)row * ld_dpt + lane * 4) = acc;
        if (lane == 0) dat[(int64_t)row * ld_dat] = sda;
    } else {
        float* p = partial + (int64_t)slot * (O + 4);
        *reinterpret_cast<float4*>(p + lane * 4) = acc;
        if (lane == 0) *reinterpret_cast<float4*>(p + O) = make_float4(sda, 0.f, 0.f, 0.f);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void gat_agg_bwd_finish_kernel(const int4* __restrict__ long_rows, int n_long,
                                                                 const float* __restrict__ partial,
                                                                 float* __restrict__ dpt, int64_t ld_dpt,
                                                                 float* __restrict__ dat, int64_t ld_dat) {
    constexpr int O = LPR * 4;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / LPR);
    const int lane = threadIdx.x & (LPR - 1);
    if (gid >= n_long) return;
    const int4 lr = long_rows[gid];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    for (int q = lr.y; q < lr.z; ++q) {
        const float* p = partial + (int64_t)q * (O + 4);
        const float4 v = *reinterpret_cast<const float4*>(p + lane * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        s += p[O];
    }
    *reinterpret_cast<float4*>(dpt + (int64_t)lr.x * ld_dpt + lane * 4) = acc;
    if (lane == 0) dat[(int64_t)lr.x * ld_dat] = s;
}

// Two-stage max-path correction for large edge lists: block sums / first arg-max per block, then one block combines
// them in block order and applies  da[e*] -= S  (and, when given, dat[tgt_row(e*)] -= S for sums already formed).
__global__ __launch_bounds__(256) void gat_maxpath_part_kernel(const float* __restrict__ a, const float* __restrict__ amax,
                                                               const float* __restrict__ da, int n_edges,
                                                               float* __restrict__ psum, int* __restrict__ pidx) {
    __shared__ float ssum[4];
    __shared__ int sidx[4];
    const float m = amax[0];
    float s = 0.f;
    int first = INT32_MAX;
    const int per = (n_edges + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n_edges, lo + per);
    for (int e = lo + threadIdx.x; e < hi; e += 256) {
        s += da[e];
        if (a[e] == m && e < first) first = e;
    }
    s = wave_sum(s);
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; sidx[threadIdx.x >> 6] = first; }
    __syncthreads();
    if (threadIdx.x == 0) {
        psum[blockIdx.x] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        pidx[blockIdx.x] = min(min(sidx[0], sidx[1]), min(sidx[2], sidx[3]));
    }
}
__global__ void gat_maxpath_final_kernel(const float* __restrict__ psum, const int* __restrict__ pidx, int n_part,
                                         float* __restrict__ da, int n_edges, const int* __restrict__ tgt,
                                         float* __restrict__ dat, int64_t ld_dat) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float t = 0.f; int f = INT32_MAX;
    for (int j = 0; j < n_part; ++j) { t += psum[j]; f = min(f, pidx[j]); }
    if (f < n_edges) {
        da[f] -= t;
        if (dat && tgt) dat[(int64_t)tgt[f] * ld_dat] -= t;
    }
}

// One block: logits of all edges and their maximum in a single launch (E <= 32 768).
__global__ __launch_bounds__(1024) void gat_logits_small_kernel(Proj pv, const float* __restrict__ bw,
                                                                const int* __restrict__ src, const int* __restrict__ tgt,
                                                                int n_edges, float* __restrict__ a, float* __restrict__ amax) {
    __shared__ float sm[16];
    float m = -INFINITY;
    const float b = bw ? bw[0] : 0.f;
    for (int e = threadIdx.x; e < n_edges; e += 1024) {
        const float v = pv.as[(int64_t)src[e] * pv.lda] + pv.at[(int64_t)tgt[e] * pv.lda] + b;
        a[e] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = sm[0];
        for (int j = 1; j < 16; ++j) r = fmaxf(r, sm[j]);
        amax[0] = r;
    }
}

// The path of the gradient through the global maximum (GAT/layers.py:47: a - max(a)): da[e*] -= sum_e da[e] with e* the
// first edge attaining the maximum.  One block; fixed summation order.
__global__ __launch_bounds__(1024) void gat_maxpath_kernel(const float* __restrict__ a, const float* __restrict__ amax,
                                                           float* __restrict__ da, int n_edges) {
    __shared__ float ssum[16];
    __shared__ int sidx[16];
    const float m = amax[0];
    float s = 0.f;
    int first = INT32_MAX;
    for (int e = threadIdx.x; e < n_edges; e += 1024) {
        s += da[e];
        if (a[e] == m && e < first) first = e;
    }
    s = wave_sum(s);
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; sidx[threadIdx.x >> 6] = first; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f; int f = INT32_MAX;
        for (int j = 0; j < 16; ++j) { t += ssum[j]; f = min(f, sidx[j]); }
        if (f < n_edges) da[f] -= t;
    }
}

// Edge cotangents back to the node-level projections: a wave per node sums dz / da over the edges leaving it
// (source incidence) and over the edges entering it (target incidence) - the four incidence products of the VJP
// in one launch.
template <int MAXC>
__global__ __launch_bounds__(256) void gat_scatter_kernel(const int* __restrict__ rp_s, const int* __restrict__ e_s,
                                                          const int* __restrict__ rp_t, const int* __restrict__ e_t,
                                                          const float* __restrict__ dz, const float* __restrict__ da,
                                                          int o, int n_rows, int G,
                                                          float* __restrict__ dps, int64_t lds, float* __restrict__ dpt,
                                                          int64_t ldt, float* __restrict__ das, float* __restrict__ dat,
                                                          int64_t lda) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_rows) return;
    const int l = threadIdx.x & 63, lane = l & (G - 1), sg = l / G, ns = 64 / G;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int* rp = side == 0 ? rp_s : rp_t;
        const int* ee = side == 0 ? e_s : e_t;
        float acc[MAXC], sa = 0.f;
#pragma unroll
        for (int q = 0; q < MAXC; ++q) acc[q] = 0.f;
        const int end = rp[v + 1];
        for (int k = rp[v] + sg; k < end; k += ns) {
            const int e = ee[k];
            sa += da[e];
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                const int c = lane + q * G;
                if (c < o) acc[q] += dz[(int64_t)e * o + c];
            }
        }
        for (int off = G; off < 64; off <<= 1) {
            sa += __shfl_xor(sa, off, 64);
#pragma unroll
            for (int q = 0; q < MAXC; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
        }
        float* dp = side == 0 ? dps + (int64_t)v * lds : dpt + (int64_t)v * ldt;
        if (sg == 0) {
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                const int c = lane + q * G;
                if (c < o) dp[c] = acc[q];
            }
        }
        if (l == 0) (side == 0 ? das : dat)[(int64_t)v * lda] = sa;
    }
}

// a_t' and the time row of a weight gradient: at (+)= <g_row0, W_row0>, g_row0 *= t  (one tiny block).
__global__ __launch_bounds__(256) void time_row_fixup_kernel(float* __restrict__ g_row0, const float* __restrict__ w_row0,
                                                             int len, float t, float* __restrict__ at, int accumulate) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int c = threadIdx.x; c < len; c += 256) s += g_row0[c] * w_row0[c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]); at[0] = accumulate ? at[0] + r : r; }
    for (int c = threadIdx.x; c < len; c += 256) g_row0[c] *= t;
}

// ---------------- QC edge-conditioned messages ------------------------------------------------
// block per target row v: M_v = sum_{(e,val) in row v} val * A_e * x[src_e]
// An edge matrix goes through LDS in tiles of whole rows, read from memory as ONE flat run with every load of a thread in
// flight (h = 73: the whole 21 KB matrix, 21 loads per thread), the next tile's loads (and the next edge's index, value
// and source row) requested into registers BEFORE the current tile is multiplied; thread r then sums row r out of LDS.
// (The first version gave a wave four matrix rows per trip: five dependent trips per edge with 4-8 loads in flight, then
// a 24-step shuffle reduction - 37 us per launch for a 760-edge mini-batch whose matrices are 16 MB: a latency chain.)
constexpr int kMsgTile = 8192;        // floats of LDS per tile (32 KB)
// <row, x> over h floats of LDS, eight elements per trip: sixteen independent LDS reads are requested before the first
// multiply-add (two elements per trip made a trip one LDS round trip: 37 trips of ~150 cycles per matrix row)
__device__ __forceinline__ float lds_dot(const float* __restrict__ row, const float* __restrict__ x, int h) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int j = 0;
    for (; j + 8 <= h; j += 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = row[j + u]; b[u] = x[j + u]; }
        s0 = fmaf(a[0], b[0], s0); s1 = fmaf(a[1], b[1], s1); s2 = fmaf(a[2], b[2], s2); s3 = fmaf(a[3], b[3], s3);
        s0 = fmaf(a[4], b[4], s0); s1 = fmaf(a[5], b[5], s1); s2 = fmaf(a[6], b[6], s2); s3 = fmaf(a[7], b[7], s3);
    }
    for (; j < h; ++j) s0 = fmaf(row[j], x[j], s0);
    return (s0 + s1) + (s2 + s3);
}
// XP = registers of a prefetched source row: 1 (h <= 256) or 16 (h <= 4096).  The index triples (edge id, value, source
// atom) of up to 256 of the row's edges are fetched together into LDS before the first matrix is requested, and matrix
// tiles are requested TWO steps ahead (two register sets): per edge the block then waits for at most one load round
// trip instead of three dependent ones (edge id -> source atom -> rows), which was 30 of the launch's 37 us.
template <int XP>
__global__ __launch_bounds__(256) void edge_matvec_fwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                              const float* __restrict__ val, const int* __restrict__ src,
                                                              const float* __restrict__ A, const float* __restrict__ X,
                                                              int64_t ldx, int h, float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PF = kMsgTile / 256;                            // registers of a prefetched tile
    __shared__ int es[256], ss[256];
    __shared__ float vs[256];
    const int hp = (h + 3) & ~3;
    float* xs = smem;              // [h]
    float* macc = smem + hp;       // [h]
    float* tile = smem + 2 * hp;   // [rows per tile][h]
    const int v = blockIdx.x, tid = threadIdx.x;
    const int rb = rowptr[v], ne = rowptr[v + 1] - rb;
    const int rpt = kMsgTile / h > 0 ? kMsgTile / h : 1, ntiles = (h + rpt - 1) / rpt;
    for (int i = tid; i < h; i += 256) macc[i] = 0.f;
    float pa[PF], pb[PF], xa[XP], xb[XP];
    for (int base = 0; base < ne; base += 256) {                  // (a row with more than 256 edges: another round)
        const int cnt = min(256, ne - base), T = cnt * ntiles;
        __syncthreads();                                           // the previous round's index triples are done with
        if (tid < cnt) {
            const int k = rb + base + tid, e = eid ? eid[k] : k;
            es[tid] = e; vs[tid] = val ? val[k] : 1.f; ss[tid] = src[e];
        }
        __syncthreads();
        auto issue = [&](int it, float (&pre)[PF], float (&xpre)[XP]) {
            const int q = it / ntiles, ti = it - q * ntiles, e = es[q];
            const int i0 = ti * rpt, run = min(rpt, h - i0) * h;
            const float* Ar = A + (int64_t)e * h * h + (int64_t)i0 * h;
#pragma unroll
            for (int u = 0; u < PF; ++u) { const int p = tid + 256 * u; pre[u] = Ar[p < run ? p : run - 1]; }   // unconditional, clamped:
            {                                                                                              // no load under a branch
                const float* xr = X + (int64_t)ss[q] * ldx;
#pragma unroll
                for (int u = 0; u < XP; ++u) { const int j = tid + 256 * u; xpre[u] = xr[j < h ? j : h - 1]; }
            }
        };
        float vv = 0.f;
        auto step = [&](int it, float (&pre)[PF], float (&xpre)[XP]) {
            const int q = it / ntiles, ti = it - q * ntiles, i0 = ti * rpt, rows = min(rpt, h - i0), run = rows * h;
            __syncthreads();                                       // the previous tile (and macc's zeros) are done with
#pragma unroll
            for (int u = 0; u < PF; ++u) { const int p = tid + 256 * u; if (p < run) tile[p] = pre[u]; }
            if (ti == 0) {
                vv = vs[q];
#pragma unroll
                for (int u = 0; u < XP; ++u) { const int j = tid + 256 * u; if (j < h) xs[j] = xpre[u]; }
            }
            __syncthreads();
            issue(it + 2 < T ? it + 2 : T - 1, pre, xpre);         // this register set is free again: two steps ahead (past the end: the last tile again)
            for (int r = tid; r < rows; r += 256)
                macc[i0 + r] += vv * lds_dot(tile + r * h, xs, h);  // row i0 + r belongs to this thread in every tile of every edge
        };
        issue(0, pa, xa);
        if (T > 1) issue(1, pb, xb);
        for (int it = 0; it < T; it += 2) {
            step(it, pa, xa);
            if (it + 1 < T) step(it + 1, pb, xb);
        }
    }
    __syncthreads();
    for (int i = tid; i < h; i += 256) out[(int64_t)v * ldo + i] = macc[i];
}

// block per edge e: msg[e, :] = A_e * x[src_e]  (large batches: every edge matrix is streamed by its own workgroup and
// the per-target sum is a separate SpMM over Etgt; the fused per-target kernel above serialises a target's edges).
// The matrix goes through LDS in tiles of whole rows, read from memory as one flat run (all loads of a tile in flight,
// full-width accesses whatever h is); thread i then sums row i out of LDS - no cross-lane reduction.
__global__ __launch_bounds__(256) void edge_matvec_msg_kernel(const int* __restrict__ src, const float* __restrict__ A,
                                                              const float* __restrict__ X, int64_t ldx, int h,
                                                              float* __restrict__ msg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [h]
    float* tile = smem + h;           // [rows_per_tile * h]
    const int e = blockIdx.x;
    for (int j = threadIdx.x; j < h; j += 256) xs[j] = X[(int64_t)src[e] * ldx + j];
    const float* Ae = A + (int64_t)e * h * h;
    const int rpt = kMsgTile / h > 0 ? kMsgTile / h : 1;
    for (int i0 = 0; i0 < h; i0 += rpt) {
        const int rows = min(rpt, h - i0), run = rows * h;
        __syncthreads();                                       // previous tile consumed (and xs written)
        const float* Ar = Ae + (int64_t)i0 * h;
        for (int p = threadIdx.x; p < run; p += 256) tile[p] = Ar[p];
        __syncthreads();
        for (int r = threadIdx.x; r < rows; r += 256) msg[(int64_t)e * h + i0 + r] = lds_dot(tile + r * h, xs, h);
    }
}

// block per edge e: dm = val_e * dM[tgt_e]; dA_e = dm (x) x[src_e]; dxe[e] = A_e^T dm
__global__ __launch_bounds__(256) void edge_matvec_bwd_kernel(const int* __restrict__ erow, const float* __restrict__ eval,
                                                              const int* __restrict__ src, const float* __restrict__ A,
                                                              const float* __restrict__ X, int64_t ldx,
                                                              const float* __restrict__ dM, int64_t ldm, int h,
                                                              float* __restrict__ dA, float* __restrict__ dxe) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;        // [h]
    float* dm = smem + h;    // [h]
    const int e = blockIdx.x;
    const int row = erow[e];
    const float vv = eval ? eval[e] : 1.f;
    for (int j = threadIdx.x; j < h; j += 256) {
        xs[j] = X[(int64_t)src[e] * ldx + j];
        dm[j] = row >= 0 ? vv * dM[(int64_t)row * ldm + j] : 0.f;
    }
    __syncthreads();
    const int64_t base = (int64_t)e * h * h;
    if (dA) {
        for (int idx = threadIdx.x; idx < h * h; idx += 256) {
            const int i = idx / h, j = idx - i * h;
            dA[base + idx] = dm[i] * xs[j];
        }
    }
    if (dxe) {
        for (int j = threadIdx.x; j < h; j += 256) {
            // eight matrix rows per trip, their loads requested together (one row per trip made 73 dependent round trips)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float* Aj = A + base + j;
            int i = 0;
            for (; i + 8 <= h; i += 8) {
                float a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = Aj[(int64_t)(i + u) * h];
                s0 = fmaf(a[0], dm[i], s0); s1 = fmaf(a[1], dm[i + 1], s1); s2 = fmaf(a[2], dm[i + 2], s2); s3 = fmaf(a[3], dm[i + 3], s3);
                s0 = fmaf(a[4], dm[i + 4], s0); s1 = fmaf(a[5], dm[i + 5], s1); s2 = fmaf(a[6], dm[i + 6], s2); s3 = fmaf(a[7], dm[i + 7], s3);
            }
            for (; i < h; ++i) s0 = fmaf(Aj[(int64_t)i * h], dm[i], s0);
            dxe[(int64_t)e * h + j] = (s0 + s1) + (s2 + s3);
        }
    }
}

// block per edge e: dA_e = sum_t (val_e dM_t[tgt_e]) (x) X_t[src_e] over the n_terms message steps that used the SAME edge
// matrices (QC/mpnn.py:27-30 runs T steps on one edge_data; autograd would write T arrays of E h^2 floats and add them in
// T - 1 more passes: here the sum is formed once, 4 E h^2 bytes written in total)
struct OuterTerms { int n; const float* dM[GODE_MAX_TERMS]; const float* X[GODE_MAX_TERMS]; };
__global__ __launch_bounds__(256) void edge_outer_sum_kernel(const int* __restrict__ erow, const float* __restrict__ eval,
                                                             const int* __restrict__ src, OuterTerms tm, int64_t ldx,
                                                             int64_t ldm, int h, float* __restrict__ dA) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                    // [n][h]
    float* dm = smem + tm.n * h;         // [n][h]
    const int e = blockIdx.x;
    const int row = erow[e], sc = src[e];
    const float vv = eval ? eval[e] : 1.f;
    for (int idx = threadIdx.x; idx < tm.n * h; idx += 256) {
        const int t = idx / h, j = idx - t * h;
        xs[idx] = tm.X[t][(int64_t)sc * ldx + j];
        dm[idx] = row >= 0 ? vv * tm.dM[t][(int64_t)row * ldm + j] : 0.f;
    }
    __syncthreads();
    const int64_t base = (int64_t)e * h * h;
    for (int idx = threadIdx.x; idx < h * h; idx += 256) {
        const int i = idx / h, j = idx - i * h;
        float v = 0.f;
        for (int t = 0; t < tm.n; ++t) v = fmaf(dm[t * h + i], xs[t * h + j], v);
        dA[base + idx] = v;
    }
}

// ---- H independent attention heads on one graph ------------------------------------------------------------------
// The heads run as ONE head on the H-fold graph: virtual node v*H + h carries head h of node v (an N x H*o projection
// matrix IS the (N*H) x o matrix of the virtual nodes), edge (s -> t) becomes the H edges (s*H+h -> t*H+h), and every
// aggregation kernel above runs unchanged.  What does not carry over is the GLOBAL maximum of GAT/layers.py:47, which
// is per head: the logits are shifted by their head's maximum here (so each head's maximum is exactly 0 and the
// aggregation kernels are given amax = 0), and the gradient path through the maximum is applied per head.
// Head of an edge = tgt % H.  Deterministic: maxima are order-free, sums run in a fixed order.
constexpr int kMaxHeads = 64;
constexpr int kHeadBlocks = 512;

__device__ __forceinline__ int float_key(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(256) void gat_logits_heads_part_kernel(Proj pv, const float* __restrict__ bw,
                                                                    const int* __restrict__ src,
                                                                    const int* __restrict__ tgt, int n_edges, int H,
                                                                    float* __restrict__ a, float* __restrict__ pmax) {
    __shared__ int hm[kMaxHeads];
    if (threadIdx.x < H) hm[threadIdx.x] = float_key(-INFINITY);
    __syncthreads();
    const int per = (n_edges + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n_edges, lo + per);
    for (int e = lo + threadIdx.x; e < hi; e += 256) {
        const int t = tgt[e];
        const float v = pv.as[(int64_t)src[e] * pv.lda] + pv.at[(int64_t)t * pv.lda] + (bw ? bw[t % H] : 0.f);
        a[e] = v;
        atomicMax(&hm[t % H], float_key(v));
    }
    __syncthreads();
    if (threadIdx.x < H) pmax[blockIdx.x * H + threadIdx.x] = key_float(hm[threadIdx.x]);
}

// Both steps in one block (small edge lists): every thread revisits the entries it wrote.
__global__ __launch_bounds__(1024) void gat_logits_heads_small_kernel(Proj pv, const float* __restrict__ bw,
                                                                      const int* __restrict__ src,
                                                                      const int* __restrict__ tgt, int n_edges, int H,
                                                                      float* __restrict__ a, float* __restrict__ hmax) {
    __shared__ int hm[kMaxHeads];
    if (threadIdx.x < H) hm[threadIdx.x] = float_key(-INFINITY);
    __syncthreads();
    for (int e = threadIdx.x; e < n_edges; e += 1024) {
        const int t = tgt[e];
        const float v = pv.as[(int64_t)src[e] * pv.lda] + pv.at[(int64_t)t * pv.lda] + (bw ? bw[t % H] : 0.f);
        a[e] = v;
        atomicMax(&hm[t % H], float_key(v));
    }
    __syncthreads();
    if (threadIdx.x < H && hmax) hmax[threadIdx.x] = key_float(hm[threadIdx.x]);
    for (int e = threadIdx.x; e < n_edges; e += 1024) a[e] -= key_float(hm[tgt[e] % H]);
}

// One block (small edge lists, H <= 8 at the call site): per-thread LDS columns per head, fixed-order sums, first arg-max per head.
__global__ __launch_bounds__(1024) void gat_maxpath_heads_small_kernel(const float* __restrict__ a, float* __restrict__ da,
                                                                       const int* __restrict__ tgt, int n_edges, int H) {
    extern __shared__ float col[];                    // [H][1024]
    __shared__ int hidx[16];
    __shared__ float hsum[16];
    for (int h = 0; h < H; ++h) col[h * 1024 + threadIdx.x] = 0.f;
    if (threadIdx.x < H) hidx[threadIdx.x] = INT32_MAX;
    __syncthreads();
    for (int e = threadIdx.x; e < n_edges; e += 1024) {
        const int h = tgt[e] % H;
        col[h * 1024 + threadIdx.x] += da[e];
        if (a[e] == 0.f) atomicMin(&hidx[h], e);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int h = w; h < H; h += 16) {
        const float* c = col + h * 1024;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += c[lane + 64 * k];
        v = wave_sum(v);
        if (lane == 0) hsum[h] = v;
    }
    __syncthreads();
    if (threadIdx.x < H && hidx[threadIdx.x] < n_edges) da[hidx[threadIdx.x]] -= hsum[threadIdx.x];
}

__global__ __launch_bounds__(256) void gat_logits_heads_shift_kernel(const int* __restrict__ tgt, int n_edges, int H,
                                                                     int n_part, const float* __restrict__ pmax,
                                                                     float* __restrict__ a, float* __restrict__ hmax) {
    __shared__ float hm[kMaxHeads];
    if (threadIdx.x < H) {
        float m = -INFINITY;
        for (int b = 0; b < n_part; ++b) m = fmaxf(m, pmax[b * H + threadIdx.x]);
        hm[threadIdx.x] = m;
        if (blockIdx.x == 0 && hmax) hmax[threadIdx.x] = m;
    }
    __syncthreads();
    const int per = (n_edges + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n_edges, lo + per);
    for (int e = lo + threadIdx.x; e < hi; e += 256) a[e] -= hm[tgt[e] % H];
}

// per block and head: sum of da and first edge whose shifted logit is 0 (the head's arg-max)
__global__ __launch_bounds__(256) void gat_maxpath_heads_part_kernel(const float* __restrict__ a, const float* __restrict__ da,
                                                                     const int* __restrict__ tgt, int n_edges, int H,
                                                                     float* __restrict__ psum, int* __restrict__ pidx,
                                                                     const float* __restrict__ pmax, int n_pmax) {
    extern __shared__ float col[];                    // [H][256]: thread t accumulates its own column
    __shared__ int hidx[kMaxHeads];
    __shared__ float hmx[kMaxHeads];                  // the value an arg-max logit has: 0 (shifted logits) or the head's maximum (raw)
    for (int h = 0; h < H; ++h) col[h * 256 + threadIdx.x] = 0.f;
    if (threadIdx.x < H) {
        hidx[threadIdx.x] = INT32_MAX;
        float m = 0.f;
        if (pmax) { m = -INFINITY; for (int b = 0; b < n_pmax; ++b) m = fmaxf(m, pmax[b * H + threadIdx.x]); }
        hmx[threadIdx.x] = m;
    }
    __syncthreads();
    const int per = (n_edges + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n_edges, lo + per);
    for (int e = lo + threadIdx.x; e < hi; e += 256) {
        const int h = tgt[e] % H;
        col[h * 256 + threadIdx.x] += da[e];
        if (a[e] == hmx[h]) atomicMin(&hidx[h], e);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int h = w; h < H; h += 4) {
        const float* c = col + h * 256;
        const float v = wave_sum((c[lane] + c[lane + 64]) + (c[lane + 128] + c[lane + 192]));
        if (lane == 0) { psum[blockIdx.x * H + h] = v; pidx[blockIdx.x * H + h] = hidx[h]; }
    }
}

__global__ __launch_bounds__(64) void gat_maxpath_heads_final_kernel(const float* __restrict__ psum, const int* __restrict__ pidx,
                                                                     int n_part, int H, float* __restrict__ da, int n_edges,
                                                                     const int* __restrict__ tgt, float* __restrict__ dat,
                                                                     int64_t ld_dat) {
    const int h = threadIdx.x;
    if (h >= H) return;
    float t = 0.f; int f = INT32_MAX;
    for (int b = 0; b < n_part; ++b) { t += psum[b * H + h]; f = min(f, pidx[b * H + h]); }
    if (f < n_edges) {
        da[f] -= t;
        if (dat) dat[(int64_t)tgt[f] * ld_dat] -= t;
    }
}

int pow2_group(int o) { int G = 1; while (G < o && G < 64) G <<= 1; return G; }

}  // namespace

extern "C" int64_t gode_gat_logits_scratch_bytes(int64_t n_edges) {
    int64_t b = (n_edges + 255) / 256; if (b > 1024) b = 1024; if (b < 1) b = 1;
    return b * (int64_t)sizeof(float);
}

namespace {

constexpr int64_t kWaveRows = 65536;      // below this a whole wave works on one target row
constexpr int64_t kOneBlockEdges = 32768; // below this logits + maximum are one launch

int check_proj(const Proj& p, int64_t o) {
    if (!p.ps || !p.pt || !p.as || !p.at) return GODE_E_NULLPTR;
    if (p.lds < o || p.ldt < o || p.lda < 1) return GODE_E_SHAPE;
    return 0;
}

// the prefetched-index kernels: float4 lanes over o in {4, 8, 16, 32, 64} (4 and 8: four rows per wave), 16-byte aligned rows
// (o = 8 - eight heads of 8 on the H-fold graph - runs four rows per wave, GW = 16: with a wave per row two lanes per edge left
// the prefetch stage mostly idle and the plain wave kernels were faster, 9.2 / 10.5 us against 12.0 / 18.9 us on Citeseer)
bool pf_ok(const Proj& pv, int64_t o, const void* p0, const void* p1) {
    if (o != 4 && o != 8 && o != 16 && o != 32 && o != 64) return false;
    if ((pv.lds % 4) || (pv.ldt % 4)) return false;
    const uintptr_t al = (uintptr_t)pv.ps | (uintptr_t)pv.pt | (uintptr_t)p0 | (uintptr_t)p1;
    return !(al & 15);
}

int launch_logits(const Proj& pv, const float* bw, const int32_t* src, const int32_t* tgt, int64_t n_edges,
                  float* a, float* amax, float* scratch, hipStream_t s) {
    if (n_edges <= kOneBlockEdges) {
        hipLaunchKernelGGL(gat_logits_small_kernel, dim3(1), dim3(1024), 0, s, pv, bw, src, tgt, (int)n_edges, a, amax);
        GODE_LAUNCH_CHECK();
        return 0;
    }
    int64_t b = (n_edges + 255) / 256; if (b > 1024) b = 1024; if (b < 1) b = 1;
    hipLaunchKernelGGL(gat_logits_kernel, dim3((unsigned)b), dim3(256), 0, s, pv, bw, src, tgt, (int)n_edges, a, scratch);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, s, (const float*)scratch, (int)b, amax);
    GODE_LAUNCH_CHECK();
    return 0;
}

int launch_agg_fwd(const int32_t* rowptr, const int32_t* eid, const float* val, const int32_t* src, const int32_t* tgt,
                   const Proj& pv, int64_t o, const float* bf, const float* a, const float* amax, float eps,
                   int64_t n_rows, float* out, float* w_out, float* den_out, hipStream_t s, HeadMax hm = HeadMax{nullptr, 0, 0}) {
    const int G = pow2_group((int)o);
    const int maxc = (int)((o + G - 1) / G);
    const bool wave = n_rows <= kWaveRows;
    if (hm.H > 0 && !wave) return GODE_E_UNSUPPORTED;          // per-head partial maxima: wave kernels only
    const int64_t blocks = wave ? (n_rows + 3) / 4 : (n_rows * G + 255) / 256;
    if (wave && pf_ok(pv, o, out, bf)) {
#define GODE_PF(L, W) hipLaunchKernelGGL((gat_agg_fwd_pf_kernel<L, W>), dim3((unsigned)((n_rows + 4 * (64 / W) - 1) / (4 * (64 / W)))), dim3(256), 0, s, \
                                         rowptr, eid, val, src, tgt, pv, bf, a, amax, hm, eps, (int)n_rows, out, w_out, den_out)
        switch ((int)o / 4) { case 1: GODE_PF(1, 16); break; case 2: GODE_PF(2, 16); break; case 4: GODE_PF(4, 64); break; case 8: GODE_PF(8, 64); break; default: GODE_PF(16, 64); break; }
#undef GODE_PF
        GODE_LAUNCH_CHECK();
        return 0;
    }
#define GODE_AGG(M)                                                                                              \
    do {                                                                                                         \
        if (wave) hipLaunchKernelGGL(gat_agg_fwd_wave_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, eid, val, \
                                     src, tgt, pv, (int)o, bf, a, amax, hm, eps, (int)n_rows, G, out, w_out, den_out); \
        else hipLaunchKernelGGL(gat_agg_fwd_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, eid, val, \
                                src, tgt, pv, (int)o, bf, a, amax, eps, (int)n_rows, G, out, w_out, den_out);     \
    } while (0)
    if (maxc <= 1) GODE_AGG(1); else if (maxc <= 2) GODE_AGG(2); else if (maxc <= 4) GODE_AGG(4); else GODE_AGG(8);
#undef GODE_AGG
    GODE_LAUNCH_CHECK();
    return 0;
}

int launch_agg_bwd(const int32_t* rowptr, const int32_t* eid, const float* val, const int32_t* src, const int32_t* tgt,
                   const Proj& pv, int64_t o, const float* bf, const float* w, const float* den, const float* out,
                   const float* dout, const LinComb& cot, float cot_scale, int64_t n_rows, float* dz, float* da,
                   hipStream_t s) {
    const int G = pow2_group((int)o);
    const int maxc = (int)((o + G - 1) / G);
    const bool wave = n_rows <= kWaveRows || cot.n > 0;      // the cotangent-combining form exists as wave kernel only
    const int64_t blocks = wave ? (n_rows + 3) / 4 : (n_rows * G + 255) / 256;
    bool cot_al = true;
    for (int j = 0; j < cot.n; ++j) cot_al = cot_al && !(((uintptr_t)cot.ptr[j]) & 15);
    if (wave && n_rows <= kWaveRows && cot_al && pf_ok(pv, o, out, bf) && !((((uintptr_t)dz) | ((uintptr_t)dout)) & 15)) {
#define GODE_PF(L, W) hipLaunchKernelGGL((gat_agg_bwd_pf_kernel<L, W>), dim3((unsigned)((n_rows + 4 * (64 / W) - 1) / (4 * (64 / W)))), dim3(256), 0, s, \
                                         rowptr, eid, val, src, tgt, pv, bf, w, den, out, dout, cot, cot_scale, (int)n_rows, dz, da)
        switch ((int)o / 4) { case 1: GODE_PF(1, 16); break; case 2: GODE_PF(2, 16); break; case 4: GODE_PF(4, 64); break; case 8: GODE_PF(8, 64); break; default: GODE_PF(16, 64); break; }
#undef GODE_PF
        GODE_LAUNCH_CHECK();
        return 0;
    }
#define GODE_AGG(M)                                                                                              \
    do {                                                                                                         \
        if (wave) hipLaunchKernelGGL(gat_agg_bwd_wave_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, eid, val, \
                                     src, tgt, pv, (int)o, bf, w, den, out, dout, cot, cot_scale, (int)n_rows, G, dz, da); \
        else hipLaunchKernelGGL(gat_agg_bwd_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, eid, val, \
                                src, tgt, pv, (int)o, bf, w, den, out, dout, (int)n_rows, G, dz, da);             \
    } while (0)
    if (maxc <= 1) GODE_AGG(1); else if (maxc <= 2) GODE_AGG(2); else if (maxc <= 4) GODE_AGG(4); else GODE_AGG(8);
#undef GODE_AGG
    GODE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ---- the same three steps over separately stored projections, plus the fused pieces of the ODE-function VJP --------
extern "C" int gode_gat_logits_f32(const gode_gat_proj_t* proj, const float* bw, const int32_t* src, const int32_t* tgt,
                                   int64_t n_edges, float* a, float* amax, float* scratch, void* stream) {
    if (!proj || !amax || !scratch) return GODE_E_NULLPTR;
    if (n_edges < 0) return GODE_E_SHAPE;
    if (n_edges > INT32_MAX) return GODE_E_RANGE;
    const Proj pv = proj_of(proj);
    if (n_edges > 0) { int rc = check_proj(pv, 1); if (rc) return rc; if (!src || !tgt || !a) return GODE_E_NULLPTR; }
    return launch_logits(pv, bw, src, tgt, n_edges, a, amax, scratch, (hipStream_t)stream);
}

namespace {
bool rec_ok(const gode_graph_t* mt, const Proj& pv, int64_t o, const void* p0, const void* p1) {
    if (!mt->items || mt->col || mt->n_rows <= kWaveRows) return false;      // records + target-sorted edges only
    if (o % 4 || o < 16 || o > 256 || (o & (o - 1))) return false;
    if (mt->n_long > 0 && !mt->partial) return false;
    if ((pv.lds % 4) || (pv.ldt % 4)) return false;
    const uintptr_t al = (uintptr_t)pv.ps | (uintptr_t)pv.pt | (uintptr_t)p0 | (uintptr_t)p1;
    return !(al & 15);
}
}  // namespace

extern "C" int gode_gat_agg_f32_fwd(const gode_graph_t* mt, const int32_t* src, const int32_t* tgt,
                                    const gode_gat_proj_t* proj, int64_t o, const float* bf, const float* a,
                                    const float* amax, float eps, float* out, float* w_out, float* den_out, void* stream) {
    if (!mt) return GODE_E_NULLPTR;
    const int64_t n_rows = mt->n_rows;
    if (n_rows < 0 || o <= 0) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!proj || !mt->rowptr || !amax || !out || !den_out) return GODE_E_NULLPTR;
    if (mt->nnz > 0 && (!src || !tgt || !a || !w_out)) return GODE_E_NULLPTR;        // an edgeless graph has no edge arrays
    if (n_rows > INT32_MAX || o > 512) return (o > 512) ? GODE_E_UNSUPPORTED : GODE_E_RANGE;
    const Proj pv = proj_of(proj);
    int rc = check_proj(pv, o); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (rec_ok(mt, pv, o, out, bf ? (const void*)bf : (const void*)out)) {
        const int lpr = (int)(o / 4);
        const int64_t blocks = (mt->n_items * lpr + 255) / 256;
        const int64_t fblocks = (mt->n_long * lpr + 255) / 256;
#define GODE_REC(L)                                                                                                   \
        do {                                                                                                          \
            hipLaunchKernelGGL(gat_agg_fwd_rec_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, s, (const int4*)mt->items, \
                               (int)mt->n_items, src, mt->val, pv, bf, a, amax, eps, out, w_out, den_out, mt->partial); \
            if (mt->n_long > 0)                                                                                       \
                hipLaunchKernelGGL(gat_agg_fwd_finish_kernel<L>, dim3((unsigned)fblocks), dim3(256), 0, s,            \
                                   (const int4*)mt->long_rows, (int)mt->n_long, (const float*)mt->partial, eps, out, den_out); \
        } while (0)
        switch (lpr) { case 4: GODE_REC(4); break; case 8: GODE_REC(8); break; case 16: GODE_REC(16); break;
                       case 32: GODE_REC(32); break; default: GODE_REC(64); break; }
#undef GODE_REC
        GODE_LAUNCH_CHECK();
        return 0;
    }
    return launch_agg_fwd(mt->rowptr, mt->col, mt->val, src, tgt, pv, o, bf, a, amax, eps, n_rows, out, w_out, den_out, s);
}

extern "C" int gode_gat_agg_f32_bwd(const gode_graph_t* mt, const int32_t* src, const int32_t* tgt,
                                    const gode_gat_proj_t* proj, int64_t o, const float* bf, const float* w,
                                    const float* den, const float* out, const float* dout, const gode_lincomb_t* cot,
                                    float cot_scale, float* dz, float* da, float* dpt, int64_t ld_dpt, float* dat,
                                    int64_t ld_dat, int32_t* did_target_sums, void* stream) {
    if (!mt) return GODE_E_NULLPTR;
    const int64_t n_rows = mt->n_rows;
    if (did_target_sums) *did_target_sums = 0;
    if (n_rows < 0 || o <= 0) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!proj || !mt->rowptr || !den || !out) return GODE_E_NULLPTR;
    if (mt->nnz > 0 && (!src || !tgt || !w || !dz || !da)) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || o > 512) return (o > 512) ? GODE_E_UNSUPPORTED : GODE_E_RANGE;
    const bool use_cot = cot && cot->n > 0;
    if (use_cot) { int rc = check_lincomb(cot, true); if (rc) return rc; }
    else if (!dout) return GODE_E_NULLPTR;
    const Proj pv = proj_of(proj);
    int rc = check_proj(pv, o); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const LinComb lc = make_lincomb(use_cot ? cot : nullptr);
    if (dpt && dat && did_target_sums && ld_dpt >= o && !(ld_dpt % 4) && ld_dat >= 1 && !(((uintptr_t)dpt | (uintptr_t)dz) & 15) &&
        (!use_cot || lincomb_aligned16(cot)) && rec_ok(mt, pv, o, out, use_cot ? (const void*)out : (const void*)dout)) {
        const int lpr = (int)(o / 4);
        const int64_t blocks = (mt->n_items * lpr + 255) / 256;
        const int64_t fblocks = (mt->n_long * lpr + 255) / 256;
#define GODE_REC(L)                                                                                                   \
        do {                                                                                                          \
            hipLaunchKernelGGL(gat_agg_bwd_rec_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, s, (const int4*)mt->items, \
                               (int)mt->n_items, src, mt->val, pv, bf, w, den, out, dout, lc, cot_scale, dz, da, dpt,  \
                               ld_dpt, dat, ld_dat, mt->partial);                                                     \
            if (mt->n_long > 0)                                                                                       \
                hipLaunchKernelGGL(gat_agg_bwd_finish_kernel<L>, dim3((unsigned)fblocks), dim3(256), 0, s,            \
                                   (const int4*)mt->long_rows, (int)mt->n_long, (const float*)mt->partial, dpt, ld_dpt, \
                                   dat, ld_dat);                                                                      \
        } while (0)
        switch (lpr) { case 4: GODE_REC(4); break; case 8: GODE_REC(8); break; case 16: GODE_REC(16); break;
                       case 32: GODE_REC(32); break; default: GODE_REC(64); break; }
#undef GODE_REC
        GODE_LAUNCH_CHECK();
        *did_target_sums = 1;
        return 0;
    }
    return launch_agg_bwd(mt->rowptr, mt->col, mt->val, src, tgt, pv, o, bf, w, den, out, dout, lc, cot_scale, n_rows,
                          dz, da, s);
}

extern "C" int64_t gode_gat_maxpath_scratch_bytes(int64_t n_edges) {
    (void)n_edges;
    return 1024 * (int64_t)(sizeof(float) + sizeof(int));
}

extern "C" int gode_gat_maxpath_f32(const float* a, const float* amax, float* da, int64_t n_edges,
                                    const int32_t* tgt, float* dat, int64_t ld_dat, void* scratch, void* stream) {
    if (n_edges < 0) return GODE_E_SHAPE;
    if (n_edges == 0) return 0;
    if (!a || !amax || !da) return GODE_E_NULLPTR;
    if (n_edges > INT32_MAX) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    if (n_edges <= kOneBlockEdges && !dat) {
        hipLaunchKernelGGL(gat_maxpath_kernel, dim3(1), dim3(1024), 0, s, a, amax, da, (int)n_edges);
        GODE_LAUNCH_CHECK();
        return 0;
    }
    if (!scratch) return GODE_E_NULLPTR;
    int64_t nb = (n_edges + 4095) / 4096; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    float* psum = (float*)scratch;
    int* pidx = (int*)(psum + 1024);
    hipLaunchKernelGGL(gat_maxpath_part_kernel, dim3((unsigned)nb), dim3(256), 0, s, a, amax, (const float*)da, (int)n_edges,
                       psum, pidx);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gat_maxpath_final_kernel, dim3(1), dim3(64), 0, s, (const float*)psum, (const int*)pidx, (int)nb, da,
                       (int)n_edges, tgt, dat, ld_dat);
    GODE_LAUNCH_CHECK();
    return 0;
}

namespace {
constexpr int64_t kSmallHeadEdges = 8192;         // below this the per-head maximum steps are single-block launches
int head_blocks(int64_t n_edges) {
    int64_t b = (n_edges + 1023) / 1024; if (b > kHeadBlocks) b = kHeadBlocks; if (b < 1) b = 1;
    return (int)b;
}
}  // namespace

// three regions of kHeadBlocks * heads entries: partial sums, partial arg-max indices (the max-path step), partial maxima
// (kept from the logits launch to the max-path launch of an adjoint stage on the raw-logit route)
extern "C" int64_t gode_gat_heads_scratch_bytes(int64_t n_edges, int64_t heads) {
    (void)n_edges;
    if (heads < 1) heads = 1;
    return (int64_t)kHeadBlocks * heads * (int64_t)(2 * sizeof(float) + sizeof(int));
}
extern "C" int64_t gode_gat_heads_parts(int64_t n_edges) { return head_blocks(n_edges); }

extern "C" int gode_gat_logits_heads_f32(const gode_gat_proj_t* proj, const float* bw, const int32_t* src,
                                         const int32_t* tgt, int64_t n_edges, int64_t heads, float* a, float* hmax,
                                         void* scratch, void* stream) {
    if (!proj) return GODE_E_NULLPTR;
    if (n_edges < 0 || heads < 1) return GODE_E_SHAPE;
    if (heads > kMaxHeads) return GODE_E_UNSUPPORTED;
    if (n_edges > INT32_MAX) return GODE_E_RANGE;
    if (n_edges == 0) return 0;
    const Proj pv = proj_of(proj);
    if (!pv.as || !pv.at || !src || !tgt || !a || !scratch) return GODE_E_NULLPTR;
    if (pv.lda < 1) return GODE_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (n_edges <= kSmallHeadEdges) {
        hipLaunchKernelGGL(gat_logits_heads_small_kernel, dim3(1), dim3(1024), 0, s, pv, bw, src, tgt, (int)n_edges,
                           (int)heads, a, hmax);
        GODE_LAUNCH_CHECK();
        return 0;
    }
    const int nb = head_blocks(n_edges);
    float* pmax = (float*)scratch;
    hipLaunchKernelGGL(gat_logits_heads_part_kernel, dim3(nb), dim3(256), 0, s, pv, bw, src, tgt, (int)n_edges, (int)heads, a, pmax);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gat_logits_heads_shift_kernel, dim3(nb), dim3(256), 0, s, tgt, (int)n_edges, (int)heads, nb,
                       (const float*)pmax, a, hmax);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gat_maxpath_heads_f32(const float* a, float* da, int64_t n_edges, int64_t heads, const int32_t* tgt,
                                          float* dat, int64_t ld_dat, void* scratch, void* stream) {
    if (n_edges < 0 || heads < 1) return GODE_E_SHAPE;
    if (heads > kMaxHeads) return GODE_E_UNSUPPORTED;
    if (n_edges > INT32_MAX) return GODE_E_RANGE;
    if (n_edges == 0) return 0;
    if (!a || !da || !tgt || !scratch) return GODE_E_NULLPTR;
    hipStream_t s = (hipStream_t)stream;
    if (n_edges <= kSmallHeadEdges && heads <= 8 && !dat) {          // 4 KB of LDS per head
        hipLaunchKernelGGL(gat_maxpath_heads_small_kernel, dim3(1), dim3(1024), (size_t)heads * 1024 * sizeof(float), s, a,
                           da, tgt, (int)n_edges, (int)heads);
        GODE_LAUNCH_CHECK();
        return 0;
    }
    const int nb = head_blocks(n_edges);
    float* psum = (float*)scratch;
    int* pidx = (int*)(psum + (int64_t)kHeadBlocks * heads);
    { const int rc = gode_set_lds_once((const void*)gat_maxpath_heads_part_kernel, (size_t)heads * 256 * sizeof(float));
      if (rc) return rc; }
    hipLaunchKernelGGL(gat_maxpath_heads_part_kernel, dim3(nb), dim3(256), (size_t)heads * 256 * sizeof(float), s, a,
                       (const float*)da, tgt, (int)n_edges, (int)heads, psum, pidx, (const float*)nullptr, 0);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gat_maxpath_heads_final_kernel, dim3(1), dim3(64), 0, s, (const float*)psum, (const int*)pidx, nb,
                       (int)heads, da, (int)n_edges, tgt, dat, ld_dat);
    GODE_LAUNCH_CHECK();
    return 0;
}

// ---- the raw-logit route of the H-head function: no launch that shifts the logits ---------------------------------
// logits_heads_raw: a[e] = As[src] + At[tgt] + bw[head] and the per-block partial maxima (third scratch region);
// agg_heads_fwd: the aggregation with every row shifted by its head's maximum, reduced from those partials in the kernel;
// maxpath_heads_raw: the max-path step on raw logits (arg-max = first edge whose logit equals its head's maximum).
extern "C" int gode_gat_logits_heads_raw_f32(const gode_gat_proj_t* proj, const float* bw, const int32_t* src,
                                             const int32_t* tgt, int64_t n_edges, int64_t heads, float* a, void* scratch,
                                             void* stream) {
    if (!proj) return GODE_E_NULLPTR;
    if (n_edges < 0 || heads < 1) return GODE_E_SHAPE;
    if (heads > kMaxHeads) return GODE_E_UNSUPPORTED;
    if (n_edges > INT32_MAX) return GODE_E_RANGE;
    if (n_edges == 0) return 0;
    const Proj pv = proj_of(proj);
    if (!pv.as || !pv.at || !src || !tgt || !a || !scratch) return GODE_E_NULLPTR;
    if (pv.lda < 1) return GODE_E_SHAPE;
    const int nb = head_blocks(n_edges);
    float* pmax = (float*)scratch + 2 * (int64_t)kHeadBlocks * heads;
    hipLaunchKernelGGL(gat_logits_heads_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pv, bw, src, tgt, (int)n_edges,
                       (int)heads, a, pmax);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gat_agg_heads_f32_fwd(const gode_graph_t* mt, const int32_t* src, const int32_t* tgt,
                                          const gode_gat_proj_t* proj, int64_t o, const float* bf, const float* a,
                                          const void* scratch, int64_t n_edges, int64_t heads, float eps, float* out,
                                          float* w_out, float* den_out, void* stream) {
    if (!mt || !proj || !scratch) return GODE_E_NULLPTR;
    const int64_t n_rows = mt->n_rows;
    if (n_rows < 0 || o <= 0 || heads < 1 || n_edges < 0) return GODE_E_SHAPE;
    if (heads > kMaxHeads) return GODE_E_UNSUPPORTED;
    if (n_rows == 0) return 0;
    if (!mt->rowptr || !out || !den_out) return GODE_E_NULLPTR;
    if (mt->nnz > 0 && (!src || !tgt || !a || !w_out)) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || o > 512) return (o > 512) ? GODE_E_UNSUPPORTED : GODE_E_RANGE;
    const Proj pv = proj_of(proj);
    int rc = check_proj(pv, o); if (rc) return rc;
    HeadMax hm;
    hm.pmax = (const float*)scratch + 2 * (int64_t)kHeadBlocks * heads; hm.n_part = n_edges > 0 ? head_blocks(n_edges) : 0; hm.H = (int)heads;
    return launch_agg_fwd(mt->rowptr, mt->col, mt->val, src, tgt, pv, o, bf, a, nullptr, eps, n_rows, out, w_out, den_out,
                          (hipStream_t)stream, hm);
}

// the first half only: per-block sums of da and arg-max candidates stay in `scratch`; gode_gat_dense_vjp_small_f32 closes the
// step itself (it subtracts the heads' sums from the two dA2 entries the arg-max edge feeds, when it loads those rows)
extern "C" int gode_gat_maxpath_heads_part_f32(const float* a, const float* da, int64_t n_edges, int64_t heads,
                                               const int32_t* tgt, void* scratch, void* stream) {
    if (n_edges < 0 || heads < 1) return GODE_E_SHAPE;
    if (heads > kMaxHeads) return GODE_E_UNSUPPORTED;
    if (n_edges > INT32_MAX) return GODE_E_RANGE;
    if (n_edges == 0) return 0;
    if (!a || !da || !tgt || !scratch) return GODE_E_NULLPTR;
    const int nb = head_blocks(n_edges);
    float* psum = (float*)scratch;
    int* pidx = (int*)(psum + (int64_t)kHeadBlocks * heads);
    const float* pmax = psum + 2 * (int64_t)kHeadBlocks * heads;
    { const int rc = gode_set_lds_once((const void*)gat_maxpath_heads_part_kernel, (size_t)heads * 256 * sizeof(float));
      if (rc) return rc; }
    hipLaunchKernelGGL(gat_maxpath_heads_part_kernel, dim3(nb), dim3(256), (size_t)heads * 256 * sizeof(float), (hipStream_t)stream, a,
                       da, tgt, (int)n_edges, (int)heads, psum, pidx, pmax, nb);
    GODE_LAUNCH_CHECK();
    return 0;
}
extern "C" int64_t gode_gat_heads_block_cap(void) { return kHeadBlocks; }

extern "C" int gode_gat_maxpath_heads_raw_f32(const float* a, float* da, int64_t n_edges, int64_t heads, const int32_t* tgt,
                                              float* dat, int64_t ld_dat, void* scratch, void* stream) {
    if (n_edges < 0 || heads < 1) return GODE_E_SHAPE;
    if (heads > kMaxHeads) return GODE_E_UNSUPPORTED;
    if (n_edges > INT32_MAX) return GODE_E_RANGE;
    if (n_edges == 0) return 0;
    if (!a || !da || !tgt || !scratch) return GODE_E_NULLPTR;
    hipStream_t s = (hipStream_t)stream;
    const int nb = head_blocks(n_edges);
    float* psum = (float*)scratch;
    int* pidx = (int*)(psum + (int64_t)kHeadBlocks * heads);
    const float* pmax = psum + 2 * (int64_t)kHeadBlocks * heads;
    { const int rc = gode_set_lds_once((const void*)gat_maxpath_heads_part_kernel, (size_t)heads * 256 * sizeof(float));
      if (rc) return rc; }
    hipLaunchKernelGGL(gat_maxpath_heads_part_kernel, dim3(nb), dim3(256), (size_t)heads * 256 * sizeof(float), s, a,
                       (const float*)da, tgt, (int)n_edges, (int)heads, psum, pidx, pmax, nb);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gat_maxpath_heads_final_kernel, dim3(1), dim3(64), 0, s, (const float*)psum, (const int*)pidx, nb,
                       (int)heads, da, (int)n_edges, tgt, dat, ld_dat);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gat_scatter_f32(const int32_t* rowptr_src, const int32_t* eid_src, const int32_t* rowptr_tgt,
                                    const int32_t* eid_tgt, const float* dz, const float* da, int64_t o, int64_t n_rows,
                                    float* dps, int64_t ld_s, float* dpt, int64_t ld_t, float* das, float* dat,
                                    int64_t ld_a, void* stream) {
    if (n_rows < 0 || o <= 0 || ld_s < o || ld_t < o || ld_a < 1) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    // eid_* / dz / da may be NULL for an edgeless graph (every row is empty and nothing is dereferenced)
    if (!rowptr_src || !rowptr_tgt || !dps || !dpt || !das || !dat) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || o > 512) return (o > 512) ? GODE_E_UNSUPPORTED : GODE_E_RANGE;
    const int G = pow2_group((int)o);
    const int maxc = (int)((o + G - 1) / G);
    const int64_t blocks = (n_rows + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
    if ((o == 4 || o == 8 || o == 16 || o == 32 || o == 64) && !(ld_s % 4) && !(ld_t % 4) &&
        !((((uintptr_t)dz) | ((uintptr_t)dps) | ((uintptr_t)dpt)) & 15)) {
#define GODE_PF(L, W) hipLaunchKernelGGL((gat_scatter_pf_kernel<L, W>), dim3((unsigned)((n_rows + 4 * (64 / W) - 1) / (4 * (64 / W)))), dim3(256), 0, s, \
                                         rowptr_src, eid_src, rowptr_tgt, eid_tgt, dz, da, (int)n_rows, dps, ld_s, dpt, ld_t, das, dat, ld_a)
        switch ((int)o / 4) { case 1: GODE_PF(1, 16); break; case 2: GODE_PF(2, 16); break; case 4: GODE_PF(4, 64); break; case 8: GODE_PF(8, 64); break; default: GODE_PF(16, 64); break; }
#undef GODE_PF
        GODE_LAUNCH_CHECK();
        return 0;
    }
#define GODE_SC(M) hipLaunchKernelGGL(gat_scatter_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, rowptr_src, eid_src, \
                                      rowptr_tgt, eid_tgt, dz, da, (int)o, (int)n_rows, G, dps, ld_s, dpt, ld_t, das, dat, ld_a)
    if (maxc <= 1) GODE_SC(1); else if (maxc <= 2) GODE_SC(2); else if (maxc <= 4) GODE_SC(4); else GODE_SC(8);
#undef GODE_SC
    GODE_LAUNCH_CHECK();
    return 0;
}

// the same for three weight blocks in one launch: at = sum_b <g_b, w_b> (added in block order), every g_b *= t
__global__ __launch_bounds__(256) void time_row_fixup3_kernel(float* g0, const float* w0, int l0, float* g1, const float* w1,
                                                              int l1, float* g2, const float* w2, int l2, float t,
                                                              float* __restrict__ at) {
    __shared__ float sm[4];
    float* gs[3] = {g0, g1, g2};
    const float* ws[3] = {w0, w1, w2};
    const int ls[3] = {l0, l1, l2};
    float total = 0.f;
    for (int b = 0; b < 3; ++b) {
        float s = 0.f;
        for (int c = threadIdx.x; c < ls[b]; c += 256) s += gs[b][c] * ws[b][c];
        s = wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
        __syncthreads();
        total += (sm[0] + sm[1]) + (sm[2] + sm[3]);
        for (int c = threadIdx.x; c < ls[b]; c += 256) gs[b][c] *= t;
    }
    if (threadIdx.x == 0) at[0] = total;
}

extern "C" int gode_time_row_fixup3_f32(float* g0, const float* w0, int64_t l0, float* g1, const float* w1, int64_t l1,
                                        float* g2, const float* w2, int64_t l2, float t, float* at, void* stream) {
    if (l0 <= 0 || l1 <= 0 || l2 <= 0) return GODE_E_SHAPE;
    if (!g0 || !w0 || !g1 || !w1 || !g2 || !w2 || !at) return GODE_E_NULLPTR;
    if (l0 > INT32_MAX || l1 > INT32_MAX || l2 > INT32_MAX) return GODE_E_RANGE;
    hipLaunchKernelGGL(time_row_fixup3_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, g0, w0, (int)l0, g1, w1, (int)l1,
                       g2, w2, (int)l2, t, at);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_time_row_fixup_f32(float* g_row0, const float* w_row0, int64_t len, float t, float* at,
                                       int accumulate, void* stream) {
    if (len <= 0) return GODE_E_SHAPE;
    if (!g_row0 || !w_row0 || !at) return GODE_E_NULLPTR;
    if (len > INT32_MAX) return GODE_E_RANGE;
    hipLaunchKernelGGL(time_row_fixup_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, g_row0, w_row0, (int)len, t, at,
                       accumulate ? 1 : 0);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_edge_matvec_f32_fwd(const int32_t* rowptr, const int32_t* eid, const float* val,
                                        const int32_t* src, const float* A, const float* X, int64_t ldx,
                                        int64_t h, int64_t n_rows, float* out, int64_t ldo, void* stream) {
    if (n_rows < 0 || h <= 0 || ldx < h || ldo < h) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!rowptr || !eid || !src || !A || !X || !out) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || h > 4096) return GODE_E_RANGE;
    const int64_t rpt = kMsgTile / h > 0 ? kMsgTile / h : 1;
    const size_t lds = (size_t)(2 * ((h + 3) & ~(int64_t)3) + (rpt < h ? rpt : h) * h) * sizeof(float);
    if (h <= 256) {
        hipLaunchKernelGGL(edge_matvec_fwd_kernel<1>, dim3((unsigned)n_rows), dim3(256), lds,
                           (hipStream_t)stream, rowptr, eid, val, src, A, X, ldx, (int)h, out, ldo);
    } else {
        if (lds > 48 * 1024) { const int rc = gode_set_lds_once((const void*)edge_matvec_fwd_kernel<16>, lds); if (rc) return rc; }
        hipLaunchKernelGGL(edge_matvec_fwd_kernel<16>, dim3((unsigned)n_rows), dim3(256), lds,
                           (hipStream_t)stream, rowptr, eid, val, src, A, X, ldx, (int)h, out, ldo);
    }
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_edge_matvec_msg_f32(const int32_t* src, const float* A, const float* X, int64_t ldx, int64_t h,
                                        int64_t n_edges, float* msg, void* stream) {
    if (n_edges < 0 || h <= 0 || ldx < h) return GODE_E_SHAPE;
    if (n_edges == 0) return 0;
    if (!src || !A || !X || !msg) return GODE_E_NULLPTR;
    if (n_edges > INT32_MAX || h > 4096) return GODE_E_RANGE;
    const int64_t rpt = kMsgTile / h > 0 ? kMsgTile / h : 1;
    const size_t lds = (size_t)(h + (rpt < h ? rpt : h) * h) * sizeof(float);
    hipLaunchKernelGGL(edge_matvec_msg_kernel, dim3((unsigned)n_edges), dim3(256), lds, (hipStream_t)stream, src, A, X,
                       ldx, (int)h, msg);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_edge_matvec_f32_bwd(const int32_t* edge_row, const float* edge_val, const int32_t* src,
                                        const float* A, const float* X, int64_t ldx, const float* dM, int64_t ldm,
                                        int64_t h, int64_t n_edges, float* dA, float* dxe, void* stream) {
    if (n_edges < 0 || h <= 0 || ldx < h || ldm < h) return GODE_E_SHAPE;
    if (n_edges == 0) return 0;
    if (!edge_row || !src || !A || !X || !dM) return GODE_E_NULLPTR;
    if (n_edges > INT32_MAX || h > 4096) return GODE_E_RANGE;
    hipLaunchKernelGGL(edge_matvec_bwd_kernel, dim3((unsigned)n_edges), dim3(256), (size_t)2 * h * sizeof(float),
                       (hipStream_t)stream, edge_row, edge_val, src, A, X, ldx, dM, ldm, (int)h, dA, dxe);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_edge_outer_sum_f32(const int32_t* edge_row, const float* edge_val, const int32_t* src, int32_t n_terms,
                                       const float* const* dM /* host[n_terms] */, const float* const* X /* host[n_terms] */,
                                       int64_t ldx, int64_t ldm, int64_t h, int64_t n_edges, float* dA, void* stream) {
    if (n_edges < 0 || h <= 0 || ldx < h || ldm < h) return GODE_E_SHAPE;
    if (n_terms < 1 || n_terms > GODE_MAX_TERMS) return GODE_E_RANGE;
    if (n_edges == 0) return 0;
    if (!edge_row || !src || !dM || !X || !dA) return GODE_E_NULLPTR;
    if (n_edges > INT32_MAX || h > 1024) return GODE_E_RANGE;
    OuterTerms tm;
    tm.n = n_terms;
    for (int t = 0; t < GODE_MAX_TERMS; ++t) {
        tm.dM[t] = t < n_terms ? dM[t] : nullptr; tm.X[t] = t < n_terms ? X[t] : nullptr;
        if (t < n_terms && (!dM[t] || !X[t])) return GODE_E_NULLPTR;
    }
    const size_t lds = (size_t)2 * n_terms * h * sizeof(float);
    if (lds > 48 * 1024) { const int rc = gode_set_lds_once((const void*)edge_outer_sum_kernel, lds); if (rc) return rc; }
    hipLaunchKernelGGL(edge_outer_sum_kernel, dim3((unsigned)n_edges), dim3(256), lds, (hipStream_t)stream, edge_row, edge_val, src, tm,
                       ldx, ldm, (int)h, dA);
    GODE_LAUNCH_CHECK();
    return 0;
}
