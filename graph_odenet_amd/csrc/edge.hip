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

__device__ __forceinline__ float block_max(float v) {
    __shared__ float sm[4];
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// a[e] = P[src,2o] + P[tgt,2o+1] + bw ; block maxima -> pmax[block]
__global__ __launch_bounds__(256) void gat_logits_kernel(const float* __restrict__ P, int64_t ldp, int o,
                                                         const float* __restrict__ bw,
                                                         const int* __restrict__ src, const int* __restrict__ tgt,
                                                         int n_edges, float* __restrict__ a, float* __restrict__ pmax) {
    float m = -INFINITY;
    const float b = bw ? bw[0] : 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n_edges; e += gridDim.x * 256) {
        const float v = P[(int64_t)src[e] * ldp + 2 * o] + P[(int64_t)tgt[e] * ldp + 2 * o + 1] + b;
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
                                                          const float* __restrict__ P, int64_t ldp, int o,
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
        const int e = eid[k];
        const float w = expf(a[e] - m);
        const float we = (val ? val[k] : 1.f) * w;
        if (lane == 0) w_out[e] = w;
        s += we;
        const float* ps = P + (int64_t)src[e] * ldp;
        const float* pt = P + (int64_t)tgt[e] * ldp + o;
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
                                                          const float* __restrict__ P, int64_t ldp, int o,
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
        const int e = eid[k];
        const float we = w[e];
        const float vv = val ? val[k] : 1.f;
        const float* ps = P + (int64_t)src[e] * ldp;
        const float* pt = P + (int64_t)tgt[e] * ldp + o;
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

// ---------------- QC edge-conditioned messages ------------------------------------------------
// block per target row v: M_v = sum_{(e,val) in row v} val * A_e * x[src_e]
__global__ __launch_bounds__(256) void edge_matvec_fwd_kernel(const int* __restrict__ rowptr, const int* __restrict__ eid,
                                                              const float* __restrict__ val, const int* __restrict__ src,
                                                              const float* __restrict__ A, const float* __restrict__ X,
                                                              int64_t ldx, int h, float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;          // [h]
    float* macc = smem + h;    // [h]
    const int v = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < h; i += 256) macc[i] = 0.f;
    for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
        const int e = eid[k];
        const float vv = val ? val[k] : 1.f;
        __syncthreads();
        for (int j = threadIdx.x; j < h; j += 256) xs[j] = X[(int64_t)src[e] * ldx + j];
        __syncthreads();
        const float* Ae = A + (int64_t)e * h * h;
        for (int i = wave; i < h; i += 4) {
            float s = 0.f;
            for (int j = lane; j < h; j += 64) s = fmaf(Ae[(int64_t)i * h + j], xs[j], s);
            s = wave_sum(s);
            if (lane == 0) macc[i] += vv * s;     // row i is owned by exactly one wave
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h; i += 256) out[(int64_t)v * ldo + i] = macc[i];
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
            float s = 0.f;
            for (int i = 0; i < h; ++i) s = fmaf(A[base + (int64_t)i * h + j], dm[i], s);
            dxe[(int64_t)e * h + j] = s;
        }
    }
}

int pow2_group(int o) { int G = 1; while (G < o && G < 64) G <<= 1; return G; }

}  // namespace

extern "C" int64_t gode_edge_softmax_scratch_bytes(int64_t n_edges) {
    int64_t b = (n_edges + 255) / 256; if (b > 1024) b = 1024; if (b < 1) b = 1;
    return b * (int64_t)sizeof(float);
}

extern "C" int gode_edge_softmax_logits_f32(const float* P, int64_t ldp, int64_t o, const float* bw,
                                            const int32_t* src, const int32_t* tgt, int64_t n_edges,
                                            float* a, float* amax, float* scratch, void* stream) {
    if (n_edges < 0 || o <= 0 || ldp < 2 * o + 2) return GODE_E_SHAPE;
    if (!amax || !scratch) return GODE_E_NULLPTR;
    if (n_edges > 0 && (!P || !src || !tgt || !a)) return GODE_E_NULLPTR;
    if (n_edges > INT32_MAX || o > (1 << 20)) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    int64_t b = (n_edges + 255) / 256; if (b > 1024) b = 1024; if (b < 1) b = 1;
    hipLaunchKernelGGL(gat_logits_kernel, dim3((unsigned)b), dim3(256), 0, s, P, ldp, (int)o, bw, src, tgt,
                       (int)n_edges, a, scratch);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_max_kernel, dim3(1), dim3(256), 0, s, (const float*)scratch, (int)b, amax);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_edge_softmax_agg_f32_fwd(const int32_t* rowptr, const int32_t* eid, const float* val,
                                             const int32_t* src, const int32_t* tgt,
                                             const float* P, int64_t ldp, int64_t o, const float* bf,
                                             const float* a, const float* amax, float eps, int64_t n_rows,
                                             float* out, float* w_out, float* den_out, void* stream) {
    if (n_rows < 0 || o <= 0 || ldp < 2 * o + 2) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!rowptr || !eid || !src || !tgt || !P || !a || !amax || !out || !w_out || !den_out) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || o > 512) return (o > 512) ? GODE_E_UNSUPPORTED : GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    const int G = pow2_group((int)o);
    const int64_t blocks = (n_rows * G + 255) / 256;
    const int maxc = (int)((o + G - 1) / G);
#define GODE_AGG(M) hipLaunchKernelGGL(gat_agg_fwd_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, eid, val, \
                                       src, tgt, P, ldp, (int)o, bf, a, amax, eps, (int)n_rows, G, out, w_out, den_out)
    if (maxc <= 1) GODE_AGG(1); else if (maxc <= 2) GODE_AGG(2); else if (maxc <= 4) GODE_AGG(4); else GODE_AGG(8);
#undef GODE_AGG
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_edge_softmax_agg_f32_bwd(const int32_t* rowptr, const int32_t* eid, const float* val,
                                             const int32_t* src, const int32_t* tgt,
                                             const float* P, int64_t ldp, int64_t o, const float* bf,
                                             const float* w, const float* den, const float* out, const float* dout,
                                             int64_t n_rows, float* dz, float* da, void* stream) {
    if (n_rows < 0 || o <= 0 || ldp < 2 * o + 2) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!rowptr || !eid || !src || !tgt || !P || !w || !den || !out || !dout || !dz || !da) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || o > 512) return (o > 512) ? GODE_E_UNSUPPORTED : GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    const int G = pow2_group((int)o);
    const int64_t blocks = (n_rows * G + 255) / 256;
    const int maxc = (int)((o + G - 1) / G);
#define GODE_AGG(M) hipLaunchKernelGGL(gat_agg_bwd_kernel<M>, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, eid, val, \
                                       src, tgt, P, ldp, (int)o, bf, w, den, out, dout, (int)n_rows, G, dz, da)
    if (maxc <= 1) GODE_AGG(1); else if (maxc <= 2) GODE_AGG(2); else if (maxc <= 4) GODE_AGG(4); else GODE_AGG(8);
#undef GODE_AGG
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
    hipLaunchKernelGGL(edge_matvec_fwd_kernel, dim3((unsigned)n_rows), dim3(256), (size_t)2 * h * sizeof(float),
                       (hipStream_t)stream, rowptr, eid, val, src, A, X, ldx, (int)h, out, ldo);
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
