// set2set.hip — the whole Set2Set readout loop as ONE launch per direction (gfx950).
//
// Reference: QC/set2set.py:50-75.  For `processing_steps` rounds: q_t = LSTM(q*_{t-1}), e_i = <x_i, q_t[batch_i]>,
// a = softmax of e within each graph (the reference: a Python loop over the graphs of the batch), r_t = scatter_add(a x),
// q*_t = [q_t | r_t].  Nothing couples two graphs of a batch: row b of the LSTM state, of q_t and of r_t depend on the
// nodes of graph b only.  So ONE workgroup walks graph b through all processing steps - no launch boundary, no grid
// synchronisation, the recurrent state in LDS - where the per-step form needs 2 launches per step and direction
// (csrc/lstm.hip + csrc/segment.hip: 48 launches of ~11 us for the 12 steps of MPNN_ENN_K_Set2Set, 0.5 ms of a 2.4 ms
// training step, profiles/r04_qc_mpnn_kernel_stats.txt).
//
// Forward, per step: the gate rows  [W_ih | W_hh] [q* | h] + b  from the TRANSPOSED weights Wt[3H][4H] (loads coalesced
// over the rows, the state broadcast from LDS; the 1 024 threads split the contraction three ways and keep 16 loads in
// flight each - a step is a chain of dependent latencies, so loads in flight are what counts: the first version, one
// thread per row with 8 loads in flight, took 18 us per step, as long as the two launches it replaced); threads u < H
// apply the cell; then the segment attention of seg_attn_fwd_kernel (16 waves over the graph's nodes, lanes over features)
// on the graph's node rows, staged in LDS once for all steps.  Saved: q*_t, c_t, the gate activations, the attention
// weights.
// Backward, per step in reverse: the segment-attention backward (dx accumulated in place: a graph's node rows belong
// to its workgroup), the gate cotangents dG_t (written out), and d[q* | h] = W^T dG_t with thread k over the columns
// (W_ih, W_hh as stored: loads coalesced over k).  The weight gradients need no atomics and no per-step pass:
//     dW_ih = sum_t dG_t^T q*_{t-1} = DG^T QS   - one small product over the (steps x graphs) rows written here,
//     dW_hh = dW_ih[:, :H]                      - because h_{t-1} IS the left half of q*_{t-1},
//     db    = column sums of DG,
// formed by the caller with gode_gemm_f32 / gode_colsum_f32.  Everything is summed in a fixed order.
// Bound: latency (20 workgroups, 12 dependent steps, 256 KB of weights from L2 per step and workgroup).
#include "common.h"

namespace {

constexpr int NT = 1024, kWaves = NT / 64;
constexpr int kXCap = 12288;              // floats of LDS for the graph's own node rows (48 KB); larger graphs read x from L2

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ float blk_sum(float v, float* red) {     // v wave-uniform
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kWaves; ++j) s += red[j];
    return s;
}
__device__ __forceinline__ float blk_max(float v, float* red) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int j = 1; j < kWaves; ++j) s = fmaxf(s, red[j]);
    return s;
}

// out[j] = init[j] + sum_k M[k * ld + j] * v[k]  for j < nj, k < nk: the matrix-vector product of both directions (forward:
// M = Wt, v = [q* | h]; backward: M = W_ih / W_hh, v = dG).  The step is a chain of dependent latencies, so what counts is
// loads in flight: the NT threads split the k range KS ways (KS = NT / nj rounded down to what covers nj), every thread
// keeps 16 independent loads in flight, and the KS partial sums of an output are added in fixed order.
// part: NT floats of LDS.  Ends with a barrier.
__device__ __forceinline__ void matvec_cols(const float* __restrict__ M, int64_t ld, const float* v, int nj, int nk,
                                            const float* init0, const float* init1, float* out, float* part)
{
    const int tid = threadIdx.x;
    const int jp = nj < NT ? ((nj + 63) & ~63) : NT;                // threads per k slice
    const int ks_n = NT / jp > 0 ? NT / jp : 1;
    for (int j0 = 0; j0 < nj; j0 += jp) {                           // one pass unless nj > NT
        const int ks = tid / jp, j = j0 + tid - ks * jp;
        const int klen = (nk + ks_n - 1) / ks_n, k0 = ks * klen, k1 = k0 + klen < nk ? k0 + klen : nk;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (ks < ks_n && j < nj) {
            const float* mc = M + j;
            int k = k0;
            for (; k + 16 <= k1; k += 16) {
                float m[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) m[u] = mc[(int64_t)(k + u) * ld];
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    a0 = fmaf(m[u], v[k + u], a0); a1 = fmaf(m[u + 1], v[k + u + 1], a1);
                    a2 = fmaf(m[u + 2], v[k + u + 2], a2); a3 = fmaf(m[u + 3], v[k + u + 3], a3);
                }
            }
            for (; k < k1; ++k) a0 = fmaf(mc[(int64_t)k * ld], v[k], a0);
        }
        part[tid] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tid < jp && j0 + tid < nj) {
            float s = (init0 ? init0[j0 + tid] : 0.f) + (init1 ? init1[j0 + tid] : 0.f);
            for (int q = 0; q < ks_n; ++q) s += part[q * jp + tid];
            out[j0 + tid] = s;
        }
        __syncthreads();
    }
}

// LDS (floats): zin[3H] = [q* | h], pre[4H], cst[H], hq[H], racc[kWaves][H], part[NT], red[kWaves], xs[<= kXCap]
__global__ __launch_bounds__(NT) void set2set_fwd_kernel(const int32_t* __restrict__ segptr, const int32_t* __restrict__ perm,
                                                         const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ Wt /* [3H][4H] */,
                                                         const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                         int H, int T, int B, int64_t N, float* __restrict__ qs,
                                                         float* __restrict__ cs, float* __restrict__ gates,
                                                         float* __restrict__ att)
{
    extern __shared__ float sm[];
    float* zin = sm;
    float* pre = zin + 3 * H;
    float* cst = pre + 4 * H;
    float* hq = cst + H;
    float* racc = hq + H;
    float* part = racc + kWaves * H;
    float* red = part + NT;
    float* xs = red + kWaves;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int beg = segptr[b], end = segptr[b + 1], nb = end - beg;
    const int G4 = 4 * H, K3 = 3 * H;
    const bool staged = (int64_t)nb * H <= kXCap;                   // the graph's node rows live in LDS for all steps
    if (staged)
        for (int i = tid; i < nb * H; i += NT) {
            const int k = i / H, f = i - k * H;
            xs[i] = x[(int64_t)(perm ? perm[beg + k] : beg + k) * ldx + f];
        }
    for (int k = tid; k < K3; k += NT) zin[k] = 0.f;
    for (int u = tid; u < H; u += NT) {
        cst[u] = 0.f;
        cs[(int64_t)b * H + u] = 0.f;                               // c_0
    }
    for (int k = tid; k < 2 * H; k += NT) qs[(int64_t)b * 2 * H + k] = 0.f;   // q*_0
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        // ---- LSTM cell, row b: pre = [W_ih | W_hh] [q* | h] + b_ih + b_hh
        matvec_cols(Wt, G4, zin, G4, K3, b_ih, b_hh, pre, part);
        float* gt = gates + ((int64_t)t * B + b) * G4;
        float* cn_out = cs + ((int64_t)(t + 1) * B + b) * H;
        for (int u = tid; u < H; u += NT) {
            const float ig = sigm(pre[u]), fg = sigm(pre[H + u]), gg = tanhf(pre[2 * H + u]), og = sigm(pre[3 * H + u]);
            const float cn = fg * cst[u] + ig * gg;
            cst[u] = cn;
            cn_out[u] = cn;
            hq[u] = og * tanhf(cn);
            gt[u] = ig; gt[H + u] = fg; gt[2 * H + u] = gg; gt[3 * H + u] = og;
        }
        __syncthreads();
        // ---- segment attention of graph b with query hq (the passes of seg_attn_fwd_kernel; logits kept in att[])
        float* at = att + (int64_t)t * N;
        float m = -INFINITY;
        for (int k = w; k < nb; k += kWaves) {
            const int node = perm ? perm[beg + k] : beg + k;
            const float* xr = staged ? xs + k * H : x + (int64_t)node * ldx;
            float dot = 0.f;
            for (int f = lane; f < H; f += 64) dot = fmaf(xr[f], hq[f], dot);
            dot = wave_sum(dot);
            if (lane == 0) at[node] = dot;
            m = fmaxf(m, dot);
        }
        m = blk_max(m, red);
        float s = 0.f;
        for (int f = lane; f < H; f += 64) racc[w * H + f] = 0.f;
        for (int k = w; k < nb; k += kWaves) {
            const int node = perm ? perm[beg + k] : beg + k;
            const float* xr = staged ? xs + k * H : x + (int64_t)node * ldx;
            float e = (lane == 0) ? at[node] : 0.f;
            e = __shfl(e, 0, 64);
            const float wgt = expf(e - m);
            s += wgt;
            for (int f = lane; f < H; f += 64) racc[w * H + f] = fmaf(wgt, xr[f], racc[w * H + f]);
            if (lane == 0) at[node] = wgt;
        }
        const float S = blk_sum(s, red);                            // also orders racc[] and at[] for the block
        const float inv = nb > 0 ? 1.f / S : 0.f;
        float* qo = qs + ((int64_t)(t + 1) * B + b) * 2 * H;
        for (int f = tid; f < H; f += NT) {
            float r = 0.f;
            for (int j = 0; j < kWaves; ++j) r += racc[j * H + f];
            r *= inv;
            const float h = hq[f];
            zin[f] = h; zin[H + f] = r; zin[2 * H + f] = h;
            qo[f] = h; qo[H + f] = r;
        }
        for (int k = tid; k < nb; k += NT) {
            const int node = perm ? perm[beg + k] : beg + k;
            at[node] *= inv;
        }
        __syncthreads();
    }
}

// LDS (floats): dqs[2H], dhn[H], dcn[H], dG[4H], qv[H], drv[H], dht[H], qacc[kWaves][H], part[NT], red[kWaves], xs[<= kXCap]
__global__ __launch_bounds__(NT) void set2set_bwd_kernel(const int32_t* __restrict__ segptr, const int32_t* __restrict__ perm,
                                                         const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ w_ih /* [4H][2H] */,
                                                         const float* __restrict__ w_hh /* [4H][H] */, int H, int T, int B,
                                                         int64_t N, const float* __restrict__ qs, const float* __restrict__ cs,
                                                         const float* __restrict__ gates, const float* __restrict__ att,
                                                         const float* __restrict__ dq_final, float* __restrict__ dx,
                                                         float* __restrict__ DG)
{
    extern __shared__ float sm[];
    float* dqs = sm;
    float* dhn = dqs + 2 * H;
    float* dcn = dhn + H;
    float* dG = dcn + H;
    float* qv = dG + 4 * H;
    float* drv = qv + H;
    float* dht = drv + H;
    float* qacc = dht + H;
    float* part = qacc + kWaves * H;
    float* red = part + NT;
    float* xs = red + kWaves;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int beg = segptr[b], end = segptr[b + 1], nb = end - beg;
    const int G4 = 4 * H;
    const bool staged = (int64_t)nb * H <= kXCap;
    if (staged)
        for (int i = tid; i < nb * H; i += NT) {
            const int k = i / H, f = i - k * H;
            xs[i] = x[(int64_t)(perm ? perm[beg + k] : beg + k) * ldx + f];
        }
    for (int k = tid; k < 2 * H; k += NT) dqs[k] = dq_final[(int64_t)b * 2 * H + k];
    for (int u = tid; u < H; u += NT) { dhn[u] = 0.f; dcn[u] = 0.f; }
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const float* qt = qs + ((int64_t)(t + 1) * B + b) * 2 * H;     // q*_t = [q_t | r_t]
        for (int f = tid; f < H; f += NT) { qv[f] = qt[f]; drv[f] = dqs[H + f]; }
        __syncthreads();
        // ---- segment attention backward (seg_attn_bwd_kernel): da_i = <x_i, dr>, S = sum a_i da_i, de_i = a_i (da_i - S),
        //      dx_i (+)= a_i dr + de_i q,  dq = sum_i de_i x_i
        const float* at = att + (int64_t)t * N;
        float ts = 0.f;
        for (int k = w; k < nb; k += kWaves) {
            const int node = perm ? perm[beg + k] : beg + k;
            const float* xr = staged ? xs + k * H : x + (int64_t)node * ldx;
            float dot = 0.f;
            for (int f = lane; f < H; f += 64) dot = fmaf(xr[f], drv[f], dot);
            ts = fmaf(at[node], wave_sum(dot), ts);
        }
        const float S = blk_sum(ts, red);
        for (int f = lane; f < H; f += 64) qacc[w * H + f] = 0.f;
        for (int k = w; k < nb; k += kWaves) {
            const int node = perm ? perm[beg + k] : beg + k;
            const float* xr = staged ? xs + k * H : x + (int64_t)node * ldx;
            float dot = 0.f;
            for (int f = lane; f < H; f += 64) dot = fmaf(xr[f], drv[f], dot);
            const float ai = at[node];
            const float de = ai * (wave_sum(dot) - S);
            float* dxr = dx + (int64_t)node * H;
            for (int f = lane; f < H; f += 64) {
                const float v = fmaf(ai, drv[f], de * qv[f]);
                dxr[f] = (t == T - 1) ? v : dxr[f] + v;            // this workgroup owns the row: plain read-modify-write
                qacc[w * H + f] = fmaf(de, xr[f], qacc[w * H + f]);
            }
        }
        __syncthreads();
        // ---- cotangent of q_t = h_t: the left half of dq*_t + the next cell's dh + the attention's dq
        for (int u = tid; u < H; u += NT) {
            float v = 0.f;
            for (int j = 0; j < kWaves; ++j) v += qacc[j * H + u];
            dht[u] = (dqs[u] + dhn[u]) + v;
        }
        __syncthreads();
        // ---- gate cotangents (lstm_gate_cotangents)
        const float* gt = gates + ((int64_t)t * B + b) * G4;
        const float* cn_ = cs + ((int64_t)(t + 1) * B + b) * H;
        const float* cp_ = cs + ((int64_t)t * B + b) * H;
        float* dgo = DG + ((int64_t)t * B + b) * G4;
        for (int u = tid; u < H; u += NT) {
            const float ig = gt[u], fg = gt[H + u], gg = gt[2 * H + u], og = gt[3 * H + u];
            const float tc = tanhf(cn_[u]);
            const float dh = dht[u];
            const float dct = dcn[u] + dh * og * (1.f - tc * tc);
            const float d0 = dct * gg * ig * (1.f - ig), d1 = dct * cp_[u] * fg * (1.f - fg);
            const float d2 = dct * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
            dG[u] = d0; dG[H + u] = d1; dG[2 * H + u] = d2; dG[3 * H + u] = d3;
            dgo[u] = d0; dgo[H + u] = d1; dgo[2 * H + u] = d2; dgo[3 * H + u] = d3;
            dcn[u] = dct * fg;
        }
        __syncthreads();
        // ---- d[q*_{t-1} | h_{t-1}] = W^T dG (W_ih, W_hh as stored: the columns are contiguous over the threads)
        matvec_cols(w_ih, 2 * H, dG, 2 * H, G4, nullptr, nullptr, dqs, part);
        matvec_cols(w_hh, H, dG, H, G4, nullptr, nullptr, dhn, part);
    }
}

}  // namespace

extern "C" int gode_set2set_supported(int64_t H) { return H > 0 && H <= 512; }     // (11 + 16) H + 13 K floats of LDS

extern "C" int gode_set2set_f32_fwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx, const float* Wt,
                                    const float* b_ih, const float* b_hh, int64_t n_graphs, int64_t H, int64_t steps,
                                    int64_t n_nodes, float* qs, float* cs, float* gates, float* att, void* stream)
{
    if (n_graphs < 0 || H <= 0 || steps <= 0 || n_nodes < 0 || ldx < H) return GODE_E_SHAPE;
    if (n_graphs == 0) return 0;
    if (!segptr || !x || !Wt || !qs || !cs || !gates || !att) return GODE_E_NULLPTR;
    if (!gode_set2set_supported(H)) return GODE_E_UNSUPPORTED;
    if (n_graphs > INT32_MAX || steps > 4096) return GODE_E_RANGE;
    const size_t lds = (size_t)(3 * H + 4 * H + H + H + kWaves * H + NT + kWaves + kXCap) * sizeof(float);
    const int rc = gode_set_lds_once(reinterpret_cast<const void*>(set2set_fwd_kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(set2set_fwd_kernel, dim3((unsigned)n_graphs), dim3(NT), lds, (hipStream_t)stream, segptr, perm, x, ldx, Wt,
                       b_ih, b_hh, (int)H, (int)steps, (int)n_graphs, n_nodes, qs, cs, gates, att);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_set2set_f32_bwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx, const float* w_ih,
                                    const float* w_hh, int64_t n_graphs, int64_t H, int64_t steps, int64_t n_nodes,
                                    const float* qs, const float* cs, const float* gates, const float* att,
                                    const float* dq_final, float* dx, float* DG, void* stream)
{
    if (n_graphs < 0 || H <= 0 || steps <= 0 || n_nodes < 0 || ldx < H) return GODE_E_SHAPE;
    if (n_graphs == 0) return 0;
    if (!segptr || !x || !w_ih || !w_hh || !qs || !cs || !gates || !att || !dq_final || !dx || !DG) return GODE_E_NULLPTR;
    if (!gode_set2set_supported(H)) return GODE_E_UNSUPPORTED;
    if (n_graphs > INT32_MAX || steps > 4096) return GODE_E_RANGE;
    const size_t lds = (size_t)(2 * H + H + H + 4 * H + 3 * H + kWaves * H + NT + kWaves + kXCap) * sizeof(float);
    const int rc = gode_set_lds_once(reinterpret_cast<const void*>(set2set_bwd_kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(set2set_bwd_kernel, dim3((unsigned)n_graphs), dim3(NT), lds, (hipStream_t)stream, segptr, perm, x, ldx,
                       w_ih, w_hh, (int)H, (int)steps, (int)n_graphs, n_nodes, qs, cs, gates, att, dq_final, dx, DG);
    GODE_LAUNCH_CHECK();
    return 0;
}
