// lstm.hip — the LSTM cell of the Set2Set readout, forward and backward as ONE launch each (gfx950).
//
// Replaces, for the single-layer `self.lstm` of QC/set2set.py:44-47 stepped once per processing step (:61), the library
// path torch.lstm_cell takes on this image: two Tensile GEMMs + a cell kernel forward; a cell kernel, four GEMMs, two
// bias reductions backward - ~60 library launches per training step of MPNN_ENN_K_Set2Set (12 processing steps), each
// on a batch of 20 rows (`profiles/r03_qc_mpnn_kernel_stats.txt`).  The products are tiny (B = 20 graphs, input 2h = 146,
// hidden h = 73: 1.3 MFLOP) and the 256 KB of weights live in L2: the step is launch-bound, so the cell is laid out for
// one launch per direction, not for the matrix pipe.
//
//   gates = x W_ih^T + b_ih + h W_hh^T + b_hh            (B x 4H, gate order i, f, g, o as torch.nn.LSTM)
//   c' = sigmoid(f) c + sigmoid(i) tanh(g);   h' = sigmoid(o) tanh(c')
//
// Forward: one block per hidden unit u, wave g of the block owns gate row g H + u.  The unit's four rows of [W_ih | W_hh]
// and up to 64 rows of [x | h] are staged in LDS (every load in flight at once), then LANE b forms the dot product of row
// b with the wave's gate row - broadcast reads of the weights, conflict-free reads of the rows (odd stride), no wave
// reductions; the first wave applies the cell to unit u of every row.  Saved for the backward pass: the four
// activations per unit.
// Backward: every block recomputes the B x 4H gate cotangents from the saved activations into LDS (cheap: 5 840 values),
// then block kc owns four consecutive columns k of [x | h]: d[x|h][:, k] = dG W[:, k] (a wave per column, lane b the
// batch row) and d[W_ih | W_hh][:, k] = dG^T [x|h][:, k] (threads over gate rows); block 0 also writes the bias
// gradients (column sums of dG) and dc.  Everything is summed in a fixed order: deterministic.
#include "common.h"

namespace {

constexpr int kLstmMaxLds = 24576;      // floats of LDS a launch may ask for (96 KB)

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                            const float* __restrict__ c, const float* __restrict__ w_ih,
                                                            const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                            const float* __restrict__ b_hh, int B, int I, int H, int CB,
                                                            float* __restrict__ h_out, float* __restrict__ c_out,
                                                            float* __restrict__ gates)
{
    extern __shared__ float smem[];
    const int u = blockIdx.x, g = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j = g * H + u, K = I + H, KP = K | 1;            // odd row stride: lane b reads xs[b][k] without bank conflicts
    float* xs = smem;                                          // CB x KP: a chunk of rows of [x | h]
    float* wrow = smem + (int64_t)CB * KP;                     // 4 x K: the unit's four gate rows of [W_ih | W_hh]
    float* pre = wrow + 4 * K;                                 // 4 x 64
    for (int k = l; k < K; k += 64) wrow[g * K + k] = k < I ? w_ih[(int64_t)j * I + k] : w_hh[(int64_t)j * H + (k - I)];
    const float bias = (b_ih ? b_ih[j] : 0.f) + (b_hh ? b_hh[j] : 0.f);
    for (int b0 = 0; b0 < B; b0 += CB) {
        const int nb = B - b0 < CB ? B - b0 : CB;
        // the chunk's rows into LDS with every load in flight at once, then LANE b forms the dot product of row b with
        // the wave's gate row (broadcast reads): no wave reduction, no chain of B global round trips (first version:
        // a reduction per row straight from global memory, 34 us for 20 rows)
        for (int idx = threadIdx.x; idx < nb * K; idx += 256) {
            const int bb = idx / K, k = idx - bb * K;
            xs[bb * KP + k] = k < I ? x[(int64_t)(b0 + bb) * I + k] : h[(int64_t)(b0 + bb) * H + (k - I)];
        }
        __syncthreads();
        float d = bias;
        if (l < nb) {
            const float* xr = xs + l * KP;
            const float* wr = wrow + g * K;
            for (int k = 0; k < K; ++k) d = fmaf(wr[k], xr[k], d);
        }
        pre[g * 64 + l] = d;
        __syncthreads();
        if (g == 0 && l < nb) {
            const int b = b0 + l;
            const float ig = sigmoidf_(pre[l]), fg = sigmoidf_(pre[64 + l]), gg = tanhf(pre[128 + l]), og = sigmoidf_(pre[192 + l]);
            const float cn = fg * c[(int64_t)b * H + u] + ig * gg;
            c_out[(int64_t)b * H + u] = cn;
            h_out[(int64_t)b * H + u] = og * tanhf(cn);
            if (gates) {
                float* gp = gates + (int64_t)b * 4 * H + u;
                gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            }
        }
        __syncthreads();
    }
}

// dG[b][g H + u] from the saved activations and the cotangents of (h', c')
__device__ __forceinline__ void lstm_gate_cotangents(float* dG, const float* gates, const float* c, const float* c_out,
                                                     const float* dh, const float* dc, int B, int H, int GP, float* dc_prev /* nullable */)
{
    constexpr int UN = 4;                                        // items per thread with all their loads in flight
    for (int base = 0; base < B * H; base += 256 * UN) {
        float ig[UN], fg[UN], gg[UN], og[UN], cn[UN], cp[UN], dhv[UN], dcv[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int idx = base + threadIdx.x + 256 * q;
            const bool ok = idx < B * H;
            const int b = ok ? idx / H : 0, u = ok ? idx % H : 0, id = ok ? idx : 0;
            const float* gp = gates + (int64_t)b * 4 * H + u;
            ig[q] = gp[0]; fg[q] = gp[H]; gg[q] = gp[2 * H]; og[q] = gp[3 * H];
            cn[q] = c_out[id]; cp[q] = c[id];
            dhv[q] = dh ? dh[id] : 0.f; dcv[q] = dc ? dc[id] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int idx = base + threadIdx.x + 256 * q;
            if (idx < B * H) {
                const int b = idx / H, u = idx % H;
                const float tc = tanhf(cn[q]);
                const float dct = dcv[q] + dhv[q] * og[q] * (1.f - tc * tc);
                float* d = dG + (int64_t)b * GP + u;
                d[0] = dct * gg[q] * ig[q] * (1.f - ig[q]);
                d[H] = dct * cp[q] * fg[q] * (1.f - fg[q]);
                d[2 * H] = dct * ig[q] * (1.f - gg[q] * gg[q]);
                d[3 * H] = dhv[q] * tc * og[q] * (1.f - og[q]);
                if (dc_prev) dc_prev[idx] = dct * fg[q];
            }
        }
    }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                            const float* __restrict__ c, const float* __restrict__ w_ih,
                                                            const float* __restrict__ w_hh, const float* __restrict__ gates,
                                                            const float* __restrict__ c_out, const float* __restrict__ dh_out,
                                                            const float* __restrict__ dc_out, int B, int I, int H,
                                                            float* __restrict__ dx, float* __restrict__ dh, float* __restrict__ dc,
                                                            float* __restrict__ dw_ih, float* __restrict__ dw_hh,
                                                            float* __restrict__ db_ih, float* __restrict__ db_hh)
{
    extern __shared__ float smem[];
    const int G4 = 4 * H, K = I + H, GP = G4 | 1;               // odd row stride of dG: lane b reads dG[b][j] conflict-free
    float* dG = smem;                                           // B x GP
    float* xs = dG + (int64_t)B * GP;                           // B x 4: this block's columns of [x | h]
    float* wcol = xs + 4 * B;                                   // 4 x G4: the same columns of [W_ih | W_hh]
    const int k0 = 4 * blockIdx.x;
    for (int idx = threadIdx.x; idx < 4 * G4; idx += 256) {     // requested first: independent of the gate cotangents
        const int q = idx / G4, j = idx - q * G4, k = k0 + q;
        wcol[idx] = k < I ? w_ih[(int64_t)j * I + k] : (k < K ? w_hh[(int64_t)j * H + (k - I)] : 0.f);
    }
    for (int idx = threadIdx.x; idx < B * 4; idx += 256) {
        const int b = idx >> 2, k = k0 + (idx & 3);
        xs[idx] = k < I ? x[(int64_t)b * I + k] : (k < K ? h[(int64_t)b * H + (k - I)] : 0.f);
    }
    lstm_gate_cotangents(dG, gates, c, c_out, dh_out, dc_out, B, H, GP, blockIdx.x == 0 ? dc : nullptr);
    __syncthreads();
    // weight gradients of the four columns: thread t owns gate rows t, t + 256, ...
    for (int j = threadIdx.x; j < G4; j += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, s = 0.f;
        for (int b = 0; b < B; ++b) {
            const float gq = dG[(int64_t)b * GP + j];
            a0 = fmaf(gq, xs[4 * b], a0); a1 = fmaf(gq, xs[4 * b + 1], a1);
            a2 = fmaf(gq, xs[4 * b + 2], a2); a3 = fmaf(gq, xs[4 * b + 3], a3);
            s += gq;
        }
        const float av[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q;
            if (k < I) dw_ih[(int64_t)j * I + k] = av[q];
            else if (k < K) dw_hh[(int64_t)j * H + (k - I)] = av[q];
        }
        if (blockIdx.x == 0) { if (db_ih) db_ih[j] = s; if (db_hh) db_hh[j] = s; }
    }
    // input cotangents: wave w owns column k0 + w, LANE b the batch row (broadcast reads of the column, no reduction)
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, k = k0 + w;
    if (k < K) {
        const float* wc = wcol + w * G4;
        for (int b = l; b < B; b += 64) {
            const float* gr = dG + (int64_t)b * GP;
            float d = 0.f;
            for (int j = 0; j < G4; ++j) d = fmaf(wc[j], gr[j], d);
            if (k < I) { if (dx) dx[(int64_t)b * I + k] = d; }
            else if (dh) dh[(int64_t)b * H + (k - I)] = d;
        }
    }
}

}  // namespace

extern "C" int gode_lstm_cell_supported(int64_t B, int64_t I, int64_t H) {
    // LDS: backward B x (4H + 1) gate cotangents + 4 B + 16 H floats; forward at least one row of [x | h] + 4 (I + H) + 256
    return B > 0 && I > 0 && H > 0 && B * (4 * H + 1) + 4 * B + 16 * H <= kLstmMaxLds && 6 * (I + H) + 256 <= kLstmMaxLds;
}

extern "C" int gode_lstm_cell_f32_fwd(const float* x, const float* h, const float* c, const float* w_ih, const float* w_hh,
                                      const float* b_ih, const float* b_hh, int64_t B, int64_t I, int64_t H, float* h_out,
                                      float* c_out, float* gates, void* stream)
{
    if (B < 0 || I <= 0 || H <= 0) return GODE_E_SHAPE;
    if (B == 0) return 0;
    if (!x || !h || !c || !w_ih || !w_hh || !h_out || !c_out) return GODE_E_NULLPTR;
    if (!gode_lstm_cell_supported(B, I, H)) return GODE_E_UNSUPPORTED;
    const int64_t K = I + H, KP = K | 1;
    int64_t cb = (kLstmMaxLds - 256 - 4 * K) / KP;             // rows of [x | h] per round in LDS, at most one wave's lanes
    if (cb > 64) cb = 64;
    if (cb > B) cb = B;
    const size_t lds = (size_t)(cb * KP + 4 * K + 256) * sizeof(float);
    int rc = gode_set_lds_once(reinterpret_cast<const void*>(lstm_cell_fwd_kernel), lds); if (rc) return rc;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((unsigned)H), dim3(256), lds, (hipStream_t)stream, x, h, c, w_ih, w_hh, b_ih, b_hh,
                       (int)B, (int)I, (int)H, (int)cb, h_out, c_out, gates);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_lstm_cell_f32_bwd(const float* x, const float* h, const float* c, const float* w_ih, const float* w_hh,
                                      const float* gates, const float* c_out, const float* dh_out, const float* dc_out,
                                      int64_t B, int64_t I, int64_t H, float* dx, float* dh, float* dc, float* dw_ih,
                                      float* dw_hh, float* db_ih, float* db_hh, void* stream)
{
    if (B < 0 || I <= 0 || H <= 0) return GODE_E_SHAPE;
    if (B == 0) return 0;
    if (!x || !h || !c || !w_ih || !w_hh || !gates || !c_out || !dc || !dw_ih || !dw_hh) return GODE_E_NULLPTR;
    if (!gode_lstm_cell_supported(B, I, H)) return GODE_E_UNSUPPORTED;
    const size_t lds = (size_t)(B * ((4 * H) | 1) + 4 * B + 16 * H) * sizeof(float);
    int rc = gode_set_lds_once(reinterpret_cast<const void*>(lstm_cell_bwd_kernel), lds); if (rc) return rc;
    const int64_t blocks = (I + H + 3) / 4;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, h, c, w_ih, w_hh, gates,
                       c_out, dh_out, dc_out, (int)B, (int)I, (int)H, dx, dh, dc, dw_ih, dw_hh, db_ih, db_hh);
    GODE_LAUNCH_CHECK();
    return 0;
}
