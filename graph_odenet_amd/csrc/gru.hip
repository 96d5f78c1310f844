// gru.hip — the node update of the QC message-passing layer as fused kernels (gfx950).
//
// Replaces `x = self.update_net(torch.cat([x, node_msg], 1), x)` with update_net = nn.GRUCell(2h, h) at
// QC/mpnn.py:12,30 of the reference (the GRU half), and its autograd:
//
//   gi = [x | m] W_ih^T + b_ih      gh = x W_hh^T + b_hh                (W_ih: 3h x 2h, W_hh: 3h x h; gates r, z, n)
//   r = sigmoid(gi_r + gh_r)   z = sigmoid(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   out = (1 - z) n + z x
//
// One launch forward (no concatenated input, no N x 3h gate matrices in memory besides the 4h values per row the
// backward pass needs), three launches backward (gate derivatives + input gradients; weight / bias gradient partials
// over row chunks; their fixed-order reduction).  The shapes are small and irregular (QM9 batch: N ~ 360 rows, h = 73,
// 48 K weights): plain fp32 FMA out of LDS-staged rows, weights streamed from L2 - launch-bound work, where the
// library path costs ~4 launches forward and ~10 backward, each re-tuned for every new N.
// Bound: launch latency (35 MFLOP per product at N = 360).
#include "common.h"

namespace {

constexpr int RB = 8;            // rows per block
constexpr int WCHUNK = 64;       // rows per weight-gradient partial

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

static_assert(RB == 8, "gru_sweep reads the eight rows of a gate column as two float4");
// a[rr] += sum_j dg[j][rr] W[j][col], j < h3 (dg in LDS as [3h][RB]; W row-major with leading dimension ldw): 32 weight
// rows per trip, every load unconditional (clamped row index; a surplus row multiplies a zero).  16 rows per trip made the
// loop 28 dependent L2 round trips - 42 us per backward launch at N = 380 (87 us before any were overlapped).
__device__ __forceinline__ void gru_sweep(const float* __restrict__ W, int ldw, const float* __restrict__ dg, int col, int h3,
                                          float (&a)[RB]) {
    for (int j0 = 0; j0 < h3; j0 += 32) {
        float w[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { const int j = j0 + u; w[u] = W[(int64_t)(j < h3 ? j : h3 - 1) * ldw + col]; }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int j = j0 + u, jc = j < h3 ? j : h3 - 1;
            const float wv = j < h3 ? w[u] : 0.f;
            // dg is [j][RB] in LDS: the eight rows' values of gate column j are two 16-byte reads (row-major [RB][3h] took
            // eight 4-byte reads per j: 3 500 LDS instructions per thread, which - not the weight loads - bounded the kernel)
            const float4 d0 = *reinterpret_cast<const float4*>(dg + jc * RB), d1 = *reinterpret_cast<const float4*>(dg + jc * RB + 4);
            a[0] = fmaf(d0.x, wv, a[0]); a[1] = fmaf(d0.y, wv, a[1]); a[2] = fmaf(d0.z, wv, a[2]); a[3] = fmaf(d0.w, wv, a[3]);
            a[4] = fmaf(d1.x, wv, a[4]); a[5] = fmaf(d1.y, wv, a[5]); a[6] = fmaf(d1.z, wv, a[6]); a[7] = fmaf(d1.w, wv, a[7]);
        }
    }
}

// forward: block = RB rows.  LDS: xm[2h][RB] | gi[RB][3h] | gh[RB][3h]
__global__ __launch_bounds__(256) void gru_fwd_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                      const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                      const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                      int n, int h, float* __restrict__ out, float* __restrict__ gates)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int h2 = 2 * h, h3 = 3 * h;
    float* xm = smem;
    float* gi = xm + RB * h2;
    float* gh = gi + RB * h3;
    const int row0 = blockIdx.x * RB;
    for (int idx = threadIdx.x; idx < RB * h2; idx += 256) {
        const int rr = idx / h2, k = idx % h2, row = row0 + rr;
        float v = 0.f;
        if (row < n) v = k < h ? x[(int64_t)row * h + k] : m[(int64_t)row * h + (k - h)];
        xm[k * RB + rr] = v;                            // [input column][row]: the eight rows of a column are two 16-byte reads
    }
    __syncthreads();
    auto fma8 = [](float (&acc)[RB], float w, const float* __restrict__ col) {
        const float4 d0 = *reinterpret_cast<const float4*>(col), d1 = *reinterpret_cast<const float4*>(col + 4);
        acc[0] = fmaf(w, d0.x, acc[0]); acc[1] = fmaf(w, d0.y, acc[1]); acc[2] = fmaf(w, d0.z, acc[2]); acc[3] = fmaf(w, d0.w, acc[3]);
        acc[4] = fmaf(w, d1.x, acc[4]); acc[5] = fmaf(w, d1.y, acc[5]); acc[6] = fmaf(w, d1.z, acc[6]); acc[7] = fmaf(w, d1.w, acc[7]);
    };
    for (int j = threadIdx.x; j < h3; j += 256) {       // gate column j of every row of the block: weights read once
        float ai[RB], ah[RB];
        const float bi = b_ih ? b_ih[j] : 0.f, bh = b_hh ? b_hh[j] : 0.f;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) { ai[rr] = bi; ah[rr] = bh; }
        const float* wi = w_ih + (int64_t)j * h2;
        const float* wh = w_hh + (int64_t)j * h;
        int k = 0;
        for (; k + 8 <= h2; k += 8) {                   // 8 weight loads in flight
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = wi[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) fma8(ai, w8[u], xm + (k + u) * RB);
        }
        for (; k < h2; ++k) {
            fma8(ai, wi[k], xm + k * RB);
        }
        k = 0;
        for (; k + 8 <= h; k += 8) {
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = wh[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) fma8(ah, w8[u], xm + (k + u) * RB);
        }
        for (; k < h; ++k) {
            fma8(ah, wh[k], xm + k * RB);
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) { gi[rr * h3 + j] = ai[rr]; gh[rr * h3 + j] = ah[rr]; }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < RB * h; idx += 256) {
        const int rr = idx / h, c = idx % h, row = row0 + rr;
        if (row >= n) continue;
        const float r = sigmoidf_(gi[rr * h3 + c] + gh[rr * h3 + c]);
        const float z = sigmoidf_(gi[rr * h3 + h + c] + gh[rr * h3 + h + c]);
        const float hn = gh[rr * h3 + 2 * h + c];
        const float nn = tanhf(gi[rr * h3 + 2 * h + c] + r * hn);
        const float xv = xm[c * RB + rr];
        out[(int64_t)row * h + c] = (1.0f - z) * nn + z * xv;
        if (gates) {
            float* g = gates + (int64_t)row * 4 * h;
            g[c] = r; g[h + c] = z; g[2 * h + c] = nn; g[3 * h + c] = hn;
        }
    }
}

// backward 1: gate derivatives (written out for the weight gradients) and the input gradients.
// LDS: dgi[RB][3h] | dgh[RB][3h]
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                      const float* __restrict__ w_hh, const float* __restrict__ gates,
                                                      const float* __restrict__ dout, int n, int h,
                                                      float* __restrict__ dx, float* __restrict__ dm,
                                                      float* __restrict__ dgi_out, float* __restrict__ dgh_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int h2 = 2 * h, h3 = 3 * h;
    float* dgi = smem;
    float* dgh = dgi + RB * h3;
    float* dxd = dgh + RB * h3;          // [RB][h]: dout * z (the direct path to x)
    const int row0 = blockIdx.x * RB;
    for (int idx = threadIdx.x; idx < RB * h; idx += 256) {
        const int rr = idx / h, c = idx % h, row = row0 + rr;
        float gr = 0.f, gz = 0.f, gn = 0.f, ghn = 0.f, d0 = 0.f;
        if (row < n) {
            const float* g = gates + (int64_t)row * 4 * h;
            const float r = g[c], z = g[h + c], nn = g[2 * h + c], hn = g[3 * h + c];
            const float dy = dout[(int64_t)row * h + c];
            const float xv = x[(int64_t)row * h + c];
            const float dn = dy * (1.0f - z);
            const float dz = dy * (xv - nn);
            gn = dn * (1.0f - nn * nn);
            ghn = gn * r;
            gr = gn * hn * r * (1.0f - r);
            gz = dz * z * (1.0f - z);
            d0 = dy * z;
            float* oi = dgi_out + (int64_t)row * h3;
            float* oh = dgh_out + (int64_t)row * h3;
            oi[c] = gr; oi[h + c] = gz; oi[2 * h + c] = gn;
            oh[c] = gr; oh[h + c] = gz; oh[2 * h + c] = ghn;
        }
        dgi[c * RB + rr] = gr; dgi[(h + c) * RB + rr] = gz; dgi[(2 * h + c) * RB + rr] = gn;         // [gate column][row]
        dgh[c * RB + rr] = gr; dgh[(h + c) * RB + rr] = gz; dgh[(2 * h + c) * RB + rr] = ghn;
        dxd[rr * h + c] = d0;
    }
    __syncthreads();
    // d[x | m][row][k] = sum_j dgi[row][j] W_ih[j][k]  (+ for k < h: sum_j dgh[row][j] W_hh[j][k] + dout*z)
    for (int k = threadIdx.x; k < h2; k += 256) {       // adjacent threads read adjacent weights of a row
        float a[RB];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) a[rr] = k < h ? dxd[rr * h + k] : 0.f;
        gru_sweep(w_ih, h2, dgi, k, h3, a);
        if (k < h) gru_sweep(w_hh, h, dgh, k, h3, a);
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int row = row0 + rr;
            if (row >= n) continue;
            if (k < h) { if (dx) dx[(int64_t)row * h + k] = a[rr]; }
            else if (dm) dm[(int64_t)row * h + (k - h)] = a[rr];
        }
    }
}

// backward 2: partial weight / bias gradients over a chunk of rows.  grid = (3h gate rows, chunks); a partial is
// [3h][3h + 2]: columns 0..2h-1 = dW_ih row, 2h..3h-1 = dW_hh row, 3h = db_ih, 3h+1 = db_hh.
__global__ __launch_bounds__(256) void gru_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                        const float* __restrict__ dgi, const float* __restrict__ dgh,
                                                        int n, int h, float* __restrict__ part)
{
    const int h2 = 2 * h, h3 = 3 * h, ld = h3 + 2;
    const int j = blockIdx.x;
    const int r0 = blockIdx.y * WCHUNK;
    const int r1 = min(n, r0 + WCHUNK);
    float* out = part + ((int64_t)blockIdx.y * h3 + j) * ld;
    // the chunk's gate derivatives of gate row j, once per block (every thread multiplies the same values)
    __shared__ float gi_s[WCHUNK], gh_s[WCHUNK];
    for (int q = threadIdx.x; q < r1 - r0; q += 256) {
        gi_s[q] = dgi[(int64_t)(r0 + q) * h3 + j];
        gh_s[q] = dgh[(int64_t)(r0 + q) * h3 + j];
    }
    __syncthreads();
    const int nr = r1 - r0;
    for (int k = threadIdx.x; k < ld; k += 256) {
        float acc = 0.f;
        if (k < h3) {
            const float* src = k < h ? x + k : (k < h2 ? m + (k - h) : x + (k - h2));
            const float* gs = k < h2 ? gi_s : gh_s;
            int q = 0;
            for (; q + 8 <= nr; q += 8) {               // 8 loads in flight; rows added in order (deterministic)
                float v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v8[u] = src[(int64_t)(r0 + q + u) * h];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fmaf(gs[q + u], v8[u], acc);
            }
            for (; q < nr; ++q) acc = fmaf(gs[q], src[(int64_t)(r0 + q) * h], acc);
        } else {
            const float* gs = k == h3 ? gi_s : gh_s;
            for (int q = 0; q < nr; ++q) acc += gs[q];
        }
        out[k] = acc;
    }
}

// backward 3: partials added in chunk order and scattered into the four gradients
__global__ __launch_bounds__(256) void gru_wreduce_kernel(const float* __restrict__ part, int n_part, int h,
                                                          float* __restrict__ dw_ih, float* __restrict__ dw_hh,
                                                          float* __restrict__ db_ih, float* __restrict__ db_hh)
{
    const int h2 = 2 * h, h3 = 3 * h, ld = h3 + 2;
    const int64_t total = (int64_t)h3 * ld;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int p = 0; p < n_part; ++p) s += part[(int64_t)p * total + idx];
        const int j = (int)(idx / ld), k = (int)(idx % ld);
        if (k < h2) dw_ih[(int64_t)j * h2 + k] = s;
        else if (k < h3) dw_hh[(int64_t)j * h + (k - h2)] = s;
        else if (k == h3) { if (db_ih) db_ih[j] = s; }
        else if (db_hh) db_hh[j] = s;
    }
}

}  // namespace

extern "C" int64_t gode_gru_wgrad_parts(int64_t n) { return n <= 0 ? 1 : (n + WCHUNK - 1) / WCHUNK; }

extern "C" int gode_gru_cell_f32_fwd(const float* x, const float* m, const float* w_ih, const float* w_hh,
                                     const float* b_ih, const float* b_hh, int64_t n, int64_t h, float* out,
                                     float* gates, void* stream)
{
    if (n < 0 || h <= 0) return GODE_E_SHAPE;
    if (n == 0) return 0;
    if (!x || !m || !w_ih || !w_hh || !out) return GODE_E_NULLPTR;
    if (n > INT32_MAX || h > 1024) return GODE_E_RANGE;
    const size_t lds = (size_t)RB * 8 * h * sizeof(float);
    if (lds > 160 * 1024) return GODE_E_UNSUPPORTED;
    int rc = gode_set_lds_once((const void*)gru_fwd_kernel, lds); if (rc) return rc;
    const int64_t blocks = (n + RB - 1) / RB;
    hipLaunchKernelGGL(gru_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream,
                       x, m, w_ih, w_hh, b_ih, b_hh, (int)n, (int)h, out, gates);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_gru_cell_f32_bwd(const float* x, const float* m, const float* w_ih, const float* w_hh,
                                     const float* gates, const float* dout, int64_t n, int64_t h,
                                     float* dx, float* dm, float* dgi, float* dgh, float* part,
                                     float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, void* stream)
{
    if (n < 0 || h <= 0) return GODE_E_SHAPE;
    if (!w_ih || !w_hh || (!dw_ih) != (!dw_hh)) return GODE_E_NULLPTR;
    if (n > INT32_MAX || h > 1024) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t h3 = 3 * h;
    if (n == 0 && !dw_ih) return GODE_E_SHAPE;         // partials only: there is nothing to write them from
    if (n == 0) {
        int rc = gode_zero_f32(dw_ih, h3 * 2 * h, stream); if (rc) return rc;
        rc = gode_zero_f32(dw_hh, h3 * h, stream); if (rc) return rc;
        if (db_ih) { rc = gode_zero_f32(db_ih, h3, stream); if (rc) return rc; }
        if (db_hh) { rc = gode_zero_f32(db_hh, h3, stream); if (rc) return rc; }
        return 0;
    }
    if (!x || !m || !gates || !dout || !dgi || !dgh || !part) return GODE_E_NULLPTR;
    const size_t lds = (size_t)RB * 7 * h * sizeof(float);
    if (lds > 160 * 1024) return GODE_E_UNSUPPORTED;
    const int64_t parts = gode_gru_wgrad_parts(n);
    if (parts > 65535) return GODE_E_RANGE;           // every argument check precedes the first launch
    int rc = gode_set_lds_once((const void*)gru_bwd_kernel, lds); if (rc) return rc;
    const int64_t blocks = (n + RB - 1) / RB;
    hipLaunchKernelGGL(gru_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, s, x, w_ih, w_hh, gates, dout, (int)n, (int)h,
                       dx, dm, dgi, dgh);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gru_wgrad_kernel, dim3((unsigned)h3, (unsigned)parts), dim3(256), 0, s, x, m, dgi, dgh, (int)n, (int)h, part);
    GODE_LAUNCH_CHECK();
    if (!dw_ih) return 0;                              // the caller sums the partials of several steps itself (gode_gru_wreduce_f32)
    int64_t rb = (h3 * (h3 + 2) + 255) / 256; if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(gru_wreduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, part, (int)parts, (int)h, dw_ih, dw_hh, db_ih, db_hh);
    GODE_LAUNCH_CHECK();
    return 0;
}

// dW_ih, dW_hh, db_ih, db_hh = sum of n_part partial rows (3h x (3h + 2) floats each, in order): the closing step of
// gode_gru_cell_f32_bwd on its own, for callers that let several applications of the SAME cell (QC/mpnn.py:30 inside
// `for t in range(T)`) write their partials back to back and sum them once
extern "C" int gode_gru_wreduce_f32(const float* part, int64_t n_part, int64_t h, float* dw_ih, float* dw_hh, float* db_ih,
                                    float* db_hh, void* stream)
{
    if (n_part <= 0 || h <= 0) return GODE_E_SHAPE;
    if (!part || !dw_ih || !dw_hh) return GODE_E_NULLPTR;
    if (n_part > INT32_MAX || h > 1024) return GODE_E_RANGE;
    const int64_t h3 = 3 * h;
    int64_t rb = (h3 * (h3 + 2) + 255) / 256; if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(gru_wreduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream, part, (int)n_part, (int)h, dw_ih, dw_hh,
                       db_ih, db_hh);
    GODE_LAUNCH_CHECK();
    return 0;
}
