// rk.hip — Runge-Kutta elementwise steps and reductions (gfx950).
//
// These replace the tensor arithmetic torchdiffeq performs between two calls of
// the ODE function (call site GCN/models.py:192 of the reference): stage input
// y + h*sum(a_ij k_j), solution combine y + h*sum(b_i k_i), the dopri5 error
// ratio and the initial-step norms.  One pass over each operand, 16 B per lane.
// Bound: HBM.  Algorithmic bytes: (n_terms + 1) * n * 4 for gode_lincomb_f32.
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

__global__ __launch_bounds__(256) void lincomb4_kernel(float* out, LinComb lc, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const float4 r = lc_load4(lc, i * 4);
        *reinterpret_cast<float4*>(out + i * 4) = r;
    }
}
// Up to four combinations of different lengths in ONE launch (the solution combine of an adjoint state [y, a, a_t, theta]
// on a launch-bound graph: four launches of ~4.4 us each per RK step otherwise); blockIdx.y selects the combination.
struct MultiLC { float* out[4]; LinComb lc[4]; int64_t n[4]; int vec[4]; };
__global__ __launch_bounds__(256) void lincomb_multi_kernel(MultiLC m) {
    const int c = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m.vec[c]) {
        for (const int64_t n4 = m.n[c] / 4; i < n4; i += stride) *reinterpret_cast<float4*>(m.out[c] + i * 4) = lc_load4(m.lc[c], i * 4);
    } else {
        for (; i < m.n[c]; i += stride) m.out[c][i] = lc_load1(m.lc[c], i);
    }
}
__global__ __launch_bounds__(256) void lincomb1_kernel(float* out, LinComb lc, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = lc_load1(lc, i);
}

__device__ __forceinline__ void block_reduce_store(double v, double* part) {
    __shared__ double sm[4];
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// MODE 0: dopri5 error ratio, tol = atol + rtol*max(|y0|,|y1|)
// MODE 1: scaled norm, tol = atol + rtol*|y0|
template <int MODE>
__global__ __launch_bounds__(256) void ratio_sumsq_kernel(double* part, LinComb lc, const float* y0,
                                                          const float* y1, float rtol, float atol, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (; i < n; i += stride) {
        const float e = lc_load1(lc, i);
        float m = fabsf(y0[i]);
        if (MODE == 0) m = fmaxf(m, fabsf(y1[i]));
        const float r = e / (atol + rtol * m);
        s += (double)r * (double)r;
    }
    block_reduce_store(s, part);
}

// 16-byte form of the same reduction (n % 4 == 0, aligned operands): all term loads of an iteration are in flight
// together; the scalar form above reaches 2 TB/s at 2^27 elements, this one streams.
template <int MODE>
__global__ __launch_bounds__(256) void ratio_sumsq4_kernel(double* part, LinComb lc, const float* y0,
                                                           const float* y1, float rtol, float atol, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (; i < n4; i += stride) {
        const float4 a = *reinterpret_cast<const float4*>(y0 + i * 4);
        float4 m = make_float4(fabsf(a.x), fabsf(a.y), fabsf(a.z), fabsf(a.w));
        if (MODE == 0) {
            const float4 b = *reinterpret_cast<const float4*>(y1 + i * 4);
            m = make_float4(fmaxf(m.x, fabsf(b.x)), fmaxf(m.y, fabsf(b.y)), fmaxf(m.z, fabsf(b.z)), fmaxf(m.w, fabsf(b.w)));
        }
        const float4 e = lc_load4(lc, i * 4);
        const float r0 = e.x / (atol + rtol * m.x), r1 = e.y / (atol + rtol * m.y);
        const float r2 = e.z / (atol + rtol * m.z), r3 = e.w / (atol + rtol * m.w);
        s += ((double)r0 * (double)r0 + (double)r1 * (double)r1) + ((double)r2 * (double)r2 + (double)r3 * (double)r3);
    }
    block_reduce_store(s, part);
}

// Up to four error-ratio sums in ONE pair of launches (an adaptive step of the adjoint state [y, a, a_t, theta] closes with
// four of them: eight launches of ~4 us otherwise).  Component c = blockIdx.y keeps the block decomposition it has on its
// own (nb[c] blocks, the same striding), so every sum is bit for bit the one gode_rk_errnorm_f32 forms.
struct MultiErr { LinComb lc[4]; const float* y0[4]; const float* y1[4]; int64_t n[4]; int nb[4]; int vec[4]; };
__global__ __launch_bounds__(256) void ratio_sumsq_multi_kernel(double* part, MultiErr m, float rtol, float atol) {
    const int c = blockIdx.y;
    if ((int)blockIdx.x >= m.nb[c]) return;
    const int64_t stride = (int64_t)m.nb[c] * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float* y0 = m.y0[c]; const float* y1 = m.y1[c];
    double s = 0.0;
    if (m.vec[c]) {
        for (const int64_t n4 = m.n[c] / 4; i < n4; i += stride) {
            const float4 a = *reinterpret_cast<const float4*>(y0 + i * 4);
            const float4 b = *reinterpret_cast<const float4*>(y1 + i * 4);
            const float4 mm = make_float4(fmaxf(fabsf(a.x), fabsf(b.x)), fmaxf(fabsf(a.y), fabsf(b.y)), fmaxf(fabsf(a.z), fabsf(b.z)),
                                          fmaxf(fabsf(a.w), fabsf(b.w)));
            const float4 e = lc_load4(m.lc[c], i * 4);
            const float r0 = e.x / (atol + rtol * mm.x), r1 = e.y / (atol + rtol * mm.y);
            const float r2 = e.z / (atol + rtol * mm.z), r3 = e.w / (atol + rtol * mm.w);
            s += ((double)r0 * (double)r0 + (double)r1 * (double)r1) + ((double)r2 * (double)r2 + (double)r3 * (double)r3);
        }
    } else {
        for (; i < m.n[c]; i += stride) {
            const float e = lc_load1(m.lc[c], i);
            const float mx = fmaxf(fabsf(y0[i]), fabsf(y1[i]));
            const float r = e / (atol + rtol * mx);
            s += (double)r * (double)r;
        }
    }
    __shared__ double sm[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)c * RED_BLOCKS + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ __launch_bounds__(256) void final_sum_multi_kernel(double* out, const double* part, MultiErr m) {
    __shared__ double sm[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < m.nb[c]; i += 256) s += part[(int64_t)c * RED_BLOCKS + i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = sm[0];
}

__global__ __launch_bounds__(256) void final_sum_kernel(double* out, const double* part, int nparts) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// out[j] (+)= scale * sum_p part[p][j].  Block = 32 outputs x 8 part-groups; group q adds parts
// q, q+8, ... (coalesced 128-B reads), the 8 group sums are added in a fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_parts_kernel(float* out, const float* part, int64_t n_part,
                                                           int64_t len, float scale, int accumulate,
                                                           float* out_b, const float* part_b) {
    __shared__ float sm[8][33];
    if (blockIdx.y == 1) { out = out_b; part = part_b; }      // second (out, part) pair of the same shape
    const int jj = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int64_t j = (int64_t)blockIdx.x * 32 + jj;
    float s = 0.f;
    if (j < len) {
        int64_t p = q;
        for (; p + 24 < n_part; p += 32) {
            const float a0 = part[p * len + j], a1 = part[(p + 8) * len + j];
            const float a2 = part[(p + 16) * len + j], a3 = part[(p + 24) * len + j];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; p < n_part; p += 8) s += part[p * len + j];
    }
    sm[q][jj] = s;
    __syncthreads();
    if (q == 0 && j < len) {
        float t = sm[0][jj];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sm[k][jj];
        t *= scale;
        out[j] = accumulate ? out[j] + t : t;
    }
}

// Several block-partial reductions of one adjoint stage in ONE launch (launch-bound graphs: the reduction launches of
// a stage - weight gradients, bias column sums, GroupNorm affine partials, time-row bookkeeping - cost more than the
// work).  Segment s owns blocks [blk0[s], blk0[s+1]); inside a segment the arithmetic is reduce_parts_kernel's (8
// part-groups, fixed order).  A segment may read a strided column subset of its partial rows (col0, col_stride).
// Time rows (segments with w_row0): outputs j < time_len are written as t * sum, and ONE extra block forms
// at = sum over those segments of sum_c (sum_p part[p][c]) * w_row0[c] from the partials themselves, so no block waits
// for another.
struct ReduceSegs {
    int n;
    float* out[GODE_MAX_REDUCE_SEGS];
    const float* part[GODE_MAX_REDUCE_SEGS];
    int64_t n_part[GODE_MAX_REDUCE_SEGS], ld[GODE_MAX_REDUCE_SEGS], col0[GODE_MAX_REDUCE_SEGS], cstride[GODE_MAX_REDUCE_SEGS],
            len[GODE_MAX_REDUCE_SEGS], time_len[GODE_MAX_REDUCE_SEGS];
    const float* w_row0[GODE_MAX_REDUCE_SEGS];
    int blk0[GODE_MAX_REDUCE_SEGS + 1];
    int time_block;             // index of the extra block, or -1
    float t; float* at_out;
};

__global__ __launch_bounds__(256) void reduce_segments_kernel(ReduceSegs g) {
    __shared__ float sm[8][33];
    const int jj = threadIdx.x & 31, q = threadIdx.x >> 5;
    if ((int)blockIdx.x == g.time_block) {         // the time-derivative block
        float acc = 0.f;
        for (int s = 0; s < g.n; ++s) {
            if (!g.w_row0[s]) continue;
            for (int64_t c0 = 0; c0 < g.time_len[s]; c0 += 32) {
                const int64_t c = c0 + jj;
                float v = 0.f;
                if (c < g.time_len[s])
                    for (int64_t p = q; p < g.n_part[s]; p += 8) v += g.part[s][p * g.ld[s] + g.col0[s] + c * g.cstride[s]];
                sm[q][jj] = v;
                __syncthreads();
                if (q == 0 && c < g.time_len[s]) {
                    float t = sm[0][jj];
#pragma unroll
                    for (int k = 1; k < 8; ++k) t += sm[k][jj];
                    acc = fmaf(t, g.w_row0[s][c], acc);
                }
                __syncthreads();
            }
        }
        if (q == 0) {                              // lanes 0..31 of wave 0 hold the per-column products
            for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (jj == 0) *g.at_out = acc;
        }
        return;
    }
    int s = 0;
    while (s + 1 < g.n && (int)blockIdx.x >= g.blk0[s + 1]) ++s;
    const int64_t j = (int64_t)((int)blockIdx.x - g.blk0[s]) * 32 + jj;
    const int64_t len = g.len[s], n_part = g.n_part[s], ld = g.ld[s];
    const float* part = g.part[s] + g.col0[s] + j * g.cstride[s];
    float v = 0.f;
    if (j < len) {
        int64_t p = q;
        for (; p + 24 < n_part; p += 32) {
            const float a0 = part[p * ld], a1 = part[(p + 8) * ld];
            const float a2 = part[(p + 16) * ld], a3 = part[(p + 24) * ld];
            v += a0; v += a1; v += a2; v += a3;
        }
        for (; p < n_part; p += 8) v += part[p * ld];
    }
    sm[q][jj] = v;
    __syncthreads();
    if (q == 0 && j < len) {
        float t = sm[0][jj];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sm[k][jj];
        if (g.w_row0[s] && j < g.time_len[s]) t *= g.t;
        g.out[s][j] = t;
    }
}

// 16-byte form for long outputs (the weight-gradient reduction: 512 partials x 16 512 floats at d = 128): a block owns 64
// outputs (16 lanes x float4, 256 contiguous bytes per partial row) and 16 part-groups stride the partials with four
// loads in flight each; the groups are added in a fixed order.
__global__ __launch_bounds__(256) void reduce_parts4_kernel(float* out, const float* part, int64_t n_part,
                                                            int64_t len, float scale, int accumulate,
                                                            float* out_b, const float* part_b) {
    __shared__ float4 sm[16][17];
    if (blockIdx.y == 1) { out = out_b; part = part_b; }
    const int jj = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t j = ((int64_t)blockIdx.x * 16 + jj) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < len) {
        int64_t p = q;
        for (; p + 48 < n_part; p += 64) {
            const float4 a0 = *reinterpret_cast<const float4*>(part + p * len + j);
            const float4 a1 = *reinterpret_cast<const float4*>(part + (p + 16) * len + j);
            const float4 a2 = *reinterpret_cast<const float4*>(part + (p + 32) * len + j);
            const float4 a3 = *reinterpret_cast<const float4*>(part + (p + 48) * len + j);
            s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
            s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
            s.x += a2.x; s.y += a2.y; s.z += a2.z; s.w += a2.w;
            s.x += a3.x; s.y += a3.y; s.z += a3.z; s.w += a3.w;
        }
        for (; p < n_part; p += 16) {
            const float4 a = *reinterpret_cast<const float4*>(part + p * len + j);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    sm[q][jj] = s;
    __syncthreads();
    if (q == 0 && j < len) {
        float4 t = sm[0][jj];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 a = sm[k][jj]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
        t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale;
        float4* o = reinterpret_cast<float4*>(out + j);
        if (accumulate) { const float4 c = *o; t.x += c.x; t.y += c.y; t.z += c.z; t.w += c.w; }
        *o = t;
    }
}

// column sums with 16-B loads: thread owns 4 columns, TPR = d/4 threads cover a row.
__global__ __launch_bounds__(256) void colsum4_kernel(float* part, const float* X, int64_t n_rows, int d, int64_t rpb) {
    const int tpr = d >> 2;
    const int rows_par = 256 / tpr;
    const int c4 = (threadIdx.x % tpr) * 4, ph = threadIdx.x / tpr;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = r0 + rpb < n_rows ? r0 + rpb : n_rows;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ph < rows_par) {
        int64_t r = r0 + ph;
        for (; r + 3 * rows_par < r1; r += 4 * rows_par) {
            const float4 a = *reinterpret_cast<const float4*>(X + r * d + c4);
            const float4 b = *reinterpret_cast<const float4*>(X + (r + rows_par) * d + c4);
            const float4 c = *reinterpret_cast<const float4*>(X + (r + 2 * rows_par) * d + c4);
            const float4 e = *reinterpret_cast<const float4*>(X + (r + 3 * rows_par) * d + c4);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
            s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w;
        }
        for (; r < r1; r += rows_par) {
            const float4 a = *reinterpret_cast<const float4*>(X + r * d + c4);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    __shared__ float4 sm[256];
    sm[threadIdx.x] = (ph < rows_par) ? s : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (ph == 0) {
        float4 t = sm[threadIdx.x];
        for (int p = 1; p < rows_par; ++p) {
            const float4 a = sm[p * tpr + threadIdx.x];
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
        }
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * d + c4) = t;
    }
}

// column sums, stage 1: block b sums rows [b*rpb, (b+1)*rpb) -> part[b][d]
__global__ __launch_bounds__(256) void colsum_kernel(float* part, const float* X, int64_t n_rows, int d, int64_t rpb) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = r0 + rpb < n_rows ? r0 + rpb : n_rows;
    // thread t owns column c = t % dpad, row phase t / dpad
    for (int c = threadIdx.x; c < d; c += 256) part[(int64_t)blockIdx.x * d + c] = 0.f;
    __syncthreads();
    if (d <= 256) {
        const int lanes_per_row = d;              // threads covering one row
        const int rows_par = 256 / lanes_per_row; // rows processed concurrently
        const int c = threadIdx.x % lanes_per_row;
        const int ph = threadIdx.x / lanes_per_row;
        float s = 0.f;
        if (ph < rows_par)
            for (int64_t r = r0 + ph; r < r1; r += rows_par) s += X[r * d + c];
        __shared__ float sm[256];
        sm[threadIdx.x] = (ph < rows_par) ? s : 0.f;
        __syncthreads();
        if (ph == 0) {
            float t = 0.f;
            for (int p = 0; p < rows_par; ++p) t += sm[p * lanes_per_row + c];
            part[(int64_t)blockIdx.x * d + c] = t;
        }
    } else {
        for (int c = threadIdx.x; c < d; c += 256) {
            float s = 0.f;
            for (int64_t r = r0; r < r1; ++r) s += X[r * d + c];
            part[(int64_t)blockIdx.x * d + c] = s;
        }
    }
}

int red_blocks(int64_t n) {
    int64_t b = (n + 1023) / 1024;
    if (b < 1) b = 1;
    if (b > RED_BLOCKS) b = RED_BLOCKS;
    return (int)b;
}

// Column sums of a WIDE, short matrix (the bias gradients of the QC edge encoder: 760 x 5329 and 760 x 2667 - the
// row-block kernels above would run on 3 blocks there: 330-550 us): a block owns 32 columns, its 8 row groups stride
// over the rows with eight loads in flight each and are added in a fixed order; writes the result itself.
__global__ __launch_bounds__(256) void colsum_wide_kernel(float* __restrict__ out, const float* __restrict__ X, int64_t n_rows, int d,
                                                         float scale, int accumulate) {
    __shared__ float sm[8][33];
    const int jj = threadIdx.x & 31, qq = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + jj;
    float s = 0.f;
    if (c < d) {
        int64_t r = qq;
        for (; r + 56 < n_rows; r += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = X[(r + 8 * u) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < n_rows; r += 8) s += X[r * d + c];
    }
    sm[qq][jj] = s;
    __syncthreads();
    if (qq == 0 && c < d) {
        float t = sm[0][jj];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sm[k][jj];
        out[c] = accumulate ? out[c] + scale * t : scale * t;
    }
}

int64_t colsum_blocks(int64_t n_rows) {
    int64_t b = (n_rows + 255) / 256;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return b;
}

}  // namespace

extern "C" int gode_lincomb_f32(float* out, const gode_lincomb_t* lc, int64_t n, void* stream) {
    if (n < 0) return GODE_E_SHAPE;
    if (n == 0) return 0;
    if (!out) return GODE_E_NULLPTR;
    int rc = check_lincomb(lc, true); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    LinComb d = make_lincomb(lc);
    if (n % 4 == 0 && !(((uintptr_t)out) & 15) && lincomb_aligned16(lc)) {
        const int64_t n4 = n / 4;
        int64_t blocks = (n4 + 255) / 256; if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(lincomb4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, d, n4);
    } else {
        int64_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(lincomb1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, d, n);
    }
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_lincomb_multi_f32(float* const* outs, const gode_lincomb_t* lcs, const int64_t* ns, int32_t count, void* stream) {
    if (!outs || !lcs || !ns) return GODE_E_NULLPTR;
    if (count < 1 || count > 4) return GODE_E_RANGE;
    MultiLC m;
    int64_t longest = 0;
    for (int c = 0; c < 4; ++c) { m.out[c] = nullptr; m.lc[c] = make_lincomb(nullptr); m.n[c] = 0; m.vec[c] = 0; }
    for (int c = 0; c < count; ++c) {
        if (ns[c] < 0) return GODE_E_SHAPE;
        if (ns[c] == 0) continue;
        if (!outs[c]) return GODE_E_NULLPTR;
        int rc = check_lincomb(&lcs[c], true); if (rc) return rc;
        m.out[c] = outs[c]; m.lc[c] = make_lincomb(&lcs[c]); m.n[c] = ns[c];
        m.vec[c] = (ns[c] % 4 == 0 && !(((uintptr_t)outs[c]) & 15) && lincomb_aligned16(&lcs[c])) ? 1 : 0;
        const int64_t work = m.vec[c] ? ns[c] / 4 : ns[c];
        if (work > longest) longest = work;
    }
    if (longest == 0) return 0;
    int64_t blocks = (longest + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lincomb_multi_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, (hipStream_t)stream, m);
    GODE_LAUNCH_CHECK();
    return 0;
}

// out[0..n) = 0 as a kernel launch (internal; declared in common.h).  hipMemsetAsync is avoided on purpose: inside a
// HIP-graph capture it becomes a memset node, and replays of captured solves with memset nodes returned non-finite
// GroupNorm gradients on this ROCm (DESIGN.md section 2, round-2 finding) - kernels only on the captured paths.
int gode_zero_f32(float* out, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (!out) return GODE_E_NULLPTR;
    LinComb none = make_lincomb(nullptr);
    int64_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lincomb1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, none, n);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gode_rk_errnorm_scratch_bytes(void) { return (int64_t)4 * RED_BLOCKS * sizeof(double); }   // four components (multi form)

extern "C" int gode_rk_errnorm_multi_f32(double* out, const float* const* y0, const float* const* y1, const gode_lincomb_t* elcs,
                                         const int64_t* ns, int32_t count, float rtol, float atol, void* scratch, void* stream) {
    if (!out || !y0 || !y1 || !elcs || !ns || !scratch) return GODE_E_NULLPTR;
    if (count < 1 || count > 4) return GODE_E_RANGE;
    MultiErr m;
    int nbmax = 0;
    for (int c = 0; c < 4; ++c) { m.lc[c] = make_lincomb(nullptr); m.y0[c] = m.y1[c] = nullptr; m.n[c] = 0; m.nb[c] = 0; m.vec[c] = 0; }
    for (int c = 0; c < count; ++c) {
        if (ns[c] <= 0) return GODE_E_SHAPE;
        if (!y0[c] || !y1[c]) return GODE_E_NULLPTR;
        int rc = check_lincomb(&elcs[c], true); if (rc) return rc;
        m.lc[c] = make_lincomb(&elcs[c]); m.y0[c] = y0[c]; m.y1[c] = y1[c]; m.n[c] = ns[c];
        m.nb[c] = red_blocks(ns[c]);                           // as gode_rk_errnorm_f32 decomposes this component
        m.vec[c] = (ns[c] % 4 == 0 && lincomb_aligned16(&elcs[c]) && !((((uintptr_t)y0[c]) | ((uintptr_t)y1[c])) & 15)) ? 1 : 0;
        if (m.nb[c] > nbmax) nbmax = m.nb[c];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ratio_sumsq_multi_kernel, dim3((unsigned)nbmax, (unsigned)count), dim3(256), 0, s, (double*)scratch, m, rtol, atol);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_sum_multi_kernel, dim3((unsigned)count), dim3(256), 0, s, out, (const double*)scratch, m);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_rk_errnorm_f32(double* out, const float* y0, const float* y1, const gode_lincomb_t* elc,
                                   float rtol, float atol, int64_t n, void* scratch, void* stream) {
    if (n <= 0) return GODE_E_SHAPE;
    if (!out || !y0 || !y1 || !scratch) return GODE_E_NULLPTR;
    int rc = check_lincomb(elc, true); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    if (n % 4 == 0 && lincomb_aligned16(elc) && !((((uintptr_t)y0) | ((uintptr_t)y1)) & 15))
        hipLaunchKernelGGL(ratio_sumsq4_kernel<0>, dim3(nb), dim3(256), 0, s, (double*)scratch, make_lincomb(elc),
                           y0, y1, rtol, atol, n / 4);
    else
        hipLaunchKernelGGL(ratio_sumsq_kernel<0>, dim3(nb), dim3(256), 0, s, (double*)scratch, make_lincomb(elc),
                           y0, y1, rtol, atol, n);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, out, (const double*)scratch, nb);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_rk_scaled_sumsq_f32(double* out, const gode_lincomb_t* lc, const float* y, float rtol,
                                        float atol, int64_t n, void* scratch, void* stream) {
    if (n <= 0) return GODE_E_SHAPE;
    if (!out || !y || !scratch) return GODE_E_NULLPTR;
    int rc = check_lincomb(lc, true); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    if (n % 4 == 0 && lincomb_aligned16(lc) && !(((uintptr_t)y) & 15))
        hipLaunchKernelGGL(ratio_sumsq4_kernel<1>, dim3(nb), dim3(256), 0, s, (double*)scratch, make_lincomb(lc),
                           y, y, rtol, atol, n / 4);
    else
        hipLaunchKernelGGL(ratio_sumsq_kernel<1>, dim3(nb), dim3(256), 0, s, (double*)scratch, make_lincomb(lc),
                           y, y, rtol, atol, n);
    GODE_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, out, (const double*)scratch, nb);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_reduce_parts_f32(float* out, const float* part, int64_t n_part, int64_t len, float scale,
                                     int accumulate, void* stream) {
    if (n_part < 0 || len < 0) return GODE_E_SHAPE;
    if (len == 0) return 0;
    if (!out || (n_part > 0 && !part)) return GODE_E_NULLPTR;
    if (len >= 4096 && n_part >= 64 && len % 4 == 0 && !((((uintptr_t)out) | ((uintptr_t)part)) & 15)) {
        hipLaunchKernelGGL(reduce_parts4_kernel, dim3((unsigned)((len / 4 + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                           out, part, n_part, len, scale, accumulate, (float*)nullptr, (const float*)nullptr);
        GODE_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((len + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       out, part, n_part, len, scale, accumulate, (float*)nullptr, (const float*)nullptr);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_reduce_segments_f32(const gode_reduce_seg_t* segs, int32_t n_segs, float t, float* at, void* stream) {
    if (!segs) return GODE_E_NULLPTR;
    if (n_segs < 1 || n_segs > GODE_MAX_REDUCE_SEGS) return GODE_E_RANGE;
    ReduceSegs g;
    g.n = n_segs;
    int blk = 0;
    bool any_time = false;
    for (int s = 0; s < GODE_MAX_REDUCE_SEGS; ++s) {
        g.out[s] = nullptr; g.part[s] = nullptr; g.w_row0[s] = nullptr;
        g.n_part[s] = g.ld[s] = g.col0[s] = g.len[s] = g.time_len[s] = 0; g.cstride[s] = 1;
    }
    for (int s = 0; s < n_segs; ++s) {
        const gode_reduce_seg_t& q = segs[s];
        if (!q.out || (q.n_part > 0 && !q.part)) return GODE_E_NULLPTR;
        if (q.len <= 0 || q.n_part < 0 || q.col_stride < 1 || q.col0 < 0 || q.ld < q.col0 + (q.len - 1) * q.col_stride + 1)
            return GODE_E_SHAPE;
        if (q.len > (int64_t)INT32_MAX * 16) return GODE_E_RANGE;
        g.out[s] = q.out; g.part[s] = q.part; g.n_part[s] = q.n_part; g.ld[s] = q.ld; g.col0[s] = q.col0;
        g.cstride[s] = q.col_stride; g.len[s] = q.len;
        if (q.w_row0 && q.time_len > 0) {
            if (!at || q.time_len > q.len) return q.time_len > q.len ? GODE_E_SHAPE : GODE_E_NULLPTR;
            g.w_row0[s] = q.w_row0; g.time_len[s] = q.time_len; any_time = true;
        }
        g.blk0[s] = blk;
        blk += (int)((q.len + 31) / 32);
    }
    for (int s = n_segs; s <= GODE_MAX_REDUCE_SEGS; ++s) g.blk0[s] = blk;
    g.time_block = any_time ? blk : -1;
    g.t = t; g.at_out = at;
    hipLaunchKernelGGL(reduce_segments_kernel, dim3((unsigned)(blk + (any_time ? 1 : 0))), dim3(256), 0, (hipStream_t)stream, g);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_reduce_parts2_f32(float* out_a, const float* part_a, float* out_b, const float* part_b,
                                      int64_t n_part, int64_t len, float scale, int accumulate, void* stream) {
    if (n_part < 0 || len < 0) return GODE_E_SHAPE;
    if (len == 0) return 0;
    if (!out_a || !out_b || (n_part > 0 && (!part_a || !part_b))) return GODE_E_NULLPTR;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((len + 31) / 32), 2), dim3(256), 0, (hipStream_t)stream,
                       out_a, part_a, n_part, len, scale, accumulate, out_b, part_b);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gode_colsum_scratch_bytes(int64_t n_rows, int64_t d) {
    return colsum_blocks(n_rows) * d * (int64_t)sizeof(float);
}

// stage 1 of gode_colsum_f32 alone: block partials into `scratch` (gode_colsum_scratch_bytes), their count in *n_parts
extern "C" int gode_colsum_parts_f32(const float* X, int64_t n_rows, int64_t d, float* scratch, int64_t* n_parts, void* stream) {
    if (n_rows < 0 || d <= 0) return GODE_E_SHAPE;
    if (!scratch || !n_parts || (n_rows > 0 && !X)) return GODE_E_NULLPTR;
    if (d > INT32_MAX) return GODE_E_RANGE;
    const int64_t nb = colsum_blocks(n_rows);
    const int64_t rpb = (n_rows + nb - 1) / nb;
    if (d % 4 == 0 && d <= 1024 && !(((uintptr_t)X) & 15) && !(((uintptr_t)scratch) & 15))
        hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, scratch, X, n_rows, (int)d, rpb > 0 ? rpb : 1);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, scratch, X, n_rows, (int)d, rpb > 0 ? rpb : 1);
    GODE_LAUNCH_CHECK();
    *n_parts = nb;
    return 0;
}

extern "C" int gode_colsum_f32(float* out, const float* X, int64_t n_rows, int64_t d, float scale,
                               int accumulate, float* scratch, void* stream) {
    if (n_rows < 0 || d <= 0) return GODE_E_SHAPE;
    if (!out || !scratch || (n_rows > 0 && !X)) return GODE_E_NULLPTR;
    if (d > INT32_MAX) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    if (d >= 512 && n_rows <= 4 * d) {               // wide and short: one launch, blocks over the columns
        hipLaunchKernelGGL(colsum_wide_kernel, dim3((unsigned)((d + 31) / 32)), dim3(256), 0, s, out, X, n_rows, (int)d, scale, accumulate);
        GODE_LAUNCH_CHECK();
        return 0;
    }
    const int64_t nb = colsum_blocks(n_rows);
    const int64_t rpb = (n_rows + nb - 1) / nb;
    if (d % 4 == 0 && d <= 1024 && !(((uintptr_t)X) & 15) && !(((uintptr_t)scratch) & 15))
        hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)nb), dim3(256), 0, s, scratch, X, n_rows, (int)d, rpb > 0 ? rpb : 1);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nb), dim3(256), 0, s, scratch, X, n_rows, (int)d, rpb > 0 ? rpb : 1);
    GODE_LAUNCH_CHECK();
    return gode_reduce_parts_f32(out, scratch, nb, d, scale, accumulate, stream);
}
