// spmm.hip — CSR sparse aggregation  Z = A*X (+bias) with fused epilogue (gfx950).
//
// Replaces torch.spmm at GCN/layers.py:33,71 (+bias :35,73, relu GCN/models.py:178),
// GAT/layers.py:53,55 and QC/mpnn.py:29 / QC/layers.py:145 of the reference.
//
// Work decomposition: the host builds an nnz-balanced record list once per graph
// (graph_odenet_amd/graph.py): {row, begin, end, slot}.  A group of LPR lanes owns one
// record; each lane keeps VEC=4 consecutive feature columns, so a group reads one
// X row as LPR*16 contiguous bytes (d=128: 32 lanes x 16 B = 512 B per gathered row,
// two records per wave64).  Column indices / values of a record are loaded
// cooperatively (coalesced) and broadcast inside the group with ds_bpermute.
// Rows longer than the split length are cut into several records that write raw
// partial sums; spmm_finish adds them in record order (deterministic, no atomics).
//
// Bound: HBM.  Algorithmic bytes per launch: nnz*(4+4+4d) + (N+1)*4 + N*d*4.
#include "common.h"
#include "prof.h"

namespace {

struct Epilogue {
    const float* bias;
    int relu;
    float alpha;     // Y = (sum pre) + alpha * act(Z + bias)
    LinComb pre;     // n == 0: Y = alpha * act(Z + bias)
    LinComb cot;
    float* Y2;
    float* colpart;  // with Y2 (vec4 kernels, a thread group per record): block b writes the column sums of the Y2 rows it stored
                     // to colpart[b][d]; the finishing launch continues at row `colpart_row0`
    int64_t colpart_row0;
};

template <int LPR>
__device__ __forceinline__ float4 epilogue_store4(const Epilogue& ep, float4 z, int row, int lane,
                                                  int64_t d, float* Y, int64_t ldy) {
    if (ep.bias) {
        const float4 b = *reinterpret_cast<const float4*>(ep.bias + lane * 4);
        z.x += b.x; z.y += b.y; z.z += b.z; z.w += b.w;
    }
    float4 y = z;
    if (ep.relu) { y.x = fmaxf(z.x, 0.f); y.y = fmaxf(z.y, 0.f); y.z = fmaxf(z.z, 0.f); y.w = fmaxf(z.w, 0.f); }
    if (ep.pre.n > 0) {          // fused RK solution combine: rows of the pre-terms have ld = d
        const float4 p = lc_load4(ep.pre, (int64_t)row * d + lane * 4);
        y.x = fmaf(ep.alpha, y.x, p.x); y.y = fmaf(ep.alpha, y.y, p.y);
        y.z = fmaf(ep.alpha, y.z, p.z); y.w = fmaf(ep.alpha, y.w, p.w);
    } else if (ep.alpha != 1.f) {
        y.x *= ep.alpha; y.y *= ep.alpha; y.z *= ep.alpha; y.w *= ep.alpha;
    }
    *reinterpret_cast<float4*>(Y + (int64_t)row * ldy + lane * 4) = y;
    if (ep.Y2) {
        const int64_t o = (int64_t)row * d + lane * 4;
        float4 g = lc_load4(ep.cot, o);
        g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f;
        g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
        *reinterpret_cast<float4*>(ep.Y2 + o) = g;
        return g;
    }
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

// Column sums of the Y2 rows a block stored (the bias gradient of a layer is colsum of its masked cotangent: formed here
// the 2^20 x 128 array is never read again for it - the caller reduces gridDim.x partial rows instead, 1/8 of the bytes).
// Every thread of the block calls this (threads without a row pass zeros); rows are added in thread-group order.
template <int LPR>
__device__ __forceinline__ void block_colsum_store(const float4 g, float* __restrict__ colpart_row) {
    __shared__ float4 cs[256];
    cs[threadIdx.x] = g;
    __syncthreads();
    if (threadIdx.x < LPR) {
        float4 t = cs[threadIdx.x];
#pragma unroll
        for (int r = 1; r < 256 / LPR; ++r) {
            const float4 a = cs[r * LPR + threadIdx.x];
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
        }
        *reinterpret_cast<float4*>(colpart_row + 4 * threadIdx.x) = t;
    }
}

// d == 4*LPR, X/Y rows 16-byte aligned.
template <int LPR>
__global__ __launch_bounds__(256) void spmm_vec4_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const int4* __restrict__ items, int n_items, float* __restrict__ partial,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, Epilogue ep)
{
    constexpr int U = 4;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / LPR);
    const int lane = threadIdx.x & (LPR - 1);
    const bool active = gid < n_items;
    int row = 0, b = 0, e = 0, slot = -1;
    if (active) {
        if (items) { const int4 it = items[gid]; row = it.x; b = it.y; e = it.z; slot = it.w; }
        else { row = gid; b = rowptr[gid]; e = rowptr[gid + 1]; }
    }
    const int len = e - b;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Xl = X + lane * 4;

    for (int base = 0; __any(base < len); base += LPR) {
        int c = 0; float v = 0.f;
        if (base + lane < len) {
            c = col[b + base + lane];
            v = val ? val[b + base + lane] : 1.f;
        }
        const int cnt = len - base;   // may be <= 0 or > LPR
        for (int k = 0; k < LPR; k += U) {
            if (!__any(k < cnt)) break;
            int cc[U]; float vv[U]; float4 xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cc[u] = __shfl(c, k + u, LPR);
                vv[u] = __shfl(v, k + u, LPR);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + u < cnt && k + u < LPR)
                    xv[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)cc[u] * ldx);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc.x = fmaf(vv[u], xv[u].x, acc.x); acc.y = fmaf(vv[u], xv[u].y, acc.y);
                acc.z = fmaf(vv[u], xv[u].z, acc.z); acc.w = fmaf(vv[u], xv[u].w, acc.w);
            }
        }
    }
    if (ep.colpart) {                                  // (block-uniform) nobody leaves before the block's column sums
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {
            if (slot < 0) g = epilogue_store4<LPR>(ep, acc, row, lane, (int64_t)LPR * 4, Y, ldy);
            else *reinterpret_cast<float4*>(partial + (int64_t)slot * (LPR * 4) + lane * 4) = acc;
        }
        block_colsum_store<LPR>(g, ep.colpart + (int64_t)blockIdx.x * (LPR * 4));
        return;
    }
    if (!active) return;
    if (slot < 0) epilogue_store4<LPR>(ep, acc, row, lane, (int64_t)LPR * 4, Y, ldy);
    else *reinterpret_cast<float4*>(partial + (int64_t)slot * (LPR * 4) + lane * 4) = acc;
}

// Small-graph variant: a whole wave per record.  The wave's 64/LPR sub-groups of LPR lanes each take every
// (64/LPR)-th chunk of LPR nonzeros of the SAME record and their partial sums are added with xor-shuffles, so a
// long row costs len/(64/LPR) dependent gathers instead of len (on Cora at d=16 one 169-neighbour row was the whole
// 34 us of the launch) and no row needs the split / finish pass.  Used when there are too few records to fill the
// chip anyway (launch_vec4).
template <int LPR>
__global__ __launch_bounds__(256) void spmm_vec4_wave_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const int4* __restrict__ items, int n_items, float* __restrict__ partial,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, Epilogue ep)
{
    const int gid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);      // one record per wave
    const int wl = threadIdx.x & 63;
    const int lane = wl & (LPR - 1), sub = wl / LPR;
    if (gid >= n_items) return;                                                            // wave-uniform
    int row, b, e, slot = -1;
    if (items) { const int4 it = items[gid]; row = it.x; b = it.y; e = it.z; slot = it.w; }
    else { row = gid; b = rowptr[gid]; e = rowptr[gid + 1]; }
    const int len = e - b;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Xl = X + lane * 4;
    for (int base = sub * LPR; base - sub * LPR < len; base += 64) {                      // same trip count in every sub-group
        int c = 0; float v = 0.f;
        if (base + lane < len) {
            c = col[b + base + lane];
            v = val ? val[b + base + lane] : 1.f;
        }
        const int cnt = len - base;
        constexpr int U = LPR < 4 ? LPR : 4;
        for (int k = 0; k < LPR; k += U) {
            if (!__any(k < cnt)) break;
            int cc[U]; float vv[U]; float4 xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { cc[u] = __shfl(c, k + u, LPR); vv[u] = __shfl(v, k + u, LPR); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k + u < cnt) xv[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)cc[u] * ldx);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc.x = fmaf(vv[u], xv[u].x, acc.x); acc.y = fmaf(vv[u], xv[u].y, acc.y);
                acc.z = fmaf(vv[u], xv[u].z, acc.z); acc.w = fmaf(vv[u], xv[u].w, acc.w);
            }
        }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (sub != 0) return;
    if (slot < 0) epilogue_store4<LPR>(ep, acc, row, lane, (int64_t)LPR * 4, Y, ldy);
    else *reinterpret_cast<float4*>(partial + (int64_t)slot * (LPR * 4) + lane * 4) = acc;
}

template <int LPR>
__global__ __launch_bounds__(256) void spmm_finish_vec4_kernel(
    const int4* __restrict__ long_rows, int n_long, const float* __restrict__ partial,
    float* __restrict__ Y, int64_t ldy, Epilogue ep)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / LPR);
    const int lane = threadIdx.x & (LPR - 1);
    const bool active = gid < n_long;
    if (!active && !ep.colpart) return;
    const int4 lr = active ? long_rows[gid] : make_int4(0, 0, 0, 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = lr.y;
    for (; s + 3 < lr.z; s += 4) {          // 4 independent loads in flight, added in slot order
        const float4 p0 = *reinterpret_cast<const float4*>(partial + (int64_t)s * (LPR * 4) + lane * 4);
        const float4 p1 = *reinterpret_cast<const float4*>(partial + (int64_t)(s + 1) * (LPR * 4) + lane * 4);
        const float4 p2 = *reinterpret_cast<const float4*>(partial + (int64_t)(s + 2) * (LPR * 4) + lane * 4);
        const float4 p3 = *reinterpret_cast<const float4*>(partial + (int64_t)(s + 3) * (LPR * 4) + lane * 4);
        acc.x += p0.x; acc.y += p0.y; acc.z += p0.z; acc.w += p0.w;
        acc.x += p1.x; acc.y += p1.y; acc.z += p1.z; acc.w += p1.w;
        acc.x += p2.x; acc.y += p2.y; acc.z += p2.z; acc.w += p2.w;
        acc.x += p3.x; acc.y += p3.y; acc.z += p3.z; acc.w += p3.w;
    }
    for (; s < lr.z; ++s) {
        const float4 p = *reinterpret_cast<const float4*>(partial + (int64_t)s * (LPR * 4) + lane * 4);
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    if (ep.colpart) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) g = epilogue_store4<LPR>(ep, acc, lr.x, lane, (int64_t)LPR * 4, Y, ldy);
        block_colsum_store<LPR>(g, ep.colpart + (ep.colpart_row0 + blockIdx.x) * (LPR * 4));
        return;
    }
    epilogue_store4<LPR>(ep, acc, lr.x, lane, (int64_t)LPR * 4, Y, ldy);
}

// Generic path: any d, any alignment.  A group of G lanes (power of two <= 64) owns a record.
__device__ __forceinline__ void epilogue_store1(const Epilogue& ep, float z, int row, int c, int64_t d,
                                                float* Y, int64_t ldy) {
    if (ep.bias) z += ep.bias[c];
    float y = ep.relu ? fmaxf(z, 0.f) : z;
    if (ep.pre.n > 0) y = fmaf(ep.alpha, y, lc_load1(ep.pre, (int64_t)row * d + c));
    else y *= ep.alpha;
    Y[(int64_t)row * ldy + c] = y;
    if (ep.Y2) {
        const int64_t o = (int64_t)row * d + c;
        const float g = lc_load1(ep.cot, o);
        ep.Y2[o] = z > 0.f ? g : 0.f;
    }
}

__global__ __launch_bounds__(256) void spmm_generic_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const int4* __restrict__ items, int n_items, float* __restrict__ partial,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int d, int G, Epilogue ep)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / G);
    const int lane = threadIdx.x & (G - 1);
    if (gid >= n_items) return;
    int row, b, e, slot = -1;
    if (items) { const int4 it = items[gid]; row = it.x; b = it.y; e = it.z; slot = it.w; }
    else { row = gid; b = rowptr[gid]; e = rowptr[gid + 1]; }
    // widths that are not a multiple of 4 (h = 73 of the QC models; 7 classes arrive padded to 8 and never come here):
    // lanes over consecutive columns (coalesced row reads), four gathers in flight per lane, the sum in CSR order
    for (int c = lane; c < d; c += G) {
        float acc = 0.f;
        int j = b;
        for (; j + 4 <= e; j += 4) {
            const int c0 = col[j], c1 = col[j + 1], c2 = col[j + 2], c3 = col[j + 3];
            const float x0 = X[(int64_t)c0 * ldx + c], x1 = X[(int64_t)c1 * ldx + c], x2 = X[(int64_t)c2 * ldx + c],
                        x3 = X[(int64_t)c3 * ldx + c];
            float v0 = 1.f, v1 = 1.f, v2 = 1.f, v3 = 1.f;
            if (val) { v0 = val[j]; v1 = val[j + 1]; v2 = val[j + 2]; v3 = val[j + 3]; }
            acc = fmaf(v0, x0, acc); acc = fmaf(v1, x1, acc); acc = fmaf(v2, x2, acc); acc = fmaf(v3, x3, acc);
        }
        for (; j < e; ++j) {
            const float v = val ? val[j] : 1.f;
            acc = fmaf(v, X[(int64_t)col[j] * ldx + c], acc);
        }
        if (slot < 0) epilogue_store1(ep, acc, row, c, d, Y, ldy);
        else partial[(int64_t)slot * d + c] = acc;
    }
}

// One column (sparse matrix x vector: the source-side sum of the attention-logit cotangents): a wave per record, the
// lanes stride over its non-zeros (the generic kernel above would walk a 1 024-entry record with ONE thread).
__global__ __launch_bounds__(256) void spmv_wave_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const int4* __restrict__ items, int n_items, float* __restrict__ partial,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, Epilogue ep)
{
    const int gid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (gid >= n_items) return;
    int row, b, e, slot = -1;
    if (items) { const int4 it = items[gid]; row = it.x; b = it.y; e = it.z; slot = it.w; }
    else { row = gid; b = rowptr[gid]; e = rowptr[gid + 1]; }
    float acc = 0.f;
    for (int j = b + lane; j < e; j += 64) acc = fmaf(val ? val[j] : 1.f, X[(int64_t)col[j] * ldx], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
        if (slot < 0) epilogue_store1(ep, acc, row, 0, 1, Y, ldy);
        else partial[slot] = acc;
    }
}

__global__ __launch_bounds__(256) void spmm_finish_generic_kernel(
    const int4* __restrict__ long_rows, int n_long, const float* __restrict__ partial,
    float* __restrict__ Y, int64_t ldy, int d, int G, Epilogue ep)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = (int)(tid / G);
    const int lane = threadIdx.x & (G - 1);
    if (gid >= n_long) return;
    const int4 lr = long_rows[gid];
    for (int c = lane; c < d; c += G) {
        float acc = 0.f;
        for (int s = lr.y; s < lr.z; ++s) acc += partial[(int64_t)s * d + c];
        epilogue_store1(ep, acc, lr.x, c, d, Y, ldy);
    }
}

template <int LPR>
int launch_vec4(const int* rowptr, const int* col, const float* val, const int4* items, int n_items,
                const int4* long_rows, int n_long, float* partial, const float* X, int64_t ldx,
                float* Y, int64_t ldy, const Epilogue& ep, hipStream_t s) {
    if (LPR < 64 && n_items > 0 && n_items <= 65536) {         // too few records to fill the chip: a wave per record
        if (ep.colpart) return GODE_E_UNSUPPORTED;             // (gode_spmm_y2_colsum_rows is 0 for such a graph)
        const int64_t wb = ((int64_t)n_items * 64 + 255) / 256;
        const int slot = gode_prof_begin(s, (int64_t)LPR * 4, n_items, (int64_t)ep.pre.n + ep.cot.n + (ep.Y2 ? 1 : 0));
        hipLaunchKernelGGL(spmm_vec4_wave_kernel<LPR>, dim3((unsigned)wb), dim3(256), 0, s,
                           rowptr, col, val, items, n_items, partial, X, ldx, Y, ldy, ep);
        gode_prof_end(s, slot);
        GODE_LAUNCH_CHECK();
        if (n_long > 0) {
            const int64_t b2 = ((int64_t)n_long * LPR + 255) / 256;
            hipLaunchKernelGGL(spmm_finish_vec4_kernel<LPR>, dim3((unsigned)b2), dim3(256), 0, s,
                               long_rows, n_long, partial, Y, ldy, ep);
            GODE_LAUNCH_CHECK();
        }
        return 0;
    }
    const int64_t threads = (int64_t)n_items * LPR;
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 0) {
        const int slot = gode_prof_begin(s, (int64_t)LPR * 4, n_items, (int64_t)ep.pre.n + ep.cot.n + (ep.Y2 ? 1 : 0));
        hipLaunchKernelGGL(spmm_vec4_kernel<LPR>, dim3((unsigned)blocks), dim3(256), 0, s,
                           rowptr, col, val, items, n_items, partial, X, ldx, Y, ldy, ep);
        gode_prof_end(s, slot);
        GODE_LAUNCH_CHECK();
    }
    if (n_long > 0) {
        const int64_t b2 = ((int64_t)n_long * LPR + 255) / 256;
        Epilogue ep2 = ep;
        ep2.colpart_row0 = blocks;                             // its partial rows follow the main launch's
        hipLaunchKernelGGL(spmm_finish_vec4_kernel<LPR>, dim3((unsigned)b2), dim3(256), 0, s,
                           long_rows, n_long, partial, Y, ldy, ep2);
        GODE_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" int64_t gode_spmm_y2_colsum_rows(int64_t n_items, int64_t n_long, int64_t d)
{
    if (n_items <= 0 || n_long < 0 || d <= 0 || d % 4) return 0;
    const int64_t lpr = d / 4;
    if (lpr > 64 || (lpr & (lpr - 1))) return 0;              // the 16-byte-lane kernels only
    if (lpr < 64 && n_items <= 65536) return 0;                // small graphs run a wave per record: no per-block sums
    return (n_items * lpr + 255) / 256 + (n_long * lpr + 255) / 256;
}

extern "C" int gode_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                 const int32_t* items, int64_t n_items,
                                 const int32_t* long_rows, int64_t n_long, float* partial,
                                 const float* X, int64_t ldx, float* Y, int64_t ldy,
                                 int64_t n_rows, int64_t d,
                                 const gode_spmm_epilogue_t* epi, void* stream)
{
    if (n_rows < 0 || d <= 0 || ldx < d || ldy < d) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!rowptr || !col || !X || !Y) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX || d > (1 << 20)) return GODE_E_RANGE;
    if (!items) { n_items = n_rows; n_long = 0; }
    if (n_items < 0 || n_items > INT32_MAX || n_long < 0 || n_long > INT32_MAX) return GODE_E_RANGE;
    if (n_long > 0 && (!long_rows || !partial)) return GODE_E_NULLPTR;
    if (n_items * 64 >= ((int64_t)1 << 32)) return GODE_E_RANGE;      // a launch holds fewer than 2^32 threads
    const float* bias = epi ? epi->bias : nullptr;
    float* Y2 = epi ? epi->Y2 : nullptr;
    const gode_lincomb_t* cot = (epi && Y2) ? &epi->cot : nullptr;
    const gode_lincomb_t* pre = (epi && epi->pre.n > 0) ? &epi->pre : nullptr;
    if (Y2) { int rc = check_lincomb(cot, true); if (rc) return rc; }
    if (pre) { int rc = check_lincomb(pre, true); if (rc) return rc; }
    hipStream_t s = (hipStream_t)stream;

    Epilogue ep;
    ep.bias = bias; ep.relu = epi ? epi->relu : 0; ep.alpha = epi ? epi->alpha : 1.f;
    ep.pre = make_lincomb(pre); ep.cot = make_lincomb(cot); ep.Y2 = Y2;
    ep.colpart = (epi && Y2) ? epi->Y2_colsum : nullptr; ep.colpart_row0 = 0;
    if (ep.colpart && ((((uintptr_t)ep.colpart) & 15) || gode_spmm_y2_colsum_rows(n_items, n_long, d) == 0)) return GODE_E_UNSUPPORTED;

    const bool al = !(((uintptr_t)X) & 15) && !(((uintptr_t)Y) & 15) && (ldx % 4 == 0) && (ldy % 4 == 0) &&
                    (!bias || !(((uintptr_t)bias) & 15)) && (!partial || !(((uintptr_t)partial) & 15)) &&
                    (!Y2 || (!(((uintptr_t)Y2) & 15) && lincomb_aligned16(cot))) && lincomb_aligned16(pre);
    const int4* it4 = reinterpret_cast<const int4*>(items);
    const int4* lr4 = reinterpret_cast<const int4*>(long_rows);
    if (al && d % 4 == 0) {
        switch (d / 4) {
#define GODE_CASE(L) case L: return launch_vec4<L>(rowptr, col, val, it4, (int)n_items, lr4, (int)n_long, partial, X, ldx, Y, ldy, ep, s);
            GODE_CASE(1) GODE_CASE(2) GODE_CASE(4) GODE_CASE(8) GODE_CASE(16) GODE_CASE(32) GODE_CASE(64)
#undef GODE_CASE
            default: break;
        }
    }
    if (ep.colpart) return GODE_E_UNSUPPORTED;                 // unaligned operands: no per-block column sums on the generic kernels
    int G = 1; while (G < d && G < 64) G <<= 1;
    if (d == 1 && it4 != nullptr) {                    // record lists only: whole short rows are cheaper one thread each
        const int64_t blocks = ((int64_t)n_items * 64 + 255) / 256;
        const int slot = gode_prof_begin(s, 1, n_items, (int64_t)ep.pre.n + ep.cot.n + (ep.Y2 ? 1 : 0));
        hipLaunchKernelGGL(spmv_wave_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           rowptr, col, val, it4, (int)n_items, partial, X, ldx, Y, ldy, ep);
        gode_prof_end(s, slot);
        GODE_LAUNCH_CHECK();
    } else {
        const int64_t blocks = ((int64_t)n_items * G + 255) / 256;
        hipLaunchKernelGGL(spmm_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           rowptr, col, val, it4, (int)n_items, partial, X, ldx, Y, ldy, (int)d, G, ep);
        GODE_LAUNCH_CHECK();
    }
    if (n_long > 0) {
        const int64_t b2 = ((int64_t)n_long * G + 255) / 256;
        hipLaunchKernelGGL(spmm_finish_generic_kernel, dim3((unsigned)b2), dim3(256), 0, s,
                           lr4, (int)n_long, partial, Y, ldy, (int)d, G, ep);
        GODE_LAUNCH_CHECK();
    }
    return 0;
}
