// convert.hip — index bookkeeping of a small assignment matrix in ONE launch (gfx950).
//
// A QC mini-batch (QC/datasets/utils.py:153-217: 20 molecules, ~360 atoms, ~760 directed edges) arrives every training
// step with new sizes, and three assignment vectors have to become CSR before the message kernels can run: edge -> target
// atom (Etgt), edge -> source atom (Esrc, for the backward pass), atom -> graph (batch, for the readouts).  With torch
// ops each conversion is a stable sort, a count, a prefix sum and a handful of casts - ~14 launches, ~40 per batch,
// more GPU time than the message round they prepare.  Here one workgroup does a conversion: counts by LDS integer
// atomics (exact, order-free), a block-wide exclusive scan, and the stable position of every entry from its rank among
// the earlier entries of the same row (a quadratic scan over LDS, sixteen entries per step: 290 K comparisons for 760 edges).
// Limits: n_entries <= 2048, n_rows <= 4096; larger matrices take the sort-based path (graph.csr_from_assignment).
#include "common.h"

namespace {

constexpr int kMaxEntries = 2048, kMaxRows = 4096, NT = 1024;

__global__ __launch_bounds__(NT) void assign_csr_kernel(const int64_t* __restrict__ index, int E, int n_rows,
                                                        int32_t* __restrict__ rowptr, int32_t* __restrict__ order,
                                                        const float* __restrict__ vals, float* __restrict__ vals_out)
{
    __shared__ __attribute__((aligned(16))) int idx[kMaxEntries];
    __shared__ int cnt[kMaxRows + 1];
    __shared__ int wsum[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int r = tid; r <= n_rows; r += NT) cnt[r] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += NT) {
        const int64_t v = index[e];
        const int r = (v >= 0 && v < n_rows) ? (int)v : -1;          // entries outside the matrix are dropped
        idx[e] = r;
        if (r >= 0) atomicAdd(&cnt[r], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0 .. n_rows): thread t owns rows 4t .. 4t + 3
    int c[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int r = 4 * tid + q; c[q] = r < n_rows ? cnt[r] : 0; s += c[q]; }
    int inc = s;                                                     // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0;
    for (int j = 0; j < w; ++j) base += wsum[j];
    int run = base + inc - s;
    __syncthreads();                                                 // every thread has read its counts
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * tid + q;
        if (r < n_rows) { cnt[r] = run; rowptr[r] = run; }
        run += c[q];
    }
    if (tid == NT - 1) rowptr[n_rows] = run;                         // = number of entries inside the matrix
    __syncthreads();
    // stable placement: position = start of the row + number of earlier entries of the same row
    for (int e = tid; e < E; e += NT) {
        const int r = idx[e];
        if (r < 0) continue;
        // sixteen entries per iteration from four independent 16-byte LDS reads (all lanes read the same address: a
        // broadcast): one entry per iteration was a chain of ~800 LDS round trips, 15 of the launch's 22 us
        int rank = 0;
        const int e16 = e & ~15;
        for (int k = 0; k < e16; k += 16) {
            const int4 a = *reinterpret_cast<const int4*>(idx + k), b = *reinterpret_cast<const int4*>(idx + k + 4);
            const int4 c2 = *reinterpret_cast<const int4*>(idx + k + 8), d2 = *reinterpret_cast<const int4*>(idx + k + 12);
            rank += (a.x == r) + (a.y == r) + (a.z == r) + (a.w == r) + (b.x == r) + (b.y == r) + (b.z == r) + (b.w == r);
            rank += (c2.x == r) + (c2.y == r) + (c2.z == r) + (c2.w == r) + (d2.x == r) + (d2.y == r) + (d2.z == r) + (d2.w == r);
        }
        for (int k = e16; k < e; ++k) rank += (idx[k] == r);
        const int pos = cnt[r] + rank;
        order[pos] = e;
        if (vals_out) vals_out[pos] = vals ? vals[e] : 1.f;
    }
}

// index[c] = first row r with M[r][c] != 0 (0 for an all-zero column): the reference collate's dense N x E target matrix
// (QC/datasets/utils.py:194-214, one entry per edge column) back to the per-edge target vector, on the device and without
// a validity check.  What `(M != 0).to(uint8).argmax(0)` computes in three library launches, 50-80 us for a 380 x 760 batch
// (the arg-max reduction walks the columns with a handful of threads).  A block owns 32 columns x 8 row slices: coalesced
// 128-byte row segments, eight rows per trip in flight, the slices' first hits combined through LDS.
__global__ __launch_bounds__(256) void dense_first_nonzero_kernel(const float* __restrict__ M, int64_t ld, int n_rows, int n_cols,
                                                                 int64_t* __restrict__ index)
{
    __shared__ int first[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    const int per = (n_rows + 7) / 8, r0 = sl * per, r1 = min(n_rows, r0 + per);
    const int cc = c < n_cols ? c : n_cols - 1;
    int f = INT32_MAX;
    for (int r = r0; r < r1; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = M[(int64_t)(r + u < r1 ? r + u : r1 - 1) * ld + cc];        // unconditional, clamped
#pragma unroll
        for (int u = 7; u >= 0; --u) if (r + u < r1 && v[u] != 0.f) f = min(f, r + u);
    }
    first[sl][cl] = f;
    __syncthreads();
    if (sl == 0 && c < n_cols) {
        int m = first[0][cl];
#pragma unroll
        for (int q = 1; q < 8; ++q) m = min(m, first[q][cl]);
        index[c] = m == INT32_MAX ? 0 : m;
    }
}

}  // namespace

extern "C" int gode_dense_first_nonzero_f32(const float* M, int64_t ld, int64_t n_rows, int64_t n_cols, int64_t* index, void* stream)
{
    if (n_rows < 0 || n_cols < 0 || ld < n_cols) return GODE_E_SHAPE;
    if (n_cols == 0) return 0;
    if (!index || (n_rows > 0 && !M)) return GODE_E_NULLPTR;
    if (n_rows > INT32_MAX - 16 || n_cols > INT32_MAX - 64) return GODE_E_RANGE;
    if (n_rows == 0) return GODE_E_SHAPE;                          // (torch raises on an arg-max over an empty dimension as well)
    hipLaunchKernelGGL(dense_first_nonzero_kernel, dim3((unsigned)((n_cols + 31) / 32)), dim3(256), 0, (hipStream_t)stream, M, ld,
                       (int)n_rows, (int)n_cols, index);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_assign_csr_supported(int64_t n_entries, int64_t n_rows) {
    return n_entries >= 0 && n_entries <= kMaxEntries && n_rows > 0 && n_rows <= kMaxRows;
}

extern "C" int gode_assign_csr_i32(const int64_t* index, int64_t n_entries, int64_t n_rows, int32_t* rowptr, int32_t* order,
                                   const float* vals, float* vals_out, void* stream)
{
    if (n_entries < 0 || n_rows <= 0) return GODE_E_SHAPE;
    if (!rowptr || (n_entries > 0 && (!index || !order))) return GODE_E_NULLPTR;
    if (!gode_assign_csr_supported(n_entries, n_rows)) return GODE_E_UNSUPPORTED;
    hipLaunchKernelGGL(assign_csr_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, index, (int)n_entries, (int)n_rows, rowptr, order,
                       vals, vals_out);
    GODE_LAUNCH_CHECK();
    return 0;
}
