// rect.hip — the RECTANGULAR dense products of a GraphConvolution layer (in_features != out_features) on the exact
// fp32 matrix instruction (v_mfma_f32_16x16x4_f32), gfx950.
//
// Replaces `support = torch.mm(input, self.weight)` at GCN/layers.py:32 of the reference and its autograd for the
// layers the square kernels of gemm.hip do not cover: Cora's 16 -> 7 output layer, the benchmark's 128 -> 16 output
// layer, a dense 500 -> 16 input layer (a 98.7 %-zero feature matrix goes through gode_spmm_csr_f32 instead:
// graph_odenet_amd/layers.py).  Three entry points:
//
//   gode_rect_gemm_f32      S  = X W        (n x K)(K x M)          -> n x M   (leading dimension lds >= M, pad columns zeroed)
//   gode_rect_gemm_nt_f32   dX = dS W^T     (n x M)(K x M)^T        -> n x K
//   gode_rect_wgrad_f32     dW = X^T dS     (n x K)^T (n x M)       -> block partials of K x M (gode_reduce_parts_f32)
//
// These products are tall and skinny (n up to 2^20 rows, K and M between 3 and a few thousand): every one of them is
// bound by reading its n x K or n x M operand ONCE from HBM (the matrix work is a few per cent of the fp32 MFMA peak);
// the kernels are laid out for that stream.  Products: a wave owns 16-row tiles; the streamed operand is loaded in the
// MEMORY layout (lane m: row m >> 2, 16-byte chunk m & 3 - one wave instruction = 16 runs of 64 contiguous bytes) and
// moved to the MFMA layout with one ds_bpermute per register (gemm.hip); the product is formed transposed (D[col][row])
// so that a lane ends with four consecutive output columns of one row; the small weight matrix is read through L1 / L2.
// Weight gradient: the reduction runs over the rows, four rows per MFMA; a lane reads float4 along the feature axis
// (16 lanes x 16 B = 256 contiguous bytes per row) and the four components feed four different output tiles, which
// only permutes the rows of dW inside a 64-row block (undone at the store).
#include "common.h"
#include "dense_common.h"

namespace {

constexpr int kRectBlocks = 1024;          // 4 blocks per CU (no LDS, < 64 registers)

__device__ __forceinline__ float4 to_f_layout(const float4 v, int src_x4) {
    return make_float4(__int_as_float(__builtin_amdgcn_ds_bpermute(src_x4, __float_as_int(v.x))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_x4, __float_as_int(v.y))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_x4, __float_as_int(v.z))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(src_x4, __float_as_int(v.w))));
}

// out[row][c_base + ...] = sum_k in[row][k] * Wop(k, col):  NT column tiles of 16 per wave pass.
//   TRANSW = false:  Wop(k, col) = W[k * M + col]     (S = X W;     K = inner = rows of W, M = columns of W = outputs)
//   TRANSW = true :  Wop(k, col) = W[col * Kin + k]   (dX = dS W^T; inner = columns of W, outputs = rows of W)
// VEC: in / out rows are 16-byte aligned and ld % 4 == 0 (16-byte accesses); else dword accesses, same lane layout.
template <int NT, bool TRANSW, bool VEC>
__global__ __launch_bounds__(256) void rect_gemm_kernel(const float* __restrict__ in, int64_t ld_in, int n_rows, int inner,
                                                        const float* __restrict__ W, int w_ld, int n_out,
                                                        float* __restrict__ out, int64_t ld_out, int out_cols /* columns written, >= n_out: pad = 0 */)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int r = l & 15, g = l >> 4;          // MFMA layout: row r, k-slot / column quad g
    const int mr = l >> 2, mg = l & 3;         // memory layout
    const int to_f = (4 * r + g) * 4, to_m = (mg * 16 + mr) * 4;
    const int c_base = blockIdx.y * (16 * NT);
    const int n_tiles = (n_rows + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        f32x4 acc[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // One 16-wide k chunk: the streamed operand in the memory layout (lane m: row m >> 2, floats 4 (m & 3) .. + 3 of
        // the chunk: four lanes cover the 64 contiguous bytes of a row; rows that are not 16-byte aligned - K = 3703,
        // 1433, 73 - take four dword loads per lane instead of one 16-byte load; the first version read unaligned
        // operands in the MFMA layout, adjacent lanes 4 K bytes apart: 0.68 ms for Citeseer's 3327 x 3703 input layer) and
        // the 4 x NT weight values of the chunk's MFMAs.  The NEXT chunk is requested before the current one is
        // multiplied: on a short operand (a QM9 mini-batch: 380 rows, K = 73 - every wave has ONE tile) the loop was a
        // chain of five load round trips, 34 us per launch.
        // Every load of a chunk is unconditional, from clamped coordinates (a load under a branch is waited for at the
        // join, which would put the wait in front of the MFMAs); out-of-range values are zeroed when they are used.
        float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
        float wn[4][NT];
        const int row = tile * 16 + mr, rowc = row < n_rows ? row : n_rows - 1;
        auto fetch = [&](int k0) {
            const int k = k0 + 4 * mg;
            const float* p = in + (int64_t)rowc * ld_in;
            if (VEC) {                                            // rows are 16-byte aligned and ld % 4 == 0: the clamped chunk lies inside the row
                const int k4 = k < inner ? k : ((inner - 1) & ~3);
                xn = ld4(p + k4);
            } else {
                xn.x = p[k < inner ? k : inner - 1]; xn.y = p[k + 1 < inner ? k + 1 : inner - 1];
                xn.z = p[k + 2 < inner ? k + 2 : inner - 1]; xn.w = p[k + 3 < inner ? k + 3 : inner - 1];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kk = k0 + 4 * g + c, kc = kk < inner ? kk : inner - 1;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int col = c_base + 16 * tt + r, cc = col < n_out ? col : n_out - 1;
                    wn[c][tt] = TRANSW ? W[(int64_t)cc * w_ld + kc] : W[(int64_t)kc * w_ld + cc];
                }
            }
        };
        fetch(0);
        for (int k0 = 0; k0 < inner; k0 += 16) {
            float4 xv = xn;
            {
                const int k = k0 + 4 * mg;
                const bool rok = row < n_rows;
                xv.x = (rok && k < inner) ? xv.x : 0.f; xv.y = (rok && k + 1 < inner) ? xv.y : 0.f;
                xv.z = (rok && k + 2 < inner) ? xv.z : 0.f; xv.w = (rok && k + 3 < inner) ? xv.w : 0.f;
            }
            xv = to_f_layout(xv, to_f);
            float wv[4][NT];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
                    wv[c][tt] = (k0 + 4 * g + c < inner && c_base + 16 * tt + r < n_out) ? wn[c][tt] : 0.f;
            fetch(k0 + 16 < inner ? k0 + 16 : k0);               // (the last chunk again: no load under a branch)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xb = c == 0 ? xv.x : (c == 1 ? xv.y : (c == 2 ? xv.z : xv.w));
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][tt], xb, acc[tt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const float4 o = to_f_layout(make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]), to_m);   // -> memory layout
            const int row = tile * 16 + mr, col = c_base + 16 * tt + 4 * mg;
            if (row < n_rows && col < out_cols) {
                // columns >= n_out come out as exact zeros (their weights were read as zero)
                float* p = out + (int64_t)row * ld_out + col;
                if (VEC && col + 3 < out_cols) *reinterpret_cast<float4*>(p) = o;
                else { p[0] = o.x; if (col + 1 < out_cols) p[1] = o.y; if (col + 2 < out_cols) p[2] = o.z; if (col + 3 < out_cols) p[3] = o.w; }
            }
        }
    }
}

// dW block partial: part[block][k][m] = sum over the block's rows of X[row][k] * dS[row][m], for the K-block
// blockIdx.y (64 features) and all M <= 16 * MT columns.  Lane (r, g) of a step holds row 4 step + g: float4
// X[row][64 kb + 4 r ..] (component q feeds output tile q, whose row index r stands for feature 64 kb + 4 r + q) and
// dS[row][16 mt + r].
template <int MT, bool VECX>
__global__ __launch_bounds__(256) void rect_wgrad_kernel(const float* __restrict__ X, int64_t ldx, int n_rows, int K,
                                                         const float* __restrict__ dS, int64_t ldds, int M,
                                                         float* __restrict__ part, int per_wave)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const int kb = blockIdx.y;
    const int kcol = 64 * kb + 4 * r;
    f32x4 acc[4][MT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[q][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n_steps = (n_rows + 3) / 4;
    for (int step = blockIdx.x * 4 + wave; step < n_steps; step += gridDim.x * 4) {
        const int row = 4 * step + g;
        const bool ok = row < n_rows;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && kcol < K) {
            const float* p = X + (int64_t)row * ldx + kcol;
            if (VECX && kcol + 3 < K) xv = ld4(p);
            else { xv.x = p[0]; if (kcol + 1 < K) xv.y = p[1]; if (kcol + 2 < K) xv.z = p[2]; if (kcol + 3 < K) xv.w = p[3]; }
        }
        float b[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = (ok && 16 * mt + r < M) ? dS[(int64_t)row * ldds + 16 * mt + r] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = q == 0 ? xv.x : (q == 1 ? xv.y : (q == 2 ? xv.z : xv.w));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[q][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[mt], acc[q][mt], 0, 0, 0);
        }
    }
    // acc[q][mt][e] is the entry (feature 64 kb + 4 (4 g + e) + q, column 16 mt + r)
    if (per_wave) {
        // few rows (a QM9 mini-batch: 380 atoms, 24 blocks): every wave writes its own partial - four times the partial
        // rows for the reduction launch, which reads 2 MB instead of 0.5, but no four-phase LDS reduction here (the
        // kernel was 20 us for a 4 MFLOP product)
        float* outw = part + ((int64_t)blockIdx.x * 4 + wave) * K * M;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = 64 * kb + 4 * (4 * g + e) + q, m = 16 * mt + r;
                    if (kk < K && m < M) outw[(int64_t)kk * M + m] = acc[q][mt][e];
                }
        return;
    }
    // block partial: the four waves add through LDS in wave order (fixed order: deterministic)
    __shared__ float red[64 * 16 * MT];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = (4 * (4 * g + e) + q) * (16 * MT) + 16 * mt + r;
                        red[idx] = (w == 0 ? 0.f : red[idx]) + acc[q][mt][e];
                    }
        }
        __syncthreads();
    }
    float* out = part + (int64_t)blockIdx.x * K * M;
    for (int idx = threadIdx.x; idx < 64 * 16 * MT; idx += 256) {
        const int kk = 64 * kb + idx / (16 * MT), m = idx % (16 * MT);
        if (kk < K && m < M) out[(int64_t)kk * M + m] = red[idx];
    }
}

int64_t rect_blocks(int64_t n_rows) {
    int64_t b = ((n_rows + 15) / 16 + 3) / 4;
    if (b < 1) b = 1;
    if (b > kRectBlocks) b = kRectBlocks;
    return b;
}

template <bool TRANSW>
int rect_launch(const float* in, int64_t ld_in, int64_t n_rows, int64_t inner, const float* W, int64_t w_ld, int64_t n_out,
                float* out, int64_t ld_out, int64_t out_cols, hipStream_t s)
{
    const bool vec = !(((uintptr_t)in) & 15) && !(((uintptr_t)out) & 15) && ld_in % 4 == 0 && ld_out % 4 == 0;
    const int64_t tiles = (out_cols + 15) / 16;
    const int nt = tiles >= 8 ? 8 : (tiles > 4 ? 8 : (tiles > 2 ? 4 : (tiles > 1 ? 2 : 1)));
    const dim3 grid((unsigned)rect_blocks(n_rows), (unsigned)((tiles + nt - 1) / nt));
#define GODE_RECT(NTV, VECV)                                                                                       \
    hipLaunchKernelGGL((rect_gemm_kernel<NTV, TRANSW, VECV>), grid, dim3(256), 0, s, in, ld_in, (int)n_rows, (int)inner, \
                       W, (int)w_ld, (int)n_out, out, ld_out, (int)out_cols)
#define GODE_RECT_NT(VECV) { if (nt == 8) GODE_RECT(8, VECV); else if (nt == 4) GODE_RECT(4, VECV);                 \
                             else if (nt == 2) GODE_RECT(2, VECV); else GODE_RECT(1, VECV); }
    if (vec) GODE_RECT_NT(true) else GODE_RECT_NT(false)
#undef GODE_RECT_NT
#undef GODE_RECT
    GODE_LAUNCH_CHECK();
    return 0;
}

int rect_check(const void* a, const void* b, const void* c, int64_t n_rows, int64_t K, int64_t M) {
    if (n_rows < 0 || K <= 0 || M <= 0) return GODE_E_SHAPE;
    if (n_rows > INT32_MAX - 64 || K > (1 << 20) || M > (1 << 20)) return GODE_E_RANGE;
    if (n_rows > 0 && (!a || !b || !c)) return GODE_E_NULLPTR;
    return 0;
}

}  // namespace

extern "C" int gode_rect_gemm_f32(const float* X, int64_t ldx, int64_t n_rows, int64_t K, const float* W, int64_t M,
                                  float* S, int64_t lds, void* stream)
{
    int rc = rect_check(X, W, S, n_rows, K, M); if (rc) return rc;
    if (ldx < K || lds < M) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    return rect_launch<false>(X, ldx, n_rows, K, W, M, M, S, lds, lds, (hipStream_t)stream);
}

extern "C" int gode_rect_gemm_nt_f32(const float* dS, int64_t ldds, int64_t n_rows, int64_t M, const float* W, int64_t K,
                                     float* dX, int64_t lddx, void* stream)
{
    int rc = rect_check(dS, W, dX, n_rows, K, M); if (rc) return rc;
    if (ldds < M || lddx < K) return GODE_E_SHAPE;
    if (n_rows == 0) return 0;
    return rect_launch<true>(dS, ldds, n_rows, M, W, M, K, dX, lddx, K, (hipStream_t)stream);
}

static int64_t rect_wgrad_blocks(int64_t n_rows) {
    int64_t b = ((n_rows + 3) / 4 + 3) / 4;            // 16 rows per block and pass
    if (b < 1) b = 1;
    if (b > 512) b = 512;
    return b;
}
static constexpr int64_t kPerWaveBlocks = 32;                 // up to 512 rows: a partial row per WAVE (see rect_wgrad_kernel)

extern "C" int64_t gode_rect_wgrad_parts(int64_t n_rows) {
    const int64_t b = rect_wgrad_blocks(n_rows);
    return b <= kPerWaveBlocks ? 4 * b : b;
}

extern "C" int gode_rect_wgrad_f32(const float* X, int64_t ldx, int64_t n_rows, int64_t K, const float* dS, int64_t ldds,
                                   int64_t M, float* part, void* stream)
{
    int rc = rect_check(X, dS, part, n_rows, K, M); if (rc) return rc;
    if (!part) return GODE_E_NULLPTR;
    if (ldx < K || ldds < M) return GODE_E_SHAPE;
    if (M > 128) return GODE_E_UNSUPPORTED;            // the caller splits wider outputs into column blocks
    const int64_t blocks = rect_wgrad_blocks(n_rows);
    const int per_wave = blocks <= kPerWaveBlocks ? 1 : 0;
    const dim3 grid((unsigned)blocks, (unsigned)((K + 63) / 64));
    const bool vecx = !(((uintptr_t)X) & 15) && ldx % 4 == 0;
    hipStream_t s = (hipStream_t)stream;
    const int mt = M > 64 ? 8 : (M > 32 ? 4 : (M > 16 ? 2 : 1));
#define GODE_RWG(MTV, VX) hipLaunchKernelGGL((rect_wgrad_kernel<MTV, VX>), grid, dim3(256), 0, s, X, ldx, (int)n_rows, (int)K, dS, ldds, (int)M, part, per_wave)
#define GODE_RWG_MT(VX) { if (mt == 8) GODE_RWG(8, VX); else if (mt == 4) GODE_RWG(4, VX); else if (mt == 2) GODE_RWG(2, VX); else GODE_RWG(1, VX); }
    if (vecx) GODE_RWG_MT(true) else GODE_RWG_MT(false)
#undef GODE_RWG_MT
#undef GODE_RWG
    GODE_LAUNCH_CHECK();
    return 0;
}

// the same followed by the fixed-order sum of the partials into dW (K x M, contiguous): one call from the host instead of two
extern "C" int gode_rect_wgrad_sum_f32(const float* X, int64_t ldx, int64_t n_rows, int64_t K, const float* dS, int64_t ldds,
                                       int64_t M, float* part, float* dW, void* stream)
{
    if (!dW) return GODE_E_NULLPTR;
    const int rc = gode_rect_wgrad_f32(X, ldx, n_rows, K, dS, ldds, M, part, stream);
    if (rc) return rc;
    if (n_rows == 0) return gode_zero_f32(dW, K * M, stream);
    return gode_reduce_parts_f32(dW, part, gode_rect_wgrad_parts(n_rows), K * M, 1.f, 0, stream);
}
