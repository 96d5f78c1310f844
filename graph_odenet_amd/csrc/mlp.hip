// mlp.hip — the dense products of the QC edge encoder and its update step, gfx950.
//
// Replaces, for the hot path of the QC models (the EdgeEncoderMLP half):
//   * QC/layers.py:46-86  EdgeEncoderMLP = TransitionMLP(5 -> 2667 -> 5329): `self.f(self.linear(input))`,
//     `torch.mm(input, self.weight) + self.bias` on E ~ 760 edge rows, and their autograd - the largest products of a
//     QM9 training step (2 x 760 x 2667 x 5329 = 21.6 GFLOP each: H W2, dA W2^T, H^T dA);
//   * QC/train_egcn.py's `optimizer.step()` (torch.optim.Adam over 14.3 M parameters) as ONE launch.
//
// gode_gemm_f32: C = op(A) op(B) on the exact fp32 matrix instruction (v_mfma_f32_32x32x2_f32: bit for bit a k-ordered
// fmaf chain), 128 x 128 x 16 block tiles staged k-major in LDS (conflict-free operand reads), 4 waves of 64 x 64 each,
// next tile's global loads in registers while the current one multiplies; arbitrary M, N, K and leading dimensions
// (2667 and 5329 floats per row: rows are not 16-byte aligned, so operands are loaded as dwords; they live in L2 /
// Infinity Cache - 8, 16 and 57 MB - and the kernel is matrix-bound, not load-bound).  Fused epilogue: + bias[col],
// relu, or * (mask[row][col] > 0) (the relu mask of the hidden layer in dH = (dA W2^T) * [H > 0]).
// Bound: fp32 MFMA (157.3 TFLOP/s): 0.14 ms per product at the peak.
//
// gode_adam_f32: one launch over a table of (param, grad, exp_avg, exp_avg_sq, length) - torch.optim.Adam's update
// (L2 weight decay folded into the gradient, bias corrections from a device-resident step counter so that the launch
// can sit inside a HIP graph).  Bound: HBM (16 B read + 12 B written per parameter).
#include "common.h"
#include "dense_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Loads this thread's ROWS * BK / 256 elements of a (ROWS x BK) operand tile into registers / stores them k-major into LDS
// (row stride ROWS + 1 floats: both kinds of store and the operand reads are conflict-free).
//   KCONTIG: source is [row][k] with k contiguous (A of C = A B; A and B of C = A B^T)
//   !KCONTIG: source is [k][row] with row contiguous (B of C = A B; A and B of C = A^T B)
// Thread t, element j:  KCONTIG: k = t % BK, row = t / BK + (256 / BK) j - BK lanes read the 4 BK contiguous bytes of a
//                                row's k-tile (rows of 2667 / 5329 floats are only 4-byte aligned: dword loads).  (The
//                                first version read four lanes x 4 B at a 16-byte stride per row: 40-59 TFLOP/s against
//                                87 for the row-contiguous operands.)
//                       else   : row = (t & 31) + 32 (j % (ROWS/32)), k = (t >> 5) + 8 (j / (ROWS/32)) - coalesced
template <bool KCONTIG, int ROWS, int BK>
__device__ __forceinline__ void tile_coords(int j, int& row, int& k) {
    const int t = threadIdx.x;
    if (KCONTIG) { k = t % BK; row = t / BK + (256 / BK) * j; }
    else { row = (t & 31) + 32 * (j % (ROWS / 32)); k = (t >> 5) + 8 * (j / (ROWS / 32)); }
}
template <bool KCONTIG, int ROWS, int BK>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int64_t ld, int row0, int n_rows, int k0, int n_k,
                                          float (&v)[ROWS * BK / 256]) {
#pragma unroll
    for (int j = 0; j < ROWS * BK / 256; ++j) {
        int row, k;
        tile_coords<KCONTIG, ROWS, BK>(j, row, k);
        // unconditional loads from clamped coordinates (a load under a branch is waited for at the join, which would
        // put the wait in front of the matrix phase); out-of-range elements are zeroed afterwards
        const bool ok = row0 + row < n_rows && k0 + k < n_k;
        const int rr = row0 + row < n_rows ? row0 + row : n_rows - 1, kk = k0 + k < n_k ? k0 + k : n_k - 1;
        const int64_t off = KCONTIG ? (int64_t)rr * ld + kk : (int64_t)kk * ld + rr;
        const float x = src[off];
        v[j] = ok ? x : 0.f;
    }
}
template <bool KCONTIG, int ROWS, int BK>
__device__ __forceinline__ void tile_store(float* tile /* [BK][ROWS + 1] */, const float (&v)[ROWS * BK / 256]) {
#pragma unroll
    for (int j = 0; j < ROWS * BK / 256; ++j) {
        int row, k;
        tile_coords<KCONTIG, ROWS, BK>(j, row, k);
        tile[k * (ROWS + 1) + row] = v[j];
    }
}

// C[M x N] = opA(A) opB(B):  A_KC: A given as [M][K] (else [K][M]);  B_KC: B given as [N][K] (else [K][N]).
// MT / NTW = 32-row / 32-column tiles per wave: block tile (64 MT) x (64 NTW) - 128 x 128, 64 x 128 or 64 x 64, chosen by
// the launcher; BK = k depth of a staged tile (16 for the largest tile: 64 KB of static LDS; 32 otherwise - half the
// barriers per flop).
template <bool A_KC, bool B_KC, int MT, int NTW, int BK>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                          const float* __restrict__ bias, int relu,
                                                          const float* __restrict__ mask, int64_t ldmask,
                                                          int k_chunk, int64_t c_zstride)
{
    constexpr int TM = 64 * MT, TN = 64 * NTW;                    // block tile rows / columns
    // gridDim.z > 1: block z multiplies the k range [z k_chunk, (z + 1) k_chunk) and writes its raw partial product to
    // C + z c_zstride (gode_gemm_splitk_f32: tall contractions with a handful of output tiles)
    const int kbeg = blockIdx.z * k_chunk, kend = min(K, kbeg + k_chunk);
    C += blockIdx.z * c_zstride;
    constexpr int LDA = TM + 1, LDB = TN + 1;
    __shared__ float As[2][BK * LDA];
    __shared__ float Bs[2][BK * LDB];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wm = (wave >> 1) * 32 * MT, wn = (wave & 1) * 32 * NTW;   // this wave's (32 MT) x (32 NTW) part of the block tile
    const int li = l & 31, lk = l >> 5;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    float ra[TM * BK / 256], rb[TN * BK / 256];
    tile_load<A_KC, TM, BK>(A, lda, m0, M, kbeg, kend, ra);
    tile_load<B_KC, TN, BK>(B, ldb, n0, N, kbeg, kend, rb);
    tile_store<A_KC, TM, BK>(As[0], ra);
    tile_store<B_KC, TN, BK>(Bs[0], rb);
    __syncthreads();
    const int n_kt = (kend - kbeg + BK - 1) / BK;
    for (int kt = 0; kt < n_kt; ++kt) {
        const int cur = kt & 1;
        tile_load<A_KC, TM, BK>(A, lda, m0, M, kbeg + (kt + 1) * BK, kend, ra);     // next tile (zeros past the range): in flight
        tile_load<B_KC, TN, BK>(B, ldb, n0, N, kbeg + (kt + 1) * BK, kend, rb);     // during the matrix phase
        const float* as = As[cur] + lk * LDA + wm + li;
        const float* bs = Bs[cur] + lk * LDB + wn + li;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float av[MT], bv[NTW];
#pragma unroll
            for (int a = 0; a < MT; ++a) av[a] = as[2 * ks * LDA + 32 * a];
#pragma unroll
            for (int b = 0; b < NTW; ++b) bv[b] = bs[2 * ks * LDB + 32 * b];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NTW; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < n_kt) {
            tile_store<A_KC, TM, BK>(As[cur ^ 1], ra);            // the other buffer: last read in iteration kt - 1,
            tile_store<B_KC, TN, BK>(Bs[cur ^ 1], rb);            // which every wave left through the barrier below
        }
        __syncthreads();
    }
    // D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            const int col = n0 + wn + 32 * b + li;
            const float bvv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row < M && col < N) {
                    float v = acc[a][b][e] + bvv;
                    if (relu) v = fmaxf(v, 0.f);
                    if (mask) v = mask[(int64_t)row * ldmask + col] > 0.f ? v : 0.f;
                    C[(int64_t)row * ldc + col] = v;
                }
            }
        }
}

// ---- Adam -----------------------------------------------------------------------------------------------------------
struct AdamChunk { int32_t tensor; int32_t pad; int64_t start; };   // 16 B per chunk of kAdamChunk elements
constexpr int kAdamChunk = 8192;

__global__ void adam_tick_kernel(float* state /* [step, 1/bc1, 1/sqrt(bc2)] */, float beta1, float beta2) {
    const float step = state[0] + 1.f;
    state[0] = step;
    // bias corrections in double: 1 - beta^t loses every digit in float for small t * (1 - beta)
    state[1] = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    state[2] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
}

// torch.optim.Adam (no amsgrad, no maximize), single-tensor formula: g += wd * p; m.lerp_(g, 1 - b1);
// v = b2 v + (1 - b2) g g; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
// The tensor table travels BY VALUE in the kernel arguments (2.5 KB): nothing to upload when a gradient tensor was
// re-allocated, and a captured launch carries the addresses it was captured with.
__global__ __launch_bounds__(256) void adam_kernel(gode_adam_args_t a, const AdamChunk* __restrict__ chunks,
                                                   const float* __restrict__ state, float lr, float beta1, float beta2,
                                                   float eps, float wd)
{
    const AdamChunk ch = chunks[blockIdx.x];
    float* p = a.param[ch.tensor];
    const float* g = a.grad[ch.tensor];
    float* m = a.exp_avg[ch.tensor];
    float* v = a.exp_avg_sq[ch.tensor];
    const int64_t len = a.len[ch.tensor];
    const float step_size = lr * state[1], inv_bc2_sqrt = state[2];
    const int64_t end = ch.start + kAdamChunk < len ? ch.start + kAdamChunk : len;
    for (int64_t i = ch.start + threadIdx.x; i < end; i += 256) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        float mi = m[i];
        mi = mi + (gi - mi) * (1.f - beta1);
        const float vi = fmaf(1.f - beta2, gi * gi, beta2 * v[i]);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

}  // namespace

extern "C" int gode_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu,
                             const float* mask, int64_t ldmask, void* stream)
{
    if (M < 0 || N < 0 || K < 0) return GODE_E_SHAPE;
    if (M == 0 || N == 0) return 0;
    if (K == 0) return GODE_E_SHAPE;
    if (!A || !B || !C) return GODE_E_NULLPTR;
    if (M > INT32_MAX - 256 || N > INT32_MAX - 256 || K > INT32_MAX - 256) return GODE_E_RANGE;
    // A is M x K (trans_a = 0, row-major, lda >= K) or K x M (trans_a = 1, lda >= M); B is K x N (trans_b = 0, ldb >= N)
    // or N x K (trans_b = 1, ldb >= K)
    if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N || (mask && ldmask < N)) return GODE_E_SHAPE;
    // block tile 128 x 128, 64 x 128 or 64 x 64: the largest one that still gives two blocks per CU (a block is one wave
    // per SIMD: its barriers and load latencies hide under the co-resident block's matrix work), else the smallest.
    // Measured on the three products of a 760-edge batch (tools/dev/gemm_probe.py), 128^2 / 64x128 / 64^2:
    // H W2 0.314 / 0.264 / 0.268 ms, dA W2^T 0.615 / 0.415 / 0.305 ms, H^T dA 0.252 / 0.245 / 0.275 ms.
    auto blocks_of = [&](int64_t tm, int64_t tn) { return ((M + tm - 1) / tm) * ((N + tn - 1) / tn); };
    const int shape = blocks_of(128, 128) >= 500 ? 22 : (blocks_of(64, 128) >= 500 ? 12 : 11);
    const int64_t tm = shape == 22 ? 128 : 64, tn = shape == 11 ? 64 : 128;
    const dim3 grid((unsigned)((N + tn - 1) / tn), (unsigned)((M + tm - 1) / tm));
    if (grid.y > 65535) return GODE_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
#define GODE_GEMM2(AKC, BKC, MTV, NTV, BKV) hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, MTV, NTV, BKV>), grid, dim3(256), 0, s, A, lda, \
                                                               B, ldb, C, ldc, (int)M, (int)N, (int)K, bias, relu ? 1 : 0, mask, ldmask, \
                                                               (int)K, (int64_t)0)
#define GODE_GEMM(AKC, BKC) { if (shape == 22) GODE_GEMM2(AKC, BKC, 2, 2, 16); else if (shape == 12) GODE_GEMM2(AKC, BKC, 1, 2, 32); \
                              else GODE_GEMM2(AKC, BKC, 1, 1, 32); }
    if (!trans_a && !trans_b) GODE_GEMM(true, false)
    else if (!trans_a && trans_b) GODE_GEMM(true, true)
    else if (trans_a && !trans_b) GODE_GEMM(false, false)
    else GODE_GEMM(false, true)
#undef GODE_GEMM
#undef GODE_GEMM2
    GODE_LAUNCH_CHECK();
    return 0;
}

// Tall contractions with a handful of output tiles (weight gradients x^T dy of the QC models' small layers: 73 x 42 from
// 380 rows, 5 x 2667 from 760): one block per output tile walks the whole contraction, 12-24 dependent k-steps - 15-28 us
// for a few MFLOP.  Here the contraction is cut into parts of >= 64 (a multiple of 32) and block (x, y, z) writes the
// raw product of part z to part[z][M][N]; the caller adds the parts (gode_reduce_parts_f32: fixed order).
static int splitk_chunk(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K < 192) return 0;
    const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
    if (tiles > 64) return 0;
    int64_t parts = K / 64;
    if (parts > 16) parts = 16;
    while (parts > 1 && tiles * parts > 1024) --parts;
    if (parts < 2) return 0;
    return (int)((((K + parts - 1) / parts) + 31) & ~(int64_t)31);
}
extern "C" int64_t gode_gemm_splitk_parts(int64_t M, int64_t N, int64_t K) {
    const int ch = splitk_chunk(M, N, K);
    return ch > 0 ? (K + ch - 1) / ch : 1;
}
extern "C" int gode_gemm_splitk_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                    const float* B, int64_t ldb, float* part, float* C, void* stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return GODE_E_SHAPE;
    if (!A || !B || !part) return GODE_E_NULLPTR;
    if (M > INT32_MAX - 256 || N > INT32_MAX - 256 || K > INT32_MAX - 256) return GODE_E_RANGE;
    if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N)) return GODE_E_SHAPE;
    const int ch = splitk_chunk(M, N, K);
    if (ch <= 0) return GODE_E_UNSUPPORTED;                       // gode_gemm_splitk_parts says 1: use gode_gemm_f32
    const int64_t parts = (K + ch - 1) / ch;
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)parts);
    hipStream_t s = (hipStream_t)stream;
#define GODE_GEMMK(AKC, BKC) hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, 1, 1, 32>), grid, dim3(256), 0, s, A, lda, B, ldb, part, N, \
                                                (int)M, (int)N, (int)K, (const float*)nullptr, 0, (const float*)nullptr, (int64_t)0, ch, M * N)
    if (!trans_a && !trans_b) GODE_GEMMK(true, false);
    else if (!trans_a && trans_b) GODE_GEMMK(true, true);
    else if (trans_a && !trans_b) GODE_GEMMK(false, false);
    else GODE_GEMMK(false, true);
#undef GODE_GEMMK
    GODE_LAUNCH_CHECK();
    // C (nullable, M x N contiguous): the sum of the parts, by the library's fixed-order reduction - one call from the host
    // instead of two (the QC step is bound by the host's launch rate on a slow host)
    return C ? gode_reduce_parts_f32(C, part, parts, M * N, 1.f, 0, stream) : 0;
}

extern "C" int64_t gode_adam_chunk(void) { return kAdamChunk; }

extern "C" int gode_adam_tick_f32(float* state, float beta1, float beta2, void* stream)
{
    if (!state) return GODE_E_NULLPTR;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int gode_adam_f32(const gode_adam_args_t* args, int32_t n_tensors, const void* chunks, int64_t n_chunks,
                             const float* state, float lr, float beta1, float beta2, float eps, float weight_decay,
                             void* stream)
{
    if (n_chunks < 0 || n_tensors < 0 || n_tensors > GODE_ADAM_MAX_TENSORS) return GODE_E_SHAPE;
    if (!state || !args) return GODE_E_NULLPTR;
    if (n_chunks == 0) return 0;
    if (!chunks) return GODE_E_NULLPTR;
    if (n_chunks > INT32_MAX) return GODE_E_RANGE;
    for (int i = 0; i < n_tensors; ++i)
        if (args->len[i] > 0 && (!args->param[i] || !args->grad[i] || !args->exp_avg[i] || !args->exp_avg_sq[i])) return GODE_E_NULLPTR;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, *args,
                       (const AdamChunk*)chunks, state, lr, beta1, beta2, eps, weight_decay);
    GODE_LAUNCH_CHECK();
    return 0;
}
