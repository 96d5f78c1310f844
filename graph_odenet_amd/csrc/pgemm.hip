// pgemm.hip — large fp32 products on the bf16 matrix cores from exact three-way cuts, gfx950.
//
// Replaces, for the QC edge encoder (QC/layers.py:46-86, EdgeEncoderMLP = TransitionMLP(5 -> 2667 -> 5329) on the ~760
// edge rows of a QM9 batch) the three 21.6 GFLOP products of a training step
//     A  = H W2          (forward,  760 x 2667 . 2667 x 5329)
//     dH = dA W2^T       (autograd, 760 x 5329 . 5329 x 2667)
//     dW2 = H^T dA       (autograd, 2667 x 760 . 760 x 5329)
// which csrc/mlp.hip's exact-fp32 MFMA kernel runs at 64-81 TFLOP/s (0.40-0.51 of the 157 TFLOP/s fp32 matrix peak:
// on gfx950 an fp32 MFMA wave keeps the other waves of its SIMD from issuing their loads and stores, so matrix time
// and memory time add).
//
// Arithmetic (as csrc/gemm_pc.hip): every fp32 operand is cut by TRUNCATION into three bf16 numbers, x = hi + mid + lo
// exactly (8 + 8 + 8 significant bits); a product is accumulated in fp32 on v_mfma_f32_16x16x32_bf16 from eight of the
// nine piece products - lo*lo, below 2^-32 of the product, is left out.  hi*hi goes into one accumulator, the seven
// smaller products (<= 2^-8 of it) into a second one, added once at the end: the large accumulator is rounded K/32
// times instead of 8 K/32 times.  The result is an fp32 result (tests: 3e-6 of max|C| against float64, the bar of the
// fp32 kernel).
//
// Two steps.  (1) gode_cut_bf16x3_f32: one streaming pass writes the three piece PLANES of a matrix, bf16 row-major with
// both dimensions zero-padded to multiples of 128 (rows of 2667 / 5329 floats are only 4-byte aligned; the planes' rows
// are 256-byte aligned, and the padding makes every tile of the product whole: no bounds logic in the loop).  A matrix
// is cut once and used by every product it enters (W2: forward and dH; H: forward and dW2; dA: dH and dW2).
// (2) gode_pgemm_bf16x3: C = op(A) op(B) from planes.  128 x 128 block tile, 32-deep k-steps, 512 threads = 8 waves
// (2 x 4, 64 x 32 each: 8 accumulator tiles x 2), two waves per SIMD.  Operand tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no registers, no VALU), three stages of 48 KB, the loads of step t + 2 in flight while step t
// multiplies (counted s_waitcnt vmcnt, one raw s_barrier per k-step).  An operand whose contraction index is contiguous
// in memory ([rows][k]) is read from LDS with ds_read_b128; one whose rows are contiguous ([k][rows]) with
// ds_read_b64_tr_b16, the transposing read - so all four op() combinations run from the same row-major planes.  Both LDS
// images are XOR-swizzled on the SOURCE address of the DMA (the destination of an LDS-DMA is lane-linear) and on the
// read address: conflict-free ds_read_b128 / ds_read_b64_tr_b16.  Per k-step and block: 512 MFMAs (2 048 matrix cycles
// per SIMD) for 48 KB of operand bytes; blocks are numbered so that the 32 blocks of an XCD share operand panels in
// its L2.  Bound: bf16 MFMA (8 x 2 M N K flop at 2.5 PFLOP/s: 69 us for the edge encoder's products).
// Fused epilogue as gode_gemm_f32: + bias[col], relu, * (mask[row][col] > 0).
#include "common.h"
#include "dense_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS char* lds_ptr;                    // every LDS access of the kernel goes through explicit LDS pointers

constexpr int TM = 128, TN = 128, BK = 32, STAGES = 3;
constexpr int PIECE_B = TM * BK * 2;             // 8 KB: one piece of one operand tile (either image)
constexpr int OPER_B = 3 * PIECE_B;              // 24 KB
constexpr int STAGE_B = 2 * OPER_B;              // 48 KB
constexpr int WM = 2, WN = 4;                    // wave grid
constexpr int MI = TM / WM / 16, NJ = TN / WN / 16;   // 4 x 2 accumulator tiles of 16 x 16 per wave

// ---- the cut ---------------------------------------------------------------------------------------------------------
// X[R][C] fp32 (leading dimension ld, any alignment) -> planes[3][R_pad][C_pad] bf16, zero outside R x C.
// A thread owns column pairs (2 bf16 = one dword per plane): loads and stores are lane-contiguous.
constexpr int CUT_PAIRS = 4;                     // column pairs per thread, 512 columns apart
__global__ __launch_bounds__(256) void cut_kernel(const float* __restrict__ X, int64_t ld, int R, int C,
                                                  unsigned* __restrict__ planes, int R_pad, int C_pad)
{
    const int row = blockIdx.x;                                   // grid.x: rows (up to 2^31 - 1), grid.y: column blocks
    const int c0 = blockIdx.y * (512 * CUT_PAIRS) + 2 * threadIdx.x;
    float v[CUT_PAIRS][2];
#pragma unroll
    for (int j = 0; j < CUT_PAIRS; ++j) {
        const int c = c0 + 512 * j;
        const bool in0 = row < R && c < C, in1 = row < R && c + 1 < C;
        const float a = X[in0 ? (int64_t)row * ld + c : 0], b = X[in1 ? (int64_t)row * ld + c + 1 : 0];
        v[j][0] = in0 ? a : 0.f;
        v[j][1] = in1 ? b : 0.f;
    }
    const int64_t plane = (int64_t)R_pad * C_pad / 2;             // dwords per plane
#pragma unroll
    for (int j = 0; j < CUT_PAIRS; ++j) {
        const int c = c0 + 512 * j;
        if (c >= C_pad) continue;
        unsigned h0, m0, l0, h1, m1, l1;
        split3_trunc(v[j][0], h0, m0, l0);
        split3_trunc(v[j][1], h1, m1, l1);
        const int64_t o = ((int64_t)row * C_pad + c) / 2;
        planes[o] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);               // {hi16(x0), hi16(x1)}: x0 at the lower address
        planes[plane + o] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
        planes[2 * plane + o] = __builtin_amdgcn_perm(l1, l0, 0x07060302u);
    }
}

// ---- LDS images ------------------------------------------------------------------------------------------------------
// KC image (contraction index contiguous in memory): [128 rows][32 k] bf16, 64-byte rows = four 16-byte chunks; chunk c
// of row r sits at position c ^ kc_swz((r >> 2) & 3): the four 16-lane groups of a ds_read_b128 (rows l & 15, chunk
// l >> 4) then touch sixteen different 16-byte bank slots each.
__device__ __forceinline__ int kc_swz(int q) { return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1); }      // 0, 2, 3, 1
// RC image (rows contiguous in memory): [32 k][128 rows] bf16, 256-byte rows = sixteen chunks; chunk ch of k-row kr sits
// at position ch ^ rc_swz(kr) - the dual-use image (b) of the CDNA4 guide: conflict-free ds_read_b64_tr_b16.
__device__ __forceinline__ int rc_swz(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }

// LDS reads are inline asm on purpose: hipcc orders a compiler-visible LDS read behind EVERY LDS-DMA still in flight
// (s_waitcnt vmcnt(0) in front of the first ds_read of a k-step), which would drain the two-steps-ahead pipeline.  The
// asm reads are ordered by hand: counted vmcnt + barrier before, s_waitcnt lgkmcnt(0) + sched_barrier after.
__device__ __forceinline__ bf16x8_t lds_read_b128(unsigned addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ s16x4 lds_read_tr16_b64(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <bool KC>
struct Operand {
    // per-lane element offset (in bf16 elements, relative to plane + tile origin) of the 16 bytes this lane's LDS-DMA
    // fetches: wave w fills bytes [1024 w, 1024 w + 1024) of a piece image
    static __device__ __forceinline__ int64_t dma_src(int wave, int lane, int64_t ld) {
        if (KC) {
            const int r = lane >> 2, cp = lane & 3;                       // row 16 w + r, position cp
            return (int64_t)(16 * wave + r) * ld + 8 * (cp ^ kc_swz((r >> 2) & 3));
        } else {
            const int kr = 4 * wave + (lane >> 4), cp = lane & 15;        // k-row kr, position cp
            return (int64_t)kr * ld + 8 * (cp ^ rc_swz(kr));
        }
    }
    // element step of the source per k-step
    static __device__ __forceinline__ int64_t k_step(int64_t ld) { return KC ? BK : (int64_t)BK * ld; }
    // byte offsets inside a piece image of this lane's reads for the 16-row subtile starting at row r0 (a multiple of 16):
    // KC: one ds_read_b128 (o0); RC: two ds_read_b64_tr_b16 (k-rows 8 g + q and 8 g + 4 + q)
    static __device__ __forceinline__ void read_offsets(int r0, int lane, unsigned& o0, unsigned& o1) {
        if (KC) {
            const int r = lane & 15, c = lane >> 4;
            o0 = (r0 + r) * 64 + ((c ^ kc_swz((r >> 2) & 3)) << 4);
            o1 = 0;
        } else {
            const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
            const int ch = (r0 >> 3) + (p4 >> 1);
            const int kr0 = 8 * g + q, kr1 = kr0 + 4;
            o0 = kr0 * 256 + ((ch ^ rc_swz(kr0)) << 4) + 8 * (p4 & 1);
            o1 = kr1 * 256 + ((ch ^ rc_swz(kr1)) << 4) + 8 * (p4 & 1);
        }
    }
    static __device__ __forceinline__ bf16x8_t frag(unsigned img, unsigned o0, unsigned o1) {
        if (KC) return lds_read_b128(img + o0);
        const s16x4 a = lds_read_tr16_b64(img + o0), b = lds_read_tr16_b64(img + o1);
        const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    }
};

__device__ __forceinline__ void dma16(const short* src, lds_ptr lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// ---- epilogues -------------------------------------------------------------------------------------------------------
// D layout of v_mfma_f32_16x16x32_bf16: col = lane & 15, row = 4 (lane >> 4) + e.  Epilogue operands are loaded from
// clamped coordinates (no load under a divergent branch); only the store is guarded.
__device__ __forceinline__ void store_tile(const f32x4 (&v)[MI][NJ], int m0, int n0, int wm, int wn, int lane,
                                           float* __restrict__ C, int64_t ldc, int M, int N, const float* __restrict__ bias,
                                           int relu, const float* __restrict__ mask, int64_t ldmask)
{
    const bool has_bias = bias != nullptr, has_mask = mask != nullptr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn + 16 * j + (lane & 15);
        const int colc = col < N ? col : N - 1;
        const float bv = has_bias ? bias[colc] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rb = m0 + wm + 16 * i + 4 * (lane >> 4);
            float mv[4] = {1.f, 1.f, 1.f, 1.f};
            if (has_mask) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rowc = rb + e < M ? rb + e : M - 1;
                    mv[e] = mask[(int64_t)rowc * ldmask + colc];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[i][j][e] + bv;
                if (relu) x = fmaxf(x, 0.f);
                x = mv[e] > 0.f ? x : 0.f;
                if (rb + e < M && col < N) C[(int64_t)(rb + e) * ldc + col] = x;
            }
        }
    }
}

// A partial tile (a block that multiplied only part of a tile's contraction) in the workspace: raw sums in accumulator
// order, 16 bytes per lane, lane-contiguous.  Slot 2 b + 0: the FIRST tile block b touched, slot 2 b + 1: its last.
constexpr int64_t SLOT_F = (int64_t)TM * TN;     // floats per slot
__device__ __forceinline__ f32x4* slot_ptr(float* ws, int64_t slot, int i, int j, int thread) {
    return reinterpret_cast<f32x4*>(ws + slot * SLOT_F) + ((i * NJ + j) * 512 + thread);
}

// block of the launch whose share [W b / nwg, W (b + 1) / nwg) of the W = ntile * nk k-steps holds k-step x
__device__ __forceinline__ int owner_of(int64_t x, int64_t W, int nwg) { return (int)(((x + 1) * nwg - 1) / W); }

// C[M x N] = op(A) op(B) from piece planes.  A_KC: A stored [M][K] (else [K][M]);  B_KC: B stored [N][K] (else [K][N]).
// lda / ldb: padded row lengths of the planes (elements); pa / pb: elements per plane.  nk: k-steps of 32 (the planes are
// zero beyond K).  PRODUCTS: 8, or 6 (mid*lo and lo*mid, 2^-24 of a product each, left out as well).
//
// Work division: the ntile * nk k-steps of the product, tiles in m-fastest order and k inside a tile, are dealt in equal
// CONTIGUOUS shares to the gridDim.x blocks.  gridDim.x = ntile: a block per tile, the plain case.  gridDim.x = number of
// CUs when the tiles do not fill whole rounds of the chip (294 tiles on 256 CUs: the second round is 85 % idle; 147
// tiles: 43 % of the chip idle): every block then multiplies the same number of k-steps, a share may start and end in
// the middle of a tile, whole tiles in between go straight to C and the (at most two) partial ones to workspace slots,
// which pgemm_finish_kernel adds in block order - a fixed order, results do not depend on timing.  The LDS pipeline
// runs through tile changes (the loads of the next tile's first steps are in flight during an epilogue).
template <bool A_KC, bool B_KC, int PRODUCTS>
__global__ __launch_bounds__(512, 2) void pgemm_kernel(const short* __restrict__ Ap, int64_t lda, int64_t pa,
                                                       const short* __restrict__ Bp, int64_t ldb, int64_t pb,
                                                       float* __restrict__ C, int64_t ldc, int M, int N, int nk,
                                                       int tiles_m, int tiles_n, float* __restrict__ ws,
                                                       const float* __restrict__ bias, int relu,
                                                       const float* __restrict__ mask, int64_t ldmask)
{
    extern __shared__ __attribute__((aligned(1024))) char lds_generic[];  // STAGES x 48 KB, the only LDS object
    lds_ptr const lds = (lds_ptr)lds_generic;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // share of this block: ids are dealt to the XCDs round-robin by the hardware (b and b + 8 share an L2); renumber so
    // that one XCD works on CONSECUTIVE shares: its blocks share the B panel of a column tile and re-use the A panels of
    // all row tiles (speed only - any placement is correct)
    const int ntile = tiles_m * tiles_n, nwg = (int)gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int64_t W = (int64_t)ntile * nk;
    const bool per_tile = nwg == ntile;                            // (exact, and no 64-bit overflow for huge tile counts)
    const int64_t f0 = per_tile ? (int64_t)wid * nk : W * wid / nwg;
    const int n = per_tile ? nk : (int)(W * (wid + 1) / nwg - f0);  // k-steps of this block (block-uniform)
    if (n <= 0) return;
    const int wm = (wave / WN) * (TM / WM), wn = (wave % WN) * (TN / WN);

    // LDS-DMA sources: piece p of A / B, this lane's 16 bytes of the wave's 1 KB slice; a cursor (tile it, k-step ik) runs
    // two k-steps ahead of the multiplication and steps over tile ends
    const int64_t adma = Operand<A_KC>::dma_src(wave, lane, lda), bdma = Operand<B_KC>::dma_src(wave, lane, ldb);
    const int64_t astep = Operand<A_KC>::k_step(lda), bstep = Operand<B_KC>::k_step(ldb);
    lds_ptr const wbase = lds + 1024 * wave;
    int it = (int)(f0 / nk), ik = (int)(f0 - (int64_t)it * nk);
    const short *ia, *ib;
    auto seek = [&](int t, int k) {
        const int tn = t / tiles_m, tm = t - tn * tiles_m;
        ia = Ap + (A_KC ? (int64_t)tm * TM * lda : (int64_t)tm * TM) + adma + k * astep;
        ib = Bp + (B_KC ? (int64_t)tn * TN * ldb : (int64_t)tn * TN) + bdma + k * bstep;
    };
    seek(it, ik);
    auto issue = [&](int stage) {
        lds_ptr d = wbase + stage * STAGE_B;
#pragma unroll
        for (int p = 0; p < 3; ++p) dma16(ia + p * pa, d + p * PIECE_B);
#pragma unroll
        for (int p = 0; p < 3; ++p) dma16(ib + p * pb, d + OPER_B + p * PIECE_B);
        ia += astep;
        ib += bstep;
        if (++ik == nk) { ik = 0; ++it; seek(it < ntile ? it : 0, 0); }       // no loads in this branch
    };

    f32x4 big[MI][NJ], small[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) { big[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; small[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // byte offsets of this lane's operand reads inside a piece image (constant over the k loop)
    unsigned ao[MI][2], bo[NJ][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) Operand<A_KC>::read_offsets(wm + 16 * i, lane, ao[i][0], ao[i][1]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) Operand<B_KC>::read_offsets(wn + 16 * j, lane, bo[j][0], bo[j][1]);
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= (WM * WN) / 2;       // wave-uniform by construction

    // the multiplication's cursor: tile ct, k-step ck, first k-step of this block in the tile k_lo, tiles seen so far seg
    int ct = (int)(f0 / nk), ck = (int)(f0 - (int64_t)ct * nk), k_lo = ck, seg = 0;
    issue(0);
    if (n > 1) issue(1);
    for (int s = 0; s < n; ++s) {
        // this wave's share of step s has landed (the six DMAs of step s + 1 may stay in flight) ...
        if (s + 1 < n) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and after the barrier everybody's has, and everybody has finished reading step s - 1, whose buffer the
        // DMAs of step s + 2 may now overwrite
        __builtin_amdgcn_s_barrier();
        // The two waves of a SIMD (w and w + 4) leave the barrier together and would spend the same ~600 cycles issuing
        // their six DMAs while the matrix pipe idles (counters, profiles/r04_pgemm_pmc.txt: pipe busy 51 % of the launch).
        // Waves 0-3 issue theirs now, waves 4-7 after the MFMAs of their second slice: the partner multiplies meanwhile.
        if (!late && s + 2 < n) issue((s + 2) % STAGES);
        const unsigned st = lds_base + (s % STAGES) * STAGE_B;
        bf16x8_t a[MI][3], b[NJ][3];
        // operand reads run one 16-row slice of A ahead of the MFMAs: all of B and slice 0, then slice i + 1 is requested
        // before the 16 MFMAs of slice i are issued
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = Operand<B_KC>::frag(st + OPER_B + p * PIECE_B, bo[j][0], bo[j][1]);
#pragma unroll
        for (int p = 0; p < 3; ++p) a[0][p] = Operand<A_KC>::frag(st + p * PIECE_B, ao[0][0], ao[0][1]);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < MI) {
#pragma unroll
                for (int p = 0; p < 3; ++p) a[i + 1][p] = Operand<A_KC>::frag(st + p * PIECE_B, ao[i + 1][0], ao[i + 1][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 sm = small[i][j];
                if (PRODUCTS == 8) {
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[j][1], sm, 0, 0, 0);      // lo*mid
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][2], sm, 0, 0, 0);      // mid*lo
                }
                sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[j][0], sm, 0, 0, 0);          // lo*hi
                sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][2], sm, 0, 0, 0);          // hi*lo
                sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][1], sm, 0, 0, 0);          // mid*mid
                sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][0], sm, 0, 0, 0);          // mid*hi
                sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][1], sm, 0, 0, 0);          // hi*mid
                small[i][j] = sm;
                big[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][0], big[i][j], 0, 0, 0);   // hi*hi
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (late && i == MI / 2 - 1 && s + 2 < n) {
                issue((s + 2) % STAGES);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++ck < nk && s + 1 < n) continue;
        // end of this block's part of tile ct: k-steps [k_lo, ck) of it
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) big[i][j] += small[i][j];
        if (k_lo == 0 && ck == nk) {
            const int tn = ct / tiles_m, tm = ct - tn * tiles_m;
            store_tile(big, tm * TM, tn * TN, wm, wn, lane, C, ldc, M, N, bias, relu, mask, ldmask);
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) *slot_ptr(ws, 2 * (int64_t)wid + (seg > 0), i, j, threadIdx.x) = big[i][j];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) { big[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; small[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        ++ct; ck = 0; k_lo = 0; ++seg;
        // (The epilogue's stores share the vector-memory counter with the DMAs in flight.  The counted wait of the next
        // step stays correct: loads complete in order among loads and the DMAs of step s + 1 are older than those of
        // s + 2, so "at most six operations outstanding" still implies that step s + 1 has landed - stores can only make
        // the wait longer, never shorter.)
    }
}

// The tiles that more than one block multiplied: C tile = sum of the blocks' partial tiles, in block order, + epilogue.
// One block per tile, same thread -> element map as the product kernel; tiles a single block finished return at once.
__global__ __launch_bounds__(512) void pgemm_finish_kernel(float* __restrict__ ws, int nwg, int nk, int tiles_m, int tiles_n,
                                                           float* __restrict__ C, int64_t ldc, int M, int N,
                                                           const float* __restrict__ bias, int relu,
                                                           const float* __restrict__ mask, int64_t ldmask)
{
    const int t = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t W = (int64_t)tiles_m * tiles_n * nk, x0 = (int64_t)t * nk;
    const int b_lo = owner_of(x0, W, nwg), b_hi = owner_of(x0 + nk - 1, W, nwg);
    if (b_lo == b_hi) return;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = b_lo; b <= b_hi; ++b) {
        const int64_t slot = 2 * (int64_t)b + (W * b / nwg < x0);       // a block that began in an earlier tile: its last tile
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] += *slot_ptr(ws, slot, i, j, threadIdx.x);
    }
    const int tn = t / tiles_m, tm = t - tn * tiles_m;
    const int wm = (wave / WN) * (TM / WM), wn = (wave % WN) * (TN / WN);
    store_tile(acc, tm * TM, tn * TN, wm, wn, lane, C, ldc, M, N, bias, relu, mask, ldmask);
}

// Blocks of the launch (measured, tools/dev/pgemm_bench.py --sweep, 256 CUs; every partial tile costs a 64 KB store, its
// share of the finishing launch and a tile change inside the block, so splitting has to win back ~25 us):
//   a block per tile            when the tiles fill their rounds of the chip to >= 80 % (252 / 210 of 256: 151 / ~150 us),
//                               when there is no workspace, or when a block would walk through more than 1.5 tiles
//                               (H^T dA, 882 tiles of 24 k-steps: 176 us against 191 with shares);
//   s blocks per tile           the contraction cut in s ALIGNED parts when that fills ONE round to >= 80 % (dA W2^T at
//                               126 tiles: 2 x 126 blocks, 147 us against 168 with free shares - one partial per block);
//   one block per CU, equal shares of the k-steps
//                               otherwise (294 tiles: 193 us against ~230 for two rounds; 147 tiles: 197 against 230).
int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}
int choose_blocks(int64_t ntile, int64_t nk, bool have_ws) {
    const int64_t cus = cu_count(), rounds = (ntile + cus - 1) / cus;
    if (!have_ws || ntile * 100 >= rounds * cus * 80 || ntile * nk < cus * 8 || ntile > (int64_t)1 << 24) return (int)ntile;
    for (int64_t sp = 8; sp >= 2; --sp)
        if (ntile * sp <= cus && ntile * sp * 100 >= cus * 80 && nk / sp >= 16) return (int)(ntile * sp);
    if (ntile * 2 > cus * 3) return (int)ntile;
    return (int)cus;
}

template <bool A_KC, bool B_KC>
int launch(int products, const short* Ap, int64_t lda, int64_t pa, const short* Bp, int64_t ldb, int64_t pb, float* C,
           int64_t ldc, int M, int N, int nk, const float* bias, int relu, const float* mask, int64_t ldmask, float* ws,
           hipStream_t s)
{
    const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
    const size_t lds = (size_t)STAGES * STAGE_B;
    const int nwg = choose_blocks((int64_t)tiles_m * tiles_n, nk, ws != nullptr);
    const dim3 grid(nwg), block(512);
    if (products == 6) {
        auto* k = pgemm_kernel<A_KC, B_KC, 6>;
        const int rc = gode_set_lds_once((const void*)k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, block, lds, s, Ap, lda, pa, Bp, ldb, pb, C, ldc, M, N, nk, tiles_m, tiles_n, ws, bias, relu, mask, ldmask);
    } else {
        auto* k = pgemm_kernel<A_KC, B_KC, 8>;
        const int rc = gode_set_lds_once((const void*)k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, block, lds, s, Ap, lda, pa, Bp, ldb, pb, C, ldc, M, N, nk, tiles_m, tiles_n, ws, bias, relu, mask, ldmask);
    }
    GODE_LAUNCH_CHECK();
    if (nwg != tiles_m * tiles_n) {
        hipLaunchKernelGGL(pgemm_finish_kernel, dim3(tiles_m * tiles_n), dim3(512), 0, s, ws, nwg, nk, tiles_m, tiles_n, C, ldc,
                           M, N, bias, relu, mask, ldmask);
        GODE_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" int64_t gode_cut_pad(int64_t n) { return n <= 0 ? 0 : (n + 127) / 128 * 128; }

extern "C" int gode_cut_bf16x3_f32(const float* X, int64_t ld, int64_t R, int64_t C, void* planes, void* stream)
{
    if (R < 0 || C < 0 || ld < C) return GODE_E_SHAPE;
    if (R == 0 || C == 0) return 0;
    if (!X || !planes) return GODE_E_NULLPTR;
    if (R > INT32_MAX - 256 || C > INT32_MAX - 4096) return GODE_E_RANGE;
    if (((uintptr_t)planes) & 15) return GODE_E_ALIGN;
    const int R_pad = (int)gode_cut_pad(R), C_pad = (int)gode_cut_pad(C);
    const int per_block = 512 * CUT_PAIRS;
    if ((C_pad + per_block - 1) / per_block > 65535) return GODE_E_RANGE;
    hipLaunchKernelGGL(cut_kernel, dim3(R_pad, (C_pad + per_block - 1) / per_block), dim3(256), 0, (hipStream_t)stream,
                       X, ld, (int)R, (int)C, reinterpret_cast<unsigned*>(planes), R_pad, C_pad);
    GODE_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gode_pgemm_workspace_bytes(int64_t M, int64_t N, int64_t K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int64_t tiles = (gode_cut_pad(M) / TM) * (gode_cut_pad(N) / TN);
    const int nwg = choose_blocks(tiles, (K + BK - 1) / BK, true);
    return nwg != tiles ? 2 * (int64_t)nwg * SLOT_F * 4 : 0;      // two partial-tile slots per block
}

extern "C" int gode_pgemm_bf16x3(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void* A_planes,
                                 const void* B_planes, float* C, int64_t ldc, const float* bias, int relu,
                                 const float* mask, int64_t ldmask, int products, void* workspace, void* stream)
{
    if (M < 0 || N < 0 || K < 0) return GODE_E_SHAPE;
    if (M == 0 || N == 0) return 0;
    if (K == 0) return GODE_E_SHAPE;
    if (!A_planes || !B_planes || !C) return GODE_E_NULLPTR;
    if (M > INT32_MAX - 256 || N > INT32_MAX - 256 || K > INT32_MAX - 256) return GODE_E_RANGE;
    if (ldc < N || (mask && ldmask < N)) return GODE_E_SHAPE;
    if (products != 8 && products != 6) return GODE_E_UNSUPPORTED;
    if ((((uintptr_t)A_planes) | ((uintptr_t)B_planes)) & 15) return GODE_E_ALIGN;
    const int64_t Mp = gode_cut_pad(M), Np = gode_cut_pad(N), Kp = gode_cut_pad(K);
    if ((Mp / TM) * (Np / TN) > INT32_MAX / 2) return GODE_E_RANGE;
    // planes of A as stored: M x K (trans_a = 0) or K x M; of B: K x N (trans_b = 0) or N x K
    const int64_t lda = trans_a ? Mp : Kp, ldb = trans_b ? Kp : Np;
    const int64_t pa = Mp * Kp, pb = Np * Kp;
    const short* Ap = reinterpret_cast<const short*>(A_planes);
    const short* Bp = reinterpret_cast<const short*>(B_planes);
    hipStream_t s = (hipStream_t)stream;
    float* ws = reinterpret_cast<float*>(workspace);          // nullable: without it the contraction is never split
    if (ws && (((uintptr_t)ws) & 15)) return GODE_E_ALIGN;
    const bool akc = !trans_a, bkc = trans_b != 0;
    if (akc && bkc) return launch<true, true>(products, Ap, lda, pa, Bp, ldb, pb, C, ldc, (int)M, (int)N, (int)((K + BK - 1) / BK), bias, relu, mask, ldmask, ws, s);
    if (akc && !bkc) return launch<true, false>(products, Ap, lda, pa, Bp, ldb, pb, C, ldc, (int)M, (int)N, (int)((K + BK - 1) / BK), bias, relu, mask, ldmask, ws, s);
    if (!akc && bkc) return launch<false, true>(products, Ap, lda, pa, Bp, ldb, pb, C, ldc, (int)M, (int)N, (int)((K + BK - 1) / BK), bias, relu, mask, ldmask, ws, s);
    return launch<false, false>(products, Ap, lda, pa, Bp, ldb, pb, C, ldc, (int)M, (int)N, (int)((K + BK - 1) / BK), bias, relu, mask, ldmask, ws, s);
}
