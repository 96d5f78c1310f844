// gemm_pc.hip — the dense products of the ODE function at d = 128 in PRODUCER / CONSUMER form on the bf16 matrix
// cores, from exact three-way cuts of the fp32 operands (gfx950).
//
//   forward  S  = [t | GN(x)] W            (GCN/models.py:175-177 + GCN/layers.py:70 of the reference)
//   VJP      dx = GN'(x)^T (dS W1^T)        (their autograd)
//
// Why this form (measured): v_mfma_f32_16x16x4_f32 keeps every other wave of its SIMD from issuing
// anything, so an exact-fp32 MFMA kernel pays matrix time + memory time + GroupNorm time in series (0.49 ms for the
// VJP, 0.47 ms for a 3-4 term forward product at 2^20 x 128, against 0.27-0.45 ms of traffic).  v_mfma_f32_16x16x32_bf16
// does not do that; the weight gradient already runs this way (gemm.hip: wgrad_split_kernel, 0.44 -> 0.30 ms).
//
// Arithmetic.  Every fp32 operand is cut by truncation into three bf16 numbers, x = hi + mid + lo EXACTLY (8 + 8 + 8
// significant bits, nothing rounded); a product is accumulated in fp32 from eight of the nine piece products in
// increasing order of magnitude (lo*mid, mid*lo, lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi); the one left out, lo*lo,
// is below 2^-32 of the product.  The result is an fp32 result: held to the same bars as the fp32-MFMA kernels
// (tests/test_gpu_kernels.py: 2e-6 relative against float64 for the products, 1e-5 for the ODE function).
//
// Schedule.  One 1 024-thread block per CU.  Waves 0-7 CONSUME: wave w owns the 16 output columns 16w .. 16w+15 of
// every row, keeps the three pieces of its 16 x 128 slab of the weight matrix in REGISTERS for the whole launch (48
// registers: the A operand of all its MFMAs - no weight traffic in the loop at all), reads the data pieces of a 32-row
// tile from LDS (24 ds_read_b128 per tile), issues 64 MFMAs per tile and finishes its rows in registers (forward: store;
// VJP: GroupNorm backward, dgamma / dbeta partial sums, RK combine, store).  Waves 8-11 and 12-15 PRODUCE alternate
// tiles: global loads (requested two tile periods ahead, each group holding one tile in flight), stage combination,
// GroupNorm forward, the three-way cut, 8-byte LDS stores into a [piece][row][k] image with 272-byte rows.  Two LDS
// buffers (51 KB), one LDS-only block barrier per tile (outstanding global loads stay in flight across it).
// Per 32-row tile and SIMD: 128 MFMAs = 2 048 matrix cycles beside ~400-700 vector instructions; the launch moves
// 2 (forward, one term) to 5 (forward, four terms) N x d arrays.
#include "common.h"
#include "dense_common.h"
#include "dense_pc.h"
#include "prof.h"

namespace {

constexpr int D = 128;
constexpr int LDK = D + 8;                  // bf16 elements per image row (272 B: conflict-free 16-byte operand reads)
constexpr int LDX = D + 4;                  // floats per row of the fp32 x tile of the VJP kernel
constexpr int kBlocks = 256;                // one block per CU
// R = rows per tile (32 or 64): one piece image is R x LDK bf16, a buffer three of them
template <int R> struct Img { static constexpr int PIECE_B = R * LDK * 2, BUF_B = 3 * PIECE_B; };

// the bf16 halves (upper 16 bits) of eight fp32 words -> one MFMA operand
__device__ __forceinline__ bf16x8 pack8(const unsigned (&u)[8]) {
    const uint2 a = pack_hi16x4(u[0], u[1], u[2], u[3]), b = pack_hi16x4(u[4], u[5], u[6], u[7]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {a.x, a.y, b.x, b.y};
    return __builtin_bit_cast(bf16x8, v);
}

// The consumer's stationary operand: pieces of the 16 x 128 weight slab  A[m][k], m = lane & 15, k = 32 kb + 8 (lane >> 4) + e.
// element(m, k) is supplied by the caller (forward: W1[k][n0 + m], VJP: W1[i0 + m][k]).
template <typename F>
__device__ __forceinline__ void load_weight_pieces(bf16x8 (&A)[4][3], F element) {
    const int l = threadIdx.x & 63, m = l & 15, g = l >> 4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        unsigned h[8], md[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3_trunc(element(m, 32 * kb + 8 * g + e), h[e], md[e], lo[e]);
        A[kb][0] = pack8(h); A[kb][1] = pack8(md); A[kb][2] = pack8(lo);
    }
}

// acc[rt] += A (16 x 128 slab, registers) * B (rows 16 rt .. 16 rt + 15 of the tile image in LDS)^T: R/16 x 4 k-blocks,
// eight piece products each, smallest first.  The operands of step s+1 are read while step s multiplies; the
// scheduling barrier per step keeps hipcc from hoisting all the reads to the top.
template <int R>
__device__ __forceinline__ void tile_product(const char* img /* + lane offset */, const bf16x8 (&A)[4][3], f32x4 (&acc)[R / 16]) {
    constexpr int PIECE_B = Img<R>::PIECE_B, NS = (R / 16) * 4;
    bf16x8 b[2][3];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) b[0][pc] = *reinterpret_cast<const bf16x8*>(img + pc * PIECE_B);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int rt = s >> 2, kb = s & 3;
        if (s + 1 < NS) {
            const int rt2 = (s + 1) >> 2, kb2 = (s + 1) & 3;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                b[(s + 1) & 1][pc] = *reinterpret_cast<const bf16x8*>(img + pc * PIECE_B + rt2 * 16 * LDK * 2 + kb2 * 64);
        }
        const bf16x8 bh = b[s & 1][0], bm = b[s & 1][1], bl = b[s & 1][2];
        f32x4 c = acc[rt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][2], bm, c, 0, 0, 0);      // lo*mid, mid*lo: 2^-24 of the product each
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][1], bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][2], bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][0], bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][1], bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][1], bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][0], bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][0], bh, c, 0, 0, 0);
        acc[rt] = c;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one row of a staging thread: its 4 consecutive columns -> 8 bytes in each of the three piece images
template <int PIECE_B>
__device__ __forceinline__ void stage_row4(char* dst /* piece 0 */, const float4 v) {
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3_trunc(v.x, h0, m0, l0); split3_trunc(v.y, h1, m1, l1); split3_trunc(v.z, h2, m2, l2); split3_trunc(v.w, h3, m3, l3);
    *reinterpret_cast<uint2*>(dst) = pack_hi16x4(h0, h1, h2, h3);
    *reinterpret_cast<uint2*>(dst + PIECE_B) = pack_hi16x4(m0, m1, m2, m3);
    *reinterpret_cast<uint2*>(dst + 2 * PIECE_B) = pack_hi16x4(l0, l1, l2, l3);
}

// Tiles of block b: b, b + stride, ...  (k-th tile of the block = b + k * stride)
struct TileWalk {
    int n_tiles, stride, mine;
    __device__ TileWalk(int n_rows, int R) {
        n_tiles = (n_rows + R - 1) / R;
        stride = gridDim.x;
        mine = (int)blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / stride + 1 : 0;
    }
    __device__ int tile(int k) const { return blockIdx.x + k * stride; }
    __device__ int clamped(int k) const { return blockIdx.x + (k < mine ? k : (mine > 0 ? mine - 1 : 0)) * stride; }
};

// ---------------------------------------------------------------------------------------------------------------
// forward:  S[row, :] = t * W[0, :] + GN(sum_j c_j x_j[row, :]) * W[1:, :]      (+ x_out = the combined input)
// NX = number of terms held raw in the producers' prefetch registers (1..4); 0 = any count, combined at load.
// R  = rows per tile: 32.  (64-row tiles - twice the bytes in flight per producer group - were measured and are
//      slower: two terms 0.326 against 0.282 ms, one term equal; tools/dev/pc_ab.py, same process, interleaved.)
// ---------------------------------------------------------------------------------------------------------------
template <int CG, int NX, bool XOUT, int R>
__global__ __launch_bounds__(1024, 1) void gn_gemm_fwd_pc_kernel(LinComb xin, int n_rows, float eps,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ W, int has_time, float t,
                                                                float* __restrict__ S, float* __restrict__ xout)
{
    constexpr int PIECE_B = Img<R>::PIECE_B, BUF_B = Img<R>::BUF_B;
    constexpr int RPT = R / 8;                    // rows per staging thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const TileWalk tw(n_rows, R);
    if (threadIdx.x >= 512) {
        // ---- producers: thread (trow, tc4) stages rows RPT trow + p, columns 4 tc4 .. + 3 (one GroupNorm group)
        const int grp = (threadIdx.x - 512) >> 8, pt = (threadIdx.x - 512) & 255;
        const int trow = pt >> 5, tcol = 4 * (pt & 31);
        const float4 gmv = gamma ? ld4(gamma + tcol) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 btv = beta ? ld4(beta + tcol) : make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int NXR = NX > 0 ? NX : 1;
        float4 xr[NXR][RPT];
        auto prefetch = [&](int k) {              // unconditional loads from clamped rows: no wait at a branch join
            const int tile = tw.clamped(k);
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
                const int row = tile * R + RPT * trow + p;
                const int64_t off = (int64_t)(row < n_rows ? row : n_rows - 1) * D + tcol;
                if (NX > 0) {
#pragma unroll
                    for (int j = 0; j < NXR; ++j) xr[j][p] = ld4(xin.ptr[j] + off);
                } else {
                    xr[0][p] = lc_load4(xin, off);
                }
            }
        };
        auto stage = [&](int k) {                 // tile k of this block -> buffer k & 1
            char* img = lds + (k & 1) * BUF_B + (RPT * trow) * LDK * 2 + tcol * 2;
            const int tile = tw.tile(k);
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
                const int row = tile * R + RPT * trow + p;
                float4 x = xr[0][p];
                if (NX > 0) {                     // the term order and arithmetic of load_tile_n / lc_load4_n
                    x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < NXR; ++j) {
                        const float c = xin.coef[j];
                        x.x = fmaf(c, xr[j][p].x, x.x); x.y = fmaf(c, xr[j][p].y, x.y);
                        x.z = fmaf(c, xr[j][p].z, x.z); x.w = fmaf(c, xr[j][p].w, x.w);
                    }
                }
                if (XOUT && row < n_rows) *reinterpret_cast<float4*>(xout + (int64_t)row * D + tcol) = x;
                float4 xn = gn_forward_v<CG>(x, eps, gmv, btv);
                if (row >= n_rows) xn = make_float4(0.f, 0.f, 0.f, 0.f);
                stage_row4<PIECE_B>(img + p * LDK * 2, xn);
            }
        };
        prefetch(grp);
        if (grp == 0 && tw.mine > 0) { stage(0); prefetch(2); }
        lds_barrier();                                                     // tile 0 staged
        for (int k = 0; k < tw.mine; ++k) {                                // while the consumers multiply tile k
            if (((k + 1) & 1) == grp && k + 1 < tw.mine) { stage(k + 1); prefetch(k + 3); }
            lds_barrier();
        }
        return;
    }
    // ---- consumers: wave w owns output columns n0 = 16 w ..; lane (r, g): row r of a 16-row part, columns n0 + 4 g ..
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const int n0 = 16 * wave;
    bf16x8 A[4][3];
    load_weight_pieces(A, [&](int m, int k) { return W[(int64_t)(k + has_time) * D + n0 + m]; });     // W1^T slab
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f};
    if (has_time) { const float4 w0 = ld4(W + n0 + 4 * g); t0 = (f32x4){t * w0.x, t * w0.y, t * w0.z, t * w0.w}; }
    const int rd_off = r * LDK * 2 + g * 16;
    lds_barrier();                                                         // tile 0 staged
    for (int k = 0; k < tw.mine; ++k) {
        f32x4 acc[R / 16];
#pragma unroll
        for (int rt = 0; rt < R / 16; ++rt) acc[rt] = t0;
        tile_product<R>(lds + (k & 1) * BUF_B + rd_off, A, acc);
        const int row0 = tw.tile(k) * R + r;
#pragma unroll
        for (int rt = 0; rt < R / 16; ++rt) {
            const int row = row0 + 16 * rt;
            if (row < n_rows)
                *reinterpret_cast<float4*>(S + (int64_t)row * D + n0 + 4 * g) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// VJP w.r.t. x:  dxn = dS W1^T ; dx = GN'(x)^T dxn ; out = out_scale * dx (+ pre)
// one partial sum of dgamma / dbeta per block.  NX = number of terms of x (1, 2; 0 = any count, combined at load).
// XLDS: the producers stage BOTH arrays - dS as piece images, the combined x as an fp32 tile (row stride D + 4 floats)
// that the consumers read in their GroupNorm backward, so x, too, is requested two tile periods ahead; !XLDS: a consumer
// lane loads its own x values at the head of the tile and uses them after the matrix phase.  Which one is taken: the
// launcher below (measured per term count).
// ---------------------------------------------------------------------------------------------------------------
template <int CG, int NX, bool XLDS>   // CG: 0 or 4
__global__ __launch_bounds__(1024, 1) void gn_gemm_bwd_pc_kernel(LinComb xin, int n_rows, float eps,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ W, int has_time,
                                                                const float* __restrict__ dS, float out_scale,
                                                                LinComb pre, float* __restrict__ dx,
                                                                float* __restrict__ dgamma_part,
                                                                float* __restrict__ dbeta_part, int n_part)
{
    constexpr int R = 32, PIECE_B = Img<R>::PIECE_B;
    constexpr int XT_B = (CG != 0 && XLDS) ? R * LDX * 4 : 0;                        // the fp32 x tile behind the piece images
    constexpr int BUF_B = Img<R>::BUF_B + XT_B;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const TileWalk tw(n_rows, R);
    if (threadIdx.x >= 512) {
        // ---- producers: the cotangent tile dS -> piece images, the combined x -> fp32 tile
        const int grp = (threadIdx.x - 512) >> 8, pt = (threadIdx.x - 512) & 255;
        const int trow = pt >> 5, tcol = 4 * (pt & 31);
        constexpr int NXR = NX > 0 ? NX : 1;
        float4 gr[4], xr[NXR][4];
        auto prefetch = [&](int k) {
            const int tile = tw.clamped(k);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = tile * R + 4 * trow + p;
                const int64_t off = (int64_t)(row < n_rows ? row : n_rows - 1) * D + tcol;
                gr[p] = ld4(dS + off);
                if (CG != 0 && XLDS) {
                    if (NX > 0) {
#pragma unroll
                        for (int j = 0; j < NXR; ++j) xr[j][p] = ld4(xin.ptr[j] + off);
                    } else {
                        xr[0][p] = lc_load4(xin, off);
                    }
                }
            }
        };
        auto stage = [&](int k) {
            char* buf = lds + (k & 1) * BUF_B;
            char* img = buf + (4 * trow) * LDK * 2 + tcol * 2;
            const int tile = tw.tile(k);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = gr[p];
                if (tile * R + 4 * trow + p >= n_rows) v = make_float4(0.f, 0.f, 0.f, 0.f);
                stage_row4<PIECE_B>(img + p * LDK * 2, v);
                if (CG != 0 && XLDS) {
                    float4 x = xr[0][p];
                    if (NX > 0) {
                        x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int j = 0; j < NXR; ++j) {
                            const float c = xin.coef[j];
                            x.x = fmaf(c, xr[j][p].x, x.x); x.y = fmaf(c, xr[j][p].y, x.y);
                            x.z = fmaf(c, xr[j][p].z, x.z); x.w = fmaf(c, xr[j][p].w, x.w);
                        }
                    }
                    *reinterpret_cast<float4*>(buf + Img<R>::BUF_B + ((4 * trow + p) * LDX + tcol) * 4) = x;
                }
            }
        };
        prefetch(grp);
        if (grp == 0 && tw.mine > 0) { stage(0); prefetch(2); }
        lds_barrier();
        for (int k = 0; k < tw.mine; ++k) {
            if (((k + 1) & 1) == grp && k + 1 < tw.mine) { stage(k + 1); prefetch(k + 3); }
            lds_barrier();
        }
        // rows of the partial buffers that no block owns (the caller's buffers hold gode_gemm_bwd_parts() rows)
        if (CG != 0 && dgamma_part)
            for (int p = gridDim.x + blockIdx.x; p < n_part; p += gridDim.x)
                for (int c = threadIdx.x - 512; c < D; c += 512) { dgamma_part[(int64_t)p * D + c] = 0.f; dbeta_part[(int64_t)p * D + c] = 0.f; }
        return;
    }
    // ---- consumers: wave w owns input channels i0 = 16 w .. of dx; lane (r, g): row r, channels i0 + 4 g .. + 3 = one group
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const int i0 = 16 * wave, c0 = i0 + 4 * g;
    bf16x8 A[4][3];
    load_weight_pieces(A, [&](int m, int k) { return W[(int64_t)(i0 + m + has_time) * D + k]; });     // W1 slab: rows i, k = n
    const float4 gm = (CG != 0 && gamma) ? ld4(gamma + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 dgs = make_float4(0.f, 0.f, 0.f, 0.f), dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int rd_off = r * LDK * 2 + g * 16;
    const int xt_off = Img<R>::BUF_B + (r * LDX + c0) * 4;
    lds_barrier();                                                         // tile 0 staged
    for (int k = 0; k < tw.mine; ++k) {
        const char* buf = lds + (k & 1) * BUF_B;
        const int row0 = tw.tile(k) * R + r;
        constexpr int NXC = NX > 0 ? NX : 1;
        float4 xd[2][NXC];                      // !XLDS: this lane's x values, requested before the matrix phase
        if (CG != 0 && !XLDS) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = row0 + 16 * rt;
                const int64_t off = (int64_t)(row < n_rows ? row : n_rows - 1) * D + c0;
                if (NX > 0) {
#pragma unroll
                    for (int j = 0; j < NXC; ++j) xd[rt][j] = ld4(xin.ptr[j] + off);
                } else {
                    xd[rt][0] = lc_load4(xin, off);
                }
            }
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        tile_product<R>(buf + rd_off, A, acc);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = row0 + 16 * rt;
            const bool valid = row < n_rows;
            const float4 dy = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
            float4 out = dy;
            if (CG != 0) {                      // four channels per group = this lane's float4 (gn_gemm_bwd_kernel, CG == 4)
                float4 x;
                if (XLDS) {
                    x = *reinterpret_cast<const float4*>(buf + xt_off + rt * 16 * LDX * 4);
                } else {
                    x = xd[rt][0];
                    if (NX > 0) {
                        x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int j = 0; j < NXC; ++j) {
                            const float c = xin.coef[j];
                            x.x = fmaf(c, xd[rt][j].x, x.x); x.y = fmaf(c, xd[rt][j].y, x.y);
                            x.z = fmaf(c, xd[rt][j].z, x.z); x.w = fmaf(c, xd[rt][j].w, x.w);
                        }
                    }
                }
                float4 mean, rstd;
                gn_stats<CG>(x, eps, mean, rstd);
                const float4 xh = make_float4((x.x - mean.x) * rstd.x, (x.y - mean.y) * rstd.y,
                                              (x.z - mean.z) * rstd.z, (x.w - mean.w) * rstd.w);
                const float4 dh = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
                if (valid) {
                    dgs.x += dy.x * xh.x; dgs.y += dy.y * xh.y; dgs.z += dy.z * xh.z; dgs.w += dy.w * xh.w;
                    dbs.x += dy.x; dbs.y += dy.y; dbs.z += dy.z; dbs.w += dy.w;
                }
                const float m1 = ((dh.x + dh.y) + (dh.z + dh.w)) * 0.25f;
                const float m2 = ((dh.x * xh.x + dh.y * xh.y) + (dh.z * xh.z + dh.w * xh.w)) * 0.25f;
                const float rs = rstd.x;
                out = make_float4(rs * (dh.x - m1 - xh.x * m2), rs * (dh.y - m1 - xh.y * m2),
                                  rs * (dh.z - m1 - xh.z * m2), rs * (dh.w - m1 - xh.w * m2));
            }
            if (valid) {
                float4 o = make_float4(out_scale * out.x, out_scale * out.y, out_scale * out.z, out_scale * out.w);
                if (pre.n > 0) {                // fused RK solution combine of the adjoint component (last stage only)
                    const float4 pv = lc_load4(pre, (int64_t)row * D + c0);
                    o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
                }
                *reinterpret_cast<float4*>(dx + (int64_t)row * D + c0) = o;
            }
        }
        lds_barrier();
    }
    if (CG != 0 && dgamma_part) {
        // the 16 rows of a lane group hold the same channels: sum over r (lane bits 0-3); every wave owns its own
        // 16 channels, so the block partial needs no exchange between waves
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            dgs.x += __shfl_xor(dgs.x, o, 64); dgs.y += __shfl_xor(dgs.y, o, 64); dgs.z += __shfl_xor(dgs.z, o, 64); dgs.w += __shfl_xor(dgs.w, o, 64);
            dbs.x += __shfl_xor(dbs.x, o, 64); dbs.y += __shfl_xor(dbs.y, o, 64); dbs.z += __shfl_xor(dbs.z, o, 64); dbs.w += __shfl_xor(dbs.w, o, 64);
        }
        if (r == 0) {
            *reinterpret_cast<float4*>(dgamma_part + (int64_t)blockIdx.x * D + c0) = dgs;
            *reinterpret_cast<float4*>(dbeta_part + (int64_t)blockIdx.x * D + c0) = dbs;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// VJP and weight gradient in ONE pass (round 4):  dx = GN'(x)^T (dS W1^T)  AND  dW = [1 | GN(x)]^T dS  from one read of x
// and dS per adjoint stage (the two launches read both arrays twice: -1 GB per stage at 2^20 x 128).
//
// 768 threads: waves 0-7 consume, waves 8-11 produce (ONE producer group with two register sets, so that two tiles are in
// flight while a third is staged; 12 waves = 3 per SIMD leave 168 registers per wave - the 16-wave layout of the two
// separate kernels leaves 128, and a consumer here holds 48 (its W1 slab) + 32 (its dW tiles) + 8 (its dx tiles)
// resident beside ~36 of operand fragments).
// Producers stage, per 32-row tile: the pieces of dS and of xn = GN(x) as ROW-MAJOR images (272-byte rows) and the fp32 x
// tile the GroupNorm backward reads.  Consumers: (1) dxn = dS W1^T as in gn_gemm_bwd_pc_kernel - weight slab in
// registers, dS rows by ds_read_b128; (2) the wave's 4 x 2 tiles of xn^T dS - both operands are wanted TRANSPOSED (8
// consecutive rows of one column): ds_read_b64_tr_b16 delivers exactly that from the same row-major images, so nothing
// is staged twice.  128 MFMAs per tile and wave.  Outputs as the two kernels': dx (+ pre-terms), one dgamma / dbeta
// partial row per block, one dW partial ((D + has_time) x D, row 0 = column sums of dS) per block.
// ---------------------------------------------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));
#define GODE_LDS_AS __attribute__((address_space(3)))

// this lane's fragment of a transposed operand: rows 8 g .. 8 g + 7 of the image, column c16 + (lane & 15)
__device__ __forceinline__ bf16x8 tr_frag(const char* img_lane /* image + lane's offset */, int col_bytes) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GODE_LDS_AS s16x4_t*)(img_lane + col_bytes));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((GODE_LDS_AS s16x4_t*)(img_lane + col_bytes + 4 * LDK * 2));
    const bf16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return v;
}

template <int CG, int NX>   // CG: 0 or 4 channels per group; NX: terms of x held raw (1, 2), 0 = any count, combined at load
__global__ __launch_bounds__(768) void gn_gemm_bwd_wgrad_pc_kernel(LinComb xin, int n_rows, float eps,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 const float* __restrict__ W, int has_time,
                                                                 const float* __restrict__ dS, float out_scale,
                                                                 LinComb pre, float* __restrict__ dx,
                                                                 float* __restrict__ dgamma_part,
                                                                 float* __restrict__ dbeta_part, int n_part,
                                                                 float* __restrict__ dW_part)
{
    constexpr int R = 32, PIECE_B = Img<R>::PIECE_B, IMG_B = 3 * PIECE_B;
    constexpr int XT_B = CG != 0 ? R * LDX * 4 : 0;
    constexpr int BUF_B = 2 * IMG_B + XT_B;                         // dS pieces | xn pieces | fp32 x tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const TileWalk tw(n_rows, R);
    float* out = dW_part + (int64_t)blockIdx.x * (D + has_time) * D;
    if (threadIdx.x >= 512) {
        // ---- producers: thread (trow, tc4) stages rows 4 trow + p, columns 4 tc4 .. + 3 (one GroupNorm group)
        const int pt = threadIdx.x - 512;
        const int trow = pt >> 5, tcol = 4 * (pt & 31);
        const float4 gmv = gamma ? ld4(gamma + tcol) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 btv = beta ? ld4(beta + tcol) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int NXR = NX > 0 ? NX : 1;
        struct Regs { float4 g[4]; float4 x[NXR][4]; };
        Regs qa, qb;
        auto prefetch = [&](Regs& q, int k) {     // unconditional loads from clamped rows
            const int tile = tw.clamped(k);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = tile * R + 4 * trow + p;
                const int64_t off = (int64_t)(row < n_rows ? row : n_rows - 1) * D + tcol;
                q.g[p] = ld4(dS + off);
                if (NX > 0) {
#pragma unroll
                    for (int j = 0; j < NXR; ++j) q.x[j][p] = ld4(xin.ptr[j] + off);
                } else {
                    q.x[0][p] = lc_load4(xin, off);
                }
            }
        };
        // NO branch around loads or around the code that consumes them: hipcc waits vmcnt(0) at the join of a branch that
        // holds a load, which would drain the OTHER register set's prefetch (issued a moment ago) in front of every stage -
        // the first version did, and ran at 4.7 us per tile instead of ~2.  Tiles past the block's last one are staged
        // from clamped rows into the buffer nobody reads any more, and count as invalid (live = false).
        auto stage = [&](Regs& q, int k) {        // tile k of this block -> buffer k & 1
            char* buf = lds + (k & 1) * BUF_B;
            const bool live = k < tw.mine;
            const int tile = tw.clamped(k);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int rr = 4 * trow + p;
                const bool valid = live && tile * R + rr < n_rows;
                float4 x = q.x[0][p];
                if (NX > 0) {                     // the term order and arithmetic of lc_load4_n
                    x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < NXR; ++j) {
                        const float c = xin.coef[j];
                        x.x = fmaf(c, q.x[j][p].x, x.x); x.y = fmaf(c, q.x[j][p].y, x.y);
                        x.z = fmaf(c, q.x[j][p].z, x.z); x.w = fmaf(c, q.x[j][p].w, x.w);
                    }
                }
                float4 xn = gn_forward_v<CG>(x, eps, gmv, btv);
                float4 gv = q.g[p];
                if (!valid) { xn = make_float4(0.f, 0.f, 0.f, 0.f); gv = make_float4(0.f, 0.f, 0.f, 0.f); }
                csum.x += gv.x; csum.y += gv.y; csum.z += gv.z; csum.w += gv.w;
                stage_row4<PIECE_B>(buf + rr * LDK * 2 + tcol * 2, gv);
                stage_row4<PIECE_B>(buf + IMG_B + rr * LDK * 2 + tcol * 2, xn);
                if (CG != 0) *reinterpret_cast<float4*>(buf + 2 * IMG_B + (rr * LDX + tcol) * 4) = x;
            }
        };
        prefetch(qa, 0);
        prefetch(qb, 1);
        stage(qa, 0);
        prefetch(qa, 2);
        lds_barrier();                                                     // tile 0 staged
        for (int k = 0; k < tw.mine;) {                                    // while the consumers work on tile k
            stage(qb, k + 1);
            prefetch(qb, k + 3);
            lds_barrier();
            if (++k >= tw.mine) break;
            stage(qa, k + 1);
            prefetch(qa, k + 3);
            lds_barrier();
            ++k;
        }
        // every consumer is past its last operand read: column sums of dS (the time row of dW), and the partial rows of
        // the GroupNorm affine gradients that no block owns
        if (has_time) {
            float* red = smem;                                             // [8][D]
            *reinterpret_cast<float4*>(red + trow * D + tcol) = csum;
        }
        __syncthreads();                                                   // pairs with the consumers' barrier below
        if (has_time && pt < D) {
            float sacc = 0.f;
            for (int p = 0; p < 8; ++p) sacc += smem[p * D + pt];
            out[pt] = sacc;
        }
        if (CG != 0 && dgamma_part)
            for (int p = gridDim.x + blockIdx.x; p < n_part; p += gridDim.x)
                for (int c = pt; c < D; c += 256) { dgamma_part[(int64_t)p * D + c] = 0.f; dbeta_part[(int64_t)p * D + c] = 0.f; }
        return;
    }
    // ---- consumers
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    const int i0 = 16 * wave, c0 = i0 + 4 * g;                            // VJP: this lane's four input channels
    bf16x8 A[4][3];
    load_weight_pieces(A, [&](int m, int k) { return W[(int64_t)(i0 + m + has_time) * D + k]; });     // W1 slab: rows i, k = n
    const float4 gm = (CG != 0 && gamma) ? ld4(gamma + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 dgs = make_float4(0.f, 0.f, 0.f, 0.f), dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int rd_off = r * LDK * 2 + g * 16;
    const int xt_off = 2 * IMG_B + (r * LDX + c0) * 4;
    const int a0 = 4 * (wave >> 2), b0 = 2 * (wave & 3);                  // weight gradient: tiles (a0 .. a0 + 3) x (b0, b0 + 1)
    const int tr_off = (8 * g + (r >> 2)) * LDK * 2 + 4 * (r & 3) * 2;    // lane 4 q + p of a group: row q, columns 4 p ..
    f32x4 wacc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) wacc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    lds_barrier();                                                         // tile 0 staged
    for (int k = 0; k < tw.mine; ++k) {
        const char* buf = lds + (k & 1) * BUF_B;
        const int row0 = tw.tile(k) * R + r;
        // (1) dxn = dS W1^T and the GroupNorm backward
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        tile_product<R>(buf + rd_off, A, acc);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = row0 + 16 * rt;
            const bool valid = row < n_rows;
            const float4 dy = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
            float4 o4 = dy;
            if (CG != 0) {
                const float4 x = *reinterpret_cast<const float4*>(buf + xt_off + rt * 16 * LDX * 4);
                float4 mean, rstd;
                gn_stats<CG>(x, eps, mean, rstd);
                const float4 xh = make_float4((x.x - mean.x) * rstd.x, (x.y - mean.y) * rstd.y,
                                              (x.z - mean.z) * rstd.z, (x.w - mean.w) * rstd.w);
                const float4 dh = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
                if (valid) {
                    dgs.x += dy.x * xh.x; dgs.y += dy.y * xh.y; dgs.z += dy.z * xh.z; dgs.w += dy.w * xh.w;
                    dbs.x += dy.x; dbs.y += dy.y; dbs.z += dy.z; dbs.w += dy.w;
                }
                const float m1 = ((dh.x + dh.y) + (dh.z + dh.w)) * 0.25f;
                const float m2 = ((dh.x * xh.x + dh.y * xh.y) + (dh.z * xh.z + dh.w * xh.w)) * 0.25f;
                const float rs = rstd.x;
                o4 = make_float4(rs * (dh.x - m1 - xh.x * m2), rs * (dh.y - m1 - xh.y * m2),
                                 rs * (dh.z - m1 - xh.z * m2), rs * (dh.w - m1 - xh.w * m2));
            }
            if (valid) {
                float4 o = make_float4(out_scale * o4.x, out_scale * o4.y, out_scale * o4.z, out_scale * o4.w);
                if (pre.n > 0) {                // fused RK solution combine of the adjoint component (last stage only)
                    const float4 pv = lc_load4(pre, (int64_t)row * D + c0);
                    o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
                }
                *reinterpret_cast<float4*>(dx + (int64_t)row * D + c0) = o;
            }
        }
        // (2) the wave's tiles of xn^T dS: transposed fragments straight from the row-major images
        const char* Gp = buf + tr_off;
        const char* Xp = buf + IMG_B + tr_off;
        bf16x8 Bf[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) Bf[b][pc] = tr_frag(Gp + pc * PIECE_B, 16 * (b0 + b) * 2);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16x8 Af[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) Af[pc] = tr_frag(Xp + pc * PIECE_B, 16 * (a0 + a) * 2);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 c = wacc[a][b];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[2], Bf[b][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[1], Bf[b][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[2], Bf[b][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[0], Bf[b][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[1], Bf[b][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[1], Bf[b][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[0], Bf[b][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[0], Bf[b][0], c, 0, 0, 0);
                wacc[a][b] = c;
            }
        }
        lds_barrier();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * (a0 + a) + 4 * g + q;
                out[(int64_t)(i + has_time) * D + 16 * (b0 + b) + r] = wacc[a][b][q];
            }
    if (CG != 0 && dgamma_part) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            dgs.x += __shfl_xor(dgs.x, o, 64); dgs.y += __shfl_xor(dgs.y, o, 64); dgs.z += __shfl_xor(dgs.z, o, 64); dgs.w += __shfl_xor(dgs.w, o, 64);
            dbs.x += __shfl_xor(dbs.x, o, 64); dbs.y += __shfl_xor(dbs.y, o, 64); dbs.z += __shfl_xor(dbs.z, o, 64); dbs.w += __shfl_xor(dbs.w, o, 64);
        }
        if (r == 0) {
            *reinterpret_cast<float4*>(dgamma_part + (int64_t)blockIdx.x * D + c0) = dgs;
            *reinterpret_cast<float4*>(dbeta_part + (int64_t)blockIdx.x * D + c0) = dbs;
        }
    }
    __syncthreads();                                                       // pairs with the producers' barrier before the time row
}

int64_t pc_blocks(int64_t n_rows, int R) {
    int64_t b = (n_rows + R - 1) / R;
    if (b < 1) b = 1;
    if (b > kBlocks) b = kBlocks;
    return b;
}

template <typename K>
int set_lds_pc(K kernel, size_t bytes) { return gode_set_lds_once(reinterpret_cast<const void*>(kernel), bytes); }

}  // namespace

int gode_pc_fwd_launch(const LinComb& lc, int64_t n_rows, float eps, const float* gamma, const float* beta,
                       const float* W, int has_time, float t, float* S, float* xout, int cg, hipStream_t s)
{
    int rc = 0;
#define GODE_FPC4(CGV, NXV, XO, RV)                                                                               \
    { const size_t lds = 2 * (size_t)Img<RV>::BUF_B;                                                               \
      const int64_t blocks = pc_blocks(n_rows, RV);                                                                \
      rc = set_lds_pc(gn_gemm_fwd_pc_kernel<CGV, NXV, XO, RV>, lds); if (rc) return rc;                            \
      const int slot = gode_prof_begin(s, D, n_rows, (int64_t)lc.n - 1 + (XO ? 1 : 0), GODE_PROF_GEMM_FWD | GODE_PROF_FORM_PC); \
      hipLaunchKernelGGL((gn_gemm_fwd_pc_kernel<CGV, NXV, XO, RV>), dim3((unsigned)blocks), dim3(1024), lds, s,    \
                         lc, (int)n_rows, eps, gamma, beta, W, has_time, t, S, xout);                             \
      gode_prof_end(s, slot);                                                                                      \
      GODE_LAUNCH_CHECK(); return 0; }
#define GODE_FPC2(CGV, NXV, RV) { if (xout) GODE_FPC4(CGV, NXV, true, RV) else GODE_FPC4(CGV, NXV, false, RV) }
#define GODE_FPC(CGV) { switch (lc.n) { case 1: GODE_FPC2(CGV, 1, 32) case 2: GODE_FPC2(CGV, 2, 32) case 3: GODE_FPC2(CGV, 3, 32) \
                                        case 4: GODE_FPC2(CGV, 4, 32) default: GODE_FPC2(CGV, 0, 32) } }
    if (cg == 0) GODE_FPC(0) else if (cg == 1) GODE_FPC(1) else if (cg == 2) GODE_FPC(2) else if (cg == 4) GODE_FPC(4)
#undef GODE_FPC
#undef GODE_FPC2
#undef GODE_FPC4
    return GODE_E_UNSUPPORTED;
}

int gode_pc_bwd_launch(const LinComb& lc, int64_t n_rows, float eps, const float* gamma, const float* W, int has_time,
                       const float* dS, float out_scale, const LinComb& pre, float* dx, float* dgamma_part,
                       float* dbeta_part, int64_t n_part, int cg, hipStream_t s)
{
    const int64_t blocks = pc_blocks(n_rows, 32);
    int rc = 0;
#define GODE_BPC3(CGV, NXV, XL)                                                                                   \
    { const size_t lds = 2 * ((size_t)Img<32>::BUF_B + ((CGV != 0 && XL) ? 32 * LDX * 4 : 0));                     \
      rc = set_lds_pc(gn_gemm_bwd_pc_kernel<CGV, NXV, XL>, lds); if (rc) return rc;                                \
      const int slot = gode_prof_begin(s, D, n_rows, (int64_t)lc.n - 1 + pre.n, GODE_PROF_GEMM_BWD | GODE_PROF_FORM_PC); \
      hipLaunchKernelGGL((gn_gemm_bwd_pc_kernel<CGV, NXV, XL>), dim3((unsigned)blocks), dim3(1024), lds, s,        \
                         lc, (int)n_rows, eps, gamma, W, has_time, dS, out_scale, pre, dx, dgamma_part, dbeta_part, \
                         (int)n_part);                                                                             \
      gode_prof_end(s, slot);                                                                                      \
      GODE_LAUNCH_CHECK(); return 0; }
    // x of the GroupNorm backward: one term - the consumers load their own values (0.33 ms against 0.40 through LDS);
    // two and more - the producers combine the terms two tile periods ahead and hand the tile over in LDS (0.38 against
    // 0.41); same process, interleaved, tools/dev/pc_ab.py
#define GODE_BPC(CGV) { if (lc.n == 1) GODE_BPC3(CGV, 1, false) else if (lc.n == 2) GODE_BPC3(CGV, 2, true) else GODE_BPC3(CGV, 0, true) }
    if (cg == 0) GODE_BPC3(0, 1, false) else if (cg == 4) GODE_BPC(4)
#undef GODE_BPC
#undef GODE_BPC3
    return GODE_E_UNSUPPORTED;
}

int64_t gode_pc_bwd_wgrad_parts(int64_t n_rows) { return pc_blocks(n_rows, 32); }

int gode_pc_bwd_wgrad_launch(const LinComb& lc, int64_t n_rows, float eps, const float* gamma, const float* beta, const float* W,
                             int has_time, const float* dS, float out_scale, const LinComb& pre, float* dx,
                             float* dgamma_part, float* dbeta_part, int64_t n_part, float* dW_part, int cg, hipStream_t s)
{
    const int64_t blocks = pc_blocks(n_rows, 32);
    int rc = 0;
#define GODE_BW(CGV, NXV)                                                                                          \
    { const size_t lds = 2 * ((size_t)2 * 3 * Img<32>::PIECE_B + (CGV != 0 ? 32 * LDX * 4 : 0));                     \
      rc = set_lds_pc(gn_gemm_bwd_wgrad_pc_kernel<CGV, NXV>, lds); if (rc) return rc;                               \
      const int slot = gode_prof_begin(s, D, n_rows, (int64_t)lc.n - 1 + pre.n, GODE_PROF_BWD_WGRAD | GODE_PROF_FORM_PC); \
      hipLaunchKernelGGL((gn_gemm_bwd_wgrad_pc_kernel<CGV, NXV>), dim3((unsigned)blocks), dim3(768), lds, s,         \
                         lc, (int)n_rows, eps, gamma, beta, W, has_time, dS, out_scale, pre, dx, dgamma_part,       \
                         dbeta_part, (int)n_part, dW_part);                                                        \
      gode_prof_end(s, slot);                                                                                       \
      GODE_LAUNCH_CHECK(); return 0; }
#define GODE_BWN(CGV) { if (lc.n == 1) GODE_BW(CGV, 1) else if (lc.n == 2) GODE_BW(CGV, 2) else GODE_BW(CGV, 0) }
    if (cg == 0) GODE_BWN(0) else if (cg == 4) GODE_BWN(4)
#undef GODE_BWN
#undef GODE_BW
    return GODE_E_UNSUPPORTED;
}
