/*
 * graphode.h — C ABI of libgraphode.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary of the graph-odenet hot path: the ATen call
 * sites the reference reaches from its layers are replaced by
 * the entry points below.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - all matrices are row-major fp32, indices are int32;
 *   - the caller owns every buffer (scratch included); nothing here allocates,
 *     frees or synchronises; all launches are asynchronous on `stream`
 *     (a hipStream_t passed as void*);
 *   - return value: 0 = success, >0 = hipError_t of a failed launch,
 *     <0 = argument-validation code (GODE_E_*); no C++ exception crosses.
 *
 * Reference call sites replaced (paths relative to the reference checkout):
 *   gode_spmm_csr_f32            GCN/layers.py:33,71   torch.spmm(adj, support) (+bias :35,73; relu GCN/models.py:178)
 *                                GAT/layers.py:53,55   torch.spmm(Mtgt, .)      QC/mpnn.py:29, QC/layers.py:145
 *   gode_gn_time_gemm_f32        GCN/models.py:175-177 GroupNorm, [t|x] concat; GCN/layers.py:70 torch.mm
 *   gode_gn_time_gemm_bwd_f32    autograd of the three sites above
 *   gode_wgrad_f32               autograd of GCN/layers.py:70 w.r.t. weight
 *   gode_rect_gemm_f32 / _nt_ / gode_rect_wgrad_f32   GCN/layers.py:32 torch.mm(input, self.weight) of a layer with
 *                                in_features != out_features, and its autograd (dY W^T, X^T dY)
 *   gode_lincomb_f32             torchdiffeq RK stage input / solution combine (call site GCN/models.py:192)
 *   gode_rk_errnorm_f32          torchdiffeq dopri5 mixed-tolerance error ratio (same call site)
 *   gode_gemm_f32                QC/layers.py:46-86 EdgeEncoderMLP (torch.mm + bias + relu on the edge rows) and its autograd
 *   gode_cut_bf16x3_f32, gode_pgemm_bf16x3   the same call sites for large products (bf16 matrix cores, exact 3-way cuts)
 *   gode_adam_f32                optimizer.step() of QC/train_egcn.py, GCN/train_res.py:97 (torch.optim.Adam)
 *   gode_gat_*                   GAT/layers.py:40-55 (and :104-120)
 *   gode_edge_matvec_*           QC/mpnn.py:27-29, QC/layers.py:143-145
 */
#ifndef GRAPHODE_H
#define GRAPHODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GODE_ABI_VERSION 1

#define GODE_E_NULLPTR  (-1)
#define GODE_E_SHAPE    (-2)
#define GODE_E_ALIGN    (-3)
#define GODE_E_RANGE    (-4)
#define GODE_E_UNSUPPORTED (-5)

#define GODE_MAX_TERMS 8

/* value(i) = sum_j coef[j] * ptr[j][i]; every ptr[j] is a contiguous array of
 * the same length.  Used for RK stage inputs y + h*sum(a_ij k_j), the RK
 * solution combine and the cotangent fed to a VJP. */
typedef struct gode_lincomb {
    int32_t      n;                      /* 0..GODE_MAX_TERMS */
    float        coef[GODE_MAX_TERMS];
    const float* ptr[GODE_MAX_TERMS];
} gode_lincomb_t;

int         gode_abi_version(void);
/* run-time tuning switches (process-wide): "gemm_split" (forward dense product at d = 128 from an exact
 * three-way bf16 cut of both operands, eight piece products, what is dropped < 2^-32 of a product: 1 always; 2, the
 * default, for launches of <= 2 terms and >= 65 536 rows - the ones the fp32-MFMA kernel does not run at the memory
 * rate; 0 never), "overlap" (0/1, default 1: two-stream adjoint schedule),
 * "wgrad_split" (8 default / 6 / 0: weight gradient at d = 128 and >= 65 536 rows formed on the bf16 matrix cores
 * from an EXACT three-way cut of every fp32 operand - 8: all piece products down to 2^-32 of a product, i.e. more
 * accurate than an fp32 FMA chain; 6: down to 2^-23; 0: fp32-MFMA kernel), "wgrad_split_small" (0/1, default 0: use
 * the bf16-piece kernels below 65 536 rows too), "bwd_pc" (1 default / 0: the VJP at d = 128 and >= 65 536 rows in the
 * same exact bf16-piece arithmetic, producer / consumer form, csrc/gemm_pc.hip; 0: fp32-MFMA kernel), "fwd_pc" (3
 * default: the forward product likewise - bit 0: launches of <= 2 terms, bit 1: launches of >= 3 terms or with x_out;
 * 0: the round-2 selection by "gemm_split").  Initial values come from GODE_GEMM_SPLIT / GODE_OVERLAP /
 * GODE_WGRAD_SPLIT / GODE_BWD_PC / GODE_FWD_PC.
 * Returns 0 or GODE_E_UNSUPPORTED. */
int         gode_set_option(const char* name, int value);
int         gode_get_option(const char* name);
const char* gode_error_string(int code);   /* host string, static storage */

/* ---- sparse aggregation -------------------------------------------------
 * Z = A * X (+ bias);  Y = (sum_j pre.coef[j]*pre.ptr[j]) + alpha * (relu ? max(Z,0) : Z)
 *   (pre.n == 0: no pre-term; used to fold the Runge-Kutta solution combine
 *    y + h*sum(b_i k_i) into the launch that produces the last stage k_s);
 * optional second output  Y2 = (sum_j cot.coef[j]*cot.ptr[j]) * (Z > 0)
 *   (the relu-masked cotangent the adjoint pass needs).  pre / cot rows have ld = d.
 * epi == NULL: Y = Z.
 *
 * A is CSR (rowptr[n_rows+1], col[nnz], val[nnz] or NULL = all ones).
 * `items` (nullable) is the nnz-balanced work list built once per graph by
 * the host: n_items records of 4 int32 {row, begin, end, slot}; a row whose
 * record has slot < 0 is complete in one record and is written with the
 * epilogue; rows split over several records (slot >= 0) write raw partial sums
 * to partial[slot*d .. ] and are finished by the `long_rows` list: n_long
 * records of 4 int32 {row, first_slot, last_slot_exclusive, 0}.
 * With items == NULL every row is one record (n_items ignored).
 */
typedef struct gode_spmm_epilogue {
    const float*   bias;    /* nullable, [d] */
    int32_t        relu;
    float          alpha;   /* 1 for a plain product */
    gode_lincomb_t pre;     /* host-side struct; n == 0 = absent */
    gode_lincomb_t cot;     /* used when Y2 != NULL */
    float*         Y2;      /* nullable */
    float*         Y2_colsum; /* nullable, with Y2: per-block column sums of the rows of Y2 - gode_spmm_y2_colsum_rows(n_items,
                               n_long, d) rows of d floats (16-byte aligned); their sum over the rows (gode_colsum_f32 on this
                               array) is colsum(Y2), the bias gradient of the layer, without reading Y2 again */
} gode_spmm_epilogue_t;

/* rows of the Y2_colsum array for a graph's record lists (0: the shape runs on kernels without it - small graphs,
 * d not 4 * 2^k <= 256 - and a call with Y2_colsum set returns GODE_E_UNSUPPORTED) */
int64_t gode_spmm_y2_colsum_rows(int64_t n_items, int64_t n_long, int64_t d);

int gode_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                      const int32_t* items, int64_t n_items,
                      const int32_t* long_rows, int64_t n_long, float* partial,
                      const float* X, int64_t ldx, float* Y, int64_t ldy,
                      int64_t n_rows, int64_t d,
                      const gode_spmm_epilogue_t* epi /* host, nullable */,
                      void* stream);

/* ---- Runge-Kutta elementwise steps -------------------------------------- */
/* out[i] = sum_j lc.coef[j]*lc.ptr[j][i],  i < n.  out may alias any ptr[j]. */
int gode_lincomb_f32(float* out, const gode_lincomb_t* lc /* host */, int64_t n, void* stream);
/* up to four such combinations of different lengths in one launch (the components of an adjoint state) */
int gode_lincomb_multi_f32(float* const* outs /* host[count] */, const gode_lincomb_t* lcs /* host[count] */,
                           const int64_t* ns /* host[count] */, int32_t count, void* stream);

/* dopri5 acceptance test (torchdiffeq _compute_error_ratio):
 *   err = sum_j elc.coef[j]*elc.ptr[j][i];  tol = atol + rtol*max(|y0[i]|,|y1[i]|)
 *   out[0] = sum_i (err/tol)^2   (fp64, deterministic two-stage reduction)
 * scratch: >= gode_rk_errnorm_scratch_bytes() bytes. */
int64_t gode_rk_errnorm_scratch_bytes(void);
int gode_rk_errnorm_f32(double* out, const float* y0, const float* y1,
                        const gode_lincomb_t* elc /* host */, float rtol, float atol,
                        int64_t n, void* scratch, void* stream);
/* up to four such sums (the components of an adjoint state) in one pair of launches; out[count]; every sum is bit for bit
 * the one gode_rk_errnorm_f32 forms.  scratch: gode_rk_errnorm_scratch_bytes() */
int gode_rk_errnorm_multi_f32(double* out, const float* const* y0 /* host[count] */, const float* const* y1,
                              const gode_lincomb_t* elcs /* host[count] */, const int64_t* ns, int32_t count, float rtol,
                              float atol, void* scratch, void* stream);

/* sum of squares of sum_j lc.coef[j]*lc.ptr[j][i] / (atol + rtol*|y[i]|) (initial-step heuristic), fp64 */
int gode_rk_scaled_sumsq_f32(double* out, const gode_lincomb_t* lc /* host */, const float* y,
                             float rtol, float atol, int64_t n, void* scratch, void* stream);

/* ---- ODEfunc dense part: S = [t | GroupNorm(x)] * W ---------------------
 * x(i,:) = sum_j xin.coef[j]*xin.ptr[j][i,:]   (stage input formed on the fly)
 * xn = GroupNorm(groups, eps)(x) * gamma + beta   (per row; gamma/beta NULL = identity affine;
 *                                                  groups == 0 = no normalisation)
 * S  = t * W[0,:] + xn * W[1:,:]     W is (d_in+1) x d_out row-major  (has_time = 1)
 * S  =              xn * W          W is  d_in    x d_out             (has_time = 0)
 */
int gode_gn_time_gemm_f32(const gode_lincomb_t* xin /* host */, int64_t n_rows, int64_t d_in,
                          int32_t groups, float eps, const float* gamma, const float* beta,
                          const float* W, int64_t d_out, int has_time, float t,
                          float* S, void* stream);
/* Same, and additionally x_out[i,:] = x(i,:) (the combined, un-normalised stage input; nullable, 16-byte aligned,
 * n_rows x d_in contiguous) so that later kernels of the same stage read one array instead of the term list. */
int gode_gn_time_gemm_xout_f32(const gode_lincomb_t* xin /* host */, int64_t n_rows, int64_t d_in,
                               int32_t groups, float eps, const float* gamma, const float* beta,
                               const float* W, int64_t d_out, int has_time, float t,
                               float* S, float* x_out, void* stream);

/* Two square products (d_out = d_in = d) over the same input in one launch: Sa = [t | xn] Wa, Sb = [t | xn] Wb
 * (the two message projections of the GAT ODE function); x_out as above. */
int gode_gn_time_gemm_pair_f32(const gode_lincomb_t* xin /* host */, int64_t n_rows, int64_t d, int32_t groups, float eps,
                               const float* gamma, const float* beta, const float* Wa, const float* Wb,
                               int has_time, float t, float* Sa, float* Sb, float* x_out, void* stream);

/* VJP of the above w.r.t. x:  dxn = dS * W[has_time:,:]^T ; dx = GroupNorm'(x)^T dxn
 * out[i,:] = (sum_j pre.coef[j]*pre.ptr[j][i,:]) + out_scale * dx[i,:]
 *   (out_scale = 1, pre = NULL for a plain VJP; pre folds the RK combine of the adjoint state).
 * Also accumulates (when non-NULL) the per-row reductions needed for the
 * parameter gradients into fp32 buffers of block partials:
 *   dgamma_part/dbeta_part : [n_part][d_in]   (n_part = gode_gemm_bwd_parts(n_rows))
 */
int64_t gode_gemm_bwd_parts(int64_t n_rows);
int gode_gn_time_gemm_bwd_f32(const gode_lincomb_t* xin /* host */, int64_t n_rows, int64_t d_in,
                              int32_t groups, float eps, const float* gamma,
                              const float* W, int64_t d_out, int has_time,
                              const float* dS, float out_scale,
                              const gode_lincomb_t* pre /* host, nullable */, float* dx,
                              float* dgamma_part, float* dbeta_part, void* stream);

/* The VJP above AND the weight gradient (gode_wgrad_f32) from ONE read of x and dS (csrc/gemm_pc.hip; round 4): per adjoint
 * stage the two launches read both N x d arrays twice.  Outputs as theirs: dx (+ pre), dgamma_part / dbeta_part
 * [gode_gemm_bwd_parts(n_rows)][d] (both nullable together), dW_part [gode_bwd_wgrad_parts(n_rows)][(d + has_time) * d]
 * (row 0 of a partial with has_time: column sums of dS).  Only where gode_bwd_wgrad_supported says so (d = 128, 0 or 4
 * channels per group, >= 65 536 rows, 16-byte aligned operands); else GODE_E_UNSUPPORTED and the caller issues the pair. */
int gode_bwd_wgrad_supported(int64_t n_rows, int64_t d_in, int64_t d_out, int32_t groups);
int64_t gode_bwd_wgrad_parts(int64_t n_rows);
int gode_gn_time_gemm_bwd_wgrad_f32(const gode_lincomb_t* xin /* host */, int64_t n_rows, int64_t d_in, int32_t groups,
                                    float eps, const float* gamma, const float* beta, const float* W, int64_t d_out,
                                    int has_time, const float* dS, float out_scale, const gode_lincomb_t* pre /* host, nullable */,
                                    float* dx, float* dgamma_part, float* dbeta_part, float* dW_part, void* stream);

/* Several block-partial reductions in ONE launch (what closes an adjoint stage on launch-bound graphs: weight-gradient
 * partials, bias column sums, GroupNorm affine partials, time-row bookkeeping).  Segment s:
 *   out[j] = sum_p part[p*ld + col0 + j*col_stride]   for j < len  (fixed summation order: deterministic);
 * with w_row0 (time row): outputs j < time_len are written as t * sum and
 *   *at = sum over the time-row segments of sum_j<time_len (their unscaled sums)[j] * w_row0[j]   (= -a^T df/dt).
 * gode_colsum_parts_f32 is the first half of gode_colsum_f32: block partials of the column sums into `scratch`
 * (*n_parts rows of d floats), to be closed by a segment here. */
#define GODE_MAX_REDUCE_SEGS 8
typedef struct gode_reduce_seg {
    float* out; const float* part;
    int64_t n_part, ld, col0, col_stride, len;
    const float* w_row0; int64_t time_len;      /* w_row0 NULL: plain segment */
} gode_reduce_seg_t;
int gode_reduce_segments_f32(const gode_reduce_seg_t* segs, int32_t n_segs, float t, float* at /* nullable */, void* stream);
int gode_colsum_parts_f32(const float* X, int64_t n_rows, int64_t d, float* scratch, int64_t* n_parts /* host */, void* stream);

/* Stand-alone GroupNorm(groups, d) on an n_rows x d matrix (the reference applies nn.GroupNorm to 2-D
 * node-feature tensors: GCN/models.py:88,133-156,565-575) and its backward; block partials of dgamma /
 * dbeta have gode_group_norm_parts(n_rows) rows. */
int64_t gode_group_norm_parts(int64_t n_rows);
int gode_group_norm_f32_fwd(const float* x, int64_t n_rows, int64_t d, int32_t groups, float eps,
                            const float* gamma /* nullable */, const float* beta /* nullable */, float* y, void* stream);
int gode_group_norm_f32_bwd(const float* x, int64_t n_rows, int64_t d, int32_t groups, float eps,
                            const float* gamma /* nullable */, const float* dy, float* dx,
                            float* dgamma_part /* nullable */, float* dbeta_part /* nullable */, void* stream);

/* dW = [1 | xn]^T * dS  ((d_in+has_time) x d_out) as n_part block partials
 * dW_part[n_part][(d_in+has_time)*d_out]; the caller sums over parts
 * (gode_reduce_parts_f32).  With has_time, row 0 holds the plain column sums of dS
 * (the gradient w.r.t. a unit time column): dW[0,:] = t * row0 and dL/dt = row0 . W[0,:]. */
int64_t gode_wgrad_parts(int64_t n_rows);
int gode_wgrad_f32(const gode_lincomb_t* xin /* host */, int64_t n_rows, int64_t d_in,
                   int32_t groups, float eps, const float* gamma, const float* beta,
                   const float* dS, int64_t d_out, int has_time,
                   float* dW_part, void* stream);

/* The rectangular dense products of GraphConvolution (GCN/layers.py:32 `support = torch.mm(input, self.weight)` with
 * in_features != out_features, and its autograd), exact fp32 (v_mfma_f32_16x16x4_f32).  Row-major operands with leading
 * dimensions; any K, M >= 1; rows 16-byte aligned with ld % 4 == 0 take the vector path.
 *   gode_rect_gemm_f32:    S[n x M] = X[n x K] W[K x M]; columns M .. lds-1 of S are written as zeros (a caller pads
 *                          M = 7 to lds = 8 so that the aggregation that follows runs on the 16-byte SpMM kernel);
 *   gode_rect_gemm_nt_f32: dX[n x K] = dS[n x M] W[K x M]^T;
 *   gode_rect_wgrad_f32:   part[p][K x M], p < gode_rect_wgrad_parts(n): block partials of X^T dS, summed by
 *                          gode_reduce_parts_f32 (M <= 128 per call). */
int gode_rect_gemm_f32(const float* X, int64_t ldx, int64_t n_rows, int64_t K, const float* W, int64_t M,
                       float* S, int64_t lds, void* stream);
int gode_rect_gemm_nt_f32(const float* dS, int64_t ldds, int64_t n_rows, int64_t M, const float* W, int64_t K,
                          float* dX, int64_t lddx, void* stream);
int64_t gode_rect_wgrad_parts(int64_t n_rows);
int gode_rect_wgrad_f32(const float* X, int64_t ldx, int64_t n_rows, int64_t K, const float* dS, int64_t ldds,
                        int64_t M, float* part, void* stream);
/* gode_rect_wgrad_f32 followed by the sum of its partials into dW (K x M, contiguous): one call */
int gode_rect_wgrad_sum_f32(const float* X, int64_t ldx, int64_t n_rows, int64_t K, const float* dS, int64_t ldds, int64_t M,
                            float* part, float* dW, void* stream);

/* C[M x N] = op(A) op(B) (+ bias[col]) (relu) (* [mask[row][col] > 0]) on the exact fp32 matrix instruction: the dense
 * products of the QC edge encoder (QC/layers.py:46-86: relu(e W1 + b1) W2 + b2 on the edge rows of a batch, and their
 * autograd dA W2^T * [H > 0], H^T dA, e^T dH).  trans_a = 0: A is M x K row-major (lda >= K); 1: A is K x M (lda >= M).
 * trans_b = 0: B is K x N (ldb >= N); 1: B is N x K (ldb >= K).  bias, mask nullable.  Any sizes / leading dimensions. */
int gode_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu,
                  const float* mask, int64_t ldmask, void* stream);
/* The same product for a TALL contraction with few output tiles (gode_gemm_splitk_parts(M, N, K) > 1: at most 64 tiles of
 * 64 x 64 and K >= 192 - the weight gradients x^T dy of small layers): the contraction is cut into that many parts,
 * part[z][M][N] (caller-owned, parts * M * N floats) receives the raw product of part z; the parts are added into C by
 * the same call, or by the caller (gode_reduce_parts_f32) when C is NULL.  No epilogue.  GODE_E_UNSUPPORTED when gode_gemm_splitk_parts is 1. */
int64_t gode_gemm_splitk_parts(int64_t M, int64_t N, int64_t K);
int gode_gemm_splitk_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                         const float* B, int64_t ldb, float* part, float* C /* nullable: M x N contiguous, receives the sum of
                         the parts (second launch of the same call) */, void* stream);

/* The same products for LARGE shapes on the bf16 matrix cores, from exact three-way cuts of the fp32 operands
 * (csrc/pgemm.hip; replaces the three 21.6 GFLOP products of the edge encoder, QC/layers.py:46-86 and autograd:
 * H W2, dA W2^T, H^T dA).  Step 1, once per matrix: gode_cut_bf16x3_f32 writes the three bf16 piece planes of X (R x C,
 * leading dimension ld, any alignment): planes = 3 x R_pad x C_pad bf16, R_pad = gode_cut_pad(R), C_pad = gode_cut_pad(C)
 * (multiples of 128), zero outside R x C; x = hi + mid + lo exactly.  16-byte aligned, caller-owned
 * (6 * R_pad * C_pad bytes).  Step 2: gode_pgemm_bf16x3 forms C = op(A) op(B) (fp32, M x N, leading dimension ldc) from the
 * planes of A AS STORED (M x K if trans_a = 0, K x M if 1) and of B AS STORED (K x N if trans_b = 0, N x K if 1), i.e. a cut
 * matrix serves every product it enters in either role.  products = 8 (every piece product >= 2^-32 of a product; an fp32
 * result) or 6 (also without mid*lo, lo*mid: 2^-24 each).  Epilogue as gode_gemm_f32: + bias[col], relu, * (mask > 0).
 * workspace: gode_pgemm_workspace_bytes(M, N, K) bytes of caller-owned scratch (16-byte aligned) or NULL - with it a
 * product of few 128 x 128 tiles (760 x 2667: 126 tiles for 256 CUs) splits its contraction over several blocks per tile,
 * which write raw partial sums there, added in fixed order by a finishing launch (deterministic, no atomics). */
int64_t gode_cut_pad(int64_t n);
int gode_cut_bf16x3_f32(const float* X, int64_t ld, int64_t R, int64_t C, void* planes, void* stream);
int64_t gode_pgemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gode_pgemm_bf16x3(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const void* A_planes,
                      const void* B_planes, float* C, int64_t ldc, const float* bias, int relu,
                      const float* mask, int64_t ldmask, int products, void* workspace, void* stream);

/* torch.optim.Adam's update (no amsgrad; L2 weight decay) of up to GODE_ADAM_MAX_TENSORS tensors in ONE launch - the
 * `optimizer.step()` of QC/train_egcn.py / GCN/train_res.py:97.  args (host; copied into the kernel arguments): device
 * addresses and lengths of the tensors; chunks (device): n_chunks x {int32 tensor, int32 0, int64 start}, the work
 * list, one entry per gode_adam_chunk() elements of a tensor; state (device): 3 floats {step, 1/(1-beta1^step),
 * 1/sqrt(1-beta2^step)}, advanced once per optimizer step by gode_adam_tick_f32 (a kernel: capturable). */
#define GODE_ADAM_MAX_TENSORS 64
typedef struct gode_adam_args {
    float*       param[GODE_ADAM_MAX_TENSORS];
    const float* grad[GODE_ADAM_MAX_TENSORS];
    float*       exp_avg[GODE_ADAM_MAX_TENSORS];
    float*       exp_avg_sq[GODE_ADAM_MAX_TENSORS];
    int64_t      len[GODE_ADAM_MAX_TENSORS];
} gode_adam_args_t;
int64_t gode_adam_chunk(void);
int gode_adam_tick_f32(float* state, float beta1, float beta2, void* stream);
int gode_adam_f32(const gode_adam_args_t* args /* host */, int32_t n_tensors, const void* chunks, int64_t n_chunks,
                  const float* state, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* out[j] (+)= scale * sum_p part[p*len + j]   (accumulate != 0 adds to out) */
int gode_reduce_parts_f32(float* out, const float* part, int64_t n_part, int64_t len,
                          float scale, int accumulate, void* stream);
/* the same for two (out, part) pairs of equal shape in one launch (e.g. dgamma and dbeta) */
int gode_reduce_parts2_f32(float* out_a, const float* part_a, float* out_b, const float* part_b,
                           int64_t n_part, int64_t len, float scale, int accumulate, void* stream);

/* column sums: out[c] (+)= scale * sum_i X[i,c]  (bias gradient) */
int gode_colsum_f32(float* out, const float* X, int64_t n_rows, int64_t d, float scale,
                    int accumulate, float* scratch /* >= gode_colsum_scratch_bytes */, void* stream);
int64_t gode_colsum_scratch_bytes(int64_t n_rows, int64_t d);

/* ---- GAT-style edge attention (GAT/layers.py:40-55, :104-120) --------------------------
 * The two Linear layers of the reference act on h = [x[src] | x[tgt]]; the caller applies them at node level and
 * hands over the projections split by role (gode_gat_proj_t below):
 *   z_e = ps[src_e, 0:o] + pt[tgt_e, 0:o] + bf        a_e = as[src_e] + at[tgt_e] + bw      amax = max_e a_e (GLOBAL, :47)
 *   w_e = exp(a_e - amax)
 *   out[v] = sum_k val[k]*w[e_k]*relu(z[e_k]) / (sum_k val[k]*w[e_k] + eps)    over the entries k of row v of Mtgt.
 * den_out[v] holds the denominator, w_out[e] the unnormalised weight (both needed by the backward).  The backward
 * returns dz (E x o: gradient w.r.t. z_e) and da (E: gradient w.r.t. a_e through w only); gode_gat_maxpath_f32 adds
 * the path through the global maximum and gode_gat_scatter_f32 / gode_spmm_csr_f32 sum both per node. */
int64_t gode_gat_logits_scratch_bytes(int64_t n_edges);

/* A CSR matrix with its balanced record list, as the multi-launch entry points take it. */
typedef struct gode_graph {
    const int32_t* rowptr; const int32_t* col; const float* val;      /* CSR, val nullable */
    const int32_t* items;  int64_t n_items;                             /* balanced records (nullable) */
    const int32_t* long_rows; int64_t n_long; float* partial;           /* split rows + their slab */
    int64_t n_rows; int64_t nnz;
} gode_graph_t;

/* Node-level projections.  Keeping the roles apart lets the caller produce the two d x d message blocks with the
 * square MFMA kernel (gode_gn_time_gemm_f32, once per block) and the two logit columns with the 2-column kernels. */
typedef struct gode_gat_proj {
    const float* ps; int64_t ld_s;      /* message part gathered by source, n x o, row stride ld_s */
    const float* pt; int64_t ld_t;      /* message part gathered by target */
    const float* as; const float* at;   /* logit parts, element v at as[v*ld_a] / at[v*ld_a] */
    int64_t ld_a;
} gode_gat_proj_t;
int gode_gat_logits_f32(const gode_gat_proj_t* proj, const float* bw, const int32_t* src, const int32_t* tgt,
                        int64_t n_edges, float* a, float* amax, float* scratch, void* stream);
/* mt: CSR by target over the edges (col = edge ids of each target in increasing order, or NULL when the edge list is
 * target-sorted so that edge k is CSR position k), val = Mtgt values (nullable), plus the balanced record list.
 * Above 65 536 targets, with col == NULL and o in {16, 32, 64, 128, 256}, the aggregation runs over the records like
 * the SpMM (streamed per-edge arrays, one coalesced o*4-byte gather per edge); mt->partial must then hold
 * n_slots * (o + 4) floats.  Otherwise a lane group (a whole wave below 65 536 targets) works on one target. */
int gode_gat_agg_f32_fwd(const gode_graph_t* mt, const int32_t* src, const int32_t* tgt,
                         const gode_gat_proj_t* proj, int64_t o, const float* bf, const float* a, const float* amax,
                         float eps, float* out, float* w_out, float* den_out, void* stream);
/* cotangent: dout (n_rows x o), or - when cot (host, nullable) has terms - cot_scale * (sum_j cot_j) masked by out > 0,
 * i.e. the stage cotangent of the adjoint solve pushed through the relu that follows the layer.
 * dz[E x o], da[E] are always written.  dpt / dat / did_target_sums (all nullable): on the record path the kernel also
 * forms dpt[v,:] = sum_{e: tgt_e = v} dz[e,:] and dat[v*ld_dat] = sum da[e] and sets *did_target_sums = 1 (host int);
 * otherwise it leaves them untouched and sets 0 (use gode_gat_scatter_f32). */
int gode_gat_agg_f32_bwd(const gode_graph_t* mt, const int32_t* src, const int32_t* tgt,
                         const gode_gat_proj_t* proj, int64_t o, const float* bf, const float* w, const float* den,
                         const float* out, const float* dout, const gode_lincomb_t* cot, float cot_scale,
                         float* dz, float* da, float* dpt, int64_t ld_dpt, float* dat, int64_t ld_dat,
                         int32_t* did_target_sums, void* stream);
/* gradient path through the global maximum of the logits (GAT/layers.py:47): da[e*] -= S, S = sum(da), e* = first
 * argmax; when dat (nullable, with tgt) holds target-side sums formed before the correction, dat[tgt[e*]*ld_dat] -= S.
 * scratch: >= gode_gat_maxpath_scratch_bytes(n_edges) bytes (may be NULL for n_edges <= 32 768 without dat). */
int64_t gode_gat_maxpath_scratch_bytes(int64_t n_edges);
int gode_gat_maxpath_f32(const float* a, const float* amax, float* da, int64_t n_edges, const int32_t* tgt,
                         float* dat, int64_t ld_dat, void* scratch, void* stream);
/* H independent heads on one graph run as ONE head on the H-fold graph (node v*H + h = head h of node v, so an
 * N x H*o projection matrix is the (N*H) x o matrix of the virtual nodes; edge e becomes the H edges
 * src*H+h -> tgt*H+h, target-sorted) through the entry points above with amax -> 0.f.  The reference's maximum
 * (GAT/layers.py:47) is per head, so the logits are computed and shifted by their head's maximum here
 * (a_e = as[src_e] + at[tgt_e] + bw[h] - max over the edges of head h = tgt_e % H), hmax[heads] (nullable)
 * receives the maxima, and the gradient path through the maximum is per head: for every head da[e*] -= S_h with e* the
 * first edge of the head whose shifted logit is 0 (and dat[tgt[e*]*ld_dat] -= S_h when dat is given).
 * heads <= 64; scratch >= gode_gat_heads_scratch_bytes(n_edges, heads) bytes. */
int64_t gode_gat_heads_scratch_bytes(int64_t n_edges, int64_t heads);
int gode_gat_logits_heads_f32(const gode_gat_proj_t* proj, const float* bw /* nullable: heads logit biases */,
                              const int32_t* src, const int32_t* tgt, int64_t n_edges, int64_t heads, float* a,
                              float* hmax, void* scratch, void* stream);
int gode_gat_maxpath_heads_f32(const float* a, float* da, int64_t n_edges, int64_t heads, const int32_t* tgt,
                               float* dat, int64_t ld_dat, void* scratch, void* stream);
/* The same three steps WITHOUT the launch that shifts the logits (launch-bound graphs): logits_heads_raw leaves raw logits in
 * a and per-block partial maxima in `scratch` (which must then stay untouched until the stage's max-path step);
 * agg_heads_fwd is gode_gat_agg_f32_fwd on the H-fold graph with every virtual row shifted by its head's maximum, reduced
 * from those partials inside the kernel (same arithmetic: exp(a - max)); maxpath_heads_raw is the max-path step on raw
 * logits (e* = the first edge whose logit EQUALS its head's maximum).  Up to 65 536 virtual rows. */
int64_t gode_gat_heads_parts(int64_t n_edges);
int gode_gat_logits_heads_raw_f32(const gode_gat_proj_t* proj, const float* bw, const int32_t* src, const int32_t* tgt,
                                  int64_t n_edges, int64_t heads, float* a, void* scratch, void* stream);
int gode_gat_agg_heads_f32_fwd(const gode_graph_t* mt, const int32_t* src, const int32_t* tgt, const gode_gat_proj_t* proj,
                               int64_t o, const float* bf, const float* a, const void* scratch, int64_t n_edges,
                               int64_t heads, float eps, float* out, float* w_out, float* den_out, void* stream);
int gode_gat_maxpath_heads_raw_f32(const float* a, float* da, int64_t n_edges, int64_t heads, const int32_t* tgt,
                                   float* dat, int64_t ld_dat, void* scratch, void* stream);
/* the first half of maxpath_heads_raw only (per-block sums and arg-max candidates left in `scratch`): the step is closed by
 * gode_gat_dense_vjp_small_f32(..., maxpath_scratch = scratch, ...) */
int gode_gat_maxpath_heads_part_f32(const float* a, const float* da, int64_t n_edges, int64_t heads, const int32_t* tgt,
                                    void* scratch, void* stream);
int64_t gode_gat_heads_block_cap(void);
/* dps[v,:] = sum_{e: src_e = v} dz[e,:], dpt[v,:] = sum_{e: tgt_e = v} dz[e,:], das / dat likewise from da; the
 * incidence lists are CSR (rowptr over nodes, eid = edge ids in increasing order). */
int gode_gat_scatter_f32(const int32_t* rowptr_src, const int32_t* eid_src, const int32_t* rowptr_tgt,
                         const int32_t* eid_tgt, const float* dz, const float* da, int64_t o, int64_t n_rows,
                         float* dps, int64_t ld_s, float* dpt, int64_t ld_t, float* das, float* dat, int64_t ld_a,
                         void* stream);
/* time row of a weight gradient inside the adjoint: at (+)= <g_row0, w_row0>;  g_row0 *= t  (len floats) */
int gode_time_row_fixup_f32(float* g_row0, const float* w_row0, int64_t len, float t, float* at, int accumulate,
                            void* stream);
/* three weight blocks in one launch: at = <g0,w0> + <g1,w1> + <g2,w2>;  every g_b *= t */
int gode_time_row_fixup3_f32(float* g0, const float* w0, int64_t l0, float* g1, const float* w1, int64_t l1,
                             float* g2, const float* w2, int64_t l2, float t, float* at, void* stream);

/* ---- QC edge-conditioned messages (QC/mpnn.py:27-29, QC/layers.py:143-145) ----------------
 * out[v,:] = sum_k val[k] * A[e_k] (h x h, row-major) * X[src[e_k], :]   over row v of Etgt (CSR).
 * Backward (one block per edge): dm = edge_val[e] * dM[edge_row[e], :] (edge_row < 0: edge unused),
 *   dA[e] = dm (x) X[src[e]]   (nullable),   dxe[e,:] = A[e]^T dm   (nullable; the caller sums dxe
 *   per source node with gode_spmm_csr_f32 over the src incidence matrix). */
int gode_edge_matvec_f32_fwd(const int32_t* rowptr, const int32_t* eid, const float* val,
                             const int32_t* src, const float* A, const float* X, int64_t ldx,
                             int64_t h, int64_t n_rows, float* out, int64_t ldo, void* stream);
/* msg[e,:] = A[e] * X[src[e],:] for every edge (n_edges x h); with gode_spmm_csr_f32 over Etgt this is the two-launch
 * form of gode_edge_matvec_f32_fwd for large batches (a workgroup per edge instead of per target). */
int gode_edge_matvec_msg_f32(const int32_t* src, const float* A, const float* X, int64_t ldx, int64_t h,
                             int64_t n_edges, float* msg, void* stream);
int gode_edge_matvec_f32_bwd(const int32_t* edge_row, const float* edge_val, const int32_t* src,
                             const float* A, const float* X, int64_t ldx, const float* dM, int64_t ldm,
                             int64_t h, int64_t n_edges, float* dA, float* dxe, void* stream);
/* dA_e = sum_t (val_e dM[t][tgt_e]) (x) X[t][src_e]: the edge-matrix gradient of n_terms <= 8 message steps that used the same
 * edge matrices (QC/mpnn.py:27-30; QC/layers.py:143-145 inside a layer stack) in ONE pass - instead of n_terms calls of
 * gode_edge_matvec_f32_bwd with dA and n_terms - 1 additions (each step then asks that call for dxe only) */
int gode_edge_outer_sum_f32(const int32_t* edge_row, const float* edge_val /* nullable */, const int32_t* src, int32_t n_terms,
                            const float* const* dM /* host */, const float* const* X /* host */, int64_t ldx, int64_t ldm,
                            int64_t h, int64_t n_edges, float* dA, void* stream);

/* ---- Set2Set attention readout over the graphs of a batch (QC/set2set.py:59-75) --------------
 * The nodes of graph b are perm[segptr[b] .. segptr[b+1]) (perm nullable: node ids are the positions).
 *   e_i = <x_i, q_b>,  a = softmax(e) within the graph,  r_b = sum_i a_i x_i.
 * fwd writes a[N] (attention weight per node id) and r[n_seg x h]; a graph without nodes gives r_b = 0.
 * bwd takes dr[n_seg x h] and writes dx[N x h] (rows of nodes in segments; contiguous, ld = h) and dq[n_seg x h].
 * h <= 1024.  The reference evaluates this with a Python loop over the graphs. */
int gode_segment_attention_f32_fwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx,
                                   const float* q, int64_t n_seg, int64_t h, float* a, float* r, void* stream);
int gode_segment_attention_f32_bwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx,
                                   const float* q, const float* a, const float* dr, int64_t n_seg, int64_t h,
                                   float* dx, float* dq, void* stream);
/* ---- CSR of a small assignment matrix in one launch (csrc/convert.hip) -----------------------------------------
 * The (n_rows x n_entries) matrix with one entry per column e at row index[e] - edge -> target atom, edge -> source atom,
 * atom -> graph of a QC mini-batch (QC/datasets/utils.py:194-214) - as int32 CSR: rowptr[n_rows + 1], order[n_entries] =
 * the column ids grouped by row, ascending inside a row (stable), vals_out[k] = vals[order[k]] (both nullable; vals null,
 * vals_out not: ones).  Entries with an index outside [0, n_rows) are dropped (rowptr[n_rows] counts what is left).
 * One workgroup; n_entries <= 2048, n_rows <= 4096 (gode_assign_csr_supported), else GODE_E_UNSUPPORTED. */
int gode_assign_csr_supported(int64_t n_entries, int64_t n_rows);
/* index[c] = first row r with M[r][c] != 0 (0 for an all-zero column), M row-major n_rows x n_cols fp32 with leading dimension
 * ld: the reference collate's dense N x E target matrix (QC/datasets/utils.py:194-214) back to the per-edge target vector -
 * `(M != 0).to(uint8).argmax(0)` in one launch (round 4) */
int gode_dense_first_nonzero_f32(const float* M, int64_t ld, int64_t n_rows, int64_t n_cols, int64_t* index, void* stream);
int gode_assign_csr_i32(const int64_t* index, int64_t n_entries, int64_t n_rows, int32_t* rowptr, int32_t* order,
                        const float* vals, float* vals_out, void* stream);

/* ---- the whole Set2Set readout loop (QC/set2set.py:50-75) as ONE launch per direction (csrc/set2set.hip) ----------
 * processing_steps rounds of { q_t = LSTMCell(q*_{t-1}); r_t = segment attention of x with query q_t; q*_t = [q_t | r_t] }.
 * Row b of every quantity depends on the nodes of graph b only, so one workgroup walks graph b through all rounds.
 * Wt: the transposed weights [W_ih | W_hh]^T, (3H) x (4H) row-major (gate order i, f, g, o as torch.nn.LSTM); b_ih, b_hh
 * nullable.  fwd writes (caller-owned, any previous content): qs (steps+1) x n_graphs x 2H with qs[t+1] = q*_t (qs[steps]
 * is the module's output), cs (steps+1) x n_graphs x H, gates steps x n_graphs x 4H (activations), att steps x n_nodes.
 * bwd takes dq_final (n_graphs x 2H) and those arrays; writes dx (n_nodes x H, contiguous) and DG (steps x n_graphs x 4H,
 * the gate cotangents).  The weight gradients follow without another pass over the nodes:
 * dW_ih = DG^T QS (rows t, b: DG[t][b] and qs[t][b]), dW_hh = dW_ih[:, :H] (h_{t-1} is the left half of q*_{t-1}),
 * db_ih = db_hh = column sums of DG.  H <= 512. */
int gode_set2set_supported(int64_t H);
int gode_set2set_f32_fwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx, const float* Wt,
                         const float* b_ih, const float* b_hh, int64_t n_graphs, int64_t H, int64_t steps, int64_t n_nodes,
                         float* qs, float* cs, float* gates, float* att, void* stream);
int gode_set2set_f32_bwd(const int32_t* segptr, const int32_t* perm, const float* x, int64_t ldx, const float* w_ih,
                         const float* w_hh, int64_t n_graphs, int64_t H, int64_t steps, int64_t n_nodes, const float* qs,
                         const float* cs, const float* gates, const float* att, const float* dq_final, float* dx,
                         float* DG, void* stream);

/* ---- whole rk4 integrations of the GCN ODE function in one call (host-launch-bound sizes) -----
 * f(t, x) = relu(A * ([t | GroupNorm(x)] * W) + b)   (ODEfunc.forward, GCN/models.py:172-179).
 * 3/8-rule steps on a uniform grid from t0 to t1 (t1 < t0 for the adjoint pass); every launch goes to
 * `stream`, nothing is allocated or synchronised, so the call can be captured into a HIP graph.
 * The solution / stage buffers swap roles from step to step: the final state is returned through
 * the *_result pointers (each equals one of the buffers passed in).
 * Adjoint: (y, a, theta) integrate d/dt [y, a, theta] = [f, -a^T df/dy, -a^T df/dtheta]; `theta` is the
 * packed vector [dW ((d+1)*d) | db (d) | dgamma (d) | dbeta (d) | dt (1)] of gode_gcn_ode_theta_len(d)
 * floats and is updated in place.  With ws->S2 set, the adjoint runs its forward-recompute chain and its
 * VJP chain on two streams (one private side stream per process, joined back into `stream` before the
 * call returns; GODE_OVERLAP=0 disables it): one adjoint solve at a time per process. */

typedef struct gode_gcn_odefunc {
    gode_graph_t A, AT;                 /* adjacency and its transpose */
    int64_t n, d;                       /* nodes, hidden width */
    int32_t groups; float eps;          /* GroupNorm(groups, d), eps */
    const float* W;                     /* (d+1) x d, row 0 = time row */
    const float* b; const float* gamma; const float* beta;
} gode_gcn_odefunc_t;

typedef struct gode_rk4_workspace {
    float* S; float* dZ; float* dS;     /* n x d each (dZ, dS: adjoint only) */
    float* S2;                          /* nullable: second n x d buffer; enables the two-stream adjoint schedule */
    float* ky[4]; float* ka[4];         /* n x d stage buffers (ka: adjoint only) */
    float* ktheta[4];                   /* gode_gcn_ode_theta_len(d) floats each (adjoint only) */
    float* wpart;                       /* gode_wgrad_parts(n) * (d+1)*d floats */
    float* gpart; float* bpart;         /* gode_gemm_bwd_parts(n) * d floats each */
    float* colsum_scratch;              /* gode_colsum_scratch_bytes(n, d) bytes */
    float* X[2];                        /* nullable pair of n x d buffers (adjoint): the combined input of a stage with
                                           3 or 4 terms is written once by its forward launch and read as one array by
                                           the VJP and weight-gradient launches */
    float* small_part;                  /* nullable: 6 * gode_gcn_small_parts(n) * gode_gcn_small_part_len(d) floats (rk4 uses 4 of
                                           the 6 slots, a dopri5 step all of them) - enables
                                           the fused launch-bound path of the adjoint drivers (csrc/small.hip) */
    float* y2_colsum;                   /* nullable: gode_spmm_y2_colsum_rows(A.n_items, A.n_long, d) * d floats - the bias
                                           gradient of an adjoint stage is reduced from the per-block column sums the
                                           forward-recompute SpMM leaves there instead of from dZ */
} gode_rk4_workspace_t;

/* Launch-bound graphs (n <= 65 536, d in {16, 32}, 1 / 2 / 4 channels per GroupNorm group): ODEfunc.forward
 * (GCN/models.py:172-179) as ONE launch and its VJPs as one more, by re-association - z_i = (sum_j a_ij [t | GN(x_j)]) W + b,
 * everything after the gather row-local (csrc/small.hip).  gode_gcn_feval_small_f32: out = (sum pre) + alpha * relu(z),
 * Y2 (nullable) = (sum cot) * [z > 0].  gode_gcn_vjp_small_f32: ka = (sum pre) + out_scale * GN'(x)^T ((A^T dZ) W1^T) and
 * one block partial row per block in `part` (gode_gcn_small_parts(n) rows of gode_gcn_small_part_len(d) floats:
 * [ [1|xn]^T dS ((d+1) x d) | colsum(dZ) | dgamma | dbeta | the block's share of a_t ]); gode_gcn_small_finish_f32 adds them into a theta-k
 * vector [W | b | gamma | beta | a_t] (row 0 of W scaled by t, a_t = colsum(dS) . W[0,:]).  The rk4 / dopri5 drivers below
 * take this path by themselves (option "small_fused", default 1).  GODE_E_UNSUPPORTED outside the shapes above. */
int     gode_gcn_small_supported(int64_t n_rows, int64_t d, int32_t groups);
int64_t gode_gcn_small_parts(int64_t n_rows);
int64_t gode_gcn_small_part_len(int64_t d);
int gode_gcn_feval_small_f32(const gode_gcn_odefunc_t* f, const gode_lincomb_t* xin /* host */, float t, float alpha,
                             const gode_lincomb_t* pre /* host, nullable */, const gode_lincomb_t* cot /* host, with Y2 */,
                             float* Y2 /* nullable */, float* out, void* stream);
/* the same launch, which also writes the NEXT stage's combined input x_next = sum_j next.coef[j] * next.ptr[j] row by row
 * (a term that names `out` is this launch's result): the next evaluation then gathers one array per neighbour instead
 * of one per term of its stage input.  x_next must not be `out` or a term of xin.  next, x_next nullable together. */
int gode_gcn_feval_small_next_f32(const gode_gcn_odefunc_t* f, const gode_lincomb_t* xin /* host */, float t, float alpha,
                                  const gode_lincomb_t* pre /* host, nullable */, const gode_lincomb_t* cot /* host, with Y2 */,
                                  float* Y2 /* nullable */, float* out, const gode_lincomb_t* next /* host */, float* x_next,
                                  void* stream);
int gode_gcn_vjp_small_f32(const gode_gcn_odefunc_t* f, const gode_lincomb_t* xin /* host */, const float* dZ,
                           float out_scale, const gode_lincomb_t* pre /* host, nullable */, float* ka, float* part,
                           void* stream);
int gode_gcn_small_finish_f32(const gode_gcn_odefunc_t* f, const float* part, float* ktheta, float t, void* stream);
/* n_stages <= 8 stages in one launch: stage s reads part + s * parts * part_len, writes ktheta[s] (time ts[s]); each result is
 * bit for bit gode_gcn_small_finish_f32's (the dopri5 step driver closes its six stages at the end of the step) */
int gode_gcn_small_finish_multi_f32(const gode_gcn_odefunc_t* f, const float* part, int32_t n_stages,
                                    float* const* ktheta /* host */, const float* ts /* host */, void* stream);
/* fixed-grid solves: the four stages of an RK step write their partials back to back (stage s at
 * part + s * parts * part_len) and ONE launch per step adds sum_s wb[s] * (stage derivative) to the packed small
 * components theta = [W | b | gamma | beta | a_t] (wb[s] = h * b_s, ts[s] = stage times; host arrays of 4) */
int gode_gcn_small_finish4_f32(const gode_gcn_odefunc_t* f, const float* part, float* theta, const float* wb /* host */,
                               const float* ts /* host */, void* stream);

int64_t gode_gcn_ode_theta_len(int64_t d);
int gode_gcn_ode_rk4_forward(const gode_gcn_odefunc_t* f, float* y, float** result,
                             const gode_rk4_workspace_t* ws, float t0, float t1, int32_t n_steps, void* stream);
int gode_gcn_ode_rk4_adjoint(const gode_gcn_odefunc_t* f, float* y, float* a, float* theta,
                             float** y_result, float** a_result,
                             const gode_rk4_workspace_t* ws, float t0, float t1, int32_t n_steps, void* stream);

/* One Dormand-Prince 5(4) step of the same ODE function per call (adaptive solves on launch-bound sizes; the controller
 * - step-size selection, accept / reject, interpolation - stays with the caller, as in torchdiffeq).
 * forward: k[0] = f(t, y) on entry (FSAL); on return k[1..6] hold the other stages, y1 the 5th-order solution and
 *   sums[0] = sum_i (err_i / (atol + rtol*max(|y_i|, |y1_i|)))^2 as an fp64 device scalar.
 * adjoint: the augmented state is (y, a, theta) with theta = [W | b | gamma | beta | a_t] packed
 *   (gode_gcn_ode_theta_len floats; ktheta stages likewise); h < 0.  sums[0..3] = the same sums for y, a, a_t and the
 *   flattened parameters (the tensors torchdiffeq's mixed-tolerance norm keeps apart).
 * ws as for the rk4 driver (S2 / X optional); err_scratch >= gode_rk_errnorm_scratch_bytes(). */
int gode_gcn_ode_dopri5_step_forward(const gode_gcn_odefunc_t* f, const float* y, float* const* k /* 7 */, float* y1,
                                     const gode_rk4_workspace_t* ws, double t, double h, float rtol, float atol,
                                     double* sums, void* err_scratch, void* stream);
int gode_gcn_ode_dopri5_step_adjoint(const gode_gcn_odefunc_t* f, const float* y, const float* a, const float* theta,
                                     float* const* ky /* 7 */, float* const* ka /* 7 */, float* const* ktheta /* 7 */,
                                     float* y1, float* a1, float* theta1, const gode_rk4_workspace_t* ws,
                                     double t, double h, float rtol, float atol, double* sums /* 4 */,
                                     void* err_scratch, void* stream);

/* The same for the GAT ODE function  f(t, x) = relu(EdgeAttention([t | GroupNorm(x)]))  (GAT/models.py:172-179 ->
 * GAT/layers.py:95-122), with the two Linear layers packed by role: Wsrc, Wtgt ((d+1) x d: message parts by source /
 * target, time row first) and Wlog ((d+1) x 2: the two logit columns).  Adjoint state: (y, a, a_t, theta) with
 * theta = [Wsrc | Wtgt | Wlog | bf | bw | gamma | beta]  (gode_gat_ode_theta_len floats). */
typedef struct gode_gat_odefunc {
    gode_graph_t mt;                    /* CSR of Mtgt (col NULL: target-sorted edge list), + record list on large graphs */
    gode_graph_t ms_inc, mt_inc;        /* incidence CSR by source / by target (pattern only; adjoint) */
    const int32_t* src; const int32_t* tgt; int64_t n_edges;
    int64_t n, d; int32_t groups; float eps_gn; float eps;
    const float* Wsrc; const float* Wtgt; const float* Wlog;
    const float* bf; const float* bw; const float* gamma; const float* beta;
    /* H heads side by side (graph_odenet_amd/gat_heads.py; 0 or 1: one head).  With heads = H > 1 the graphs, src, tgt
     * and n_edges above are those of the H-fold graph (virtual node v*H + h = head h of node v; n stays the number of
     * real nodes, d = H * o), Wlog is (d+1) x 2H, bw holds H logit biases, bf (d) is added to the target-side
     * projection, and theta = [Wsrc | Wtgt | Wlog | bf | bw (H) | gamma | beta] (gode_gat_ode_theta_len_heads). */
    int32_t heads;
    /* nullable: gode_gat_small_pack_f32 of (Wsrc, Wtgt, Wlog), refreshed with them - launch-bound graphs only */
    const float* Wpacked;
} gode_gat_odefunc_t;

typedef struct gode_gat_workspace {
    float* X; float* Ps; float* Pt; float* A2;          /* n x d, n x d, n x d, n x 2 */
    float* a; float* amax; float* wgt; float* den;      /* E, 1, E, n */
    void* logits_scratch;                               /* gode_gat_logits_scratch_bytes(E) */
    /* adjoint only */
    float* dz; float* da;                               /* E x d, E */
    float* dPs; float* dPt; float* dA2; float* pair;    /* n x d, n x d, n x 2, 2 floats */
    float* gp; float* bp;                               /* 3 * gode_gemm_bwd_parts(n) * d floats each */
    float* wp[3];                                       /* gode_wgrad_parts(n) * (d+1)*d, same, * (d+1)*2 */
    void* maxpath_scratch; void* colsum_scratch;        /* gode_gat_maxpath_scratch_bytes(E); gode_colsum_scratch_bytes(n, d) */
    /* heads > 1 only: A2 / dA2 are n x 2H; pair holds 2H floats; zeros (max(o, 1) floats, all 0, read-only);
     * heads_scratch >= gode_gat_heads_scratch_bytes(n_edges, heads) */
    const float* zeros; void* heads_scratch;
    /* second column-sum scratch (gode_colsum_scratch_bytes(n, 2 * max(heads, 1))): with it, and up to 65 536 rows, the
     * reductions that close an adjoint stage run as one launch (gode_reduce_segments_f32); NULL: separate launches */
    void* colsum_scratch2;
    /* nullable: gode_gat_small_parts(n, d) * gode_gat_small_part_len(d, heads) floats - with it, and where
     * gode_gat_small_supported, the dense half of a stage runs on the one-launch kernels below */
    float* small_part;
} gode_gat_workspace_t;

/* ---- GAT ODE function on launch-bound graphs: the row-local half of an evaluation / of an adjoint stage as ONE launch
 * each (csrc/gat_small.hip).  Replaces, for n <= 65 536 nodes, d in {16, 32, 64}, 1 / 2 / 4 channels per GroupNorm group
 * and up to 8 heads, the node-level forms of `self.f` and `self.w` of GAT/layers.py:43,45 (Wsrc, Wtgt: (d+1) x d, Wlog:
 * (d+1) x 2H, row 0 = the time column's weights; see gode_gat_odefunc_t) and their autograd:
 *   project    Ps = [t|GN(x)] Wsrc, Pt = [t|GN(x)] Wtgt (+ pt_bias, nullable: d floats), A2 = [t|GN(x)] Wlog (n x 2H);
 *              x = sum of the terms of `xin`, also written to x_out when given
 *   dense_vjp  ka = out_scale * GN'(x)^T (dPs Wsrc[1:]^T + dPt Wtgt[1:]^T + dA2 Wlog[1:]^T) + sum pre, and one partial row
 *              per block in `part`: [ dWsrc | dWtgt | dWlog (row 0 of each = the column sums of dPs / dPt / dA2) | dgamma |
 *              dbeta | colsums . time rows of the weights ]
 *   finish     ktheta = [Wsrc | Wtgt | Wlog | bf | bw | gamma | beta] (gode_gat_ode_theta_len_heads floats; time rows * t,
 *              bf = colsum(dPt), bw_h = colsum(dA2)[2h+1]) and *kat = the a_t derivative, from the partials (fixed order:
 *              deterministic) */
int     gode_gat_small_supported(int64_t n_rows, int64_t d, int32_t groups, int64_t heads);
int64_t gode_gat_small_parts(int64_t n_rows, int64_t d);
int64_t gode_gat_small_part_len(int64_t d, int64_t heads);
int gode_gat_project_small_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d, int32_t groups, float eps,
                               const float* gamma, const float* beta, const float* Wsrc, const float* Wtgt,
                               const float* Wlog, int64_t heads, const float* pt_bias, float t, float* Ps, float* Pt,
                               float* A2, float* x_out, const float* packed /* nullable: gode_gat_small_pack_f32 */, void* stream);
int gode_gat_dense_vjp_small_f32(const gode_lincomb_t* xin, int64_t n_rows, int64_t d, int32_t groups, float eps,
                                 const float* gamma, const float* beta, const float* Wsrc, const float* Wtgt,
                                 const float* Wlog, int64_t heads, const float* dPs, const float* dPt, const float* dA2,
                                 float out_scale, const gode_lincomb_t* pre, float* ka, float* part,
                                 const void* maxpath_scratch /* nullable: the scratch gode_gat_maxpath_heads_part_f32 filled;
                                 the per-head max-path sums are then taken off the two dA2 entries of each head's arg-max
                                 edge while the rows are loaded */, const int32_t* esrc, const int32_t* etgt, int64_t n_edges,
                                 const float* packed /* nullable: gode_gat_small_pack_f32 */, void* stream);
int gode_gat_small_finish_f32(const float* part, int64_t n_rows, int64_t d, int64_t heads, float t, float* ktheta,
                              float* kat, void* stream);
/* The closing launches of the n_slots <= 4 stages of ONE fixed-grid Runge-Kutta step as one: theta += sum_s w[s] k_theta(s),
 * a_t += sum_s w[s] k_at(s), stage s evaluated at ts[s], its partials at part + s * parts * part_len (round 4: the small
 * components of the adjoint state are linear in the ODE and never read inside a fixed-grid step). */
int gode_gat_small_finish_step_f32(const float* part, int64_t n_rows, int64_t d, int64_t heads, int32_t n_slots,
                                   const float* ts /* host */, const float* w /* host */, float* theta, float* at,
                                   void* stream);
/* The LDS images of [Wsrc | Wtgt | Wlog] both kernels stage (row-major with padded logit columns for project, transposed
 * without the time row for dense_vjp), formed once per solve - the weights do not change inside one - so that staging is
 * a straight 16-byte copy: packed holds gode_gat_small_pack_len(d, heads) floats.  Passing NULL for `packed` makes the
 * kernels lay the images out themselves (element by element). */
int64_t gode_gat_small_pack_len(int64_t d, int64_t heads);
int gode_gat_small_pack_f32(const float* Wsrc, const float* Wtgt, const float* Wlog, int64_t d, int64_t heads, float* packed,
                            void* stream);

int64_t gode_gat_ode_theta_len(int64_t d);
int64_t gode_gat_ode_theta_len_heads(int64_t d, int64_t heads);
int gode_gat_ode_dopri5_step_forward(const gode_gat_odefunc_t* f, const float* y, float* const* k /* 7 */, float* y1,
                                     const gode_gat_workspace_t* ws, double t, double h, float rtol, float atol,
                                     double* sums, void* err_scratch, void* stream);
int gode_gat_ode_dopri5_step_adjoint(const gode_gat_odefunc_t* f, const float* y, const float* a, const float* a_t,
                                     const float* theta, float* const* ky, float* const* ka, float* const* ka_t,
                                     float* const* ktheta, float* y1, float* a1, float* a_t1, float* theta1,
                                     const gode_gat_workspace_t* ws, double t, double h, float rtol, float atol,
                                     double* sums /* 4 */, void* err_scratch, void* stream);

/* ---- Set2Set readout: LSTM cell, one launch per direction (csrc/lstm.hip; replaces the single-layer `self.lstm` step of
 * QC/set2set.py:44-47,61 = torch.lstm_cell: two library GEMMs + a cell kernel forward, four GEMMs + reductions backward).
 * x: B x I, h, c: B x H, w_ih: 4H x I, w_hh: 4H x H, b_ih, b_hh: 4H (nullable), gate order i, f, g, o as torch.nn.LSTM.
 * fwd: h_out, c_out (B x H); gates (nullable; B x 4H) receives the four activations per unit - what the backward reads.
 * bwd: dh_out, dc_out (B x H, either nullable = zero) -> dx (B x I, nullable), dh (nullable), dc (B x H), dw_ih, dw_hh and
 * db_ih, db_hh (nullable).  Fixed summation order.  supported: B (4H + 5) + 16 H <= 24 576 and 6 (I + H) + 256 <= 24 576
 * (floats of LDS). */
int gode_lstm_cell_supported(int64_t B, int64_t I, int64_t H);
int gode_lstm_cell_f32_fwd(const float* x, const float* h, const float* c, const float* w_ih, const float* w_hh,
                           const float* b_ih, const float* b_hh, int64_t B, int64_t I, int64_t H, float* h_out,
                           float* c_out, float* gates, void* stream);
int gode_lstm_cell_f32_bwd(const float* x, const float* h, const float* c, const float* w_ih, const float* w_hh,
                           const float* gates, const float* c_out, const float* dh_out, const float* dc_out, int64_t B,
                           int64_t I, int64_t H, float* dx, float* dh, float* dc, float* dw_ih, float* dw_hh,
                           float* db_ih, float* db_hh, void* stream);
/* ---- QC node update: fused GRU cell (replaces nn.GRUCell(2h, h) applied to ([x | m], x), QC/mpnn.py:12,30) -----
 * x, m: n x h (state and aggregated messages; the concatenation [x | m] is never formed).  w_ih: 3h x 2h, w_hh: 3h x h,
 * b_ih, b_hh: 3h (nullable), gate order r, z, n as torch.nn.GRUCell.  out: n x h.  gates (nullable; n x 4h) receives
 * r, z, n and W_hn x + b_hn per row - what the backward pass reads.
 * bwd: dout n x h -> dx, dm (n x h, nullable), the gate derivatives dgi, dgh (n x 3h each, caller-owned, also outputs),
 * and the parameter gradients dw_ih, dw_hh, db_ih, db_hh (bias pointers nullable); `part` is scratch of
 * gode_gru_wgrad_parts(n) * 3h * (3h + 2) floats (row-chunk partials, added in fixed order: deterministic). */
int64_t gode_gru_wgrad_parts(int64_t n);
int gode_gru_cell_f32_fwd(const float* x, const float* m, const float* w_ih, const float* w_hh, const float* b_ih,
                          const float* b_hh, int64_t n, int64_t h, float* out, float* gates, void* stream);
int gode_gru_cell_f32_bwd(const float* x, const float* m, const float* w_ih, const float* w_hh, const float* gates,
                          const float* dout, int64_t n, int64_t h, float* dx, float* dm, float* dgi, float* dgh,
                          float* part, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, void* stream);
/* dw_ih = dw_hh = NULL above: the call stops after writing its gode_gru_wgrad_parts(n) partial rows (3h (3h + 2) floats each)
 * to `part`; gode_gru_wreduce_f32 sums n_part such rows - e.g. those of the T applications of one cell inside a message
 * passing loop, written back to back - into the four parameter gradients (db_* nullable) */
int gode_gru_wreduce_f32(const float* part, int64_t n_part, int64_t h, float* dw_ih, float* dw_hh, float* db_ih,
                         float* db_hh, void* stream);

/* ---- measurement aid (bench.py): HIP-event brackets around the dominant kernels ----------
 * While a profiler is enabled (process-wide; one measuring client at a time), every gode_spmm_csr_f32 main-kernel launch
 * and every MFMA-path launch of the dense kernels (gn_gemm_fwd / gn_gemm_bwd / wgrad) records a start/stop event pair
 * on the stream it is launched on (up to `capacity` launches).
 * gode_prof_read waits for the recorded events and returns the number of launches read, with
 * per-launch milliseconds, feature width d, record count (SpMM) or row count (dense) and `extra` = number of
 * additional n_rows x d operand arrays the launch read or wrote besides its plain operands (SpMM: pre terms +
 * cotangent terms + Y2; dense: stage terms beyond the first, + x_out, + pre terms).
 * gode_prof_kinds: what each launch was - GODE_PROF_SPMM / _GEMM_FWD / _GEMM_BWD / _WGRAD. */
#define GODE_PROF_SPMM     0
#define GODE_PROF_GEMM_FWD 1
#define GODE_PROF_GEMM_BWD 2
#define GODE_PROF_WGRAD    3
#define GODE_PROF_BWD_WGRAD 4      /* VJP + weight gradient in one pass (gode_gn_time_gemm_bwd_wgrad_f32): 2 x 2 N d^2 flop */
/* which kernel of the family ran, OR-ed into the kind of a dense launch (kind & 0xff = family, kind >> 8 = form):
 * exact-fp32 MFMA kernel; bf16-piece kernel, every wave loading + cutting + multiplying; bf16-piece kernel in
 * producer / consumer form (wgrad_split_kernel, gemm_pc.hip) */
#define GODE_PROF_FORM_FP32  (0 << 8)
#define GODE_PROF_FORM_SPLIT (1 << 8)
#define GODE_PROF_FORM_PC    (2 << 8)
void* gode_prof_create(int capacity);
void  gode_prof_destroy(void* prof);
void  gode_prof_enable(void* prof /* NULL = off */);
void  gode_prof_reset(void* prof);
int   gode_prof_count(void* prof);
int   gode_prof_read(void* prof, float* ms /* host */, int64_t* d /* host, nullable */,
                     int64_t* rows /* host, nullable */, int64_t* extra /* host, nullable */, int max_n);
int   gode_prof_kinds(void* prof, int32_t* kinds /* host */, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHODE_H */
