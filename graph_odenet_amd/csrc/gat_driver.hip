// gat_driver.hip — one Dormand-Prince 5(4) step of the GAT ODE function per C-ABI call.
//
// f(t, x) = relu(EdgeAttention([t | GroupNorm(x)]))  (reference: GAT/models.py:172-179 -> GAT/layers.py:95-122).
// The adaptive controller (step-size choice, accept / reject, interpolation) stays with the caller, as for the GCN
// function in ode_driver.hip; this file only sequences the launches of six stage evaluations, the solution combine
// and the error-ratio sums, so that an adaptive step on a citation-size graph is one call instead of ~200.
// The launch sequence of one evaluation is graph_odenet_amd/gat_ode.py (GatOdeField / GatOdeAdjointField).
#include "common.h"
#include "options.h"

namespace {

#define GODE_TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)
#define GODE_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return (int)e__; } while (0)

const double DPC[7] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
const double DPA[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84},
};
const double DPB[7] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0};
const double DPE[7] = {35.0 / 384 - 1951.0 / 21600, 0, 500.0 / 1113 - 22642.0 / 50085, 125.0 / 192 - 451.0 / 720,
                       -2187.0 / 6784 - -12231.0 / 42400, 11.0 / 84 - 649.0 / 6300, -1.0 / 60.0};

gode_lincomb_t dp_terms(const float* y, float* const* k, const double* coef, int count, double h, bool with_y, int64_t off = 0) {
    gode_lincomb_t lc;
    lc.n = 0;
    if (with_y) { lc.coef[0] = 1.f; lc.ptr[0] = y + off; lc.n = 1; }
    for (int j = 0; j < count; ++j)
        if (coef[j] != 0.0) { lc.coef[lc.n] = (float)(h * coef[j]); lc.ptr[lc.n] = k[j] + off; ++lc.n; }
    return lc;
}

inline int64_t n_heads(const gode_gat_odefunc_t* f) { return f->heads > 1 ? f->heads : 1; }
constexpr int64_t kMergedFinishMaxRows = 1 << 16;      // as graph_odenet_amd/gat_ode.py: MERGED_FINISH_MAX_ROWS

// projections as the edge kernels see them: with H heads the n x (H*o) matrices ARE the (n*H) x o matrices of the
// virtual nodes (row stride o), and A2 (n x 2H) is (n*H) x 2
gode_gat_proj_t proj_of(const gode_gat_odefunc_t* f, const gode_gat_workspace_t* w) {
    const int64_t o = f->d / n_heads(f);
    gode_gat_proj_t p;
    p.ps = w->Ps; p.ld_s = o; p.pt = w->Pt; p.ld_t = o; p.as = w->A2; p.at = w->A2 + 1; p.ld_a = 2;
    return p;
}

// Pt[row, :] += bf  (heads: the per-head message biases ride the target-side projection)
__global__ __launch_bounds__(256) void add_row_bias_kernel(float* __restrict__ P, const float* __restrict__ b, int64_t n, int d) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) P[i] += b[i % d];
}
// dst[h] = src[2h + 1]  (logit-bias gradients out of the column sums of dA2)
__global__ void odd_entries_kernel(float* __restrict__ dst, const float* __restrict__ src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[2 * i + 1];
}

inline bool small_dense(const gode_gat_odefunc_t* f) {
    return gode_opt_small_fused() && gode_gat_small_supported(f->n, f->d, f->groups, n_heads(f));
}

// the same condition as gat_heads.GatHeadsField.raw_logits (the Python and the C driver issue the same launches)
inline bool raw_logits(const gode_gat_odefunc_t* f) {
    return f->n * n_heads(f) <= 65536 && f->n_edges > 8192 && small_dense(f);
}

// Ps, Pt, A2 of the stage input; a multi-term input is combined once (x_out) and read back as one array afterwards
int project(const gode_gat_odefunc_t* f, const gode_gat_workspace_t* w, gode_lincomb_t* yin, float t, void* stream) {
    const int64_t n = f->n, d = f->d;
    float* xo = yin->n > 1 ? w->X : nullptr;
    const int64_t Hh = n_heads(f);
    if (small_dense(f)) {                       // launch-bound graphs: the three products (and the bias add) as one launch
        GODE_TRY(gode_gat_project_small_f32(yin, n, d, f->groups, f->eps_gn, f->gamma, f->beta, f->Wsrc, f->Wtgt, f->Wlog, Hh,
                                            Hh > 1 ? f->bf : nullptr, t, w->Ps, w->Pt, w->A2, xo, f->Wpacked, stream));
        if (xo) { yin->n = 1; yin->coef[0] = 1.f; yin->ptr[0] = xo; }
        return 0;
    }
    GODE_TRY(gode_gn_time_gemm_pair_f32(yin, n, d, f->groups, f->eps_gn, f->gamma, f->beta, f->Wsrc, f->Wtgt, 1, t, w->Ps, w->Pt,
                                        xo, stream));
    if (xo) { yin->n = 1; yin->coef[0] = 1.f; yin->ptr[0] = xo; }
    const int64_t H = n_heads(f);
    GODE_TRY(gode_gn_time_gemm_f32(yin, n, d, f->groups, f->eps_gn, f->gamma, f->beta, f->Wlog, 2 * H, 1, t, w->A2, stream));
    if (H > 1) {
        int64_t blocks = (n * d + 255) / 256; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(add_row_bias_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w->Pt, f->bf, n, (int)d);
        GODE_LAUNCH_CHECK();
    }
    return 0;
}

int eval_forward(const gode_gat_odefunc_t* f, const gode_gat_workspace_t* w, gode_lincomb_t* yin, float t, float* ky,
                 void* stream) {
    GODE_TRY(project(f, w, yin, t, stream));
    const gode_gat_proj_t pr = proj_of(f, w);
    const int64_t H = n_heads(f);
    if (H > 1) {     // logits shifted by their head's maximum (so the aggregation runs with amax = 0), biases in Pt
        if (raw_logits(f)) {     // launch-bound graphs: no launch that shifts the logits (csrc/edge.hip, HeadMax)
            GODE_TRY(gode_gat_logits_heads_raw_f32(&pr, f->bw, f->src, f->tgt, f->n_edges, H, w->a, w->heads_scratch, stream));
            return gode_gat_agg_heads_f32_fwd(&f->mt, f->src, f->tgt, &pr, f->d / H, w->zeros, w->a, w->heads_scratch, f->n_edges, H,
                                              f->eps, ky, w->wgt, w->den, stream);
        }
        GODE_TRY(gode_gat_logits_heads_f32(&pr, f->bw, f->src, f->tgt, f->n_edges, H, w->a, nullptr, w->heads_scratch, stream));
        return gode_gat_agg_f32_fwd(&f->mt, f->src, f->tgt, &pr, f->d / H, w->zeros, w->a, w->zeros, f->eps, ky, w->wgt, w->den,
                                    stream);
    }
    GODE_TRY(gode_gat_logits_f32(&pr, f->bw, f->src, f->tgt, f->n_edges, w->a, w->amax, (float*)w->logits_scratch, stream));
    return gode_gat_agg_f32_fwd(&f->mt, f->src, f->tgt, &pr, f->d, f->bf, w->a, w->amax, f->eps, ky, w->wgt, w->den, stream);
}

// theta-k layout: [Wsrc ((d+1)*d) | Wtgt ((d+1)*d) | Wlog ((d+1)*2H) | bf (d) | bw (H) | gamma (d) | beta (d)]   (H = 1: one head)
int eval_adjoint(const gode_gat_odefunc_t* f, const gode_gat_workspace_t* w, gode_lincomb_t yin, const gode_lincomb_t& ain,
                 float t, float* ky, float* ka, float* kat, float* kth, void* stream) {
    const int64_t H = n_heads(f);
    const int64_t n = f->n, d = f->d, o = d / H, nv = n * H, nW = (d + 1) * d, nL = (d + 1) * 2 * H;
    GODE_TRY(eval_forward(f, w, &yin, t, ky, stream));                  // yin now names the combined input
    const gode_gat_proj_t pr = proj_of(f, w);
    int32_t did = 0;
    GODE_TRY(gode_gat_agg_f32_bwd(&f->mt, f->src, f->tgt, &pr, o, H > 1 ? w->zeros : f->bf, w->wgt, w->den, ky, nullptr, &ain,
                                  -1.f, w->dz, w->da, w->dPt, o, w->dA2 + 1, 2, &did, stream));
    if (f->n_edges > 0) {
        if (H > 1 && raw_logits(f) && !did && w->small_part)     // first half; the dense VJP launch below closes the step
            GODE_TRY(gode_gat_maxpath_heads_part_f32(w->a, w->da, f->n_edges, H, f->tgt, w->heads_scratch, stream));
        else if (H > 1 && raw_logits(f))
            GODE_TRY(gode_gat_maxpath_heads_raw_f32(w->a, w->da, f->n_edges, H, f->tgt, did ? w->dA2 + 1 : nullptr, 2,
                                                    w->heads_scratch, stream));
        else if (H > 1)
            GODE_TRY(gode_gat_maxpath_heads_f32(w->a, w->da, f->n_edges, H, f->tgt, did ? w->dA2 + 1 : nullptr, 2,
                                                w->heads_scratch, stream));
        else
            GODE_TRY(gode_gat_maxpath_f32(w->a, w->amax, w->da, f->n_edges, did ? f->tgt : nullptr, did ? w->dA2 + 1 : nullptr, 2,
                                          w->maxpath_scratch, stream));
    }
    if (did) {
        GODE_TRY(gode_spmm_csr_f32(f->ms_inc.rowptr, f->ms_inc.col, nullptr, f->ms_inc.items, f->ms_inc.n_items,
                                   f->ms_inc.long_rows, f->ms_inc.n_long, f->ms_inc.partial, w->dz, o, w->dPs, o, nv, o,
                                   nullptr, stream));
        GODE_TRY(gode_spmm_csr_f32(f->ms_inc.rowptr, f->ms_inc.col, nullptr, f->ms_inc.items, f->ms_inc.n_items,
                                   f->ms_inc.long_rows, f->ms_inc.n_long, f->ms_inc.partial, w->da, 1, w->dA2, 2, nv, 1,
                                   nullptr, stream));
    } else {
        GODE_TRY(gode_gat_scatter_f32(f->ms_inc.rowptr, f->ms_inc.col, f->mt_inc.rowptr, f->mt_inc.col, w->dz, w->da, o, nv,
                                      w->dPs, o, w->dPt, o, w->dA2, w->dA2 + 1, 2, stream));
    }
    if (w->small_part && small_dense(f)) {      // k_a and all parameter-gradient partials in one launch, one more to close
        GODE_TRY(gode_gat_dense_vjp_small_f32(&yin, n, d, f->groups, f->eps_gn, f->gamma, f->beta, f->Wsrc, f->Wtgt, f->Wlog, H,
                                              w->dPs, w->dPt, w->dA2, 1.f, nullptr, ka, w->small_part,
                                              (H > 1 && raw_logits(f) && !did && f->n_edges > 0) ? w->heads_scratch : nullptr, f->src, f->tgt,
                                              f->n_edges, f->Wpacked, stream));
        return gode_gat_small_finish_f32(w->small_part, n, d, H, t, kth, kat, stream);
    }
    float* g_src = kth; float* g_tgt = kth + nW; float* g_log = kth + 2 * nW;
    float* g_bf = g_log + nL; float* g_bw = g_bf + d; float* g_gamma = g_bw + H; float* g_beta = g_gamma + d;
    const int64_t nb = gode_gemm_bwd_parts(n);
    const bool affine = f->groups > 0;
    const float* Wj[3] = {f->Wsrc, f->Wtgt, f->Wlog};
    const float* dPj[3] = {w->dPs, w->dPt, w->dA2};
    const int64_t dout[3] = {d, d, 2 * H};
    // launch-bound graphs: ONE reduction launch closes the stage (the same launch sequence as gat_ode.py / gat_heads.py)
    const bool merged = n <= kMergedFinishMaxRows && affine && w->colsum_scratch2 != nullptr;
    int64_t n_a = 0, n_b = 0;
    if (merged) {
        GODE_TRY(gode_colsum_parts_f32(w->dPt, n, d, (float*)w->colsum_scratch, &n_a, stream));
        GODE_TRY(gode_colsum_parts_f32(w->dA2, n, 2 * H, (float*)w->colsum_scratch2, &n_b, stream));
    } else {
        // bias gradients from the per-target node sums (every edge has exactly one target)
        GODE_TRY(gode_colsum_f32(g_bf, w->dPt, n, d, 1.f, 0, (float*)w->colsum_scratch, stream));
        GODE_TRY(gode_colsum_f32(w->pair, w->dA2, n, 2 * H, 1.f, 0, (float*)w->colsum_scratch, stream));
        hipLaunchKernelGGL(odd_entries_kernel, dim3((unsigned)((H + 63) / 64)), dim3(64), 0, (hipStream_t)stream, g_bw, (const float*)w->pair, (int)H);
        GODE_LAUNCH_CHECK();
    }
    gode_lincomb_t acc; acc.n = 1; acc.coef[0] = 1.f; acc.ptr[0] = ka;
    for (int j = 0; j < 3; ++j)
        GODE_TRY(gode_gn_time_gemm_bwd_f32(&yin, n, d, f->groups, f->eps_gn, f->gamma, Wj[j], dout[j], 1, dPj[j], 1.f,
                                           j ? &acc : nullptr, ka, affine ? w->gp + j * nb * d : nullptr,
                                           affine ? w->bp + j * nb * d : nullptr, stream));
    if (affine && !merged) GODE_TRY(gode_reduce_parts2_f32(g_gamma, w->gp, g_beta, w->bp, 3 * nb, d, 1.f, 0, stream));
    else if (!affine) GODE_TRY(gode_zero_f32(g_gamma, 2 * d, stream));
    float* gW[3] = {g_src, g_tgt, g_log};
    const int64_t lenW[3] = {nW, nW, nL};
    const int64_t npw = gode_wgrad_parts(n);
    for (int j = 0; j < 3; ++j)
        GODE_TRY(gode_wgrad_f32(&yin, n, d, f->groups, f->eps_gn, f->gamma, f->beta, dPj[j], dout[j], 1, w->wp[j], stream));
    if (merged) {
        gode_reduce_seg_t sg[7] = {};
        sg[0] = {g_src, w->wp[0], npw, nW, 0, 1, nW, f->Wsrc, d};                              // row 0 of each block: its time row
        sg[1] = {g_tgt, w->wp[1], npw, nW, 0, 1, nW, f->Wtgt, d};
        sg[2] = {g_log, w->wp[2], npw, nL, 0, 1, nL, f->Wlog, 2 * H};
        sg[3] = {g_bf, (const float*)w->colsum_scratch, n_a, d, 0, 1, d, nullptr, 0};
        sg[4] = {g_bw, (const float*)w->colsum_scratch2, n_b, 2 * H, 1, 2, H, nullptr, 0};      // odd columns of the n x 2H sums
        sg[5] = {g_gamma, w->gp, 3 * nb, d, 0, 1, d, nullptr, 0};
        sg[6] = {g_beta, w->bp, 3 * nb, d, 0, 1, d, nullptr, 0};
        return gode_reduce_segments_f32(sg, 7, t, kat, stream);
    }
    GODE_TRY(gode_reduce_parts2_f32(gW[0], w->wp[0], gW[1], w->wp[1], npw, lenW[0], 1.f, 0, stream));
    GODE_TRY(gode_reduce_parts_f32(gW[2], w->wp[2], npw, lenW[2], 1.f, 0, stream));
    // a_t' = -a^T df/dt over the three time rows; each row 0 *= t
    return gode_time_row_fixup3_f32(gW[0], Wj[0], dout[0], gW[1], Wj[1], dout[1], gW[2], Wj[2], dout[2], t, kat, stream);
}

int check_common(const gode_gat_odefunc_t* f, const gode_gat_workspace_t* w, bool adjoint) {
    if (!f || !w) return GODE_E_NULLPTR;
    if (f->n <= 0 || f->d <= 0 || f->n_edges < 0 || f->heads < 0) return GODE_E_SHAPE;
    if (f->heads > 1) {
        if (f->d % f->heads || f->heads > 64) return GODE_E_SHAPE;
        if (!w->zeros || !w->heads_scratch) return GODE_E_NULLPTR;
    }
    if (!f->Wsrc || !f->Wtgt || !f->Wlog || !f->bf || !f->bw || !f->mt.rowptr) return GODE_E_NULLPTR;
    if (!w->X || !w->Ps || !w->Pt || !w->A2 || !w->amax || !w->den || !w->logits_scratch) return GODE_E_NULLPTR;
    if (f->n_edges > 0 && (!f->src || !f->tgt || !w->a || !w->wgt)) return GODE_E_NULLPTR;
    if (adjoint) {
        if (!w->dPs || !w->dPt || !w->dA2 || !w->pair || !w->colsum_scratch || !w->maxpath_scratch || !w->wp[0] || !w->wp[1] ||
            !w->wp[2] || !f->ms_inc.rowptr || !f->mt_inc.rowptr) return GODE_E_NULLPTR;
        if (f->n_edges > 0 && (!w->dz || !w->da)) return GODE_E_NULLPTR;
        if (f->groups > 0 && (!w->gp || !w->bp)) return GODE_E_NULLPTR;
    }
    return 0;
}

}  // namespace

extern "C" int64_t gode_gat_ode_theta_len(int64_t d) { return 2 * (d + 1) * d + (d + 1) * 2 + d + 1 + 2 * d; }
extern "C" int64_t gode_gat_ode_theta_len_heads(int64_t d, int64_t heads) {
    if (heads < 1) heads = 1;
    return 2 * (d + 1) * d + (d + 1) * 2 * heads + d + heads + 2 * d;
}

extern "C" int gode_gat_ode_dopri5_step_forward(const gode_gat_odefunc_t* f, const float* y, float* const* k, float* y1,
                                                const gode_gat_workspace_t* w, double t, double h, float rtol, float atol,
                                                double* sums, void* err_scratch, void* stream)
{
    int rc = check_common(f, w, false); if (rc) return rc;
    if (!y || !k || !y1 || !sums || !err_scratch) return GODE_E_NULLPTR;
    for (int s = 0; s < 7; ++s) if (!k[s]) return GODE_E_NULLPTR;
    const int64_t nd = f->n * f->d;
    for (int s = 1; s < 7; ++s) {
        gode_lincomb_t yin = dp_terms(y, k, DPA[s], s, h, true);
        GODE_TRY(eval_forward(f, w, &yin, (float)(t + DPC[s] * h), k[s], stream));
    }
    gode_lincomb_t sol = dp_terms(y, k, DPB, 7, h, true);
    GODE_TRY(gode_lincomb_f32(y1, &sol, nd, stream));
    gode_lincomb_t err = dp_terms(nullptr, k, DPE, 7, h, false);
    return gode_rk_errnorm_f32(sums, y, y1, &err, rtol, atol, nd, err_scratch, stream);
}

extern "C" int gode_gat_ode_dopri5_step_adjoint(const gode_gat_odefunc_t* f, const float* y, const float* a,
                                                const float* a_t, const float* theta, float* const* ky, float* const* ka,
                                                float* const* kat, float* const* kth, float* y1, float* a1, float* a_t1,
                                                float* theta1, const gode_gat_workspace_t* w, double t, double h,
                                                float rtol, float atol, double* sums /* 4 */, void* err_scratch,
                                                void* stream)
{
    int rc = check_common(f, w, true); if (rc) return rc;
    if (!y || !a || !a_t || !theta || !ky || !ka || !kat || !kth || !y1 || !a1 || !a_t1 || !theta1 || !sums || !err_scratch)
        return GODE_E_NULLPTR;
    for (int s = 0; s < 7; ++s) if (!ky[s] || !ka[s] || !kat[s] || !kth[s]) return GODE_E_NULLPTR;
    const int64_t nd = f->n * f->d, P = gode_gat_ode_theta_len_heads(f->d, n_heads(f));
    for (int s = 1; s < 7; ++s) {
        gode_lincomb_t yin = dp_terms(y, ky, DPA[s], s, h, true);
        gode_lincomb_t ain = dp_terms(a, ka, DPA[s], s, h, true);
        GODE_TRY(eval_adjoint(f, w, yin, ain, (float)(t + DPC[s] * h), ky[s], ka[s], kat[s], kth[s], stream));
    }
    struct Part { const float* y0; float* const* k; float* y1; int64_t len; };
    const Part parts[4] = {{y, ky, y1, nd}, {a, ka, a1, nd}, {a_t, kat, a_t1, 1}, {theta, kth, theta1, P}};
    // the four solution combines in one launch, the four error sums in one pair (same numbers as the single forms)
    float* outs[4]; gode_lincomb_t sols[4], errs[4]; int64_t lens[4]; const float* e0[4]; const float* e1[4];
    for (int c = 0; c < 4; ++c) {
        outs[c] = parts[c].y1; lens[c] = parts[c].len; e0[c] = parts[c].y0; e1[c] = parts[c].y1;
        sols[c] = dp_terms(parts[c].y0, parts[c].k, DPB, 7, h, true);
        errs[c] = dp_terms(nullptr, parts[c].k, DPE, 7, h, false);
    }
    GODE_TRY(gode_lincomb_multi_f32(outs, sols, lens, 4, stream));
    return gode_rk_errnorm_multi_f32(sums, e0, e1, errs, lens, 4, rtol, atol, err_scratch, stream);
}
