// ode_driver.hip — whole fixed-grid integrations of the GCN ODE function as ONE C-ABI call.
//
// The reference drives its ODE function from Python (torchdiffeq, call site GCN/models.py:192).  On
// citation-graph sizes (Cora: 2708 x 16 state) every kernel runs for a few microseconds, so the
// integration is bound by the host's per-launch cost; these entry points issue the complete launch
// sequence of an rk4 (3/8 rule) forward solve or adjoint solve from C with no allocation and no
// synchronisation, which also makes the call capturable into a HIP graph by the caller.
// Launch sequence per stage = graph_odenet_amd/gcn_ode.py (GcnOdeField / GcnOdeAdjointField).
#include <map>
#include <mutex>
#include <utility>
#include "common.h"
#include "options.h"

namespace {

// step size, stage times and h*coefficient products are formed in double and rounded once, exactly as the
// Python driver (solver.py) does, so both drivers feed identical fp32 coefficients to the kernels
const double C38[4] = {0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0};
const double A38[4][3] = {{0.0, 0.0, 0.0}, {1.0 / 3.0, 0.0, 0.0}, {-1.0 / 3.0, 1.0, 0.0}, {1.0, -1.0, 1.0}};
const double B38[4] = {1.0 / 8.0, 3.0 / 8.0, 3.0 / 8.0, 1.0 / 8.0};

// terms of  y + h * sum_{j<s} A38[s][j] * k[j]
gode_lincomb_t stage_terms(const float* y, float* const* k, int s, double h) {
    gode_lincomb_t lc;
    lc.n = 0;
    lc.coef[lc.n] = 1.f; lc.ptr[lc.n] = y; ++lc.n;
    for (int j = 0; j < s; ++j)
        if (A38[s][j] != 0.0) { lc.coef[lc.n] = (float)(h * A38[s][j]); lc.ptr[lc.n] = k[j]; ++lc.n; }
    return lc;
}
// terms of  y + h * sum_{j<3} B38[j] * k[j]   (the last stage is folded into the producing launch)
gode_lincomb_t combine_terms(const float* y, float* const* k, double h) {
    gode_lincomb_t lc;
    lc.n = 0;
    lc.coef[lc.n] = 1.f; lc.ptr[lc.n] = y; ++lc.n;
    for (int j = 0; j < 3; ++j) { lc.coef[lc.n] = (float)(h * B38[j]); lc.ptr[lc.n] = k[j]; ++lc.n; }
    return lc;
}

int spmm(const gode_graph_t& g, const float* X, float* Y, int64_t d, const gode_spmm_epilogue_t* ep, void* s) {
    return gode_spmm_csr_f32(g.rowptr, g.col, g.val, g.items, g.n_items, g.long_rows, g.n_long, g.partial,
                             X, d, Y, d, g.n_rows, d, ep, s);
}

// theta-k layout: [ W ((d+1)*d) | b (d) | gamma (d) | beta (d) | a_t (1) ]
// after the weight-gradient reduction row 0 of W holds colsum(dS):  a_t' = row0 . W[0,:],  row0 *= t
__global__ void theta_fixup_kernel(float* ktheta, const float* W, float t, int d, int64_t off_at) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += blockDim.x) s += ktheta[c] * W[c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ktheta[off_at] = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) ktheta[c] *= t;
}

#define GODE_TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

// Launch-bound graphs: the four reduction launches that close an adjoint stage (weight-gradient partials, time-row
// bookkeeping, bias column sums, GroupNorm affine partials) as ONE launch (rk.hip: reduce_segments_kernel).  Above this
// many rows the separate launches are kept: the weight-gradient reduction has a 16-byte form that matters there.
constexpr int64_t kMergedFinishMaxRows = 1 << 16;

// kt = [W | b | gamma | beta | a_t] from ws->wpart, colsum partials of dZ, ws->gpart / ws->bpart
int stage_finish_merged(const gode_gcn_odefunc_t* f, const gode_rk4_workspace_t* ws, float* kt, float ts, void* stream) {
    const int64_t n = f->n, d = f->d, nW = (d + 1) * d, P = gode_gcn_ode_theta_len(d);
    int64_t cparts = 0;
    GODE_TRY(gode_colsum_parts_f32(ws->dZ, n, d, (float*)ws->colsum_scratch, &cparts, stream));
    const int64_t gp = gode_gemm_bwd_parts(n);
    gode_reduce_seg_t sg[4] = {};
    sg[0] = {kt, ws->wpart, gode_wgrad_parts(n), nW, 0, 1, nW, f->W, d};                       // dW; row 0 is the time row
    sg[1] = {kt + nW, (const float*)ws->colsum_scratch, cparts, d, 0, 1, d, nullptr, 0};           // bias: colsum(dZ)
    sg[2] = {kt + nW + d, ws->gpart, gp, d, 0, 1, d, nullptr, 0};
    sg[3] = {kt + nW + 2 * d, ws->bpart, gp, d, 0, 1, d, nullptr, 0};
    GODE_TRY(gode_reduce_segments_f32(sg, f->groups > 0 ? 4 : 2, ts, kt + (P - 1), stream));
    if (f->groups <= 0) GODE_TRY(gode_zero_f32(kt + nW + d, 2 * d, stream));
    return 0;
}

// Launch-bound graphs: ONE launch per f-eval and one per VJP (csrc/small.hip) instead of 2 + 8
bool fused_small(const gode_gcn_odefunc_t* f) {
    return gode_opt_small_fused() && gode_gcn_small_supported(f->n, f->d, f->groups);
}
gode_lincomb_t negated(gode_lincomb_t lc) { for (int j = 0; j < lc.n; ++j) lc.coef[j] = -lc.coef[j]; return lc; }

}  // namespace

extern "C" int64_t gode_gcn_ode_theta_len(int64_t d) { return (d + 1) * d + 3 * d + 1; }

extern "C" int gode_gcn_ode_rk4_forward(const gode_gcn_odefunc_t* f, float* y, float** result,
                                        const gode_rk4_workspace_t* ws, float t0, float t1, int32_t n_steps,
                                        void* stream)
{
    if (!f || !y || !ws || !result) return GODE_E_NULLPTR;
    if (n_steps <= 0 || f->n <= 0 || f->d <= 0) return GODE_E_SHAPE;
    if (!ws->S || !ws->ky[0] || !ws->ky[1] || !ws->ky[2] || !ws->ky[3]) return GODE_E_NULLPTR;
    const int64_t n = f->n, d = f->d;
    const double h = ((double)t1 - (double)t0) / n_steps;
    float* cur = y;
    float* k[4] = {ws->ky[0], ws->ky[1], ws->ky[2], ws->ky[3]};
    const bool fused = fused_small(f);
    for (int i = 0; i < n_steps; ++i) {
        const double t = (double)t0 + i * h;
        for (int s = 0; s < 4; ++s) {
            gode_lincomb_t xin = stage_terms(cur, k, s, h);
            if (fused) {
                gode_lincomb_t pre = combine_terms(cur, k, h);
                GODE_TRY(gode_gcn_feval_small_f32(f, &xin, (float)(t + C38[s] * h), s == 3 ? (float)(h * B38[3]) : 1.f,
                                                  s == 3 ? &pre : nullptr, nullptr, nullptr, k[s], stream));
                continue;
            }
            GODE_TRY(gode_gn_time_gemm_f32(&xin, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1,
                                           (float)(t + C38[s] * h), ws->S, stream));
            gode_spmm_epilogue_t ep = {};
            ep.bias = f->b; ep.relu = 1; ep.alpha = 1.f;
            if (s == 3) { ep.pre = combine_terms(cur, k, h); ep.alpha = (float)(h * B38[3]); }
            GODE_TRY(spmm(f->A, ws->S, k[s], d, &ep, stream));
        }
        float* tmp = cur; cur = k[3]; k[3] = tmp;      // k[3] holds the new solution
    }
    *result = cur;
    return 0;
}

namespace {

// Side stream + events for the two-chain schedule of the adjoint solve: one set per (device, caller stream), created
// on first use and kept for the life of the process, so that two caller streams (or two threads, each on its own
// stream) never share a side stream or an event.  The map is the only mutable state here and is mutex-guarded.
struct Overlap {
    hipStream_t side = nullptr;
    hipEvent_t sp = nullptr, gf = nullptr, spt = nullptr, wg = nullptr;
    bool ok = false;
};
Overlap* overlap_ctx(hipStream_t caller) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Overlap*> ctxs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(dev, caller);
    auto it = ctxs.find(key);
    if (it != ctxs.end()) return it->second;
    Overlap* c = new Overlap();
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess) {
        bool ok = true;
        for (hipEvent_t* ev : {&c->sp, &c->gf, &c->spt, &c->wg})
            ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
        c->ok = ok;
    }
    ctxs[key] = c;
    return c;
}
#define GODE_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return (int)e__; } while (0)

}  // namespace

// Adjoint solve.  Per stage the launches form two chains:
//   F (forward recompute):  Gf(s) = [t|GN(y_s)]W  ->  Sp(s) = relu(A . + b) (+ masked cotangent dZ)
//   B (vector-Jacobian)  :  SpT(s) = A^T dZ  ->  Gb(s) (k_a)  |  Wg(s) (dW partials)  |  colsum(dZ), reductions
// Gf(s+1) needs only k_y(s).  The SpMM launches (HBM-bound) run alone; the three MFMA-bound launches that follow
// SpT(s) run side by side: Gb(s) on the caller's stream, Wg(s) then Gf(s+1) on a side stream - each of them alone
// keeps the matrix pipe ~50 % busy (S is double-buffered when ws->S2 is given; events order every buffer reuse).
extern "C" int gode_gcn_ode_rk4_adjoint(const gode_gcn_odefunc_t* f, float* y, float* a, float* theta,
                                        float** y_result, float** a_result,
                                        const gode_rk4_workspace_t* ws, float t0, float t1, int32_t n_steps,
                                        void* stream)
{
    if (!f || !y || !a || !theta || !ws || !y_result || !a_result) return GODE_E_NULLPTR;
    if (n_steps <= 0 || f->n <= 0 || f->d <= 0) return GODE_E_SHAPE;
    for (int s = 0; s < 4; ++s) if (!ws->ky[s] || !ws->ka[s] || !ws->ktheta[s]) return GODE_E_NULLPTR;
    if (!ws->S || !ws->dZ || !ws->dS || !ws->wpart || !ws->colsum_scratch) return GODE_E_NULLPTR;
    if (f->groups > 0 && (!ws->gpart || !ws->bpart)) return GODE_E_NULLPTR;
    const int64_t n = f->n, d = f->d;
    const int64_t nW = (d + 1) * d, P = gode_gcn_ode_theta_len(d);
    const double h = ((double)t1 - (double)t0) / n_steps;   // negative: the adjoint runs from t0 (later) to t1 (earlier)
    hipStream_t hs = (hipStream_t)stream;
    // the side stream exists only for callers that ask for the two-chain schedule (never created inside a capture:
    // odeint turns the option off around its HIP-graph captures)
    // launch-bound sizes gain nothing from the second stream and close every stage with one merged reduction launch
    const bool small = f->n <= kMergedFinishMaxRows;
    Overlap* ov = (!small && ws->S2 != nullptr && gode_opt_overlap()) ? overlap_ctx(hs) : nullptr;
    const bool two = ov != nullptr && ov->ok;
    void* side = two ? (void*)ov->side : stream;
    float* Sbuf[2] = {ws->S, two ? ws->S2 : ws->S};
    float* ycur = y; float* acur = a;
    float* ky[4] = {ws->ky[0], ws->ky[1], ws->ky[2], ws->ky[3]};
    float* ka[4] = {ws->ka[0], ws->ka[1], ws->ka[2], ws->ka[3]};
    const int64_t wparts = gode_wgrad_parts(n), gparts = gode_gemm_bwd_parts(n);
    const int total = 4 * n_steps;
    // d = 128 on large graphs: Gb(s) and Wg(s) as ONE pass over x and dS (gemm_pc.hip: gn_gemm_bwd_wgrad_pc_kernel)
    const bool bw = !small && gode_bwd_wgrad_supported(n, d, d, f->groups) && gode_bwd_wgrad_parts(n) <= wparts;

    // Stage inputs with 3 or 4 terms (stages 2 and 3 of the 3/8 rule) are written out by their Gf launch so that
    // Gb and Wg read ONE n x d array instead of the term list (measured at C5: Gb 0.78 -> 0.42 ms, Wg 0.56 -> 0.43 ms
    // for +0.10 ms in Gf).  X[g&1]: Gf(g+2) is ordered after Gb(g) and Wg(g) by the spt event / side-stream order.
    const bool mat = ws->X[0] != nullptr && ws->X[1] != nullptr;
    auto x_out_of = [&](int g) -> float* { return (mat && (g % 4) >= 2) ? ws->X[g & 1] : nullptr; };

    const bool fused = fused_small(f) && ws->small_part != nullptr;
    if (!fused) {
        gode_lincomb_t yin0 = stage_terms(ycur, ky, 0, h);
        GODE_TRY(gode_gn_time_gemm_xout_f32(&yin0, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1,
                                            (float)((double)t0), Sbuf[0], x_out_of(0), stream));
    }
    bool wg_pending = false;
    float stage_t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < total; ++g) {
        const int i = g / 4, s = g % 4;
        const float ts = (float)((double)t0 + i * h + C38[s] * h);
        if (fused) {
            // two launches per stage: f-eval (+ masked cotangent dZ), VJP (+ block partials); one more per step
            const gode_lincomb_t yin = stage_terms(ycur, ky, s, h);
            const gode_lincomb_t cot = negated(stage_terms(acur, ka, s, h));          // cotangent of the VJP is -a
            const gode_lincomb_t ypre = combine_terms(ycur, ky, h), apre = combine_terms(acur, ka, h);
            GODE_TRY(gode_gcn_feval_small_f32(f, &yin, ts, s == 3 ? (float)(h * B38[3]) : 1.f, s == 3 ? &ypre : nullptr,
                                              &cot, ws->dZ, ky[s], stream));
            const int64_t slot = gode_gcn_small_parts(n) * gode_gcn_small_part_len(d);
            GODE_TRY(gode_gcn_vjp_small_f32(f, &yin, ws->dZ, s == 3 ? (float)(h * B38[3]) : 1.f, s == 3 ? &apre : nullptr,
                                            ka[s], ws->small_part + s * slot, stream));
            stage_t[s] = ts;
            if (s == 3) {
                // theta <- theta + h * sum_s b_s ktheta_s straight from the four stages' block partials: one launch per step
                const float wb[4] = {(float)(h * B38[0]), (float)(h * B38[1]), (float)(h * B38[2]), (float)(h * B38[3])};
                GODE_TRY(gode_gcn_small_finish4_f32(f, ws->small_part, theta, wb, stage_t, stream));
                float* tmp = ycur; ycur = ky[3]; ky[3] = tmp;
                tmp = acur; acur = ka[3]; ka[3] = tmp;
            }
            continue;
        }
        gode_lincomb_t yin = stage_terms(ycur, ky, s, h);     // terms of THIS stage (used by Gb / Wg below)
        if (x_out_of(g)) { yin.n = 1; yin.coef[0] = 1.f; yin.ptr[0] = x_out_of(g); }
        gode_lincomb_t ain = stage_terms(acur, ka, s, h);
        gode_spmm_epilogue_t ep = {};
        ep.bias = f->b; ep.relu = 1; ep.alpha = 1.f;
        ep.cot = ain;
        for (int j = 0; j < ep.cot.n; ++j) ep.cot.coef[j] = -ep.cot.coef[j];     // cotangent of the VJP is -a
        ep.Y2 = ws->dZ;
        // colsum(dZ), the bias gradient of the stage: from the per-block column sums Sp(g) leaves in ws->y2_colsum (1/8 of
        // dZ's bytes) when the workspace has them and the graph runs on the kernels that form them
        const int64_t y2rows = (bw && ws->y2_colsum && gode_opt_y2_colsum())
            ? gode_spmm_y2_colsum_rows(f->A.items ? f->A.n_items : f->A.n_rows, f->A.items ? f->A.n_long : 0, d) : 0;
        if (y2rows > 0) ep.Y2_colsum = ws->y2_colsum;
        gode_lincomb_t apre; apre.n = 0;
        if (s == 3) { ep.pre = combine_terms(ycur, ky, h); ep.alpha = (float)(h * B38[3]); apre = combine_terms(acur, ka, h); }
        if (two && g > 0) GODE_HIP(hipStreamWaitEvent(hs, ov->gf, 0));          // S of this stage was produced on the side stream
        GODE_TRY(spmm(f->A, Sbuf[g & 1], ky[s], d, &ep, stream));               // Sp(g): k_y (or new y) and dZ
        // pointers as the NEXT stage will see them (the y-chain swaps buffers after stage 3)
        float* ycur_n = ycur; float* ky_n[4] = {ky[0], ky[1], ky[2], ky[3]};
        if (s == 3) { ycur_n = ky[3]; ky_n[3] = ycur; }
        if (two && wg_pending) GODE_HIP(hipStreamWaitEvent(hs, ov->wg, 0));     // previous Wg still reads dS
        GODE_TRY(spmm(f->AT, ws->dZ, ws->dS, d, nullptr, stream));              // SpT(g): alone on the chip
        if (two) {
            GODE_HIP(hipEventRecord(ov->spt, hs));
            GODE_HIP(hipStreamWaitEvent(ov->side, ov->spt, 0));
        }
        float* kt = ws->ktheta[s];
        if (bw) {
            // Gb(g) + Wg(g) in one launch on the caller's stream, then Gf(g+1) behind it (a 138 KB-LDS block and a
            // forward block do not share a CU anyway); the small reductions of the stage run on the side stream beside
            // them: colsum(dZ) as soon as SpT(g) is done, the partial sums once the dense launch is
            if (two) GODE_HIP(hipStreamWaitEvent(ov->side, ov->spt, 0));
            if (y2rows > 0) GODE_TRY(gode_colsum_f32(kt + nW, ws->y2_colsum, y2rows, d, 1.f, 0, ws->colsum_scratch, side));
            else GODE_TRY(gode_colsum_f32(kt + nW, ws->dZ, n, d, 1.f, 0, ws->colsum_scratch, side));
            GODE_TRY(gode_gn_time_gemm_bwd_wgrad_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1, ws->dS,
                                                     s == 3 ? (float)(h * B38[3]) : 1.f, s == 3 ? &apre : nullptr, ka[s],
                                                     f->groups > 0 ? ws->gpart : nullptr, f->groups > 0 ? ws->bpart : nullptr,
                                                     ws->wpart, stream));
            if (two) {
                GODE_HIP(hipEventRecord(ov->sp, hs));
                GODE_HIP(hipStreamWaitEvent(ov->side, ov->sp, 0));
            }
            GODE_TRY(gode_reduce_parts_f32(kt, ws->wpart, gode_bwd_wgrad_parts(n), nW, 1.f, 0, side));
            hipLaunchKernelGGL(theta_fixup_kernel, dim3(1), dim3(256), 0, (hipStream_t)side, kt, f->W, ts, (int)d, P - 1);
            GODE_LAUNCH_CHECK();
            if (f->groups > 0) {
                GODE_TRY(gode_reduce_parts2_f32(kt + nW + d, ws->gpart, kt + nW + 2 * d, ws->bpart, gparts, d, 1.f, 0, side));
            } else {
                GODE_TRY(gode_zero_f32(kt + nW + d, 2 * d, side));
            }
            if (two) { GODE_HIP(hipEventRecord(ov->wg, ov->side)); wg_pending = true; }
            if (g + 1 < total) {                                                    // Gf(g+1), same stream as Sp(g+1)
                const int i2 = (g + 1) / 4, s2 = (g + 1) % 4;
                gode_lincomb_t yin2 = stage_terms(ycur_n, ky_n, s2, h);
                GODE_TRY(gode_gn_time_gemm_xout_f32(&yin2, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1,
                                                    (float)((double)t0 + i2 * h + C38[s2] * h), Sbuf[(g + 1) & 1],
                                                    x_out_of(g + 1), stream));
                if (two) GODE_HIP(hipEventRecord(ov->gf, hs));      // the wait at the top of the next stage finds it done
            }
            // the side chain still reads dZ and the partial buffers, which Sp(g+1) and the next dense launch overwrite
            if (two) { GODE_HIP(hipStreamWaitEvent(hs, ov->wg, 0)); wg_pending = false; }
        } else {
        // side stream, beside Gb(g) on the main stream: Wg(g), then the dense part of the NEXT stage
        const bool merged = small;                                    // launch-bound: one finishing launch per stage
        GODE_TRY(gode_wgrad_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->beta, ws->dS, d, 1, ws->wpart, side));   // Wg(g)
        if (!merged) {
            GODE_TRY(gode_reduce_parts_f32(kt, ws->wpart, wparts, nW, 1.f, 0, side));
            hipLaunchKernelGGL(theta_fixup_kernel, dim3(1), dim3(256), 0, (hipStream_t)side, kt, f->W, ts, (int)d, P - 1);
            GODE_LAUNCH_CHECK();
        }
        if (two) { GODE_HIP(hipEventRecord(ov->wg, ov->side)); wg_pending = true; }
        if (g + 1 < total) {                                                    // Gf(g+1)
            const int i2 = (g + 1) / 4, s2 = (g + 1) % 4;
            gode_lincomb_t yin2 = stage_terms(ycur_n, ky_n, s2, h);
            GODE_TRY(gode_gn_time_gemm_xout_f32(&yin2, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1,
                                                (float)((double)t0 + i2 * h + C38[s2] * h), Sbuf[(g + 1) & 1],
                                                x_out_of(g + 1), side));
            if (two) GODE_HIP(hipEventRecord(ov->gf, ov->side));
        }
        GODE_TRY(gode_gn_time_gemm_bwd_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->W, d, 1, ws->dS,
                                           s == 3 ? (float)(h * B38[3]) : 1.f, s == 3 ? &apre : nullptr, ka[s],
                                           f->groups > 0 ? ws->gpart : nullptr, f->groups > 0 ? ws->bpart : nullptr, stream));   // Gb(g)
        if (merged) {
            GODE_TRY(stage_finish_merged(f, ws, kt, ts, stream));
        } else {
            GODE_TRY(gode_colsum_f32(kt + nW, ws->dZ, n, d, 1.f, 0, ws->colsum_scratch, stream));
            if (f->groups > 0) {
                GODE_TRY(gode_reduce_parts2_f32(kt + nW + d, ws->gpart, kt + nW + 2 * d, ws->bpart, gparts, d, 1.f, 0, stream));
            } else {
                GODE_TRY(gode_zero_f32(kt + nW + d, 2 * d, stream));
            }
        }
        }
        if (s == 3) {
            // theta <- theta + h * sum b_s ktheta_s   (packed small components, one launch)
            if (two) { GODE_HIP(hipStreamWaitEvent(hs, ov->wg, 0)); wg_pending = false; }
            gode_lincomb_t tc;
            tc.n = 5; tc.coef[0] = 1.f; tc.ptr[0] = theta;
            for (int q = 0; q < 4; ++q) { tc.coef[1 + q] = (float)(h * B38[q]); tc.ptr[1 + q] = ws->ktheta[q]; }
            GODE_TRY(gode_lincomb_f32(theta, &tc, P, stream));
            float* tmp = ycur; ycur = ky[3]; ky[3] = tmp;
            tmp = acur; acur = ka[3]; ka[3] = tmp;
        }
    }
    if (two) {                      // join: nothing of this call is left running on the side stream
        GODE_HIP(hipEventRecord(ov->gf, ov->side));
        GODE_HIP(hipStreamWaitEvent(hs, ov->gf, 0));
    }
    *y_result = ycur;
    *a_result = acur;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// One Dormand-Prince 5(4) step of the same ODE function per C-ABI call (adaptive solves on launch-bound sizes).
// The controller stays with the caller: it passes the FSAL stage in k[0], gets the 5th-order solution in y1 (and a1 /
// theta1), the remaining stages in k[1..6] and the squared error-ratio sums of the tensors torchdiffeq's norm treats
// separately as fp64 device scalars, reads those (its one synchronisation per step) and decides.
// ---------------------------------------------------------------------------------------------------------------
namespace {

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

// terms of  y + h * sum_j coef[j] * k[j]  (zero coefficients dropped; first = number of leading k's considered)
gode_lincomb_t dp_terms(const float* y, float* const* k, const double* coef, int count, double h, bool with_y) {
    gode_lincomb_t lc;
    lc.n = 0;
    if (with_y) { lc.coef[0] = 1.f; lc.ptr[0] = y; lc.n = 1; }
    for (int j = 0; j < count; ++j)
        if (coef[j] != 0.0) { lc.coef[lc.n] = (float)(h * coef[j]); lc.ptr[lc.n] = k[j]; ++lc.n; }
    return lc;
}

// next / x_next (fused launch-bound path only): the stage's launch also writes the next stage's combined input
int dp_eval_forward(const gode_gcn_odefunc_t* f, const gode_rk4_workspace_t* ws, const gode_lincomb_t* yin, float t,
                    float* k_out, const gode_lincomb_t* next, float* x_next, void* stream) {
    if (fused_small(f)) return gode_gcn_feval_small_next_f32(f, yin, t, 1.f, nullptr, nullptr, nullptr, k_out, next, x_next, stream);
    GODE_TRY(gode_gn_time_gemm_f32(yin, f->n, f->d, f->groups, f->eps, f->gamma, f->beta, f->W, f->d, 1, t, ws->S, stream));
    gode_spmm_epilogue_t ep = {};
    ep.bias = f->b; ep.relu = 1; ep.alpha = 1.f;
    return spmm(f->A, ws->S, k_out, f->d, &ep, stream);
}

// One evaluation of the augmented adjoint field: k_y = f(t, y), k_a = -a^T df/dy, k_theta = [-a^T df/dW | .. b | .. gamma |
// .. beta | -a^T df/dt]  (the launch sequence of GcnOdeAdjointField._stage, single stream).
// part_slot >= 0 (fused path): the stage's block partials go to that slot of ws->small_part and the caller closes all
// stages of the step in one launch (gode_gcn_small_finish_multi_f32); -1: the stage is closed here
int dp_eval_adjoint(const gode_gcn_odefunc_t* f, const gode_rk4_workspace_t* ws, gode_lincomb_t yin,
                    const gode_lincomb_t& ain, float t, float* ky, float* ka, float* kth, const gode_lincomb_t* next,
                    float* x_next, int part_slot, void* stream) {
    const int64_t n = f->n, d = f->d, nW = (d + 1) * d, P = gode_gcn_ode_theta_len(d);
    if (fused_small(f) && ws->small_part) {
        const gode_lincomb_t cot = negated(ain);
        GODE_TRY(gode_gcn_feval_small_next_f32(f, &yin, t, 1.f, nullptr, &cot, ws->dZ, ky, next, x_next, stream));
        float* part = ws->small_part + (part_slot > 0 ? part_slot : 0) * gode_gcn_small_parts(n) * gode_gcn_small_part_len(d);
        GODE_TRY(gode_gcn_vjp_small_f32(f, &yin, ws->dZ, 1.f, nullptr, ka, part, stream));
        if (part_slot >= 0) return 0;
        return gode_gcn_small_finish_f32(f, part, kth, t, stream);
    }
    float* xo = (yin.n >= 3 && ws->X[0]) ? ws->X[0] : nullptr;
    GODE_TRY(gode_gn_time_gemm_xout_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1, t, ws->S, xo, stream));
    if (xo) { yin.n = 1; yin.coef[0] = 1.f; yin.ptr[0] = xo; }
    gode_spmm_epilogue_t ep = {};
    ep.bias = f->b; ep.relu = 1; ep.alpha = 1.f;
    ep.cot = ain;
    for (int j = 0; j < ep.cot.n; ++j) ep.cot.coef[j] = -ep.cot.coef[j];
    ep.Y2 = ws->dZ;
    GODE_TRY(spmm(f->A, ws->S, ky, d, &ep, stream));
    GODE_TRY(spmm(f->AT, ws->dZ, ws->dS, d, nullptr, stream));
    GODE_TRY(gode_gn_time_gemm_bwd_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->W, d, 1, ws->dS, 1.f, nullptr, ka,
                                       f->groups > 0 ? ws->gpart : nullptr, f->groups > 0 ? ws->bpart : nullptr, stream));
    GODE_TRY(gode_wgrad_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->beta, ws->dS, d, 1, ws->wpart, stream));
    if (n <= kMergedFinishMaxRows) return stage_finish_merged(f, ws, kth, t, stream);
    GODE_TRY(gode_reduce_parts_f32(kth, ws->wpart, gode_wgrad_parts(n), nW, 1.f, 0, stream));
    hipLaunchKernelGGL(theta_fixup_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kth, f->W, t, (int)d, P - 1);
    GODE_LAUNCH_CHECK();
    GODE_TRY(gode_colsum_f32(kth + nW, ws->dZ, n, d, 1.f, 0, ws->colsum_scratch, stream));
    if (f->groups > 0) {
        const int64_t gparts = gode_gemm_bwd_parts(n);
        GODE_TRY(gode_reduce_parts2_f32(kth + nW + d, ws->gpart, kth + nW + 2 * d, ws->bpart, gparts, d, 1.f, 0, stream));
    } else {
        GODE_TRY(gode_zero_f32(kth + nW + d, 2 * d, stream));
    }
    return 0;
}

}  // namespace

extern "C" int gode_gcn_ode_dopri5_step_forward(const gode_gcn_odefunc_t* f, const float* y, float* const* k, float* y1,
                                                const gode_rk4_workspace_t* ws, double t, double h, float rtol,
                                                float atol, double* sums, void* err_scratch, void* stream)
{
    if (!f || !y || !k || !y1 || !ws || !sums || !err_scratch) return GODE_E_NULLPTR;
    if (f->n <= 0 || f->d <= 0) return GODE_E_SHAPE;
    for (int s = 0; s < 7; ++s) if (!k[s]) return GODE_E_NULLPTR;
    if (!ws->S) return GODE_E_NULLPTR;
    const int64_t nd = f->n * f->d;
    // launch-bound graphs: from the second stage on a stage reads its combined input from X[s & 1], written row by row by
    // the stage before it (same multiply-adds in the same order as combining the terms on the fly: bit-identical)
    const bool chain = fused_small(f) && ws->X[0] && ws->X[1];
    for (int s = 1; s < 7; ++s) {
        gode_lincomb_t yin = dp_terms(y, k, DPA[s], s, h, true);
        if (chain && s >= 2) { yin.n = 1; yin.coef[0] = 1.f; yin.ptr[0] = ws->X[s & 1]; }
        gode_lincomb_t nxt; nxt.n = 0;
        if (chain && s < 6) nxt = dp_terms(y, k, DPA[s + 1], s + 1, h, true);
        GODE_TRY(dp_eval_forward(f, ws, &yin, (float)(t + DPC[s] * h), k[s], nxt.n > 0 ? &nxt : nullptr,
                                 nxt.n > 0 ? ws->X[(s + 1) & 1] : nullptr, stream));
    }
    gode_lincomb_t sol = dp_terms(y, k, DPB, 7, h, true);
    GODE_TRY(gode_lincomb_f32(y1, &sol, nd, stream));
    gode_lincomb_t err = dp_terms(nullptr, k, DPE, 7, h, false);
    return gode_rk_errnorm_f32(sums, y, y1, &err, rtol, atol, nd, err_scratch, stream);
}

extern "C" int gode_gcn_ode_dopri5_step_adjoint(const gode_gcn_odefunc_t* f, const float* y, const float* a,
                                                const float* theta, float* const* ky, float* const* ka,
                                                float* const* kth, float* y1, float* a1, float* theta1,
                                                const gode_rk4_workspace_t* ws, double t, double h, float rtol,
                                                float atol, double* sums /* 4 */, void* err_scratch, void* stream)
{
    if (!f || !y || !a || !theta || !ky || !ka || !kth || !y1 || !a1 || !theta1 || !ws || !sums || !err_scratch)
        return GODE_E_NULLPTR;
    if (f->n <= 0 || f->d <= 0) return GODE_E_SHAPE;
    for (int s = 0; s < 7; ++s) if (!ky[s] || !ka[s] || !kth[s]) return GODE_E_NULLPTR;
    if (!ws->S || !ws->dZ || !ws->dS || !ws->wpart || !ws->colsum_scratch) return GODE_E_NULLPTR;
    if (f->groups > 0 && (!ws->gpart || !ws->bpart)) return GODE_E_NULLPTR;
    const int64_t nd = f->n * f->d, P = gode_gcn_ode_theta_len(f->d);
    const bool fused = fused_small(f) && ws->small_part;
    const bool chain = fused && ws->X[0] && ws->X[1];                                   // as in the forward step
    for (int s = 1; s < 7; ++s) {
        gode_lincomb_t yin = dp_terms(y, ky, DPA[s], s, h, true);
        if (chain && s >= 2) { yin.n = 1; yin.coef[0] = 1.f; yin.ptr[0] = ws->X[s & 1]; }
        gode_lincomb_t ain = dp_terms(a, ka, DPA[s], s, h, true);
        gode_lincomb_t nxt; nxt.n = 0;
        if (chain && s < 6) nxt = dp_terms(y, ky, DPA[s + 1], s + 1, h, true);
        GODE_TRY(dp_eval_adjoint(f, ws, yin, ain, (float)(t + DPC[s] * h), ky[s], ka[s], kth[s], nxt.n > 0 ? &nxt : nullptr,
                                 nxt.n > 0 ? ws->X[(s + 1) & 1] : nullptr, fused ? s - 1 : -1, stream));
    }
    if (fused) {      // the six stages' small components in one launch: nothing inside the step reads them
        float* kout[6]; float tsv[6];
        for (int s = 1; s < 7; ++s) { kout[s - 1] = kth[s]; tsv[s - 1] = (float)(t + DPC[s] * h); }
        GODE_TRY(gode_gcn_small_finish_multi_f32(f, ws->small_part, 6, kout, tsv, stream));
    }
    gode_lincomb_t sy = dp_terms(y, ky, DPB, 7, h, true), sa = dp_terms(a, ka, DPB, 7, h, true),
                   st = dp_terms(theta, kth, DPB, 7, h, true);
    {   // the three solution combines in one launch, the four error sums in one pair (same numbers as the single forms)
        float* outs[3] = {y1, a1, theta1};
        const gode_lincomb_t sols[3] = {sy, sa, st};
        const int64_t lens[3] = {nd, nd, P};
        GODE_TRY(gode_lincomb_multi_f32(outs, sols, lens, 3, stream));
    }
    gode_lincomb_t ey = dp_terms(nullptr, ky, DPE, 7, h, false), ea = dp_terms(nullptr, ka, DPE, 7, h, false),
                   et = dp_terms(nullptr, kth, DPE, 7, h, false);
    // a_t is the last entry of the packed vector, the flattened parameters the P-1 before it
    gode_lincomb_t et_at = et;
    for (int j = 0; j < et_at.n; ++j) et_at.ptr[j] = et.ptr[j] + (P - 1);
    const float* e0[4] = {y, a, theta + (P - 1), theta};
    const float* e1[4] = {y1, a1, theta1 + (P - 1), theta1};
    const gode_lincomb_t errs[4] = {ey, ea, et_at, et};
    const int64_t ens[4] = {nd, nd, 1, P - 1};
    return gode_rk_errnorm_multi_f32(sums, e0, e1, errs, ens, 4, rtol, atol, err_scratch, stream);
}
