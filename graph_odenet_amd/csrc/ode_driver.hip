// ode_driver.hip — whole fixed-grid integrations of the GCN ODE function as ONE C-ABI call.
//
// The reference drives its ODE function from Python (torchdiffeq, call site GCN/models.py:192).  On
// citation-graph sizes (Cora: 2708 x 16 state) every kernel runs for a few microseconds, so the
// integration is bound by the host's per-launch cost; these entry points issue the complete launch
// sequence of an rk4 (3/8 rule) forward solve or adjoint solve from C with no allocation and no
// synchronisation, which also makes the call capturable into a HIP graph by the caller.
// Launch sequence per stage = graph_odenet_amd/gcn_ode.py (GcnOdeField / GcnOdeAdjointField).
#include "common.h"

namespace {

const float C38[4] = {0.f, 1.f / 3.f, 2.f / 3.f, 1.f};
const float A38[4][3] = {{0.f, 0.f, 0.f}, {1.f / 3.f, 0.f, 0.f}, {-1.f / 3.f, 1.f, 0.f}, {1.f, -1.f, 1.f}};
const float B38[4] = {1.f / 8.f, 3.f / 8.f, 3.f / 8.f, 1.f / 8.f};

// terms of  y + h * sum_{j<s} A38[s][j] * k[j]
gode_lincomb_t stage_terms(const float* y, float* const* k, int s, float h) {
    gode_lincomb_t lc;
    lc.n = 0;
    lc.coef[lc.n] = 1.f; lc.ptr[lc.n] = y; ++lc.n;
    for (int j = 0; j < s; ++j)
        if (A38[s][j] != 0.f) { lc.coef[lc.n] = h * A38[s][j]; lc.ptr[lc.n] = k[j]; ++lc.n; }
    return lc;
}
// terms of  y + h * sum_{j<3} B38[j] * k[j]   (the last stage is folded into the producing launch)
gode_lincomb_t combine_terms(const float* y, float* const* k, float h) {
    gode_lincomb_t lc;
    lc.n = 0;
    lc.coef[lc.n] = 1.f; lc.ptr[lc.n] = y; ++lc.n;
    for (int j = 0; j < 3; ++j) { lc.coef[lc.n] = h * B38[j]; lc.ptr[lc.n] = k[j]; ++lc.n; }
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
    const float h = (t1 - t0) / n_steps;
    float* cur = y;
    float* k[4] = {ws->ky[0], ws->ky[1], ws->ky[2], ws->ky[3]};
    for (int i = 0; i < n_steps; ++i) {
        const float t = t0 + i * h;
        for (int s = 0; s < 4; ++s) {
            gode_lincomb_t xin = stage_terms(cur, k, s, h);
            GODE_TRY(gode_gn_time_gemm_f32(&xin, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1,
                                           t + C38[s] * h, ws->S, stream));
            gode_spmm_epilogue_t ep = {};
            ep.bias = f->b; ep.relu = 1; ep.alpha = 1.f;
            if (s == 3) { ep.pre = combine_terms(cur, k, h); ep.alpha = h * B38[3]; }
            GODE_TRY(spmm(f->A, ws->S, k[s], d, &ep, stream));
        }
        float* tmp = cur; cur = k[3]; k[3] = tmp;      // k[3] holds the new solution
    }
    *result = cur;
    return 0;
}

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
    const float h = (t1 - t0) / n_steps;             // negative: the adjoint runs from t0 (later) to t1 (earlier)
    hipStream_t hs = (hipStream_t)stream;
    float* ycur = y; float* acur = a;
    float* ky[4] = {ws->ky[0], ws->ky[1], ws->ky[2], ws->ky[3]};
    float* ka[4] = {ws->ka[0], ws->ka[1], ws->ka[2], ws->ka[3]};
    const int64_t wparts = gode_wgrad_parts(n), gparts = gode_gemm_bwd_parts(n);
    for (int i = 0; i < n_steps; ++i) {
        const float t = t0 + i * h;
        for (int s = 0; s < 4; ++s) {
            const float ts = t + C38[s] * h;
            gode_lincomb_t yin = stage_terms(ycur, ky, s, h);
            gode_lincomb_t ain = stage_terms(acur, ka, s, h);
            GODE_TRY(gode_gn_time_gemm_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->beta, f->W, d, 1, ts, ws->S, stream));
            gode_spmm_epilogue_t ep = {};
            ep.bias = f->b; ep.relu = 1; ep.alpha = 1.f;
            ep.cot = ain;
            for (int j = 0; j < ep.cot.n; ++j) ep.cot.coef[j] = -ep.cot.coef[j];     // cotangent of the VJP is -a
            ep.Y2 = ws->dZ;
            gode_lincomb_t apre; apre.n = 0;
            if (s == 3) { ep.pre = combine_terms(ycur, ky, h); ep.alpha = h * B38[3]; apre = combine_terms(acur, ka, h); }
            GODE_TRY(spmm(f->A, ws->S, ky[s], d, &ep, stream));
            GODE_TRY(spmm(f->AT, ws->dZ, ws->dS, d, nullptr, stream));
            GODE_TRY(gode_gn_time_gemm_bwd_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->W, d, 1, ws->dS,
                                               s == 3 ? h * B38[3] : 1.f, s == 3 ? &apre : nullptr, ka[s],
                                               f->groups > 0 ? ws->gpart : nullptr, f->groups > 0 ? ws->bpart : nullptr, stream));
            GODE_TRY(gode_wgrad_f32(&yin, n, d, f->groups, f->eps, f->gamma, f->beta, ws->dS, d, 1, ws->wpart, stream));
            float* kt = ws->ktheta[s];
            GODE_TRY(gode_reduce_parts_f32(kt, ws->wpart, wparts, nW, 1.f, 0, stream));
            hipLaunchKernelGGL(theta_fixup_kernel, dim3(1), dim3(256), 0, hs, kt, f->W, ts, (int)d, P - 1);
            GODE_LAUNCH_CHECK();
            GODE_TRY(gode_colsum_f32(kt + nW, ws->dZ, n, d, 1.f, 0, ws->colsum_scratch, stream));
            if (f->groups > 0) {
                GODE_TRY(gode_reduce_parts_f32(kt + nW + d, ws->gpart, gparts, d, 1.f, 0, stream));
                GODE_TRY(gode_reduce_parts_f32(kt + nW + 2 * d, ws->bpart, gparts, d, 1.f, 0, stream));
            } else {
                hipError_t e = hipMemsetAsync(kt + nW + d, 0, (size_t)2 * d * sizeof(float), hs);
                if (e != hipSuccess) return (int)e;
            }
        }
        // theta <- theta + h * sum b_s ktheta_s   (packed small components, one launch)
        gode_lincomb_t tc;
        tc.n = 5; tc.coef[0] = 1.f; tc.ptr[0] = theta;
        for (int s = 0; s < 4; ++s) { tc.coef[1 + s] = h * B38[s]; tc.ptr[1 + s] = ws->ktheta[s]; }
        GODE_TRY(gode_lincomb_f32(theta, &tc, P, stream));
        float* tmp = ycur; ycur = ky[3]; ky[3] = tmp;
        tmp = acur; acur = ka[3]; ka[3] = tmp;
    }
    *y_result = ycur;
    *a_result = acur;
    return 0;
}
