// dense_pc.h - launchers of the producer / consumer dense kernels (gemm_pc.hip), called from the dispatch in gemm.hip.
// d = 128 only; cg = channels per GroupNorm group (forward: 0, 1, 2, 4; VJP: 0, 4); operands 16-byte aligned (checked
// by the caller).  Return 0, a hipError_t, or GODE_E_UNSUPPORTED when no kernel is instantiated for the arguments.
#pragma once
#include "common.h"

int gode_pc_fwd_launch(const LinComb& lc, int64_t n_rows, float eps, const float* gamma, const float* beta,
                       const float* W, int has_time, float t, float* S, float* xout, int cg, hipStream_t s);
int gode_pc_bwd_launch(const LinComb& lc, int64_t n_rows, float eps, const float* gamma, const float* W, int has_time,
                       const float* dS, float out_scale, const LinComb& pre, float* dx, float* dgamma_part,
                       float* dbeta_part, int64_t n_part, int cg, hipStream_t s);
// VJP and weight gradient in one pass (round 4): dW_part holds gode_pc_bwd_wgrad_parts(n_rows) partials of (128 + has_time) x 128
int64_t gode_pc_bwd_wgrad_parts(int64_t n_rows);
int gode_pc_bwd_wgrad_launch(const LinComb& lc, int64_t n_rows, float eps, const float* gamma, const float* beta, const float* W,
                             int has_time, const float* dS, float out_scale, const LinComb& pre, float* dx,
                             float* dgamma_part, float* dbeta_part, int64_t n_part, float* dW_part, int cg, hipStream_t s);
