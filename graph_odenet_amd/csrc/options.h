#pragma once
// run-time tuning switches (api.hip); initial values from the environment, changeable through gode_set_option
int gode_opt_gemm_split();   // GODE_GEMM_SPLIT (default 2): split-bf16 forward dense product at d = 128 - 1 always, 2 launches of <= 2 terms and >= 65 536 rows, 0 never
int gode_opt_overlap();      // GODE_OVERLAP (default 1): two-stream schedule of the adjoint rk4 driver
int gode_opt_wgrad_split();  // GODE_WGRAD_SPLIT (default 8): weight gradient at d = 128 from exact bf16 pieces with 8 (or 6) piece products per product; 0 = fp32-MFMA kernel
int gode_opt_wgrad_split_small();  // wgrad_split_small (default 0): use it below 65 536 rows too (tests)
int gode_opt_bwd_pc();       // GODE_BWD_PC (default 1): VJP at d = 128, >= 65 536 rows in producer / consumer form on the bf16 matrix cores (gemm_pc.hip)
int gode_opt_fwd_pc();       // GODE_FWD_PC (default 3): forward product likewise; bit 0 = launches of <= 2 terms, bit 1 = launches of >= 3 terms or with x_out
int gode_opt_small_fused();   // GODE_SMALL_FUSED (default 1): launch-bound graphs take the fused one-launch f-eval / VJP of csrc/small.hip
int gode_opt_bwd_wgrad();     // GODE_BWD_WGRAD (default 1): VJP and weight gradient of the ODE function at d = 128, >= 65 536 rows in ONE pass (gemm_pc.hip: gn_gemm_bwd_wgrad_pc_kernel) where the drivers issue both
int gode_opt_y2_colsum();     // GODE_Y2_COLSUM (default 1): the adjoint rk4 driver reduces a stage's bias gradient from the per-block column sums its forward-recompute SpMM leaves (workspace y2_colsum) instead of from dZ
