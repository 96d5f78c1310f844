#pragma once
// run-time tuning switches (api.hip); initial values from the environment, changeable through gode_set_option
int gode_opt_gemm_split();   // GODE_GEMM_SPLIT (default 0): split-bf16 forward dense product at d = 128
int gode_opt_overlap();      // GODE_OVERLAP (default 1): two-stream schedule of the adjoint rk4 driver
