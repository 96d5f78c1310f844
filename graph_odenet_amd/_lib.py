"""ctypes binding of libgraphode.so (the C ABI declared in include/graphode.h).

The library is the product: if it is missing, every op raises.  There is no
eager/PyTorch fallback behind these calls.
"""
import ctypes
import os

import torch  # noqa: F401  (imported first so that libamdhip64 is torch's copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgraphode.so")

GODE_MAX_TERMS = 8


class LinComb(ctypes.Structure):
    """Mirror of gode_lincomb_t."""
    _fields_ = [("n", ctypes.c_int32),
                ("coef", ctypes.c_float * GODE_MAX_TERMS),
                ("ptr", ctypes.c_void_p * GODE_MAX_TERMS)]


class SpmmEpilogue(ctypes.Structure):
    """Mirror of gode_spmm_epilogue_t."""
    _fields_ = [("bias", ctypes.c_void_p), ("relu", ctypes.c_int32), ("alpha", ctypes.c_float),
                ("pre", LinComb), ("cot", LinComb), ("Y2", ctypes.c_void_p), ("Y2_colsum", ctypes.c_void_p)]


class Graph(ctypes.Structure):
    """Mirror of gode_graph_t."""
    _fields_ = [("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("val", ctypes.c_void_p),
                ("items", ctypes.c_void_p), ("n_items", ctypes.c_int64),
                ("long_rows", ctypes.c_void_p), ("n_long", ctypes.c_int64), ("partial", ctypes.c_void_p),
                ("n_rows", ctypes.c_int64), ("nnz", ctypes.c_int64)]


class GcnOdeFunc(ctypes.Structure):
    """Mirror of gode_gcn_odefunc_t."""
    _fields_ = [("A", Graph), ("AT", Graph), ("n", ctypes.c_int64), ("d", ctypes.c_int64),
                ("groups", ctypes.c_int32), ("eps", ctypes.c_float),
                ("W", ctypes.c_void_p), ("b", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p)]


class GatProj(ctypes.Structure):
    """Mirror of gode_gat_proj_t."""
    _fields_ = [("ps", ctypes.c_void_p), ("ld_s", ctypes.c_int64), ("pt", ctypes.c_void_p), ("ld_t", ctypes.c_int64),
                ("as_", ctypes.c_void_p), ("at", ctypes.c_void_p), ("ld_a", ctypes.c_int64)]


class Rk4Workspace(ctypes.Structure):
    """Mirror of gode_rk4_workspace_t."""
    _fields_ = [("S", ctypes.c_void_p), ("dZ", ctypes.c_void_p), ("dS", ctypes.c_void_p), ("S2", ctypes.c_void_p),
                ("ky", ctypes.c_void_p * 4), ("ka", ctypes.c_void_p * 4), ("ktheta", ctypes.c_void_p * 4),
                ("wpart", ctypes.c_void_p), ("gpart", ctypes.c_void_p), ("bpart", ctypes.c_void_p),
                ("colsum_scratch", ctypes.c_void_p), ("X", ctypes.c_void_p * 2), ("small_part", ctypes.c_void_p),
                ("y2_colsum", ctypes.c_void_p)]


class ReduceSeg(ctypes.Structure):
    """Mirror of gode_reduce_seg_t."""
    _fields_ = [("out", ctypes.c_void_p), ("part", ctypes.c_void_p), ("n_part", ctypes.c_int64), ("ld", ctypes.c_int64),
                ("col0", ctypes.c_int64), ("col_stride", ctypes.c_int64), ("len", ctypes.c_int64),
                ("w_row0", ctypes.c_void_p), ("time_len", ctypes.c_int64)]


class GatOdeFunc(ctypes.Structure):
    """Mirror of gode_gat_odefunc_t."""
    _fields_ = [("mt", Graph), ("ms_inc", Graph), ("mt_inc", Graph), ("src", ctypes.c_void_p), ("tgt", ctypes.c_void_p),
                ("n_edges", ctypes.c_int64), ("n", ctypes.c_int64), ("d", ctypes.c_int64), ("groups", ctypes.c_int32),
                ("eps_gn", ctypes.c_float), ("eps", ctypes.c_float), ("Wsrc", ctypes.c_void_p), ("Wtgt", ctypes.c_void_p),
                ("Wlog", ctypes.c_void_p), ("bf", ctypes.c_void_p), ("bw", ctypes.c_void_p), ("gamma", ctypes.c_void_p),
                ("beta", ctypes.c_void_p), ("heads", ctypes.c_int32), ("Wpacked", ctypes.c_void_p)]


class GatWorkspace(ctypes.Structure):
    """Mirror of gode_gat_workspace_t."""
    _fields_ = [(k, ctypes.c_void_p) for k in ("X", "Ps", "Pt", "A2", "a", "amax", "wgt", "den", "logits_scratch", "dz", "da",
                                               "dPs", "dPt", "dA2", "pair", "gp", "bp")] + \
               [("wp", ctypes.c_void_p * 3), ("maxpath_scratch", ctypes.c_void_p), ("colsum_scratch", ctypes.c_void_p),
                ("zeros", ctypes.c_void_p), ("heads_scratch", ctypes.c_void_p), ("colsum_scratch2", ctypes.c_void_p),
                ("small_part", ctypes.c_void_p)]


GODE_ADAM_MAX_TENSORS = 64


class AdamArgs(ctypes.Structure):
    """Mirror of gode_adam_args_t."""
    _fields_ = [("param", ctypes.c_void_p * GODE_ADAM_MAX_TENSORS), ("grad", ctypes.c_void_p * GODE_ADAM_MAX_TENSORS),
                ("exp_avg", ctypes.c_void_p * GODE_ADAM_MAX_TENSORS), ("exp_avg_sq", ctypes.c_void_p * GODE_ADAM_MAX_TENSORS),
                ("len", ctypes.c_int64 * GODE_ADAM_MAX_TENSORS)]


c_i64 = ctypes.c_int64
c_p = ctypes.c_void_p
c_f = ctypes.c_float
c_i = ctypes.c_int

# name -> (restype, argtypes); also the list the symbol-export test walks.
SIGNATURES = {
    "gode_abi_version": (c_i, []),
    "gode_error_string": (ctypes.c_char_p, [c_i]),
    "gode_set_option": (c_i, [ctypes.c_char_p, c_i]),
    "gode_get_option": (c_i, [ctypes.c_char_p]),
    "gode_spmm_csr_f32": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64,
                                c_i64, c_i64, ctypes.POINTER(SpmmEpilogue), c_p]),
    "gode_lincomb_f32": (c_i, [c_p, ctypes.POINTER(LinComb), c_i64, c_p]),
    "gode_lincomb_multi_f32": (c_i, [c_p, c_p, c_p, ctypes.c_int32, c_p]),
    "gode_rk_errnorm_multi_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, ctypes.c_int32, c_f, c_f, c_p, c_p]),
    "gode_rk_errnorm_scratch_bytes": (c_i64, []),
    "gode_rk_errnorm_f32": (c_i, [c_p, c_p, c_p, ctypes.POINTER(LinComb), c_f, c_f, c_i64, c_p, c_p]),
    "gode_rk_scaled_sumsq_f32": (c_i, [c_p, ctypes.POINTER(LinComb), c_p, c_f, c_f, c_i64, c_p, c_p]),
    "gode_gn_time_gemm_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p,
                                    c_p, c_i64, c_i, c_f, c_p, c_p]),
    "gode_gn_time_gemm_xout_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p,
                                         c_p, c_i64, c_i, c_f, c_p, c_p, c_p]),
    "gode_gn_time_gemm_pair_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p, c_p, c_p,
                                         c_i, c_f, c_p, c_p, c_p, c_p]),
    "gode_gemm_bwd_parts": (c_i64, [c_i64]),
    "gode_gn_time_gemm_bwd_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p,
                                        c_p, c_i64, c_i, c_p, c_f, ctypes.POINTER(LinComb), c_p, c_p, c_p, c_p]),
    "gode_bwd_wgrad_supported": (c_i, [c_i64, c_i64, c_i64, ctypes.c_int32]),
    "gode_bwd_wgrad_parts": (c_i64, [c_i64]),
    "gode_gn_time_gemm_bwd_wgrad_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p, c_p, c_i64, c_i,
                                              c_p, c_f, ctypes.POINTER(LinComb), c_p, c_p, c_p, c_p, c_p]),
    "gode_group_norm_parts": (c_i64, [c_i64]),
    "gode_group_norm_f32_fwd": (c_i, [c_p, c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p, c_p, c_p]),
    "gode_group_norm_f32_bwd": (c_i, [c_p, c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "gode_rect_gemm_f32": (c_i, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "gode_rect_gemm_nt_f32": (c_i, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "gode_rect_wgrad_parts": (c_i64, [c_i64]),
    "gode_rect_wgrad_f32": (c_i, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_p]),
    "gode_rect_wgrad_sum_f32": (c_i, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "gode_gemm_f32": (c_i, [c_i, c_i, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i, c_p, c_i64, c_p]),
    "gode_gemm_splitk_parts": (c_i64, [c_i64, c_i64, c_i64]),
    "gode_gemm_splitk_f32": (c_i, [c_i, c_i, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p]),
    "gode_cut_pad": (c_i64, [c_i64]),
    "gode_cut_bf16x3_f32": (c_i, [c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "gode_pgemm_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "gode_pgemm_bf16x3": (c_i, [c_i, c_i, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i, c_p, c_i64, c_i, c_p, c_p]),
    "gode_adam_chunk": (c_i64, []),
    "gode_adam_tick_f32": (c_i, [c_p, c_f, c_f, c_p]),
    "gode_adam_f32": (c_i, [ctypes.POINTER(AdamArgs), ctypes.c_int32, c_p, c_i64, c_p, c_f, c_f, c_f, c_f, c_f, c_p]),
    "gode_gat_small_supported": (c_i, [c_i64, c_i64, ctypes.c_int32, c_i64]),
    "gode_gat_small_parts": (c_i64, [c_i64, c_i64]),
    "gode_gat_small_part_len": (c_i64, [c_i64, c_i64]),
    "gode_gat_project_small_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p, c_p, c_p, c_p,
                                         c_i64, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "gode_gat_small_pack_len": (c_i64, [c_i64, c_i64]),
    "gode_gat_small_pack_f32": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p, c_p]),
    "gode_gat_dense_vjp_small_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p, c_p, c_p, c_p,
                                           c_i64, c_p, c_p, c_p, c_f, ctypes.POINTER(LinComb), c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    "gode_gat_maxpath_heads_part_f32": (c_i, [c_p, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "gode_gat_heads_block_cap": (c_i64, []),
    "gode_gat_small_finish_f32": (c_i, [c_p, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p]),
    "gode_gat_small_finish_step_f32": (c_i, [c_p, c_i64, c_i64, c_i64, ctypes.c_int32, c_p, c_p, c_p, c_p, c_p]),
    "gode_gcn_small_supported": (c_i, [c_i64, c_i64, ctypes.c_int32]),
    "gode_gcn_small_parts": (c_i64, [c_i64]),
    "gode_gcn_small_part_len": (c_i64, [c_i64]),
    "gode_gcn_feval_small_f32": (c_i, [ctypes.POINTER(GcnOdeFunc), ctypes.POINTER(LinComb), c_f, c_f, ctypes.POINTER(LinComb),
                                       ctypes.POINTER(LinComb), c_p, c_p, c_p]),
    "gode_gcn_feval_small_next_f32": (c_i, [ctypes.POINTER(GcnOdeFunc), ctypes.POINTER(LinComb), c_f, c_f, ctypes.POINTER(LinComb),
                                            ctypes.POINTER(LinComb), c_p, c_p, ctypes.POINTER(LinComb), c_p, c_p]),
    "gode_gcn_vjp_small_f32": (c_i, [ctypes.POINTER(GcnOdeFunc), ctypes.POINTER(LinComb), c_p, c_f, ctypes.POINTER(LinComb),
                                     c_p, c_p, c_p]),
    "gode_gcn_small_finish_f32": (c_i, [ctypes.POINTER(GcnOdeFunc), c_p, c_p, c_f, c_p]),
    "gode_gcn_small_finish_multi_f32": (c_i, [ctypes.POINTER(GcnOdeFunc), c_p, ctypes.c_int32, ctypes.POINTER(c_p), ctypes.POINTER(c_f), c_p]),
    "gode_gcn_small_finish4_f32": (c_i, [ctypes.POINTER(GcnOdeFunc), c_p, c_p, c_p, c_p, c_p]),
    "gode_wgrad_parts": (c_i64, [c_i64]),
    "gode_wgrad_f32": (c_i, [ctypes.POINTER(LinComb), c_i64, c_i64, ctypes.c_int32, c_f, c_p, c_p,
                             c_p, c_i64, c_i, c_p, c_p]),
    "gode_reduce_parts_f32": (c_i, [c_p, c_p, c_i64, c_i64, c_f, c_i, c_p]),
    "gode_reduce_parts2_f32": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_f, c_i, c_p]),
    "gode_reduce_segments_f32": (c_i, [ctypes.POINTER(ReduceSeg), ctypes.c_int32, c_f, c_p, c_p]),
    "gode_colsum_parts_f32": (c_i, [c_p, c_i64, c_i64, c_p, ctypes.POINTER(c_i64), c_p]),
    "gode_colsum_scratch_bytes": (c_i64, [c_i64, c_i64]),
    "gode_spmm_y2_colsum_rows": (c_i64, [c_i64, c_i64, c_i64]),
    "gode_colsum_f32": (c_i, [c_p, c_p, c_i64, c_i64, c_f, c_i, c_p, c_p]),
    "gode_gat_logits_scratch_bytes": (c_i64, [c_i64]),
    "gode_gat_logits_f32": (c_i, [ctypes.POINTER(GatProj), c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p]),
    "gode_gat_agg_f32_fwd": (c_i, [ctypes.POINTER(Graph), c_p, c_p, ctypes.POINTER(GatProj), c_i64, c_p, c_p, c_p, c_f,
                                   c_p, c_p, c_p, c_p]),
    "gode_gat_agg_f32_bwd": (c_i, [ctypes.POINTER(Graph), c_p, c_p, ctypes.POINTER(GatProj), c_i64, c_p, c_p, c_p, c_p,
                                   c_p, ctypes.POINTER(LinComb), c_f, c_p, c_p, c_p, c_i64, c_p, c_i64,
                                   ctypes.POINTER(ctypes.c_int32), c_p]),
    "gode_gat_maxpath_scratch_bytes": (c_i64, [c_i64]),
    "gode_gat_maxpath_f32": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_p]),
    "gode_gat_heads_scratch_bytes": (c_i64, [c_i64, c_i64]),
    "gode_gat_logits_heads_f32": (c_i, [ctypes.POINTER(GatProj), c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p]),
    "gode_gat_maxpath_heads_f32": (c_i, [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p]),
    "gode_gat_heads_parts": (c_i64, [c_i64]),
    "gode_gat_logits_heads_raw_f32": (c_i, [ctypes.POINTER(GatProj), c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "gode_gat_agg_heads_f32_fwd": (c_i, [ctypes.POINTER(Graph), c_p, c_p, ctypes.POINTER(GatProj), c_i64, c_p, c_p, c_p, c_i64,
                                         c_i64, c_f, c_p, c_p, c_p, c_p]),
    "gode_gat_maxpath_heads_raw_f32": (c_i, [c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p]),
    "gode_gat_scatter_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_i64,
                                   c_p]),
    "gode_time_row_fixup_f32": (c_i, [c_p, c_p, c_i64, c_f, c_p, c_i, c_p]),
    "gode_time_row_fixup3_f32": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_p, c_i64, c_f, c_p, c_p]),
    "gode_edge_matvec_f32_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p]),
    "gode_edge_matvec_msg_f32": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "gode_edge_matvec_f32_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "gode_edge_outer_sum_f32": (c_i, [c_p, c_p, c_p, ctypes.c_int32, ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "gode_segment_attention_f32_fwd": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "gode_segment_attention_f32_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "gode_assign_csr_supported": (c_i, [c_i64, c_i64]),
    "gode_dense_first_nonzero_f32": (c_i, [c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "gode_assign_csr_i32": (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "gode_set2set_supported": (c_i, [c_i64]),
    "gode_set2set_f32_fwd": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "gode_set2set_f32_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p,
                                   c_p, c_p]),
    "gode_gcn_ode_theta_len": (c_i64, [c_i64]),
    "gode_gcn_ode_rk4_forward": (c_i, [ctypes.POINTER(GcnOdeFunc), c_p, ctypes.POINTER(c_p), ctypes.POINTER(Rk4Workspace),
                                       c_f, c_f, ctypes.c_int32, c_p]),
    "gode_gcn_ode_rk4_adjoint": (c_i, [ctypes.POINTER(GcnOdeFunc), c_p, c_p, c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_p),
                                       ctypes.POINTER(Rk4Workspace), c_f, c_f, ctypes.c_int32, c_p]),
    "gode_gcn_ode_dopri5_step_forward": (c_i, [ctypes.POINTER(GcnOdeFunc), c_p, ctypes.POINTER(c_p), c_p,
                                               ctypes.POINTER(Rk4Workspace), ctypes.c_double, ctypes.c_double, c_f, c_f,
                                               c_p, c_p, c_p]),
    "gode_gcn_ode_dopri5_step_adjoint": (c_i, [ctypes.POINTER(GcnOdeFunc), c_p, c_p, c_p, ctypes.POINTER(c_p),
                                               ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_p, c_p, c_p,
                                               ctypes.POINTER(Rk4Workspace), ctypes.c_double, ctypes.c_double, c_f, c_f,
                                               c_p, c_p, c_p]),
    "gode_gat_ode_theta_len": (c_i64, [c_i64]),
    "gode_gat_ode_theta_len_heads": (c_i64, [c_i64, c_i64]),
    "gode_gat_ode_dopri5_step_forward": (c_i, [ctypes.POINTER(GatOdeFunc), c_p, ctypes.POINTER(c_p), c_p,
                                               ctypes.POINTER(GatWorkspace), ctypes.c_double, ctypes.c_double, c_f, c_f,
                                               c_p, c_p, c_p]),
    "gode_gat_ode_dopri5_step_adjoint": (c_i, [ctypes.POINTER(GatOdeFunc), c_p, c_p, c_p, c_p, ctypes.POINTER(c_p),
                                               ctypes.POINTER(c_p), ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_p, c_p, c_p,
                                               c_p, ctypes.POINTER(GatWorkspace), ctypes.c_double, ctypes.c_double, c_f,
                                               c_f, c_p, c_p, c_p]),
    "gode_lstm_cell_supported": (c_i, [c_i64, c_i64, c_i64]),
    "gode_lstm_cell_f32_fwd": (c_i, [c_p] * 7 + [c_i64] * 3 + [c_p] * 4),
    "gode_lstm_cell_f32_bwd": (c_i, [c_p] * 9 + [c_i64] * 3 + [c_p] * 8),
    "gode_gru_wgrad_parts": (c_i64, [c_i64]),
    "gode_gru_cell_f32_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "gode_gru_cell_f32_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                    c_p]),
    "gode_gru_wreduce_f32": (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "gode_prof_create": (c_p, [c_i]),
    "gode_prof_destroy": (None, [c_p]),
    "gode_prof_enable": (None, [c_p]),
    "gode_prof_reset": (None, [c_p]),
    "gode_prof_count": (c_i, [c_p]),
    "gode_prof_read": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i]),
    "gode_prof_kinds": (c_i, [c_p, c_p, c_i]),
}

_lib = None


class GraphodeLibraryError(RuntimeError):
    pass


def load():
    """Load libgraphode.so or raise; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GraphodeLibraryError(
            "libgraphode.so not found at %s - build it with `python __graft_entry__.py` "
            "(or `make -C graph_odenet_amd/csrc`). There is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    if lib.gode_abi_version() != 1:
        raise GraphodeLibraryError("libgraphode.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gode_error_string(rc)
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", rc))


def lincomb(terms):
    """terms: list of (coef, tensor) -> LinComb struct (keeps no references; caller keeps tensors alive)."""
    lc = LinComb()
    terms = [(c, t) for (c, t) in terms if t is not None]
    if len(terms) > GODE_MAX_TERMS:
        raise ValueError("too many lincomb terms")
    lc.n = len(terms)
    for j, (c, t) in enumerate(terms):
        lc.coef[j] = float(c)
        lc.ptr[j] = t.data_ptr()
    return lc


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_handle(device_index=None):
    """hipStream_t of torch's current stream on the (current) device as an int.  torch.cuda.current_stream() builds a
    Stream object (~10 us); this is the raw query underneath it (~0.3 us) - every launch through the ABI asks once, and an
    eager QC step or an adaptive solve makes hundreds of them."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def stream_ptr():
    return ctypes.c_void_p(current_stream_handle())


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None
