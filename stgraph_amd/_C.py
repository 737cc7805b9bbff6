"""ctypes binding of libstgraph_hip.so (include/stgraph_hip.h).

This is the ONLY way the Python host side reaches the HIP kernels.  There is no
CPU or eager-PyTorch fallback: if the library has not been built, importing this
module raises, and every wrapper raises ``RuntimeError`` with the library's own
message when an entry point reports an error.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STGRAPH_AMD_LIB: a diagnosis build of the same library (tools/diag/build_*_trace.sh); the product never sets it
LIB_PATH = os.environ.get("STGRAPH_AMD_LIB") or os.path.join(_HERE, "lib", "libstgraph_hip.so")

ABI_VERSION = 27

STG_ERR_INVALID_ARGUMENT = 10001
STG_ERR_UNSUPPORTED = 10002
STG_ERR_VERTEX_RANGE = 10003
STG_ERR_WORKSPACE = 10004
STG_ERR_JIT = 10005

# every symbol include/stgraph_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = (
    "stg_abi_version", "stg_last_error_string", "stg_set_tuning",
    "stg_csr_ctor_host", "stg_graph_build_host",
    "stg_graph_build_device_workspace_bytes", "stg_graph_build_device",
    "stg_graph_build_direct_workspace_bytes", "stg_graph_build_direct_device", "stg_graph_build_direct2_device",
    "stg_graph_build_direct2_batch_device",
    "stg_rows_by_degree_workspace_bytes", "stg_rows_by_degree_device",
    "stg_edgeset_update_workspace_bytes", "stg_edgeset_update_device", "stg_edgeset_update_host", "stg_edgeset_merge_device", "stg_edgeset_step_device",
    "stg_edgeset_step_deferred_device", "stg_edgeset_emit_pending_device",
    "stg_edgeset_emit_csr_workspace_bytes", "stg_edgeset_emit_csr_device", "stg_edgeset_emit_csr_host",
    "stg_jit_compile", "stg_jit_free", "stg_jit_load", "stg_jit_get_function", "stg_jit_unload", "stg_jit_launch",
    "stg_gcn_agg", "stg_gcn_agg_edge", "stg_gcn_layer_fwd", "stg_gcn_agg_edge2", "stg_bias_act_fwd", "stg_bias_act_bwd_workspace_bytes", "stg_bias_act_bwd",
    "stg_gcn_agg_transform", "stg_edge_gather_f32", "stg_gat_score_flag", "stg_gat_fwd_k0", "stg_gat_fwd_k1", "stg_gat_bwd", "stg_gat_bwd_factored", "stg_gat_bwd_er",
    "stg_gat_fwd_k1_uniform", "stg_gat_fc_out", "stg_gat_fwd_k1_scored", "stg_gat_bwd_factored_elu",
    "stg_gat_fc_supported", "stg_gat_fc_fwd", "stg_gat_proj_supported", "stg_gat_proj_fwd", "stg_gat_proj_bwd_workspace_bytes", "stg_gat_proj_bwd",
    "stg_gemm_tn_workspace_bytes", "stg_gemm_tn_f32", "stg_gemm_tn_colsum_f32", "stg_gemm_tn_relu_mask_f32",
    "stg_gat_fc_feat_if", "stg_gat_bwd_uniform_supported", "stg_gat_bwd_prepass", "stg_gat_bwd_prepass_heads_supported", "stg_gat_bwd_prepass_heads", "stg_gat_attn_fold", "stg_gat_bwd_uniform_edges", "stg_gat_bwd_uniform_gx_fallback", "stg_gemm_tn_gated_f32", "stg_rowgemm_heads_supported", "stg_rowgemm_heads_f32", "stg_rowgemm_bits_words", "stg_rowgemm_bits_supported", "stg_rowgemm_act_bits_f32", "stg_gemm_tn_multi_workspace_bytes", "stg_gemm_tn_multi_f32", "stg_gemm_tn_form_workspace_bytes", "stg_gemm_tn_form_f32", "stg_gemm_tn_form_partial_f32", "stg_gemm_tn_reduce_multi_f32", "stg_gemm_tn_reduce_multi_blocks_f32", "stg_tgcn_pack_weights", "stg_tgcn_unfold_gate_grads", "stg_tgcn_fold_weights", "stg_link_decode_fwd_multi", "stg_rowgemm_supported", "stg_rowgemm_f32", "stg_rowgemm_strided_f32", "stg_rowgemm_act_supported", "stg_rowgemm_act_f32",
    "stg_tgcn_cell_fused_supported", "stg_tgcn_cell_fused_fwd", "stg_tgcn_cell_fused_bwd",
    "stg_tgcn_cell_fused_bwd_dx_supported", "stg_tgcn_cell_fused_bwd_dx",
    "stg_tgcn_head_supported", "stg_tgcn_head_workspace_bytes", "stg_tgcn_head_fwd", "stg_tgcn_head_fwd_acc", "stg_tgcn_head_bwd",
    "stg_xent_workspace_bytes", "stg_xent_fwd", "stg_xent_bwd", "stg_xent_bwd_colsum_workspace_bytes", "stg_xent_bwd_colsum", "stg_xent_fwd_grad_workspace_bytes", "stg_xent_fwd_grad", "stg_xent_scale_grad",
    "stg_xent_small_supported", "stg_xent_small_fwd", "stg_xent_small_bwd", "stg_mm_bwd_small_supported", "stg_mm_bwd_small", "stg_gemm_tn_small_supported", "stg_gemm_tn_small_f32",
    "stg_link_head_supported", "stg_link_head_workspace_bytes", "stg_link_head_fwd", "stg_link_head_bwd",
    "stg_tgcn_cell_prep_fwd", "stg_tgcn_cell_gates_fwd", "stg_tgcn_cell_update_fwd",
    "stg_tgcn_cell_update_bwd", "stg_tgcn_cell_gates_bwd", "stg_tgcn_cell_prep_bwd",
    "stg_tgcn_step_supported", "stg_tgcn_step_loss_partials", "stg_tgcn_step_fwd", "stg_tgcn_step_bwd",
    "stg_tgcn_window_loss", "stg_partial_sums_loss", "stg_link_decode_fwd", "stg_link_decode_bwd", "stg_degree_norm_f32",
)


def _ptr_fields(names):
    return [(n, ctypes.c_void_p) for n in names.split()]


BUILD_BATCH_MAX = 16
GEMM_REDUCE_BLOCKS = 4


class StoreEmission(ctypes.Structure):
    """stg_store_emission (include/stgraph_hip.h)."""
    _fields_ = [("keys_fwd", ctypes.c_void_p), ("keys_bwd", ctypes.c_void_p), ("E", ctypes.c_int64),
                ("fwd_row_offset", ctypes.c_void_p), ("bwd_row_offset", ctypes.c_void_p), ("fwd_column_indices", ctypes.c_void_p),
                ("bwd_column_indices", ctypes.c_void_p), ("norm", ctypes.c_void_p), ("norm_col_fwd", ctypes.c_void_p),
                ("norm_col_bwd", ctypes.c_void_p), ("flags", ctypes.c_int)]


class BuildJob(ctypes.Structure):
    """stg_build_job (include/stgraph_hip.h)."""
    _fields_ = ([("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("E", ctypes.c_int64)] +
                [(k, ctypes.c_void_p) for k in ("perm_fwd", "fwd_row_offset", "fwd_column_indices", "fwd_eids", "bwd_row_offset",
                                                "bwd_column_indices", "bwd_eids", "in_degrees", "out_degrees", "norm",
                                                "norm_col_fwd", "norm_col_bwd", "zero_counters", "workspace")] +
                [("workspace_bytes", ctypes.c_size_t), ("id", ctypes.c_int32)])


class TgcnStepFwdArgs(ctypes.Structure):
    """stg_tgcn_step_fwd_args (include/stgraph_hip.h), field for field."""
    _fields_ = (_ptr_fields("row_offsets column_indices node_ids norm_col_edge ew_edge norm x a3 H target "
                            "WcatT b3 Wz bz Wr br Wh bh W1 b1 W2 b2 P x3 Z R Ht Hn HR y y_out loss_partial clamp_mask") +
                [("N", ctypes.c_int64), ("C", ctypes.c_int32), ("Fin", ctypes.c_int32), ("Fh", ctypes.c_int32),
                 ("head", ctypes.c_int32), ("lo", ctypes.c_float), ("hi", ctypes.c_float), ("w_image", ctypes.c_void_p),
                 ("w_fold", ctypes.c_void_p), ("b_fold", ctypes.c_void_p), ("fold_status", ctypes.c_void_p),
                 ("fold_bound", ctypes.c_void_p)])


class TgcnStepBwdArgs(ctypes.Structure):
    """stg_tgcn_step_bwd_args (include/stgraph_hip.h), field for field."""
    _fields_ = (_ptr_fields("row_offsets column_indices node_ids norm_col_edge ew_edge norm zn g_y dHn g_cost "
                            "Z R Ht H Hn x3 y_out target WzT WrT WhT Wcat W1T W2 dzl drl dhl da3 dH z dyt dyo clamp_mask") +
                [("N", ctypes.c_int64), ("C", ctypes.c_int32), ("Fin", ctypes.c_int32), ("Fh", ctypes.c_int32),
                 ("head", ctypes.c_int32), ("lo", ctypes.c_float), ("hi", ctypes.c_float)] +
                _ptr_fields("link_row_ptr link_other link_eid link_y link_logits link_target") + [("link_inv_m", ctypes.c_float),
                                                                                                    ("w_image", ctypes.c_void_p),
                                                                                                    ("w_fold_t", ctypes.c_void_p),
                                                                                                    ("ld_d", ctypes.c_int32)])


class StgError(RuntimeError):
    """An entry point of libstgraph_hip.so returned a non-zero code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libstgraph_hip error {code}: {message}")
        self.code = code


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing. stgraph_amd has no CPU fallback: build the HIP library first "
            "(`make -C stgraph_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`).")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    lib.stg_abi_version.restype = ctypes.c_int
    lib.stg_abi_version.argtypes = []
    lib.stg_last_error_string.restype = ctypes.c_char_p
    lib.stg_last_error_string.argtypes = []
    lib.stg_set_tuning.restype = ctypes.c_int
    lib.stg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.stg_csr_ctor_host.restype = ctypes.c_int
    lib.stg_csr_ctor_host.argtypes = [vp, vp, vp, vp, i64, i32, ctypes.c_int] + [vp] * 7
    lib.stg_graph_build_host.restype = ctypes.c_int
    lib.stg_graph_build_host.argtypes = [vp, vp, i64, i32] + [vp] * 11
    lib.stg_graph_build_device_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_graph_build_device_workspace_bytes.argtypes = [i64, i32]
    lib.stg_graph_build_device.restype = ctypes.c_int
    lib.stg_graph_build_device.argtypes = [vp, vp, i64, i32] + [vp] * 11 + [vp, vp, ctypes.c_size_t, vp]
    lib.stg_graph_build_direct_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_graph_build_direct_workspace_bytes.argtypes = [i64, i32]
    lib.stg_graph_build_direct_device.restype = ctypes.c_int
    lib.stg_graph_build_direct_device.argtypes = [vp, vp, i64, i32] + [vp] * 11 + [vp, vp, ctypes.c_size_t, vp]
    lib.stg_graph_build_direct2_device.restype = ctypes.c_int
    lib.stg_graph_build_direct2_device.argtypes = [vp, vp, i64, i32] + [vp] * 16 + [vp, ctypes.c_size_t, vp]
    lib.stg_graph_build_direct2_batch_device.restype = ctypes.c_int
    lib.stg_graph_build_direct2_batch_device.argtypes = [ctypes.POINTER(BuildJob), i32, i32, vp, vp]
    lib.stg_rows_by_degree_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_rows_by_degree_workspace_bytes.argtypes = [i32]
    lib.stg_rows_by_degree_device.restype = ctypes.c_int
    lib.stg_rows_by_degree_device.argtypes = [vp, i32, vp, vp, ctypes.c_size_t, vp]
    lib.stg_edgeset_update_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_edgeset_update_workspace_bytes.argtypes = [i64, i64]
    lib.stg_edgeset_update_device.restype = ctypes.c_int
    lib.stg_edgeset_update_device.argtypes = [vp, vp, i64, vp, vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.stg_edgeset_update_host.restype = ctypes.c_int
    lib.stg_edgeset_update_host.argtypes = [vp, vp, i64, vp, vp, i64, vp, vp, i64, i32, vp, vp, vp]
    lib.stg_edgeset_merge_device.restype = ctypes.c_int
    lib.stg_edgeset_merge_device.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, vp]
    lib.stg_edgeset_step_device.restype = ctypes.c_int
    lib.stg_edgeset_step_device.argtypes = [vp, vp, i64, vp, vp, i64, vp, vp, i64, i32, ctypes.c_int] + [vp] * 14
    lib.stg_edgeset_step_deferred_device.restype = ctypes.c_int
    lib.stg_edgeset_step_deferred_device.argtypes = ([vp, vp, i64, vp, vp, i64, vp, vp, i64, i32, ctypes.c_int] + [vp] * 12 +
                                                     [ctypes.POINTER(StoreEmission), ctypes.POINTER(StoreEmission), vp, vp])
    lib.stg_edgeset_emit_pending_device.restype = ctypes.c_int
    lib.stg_edgeset_emit_pending_device.argtypes = [ctypes.POINTER(StoreEmission), vp]
    lib.stg_edgeset_emit_csr_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_edgeset_emit_csr_workspace_bytes.argtypes = [i32]
    lib.stg_edgeset_emit_csr_device.restype = ctypes.c_int
    lib.stg_edgeset_emit_csr_device.argtypes = [vp, vp, i64, i32, ctypes.c_int] + [vp] * 6 + [vp, ctypes.c_size_t, vp]
    lib.stg_edgeset_emit_csr_host.restype = ctypes.c_int
    lib.stg_edgeset_emit_csr_host.argtypes = [vp, vp, i64, i32, ctypes.c_int] + [vp] * 6
    lib.stg_jit_compile.restype = ctypes.c_int
    lib.stg_jit_compile.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t),
                                    ctypes.POINTER(vp)]
    lib.stg_jit_free.restype = None
    lib.stg_jit_free.argtypes = [vp]
    lib.stg_jit_load.restype = ctypes.c_int
    lib.stg_jit_load.argtypes = [vp, ctypes.POINTER(vp)]
    lib.stg_jit_get_function.restype = ctypes.c_int
    lib.stg_jit_get_function.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp)]
    lib.stg_jit_unload.restype = ctypes.c_int
    lib.stg_jit_unload.argtypes = [vp]
    lib.stg_jit_launch.restype = ctypes.c_int
    lib.stg_jit_launch.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, i32, vp, i32, vp]
    lib.stg_gcn_agg.restype = ctypes.c_int
    lib.stg_gcn_agg.argtypes = [vp] * 9 + [i32, i32, i32, vp]
    lib.stg_gcn_agg_edge2.restype = ctypes.c_int
    lib.stg_gcn_agg_edge2.argtypes = [vp] * 5 + [i32] + [vp] * 5 + [i32, i64, i32, i32, i32, i32, i32, i32, vp]
    lib.stg_gcn_agg_edge.restype = ctypes.c_int
    lib.stg_gcn_agg_edge.argtypes = [vp] * 9 + [i32, i64, i32, i32, vp]
    lib.stg_gcn_layer_fwd.restype = ctypes.c_int
    lib.stg_gcn_layer_fwd.argtypes = [vp] * 5 + [i32] + [vp] * 5 + [i32, i64, i32, vp]
    lib.stg_bias_act_fwd.restype = ctypes.c_int
    lib.stg_bias_act_fwd.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.stg_bias_act_bwd_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_bias_act_bwd_workspace_bytes.argtypes = [i32, i32]
    lib.stg_bias_act_bwd.restype = ctypes.c_int
    lib.stg_bias_act_bwd.argtypes = [vp] * 4 + [i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_gcn_agg_transform.restype = ctypes.c_int
    lib.stg_gcn_agg_transform.argtypes = [vp] * 10 + [i32, i32, i32, vp]
    lib.stg_edge_gather_f32.restype = ctypes.c_int
    lib.stg_edge_gather_f32.argtypes = [vp, vp, vp, i64, vp]
    lib.stg_gat_fc_supported.restype = ctypes.c_int
    lib.stg_gat_fc_supported.argtypes = [i32, i32, i32]
    lib.stg_gat_fc_fwd.restype = ctypes.c_int
    lib.stg_gat_fc_fwd.argtypes = [vp] * 7 + [i32, i32, i32, i32, vp]
    lib.stg_gat_fwd_k0.restype = ctypes.c_int
    lib.stg_gat_fwd_k0.argtypes = [vp] * 8 + [i32, i32, i32, f32, vp, vp]
    lib.stg_gat_score_flag.restype = ctypes.c_int
    lib.stg_gat_score_flag.argtypes = [vp, vp, i64, vp, vp]
    lib.stg_gat_fwd_k1.restype = ctypes.c_int
    lib.stg_gat_fwd_k1.argtypes = [vp] * 8 + [i32, i32, i32, i32, vp, vp]
    lib.stg_gat_bwd.restype = ctypes.c_int
    lib.stg_gat_bwd.argtypes = [vp] * 14 + [i32, i32, i32, i32, f32, vp, vp]
    lib.stg_gat_bwd_factored.restype = ctypes.c_int
    lib.stg_gat_bwd_factored.argtypes = [vp] * 13 + [i32, i32, i32, f32, vp, vp, vp]
    lib.stg_gat_bwd_factored_elu.restype = ctypes.c_int
    lib.stg_gat_bwd_factored_elu.argtypes = [vp] * 14 + [i32, i32, i32, f32, vp, vp, vp]
    lib.stg_tgcn_fold_weights.restype = ctypes.c_int
    lib.stg_tgcn_fold_weights.argtypes = [vp] * 8 + [i32, i32, vp]
    lib.stg_tgcn_unfold_gate_grads.restype = ctypes.c_int
    lib.stg_tgcn_unfold_gate_grads.argtypes = [vp] * 9 + [i32, i32, vp]
    lib.stg_gat_fwd_k1_uniform.restype = ctypes.c_int
    lib.stg_gat_fwd_k1_uniform.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    lib.stg_gat_fc_out.restype = ctypes.c_int
    lib.stg_gat_fc_out.argtypes = [vp] * 4 + [i32] * 4 + [vp]
    lib.stg_gat_fwd_k1_scored.restype = ctypes.c_int
    lib.stg_gat_fwd_k1_scored.argtypes = [vp] * 9 + [i32, i32, i32, vp, vp]
    lib.stg_gat_bwd_er.restype = ctypes.c_int
    lib.stg_gat_bwd_er.argtypes = [vp] * 5 + [i32, i32, i32, vp]
    lib.stg_gat_proj_supported.restype = ctypes.c_int
    lib.stg_gat_proj_supported.argtypes = [i32, i32]
    lib.stg_gat_proj_fwd.restype = ctypes.c_int
    lib.stg_gat_proj_fwd.argtypes = [vp] * 5 + [i64, i32, i32, vp]
    lib.stg_gat_proj_bwd_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_gat_proj_bwd_workspace_bytes.argtypes = [i64, i32, i32]
    lib.stg_gat_proj_bwd.restype = ctypes.c_int
    lib.stg_gat_proj_bwd.argtypes = [vp] * 9 + [i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_gemm_tn_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_gemm_tn_workspace_bytes.argtypes = [i64, i32, i32]
    lib.stg_gemm_tn_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_f32.argtypes = [vp, vp, vp, i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_gemm_tn_multi_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_gemm_tn_multi_workspace_bytes.argtypes = [i32, i64, i32, i32]
    lib.stg_gemm_tn_multi_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_multi_f32.argtypes = [vp, vp, i32, vp, vp, i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_rowgemm_supported.restype = ctypes.c_int
    lib.stg_rowgemm_supported.argtypes = [i32, i32]
    lib.stg_rowgemm_f32.restype = ctypes.c_int
    lib.stg_rowgemm_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, ctypes.c_int, vp]
    lib.stg_rowgemm_act_supported.restype = ctypes.c_int
    lib.stg_rowgemm_act_supported.argtypes = [i32, i32]
    lib.stg_rowgemm_act_f32.restype = ctypes.c_int
    lib.stg_rowgemm_act_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, ctypes.c_int, ctypes.c_int, vp]
    lib.stg_gat_fc_feat_if.restype = ctypes.c_int
    lib.stg_gat_fc_feat_if.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.stg_gat_bwd_uniform_supported.restype = ctypes.c_int
    lib.stg_gat_bwd_uniform_supported.argtypes = [i32, i32, i32]
    lib.stg_gat_attn_fold.restype = ctypes.c_int
    lib.stg_gat_attn_fold.argtypes = [vp] * 8 + [i32, i32, i32, vp]
    lib.stg_gat_bwd_prepass_heads_supported.restype = ctypes.c_int
    lib.stg_gat_bwd_prepass_heads_supported.argtypes = [i64, i32, i32, i32]
    lib.stg_gat_bwd_prepass_heads.restype = ctypes.c_int
    lib.stg_gat_bwd_prepass_heads.argtypes = [vp] * 8 + [i64, i32, i32, i32, ctypes.c_float, vp]
    lib.stg_gat_bwd_prepass.restype = ctypes.c_int
    lib.stg_gat_bwd_prepass.argtypes = [vp] * 5 + [i32, i32, i32, ctypes.c_float, vp, vp]
    lib.stg_gat_bwd_uniform_edges.restype = ctypes.c_int
    lib.stg_gat_bwd_uniform_edges.argtypes = [vp] * 19 + [i32, ctypes.c_float, vp, vp]
    lib.stg_gat_bwd_uniform_gx_fallback.restype = ctypes.c_int
    lib.stg_gat_bwd_uniform_gx_fallback.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.stg_gemm_tn_gated_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_gated_f32.argtypes = [vp, vp, vp, i64, i32, i32, vp, ctypes.c_size_t, vp, ctypes.c_int, vp]
    lib.stg_rowgemm_heads_supported.restype = ctypes.c_int
    lib.stg_rowgemm_heads_supported.argtypes = [i64, i32, i32, i32]
    lib.stg_rowgemm_heads_f32.restype = ctypes.c_int
    lib.stg_rowgemm_heads_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp]
    lib.stg_rowgemm_bits_words.restype = ctypes.c_size_t
    lib.stg_rowgemm_bits_words.argtypes = [i64]
    lib.stg_rowgemm_bits_supported.restype = ctypes.c_int
    lib.stg_rowgemm_bits_supported.argtypes = [i64, i32, i32]
    lib.stg_rowgemm_act_bits_f32.restype = ctypes.c_int
    lib.stg_rowgemm_act_bits_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, ctypes.c_int, ctypes.c_int, vp, vp, vp]
    lib.stg_rowgemm_strided_f32.restype = ctypes.c_int
    lib.stg_rowgemm_strided_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, ctypes.c_int, vp]
    lib.stg_gemm_tn_colsum_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_colsum_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_gemm_tn_relu_mask_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_relu_mask_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_tgcn_cell_fused_supported.restype = ctypes.c_int
    lib.stg_tgcn_cell_fused_supported.argtypes = [i32]
    lib.stg_tgcn_cell_fused_fwd.restype = ctypes.c_int
    lib.stg_tgcn_cell_fused_fwd.argtypes = [vp] * 16 + [i64, i32, f32, f32, vp]
    lib.stg_tgcn_cell_fused_bwd.restype = ctypes.c_int
    lib.stg_tgcn_cell_fused_bwd.argtypes = [vp] * 15 + [i64, i32, f32, f32, vp]
    lib.stg_tgcn_cell_fused_bwd_dx_supported.restype = ctypes.c_int
    lib.stg_tgcn_cell_fused_bwd_dx_supported.argtypes = [i32, i32]
    lib.stg_tgcn_cell_fused_bwd_dx.restype = ctypes.c_int
    lib.stg_tgcn_cell_fused_bwd_dx.argtypes = [vp] * 17 + [i64, i32, i32, f32, f32, vp]
    lib.stg_tgcn_head_supported.restype = ctypes.c_int
    lib.stg_tgcn_head_supported.argtypes = [i32, i32, i32]
    lib.stg_tgcn_head_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_tgcn_head_workspace_bytes.argtypes = [i64]
    lib.stg_tgcn_head_fwd.restype = ctypes.c_int
    lib.stg_tgcn_head_fwd.argtypes = [vp] * 10 + [i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_tgcn_head_fwd_acc.restype = ctypes.c_int
    lib.stg_tgcn_head_fwd_acc.argtypes = [vp] * 11 + [i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_tgcn_head_bwd.restype = ctypes.c_int
    lib.stg_tgcn_head_bwd.argtypes = [vp] * 11 + [i64, i32, i32, vp]
    lib.stg_xent_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_xent_workspace_bytes.argtypes = [i64, i32]
    lib.stg_xent_fwd.restype = ctypes.c_int
    lib.stg_xent_fwd.argtypes = [vp] * 6 + [i64, i32, vp, ctypes.c_size_t, vp]
    lib.stg_xent_bwd.restype = ctypes.c_int
    lib.stg_xent_bwd.argtypes = [vp] * 6 + [i64, i64, i32, vp]
    lib.stg_xent_bwd_colsum_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_xent_bwd_colsum_workspace_bytes.argtypes = [i64, i32]
    lib.stg_xent_fwd_grad_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_xent_fwd_grad_workspace_bytes.argtypes = [i64, i32]
    lib.stg_xent_fwd_grad.restype = ctypes.c_int
    lib.stg_xent_fwd_grad.argtypes = [vp] * 8 + [i64, i64, i32, vp, ctypes.c_size_t, vp]
    lib.stg_xent_scale_grad.restype = ctypes.c_int
    lib.stg_xent_scale_grad.argtypes = [vp, vp, vp, i64, i32, vp]
    lib.stg_gemm_tn_small_supported.restype = ctypes.c_int
    lib.stg_gemm_tn_small_supported.argtypes = [i64, i32, i32]
    lib.stg_gemm_tn_small_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_small_f32.argtypes = [vp, vp, vp, i64, i32, i32, vp]
    lib.stg_mm_bwd_small_supported.restype = ctypes.c_int
    lib.stg_mm_bwd_small_supported.argtypes = [i64, i32, i32]
    lib.stg_mm_bwd_small.restype = ctypes.c_int
    lib.stg_mm_bwd_small.argtypes = [vp] * 6 + [i64, i32, i32, vp]
    lib.stg_xent_small_supported.restype = ctypes.c_int
    lib.stg_xent_small_supported.argtypes = [i64, i32]
    lib.stg_xent_small_fwd.restype = ctypes.c_int
    lib.stg_xent_small_fwd.argtypes = [vp] * 6 + [i64, i32, vp]
    lib.stg_xent_small_bwd.restype = ctypes.c_int
    lib.stg_xent_small_bwd.argtypes = [vp] * 7 + [i64, i64, i32, vp]
    lib.stg_xent_bwd_colsum.restype = ctypes.c_int
    lib.stg_xent_bwd_colsum.argtypes = [vp] * 7 + [i64, i64, i32, vp, ctypes.c_size_t, vp]
    lib.stg_link_head_supported.restype = ctypes.c_int
    lib.stg_link_head_supported.argtypes = [i32, i32]
    lib.stg_link_head_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_link_head_workspace_bytes.argtypes = [i64]
    lib.stg_link_head_fwd.restype = ctypes.c_int
    lib.stg_link_head_fwd.argtypes = [vp] * 11 + [i64, i64, i32, i32, vp, ctypes.c_size_t, vp]
    lib.stg_link_head_bwd.restype = ctypes.c_int
    lib.stg_link_head_bwd.argtypes = [vp] * 13 + [i64, i64, i32, i32, vp]
    for name, nptr, tail in (("stg_tgcn_cell_prep_fwd", 6, [i64, i32, f32, f32, vp]),
                             ("stg_tgcn_cell_gates_fwd", 6, [i64, i32, vp]),
                             ("stg_tgcn_cell_update_fwd", 5, [i64, i32, vp]),
                             ("stg_tgcn_cell_update_bwd", 7, [i64, i32, vp]),
                             ("stg_tgcn_cell_gates_bwd", 5, [i64, i32, vp]),
                             ("stg_tgcn_cell_prep_bwd", 7, [i64, i32, f32, f32, vp])):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [vp] * nptr + tail
    lib.stg_gemm_tn_form_workspace_bytes.restype = ctypes.c_size_t
    lib.stg_gemm_tn_form_workspace_bytes.argtypes = [i32, i64, i32, i32, i32]
    lib.stg_gemm_tn_form_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_form_f32.argtypes = [vp, i32, vp, i32, i32, vp, i32, i32, f32, f32, i32, vp, vp, i64, i32, i32, vp,
                                         ctypes.c_size_t, vp]
    lib.stg_gemm_tn_form_partial_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_form_partial_f32.argtypes = [vp, i32, vp, i32, i32, vp, i32, i32, f32, f32, i32, i32, i64, i32, i32, vp,
                                                 ctypes.c_size_t, vp, vp]
    lib.stg_gemm_tn_reduce_multi_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_reduce_multi_f32.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    lib.stg_gemm_tn_reduce_multi_blocks_f32.restype = ctypes.c_int
    lib.stg_gemm_tn_reduce_multi_blocks_f32.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.stg_link_decode_fwd_multi.restype = ctypes.c_int
    lib.stg_link_decode_fwd_multi.argtypes = [i32, vp, vp, vp, vp, vp, i64, i32, vp]
    lib.stg_tgcn_pack_weights.restype = ctypes.c_int
    lib.stg_tgcn_pack_weights.argtypes = [vp] * 17 + [i32, i32, i32, vp]
    lib.stg_tgcn_step_supported.restype = ctypes.c_int
    lib.stg_tgcn_step_supported.argtypes = [i32, i32, i32]
    lib.stg_tgcn_step_loss_partials.restype = ctypes.c_size_t
    lib.stg_tgcn_step_loss_partials.argtypes = [i64]
    lib.stg_tgcn_step_fwd.restype = ctypes.c_int
    lib.stg_tgcn_step_fwd.argtypes = [ctypes.POINTER(TgcnStepFwdArgs), vp]
    lib.stg_tgcn_step_bwd.restype = ctypes.c_int
    lib.stg_tgcn_step_bwd.argtypes = [ctypes.POINTER(TgcnStepBwdArgs), vp]
    lib.stg_tgcn_window_loss.restype = ctypes.c_int
    lib.stg_tgcn_window_loss.argtypes = [vp, i32, i64, i64, vp, vp, vp]
    lib.stg_degree_norm_f32.restype = ctypes.c_int
    lib.stg_degree_norm_f32.argtypes = [vp, vp, vp, i64, vp]
    lib.stg_partial_sums_loss.restype = ctypes.c_int
    lib.stg_partial_sums_loss.argtypes = [vp, i32, i32, i64, f32, vp, vp, vp]
    lib.stg_link_decode_fwd.restype = ctypes.c_int
    lib.stg_link_decode_fwd.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp]
    lib.stg_link_decode_bwd.restype = ctypes.c_int
    lib.stg_link_decode_bwd.argtypes = [vp] * 8 + [i64, i64, i32, vp]
    if lib.stg_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.stg_abi_version()} != expected {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != 0:
        raise StgError(rc, lib.stg_last_error_string().decode(errors="replace"))


# launch-time knobs of the native library (performance only; "rowgemm_x3" alone selects between two arithmetics, both inside the
# fp32 kernel's error bound): include/stgraph_hip.h, stg_set_tuning.  Every key defaults to 0 (= auto).
TUNING_KEYS = ("gcn_lanes_per_row", "gcn_unroll", "gcn_long_threshold", "xw_rows", "xw_waves", "cell_rows", "gcn_tile", "gcn_block",
               "gcn_addr32", "gcn_tile_pipe", "gcn_tile_rows", "gcn_xcd_tile", "step_waves", "gcn_wide_long", "step_spread", "step_coop",
               "build_lds_count", "store_rows", "rowgemm16", "rowgemm_x3", "gemm_wide", "gemm_x3", "gemm_xcd_pair", "gemm_cyclic")
_TUNING_SET = {}


def set_tuning(key: str, value: int) -> None:
    check(lib.stg_set_tuning(key.encode(), int(value)))
    _TUNING_SET[key] = int(value)


def tuning_values() -> dict:
    """Every native knob -> the value last set through this module (0 = the library's default)."""
    return {k: _TUNING_SET.get(k, 0) for k in TUNING_KEYS}
