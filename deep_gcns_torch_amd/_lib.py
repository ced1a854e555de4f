"""ctypes binding of libdgcn.so (the C ABI declared in include/dgcn.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module
raises, and every hot-path op of the package fails loudly with it.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch  # noqa: F401  (must be imported first: libdgcn binds to the HIP runtime torch loaded)

_LIB_PATH = Path(os.environ.get("DGCN_LIB_PATH") or (Path(__file__).resolve().parent / "csrc" / "libdgcn.so"))

# aggregation modes / flags (include/dgcn.h)
AGGR_ADD, AGGR_MEAN, AGGR_MAX, AGGR_SOFTMAX, AGGR_POWER = 0, 1, 2, 3, 4
MSG_IDENTITY, MSG_RELU_EPS = 0, 1
FLAG_LEARN_T, FLAG_LEARN_P, FLAG_ADD_ROOT, FLAG_SHIFT_FLAG_IS_RANGE, FLAG_EA_IS_Z, FLAG_STATIC_ITEMS = 1, 2, 4, 8, 16, 32

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)


class DgcnGraph(C.Structure):
    """struct dgcn_graph (include/dgcn.h)."""

    _fields_ = [
        ("n_dst", C.c_int32), ("n_src", C.c_int32), ("n_edges", C.c_int32), ("reserved", C.c_int32),
        ("rowptr", C.c_void_p), ("col", C.c_void_p), ("eperm", C.c_void_p),
        ("t_rowptr", C.c_void_p), ("t_col", C.c_void_p), ("t_eperm", C.c_void_p),
        ("n_work", C.c_int32), ("n_slots", C.c_int32),
        ("work_row", C.c_void_p), ("work_beg", C.c_void_p), ("work_end", C.c_void_p), ("work_slot", C.c_void_p),
        ("t_n_work", C.c_int32), ("t_n_slots", C.c_int32),
        ("t_work_row", C.c_void_p), ("t_work_beg", C.c_void_p), ("t_work_end", C.c_void_p), ("t_work_slot", C.c_void_p),
        ("n_split", C.c_int32), ("t_n_split", C.c_int32), ("split_item", C.c_void_p), ("t_split_item", C.c_void_p),
    ]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
_SIGNATURES = {
    "dgcn_version": (C.c_int, []),
    "dgcn_strerror": (C.c_char_p, [C.c_int]),
    "dgcn_selftest_axpy_f32": (C.c_int, [C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dgcn_gen_aggr_fwd_workspace_bytes": (C.c_size_t, [C.POINTER(DgcnGraph), C.c_int32]),
    "dgcn_gen_aggr_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(DgcnGraph), C.c_int32]),
    "dgcn_gen_aggr_fwd_f32": (C.c_int, [
        C.POINTER(DgcnGraph), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
        C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_gen_aggr_bwd_f32": (C.c_int, [
        C.POINTER(DgcnGraph), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
        C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_size_t, C.c_void_p]),
    "dgcn_gen_aggr_max_mask_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "dgcn_gen_aggr_max_bwd_f32": (C.c_int, [
        C.POINTER(DgcnGraph), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
        C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
        C.c_void_p]),
    "dgcn_gen_aggr_enc_fwd_f32": (C.c_int, [
        C.POINTER(DgcnGraph), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_gen_aggr_enc_bwd_num_partials": (C.c_int32, [C.POINTER(DgcnGraph), C.c_int32]),
    "dgcn_gen_aggr_enc_bwd_f32": (C.c_int, [
        C.POINTER(DgcnGraph), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_enc_max_bwd_num_partials": (C.c_int32, [C.c_int32]),
    "dgcn_enc_max_bwd_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_void_p]),
    "dgcn_enc_compose_fwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "dgcn_enc_compose_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dgcn_gen_aggr_egemm_supported": (C.c_int32, [C.c_int32, C.c_int32]),
    "dgcn_gen_aggr_egemm_fwd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "dgcn_gen_aggr_egemm_fwd_f32": (C.c_int, [
        C.POINTER(DgcnGraph), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_egemm_max_bwd_num_partials": (C.c_int32, [C.c_int32]),
    "dgcn_egemm_max_bwd_f32": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dgcn_power_bwd_prep_f32": (C.c_int, [C.POINTER(DgcnGraph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                          C.c_int32, C.c_void_p]),
    "dgcn_softmax_bwd_prep_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_int32, C.c_void_p]),
    "dgcn_softmax_state_merge_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_graph_csr_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "dgcn_graph_csr_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "dgcn_graph_work_list_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "dgcn_graph_work_list": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_graph_coalesce_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "dgcn_graph_coalesce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_subgraph_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "dgcn_subgraph_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "dgcn_knn_dense_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "dgcn_knn_dense_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_vertex_gemm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "dgcn_dense_edge_reduce_num_partials": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "dgcn_dense_edge_reduce_fwd_f32": (C.c_int, [
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p]),
    "dgcn_dense_edge_reduce_bwd_f32": (C.c_int, [
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "dgcn_dense_edge_reduce_bwd_nsplit": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "dgcn_edgeconv_pq_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "dgcn_bn_finalize_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                       C.c_void_p, C.c_void_p]),
    "dgcn_bn_apply_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p]),
    "dgcn_bn_apply_res_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dgcn_bn_bwd_num_partials": (C.c_int32, [C.c_int32, C.c_int32]),
    "dgcn_bn_bwd_prep_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p]),
    "dgcn_bn_bwd_finalize_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "dgcn_dense_edge_reduce_bwd_inv_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "dgcn_dense_edge_reduce_bwd_inv_f32": (C.c_int, [
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dgcn_edgeconv_bwd_input_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                              C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                              C.c_void_p]),
    "dgcn_edgeconv_bwd_weight_num_partials": (C.c_int32, [C.c_int32, C.c_int32]),
    "dgcn_edgeconv_bwd_weight_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                               C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dgcn_reduce_parts_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                        C.c_void_p]),
    "dgcn_reduce_partials_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "dgcn_reduce_partials_split_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]),
    "dgcn_rows_num_partials": (C.c_int32, [C.c_int64, C.c_int32]),
    "dgcn_rows_stats_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "dgcn_rows_bn_apply_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                         C.c_int32, C.c_void_p]),
    "dgcn_rows_bn_bwd_stats_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_bn_bwd_finalize_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_void_p,
                                                C.c_void_p]),
    "dgcn_rows_bn_bwd_apply_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_ln_num_partials": (C.c_int32, [C.c_int64, C.c_int32]),
    "dgcn_rows_ln_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_ln_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_bn_act_apply_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int64, C.c_int32,
                                             C.c_void_p]),
    "dgcn_rows_bn_act_bwd_stats_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                                 C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                                 C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_bn_act_bwd_apply_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_ln_act_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_ln_act_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                           C.c_void_p]),
    "dgcn_rows_linear_supported": (C.c_int32, [C.c_int32, C.c_int32]),
    "dgcn_rows_linear_num_partials": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32]),
    "dgcn_rows_linear_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "dgcn_rows_tn_supported": (C.c_int32, [C.c_int32, C.c_int32]),
    "dgcn_rows_tn_num_partials": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32]),
    "dgcn_rows_tn_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dgcn_rows_tn_colsum_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p]),
    "dgcn_rows_msgnorm_fwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                            C.c_int64, C.c_int32, C.c_void_p]),
    "dgcn_rows_msgnorm_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
}

_lib = None


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load libdgcn.so once and declare every prototype.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} not found: build it with `python -m deep_gcns_torch_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU/eager fallback for the hot path.")
    lib = C.CDLL(os.fspath(_LIB_PATH), mode=C.RTLD_LOCAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dgcn_strerror(rc)
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def sum_partials(parts: "torch.Tensor") -> "torch.Tensor":
    """``parts.sum(0)`` of a contiguous fp32 (nparts, ...) block of per-workgroup partial sums in ONE launch
    (dgcn_reduce_partials_f32: fixed order, bit-reproducible)."""
    lib = load()
    nparts = parts.size(0)
    out = torch.empty(parts.shape[1:], device=parts.device, dtype=torch.float32)
    width = out.numel()
    if width == 0:
        return out
    if nparts == 0:
        return out.zero_()
    dev = parts.device
    with device_ctx(dev):
        check(lib.dgcn_reduce_partials_f32(parts.data_ptr(), nparts, width, out.data_ptr(), current_stream_handle(dev)),
              "dgcn_reduce_partials_f32")
    return out


def sum_partials_split(parts: "torch.Tensor"):
    """``parts.sum(0)`` of a contiguous fp32 (nparts, rows, inner) block as two contiguous tensors: ``[:, :-1]`` (rows,
    inner - 1) and ``[:, -1]`` (rows,) -- one launch, no slicing copies (dgcn_reduce_partials_split_f32)."""
    lib = load()
    nparts, rows, inner = parts.shape
    dev = parts.device
    out = torch.empty(rows, inner - 1, device=dev, dtype=torch.float32)
    last = torch.empty(rows, device=dev, dtype=torch.float32)
    if rows == 0:
        return out, last
    if nparts == 0:
        return out.zero_(), last.zero_()
    with device_ctx(dev):
        check(lib.dgcn_reduce_partials_split_f32(parts.data_ptr(), nparts, rows, inner, out.data_ptr(), last.data_ptr(),
                                                 current_stream_handle(dev)), "dgcn_reduce_partials_split_f32")
    return out, last


def ptr(t) -> int | None:
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


class _DeviceCtx:
    """torch.cuda.device without its per-call index resolution (the launches of a host-bound step enter it ~250 times)."""
    __slots__ = ("idx", "prev")

    def __init__(self, idx: int):
        self.idx, self.prev = idx, -1

    def __enter__(self):
        self.prev = torch.cuda._exchange_device(self.idx)
        return self

    def __exit__(self, *exc):
        torch.cuda._maybe_exchange_device(self.prev)
        return False


_FAST_CTX = hasattr(torch.cuda, "_exchange_device") and hasattr(torch.cuda, "_maybe_exchange_device")
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _index_of(device) -> int:
    idx = device.index if isinstance(device, torch.device) else device
    return torch.cuda.current_device() if idx is None else idx


def device_ctx(device):
    """Make ``device`` the HIP runtime's current device for the enclosed launches.  Always entered, also when torch
    already reports it as current: in autograd worker threads a ctypes launch without it costs ~25 us more per
    kernel (measured), the context manager itself ~1.3 us."""
    if _FAST_CTX:
        return _DeviceCtx(_index_of(device))
    return torch.cuda.device(device)


def current_stream_handle(device) -> int:
    """Raw hipStream_t of torch's CURRENT stream on ``device`` (queried per launch: the caller may be inside a
    ``torch.cuda.stream`` block or a graph capture)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(_index_of(device))
    return torch.cuda.current_stream(device).cuda_stream


def require_device(*tensors) -> torch.device:
    """All hot-path ops need device tensors; this is an error check, not a dispatch."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "deep_gcns_torch_amd: the message-passing hot path runs only as HIP kernels on an "
                "MI355X (got a CPU tensor). There is no CPU fallback.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev
