"""ctypes binding of libyolat_hip.so (C ABI declared in include/yolat_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing this module raises.  Build it with ``python __graft_entry__.py`` (or
``make -C yolat_vectorgraphicsrecognition_amd/csrc``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (YOLAT_LIB_PATH: another build of the same ABI — A/B measurements of one kernel on one box, tools/exp/)
LIB_PATH = os.environ.get("YOLAT_LIB_PATH") or os.path.join(_HERE, "libyolat_hip.so")

c_p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f = ctypes.c_float
c_sz = ctypes.c_size_t

YOLAT_MAX_LAYERS = 8


class ConvEval(ctypes.Structure):
    """yolat_conv_eval (include/yolat_hip.h)"""
    _fields_ = ([("Cin", c_i64)] + [(n, c_p) for n in
                                    ("W1", "b1", "s1", "t1", "W2", "b2", "s2", "t2", "Wr", "br", "Wn", "bn", "sn", "tn",
                                     "Wuv", "Wc4", "Wuvf", "uvb", "Wc4f", "t2f")] +
                [("Wfr_x6", c_p * 3), ("tfr", c_p), ("Wn_x6", c_p * 3), ("tn_fold", c_p), ("Wnx", c_p), ("tnx", c_p)])


class ModelEval(ctypes.Structure):
    """yolat_model_eval (include/yolat_hip.h)"""
    _fields_ = ([("n_blocks", ctypes.c_int32), ("n_blocks_out", ctypes.c_int32), ("n_classes", ctypes.c_int32),
                 ("reserved", ctypes.c_int32), ("C", c_i64), ("F", c_i64), ("H1", c_i64), ("H2", c_i64),
                 ("conv", ConvEval * YOLAT_MAX_LAYERS)] +
                [(n, c_p) for n in ("Wf", "bf", "sf", "tf", "Wf_hi", "Wf_mid", "Wf_lo", "tf_fold", "Wfs_hi", "Wfs_mid",
                                    "Wfs_lo", "tfs_fold", "Wfs", "bfs", "sfs", "tfs", "Wc1", "bc1", "sc1", "tc1", "Wc2",
                                    "bc2", "sc2", "tc2", "Wc3", "bc3")] +
                [("Wc_x6", c_p * 3), ("tc_fold", c_p * 3), ("Wc1_gx", c_p), ("tc1_gx", c_p)])


class BnCsrGrad(ctypes.Structure):
    """yolat_bn_csr_grad (include/yolat_hip.h)"""
    _fields_ = [("d_out", c_p), ("ld_out", c_i64), ("dst", c_p), ("inv_deg", c_p), ("Y", c_p), ("ldy", c_i64),
                ("mean", c_p), ("invstd", c_p), ("scale", c_p), ("shift", c_p), ("coef", c_p),
                ("relu", ctypes.c_int32), ("half", ctypes.c_int32)]


class ModelEvalBf16(ctypes.Structure):
    """yolat_model_eval_bf16 (include/yolat_hip.h)"""
    _fields_ = ([("base", ctypes.POINTER(ModelEval))] +
                [(n, c_p * YOLAT_MAX_LAYERS) for n in ("Wuv", "Wr", "Wn", "W2", "uv_scale", "uv_shift")] +
                [(n, c_p) for n in ("Wf", "Wfs", "Wc1", "Wc2", "Wc3")] +
                [("t2f", c_p * YOLAT_MAX_LAYERS)] +
                [(n, c_p) for n in ("Wf_fold", "Wfs_fold", "tf_fold", "tfs_fold", "conv_local")])


class Span(ctypes.Structure):
    """yolat_span"""
    _fields_ = [("ptr", c_p), ("bytes", c_i64)]


class ItemCsr(ctypes.Structure):
    """yolat_item_csr"""
    _fields_ = [("N", c_i64), ("E", c_i64), ("P", c_i64), ("row_ptr", c_p), ("src", c_p), ("dst", c_p), ("attr", c_p),
                ("seg_ptr", c_p), ("node_seg", c_p)]


YOLAT_MAX_KEYS = 8


class ItemDesc(ctypes.Structure):
    """yolat_item_desc"""
    _fields_ = [("n_keys", c_i64), ("key", Span * YOLAT_MAX_KEYS), ("rows", c_i64 * YOLAT_MAX_KEYS), ("csr", ItemCsr),
                ("fix", ctypes.c_int32 * YOLAT_MAX_KEYS), ("node_key", ctypes.c_int32), ("prop_key", ctypes.c_int32)]


class LoaderBatch(ctypes.Structure):
    """yolat_loader_batch"""
    _fields_ = [("device", c_p), ("total", c_i64), ("n_keys", c_i64), ("B", c_i64), ("N", c_i64), ("E", c_i64), ("P", c_i64),
                ("off", c_i64 * (YOLAT_MAX_KEYS + 6)), ("slices", ctypes.POINTER(c_i64)), ("slot", ctypes.c_int32),
                ("rc", ctypes.c_int32)]


class Locality(ctypes.Structure):
    """yolat_locality"""
    _fields_ = [("known", ctypes.c_int32), ("flags", ctypes.c_int32), ("max_nodes", ctypes.c_int32),
                ("max_edges", ctypes.c_int32)]


class TrainLin(ctypes.Structure):
    """yolat_train_lin"""
    _fields_ = [("W", c_p), ("b", c_p)]


class TrainBn(ctypes.Structure):
    """yolat_train_bn"""
    _fields_ = [("gamma", c_p), ("beta", c_p), ("running_mean", c_p), ("running_var", c_p), ("num_batches_tracked", c_p),
                ("momentum", c_f), ("eps", c_f)]


class TrainConv(ctypes.Structure):
    """yolat_train_conv"""
    _fields_ = [("Cin", c_i64), ("nn0", TrainLin), ("bn1", TrainBn), ("nn3", TrainLin), ("bn4", TrainBn), ("lin_r", TrainLin),
                ("node", TrainLin), ("bn_node", TrainBn)]


class TrainModel(ctypes.Structure):
    """yolat_train_model"""
    _fields_ = [("n_blocks", ctypes.c_int32), ("n_blocks_out", ctypes.c_int32), ("n_classes", ctypes.c_int32),
                ("half", ctypes.c_int32), ("C", c_i64), ("F", c_i64), ("H1", c_i64), ("H2", c_i64),
                ("conv", TrainConv * YOLAT_MAX_LAYERS), ("fus", TrainLin), ("fus_bn", TrainBn), ("fus_s", TrainLin),
                ("fus_s_bn", TrainBn), ("c1", TrainLin), ("c1_bn", TrainBn), ("c2", TrainLin), ("c2_bn", TrainBn),
                ("c3", TrainLin), ("param_base", c_p), ("grad_base", c_p)]


class AdamArgs(ctypes.Structure):
    """yolat_adam_args"""
    _fields_ = [("exp_avg", c_p), ("exp_avg_sq", c_p), ("n", c_i64), ("step", c_i64), ("lr", c_f), ("beta1", c_f),
                ("beta2", c_f), ("eps", c_f), ("weight_decay", c_f), ("grad_scale", c_f)]


class PredictTree(ctypes.Structure):
    """yolat_predict_tree"""
    _fields_ = [("R", c_i64), ("Ctot", c_i64), ("B", c_i64), ("root_row", c_p), ("root_range", c_p), ("child_ptr", c_p),
                ("child_row", c_p), ("child_range", c_p), ("image_root_ptr", c_p)]


class GraphCsr(ctypes.Structure):
    """yolat_graph_csr"""
    _fields_ = [("row_ptr", c_p), ("src", c_p), ("dst", c_p), ("attr", c_p), ("seg_ptr", c_p), ("node_seg", c_p)]


# name -> (restype, argtypes); order mirrors include/yolat_hip.h
SIGNATURES = {
    "yolat_abi_version": (c_int, []),
    "yolat_strerror": (ctypes.c_char_p, [c_int]),
    "yolat_csr_work_elems": (c_sz, [c_i64, c_i64]),
    "yolat_coo_to_csr": (c_int, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yolat_edge_mlp2_eval": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                      c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_node_side_eval": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64,
                                      c_p, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_fusion_pool_train_saved_elems": (c_sz, [c_i64, c_i64, c_i64]),
    "yolat_fusion_pool_train_work_elems": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "yolat_fusion_pool_train_fwd": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_f, c_f,
                                             c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p]),
    "yolat_fusion_pool_train_bwd": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_p,
                                             c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    "yolat_fusion_pool_train_bwd_parts": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_p,
                                                   c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_int, c_p]),
    "yolat_expand_ranges": (c_int, [c_p, c_p, c_i64, c_i64, c_p, c_p]),
    "yolat_subgraph_work_elems": (c_sz, [c_i64, c_i64]),
    "yolat_subgraph_reindex": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yolat_gather_rows_bytes": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
    "yolat_fixup_offsets": (c_int, [c_p, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_i64, c_p]),
    "yolat_conv_split_w1": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "yolat_node_uv_eval": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64,
                                    c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_edge_uv_mlp2_eval": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64,
                                         c_p, c_i64, c_p]),
    "yolat_edge_uv_mlp2_mean_eval": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p,
                                              c_p, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_edge_uv_mlp2_mean_eval_variant": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p,
                                                      c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_int, c_p]),
    "yolat_edge_uv_sums": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
    "yolat_conv_merge_dw1": (c_int, [c_p, c_p, c_i64, c_i64, c_p, c_i64, c_int, c_p]),
    "yolat_fusion_pair_eval": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64,
                                        c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "yolat_graph_prepare_node_uv": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64] + [c_p] * 9 +
                                    [c_p, c_i64, c_i64] + [c_p] * 8 + [c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_csc_work_elems": (c_sz, [c_i64]),
    "yolat_csc_by_source": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p]),
    "yolat_segment_ptr": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p]),
    "yolat_graph_work_elems": (c_sz, [c_i64, c_i64]),
    "yolat_graph_prepare": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p,
                                    c_p, c_p, c_p, c_p, c_p]),
    "yolat_gather_rows": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
    "yolat_bn_stats_elems": (c_sz, [c_i64, c_i64]),
    "yolat_linear_fwd": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int,
                                 c_p, c_i64, c_p, c_i64, c_p, c_p, c_int,
                                 c_p, c_i64, c_int, c_p, c_p]),
    "yolat_linear_segmax_fwd": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64,
                                        c_p]),
    "yolat_linear_fwd_wt": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_int, c_p]),
    "yolat_linear_bwd_w_work_elems": (c_sz, [c_i64, c_i64, c_i64]),
    "yolat_linear_bwd_w": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_int,
                                   c_p, c_i64, c_p, c_int, c_p, c_p]),
    "yolat_bn_finalize": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    "yolat_bn_eval_coeffs": (c_int, [c_p, c_p, c_p, c_p, c_f, c_i64, c_p, c_p, c_p]),
    "yolat_scale_shift_relu": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64, c_p]),
    "yolat_bn_bwd_work_elems": (c_sz, [c_i64, c_i64]),
    "yolat_bn_relu_bwd": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_int,
                                  c_p, c_p, c_int, c_p, c_i64, c_p, c_p]),
    "yolat_edge_lin1_fwd": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p,
                                    c_i64, c_p, c_p, c_int, c_p, c_i64, c_p, c_p]),
    "yolat_edge_lin1_bwd_w": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p,
                                      c_p, c_i64, c_p, c_int, c_p, c_p]),
    "yolat_edge_lin1_bwd_x": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
    "yolat_edge_scatter_bwd": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_int, c_p]),
    "yolat_csr_mean_fwd": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64, c_p, c_i64, c_int, c_p]),
    "yolat_csr_mean_bwd": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_segment_mean_fwd": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_segment_max_fwd": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64, c_i64, c_p, c_i64,
                                      c_p, c_p]),
    "yolat_pool_prepare": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_segment_mean_bwd": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_segment_max_bwd": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_softmax_ce_work_elems": (c_sz, [c_i64]),
    "yolat_softmax_ce": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p]),
    "yolat_adam_step": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_p]),
    "yolat_profile_enabled": (c_int, []),
    "yolat_profile_enable": (c_int, [c_int]),
    "yolat_profile_reset": (c_int, []),
    "yolat_profile_count": (c_int, []),
    "yolat_profile_get": (c_int, [c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_f), ctypes.POINTER(c_int),
                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "yolat_forward_eval_workspace_bytes": (c_sz, [ctypes.POINTER(ModelEval), c_i64, c_i64, c_i64]),
    "yolat_forward_eval": (c_int, [ctypes.POINTER(ModelEval), c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_i64,
                                   c_i64, c_p, c_i64, c_p, c_sz, c_p, c_p]),
    "yolat_forward_eval_primed": (c_int, [ctypes.POINTER(ModelEval), c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_i64,
                                          c_i64, c_i64, c_p, c_i64, c_p, c_sz, c_p, c_p]),
    "yolat_edge_uv_lin1_fwd": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p]),
    "yolat_nms_work_bytes": (c_sz, [c_i64]),
    "yolat_nms": (c_int, [c_p, c_p, c_i64, c_f, c_p, c_p, c_p, c_sz, c_p]),
    "yolat_f32_to_bf16": (c_int, [c_p, c_i64, c_p, c_p]),
    "yolat_edge_uv_lin1_fwd_h": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p]),
    "yolat_linear_fwd_h": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p,
                                   c_p, c_p]),
    "yolat_csr_mean_fwd_h": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64, c_p, c_i64, c_int, c_p]),
    "yolat_csr_mean_bwd_h": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_bn_relu_bwd_h": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_int, c_p, c_p, c_int,
                                    c_p, c_i64, c_p, c_p]),
    "yolat_linear_bwd_w_h": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_int, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64,
                                     c_p, c_int, c_p, c_p]),
    "yolat_linear_fwd_wt_h": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p]),
    "yolat_edge_uv_sums_h": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
    "yolat_split_bf16x3": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "yolat_linear_fwd_rows_x6": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p,
                                         c_p, c_p]),
    "yolat_node_uv_eval_x6": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64,
                                      c_p, c_i64, c_p, c_i64, c_p]),
    "yolat_fusion_pair_eval_x6": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p,
                                          c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "yolat_inv_degree": (c_int, [c_p, c_i64, c_p, c_p]),
    "yolat_bn_csr_work_elems": (c_sz, [c_i64, c_i64]),
    "yolat_bn_csr_bwd_stats": (c_int, [ctypes.POINTER(BnCsrGrad), c_i64, c_i64, c_p, c_p, c_int, c_p, c_p, c_p]),
    "yolat_linear_bwd_w_csr": (c_int, [ctypes.POINTER(BnCsrGrad), c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_int, c_p,
                                       c_i64, c_p, c_int, c_p, c_p]),
    "yolat_linear_fwd_wt_csr": (c_int, [ctypes.POINTER(BnCsrGrad), c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
    "yolat_bn_csr_l2_bwd_work_elems": (c_sz, []),
    "yolat_bn_csr_l2_bwd": (c_int, [ctypes.POINTER(BnCsrGrad), c_i64, c_p, c_i64, c_p, c_p, c_int, c_p, c_i64, c_p, c_i64,
                                    c_p, c_int, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yolat_bn_relu_bwd_apply": (c_int, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_int, c_p, c_p, c_i64,
                                        c_int, c_p]),
    "yolat_gemm_x6_packed_elems": (c_sz, [c_i64, c_i64]),
    "yolat_gemm_x6_pack": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "yolat_gemm_x6_pack_t": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p]),
    "yolat_edge_attr_dw_work_elems": (c_sz, [c_i64]),
    "yolat_edge_attr_dw": (c_int, [c_p, c_i64, c_int, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p]),
    "yolat_bn_apply_edge_sums_work_elems": (c_sz, [c_i64]),
    "yolat_bn_apply_edge_sums": (c_int, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_int, c_i64, c_p, c_p, c_p, c_p, c_int, c_p,
                                         c_p, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p]),
    "yolat_edge_uv_sums_v": (c_int, [c_p, c_i64, c_int, c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
    "yolat_gemm_x6_work_elems": (c_sz, [c_i64, c_i64, c_i64]),
    "yolat_gemm_x6": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_i64, c_p, c_i64, c_p, c_p]),
    "yolat_gemm_x6_stats": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p]),
    "yolat_split_bf16x3_packed_elems": (c_sz, [c_i64, c_i64]),
    "yolat_split_bf16x3_packed": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "yolat_linear_x6": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_i64, c_p, c_i64, c_p]),
    "yolat_linear_x6_pre": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_int, c_i64, c_p, c_i64, c_p]),
    "yolat_dropout_fwd": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_f, ctypes.c_uint64, c_p, c_p, c_i64, c_p]),
    "yolat_dropout_bwd": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_f, c_p, c_i64, c_p]),
    "yolat_proposals_build": (c_int, [c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p, c_i64, ctypes.c_double, c_p]),
    "yolat_proposals_count": (c_i64, [c_p]),
    "yolat_proposals_total": (c_i64, [c_p, c_int]),
    "yolat_proposals_get": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yolat_proposals_window_counts": (c_int, [c_p, c_p, c_p]),
    "yolat_proposals_free": (None, [c_p]),
    "yolat_proposals_assemble": (c_int, [c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_int,
                                         c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yolat_collate_pack": (c_int, [c_p, c_p, ctypes.POINTER(Span), c_i64, c_i64]),
    "yolat_item_csr_host": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                     c_p]),
    "yolat_collate_csr_pack": (c_int, [ctypes.POINTER(ItemCsr), c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yolat_collate_batch": (c_int, [c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p]),
    "yolat_loader_create": (c_p, [c_int]),
    "yolat_loader_submit": (c_int, [c_p, c_p, c_i64]),
    "yolat_loader_next": (c_int, [c_p, c_p, ctypes.POINTER(LoaderBatch)]),
    "yolat_loader_release": (c_int, [c_p, c_int, c_p]),
    "yolat_loader_destroy": (None, [c_p]),
    "yolat_forward_eval_csr": (c_int, [ctypes.POINTER(ModelEval), c_p, c_i64, ctypes.POINTER(GraphCsr), c_i64, c_i64,
                                        c_i64, c_p, c_i64, c_p, c_sz, c_p]),
    "yolat_forward_eval_bf16_csr": (c_int, [ctypes.POINTER(ModelEvalBf16), c_p, c_i64, ctypes.POINTER(GraphCsr), c_i64,
                                             c_i64, c_i64, c_p, c_i64, c_p, c_sz, c_p]),
    "yolat_edge_uv_mlp2_mean_eval_bf16": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p,
                                                   c_i64, c_p, c_i64, c_int, c_p]),
    "yolat_forward_eval_bf16_workspace_bytes": (c_sz, [ctypes.POINTER(ModelEvalBf16), c_i64, c_i64, c_i64]),
    "yolat_forward_eval_bf16": (c_int, [ctypes.POINTER(ModelEvalBf16), c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_i64,
                                        c_i64, c_i64, c_p, c_i64, c_p, c_sz, c_p, c_p]),
    "yolat_conv_local_pack_bytes": (c_sz, [c_i64]),
    "yolat_conv_local_pack": (c_int, [ctypes.POINTER(ModelEvalBf16), c_p, c_sz, c_p]),
    "yolat_conv_local_tune": (None, [c_int, c_int, c_int, c_p]),
    "yolat_conv_stack_local_bf16": (c_int, [ctypes.POINTER(ModelEvalBf16), c_p, c_p, c_i64, ctypes.POINTER(GraphCsr), c_i64,
                                             c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p]),
    "yolat_forward_eval_bf16_primed": (c_int, [ctypes.POINTER(ModelEvalBf16), c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, c_i64,
                                        c_i64, c_i64, c_p, c_i64, c_p, c_sz, c_p, c_p]),
    "yolat_train_step_workspace_bytes": (c_sz, [ctypes.POINTER(TrainModel), c_i64, c_i64, c_i64]),
    "yolat_train_step": (c_int, [ctypes.POINTER(TrainModel), c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p, ctypes.POINTER(GraphCsr),
                                 c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_sz, c_p, ctypes.POINTER(AdamArgs), c_int, c_p,
                                 c_p]),
    "yolat_predict_select_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "yolat_predict_select": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_i64,
                                     ctypes.POINTER(PredictTree), c_p, c_p, c_sz, c_p]),
    "yolat_predict_gather": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p]),
    "yolat_batch_locality_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "yolat_batch_locality": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_sz, c_p]),
    "yolat_conv_local_fits": (c_int, [ctypes.POINTER(Locality), c_i64]),
    "yolat_item_locality_host": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_i64, ctypes.POINTER(Locality)]),
    "yolat_conv_stack_local_bf16_coo": (c_int, [ctypes.POINTER(ModelEvalBf16), c_p, c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p,
                                                 c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_sz, c_p]),
    "yolat_forward_eval_bf16_loc": (c_int, [ctypes.POINTER(ModelEvalBf16), c_p, c_i64, c_p, c_i64, c_i64, c_p, c_p,
                                            ctypes.POINTER(GraphCsr), c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_sz, c_p,
                                            ctypes.POINTER(Locality), c_int, c_p]),
}


class YolatLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libyolat_hip.so not found at %s — the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` from the repo root." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("libyolat_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what=""):
    """Raise on a non-zero return code of a yolat_* entry point."""
    if rc != 0:
        msg = lib.yolat_strerror(int(rc))
        raise YolatLibraryError("%s failed (%d): %s" % (what or "yolat call", rc,
                                                      msg.decode() if msg else "?"))
