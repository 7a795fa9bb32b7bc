"""ctypes binding of libxmlhip.so (C ABI declared in include/xmlhip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the product path raises.
The library is built in-tree by `tvretrieval_amd/csrc/build.sh` (called from `__graft_entry__.build()`).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# XMLHIP_LIB: measurement tools point this at csrc/libxmlhip_dbg.so (the -DXML_DEBUG_VARIANTS build with the xml_debug_*
# experiment switches); the product always loads the in-tree library
LIB_PATH = os.environ.get("XMLHIP_LIB") or os.path.join(_HERE, "csrc", "libxmlhip.so")

XML_F32 = 0
XML_BF16 = 1
XML_F16 = 2       # IEEE half rows: the exact-rank FILTER operands of K6
XML_F16S = 3      # split f16 (hi + lo halves, 4 bytes per element): f32-grade values on the 16-bit MFMA pipe

ABI_VERSION = 6


class XmlHipError(RuntimeError):
    pass


class ConvseDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in
                ("nq", "nv", "kpairs", "lpad", "l_ref", "hidden", "n_mod", "merged", "ksize", "softmax", "dt")]


# name -> (restype, argtypes)   -- must list every symbol of include/xmlhip.h (tests/test_capi.py checks)
SIGNATURES = {
    "xml_abi_version": (c_int, []),
    "xml_build_arch": (c_char_p, []),
    "xml_status_string": (c_char_p, [c_int]),
    "xml_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "xml_convert": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "xml_linear_ln_relu_pos_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "xml_linear_ln_relu_pos": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                       c_void_p]),
    "xml_attention_block_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "xml_attention_block": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "xml_pack_plan": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xml_linear_ln_relu_pos_packed_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "xml_linear_ln_relu_pos_packed": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                              c_void_p, c_size_t, c_void_p]),
    "xml_attention_block_varlen_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "xml_attention_block_varlen": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                           c_void_p]),
    "xml_modular_pool_varlen": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                        c_void_p]),
    "xml_cross_attention_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "xml_cross_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_size_t, c_void_p]),
    "xml_modular_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    "xml_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "xml_linear_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "xml_l2norm_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "xml_l2norm_rows_eps": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "xml_q2c_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_void_p]),
    "xml_q2c_scores_fused": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "xml_q2c_tiled_ok": (c_int, [c_int, c_int, c_int]),
    "xml_q2c_tiled_bytes": (c_int64, [c_int64, c_int, c_int]),
    "xml_q2c_tile_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "xml_q2c_tile_rows_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "xml_q2c_scores_packed": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "xml_q2c_scores_tiled": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "xml_topk_rows_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "xml_topk_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                              c_void_p, c_size_t, c_void_p]),
    # ---- exact-rank mode (exact.hip, convse.hip) ----
    "xml_round_bf16_rows_err": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "xml_split_f16_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "xml_unsplit_f16_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "xml_pack_weights_f16s_bytes": (c_size_t, [c_int, c_int]),
    "xml_pack_weights_f16s": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "xml_linear_f16s_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "xml_linear_f16s": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_size_t,
                                c_void_p]),
    "xml_convse_rerank_ex": (c_int, [ctypes.POINTER(ConvseDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "xml_moment_topk_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "xml_convse_rerank_f16s": (c_int, [ctypes.POINTER(ConvseDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "xml_select_ge_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "xml_q2c_rescore_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "xml_q2c_rescore": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "xml_exact_certificate": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_int,
                                      c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "xml_convse_rerank_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvseDesc)]),
    "xml_convse_rerank": (c_int, [ctypes.POINTER(ConvseDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "xml_moment_topk": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_void_p]),
    "xml_nms_vcmr_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_double, c_int, c_int, c_void_p,
                                  c_void_p]),
    "xml_nms_svmr_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.c_double, c_int, c_int, c_void_p,
                                  c_void_p]),
    "xml_nms_vcmr_batched_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, ctypes.c_double,
                                          c_int, c_int, c_void_p, c_int64, c_void_p, c_int]),
    "xml_nms_svmr_batched_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, ctypes.c_double, c_int,
                                          c_int, c_void_p, c_int64, c_void_p, c_int]),
    "xml_attention_core": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "xml_conv1d_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "xml_ingest_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                                c_void_p]),
    "xml_moments_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int,
                                   c_float, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "xml_add_layernorm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                  c_void_p]),
    # ---- multi-GPU collectives (collectives.hip) ----
    "xml_rccl_available": (c_int, []),
    "xml_rccl_unique_id": (c_int, [c_void_p]),
    "xml_rccl_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "xml_rccl_comm_destroy": (c_int, [c_void_p]),
    "xml_rccl_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "xml_rccl_allreduce_avg_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "xml_rccl_allgather_topk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "xml_rccl_allgather_topk": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "xml_rccl_topk_by_owner_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "xml_rccl_topk_by_owner": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "xml_merge_shard_topk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "xml_merge_shard_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    # ---- training step (train.hip) ----
    "xml_transpose_batched": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "xml_colsum": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "xml_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "xml_add_inplace": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "xml_layernorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "xml_gemm_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int,
                                 c_void_p]),
    "xml_split_heads": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                c_void_p]),
    "xml_merge_heads": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_int,
                                c_void_p]),
    "xml_attn_softmax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "xml_q2c_tile_rows_l2norm_ok": (c_int, [c_int, c_int]),
    "xml_q2c_tile_rows_l2norm": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "xml_gemm_tn_supported": (c_int, [c_int64, c_int, c_int, c_int]),
    "xml_gemm_tn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "xml_attention_train_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "xml_attention_train_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int64, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint64, c_void_p, c_int, c_void_p]),
    "xml_attention_train_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                        c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int,
                                        c_float, ctypes.c_uint64, c_void_p, c_int, c_void_p]),
    "xml_modular_pool_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                     c_int, c_int, c_void_p]),
    "xml_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "xml_q2c_scores_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_int, c_int,
                                   c_int, c_int, c_int, c_void_p]),
    "xml_pair_sim": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "xml_pair_sim_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                 c_void_p]),
    "xml_span_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                              c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xml_rank_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p]),
    "xml_dropout": (c_int, [c_void_p, c_void_p, c_int64, c_float, ctypes.c_uint64, c_void_p, c_int, c_void_p]),
    "xml_q2c_scores_l2norm_bwd_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "xml_q2c_scores_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float,
                                          c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int,
                                          c_void_p]),
    "xml_q2c_scores_l2norm_bwd_multi": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                                c_float, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                                c_void_p, c_int64, c_int, c_void_p]),
    "xml_q2c_scores_arg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_void_p]),
    "xml_loss_combine": (c_int, [c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "xml_loss_combine_bwd": (c_int, [c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "xml_transpose_segments": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "xml_add_layernorm_drop": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                       c_float, ctypes.c_uint64, c_float, ctypes.c_uint64, c_void_p, c_void_p]),
    "xml_layernorm_bwd_drop": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_int, c_int, c_float, ctypes.c_uint64, c_float,
                                       ctypes.c_uint64, c_void_p, c_void_p]),
    "xml_layernorm_bwd_partials_bytes": (ctypes.c_size_t, [c_int64, c_int]),
    "xml_layernorm_bwd_drop_ws": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_int, c_int, c_float, ctypes.c_uint64, c_float,
                                          ctypes.c_uint64, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "xml_clip_grad_norm": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "xml_bert_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                   c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def bind(lib):
    """Attach restype / argtypes of every header symbol to a loaded library; check the ABI version."""
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise XmlHipError("libxmlhip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    if lib.xml_abi_version() != ABI_VERSION:
        raise XmlHipError("libxmlhip.so ABI %d != binding ABI %d" % (lib.xml_abi_version(), ABI_VERSION))
    return lib


def load():
    """Load libxmlhip.so (once).  Raises XmlHipError if it is missing: there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise XmlHipError(
            "libxmlhip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(tvretrieval_amd/csrc/build.sh). The HIP extension is mandatory; there is no fallback." % LIB_PATH)
    # PyTorch ships its own libamdhip64.so; it must be the first HIP runtime mapped into the process, otherwise
    # libxmlhip.so binds /opt/rocm's copy and the two runtimes do not share streams / allocations (launches fail).
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise XmlHipError("failed to load %s: %s" % (LIB_PATH, e))
    _lib = bind(lib)
    return lib


def check(status, what):
    if status != 0:
        msg = load().xml_status_string(status).decode()
        raise XmlHipError("%s failed: %s (%d)" % (what, msg, status))
