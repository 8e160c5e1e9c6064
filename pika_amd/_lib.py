"""ctypes binding of the C ABI declared in include/*.h (libpika_amd.so).

Fails loudly: a missing library is an ImportError-grade RuntimeError, never a fallback.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libpika_amd.so")
ABI_VERSION = 22

_vp, _i, _sz, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/*.h one to one
SIGNATURES = {
    "pika_amd_abi_version": (_i, []),
    "pika_rnnt_workspace_bytes": (_sz, [_i, _i, _i]),
    "pika_rnnt_loss_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pika_rnnt_loss_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pika_rnnt_loss_dense_grads": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "pika_rnnt_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pika_rnnt_fused_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pika_rnnt_fused_forward_partials": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pika_rnnt_fused_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _ll, _vp]),
    "pika_rnnt_dlogits_compact_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _ll, ctypes.c_float, _vp, _vp]),
    "pika_rnnt_fused_forward_gathered": (_i, [_vp, _ll, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp,
                                               _vp, _vp]),
    "pika_rnnt_dlogits_compact_bf16_f16in": (_i, [_vp, _ll, _vp, _vp, _i, _i, _i, _i, _i, _vp, _ll, ctypes.c_float, _vp, _vp, _vp,
                                                   _i, _vp]),
    "pika_rnnt_export_lattice": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    # include/pika_bmuf.h
    "pika_bmuf_delta": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "pika_bmuf_nan_flag": (_i, [_vp, _sz, _vp, _vp]),
    "pika_bmuf_update": (_i, [_vp, _vp, _vp, _vp, _sz, ctypes.c_float, ctypes.c_float,
                              ctypes.c_float, _vp, _vp]),
    "pika_bmuf_adam_moments": (_i, [_vp, _vp, _vp, _vp, _sz, ctypes.c_float] + [ctypes.c_float] * 6 + [_vp, _vp]),
    # include/pika_feat.h
    "pika_cmvn_apply": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "pika_specaug_apply": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    # include/pika_gemm.h
    "pika_gemm_nt": (_i, [_vp, _vp, _vp, _ll, _ll, _ll, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "pika_gemm_nt_ws": (_i, [_vp, _vp, _vp, _ll, _ll, _ll, _i, _i, _i, _i, _i, _vp, _i, _vp, ctypes.c_size_t, _vp]),
    "pika_gemm_bf16_nt": (_i, [_vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _vp, _vp]),
    "pika_gemm_bf16_nt_lse": (_i, [_vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "pika_gemm_bf16_nt_lse_f16": (_i, [_vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "pika_gemm_bf16_epilogue": (_i, [_vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _vp, _i, _i, ctypes.c_float,
                                     ctypes.c_uint, _vp, _ll, ctypes.c_float, _vp]),
    "pika_dropout_keep_mask": (_i, [_vp, _i, _i, ctypes.c_float, ctypes.c_uint, _vp]),
    "pika_set_dropout_salt": (_i, [_vp]),
    "pika_gemm_set_min_tiles": (_i, [_i]),
    "pika_gemm_bf16_dropout_residual": (_i, [_vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _vp, ctypes.c_float,
                                             ctypes.c_uint, _vp, _ll, _vp]),
    "pika_dropout_mask_cast_bf16": (_i, [_vp, _ll, _i, _i, ctypes.c_float, ctypes.c_uint, _vp, _ll, _vp]),
    "pika_gemm_bf16_ex": (_i, [_vp, _vp]),
    # include/pika_attn.h
    "pika_attention_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, ctypes.c_float, ctypes.c_uint, _vp]),
    "pika_attention_fwd_two_term": (_i, [_vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll,
                                         ctypes.c_float, ctypes.c_uint, _vp]),
    "pika_attention_infer_f16x2": (_i, [_vp, _vp, _vp, _ll, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _vp]),
    "pika_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll,
                                ctypes.c_float, ctypes.c_uint, _vp]),
    "pika_attention_keep_mask": (_i, [_vp, _i, _i, ctypes.c_float, ctypes.c_uint, _vp]),
    "pika_attention_mask_bits": (_i, [_vp, _i, _i, _vp, _vp]),
    # include/pika_ops.h
    "pika_transpose_cast": (_i, [_vp, _i, _i, _vp, _ll, _i, _vp]),
    "pika_colsum": (_i, [_vp, _ll, _i, _i, _vp, _vp, _vp]),
    "pika_colsum_bf16": (_i, [_vp, _ll, _i, _i, _vp, _vp, _vp]),
    "pika_colsum_partial_floats": (_ll, [_i, _i]),
    "pika_weight_taps_transposed_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "pika_col2im": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pika_split_bf16_terms": (_i, [_vp, _i, _i, _i, _ll, _ll, _i, _i, _i, _i, _vp, _vp]),
    "pika_split_bf16_terms2": (_i, [_vp, _vp, _vp]),
    # include/pika_las.h
    "pika_lstm_cell": (_i, [_vp, _ll, _vp, _vp, _vp, _ll, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp]),
    "pika_las_attention_work_floats": (ctypes.c_size_t, [_i, _i, _i]),
    "pika_las_mlp_attention_by_utterance": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _i, _i, _i, _i,
                                                 _vp, _vp, _vp, _vp]),
    "pika_las_step_advance": (_i, [_vp, _vp, _vp, _i, _vp]),
    "pika_las_embed_rows": (_i, [_vp, _vp, _vp, _vp, _ll, _vp, _i, _i, _vp, _vp]),
    "pika_las_fork_rows": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "pika_blstm_packed_bytes": (_ll, [_i, _i]),
    "pika_blstm_work_bytes": (_ll, [_i, _i, _i, _i]),
    "pika_blstm_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "pika_blstm_layer": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _vp]),
    "pika_blstm_status": (_i, [_vp, _vp, _vp]),
    "pika_lstm_train_packed_bytes": (_ll, [_i]),
    "pika_lstm_train_fwd_work_bytes": (_ll, [_i, _i, _i]),
    "pika_lstm_train_bwd_work_bytes": (_ll, [_i, _i, _i]),
    "pika_lstm_train_pack": (_i, [_vp, _i, _vp, _vp]),
    "pika_lstm_train_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "pika_lstm_train_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _vp]),
    "pika_lstm_train_status": (_i, [_vp, _vp, _vp]),
    # include/pika_joint.h
    "pika_joint_gate_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pika_joint_gate_bwd": (_i, [_vp, _i] + [_vp] * 8 + [_i, _i, _i, _i, _vp]),
    "pika_log_softmax_rows": (_i, [_vp, _ll, _i, _ll, ctypes.c_float, _vp]),
    "pika_log_softmax_bwd_rows": (_i, [_vp, _vp, _ll, _i, _ll, ctypes.c_float, _vp]),
    "pika_log_softmax_bwd_rows_bf16": (_i, [_vp, _vp, _vp, _ll, _i, _ll, _ll, ctypes.c_float, _vp]),
    "pika_mbr_risk_grad_rows": (_i, [_vp, _vp, _vp, _ll, _i, _ll, ctypes.c_float, _vp]),
    "pika_edit_distances": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    # include/pika_norm.h
    "pika_layer_norm_fwd": (_i, [_vp, _ll, _i, _vp, _vp, ctypes.c_float, _vp, _i, _vp, _vp, _vp, _vp]),
    "pika_layer_norm_bwd": (_i, [_vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pika_layer_norm_bwd_partial_floats": (_ll, [_ll, _i]),
    "pika_bn_stats": (_i, [_vp, _ll, _i, _vp, _vp, _vp]),
    "pika_bn_apply": (_i, [_vp, _ll, _i, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp,
                           _vp, _i, _vp, _vp, _vp]),
    "pika_bn_backward": (_i, [_vp, _i, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    # include/pika_decode.h
    "pika_incremental_attention": (_i, [_vp, _vp, _vp, _vp, _ll, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pika_beam_advance": (_i, [_vp, ctypes.c_float, _i, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp,
                               _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i,
                               _i, _vp]),
    "pika_fst_advance": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, ctypes.c_double,
                              ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "pika_fst_states_per_slot": (_i, []),
    "pika_beam_backtrack": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    # include/pika_decode_step.h
    "pika_dpack_bytes": (_sz, [_i, _i, _i]),
    "pika_dpack_weight": (_i, [_vp, _ll, _i, _i, _i, _i, _vp, _vp]),
    "pika_dgemm": (_i, [_vp, _vp]),
    "pika_dstep_prep": (_i, [_vp, _vp]),
    "pika_dstep_prep_lstm": (_i, [_vp, _vp]),
    "pika_dstep_lstm_cell": (_i, [_vp, _ll, _vp, _ll, _i, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "pika_dstep_attention": (_i, [_vp, _ll, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pika_dfc2_splits": (_i, [_i]),
    "pika_dfc2_cols_per_split": (_i, []),
    "pika_dfc2_logits": (_i, [_vp, _ll, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _vp, _ll, _vp]),
    "pika_beam_advance_logits_lds": (ctypes.c_size_t, [_i, _i, _i]),
    "pika_beam_advance_logits": (_i, [_vp, _vp, _vp, _ll, _i, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                      _vp, _vp, _vp, _vp]),
    # include/pika_optim.h
    "pika_multi_absmax": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "pika_multi_scale_by_clip": (_i, [_vp, _vp, _vp, _vp, _i, _vp, ctypes.c_float, _vp]),
    "pika_multi_sgd_nesterov": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, ctypes.c_float, ctypes.c_float, _i, _vp]),
    # include/pika_audio.h
    "pika_audio_sumsq": (_i, [_vp, _ll, _vp, _vp]),
    "pika_audio_axpby": (_i, [_vp, _vp, _ll, ctypes.c_float, ctypes.c_float, _vp]),
    "pika_audio_convolve_same": (_i, [_vp, _ll, _vp, _i, _vp, _vp]),
    "pika_audio_perturb": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _vp, _vp, _vp]),
    "pika_fbank": (_i, [_vp, _vp, _vp, _i, _ll, _i, _i, _i, ctypes.c_float, ctypes.c_float,
                        ctypes.c_ulonglong, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pika_splice_pad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
}

_lib = None


def lib():
    """Load libpika_amd.so once (torch must be imported first so that both share one HIP runtime)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "pika_amd: %s is missing -- run `python -m pika_amd.build` (or "
                "__graft_entry__.build()); there is no CPU/PyTorch fallback." % LIB_PATH)
        import torch  # noqa: F401  (loads libamdhip64 with the SONAME our library links to)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        got = handle.pika_amd_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError("pika_amd: ABI version %d != expected %d; rebuild" % (got, ABI_VERSION))
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        kind = "bad argument" if rc < 0 else "hipError_t"
        raise RuntimeError("pika_amd: %s failed (%s %d)" % (what, kind, rc))
