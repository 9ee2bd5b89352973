"""ctypes binding of libcer_hip.so (declared in include/cer_hip.h).

There is no CPU fallback: if the shared object is absent or a call fails, a
RuntimeError is raised.  ``load()`` only dlopens; compute entry points need a GPU.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint64,
                    c_void_p)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libcer_hip.so")

ACT_NONE, ACT_PRELU, ACT_LEAKY, ACT_RELU, ACT_GELU = 0, 1, 2, 3, 4
STORE_NONE, STORE_BF16, STORE_F16 = 0, 1, 2


class ConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in (
        "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "dil_h", "dil_w", "pad_t", "pad_l",
        "x_nchw", "res_stride", "Hr", "Wr", "act1", "act2")] + [("slope", c_float), ("split_k", c_int32),
                                                                ("tile", c_int32), ("x_ld", c_int32),
                                                                ("y_ld", c_int32), ("storage", c_int32),
                                                                ("x_s2d", c_int32), ("y_s2d", c_int32)]


_P = c_void_p


class ConvIO(Structure):
    """cer_conv_io: every pointer of one conv launch (see include/cer_hip.h)."""
    _fields_ = [(n, c_void_p) for n in (
        "x", "w", "x_hi", "x_lo", "w_hi", "w_lo", "in_scale", "in_shift", "bias", "alpha", "residual", "mask",
        "res_hi", "res_lo", "y", "aux", "stats", "y_hi", "y_lo", "s2", "t2", "y2_hi", "y2_lo", "bias9")]


_SIGNATURES = {
    # name: (restype, argtypes)
    "cer_last_error": (c_char_p, []),
    "cer_version": (c_int, []),
    "cer_conv_kpad": (c_int, [c_int, c_int, c_int]),
    "cer_conv_s2d_k_order": (c_int, [c_int, c_int, POINTER(c_int32)]),
    "cer_stem_conv3x3_stats_rows": (c_int, [c_int, c_int]),
    "cer_stem_conv3x3": (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P]),
    "cer_conv2d_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "cer_conv2d_stats_tiles": (c_int, [POINTER(ConvDesc), c_int]),
    "cer_conv2d_run": (c_int, [POINTER(ConvDesc), POINTER(ConvIO), _P, c_size_t, _P]),
    "cer_conv2d_b3_tile": (c_int, [POINTER(ConvDesc)]),
    "cer_conv2d_n16_tile": (c_int, [POINTER(ConvDesc)]),
    "cer_to_n16": (c_int, [_P, _P, _P, c_int, _P, c_size_t, c_int, _P]),
    "cer_from_n16": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "cer_split_bf16": (c_int, [_P, _P, _P, c_int, _P, _P, c_size_t, _P]),
    "cer_conv2d_fwd": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "cer_bn_finalize_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cer_bn_finalize": (c_int, [_P, c_int, c_int, c_double, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, c_size_t,
                                _P]),
    "cer_bn_apply_stats_tiles": (c_int, [c_int]),
    "cer_bn_apply_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, _P]),
    "cer_bn_apply_nhwc_b3": (c_int, [_P] * 14 + [c_int] * 7 + [_P]),
    "cer_bn_apply_nhwc_n16": (c_int, [_P] * 13 + [c_int] * 8 + [_P]),
    "cer_fold_bn_3x3": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "cer_pack_conv_weight": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cer_weight_norm_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "cer_weight_norm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "cer_weight_norm_fwd_packed": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cer_weight_norm_bwd_partials": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "cer_conv1d_wgrad_weight_norm_bwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P,
                                                 _P, c_size_t, _P]),
    "cer_conv_wgrad_workspace_bytes": (c_size_t, [ctypes.c_longlong, c_int, c_int, c_int]),
    "cer_conv1d_wgrad": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "cer_conv2d_wgrad": (c_int, [_P, _P, _P] + [c_int] * 12 + [_P, c_size_t, _P]),
    "cer_conv2d_wgrad_b3_workspace_bytes": (c_size_t, [c_int] * 7),
    "cer_conv2d_wgrad_b3": (c_int, [_P, _P, _P] + [c_int] * 12 + [_P, c_size_t, _P]),
    "cer_conv2d_wgrad_b3s": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 12 + [_P, c_size_t, _P]),
    "cer_prelu_fwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "cer_prelu_bwd": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, _P]),
    "cer_prelu_split": (c_int, [_P, _P, _P, _P, c_size_t, c_int, _P]),
    "cer_prelu_bwd_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, c_int, _P]),
    "cer_bn_rows_bwd_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "cer_bn_rows_bwd_add": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "cer_col_sum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cer_col_sum": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "cer_bn_bwd_sums": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "cer_act_mask_bwd": (c_int, [_P, _P, _P, _P, c_size_t, c_float, _P]),
    "cer_tblock_tail_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, c_float, _P]),
    "cer_bn_rows_fwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cer_bn_rows_fwd": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float,
                                c_float, _P, c_size_t, _P]),
    "cer_bn_rows_bwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t,
                                _P]),
    "cer_lfan_attn_fwd": (c_int, [POINTER(_P), _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cer_lfan_attn_bwd": (c_int, [POINTER(_P), _P, _P, POINTER(_P), c_int, c_int, c_int, c_int, _P]),
    "cer_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_float, _P]),
    "cer_layernorm_bwd": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "cer_cross_entropy": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "cer_dropout_mask": (c_int, [_P, c_size_t, c_float, c_uint64, c_uint64, _P]),
    "cer_copy_cols": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "cer_leaky_relu_fwd": (c_int, [_P, _P, c_size_t, c_float, _P]),
    "cer_softmax_gate_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "cer_softmax_gate_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "cer_logmel_num_frames": (c_int, [c_int, c_int]),
    "cer_logmel_fwd": (c_int, [_P, c_int, c_int, c_int, _P, c_float, _P, _P]),
    "cer_frame_examples": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cer_bert_embed_ln": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "cer_attention_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P]),
    "cer_attention_bwd": (c_int, [_P] * 11 + [c_int] * 5 + [_P] * 8 + [c_float, _P]),
    "cer_window_stitch": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "cer_eval_accumulate": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "cer_add_inplace": (c_int, [_P, _P, c_size_t, _P]),
    "cer_l2norm_rows": (c_int, [_P, _P, c_int, c_int, _P]),
    "cer_l2norm_rows_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "cer_maxpool2x2_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "cer_sgd_nesterov_flat": (c_int, [_P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_int, c_int, _P]),
    "cer_gather_rows": (c_int, [_P, _P, _P, c_int, c_int, c_uint64, _P]),
    "cer_frames_band_rows": (c_int, []),
    "cer_frames_transform": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int,
                                     c_int, c_float, c_float, _P, _P, _P]),
}

_lib = None


def exported_symbols():
    """Names that include/cer_hip.h declares (used by the CPU-side ABI test)."""
    return sorted(_SIGNATURES)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m feature_vs_text_compound_emotion_amd.build` "
            "(there is no CPU fallback for the HIP hot path)")
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Import torch FIRST so that
    # our NEEDED libamdhip64.so.7 resolves to that already-loaded runtime; loading ours first
    # would put a second HIP runtime (from /opt/rocm) in the process, which then sees no device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cer_last_error()
        raise RuntimeError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
