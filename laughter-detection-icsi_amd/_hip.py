"""ctypes binding of liblad_hip.so (C ABI declared in include/lad_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails this module raises.
PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream handle).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LAD_HIP_LIB") or os.path.join(_HERE, "liblad_hip.so")   # (LAD_HIP_LIB: A/B builds, tools/ only)

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_float = ctypes.c_float
c_double = ctypes.c_double
c_i32 = ctypes.c_int32
c_float_p = ctypes.POINTER(ctypes.c_float)


class LadHipError(RuntimeError):
    pass


class FbankCfg(ctypes.Structure):
    _fields_ = [("n_fft", ctypes.c_int32), ("frame_len", ctypes.c_int32), ("hop", ctypes.c_int32),
                ("n_mels", ctypes.c_int32), ("n_mfcc", ctypes.c_int32), ("pad_mode", ctypes.c_int32),
                ("log_mode", ctypes.c_int32), ("remove_dc", ctypes.c_int32), ("preemph", ctypes.c_float),
                ("log_floor", ctypes.c_float)]


# name -> (restype, argtypes); every symbol declared in include/lad_hip.h must be listed here
# (tests/test_cabi.py cross-checks this table against the header and the built library).
SIGNATURES = {
    "lad_version": (c_int, []),
    "lad_last_error": (ctypes.c_char_p, []),
    "lad_fbank_plan_create": (c_int, [ctypes.POINTER(FbankCfg), c_void_p, c_void_p, c_void_p,
                                      ctypes.POINTER(c_void_p)]),
    "lad_fbank_plan_destroy": (c_int, [c_void_p]),
    "lad_fbank_plan_set_kernel": (c_int, [c_void_p, c_int]),
    "lad_fbank_plan_has_fast_kernel": (c_int, [c_void_p]),
    "lad_fbank_num_frames": (c_i64, [c_void_p, c_i64]),
    "lad_fbank_forward": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p]),
    "lad_fbank_forward_long": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "lad_gather_segments": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_float, c_void_p, c_void_p]),
    "lad_assemble_windows": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_act_rows": (c_i64, [c_i64, c_i32, c_i32]),
    "lad_conv_packed_weight_floats": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "lad_conv_pack_weights": (c_int, [c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "lad_conv_pack_weights_multi": (c_int, [c_void_p, c_i32, c_void_p]),
    "lad_conv_num_tiles": (c_i64, [c_i64, c_i32, c_i32]),
    "lad_conv_fwd": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_b3_packed_weight_bytes": (c_i64, []),
    "lad_conv_b3_pack_weights": (c_int, [c_void_p, c_i32, c_void_p, c_void_p]),
    "lad_conv_b3_fwd_f32": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_conv_b3c_packed_weight_bytes": (c_i64, [c_i32]),
    "lad_conv_b3c_pack_weights": (c_int, [c_void_p, c_i32, c_void_p, c_i32, c_void_p]),
    "lad_conv_b3c_fwd_f32": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_b3c_fwd_f32_bnrelu": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_b3c_dgrad_bnstat": (c_int, [c_void_p] * 7 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_h2_packed_weight_bytes": (c_i64, [c_i32]),
    "lad_conv_h2_pack_weights_multi": (c_int, [c_void_p, c_i32, c_i32, c_void_p]),
    "lad_conv_wgrad_h2": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_h2": (c_int, [c_void_p] * 11 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_wgrad_h2_bnbwd": (c_int, [c_void_p] * 11 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_wgrad_b3c_workspace_floats": (c_i64, [c_i32]),
    "lad_conv_wgrad_b3c": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_b3_fwd_f32_gated": (c_int, [c_void_p] * 7 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_conv_b3_fwd_f32_bnrelu": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_conv_wgrad_b3_bnrelu": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_wgrad_defer_begin": (c_int, []),
    "lad_wgrad_defer_flush": (c_int, [c_void_p]),
    "lad_conv_b3_dgrad_bnstat": (c_int, [c_void_p] * 9 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_conv_wgrad_b3": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_fwd": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_bn_fold": (c_int, [c_void_p] * 5 + [c_i32, c_void_p, c_void_p, c_void_p]),
    "lad_conv_fwd_eval": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_fwd_eval": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_upsample2": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_dgrad": (c_int, [c_void_p] * 3 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_wgrad_workspace_floats": (c_i64, [c_i32, c_i32, c_i32]),
    "lad_conv_s2_fwd_fused": (c_int, [c_void_p] * 8 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_dgrad_partials": (c_i64, [c_i64, c_i32, c_i32]),
    "lad_conv_s2b3_packed_weight_bytes": (c_i64, []),
    "lad_conv_s2b3_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "lad_conv_s2b3_fwd": (c_int, [c_void_p] * 7 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_conv_s2b3_dgrad_packed_weight_bytes": (c_i64, []),
    "lad_conv_s2b3_dgrad_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "lad_conv_s2b3_pack_weights_pair": (c_int, [c_void_p] * 5),
    "lad_conv_s2b3_dgrad_partials": (c_i64, [c_i64, c_i32, c_i32]),
    "lad_conv_s2b3_dgrad": (c_int, [c_void_p] * 8 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_dgrad_fused_bnstat": (c_int, [c_void_p] * 9 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_dgrad_fused": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_wgrad_fused_workspace_floats": (c_i64, [c_i32, c_i32]),
    "lad_conv_s2_wgrad_fused": (c_int, [c_void_p] * 7 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_s2_wgrad": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_wgrad_workspace_floats": (c_i64, [c_i32, c_i32, c_i32]),
    "lad_conv_wgrad": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_stem_fwd": (c_int, [c_void_p] * 4 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_stem_fwd_eval": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i64, c_i64, c_void_p]),
    "lad_stem_wgrad_workspace_floats": (c_i64, []),
    "lad_stem_wgrad": (c_int, [c_void_p] * 4 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_stem_wgrad_bn": (c_int, [c_void_p] * 8 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_stem_bn_bwd_groups": (c_i64, [c_i64, c_i32, c_i32]),
    "lad_stem_bn_bwd_sums": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_bn_finalize": (c_int, [c_void_p, c_i64, c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                c_void_p, c_void_p]),
    "lad_stem_moments_workspace_doubles": (c_i64, []),
    "lad_stem_moments_doubles": (c_i64, []),
    "lad_stem_bn_stats": (c_int, [c_void_p] * 6 + [c_float] + [c_void_p] * 3 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_stem_bwd_onepass_workspace_floats": (c_i64, []),
    "lad_stem_bwd_onepass": (c_int, [c_void_p] * 10 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_bn_finalize_pair": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i64] + [c_void_p] * 10 + [c_float, c_void_p]),
    "lad_bn_act": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_bn_bwd_workspace_floats": (c_i64, [c_i32]),
    "lad_bn_act_bits": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_bn_bwd_bits": (c_int, [c_void_p] * 11 + [c_i64, c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_bn_bwd": (c_int, [c_void_p] * 17 + [c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_conv_fwd_bnstat": (c_int, [c_void_p] * 8 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_pool_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_pool_bwd": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_head_workspace_floats": (c_i64, [c_i64, c_i32]),
    "lad_head_fwd_train": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lad_head_fwd_train_rng": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_float, ctypes.c_uint64, c_void_p, c_void_p, c_i32,
                                       c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lad_head_fwd_eval": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p]),
    "lad_bce_metrics": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "lad_head_bwd": (c_int, [c_void_p] * 7 + [c_i64, c_i32] + [c_void_p] * 6),
    "lad_f16_packed_weight_halfs": (c_i64, [c_i32, c_i32, c_i32]),
    "lad_f16_pack_weights": (c_int, [c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "lad_f16_stem_fwd": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i64, c_i64, c_void_p]),
    "lad_f16_conv_fwd": (c_int, [c_void_p] * 6 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_f16_block_fwd": (c_int, [c_void_p] * 8 + [c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_f16_conv_s2_fwd_sc": (c_int, [c_void_p] * 9 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_f16_conv_s2_fwd_mapped_sc": (c_int, [c_void_p] * 9 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32, c_i64, c_i64, c_i32,
                                              c_i32, c_i32, c_i32, c_void_p]),
    "lad_f16_conv_s2_fwd": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_f16_conv_s2_fwd_windows": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_f16_conv_s2_fwd_mapped": (c_int, [c_void_p] * 5 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32, c_i64, c_i64, c_i32,
                                                          c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "lad_f16_block_fwd_stem_rows": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64] + [c_void_p] * 10 + [c_i64, c_i32, c_i32, c_void_p]),
    "lad_f16_conv_s2_strips_fwd": (c_int, [c_void_p] * 9 + [c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i32, c_void_p]),
    "lad_f16_tail_fwd": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32, c_i64, c_i64, c_void_p, c_void_p, c_i32,
                                 c_void_p, c_void_p]),
    "lad_f16_pool_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p]),
    "lad_grad_sumsq_partials": (c_i32, []),
    "lad_grad_sumsq": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "lad_grad_accumulate": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_void_p]),
    "lad_adam_step": (c_int, [c_void_p] * 4 + [c_i64, c_void_p] + [c_double] * 6 + [c_i64, c_void_p, c_i32, c_void_p, c_void_p]),
}

_lib = None


def lib():
    """Load liblad_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LadHipError(
                f"{LIB_PATH} not found: build it with `python laughter-detection-icsi_amd/build.py` "
                "(there is no CPU fallback for the HIP hot path)")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


LAD_ERR_INVALID = -1   # include/lad_hip.h: bad argument / unsupported shape (nothing was launched)
LAD_NOT_COVERED = 1    # ... a try-and-fall-back entry point does not cover the geometry (nothing launched, no error string)


def check(rc, what=""):
    if rc != 0:
        msg = lib().lad_last_error().decode("utf-8", "replace")
        raise LadHipError(f"{what or 'liblad_hip'} failed ({rc}): {msg}")


def require_cuda(t, name="tensor", dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise LadHipError(f"{name} must be a GPU tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise LadHipError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise LadHipError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def stream_handle(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
