"""ctypes binding of the C ABI declared in include/vcg.h (libvcg_hip.so, gfx950).

The product path fails loudly here: no CPU fallback, no oracle import.  ``load()`` raises
RuntimeError when the shared library is missing or does not load; ``require_gpu()`` raises when no
HIP device is visible.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_DIR, "libvcg_hip.so")

ACT_NONE, ACT_LRELU, ACT_PRELU, ACT_TANH = 0, 1, 2, 3
NORM_BATCH, NORM_INSTANCE = 0, 1
LOSS_MSE, LOSS_MAE = 0, 1
HEAD_NONE, HEAD_SIGMOID, HEAD_LOGSIGM, HEAD_TANH, HEAD_BILOG = 0, 1, 2, 3, 4
HEAD_KINDS = {None: 0, "none": 0, "sigmoid": 1, "log-sigm": 2, "tanh": 3, "bi-log": 4}


class ConvDesc(ctypes.Structure):
    _fields_ = [(k, c_int32) for k in ("n", "cin", "h", "w", "cout", "oh", "ow", "kh", "kw", "stride",
                                       "pad_top", "pad_left")]


class Epilogue(ctypes.Structure):
    _fields_ = [("bias", c_void_p), ("act", c_int32), ("act_alpha", c_float), ("prelu_alpha", c_void_p),
                ("residual", c_void_p)]


FINAL9X9_WFRAG_BYTES = (4 * 9 * 4 * 64 + 4) * 16      # VCG_FINAL9X9_WFRAG_BYTES
FIRST9X9_WFRAG_BYTES = 27 * 64 * 2 * 16                # VCG_FIRST9X9_WFRAG_BYTES


class EpilogueBf16(ctypes.Structure):
    _fields_ = [("scale", c_void_p), ("shift", c_void_p), ("act", c_int32), ("act_alpha", c_float),
                ("prelu_alpha", c_void_p), ("residual", c_void_p), ("stats", c_void_p), ("stats_mode", c_int32)]


STATS_NONE, STATS_BATCH, STATS_INSTANCE = 0, 1, 2
E_UNSUPPORTED = -3            # VCG_E_UNSUPPORTED

# name -> (restype, argtypes); every symbol include/vcg.h declares
_P = c_void_p
_D = POINTER(ConvDesc)
_E = POINTER(Epilogue)
_EB = POINTER(EpilogueBf16)
SIGNATURES = {
    "vcg_version": (c_char_p, []),
    "vcg_error_string": (c_char_p, [c_int]),
    "vcg_conv2d_fwd": (c_int, [_D, _P, _P, _P, _E, _P]),
    "vcg_conv2d_dgrad": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "vcg_conv2d_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv2d_wgrad": (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_conv_transpose2d_fwd": (c_int, [_D, _P, _P, _P, _E, _P]),
    "vcg_conv_transpose2d_dgrad": (c_int, [_D, _P, _P, _P, _P, _P]),
    "vcg_conv_transpose2d_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv_transpose2d_wgrad": (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_kernel_transpose": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "vcg_norm_stats_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vcg_norm_stats": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "vcg_norm_finalize": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_float, c_int, _P]),
    "vcg_sum_records": (c_int, [_P, c_int, c_int, c_float, _P, _P]),
    "vcg_bn_fold": (c_int, [_P, _P, _P, _P, _P, c_int, c_float, _P, _P, _P]),
    "vcg_norm_finalize_partials": (c_int, [_P, c_int, c_int, c_int, ctypes.c_double, _P, _P, c_float, _P, _P, _P, _P, _P, _P, c_float, c_int, _P]),
    "vcg_norm_finalize_partials_shifted": (c_int, [_P, c_int, c_int, c_int, ctypes.c_double, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, c_float, c_int, _P]),
    "vcg_conv2d_stats_records": (c_int, [_D, c_int]),
    "vcg_conv2d_fwd_stats": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "vcg_norm_act_fwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P]),
    "vcg_norm_act_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vcg_norm_act_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, _P, c_int,
                                 _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_act_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vcg_act_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_channel_sum_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vcg_channel_sum": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "vcg_dense_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vcg_dense_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "vcg_dense_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vcg_mean_reduce_workspace_bytes": (c_size_t, [c_size_t]),
    "vcg_mean_reduce": (c_int, [_P, c_size_t, _P, _P, c_size_t, _P]),
    "vcg_pixel_loss": (c_int, [_P, _P, c_size_t, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "vcg_dilate2d": (c_int, [_P, _P, c_size_t, c_int, c_int, c_int, _P]),
    "vcg_resize2d": (c_int, [_P, _P, c_size_t, c_int, c_int, c_int, c_int, _P]),
    "vcg_crop2d": (c_int, [_P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vcg_pad2d": (c_int, [_P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vcg_copy_channels": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_size_t, _P]),
    "vcg_dropout_fwd": (c_int, [_P, _P, _P, c_size_t, c_float, ctypes.c_uint64, _P, _P]),
    "vcg_dropout_bwd": (c_int, [_P, _P, _P, c_size_t, c_float, _P]),
    "vcg_counter_inc": (c_int, [_P, _P]),
    "vcg_fill": (c_int, [_P, c_size_t, c_float, _P]),
    "vcg_axpby": (c_int, [_P, _P, c_size_t, c_float, c_float, _P]),
    "vcg_adam_keras_multi": (c_int, [_P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_float, _P]),
    "vcg_adam_keras_multi_dev": (c_int, [_P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_float, _P, _P]),
    "vcg_sigmoid_gate_fwd": (c_int, [_P, _P, _P, c_size_t, _P]),
    "vcg_sigmoid_gate_bwd": (c_int, [_P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_atanh_scale": (c_int, [_P, _P, c_size_t, c_float, _P]),
    "vcg_head_act_fwd": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "vcg_head_act_bwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "vcg_gan_loss": (c_int, [_P, _P, c_float, c_int, _P, _P, c_size_t, c_float, _P, c_size_t, c_float, _P]),
    "vcg_frames_u8_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vcg_nchw_to_frames_u8": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vcg_nhwc_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vcg_nchw_to_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vcg_maxpool2x2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vcg_maxpool2x2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    # bf16-storage path
    "vcg_pack_conv_kernel_bf16": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vcg_f32_nchw_to_bf16_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vcg_bf16_nhwc_to_f32_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vcg_conv2d_bf16_fwd": (c_int, [_D, _P, _P, _P, _EB, _P]),
    "vcg_conv2d_bf16_stats_records": (c_int, [_D, c_int32]),
    "vcg_conv_transpose2d_bf16_fwd": (c_int, [_D, _P, _P, _P, _EB, _P]),
    "vcg_norm_stats_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vcg_norm_stats_bf16": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "vcg_norm_act_fwd_bf16": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P]),
    "vcg_norm_act_bwd_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vcg_norm_act_bwd_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, _P, c_int, _P, _P, _P, _P, _P,
                                      c_size_t, _P]),
    "vcg_pack_conv3x3_c64_bf16_batch": (c_int, [_P, c_int, _P, _P]),
    "vcg_bn_fold_batch": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P]),
    "vcg_conv3ch_bf16_wfrag_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vcg_pack_conv3ch_bf16": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "vcg_conv3ch_bf16_fwd": (c_int, [_D, _P, _P, _P, c_float, _P, _P]),
    "vcg_conv3ch_bf16_dgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv3ch_bf16_dgrad": (c_int, [_D, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_conv3ch_bf16_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv3ch_bf16_wgrad": (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_conv9x9_from3_bf16_fwd_train": (c_int, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "vcg_prelu_bwd_nhwc_bf16_records": (c_int, [c_int, c_int]),
    "vcg_prelu_bwd_nhwc_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "vcg_prelu_bwd_nhwc_bf16_to_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "vcg_conv2d_cout1_nhwc_bf16_fwd": (c_int, [_D, _P, _P, _P, _P, _P]),
    "vcg_conv2d_cout1_nhwc_bf16_dgrad": (c_int, [_D, _P, _P, _P, _P]),
    "vcg_conv2d_cout1_nhwc_bf16_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv2d_cout1_nhwc_bf16_wgrad": (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_conv2d_bf16_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv2d_bf16_wgrad": (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_conv9x9_to3_bf16_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv9x9_to3_bf16_wgrad": (c_int, [_D, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_pack_first9x9_bf16": (c_int, [_P, _P, _P]),
    "vcg_pack_conv9x9_3ch_bf16": (c_int, [_P, c_int, c_int, _P, _P]),
    "vcg_conv9x9_to3_bf16_dgrad": (c_int, [_D, _P, _P, _P, c_float, _P, _P]),
    "vcg_conv9x9_to3_bf16_dgrad_chsum_records": (c_int, [_D]),
    "vcg_conv9x9_to3_bf16_dgrad_chsum": (c_int, [_D, _P, _P, _P, c_float, _P, _P, _P]),
    "vcg_conv9x9_from3_bf16_fwd": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "vcg_pack_final9x9_bf16": (c_int, [_P, _P, _P]),
    "vcg_conv9x9_to3_bf16_fwd": (c_int, [_D, _P, _P, _P, c_int, _P, _P]),
    # generic bf16 NHWC convolution
    "vcg_conv_frag_bf16_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vcg_pack_conv_frag_bf16": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "vcg_pack_conv_frag_bf16_pair": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "vcg_conv2d_nhwc_bf16_fwd": (c_int, [_D, _P, _P, _P, c_int, c_float, _P, _P]),
    "vcg_conv2d_nhwc_bf16_stats_records": (c_int, [_D, c_int]),
    "vcg_conv2d_nhwc_bf16_fwd_stats": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "vcg_conv_transpose2d_nhwc_bf16_fwd": (c_int, [_D, _P, _P, _P, c_int, c_float, _P, _P]),
    "vcg_conv2d_nhwc_bf16_dgrad": (c_int, [_D, _P, _P, _P, c_float, _P, _P]),
    "vcg_conv2d_nhwc_bf16_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv2d_nhwc_bf16_wgrad": (c_int, [_D, _P, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_conv_transpose2d_nhwc_bf16_wgrad_workspace_bytes": (c_size_t, [_D]),
    "vcg_conv_transpose2d_nhwc_bf16_wgrad": (c_int, [_D, _P, _P, _P, _P, c_size_t, _P]),
    "vcg_bf16_to_f32": (c_int, [_P, _P, c_size_t, _P]),
    "vcg_f32_to_bf16": (c_int, [_P, _P, c_size_t, _P]),
}

_lib = None


def load():
    """dlopen libvcg_hip.so and bind every declared symbol.  Raises RuntimeError on failure."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvcg_hip.so not found at %s: build it with `python %s` (hipcc, gfx950). "
            "There is no CPU fallback for the hot path." % (LIB_PATH, os.path.join(_PKG_DIR, "build.py")))
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class VcgError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().vcg_error_string(int(rc)).decode()
        if rc < 0:
            raise ValueError("%s: %s (code %d)" % (what, msg, rc))
        raise VcgError("%s: HIP error %d: %s" % (what, rc, msg))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the hot path runs only on an MI355X (gfx950) GPU; "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())
