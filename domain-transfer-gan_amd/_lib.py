"""ctypes binding of libacgan_hip.so (C ABI declared in include/acgan_hip.h).

The shared object is built in-tree (csrc/Makefile, `__graft_entry__.build()`); there is
no CPU or eager fallback: if the library is missing or a kernel call fails, this module
raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACGAN_HIP_LIB: load another build of the SAME library (A/B timing of kernel changes on one GPU box)
LIB_PATH = os.environ.get("ACGAN_HIP_LIB") or os.path.join(_HERE, "libacgan_hip.so")

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
IMPL_MFMA, IMPL_DIRECT = 0, 1
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2

c_int, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """acg_conv_desc (include/acgan_hip.h)."""
    _fields_ = [(k, c_int) for k in ("N", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "K", "stride", "pad", "pad_mode", "Cir", "Cor")]


class LatentMlpParams(ctypes.Structure):
    """acg_latent_mlp_params (include/acgan_hip.h)."""
    _fields_ = [("w", c_void_p * 4), ("b", c_void_p * 4), ("gamma", c_void_p * 3), ("beta", c_void_p * 3),
                ("run_mean", c_void_p * 3), ("run_var", c_void_p * 3)]


class LatentMlpGrads(ctypes.Structure):
    """acg_latent_mlp_grads (include/acgan_hip.h)."""
    _fields_ = [("dw", c_void_p * 4), ("db", c_void_p * 4), ("dgamma", c_void_p * 3), ("dbeta", c_void_p * 3)]


class AdamGroup(ctypes.Structure):
    """acg_adam_group (include/acgan_hip.h)."""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_size_t), ("sumsq", c_void_p)]


class PackItem(ctypes.Structure):
    """acg_pack_item (include/acgan_hip.h)."""
    _fields_ = [("w", c_void_p), ("wf", c_void_p), ("wb", c_void_p), ("Or", c_int), ("Ir", c_int), ("K", c_int), ("Ci", c_int), ("Co", c_int)]


MAX_SEGMENTS = 96


class Segments(ctypes.Structure):
    """acg_segments (include/acgan_hip.h)."""
    _fields_ = [("dst", c_void_p * MAX_SEGMENTS), ("off", c_int * MAX_SEGMENTS), ("len", c_int * MAX_SEGMENTS), ("n", c_int)]


class NormSumsDesc(ctypes.Structure):
    """acg_norm_sums (include/acgan_hip.h)."""
    _fields_ = [("x", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("gstride", c_int), ("sign_mask", c_void_p), ("act", c_int), ("part", c_void_p)]


ADAM_MAX_GROUPS = 8
_P = c_void_p
_D = ctypes.POINTER(ConvDesc)
_MP = ctypes.POINTER(LatentMlpParams)
_MG = ctypes.POINTER(LatentMlpGrads)

# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "acg_version": (c_int, []),
    "acg_last_error": (ctypes.c_char_p, []),
    "acg_last_kernel": (ctypes.c_char_p, []),
    "acg_set_conv_impl": (c_int, [c_int]),
    "acg_set_conv_precision": (c_int, [c_int]),
    "acg_minmax_scale_nhwc_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "acg_nchw_to_nhwc16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "acg_nhwc16_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "acg_concat_channels": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_size_t, _P]),
    "acg_split_channels": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_size_t, _P]),
    "acg_ncols_pad": (c_int, [c_int]),
    "acg_packed_wf_elems": (c_size_t, [c_int, c_int, c_int]),
    "acg_packed_wb_elems": (c_size_t, [c_int, c_int, c_int]),
    "acg_pack_conv_weight": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "acg_pad_vector": (c_int, [_P, c_int, _P, c_int, _P]),
    "acg_conv2d_fwd": (c_int, [_D, _P, _P, _P, _P, c_int, _P]),
    "acg_conv2d_bwd_data_workspace_bytes": (c_size_t, [_D]),
    "acg_conv2d_bwd_data": (c_int, [_D, _P, _P, _P, _P, c_size_t, _P]),
    "acg_s16_encode": (c_int, [_P, _P, c_size_t, _P]),
    "acg_s16_decode": (c_int, [_P, _P, c_size_t, _P]),
    "acg_conv2d_s16_supported": (c_int, [_D]),
    "acg_conv2d_fwd_s16": (c_int, [_D, _P, _P, _P, _P, c_int, _P, c_int, _P]),
    "acg_conv2d_bwd_data_s16": (c_int, [_D, _P, _P, _P, _P, c_size_t, _P, _P, _P, c_int, _P]),
    "acg_conv2d_fwd_s16_mask": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "acg_conv2d_bwd_data_s16_mask": (c_int, [_D, _P, _P, _P, _P, c_size_t, _P, _P]),
    "acg_conv2d_bwd_data_s16_sums_supported": (c_int, [_D]),
    "acg_conv2d_bwd_data_s16_sums": (c_int, [_D, _P, _P, _P, _P, c_size_t, _P, _P, ctypes.POINTER(NormSumsDesc), _P]),
    "acg_conv2d_bwd_data_sums_supported": (c_int, [_D]),
    "acg_conv2d_bwd_data_sums": (c_int, [_D, _P, _P, _P, _P, c_size_t, ctypes.POINTER(NormSumsDesc), _P]),
    "acg_conv2d_bwd_weight_s16": (c_int, [_D, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "acg_conv2d_bwd_weight_workspace_bytes": (c_size_t, [_D]),
    "acg_conv2d_bwd_weight": (c_int, [_D, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "acg_conv_transpose2d_fwd": (c_int, [_D, _P, _P, _P, _P, c_int, _P]),
    "acg_conv_transpose2d_fwd_stats_supported": (c_int, [_D]),
    "acg_conv_transpose2d_fwd_stats": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "acg_conv_transpose2d_bwd_data": (c_int, [_D, _P, _P, _P, _P]),
    "acg_conv_transpose2d_bwd_weight": (c_int, [_D, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    "acg_norm_workspace_bytes": (c_size_t, [c_int, c_size_t, c_int]),
    "acg_conv2d_fwd_stats_supported": (c_int, [_P]),
    "acg_conv2d_bwd_data_add_supported": (c_int, [_P]),
    "acg_conv2d_bwd_data_add": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "acg_pack_conv_weights_multi_supported": (c_int, [c_int, c_int, c_int]),
    "acg_pack_conv_weights_multi": (c_int, [_P, c_int, _P]),
    "acg_debug_mid_event": (c_int, [_P]),
    "acg_probe_mfma_rate": (c_int, [_P, c_size_t, c_int, ctypes.POINTER(ctypes.c_double), _P]),
    "acg_mask_apply": (c_int, [_P, _P, _P, c_size_t, _P]),
    "acg_dropout_apply": (c_int, [_P, _P, c_float, _P, c_size_t, _P]),
    "acg_conv2d_bwd_data_relu": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "acg_conv2d_fwd_stats": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "acg_norm_stats_from_partials": (c_int, [_P, c_int, c_size_t, c_int, c_int, c_float, c_int, _P, _P, _P]),
    "acg_norm_stats": (c_int, [_P, c_int, c_size_t, c_int, c_float, c_int, _P, _P, _P, _P, c_float, _P, c_size_t, _P]),
    "acg_bn_eval_stats": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P]),
    "acg_norm_apply": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_size_t, c_int, c_int, c_int, _P]),
    "acg_norm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_size_t, c_int, c_int,
                             c_int, c_int, _P, c_size_t, _P]),
    "acg_norm_bwd_partials": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_size_t, c_int,
                                      c_int, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    "acg_norm_bwd_sums": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_size_t, c_int, c_int, _P, c_size_t, _P]),
    "acg_norm_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_size_t, c_size_t, c_int, c_int, c_int, _P]),
    "acg_act_bwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "acg_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "acg_linear_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "acg_latent_mlp_supported": (c_int, [c_int, c_int, c_int]),
    "acg_latent_mlp_fwd": (c_int, [_MP, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P]),
    "acg_latent_mlp_bwd": (c_int, [_MP, _MG, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "acg_segments_accumulate": (c_int, [_P, ctypes.POINTER(Segments), c_int, _P]),
    "acg_spatial_mean_fwd": (c_int, [_P, _P, c_int, c_size_t, c_int, _P]),
    "acg_spatial_mean_bwd": (c_int, [_P, _P, c_int, c_size_t, c_int, _P]),
    "acg_reduce_workspace_bytes": (c_size_t, [c_size_t]),
    "acg_mse_const_fwd": (c_int, [_P, c_size_t, c_int, c_int, c_float, _P, _P, c_size_t, _P]),
    "acg_mse_const_bwd": (c_int, [_P, c_size_t, c_int, c_int, c_float, _P, _P, _P]),
    "acg_l1_fwd": (c_int, [_P, _P, c_size_t, c_int, c_int, _P, _P, c_size_t, _P]),
    "acg_l1_bwd": (c_int, [_P, _P, c_size_t, c_int, c_int, _P, _P, _P, _P]),
    "acg_mean_fwd": (c_int, [_P, c_size_t, c_int, c_int, _P, _P, c_size_t, _P]),
    "acg_sumsq": (c_int, [_P, c_size_t, _P, _P, c_size_t, _P]),
    "acg_comm_unique_id": (c_int, [_P]),
    "acg_comm_init": (c_int, [ctypes.POINTER(c_void_p), _P, c_int, c_int]),
    "acg_comm_allreduce_mean": (c_int, [_P, _P, c_size_t, _P]),
    "acg_comm_destroy": (c_int, [_P]),
    "acg_clip_adam_multi_workspace_bytes": (c_size_t, [c_int]),
    "acg_clip_adam_multi": (c_int, [ctypes.POINTER(AdamGroup), c_int, c_float, c_float, c_float, c_float, c_float, c_int, _P, _P,
                                    c_size_t, _P]),
    "acg_adam_step": (c_int, [_P, _P, _P, _P, c_size_t, _P, c_float, c_float, c_float, c_float, c_float, c_int,
                              c_int, _P]),
}

_lib = None
ABI_VERSION = 117   # include/acgan_hip.h ACG_VERSION this binding was written against


class AcgError(RuntimeError):
    pass


def load():
    """Load libacgan_hip.so and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AcgError("libacgan_hip.so not found at %s — build it with `make -C %s` "
                       "(or __graft_entry__.build()); there is no fallback path." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    got = lib.acg_version()
    if got != ABI_VERSION:   # a stale build would take shifted arguments instead of failing cleanly
        raise AcgError("%s reports ABI version %d, this binding expects %d — rebuild it (make -C %s)"
                       % (LIB_PATH, got, ABI_VERSION, os.path.join(_HERE, "csrc")))
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point; negative status -> AcgError(acg_last_error())."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise AcgError("%s failed (%d): %s" % (name, rc, lib.acg_last_error().decode()))


def query(name, *args):
    """Invoke a value-returning entry point (sizes, versions)."""
    return getattr(load(), name)(*args)
