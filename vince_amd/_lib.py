"""ctypes binding of libvince_hip.so (include/vince_hip.h).  Fails loudly: no library -> RuntimeError."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvince_hip.so")

VINCE_F32, VINCE_BF16 = 0, 1
# fp32 tensors, matrix products as three half-precision MFMAs of hi / lo halves (include/vince_hip.h): H = IEEE half halves (forward),
# B = bfloat16 halves (gradients); VINCE_F32X3 is what vince_trunk_cfg.dtype takes
VINCE_F32X3H, VINCE_F32X3B = 2, 3
VINCE_F32X3 = VINCE_F32X3H
# fp32 tensors, single bfloat16 products (op level: gradient launches); trunk cfg: x3 forward + single-product gradients ("x3f")
VINCE_F32X1B, VINCE_F32X3F = 4, 5
EPI_ACCUMULATE, EPI_RELU = 1, 2
EPI_IN_HALF_PAIRS = 8     # VINCE_F32X3H launches: the input tensor holds stored IEEE-half pairs (vince_bn_train.out_half_pairs)

c_void_p, c_int, c_int32, c_int64, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64,
                                                        ctypes.c_float, ctypes.c_size_t)


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "N", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "sh", "sw", "TA", "TB", "dh0", "dhs", "dw0", "dws",
        "wt0", "wta", "wtb", "WT", "OH", "OW", "osh", "osw", "oh0", "ow0", "Cs", "Kw")]


class BnReduce(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("y", "mask_bits", "mask_scale", "mask_shift", "mean", "invstd", "sums")]


class ConvEpi(ctypes.Structure):
    _fields_ = [("flags", c_int32), ("bias", c_void_p), ("stats", c_void_p), ("acc_mask", c_void_p),
                ("bnred", BnReduce), ("replicas", c_int32),
                ("out_scale", c_void_p), ("id_scale", c_void_p), ("id_shift", c_void_p), ("out_mask", c_void_p), ("in2", c_void_p), ("in2_channels", c_int32), ("in2_repeat", c_int32), ("out2", c_void_p),
                ("raw2", c_void_p), ("raw2_mean", c_void_p), ("mask2", c_void_p)]


class BnTrain(ctypes.Structure):
    _fields_ = [("stats", c_void_p), ("replicas", c_int32), ("count", ctypes.c_int64), ("gamma", c_void_p), ("beta", c_void_p),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("num_batches_tracked", c_void_p),
                ("momentum", c_float), ("eps", c_float), ("scale", c_void_p), ("shift", c_void_p),
                ("save_mean", c_void_p), ("save_invstd", c_void_p), ("out_sum", c_void_p), ("out_sum_replicas", c_int32),
                ("out_bf16", c_void_p), ("mask_bf16", c_void_p), ("y_centred_bf16", c_void_p), ("shadow_consts", c_void_p),
                ("out_half_pairs", c_int32)]


class InfoNCEDesc(ctypes.Structure):
    _fields_ = [("B", c_int32), ("D", c_int32), ("Bk", c_int32), ("K", c_int32), ("frames", c_int32),
                ("offdiag_neg", c_int32), ("inv_temperature", c_float)]


class TrunkCfg(ctypes.Structure):
    _fields_ = [("arch", c_int32), ("N", c_int32), ("H", c_int32), ("W", c_int32), ("dtype", c_int32)]


P = ctypes.POINTER
# name -> (restype, argtypes); every symbol declared in include/vince_hip.h
PROTOTYPES = {
    "vince_last_error": (ctypes.c_char_p, []),
    "vince_abi_version": (c_int, []),
    "vince_profile_enable": (c_int, [c_int]),
    "vince_set_side_streams": (c_int, [c_int32]),
    "vince_profile_collect": (c_int, [c_int32, c_void_p, c_void_p, c_void_p]),
    "vince_stream_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_void_p]),
    "vince_conv_igemm": (c_int, [P(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, P(ConvEpi), c_void_p]),
    "vince_conv_expand_join": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vince_conv_expand_join_next": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int, c_void_p]),
    "vince_conv3x3_strip": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "vince_conv3x3_strip_bias": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "vince_conv3x3_strip_dgrad": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, P(BnReduce), c_int32, c_void_p]),
    "vince_conv_expand_stats": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "vince_conv_expand_dgrad_masked": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int, c_void_p, c_void_p,
                                               c_void_p, c_int32, c_void_p]),
    "vince_bn3_bwd_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                      c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                      c_void_p, c_void_p]),
    "vince_bn3_bwd_finish_dw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                        c_int32, c_void_p]),
    "vince_conv_expand_dgrad": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int, c_void_p, P(BnReduce),
                                        c_int32, c_void_p]),
    "vince_conv_wgrad_det": (c_int, [P(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "vince_conv_wgrad_scratch_bytes": (c_size_t, [P(ConvDesc), c_int, c_int32]),
    "vince_conv_wgrad": (c_int, [P(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_int32, c_int, c_void_p]),
    "vince_bn_finalize": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                  c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vince_bn_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int64, c_int32, c_int, c_void_p]),
    "vince_bn_train_apply": (c_int, [c_int, c_void_p, P(BnTrain), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_int32, c_int, c_void_p]),
    "vince_bn_train_apply_gram_scratch_bytes": (ctypes.c_size_t, [c_int64, c_int32]),
    "vince_bn_train_apply_gram": (c_int, [c_int, c_void_p, P(BnTrain), c_void_p, c_int64, c_int32, c_void_p, c_void_p, ctypes.c_size_t,
                                          c_void_p]),
    "vince_bn_gram_finalize": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "vince_bn_bwd_reduce": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "vince_bn_bwd_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                   c_int32, c_void_p, c_void_p]),
    "vince_stem_pool_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
    "vince_stem_pool_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vince_stem_bwd_reduce": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                      c_int32, c_void_p]),
    "vince_stem_bwd_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vince_avgpool_fwd": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "vince_avgpool_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "vince_input_nchw_to_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p]),
    "vince_jigsaw_nchw_to_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, c_void_p]),
    "vince_input_nchw_to_rows": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_int32, c_void_p]),
    "vince_jigsaw_nchw_to_rows": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, c_int32, c_void_p]),
    "vince_input_u8hwc_to_rows": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, P(c_float), P(c_float), c_void_p,
                                          c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vince_aug_resized_crop_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                          c_int32, c_int32, c_void_p]),
    "vince_aug_resample_table_ints": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "vince_aug_resample_kmax": (c_int, [c_int32, c_int32, c_int32, c_int32]),
    "vince_aug_color_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vince_aug_blur_to_rows": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, P(c_float), P(c_float), c_void_p,
                                       c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vince_prepare_weight": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vince_transpose_f32": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "vince_prepare_weights_batched": (c_int, [c_int, c_void_p, c_int32, c_void_p]),
    "vince_nhwc_to_nchw_f32": (c_int, [c_int, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "vince_l2norm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "vince_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "vince_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vince_colsum": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "vince_nonfinite_latch": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "vince_infonce_workspace_bytes": (c_size_t, [P(InfoNCEDesc)]),
    "vince_infonce_fwd": (c_int, [P(InfoNCEDesc)] + [c_void_p] * 12),
    "vince_infonce_bwd": (c_int, [P(InfoNCEDesc)] + [c_void_p] * 11),
    "vince_sce_rows_fwd": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "vince_sce_rows_bwd": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "vince_queue_enqueue": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, P(c_int64), P(c_int32), c_void_p]),
    "vince_ema_flat": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "vince_sgd_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p]),
    "vince_trunk_create": (c_int, [P(TrunkCfg), P(c_void_p)]),
    "vince_trunk_destroy": (None, [c_void_p]),
    "vince_trunk_num_params": (c_int32, [c_void_p]),
    "vince_trunk_num_bn": (c_int32, [c_void_p]),
    "vince_trunk_param_info": (c_int, [c_void_p, c_int32, ctypes.c_char_p, c_int32, P(c_int32), P(c_int32 * 4), P(c_int32)]),
    "vince_trunk_bn_info": (c_int, [c_void_p, c_int32, ctypes.c_char_p, c_int32, P(c_int32)]),
    "vince_trunk_out_channels": (c_int32, [c_void_p]),
    "vince_trunk_out_hw": (c_int32, [c_void_p, P(c_int32), P(c_int32)]),
    "vince_trunk_workspace_bytes": (c_size_t, [c_void_p]),
    "vince_trunk_weight_cache_bytes": (c_size_t, [c_void_p]),
    "vince_trunk_prepare_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "vince_trunk_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                    c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "vince_trunk_spatial_ptr": (c_void_p, [c_void_p, c_void_p]),
    "vince_trunk_input_ptr": (c_void_p, [c_void_p, c_void_p, P(c_int32), P(c_int32)]),
    "vince_trunk_prepare_weights_folded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vince_trunk_forward_folded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                           c_void_p]),
    "vince_trunk_set_bucket_callback": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vince_trunk_set_stem_event": (c_int, [c_void_p, c_void_p]),
    "vince_trunk_stem_join": (c_int, [c_void_p, c_void_p]),
    "vince_launch_count": (ctypes.c_int64, []),
    "vince_trunk_set_shadow": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vince_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "vince_trunk_prepare_weights_part": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "vince_trunk_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                     c_void_p]),
    "vince_trunk_num_blocks": (c_int32, [c_void_p]),
}

_LIB = None


# include/vince_hip.h VINCE_ABI_VERSION (tests/test_abi_cpu.py holds the two together)
ABI_VERSION = 12


def lib():
    """The loaded library.  Raises if it has not been built (python -m vince_amd.build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "vince_amd: %s is missing -- build the HIP extension first (python -m vince_amd.build). "
                "There is no CPU fallback." % LIB_PATH)
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Loading it FIRST makes the dynamic linker bind
        # our NEEDED libamdhip64.so.7 to that same runtime instance; the other order loads a second HIP runtime from
        # /opt/rocm whose streams / device state are not torch's ("no ROCm-capable device" on the first stream op).
        import torch  # noqa: F401
        L = ctypes.CDLL(os.environ.get("VINCE_HIP_LIB", LIB_PATH))   # override: A/B measurements of kernel builds
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        built = L.vince_abi_version()
        if built != ABI_VERSION:
            raise RuntimeError("vince_amd: %s was built with ABI version %d, these bindings are version %d -- rebuild "
                               "(python -m vince_amd.build --force)" % (getattr(L, "_name", LIB_PATH), built, ABI_VERSION))
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise RuntimeError("libvince_hip error %d: %s" % (rc, lib().vince_last_error().decode()))
