"""ctypes binding of libyv1.so -- the C-ABI HIP library (include/yv1.h).

There is no CPU fallback: if the library is missing, ``lib()`` raises.  Device
pointers are taken from torch tensors (``data_ptr()``), the stream from
``torch.cuda.current_stream()``; torch is plumbing only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libyv1.so")
_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_d = ctypes.c_double
c_sz = ctypes.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/yv1.h declares
# (tests/test_abi.py checks the two against each other).
SIGNATURES = {
    "yv1_loss_workspace_bytes": (c_sz, [c_i, c_i]),
    "yv1_loss_fwd_bwd": (c_i, [c_p, c_ll, c_ll, c_ll, c_ll, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_f,
                               c_p, c_p, c_p, c_p, c_sz, c_p]),
    "yv1_scale_by_device_scalar": (c_i, [c_p, c_p, c_ll, c_p]),
    "yv1_decode_nms_batched": (c_i, [c_p, c_i, c_i, c_i, c_i, c_d, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yv1_encode_targets": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "yv1_nms": (c_i, [c_p, c_p, c_i, c_f, c_p, c_p, c_p]),
    "yv1_iou_matrix": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p]),
    "yv1_convert_cxcywh_to_xyxy": (c_i, [c_p, c_i, c_i, c_p, c_p]),
    # conv_fp8.hip
    "yv1_conv2d_fwd_nhwc_fp8": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                      c_i, c_i, c_i, c_p]),
    "yv1_conv2d_fp8_stats_rows": (c_i, [c_i, c_i]),
    "yv1_conv2d_fwd_stats_nhwc_fp8": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                            c_p]),
    "yv1_prep_weights_fp8_max_tensors": (c_i, []),
    "yv1_prep_weights_fp8_multi": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "yv1_quantize_bf16_to_fp8": (c_i, [c_p, c_i, c_p, c_i, c_ll, c_i, c_p]),
    "yv1_prep_weights_fp8": (c_i, [c_p, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "yv1_fp8_fold_bn": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    # conv.hip
    "yv1_conv2d_fwd_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "yv1_conv2d_stem_fwd_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "yv1_conv2d_fwd_bn_act_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p,
                                              c_i, c_i, c_p]),
    "yv1_conv2d_stem_fwd_bn_act_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p]),
    "yv1_conv2d_dgrad_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "yv1_conv2d_dgrad_add_masked_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i,
                                                    c_p]),
    "yv1_conv2d_dgrad_add_masked_out_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_p,
                                                        c_i, c_p, c_p]),
    "yv1_conv2d_dgrad_out_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "yv1_conv2d_dgrad_gsum_rows": (c_i, [c_i, c_i, c_i]),
    "yv1_conv2d_dgrad_cat_bias_nhwc_bf16": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                                  c_p, c_i, c_p, c_p]),
    "yv1_subsample2_nhwc_bf16": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    # bn3alg.hip
    "yv1_bn3_coeffs": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yv1_bn3_build": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yv1_bn3_dw": (c_i, [c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yv1_conv2d_stats_rows": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "yv1_conv2d_dgrad_bn_deferred_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p,
                                                     c_i, c_p, c_i, c_p, c_p, c_p]),
    "yv1_conv2d_dgrad_bn_deferred_rows": (c_i, [c_i, c_i, c_i, c_i]),
    "yv1_conv2d_dgrad_bn_sums_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p,
                                                 c_p, c_p, c_p]),
    "yv1_conv2d_dgrad_bn_sums_rows": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    # bn_deferred.hip
    "yv1_bn_bwd_finalize_deferred": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "yv1_bn_deferred_fix": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p, c_ll, c_i, c_p]),
    "yv1_pack_input_nhwc4": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    # wgrad.hip
    "yv1_conv2d_wgrad_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "yv1_conv2d_wgrad_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "yv1_conv2d_wgrad_shared_nhwc_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz,
                                                c_p]),
    "yv1_conv2d_stem_wgrad_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "yv1_conv2d_stem_wgrad_bf16": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    # elementwise.hip
    "yv1_reduce_rows": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "yv1_bn_finalize": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yv1_bn_finalize_merged": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i,
                                     c_p]),
    "yv1_bn_eval_coeffs": (c_i, [c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p]),
    "yv1_bn_finalize_apply": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p,
                                    c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                    c_p, c_i, c_p, c_i, c_p, c_i, c_ll, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p]),
    "yv1_bn_finalize_merged_apply": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i,
                                           c_p, c_i, c_p, c_i, c_ll, c_i, c_p, c_p, c_p]),
    "yv1_bn_bwd_finalize_apply": (c_i, [c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p,
                                        c_p, c_ll, c_i, c_i, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p]),
    "yv1_bn_bwd_finalize_apply_dual": (c_i, [c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i,
                                             c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_ll, c_i, c_i, c_p, c_p, c_p]),
    "yv1_bn_apply": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p]),
    "yv1_bn_apply_q8": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p, c_i, c_p]),
    "yv1_bn_reduce_rows": (c_i, [c_ll, c_i]),
    "yv1_bn_stats": (c_i, [c_p, c_i, c_ll, c_i, c_p, c_p]),
    "yv1_bn_bwd_reduce": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p]),
    "yv1_bn_bwd_finalize": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "yv1_bn_bwd_apply": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i,
                               c_p, c_i, c_p, c_i, c_i, c_p]),
    "yv1_bn_bwd_reduce_dual": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_ll, c_i, c_i, c_p, c_p, c_p]),
    "yv1_bn_bwd_apply_dual": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p,
                                    c_p, c_p, c_p, c_i, c_ll, c_i, c_i, c_p]),
    "yv1_bn_bwd_reduce_pooled": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "yv1_bn_bwd_apply_pooled": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i,
                                      c_p, c_i, c_p]),
    "yv1_stats_merge": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_p]),
    "yv1_maxpool3x3s2_fwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "yv1_bn_act_maxpool3x3s2_fwd": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "yv1_maxpool3x3s2_bwd": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "yv1_avgpool2_fwd": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "yv1_avgpool2_bwd": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "yv1_head_sigmoid_fwd": (c_i, [c_p, c_i, c_p, c_p, c_p, c_ll, c_i, c_p]),
    "yv1_head_sigmoid_bwd": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_ll, c_i, c_p]),
    "yv1_prep_weights": (c_i, [c_p, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "yv1_prep_weights_max_tensors": (c_i, []),
    "yv1_prep_weights_multi": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "yv1_prep_stem_weights": (c_i, [c_p, c_ll, c_ll, c_ll, c_ll, c_i, c_p, c_p]),
    "yv1_unpack_stem_grad": (c_i, [c_p, c_p, c_ll, c_ll, c_ll, c_ll, c_i, c_p]),
    # cfglog.hip
    "yv1_last_config": (c_i, [ctypes.c_char_p, c_i]),
    # optim.hip
    "yv1_sgd_max_tensors": (c_i, []),
    "yv1_sgd_momentum_step": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_f, c_f, c_p]),
}


class Yv1Error(RuntimeError):
    pass


def lib():
    """Loads libyv1.so once.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Yv1Error(
                "libyv1.so not found at %s -- build it with `python -m yolo_v1_amd.build` "
                "(there is no CPU fallback for the HIP hot path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_config():
    """The kernel templates the last conv / dgrad / wgrad call of this thread launched (list of names)."""
    buf = ctypes.create_string_buffer(512)
    lib().yv1_last_config(buf, 512)
    return [c for c in buf.value.decode().split(";") if c]


def stream_ptr(device=None):
    return c_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return c_p(t.data_ptr()) if t is not None else c_p(0)


LAUNCHES = [0]          # C-ABI calls made so far (every kernel launch goes through check()); ops.SideStream reads it


def check(code, what):
    LAUNCHES[0] += 1
    if code != 0:
        names = {1001: "bad argument", 1002: "unsupported shape", 1003: "workspace too small"}
        raise Yv1Error("%s failed: code %d (%s)" % (what, code, names.get(code, "hipError_t")))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Yv1Error("yolo_v1_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback"
                           % t.device)
