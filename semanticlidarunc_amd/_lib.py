"""ctypes binding of libslu_hip.so (C ABI declared in include/slu.h).

There is deliberately NO fallback: if the shared library is missing or does not export the ABI
version this package was written against, importing the ops raises.  torch is used only to own
device memory and to name the HIP stream the launches go on.
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 31
MAX_SRC = 3
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslu_hip.so")

c_f32p = C.c_void_p
c_i64p = C.c_void_p
c_f64p = C.c_void_p
c_stream = C.c_void_p


class ConvSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("C", C.c_int32), ("pixel_shuffle", C.c_int32),
                ("nbatch", C.c_int32), ("cuse", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("src", ConvSrc * MAX_SRC),
        ("nsrc", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("dil", C.c_int32), ("pad", C.c_int32),
        ("ck", C.c_int32),
        ("wpack", C.c_void_p), ("bias", C.c_void_p),
        ("has_act", C.c_int32), ("slope", C.c_float),
        ("bn_a", C.c_void_p), ("bn_b", C.c_void_p),
        ("resid", C.c_void_p), ("out", C.c_void_p),
        ("precision", C.c_int32),
        ("stats", C.c_void_p),
    ]


class H8Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("G", C.c_int32), ("nbatch", C.c_int32)]


class ConvH8Desc(C.Structure):
    _fields_ = [
        ("src", H8Src * MAX_SRC),
        ("nsrc", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("dil", C.c_int32), ("pad", C.c_int32),
        ("wpack", C.c_void_p), ("bias", C.c_void_p),
        ("has_act", C.c_int32), ("slope", C.c_float),
        ("bn_a", C.c_void_p), ("bn_b", C.c_void_p),
        ("resid", C.c_void_p), ("out", C.c_void_p),
        ("out_f32_nchw", C.c_int32),
    ]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32), ("ck", C.c_int32),
                ("dgrad", C.c_int32), ("reserved", C.c_int32), ("begin", C.c_uint64)]


class ConvTailH8Desc(C.Structure):
    _fields_ = [
        ("a1", C.c_void_p), ("a2", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("w2x2", C.c_void_p), ("w1x1", C.c_void_p),
        ("biasA", C.c_void_p), ("bnA_a", C.c_void_p), ("bnA_b", C.c_void_p),
        ("hasactA", C.c_int32), ("slopeA", C.c_float),
        ("biasB", C.c_void_p), ("bnB_a", C.c_void_p), ("bnB_b", C.c_void_p),
        ("hasactB", C.c_int32), ("slopeB", C.c_float),
        ("resid", C.c_void_p), ("out", C.c_void_p),
        ("sc_x", C.c_void_p), ("sc_w", C.c_void_p), ("sc_bias", C.c_void_p),
        ("sc_cin", C.c_int32), ("sc_hasact", C.c_int32), ("sc_slope", C.c_float),
    ]


class CtxBlockH8Desc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("C", C.c_int32),
        ("w1", C.c_void_p), ("w2", C.c_void_p), ("w3", C.c_void_p),
        ("bias1", C.c_void_p),
        ("bias2", C.c_void_p), ("bn1_a", C.c_void_p), ("bn1_b", C.c_void_p),
        ("bias3", C.c_void_p), ("bn2_a", C.c_void_p), ("bn2_b", C.c_void_p),
        ("slope", C.c_float),
        ("out", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/slu.h declares
SIGNATURES = {
    "slu_abi_version": (C.c_int, []),
    "slu_strerror": (C.c_char_p, [C.c_int]),
    "slu_conv_ck": (C.c_int, [C.c_int]),
    "slu_packed_weight_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "slu_pack_conv_weight": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "slu_pack_conv_weights_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, c_stream]),
    "slu_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), c_stream]),
    "slu_packed_weight_bytes_f16x3": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "slu_pack_conv_weight_f16x3": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "slu_conv2d_kernel_name": (C.c_int, [C.POINTER(ConvDesc), C.c_char_p, C.c_size_t]),
    "slu_bn_fold": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_int, c_f32p, c_f32p, c_stream]),
    "slu_avgpool3s2_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_avgpool3s2_bcast_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_mc_reduce": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_f32p, c_f32p,
                                c_i64p, c_stream]),
    "slu_softmax_entropy": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_f32p, c_i64p,
                                      c_stream]),
    "slu_confusion_update": (C.c_int, [c_i64p, c_i64p, C.c_int64, C.c_int, c_i64p, c_stream]),
    "slu_ece_update": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, c_i64p, c_f64p,
                                 c_f64p, c_stream]),
    "slu_softmax_nll_fwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_f64p,
                                      c_stream]),
    "slu_nll_fwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int64, c_f64p, c_i64p,
                              c_stream]),
    "slu_nll_bwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int64, c_f32p, c_f32p,
                              c_stream]),
    "slu_softmax_loss_bwd": (C.c_int, [c_f32p, c_i64p, c_f32p, C.c_float, C.c_float, C.c_float, c_f32p, C.c_int, C.c_int,
                                       C.c_int, c_f32p, c_stream]),
    "slu_bn_stats": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, c_stream]),
    "slu_bn_bwd_reduce": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, c_stream]),
    "slu_bn_coeffs_fwd": (C.c_int, [c_f64p, c_f64p, C.c_double, c_f32p, c_f32p, C.c_float, C.c_float, C.c_int, c_f32p, c_f32p, C.c_int,
                                    c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "slu_bn_coeffs_bwd": (C.c_int, [c_f64p, c_f64p, C.c_double, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p,
                                    c_f32p, c_f32p, c_stream]),
    "slu_affine_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_bn_apply_fwd": (C.c_int, [c_f32p, c_f64p, c_f64p, C.c_double, c_f32p, c_f32p, C.c_float, C.c_float, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                   C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_stream]),
    "slu_bn_act_bwd": (C.c_int, [c_f32p, c_f32p, c_f64p, c_f64p, C.c_double, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                 C.c_int, C.c_int, c_f32p, c_f64p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_stream]),
    "slu_act_affine_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                     c_f32p, c_f64p, c_stream]),
    "slu_nchw_to_nhwc": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "slu_gather_nhwc": (C.c_int, [C.POINTER(ConvSrc), C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "slu_split_grad": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p,
                                 c_stream]),
    "slu_avgpool3s2_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_dgrad_weight": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_stream]),
    "slu_wgrad_packed_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "slu_conv2d_wgrad": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   c_f32p, c_f32p, c_stream]),
    "slu_conv1x1_wgrad_nchw": (C.c_int, [c_f32p, C.POINTER(ConvSrc), C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_stream]),
    "slu_conv2d_wgrad_nchw": (C.c_int, [c_f32p, C.POINTER(ConvSrc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p,
                                        C.c_int, c_stream]),
    "slu_maxpool3s2_fwd": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_nearest_down": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_space_to_depth2": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_space_to_depth2_cat": (C.c_int, [c_f32p, C.c_int, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_depth_to_space": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_row_softmax_mul": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_packed_weight_bytes_h8": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "slu_pack_conv_weight_h8": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "slu_conv2d_h8_fwd": (C.c_int, [C.POINTER(ConvH8Desc), c_stream]),
    "slu_conv2d_h8_kernel_name": (C.c_int, [C.POINTER(ConvH8Desc), C.c_char_p, C.c_size_t]),
    "slu_nchw_to_h8": (C.c_int, [c_f32p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_h8_to_nchw": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_avgpool3s2_h8": (C.c_int, [C.c_void_p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_pixel_shuffle_h8": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_dirichlet_head": (C.c_int, [c_f32p, C.c_longlong, c_f32p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_f32p,
                                     c_f32p, c_f32p, c_f32p, c_i64p, c_stream]),
    "slu_dirichlet_uncertainty": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_f32p, c_f32p, c_i64p, c_stream]),
    "slu_auroc_scores": (C.c_int, [c_f32p, c_i64p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, c_f32p,
                                   C.c_void_p, c_stream]),
    "slu_auroc_workspace_bytes": (C.c_size_t, [C.c_longlong]),
    "slu_auroc_compute": (C.c_int, [c_f32p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_size_t, c_f64p, c_f32p, C.c_void_p, c_stream]),
    "slu_ua_samples": (C.c_int, [c_i64p, c_i64p, c_f32p, C.c_longlong, c_i64p, C.c_int, c_f32p, C.c_void_p, c_stream]),
    "slu_binned_counts": (C.c_int, [c_f32p, C.c_void_p, C.c_longlong, c_f32p, C.c_int, c_i64p, c_i64p, c_stream]),
    "slu_ece_samples": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, c_f32p, C.c_void_p, c_stream]),
    "slu_binned_stats": (C.c_int, [c_f32p, C.c_void_p, C.c_longlong, c_f32p, C.c_int, c_i64p, c_i64p, c_f64p, c_stream]),
    "slu_tversky_fwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                                  c_f64p, c_f32p, c_f32p, c_f32p, c_stream]),
    "slu_tversky_bwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float, c_f32p, c_f32p,
                                  C.c_int, c_f32p, c_stream]),
    "slu_dirichlet_loss_fwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int64, c_f64p, c_i64p,
                                         c_stream]),
    "slu_dirichlet_loss_bwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int64, c_f32p, c_f32p,
                                         c_stream]),
    "slu_ctx_block_h8_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "slu_ctx_block_h8_fwd": (C.c_int, [C.POINTER(CtxBlockH8Desc), c_stream]),
    "slu_bilinear_upsample": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_groupnorm_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, c_f32p, c_f32p, c_f32p, c_stream]),
    "slu_spatial_softmax_gate": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_conv_tail_h8_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "slu_conv_tail_h8_shortcut_supported": (C.c_int, [C.c_int, C.c_int]),
    "slu_conv_tail_h8_fwd": (C.c_int, [C.POINTER(ConvTailH8Desc), c_stream]),
    "slu_head_mc_h8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_f32p, C.c_int, C.c_float, c_f32p, c_f32p, c_f32p,
                                 c_i64p, c_stream]),
    "slu_build_normals": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_stream]),
    "slu_group_by_class_workspace_bytes": (C.c_size_t, [C.c_longlong]),
    "slu_group_by_class": (C.c_int, [c_i64p, c_f32p, C.c_longlong, C.c_int, c_f32p, c_i64p, C.c_void_p, C.c_size_t, c_stream]),
    "slu_dirichlet_loss_fwd_ex": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_int, C.c_int64,
                                            c_f64p, c_i64p, c_stream]),
    "slu_dirichlet_loss_bwd_ex": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_int, C.c_int64,
                                            c_f32p, c_f32p, c_stream]),
    "slu_spherical_projection_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "slu_spherical_projection": (C.c_int, [c_f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_size_t,
                                           c_f32p, c_f64p, c_stream]),
    "slu_spherical_projection_ex": (C.c_int, [c_f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, c_f64p, C.c_int, C.c_int,
                                              C.c_int, C.c_void_p, C.c_size_t, c_f32p, c_f64p, c_stream]),
    "slu_kitti_decode": (C.c_int, [c_f32p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, c_f64p, C.c_void_p, c_stream]),
    "slu_range_image_split": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, c_stream]),
    "slu_pointwise_fwd": (C.c_int, [c_f32p, c_f32p, C.c_size_t, C.c_int, C.c_float, c_stream]),
    "slu_pointwise_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_size_t, C.c_int, C.c_float, c_stream]),
    "slu_maxpool3s2_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_nearest_down_bwd": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_replace_tail_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_replace_tail_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_row_softmax_mul_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_depth_to_space_bwd": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_bilinear_upsample_bwd": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_groupnorm_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, c_f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    c_stream]),
    "slu_spatial_softmax_gate_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_dropout_draw": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, c_f32p, C.c_longlong, c_stream]),
    "slu_dwconv3x3_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_dwconv3x3_wgrad": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_global_avgpool": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_se_gate": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_resize_nearest_hwc": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, c_stream]),
    "slu_lovasz_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "slu_lovasz_fwd": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint, C.c_void_p, C.c_size_t, c_f32p,
                                 c_f32p, c_f32p, c_stream]),
}

_lib = None


class SluError(RuntimeError):
    pass


def load():
    """Load (once) and type the shared library.  Raises if it is absent or ABI-incompatible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SluError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C semanticlidarunc_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    v = lib.slu_abi_version()
    if v != ABI_VERSION:
        raise SluError(f"libslu_hip.so ABI version {v} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().slu_strerror(code).decode()
        raise SluError(f"{what}: {msg} (code {code})")
