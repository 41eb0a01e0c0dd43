"""ctypes binding of libfrcnn_hip.so (the C ABI declared in include/frcnn_hip.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.  The library is
built by ``faster_rcnn_pytorch_multimodal_amd.build.build()`` (``__graft_entry__.build()``).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint32, c_void_p

from .build import LIB_PATH

_LIB = None

c_float_p = c_void_p  # device pointers travel as integers
_P = c_void_p

# name -> (restype, argtypes); mirrors include/frcnn_hip.h one to one.
PROTOTYPES = {
    "frcnn_version": (c_int, []),
    "frcnn_last_error": (c_char_p, []),
    "frcnn_conv2d_fwd_ws_bytes": (c_size_t, [c_int] * 10),
    "frcnn_conv2d_fwd": (c_int, [_P, _P, _P, _P, _P, _P] + [c_int] * 11 + [_P, c_size_t, _P]),
    "frcnn_conv2d_fwd_pre": (c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 11 + [_P, c_size_t, _P]),
    "frcnn_conv2d_winograd_filter_bytes": (c_size_t, [c_int, c_int]),
    "frcnn_conv2d_winograd_filter": (c_int, [_P, _P, c_int, c_int, _P]),
    "frcnn_conv2d_set_tile": (c_int, [c_int, c_int]),
    "frcnn_conv2d_set_staging": (c_int, [c_int]),
    "frcnn_conv2d_set_algo": (c_int, [c_int]),
    "frcnn_conv2d_plan_algo": (c_int, [c_int] * 10),
    "frcnn_conv2d_set_autotune": (c_int, [c_int]),
    "frcnn_conv2d_profile_begin": (c_int, []),
    "frcnn_conv2d_profile_end": (c_int, [POINTER(c_float), POINTER(c_int), POINTER(c_int), c_int]),
    "frcnn_conv2d_clear_plans": (c_int, []),
    "frcnn_conv2d_export_plans": (c_int, [POINTER(c_int), c_int]),
    "frcnn_conv2d_import_plans": (c_int, [POINTER(c_int), c_int]),
    "frcnn_conv2d_transpose_filter": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "frcnn_conv2d_bwd_data_ws_bytes": (c_size_t, [c_int] * 9),
    "frcnn_conv2d_bwd_data": (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P, c_size_t, _P]),
    "frcnn_conv2d_bwd_data_pre": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 9 + [_P, c_size_t, _P]),
    "frcnn_conv2d_bwd_data_act": (c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 9 + [_P, c_size_t, _P]),
    "frcnn_conv2d_bwd_weight_ws_bytes": (c_size_t, [c_int] * 9),
    "frcnn_conv2d_bwd_weight_counters": (c_int, [c_int] * 4),
    "frcnn_conv2d_wgrad_set_variant": (c_int, [c_int]),
    "frcnn_conv2d_wgrad_set_plan": (c_int, [c_int, c_int]),
    "frcnn_conv2d_bwd_weight": (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P, c_size_t, _P, _P]),
    "frcnn_conv2d_bwd_weight_acc": (c_int, [_P, _P, _P, c_int, _P] + [c_int] * 9 + [_P, c_size_t, _P, _P]),
    "frcnn_conv2d_bwd_weight_acc_grouped_ws_bytes": (c_size_t, [c_int] * 5),
    "frcnn_conv2d_bwd_weight_acc_grouped": (c_int, [_P, _P, _P, c_int, c_int] + [c_int] * 9 + [_P, c_size_t, _P]),
    "frcnn_maxpool3x3s2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "frcnn_maxpool3x3s2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "frcnn_pad_channels": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "frcnn_prep_image_out_size": (c_int, [c_int, c_int, c_float, POINTER(c_int), POINTER(c_int)]),
    "frcnn_prep_image": (c_int, [_P, c_int, c_int, c_float, POINTER(c_double), POINTER(c_double), POINTER(c_int), c_int, _P,
                                 _P]),
    "frcnn_generate_anchors": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "frcnn_rpn_decode_clip": (c_int, [_P, c_int, _P, _P, _P, POINTER(c_float), c_int, c_int, _P, _P, _P]),
    "frcnn_bbox_transform_inv": (c_int, [_P, c_int, _P, c_int, c_int, c_float, _P, _P]),
    "frcnn_clip_boxes": (c_int, [_P, c_int, POINTER(c_float), _P, _P]),
    "frcnn_lidar_bbox_transform_inv": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_float, _P, _P]),
    "frcnn_uncertainty_transform_inv": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_float, c_int, c_int, _P, _P]),
    "frcnn_sort_topk_desc_ws_bytes": (c_size_t, [c_int, c_int]),
    "frcnn_sort_topk_desc": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "frcnn_gather_rows": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "frcnn_nms_set_suppress_at_equal": (c_int, [c_int]),
    "frcnn_nms_get_suppress_at_equal": (c_int, []),
    "frcnn_nms_ws_bytes": (c_size_t, [c_int]),
    "frcnn_nms": (c_int, [_P, _P, c_int, c_float, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "frcnn_make_rois": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P]),
    "frcnn_roi_align_set_variant": (c_int, [c_int]),
    "frcnn_roi_align_fwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "frcnn_roi_align_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_float, c_int, _P, c_int, _P,
                                    _P, c_size_t, _P]),
    "frcnn_roi_align_fwd_affine": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_float, c_int, _P,
                                           c_int, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "frcnn_fpn_level_map": (c_int, [_P, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P]),
    "frcnn_head_fc_softmax_decode": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, POINTER(c_float),
                                             POINTER(c_float), c_float, _P, _P, _P, _P, _P, _P]),
    "frcnn_head_fc_softmax_decode_lidar": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P,
                                                   POINTER(c_float), POINTER(c_float), c_float, _P, _P, _P, _P, _P, _P]),
    "frcnn_generate_anchors_3d": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "frcnn_filter_set_variant": (c_int, [c_int]),
    "frcnn_filter_per_class_lidar": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, c_int, c_int, _P, _P, _P, _P,
                                             c_size_t, _P]),
    "frcnn_act_bwd": (c_int, [_P, _P, _P, c_int, c_int64, c_int, _P, _P, _P]),
    "frcnn_bev_voxelize_grid": (c_int, [POINTER(c_float), POINTER(c_float), POINTER(c_int)]),
    "frcnn_bev_voxelize_ws_bytes": (c_size_t, [c_int, POINTER(c_float), POINTER(c_float), c_int]),
    "frcnn_bev_voxelize": (c_int, [_P, c_int, c_int, POINTER(c_float), POINTER(c_float), c_float, c_int, c_int, c_int,
                                   c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_bn_train_ws_bytes": (c_size_t, [c_int]),
    "frcnn_bn_train_counters": (c_int, [c_int]),
    "frcnn_bn_train_fwd": (c_int, [_P, c_int64, c_int, _P, _P, c_float, c_float, _P, _P, _P, c_int, _P, _P, _P, _P,
                                   c_size_t, _P, _P]),
    "frcnn_bn_train_bwd": (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, _P, c_size_t, _P,
                                   _P]),
    "frcnn_spatial_mean_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "frcnn_spatial_mean_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "frcnn_upsample_bilinear_add_fwd": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
    "frcnn_upsample_bilinear_bwd": (c_int, [_P, _P] + [c_int] * 6 + [_P]),
    "frcnn_labelled_pixels_ws_bytes": (c_size_t, [c_int]),
    "frcnn_labelled_pixels": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "frcnn_gather_patches": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "frcnn_scatter_add_patches": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "frcnn_roi_align_bwd": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, c_int, _P, c_int, _P, _P]),
    "frcnn_roi_align_bwd_planned": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, c_int, _P, c_int, _P, _P,
                                            c_size_t, _P]),
    "frcnn_rpn_loss_ws_bytes": (c_size_t, []),
    "frcnn_rpn_loss": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, c_size_t, _P]),
    "frcnn_det_loss": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_float, c_float, _P, _P, _P, _P]),
    "frcnn_det_loss_lidar": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, POINTER(c_float), c_int, c_float, c_float, _P,
                                     _P, _P, _P]),
    "frcnn_det_loss_aleatoric": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, POINTER(c_float), c_int, c_float,
                                         c_float, _P, _P, _P, _P, _P]),
    "frcnn_mc_bbox_var": (c_int, [_P, c_int, c_int64, _P, _P]),
    "frcnn_mc_cls_stats": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "frcnn_mc_mean": (c_int, [_P, c_int, c_int64, _P, _P]),
    "frcnn_roi_align_fwd_split": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, c_int, c_int, _P, _P, _P, _P,
                                           c_int, c_int, _P, c_size_t, _P]),
    "frcnn_set_memops_mode": (c_int, [c_int]),
    "frcnn_get_memops_mode": (c_int, []),
    "frcnn_settings_signature": (ctypes.c_uint, []),
    "frcnn_dropout_fwd": (c_int, [_P, c_int64, c_int, c_float, c_uint32, _P, c_uint32, _P, _P]),
    "frcnn_dropout_bwd": (c_int, [_P, c_int64, c_int, c_float, c_uint32, _P, c_uint32, _P, _P]),
    "frcnn_logit_distort": (c_int, [_P, _P, c_int64, c_int, c_uint32, _P, c_uint32, c_int, _P, _P, _P]),
    "frcnn_bayesian_cross_entropy": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_uint32, _P, c_uint32, c_int, c_float, _P, _P, _P,
                                             _P, _P]),
    "frcnn_exp": (c_int, [_P, c_int64, _P, _P]),
    "frcnn_bbox_transform": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P]),
    "frcnn_lidar_bbox_transform": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P, _P]),
    "frcnn_bbox_overlaps": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, _P]),
    "frcnn_anchor_target_layer_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "frcnn_anchor_target_layer": (c_int, [_P, c_int, _P, c_int, _P, POINTER(c_float), c_int, c_float, c_float, c_float,
                                          c_uint32, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "frcnn_proposal_target_layer": (c_int, [_P, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_float, c_float, c_float,
                                            c_float, POINTER(c_float), POINTER(c_float), c_uint32, _P, _P, _P, _P, _P, _P,
                                            _P, _P, _P, _P, _P]),
    "frcnn_proposal_target_layer_lidar": (c_int, [_P, _P, _P, c_int, _P, _P, _P, c_int, _P, c_int, c_int, c_float, c_float,
                                                  c_float, c_float, POINTER(c_float), POINTER(c_float), c_uint32, _P, _P, _P,
                                                  _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frcnn_filter_per_class_ws_bytes": (c_size_t, [c_int, c_int]),
    "frcnn_filter_per_class": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int,
                                       c_int, _P, _P, _P, _P, c_size_t, _P]),
}


class HipError(RuntimeError):
    """A libfrcnn_hip.so entry point returned a negative status."""


def library_path():
    return LIB_PATH


def load():
    """dlopen libfrcnn_hip.so and type every exported entry point.  Raises if the library is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfrcnn_hip.so is not built (%s). Run `python -m faster_rcnn_pytorch_multimodal_amd.build` "
            "or __graft_entry__.build(); this package has no non-HIP execution path." % LIB_PATH)
    # torch ships its own libamdhip64: it must be in the process BEFORE this library is opened, so that both resolve
    # to ONE HIP runtime (two runtimes = two device contexts: torch's pointers would mean nothing to the kernels and
    # the second runtime reports "no ROCm-capable device").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().frcnn_last_error()
        raise HipError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))


def float_array(values):
    arr = (c_float * len(values))(*[float(v) for v in values])
    return arr
