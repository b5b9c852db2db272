"""ctypes binding of libptmi355.so (C ABI in include/ptmi355.h).

The HIP library is THE compute path.  There is no CPU/eager fallback: if the shared object is missing or a
symbol is absent this module raises at first use, loudly.
"""
import ctypes
import os

# torch must be loaded FIRST: it bundles its own libamdhip64.so.7, and libptmi355.so has to resolve to that same
# HIP runtime instance (same soname) so that streams / device pointers are shared with the caching allocator.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptmi355.so")

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes); mirrors include/ptmi355.h one to one
SIGNATURES = {
    "ptmi_last_error": (ctypes.c_char_p, []),
    "ptmi_abi_version": (_i, []),
    "ptmi_conv3x3_bm": (_i, [_i]),
    "ptmi_conv3x3_ck": (_i, [_i]),
    "ptmi_conv3x3_packed_floats": (_i64, [_i, _i]),
    "ptmi_conv3x3_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ptmi_conv3x3_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wgrad_ws_floats": (_i64, [_i, _i, _i, _i, _i]),
    "ptmi_conv3x3_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino_packed_floats": (_i64, [_i, _i]),
    "ptmi_conv3x3_wino_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino_fwd_fits": (_i, [_i, _i, _i, _i]),
    "ptmi_conv3x3_wino_wgrad_fits": (_i, [_i, _i]),
    "ptmi_conv3x3_wino4_packed_floats": (_i64, [_i, _i]),
    "ptmi_conv3x3_wino4_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino4_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino4_fwd_sched": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ptmi_conv3x3_wino4_fwd_fits": (_i, [_i, _i, _i, _i]),
    "ptmi_conv3x3_wino4p_packed_floats": (_i64, [_i, _i]),
    "ptmi_conv3x3_wino4p_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino4p_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino4p_fwd_sched": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ptmi_conv3x3_wino4p_fwd_fits": (_i, [_i, _i, _i, _i]),
    "ptmi_conv3x3_wino4_wgrad_fits": (_i, [_i, _i]),
    "ptmi_conv3x3_wino4_wgrad_ws_floats": (_i64, [_i, _i, _i, _i, _i]),
    "ptmi_conv3x3_wino4_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino4_wgrad_ws_floats_waves": (_i64, [_i, _i, _i, _i, _i, _i]),
    "ptmi_conv3x3_wino4_wgrad_waves": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino_wgrad_ws_floats_waves": (_i64, [_i, _i, _i, _i, _i, _i]),
    "ptmi_conv3x3_wino_wgrad_waves": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_conv3x3_wino_wgrad_ws_floats": (_i64, [_i, _i, _i, _i, _i]),
    "ptmi_conv3x3_wino_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_p8_plane_pixels": (_i64, [_i, _i, _i]),
    "ptmi_p8_planes": (_i, [_i]),
    "ptmi_p8_from_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ptmi_p8_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ptmi_p8_maxpool2x2_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ptmi_p8_maxpool2x2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ptmi_p8_relu_bwd": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "ptmi_p8_packed_elems": (_i64, [_i, _i]),
    "ptmi_p8_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ptmi_p8_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_p8_wgrad_fits": (_i, [_i, _i, _i, _i, _i]),
    "ptmi_p8_gemm_nt_fits": (_i, [_i, _i, _i]),
    "ptmi_p8_wgrad_ws_floats": (_i64, [_i, _i, _i, _i, _i]),
    "ptmi_p8_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_p8_conv3x3_waves": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_p8_wgrad_ws_floats_waves": (_i64, [_i, _i, _i, _i, _i, _i]),
    "ptmi_p8_wgrad_waves": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_p8m_elems": (_i64, [_i, _i]),
    "ptmi_p8m_pack": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp]),
    "ptmi_p8_gemm_nt_ws_floats": (_i64, [_i, _i, _i]),
    "ptmi_p8_gemm_nt": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ptmi_relu_bwd": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "ptmi_maxpool2x2_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ptmi_maxpool2x2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ptmi_gemm_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _vp, _i64, _vp]),
    "ptmi_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _vp, _i64, _vp]),
    "ptmi_gemm_ws_floats": (_i64, [_i, _i, _i, _i]),
    "ptmi_colsum": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ptmi_colsum_ws_floats": (_i64, [_i, _i]),
    "ptmi_colsum_ws": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ptmi_rowsum_batched": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ptmi_roi_align_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ptmi_roi_align_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ptmi_roi_align_ws_bytes": (_i64, [_i, _i, _i]),
    "ptmi_roi_align_fwd_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ptmi_roi_align_bwd_ws_bytes": (_i64, [_i, _i, _i]),
    "ptmi_roi_align_fwd_p8m_fits": (_i, [_i, _i, _i, _i]),
    "ptmi_roi_align_fwd_p8m": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ptmi_roi_align_bwd_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ptmi_grid_anchors": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "ptmi_apply_deltas": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i64, _f, _f, _f, _f, _f, _vp]),
    "ptmi_get_deltas": (_i, [_vp, _vp, _vp, _i64, _f, _f, _f, _f, _vp]),
    "ptmi_iou_match": (_i, [_vp, _vp, _i, _i64, ctypes.POINTER(_f), ctypes.POINTER(_i), _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ptmi_iou_match_batched": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i64, ctypes.POINTER(_f), ctypes.POINTER(_i), _i, _i,
                                    _vp, _vp, _vp, _vp, _vp]),
    "ptmi_rpn_subsample_relabel": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "ptmi_sample_by_keys": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ptmi_segsort_ws_bytes": (_i64, [_i64, _i]),
    "ptmi_segsort_desc": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _i64, _vp]),
    "ptmi_segsort_topk_fits": (_i, [_i64, _i64]),
    "ptmi_segsort_topk_desc": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _i64, _vp]),
    "ptmi_rpn_prepare": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp]),
    "ptmi_nms_ws_bytes": (_i64, [_i64, _i]),
    "ptmi_nms_batched": (_i, [_vp, _vp, _vp, _i, _i64, _f, _i, _vp, _vp, _vp, _vp]),
    "ptmi_roi_infer_prepare": (_i, [_vp] * 11 + [_i64, _i, _i, _f, _f, _f, _f, _f, _f, _vp]),
    "ptmi_roi_infer_nms_boxes": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ptmi_bce_logits_sum": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _vp, _vp]),
    "ptmi_gaussian_nll_sum": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _vp, _vp, _vp]),
    "ptmi_softmax_ce_mean": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "ptmi_softmax_rows": (_i, [_vp, _vp, _i64, _i, _vp]),
    "ptmi_soft_ce_efl": (_i, [_vp, _vp, _i64, _i, _f, _f, _i, _f, _vp, _vp, _vp, _vp]),
    "ptmi_rpn_soft_obj_loss": (_i, [_vp, _vp, _i64, _i, _f, _f, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "ptmi_kl_efl_loss": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "ptmi_get_deltas_bwd_src": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _vp, _vp]),
    "ptmi_hold_cus": (_i, [_vp, _i, _i, _vp]),
    "ptmi_ema_update": (_i, [_vp, _vp, _i64, _f, _f, _vp]),
    "ptmi_sumsq": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "ptmi_clip_sgd_step": (_i, [_vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _i, _vp]),
    "ptmi_scale_by_clip": (_i, [_vp, _i64, _vp, _f, _vp]),
    "ptmi_preprocess_image": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _vp]),
    "ptmi_shrink_paste": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ptmi_preprocess_batched": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _f, _f, _f, _f, _vp]),
    "ptmi_shrink_paste_batched": (_i, [_vp, _i, _i64, _i, _i, _i, _vp]),
    "ptmi_aug_gray_sum_batched": (_i, [_vp, _i, _i64, _vp, _vp]),
    "ptmi_aug_color_batched": (_i, [_vp, _i, _i64, _vp, _vp]),
    "ptmi_aug_box_blur_batched": (_i, [_vp, _i, _i64, _vp]),
    "ptmi_aug_hflip_batched": (_i, [_vp, _i, _i64, _vp]),
    "ptmi_aug_resize_pass_batched": (_i, [_vp, _i, _i64, _vp]),
}

_lib = None


class PtmiError(RuntimeError):
    pass


def load():
    """Load libptmi355.so and bind every symbol of the ABI.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtmiError(
            f"{LIB_PATH} not found: the HIP extension is the only compute path of probabilisticteacher_amd "
            "(no CPU/eager fallback).  Build it with `python -m probabilisticteacher_amd.build_ext`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PtmiError(f"libptmi355.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point and raise PtmiError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise PtmiError(f"{name} failed ({rc}): {lib.ptmi_last_error().decode()}")
