"""ctypes binding of libe4s_hip.so (include/e4s_hip.h).

This is the thin host-side shim a maintainer of the reference would add in place of
``torch.utils.cpp_extension.load(...)`` (src/models/stylegan2/op/fused_act.py:8-15,
upfirdn2d.py:7-14): PyTorch supplies device memory (``Tensor.data_ptr()``) and the current HIP
stream; the library gets raw pointers and sizes.  There is NO fallback: if the shared object is
missing, or a tensor is not a contiguous fp32 tensor on a ROCm device, we raise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("E4S_LIB_PATH") or os.path.join(_HERE, "libe4s_hip.so")      # (E4S_LIB_PATH: A/B runs of two builds)
ABI_VERSION = 14

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float
c_d = ctypes.c_double


STYLE_GRAD_MAX_JOBS = 32


class StyleGradJob(ctypes.Structure):
    """Mirror of ``e4s_style_grad_job`` (include/e4s_hip.h)."""
    _fields_ = [
        ("ds_raw", c_p), ("dd_d", c_p), ("d", c_p), ("s", c_p), ("wsq", c_p), ("dws", c_p), ("w3", c_p), ("wmod", c_p), ("ds_total", c_p),
        ("conv_scale", c_f), ("mod_scale", c_f), ("G", c_i), ("Cin", c_i), ("Cout", c_i), ("slot", c_i), ("masked", c_i),
    ]


class ConvParams(ctypes.Structure):
    """Mirror of ``e4s_conv_params`` (include/e4s_hip.h)."""
    _fields_ = [
        ("x", c_p), ("w", c_p), ("y", c_p), ("rows", c_p), ("tiles", c_p), ("meta", c_p), ("tiles_cap", c_i),
        ("B", c_i), ("Ha", c_i), ("Wa", c_i),
        ("Hi", c_i), ("Wi", c_i), ("Ho", c_i), ("Wo", c_i), ("Cin", c_i), ("Cout", c_i),
        ("istride", c_i), ("ostride", c_i), ("ntaps", c_i), ("ncls", c_i),
        ("in_scale", c_p), ("out_scale", c_p), ("groups_per_batch", c_i),
        ("labels", c_p), ("Hm", c_i), ("Wm", c_i),
        ("noise", c_p), ("noise_w", c_p), ("noise_bstride", c_l), ("noise_per_channel", c_i),
        ("bias", c_p), ("slope", c_p), ("act", c_i), ("alpha", c_f), ("gain", c_f),
        ("in_stats", c_p), ("y_cstride", c_i), ("splitk_ws", c_p), ("stats_ws", c_p), ("stats_slots", c_i), ("tap_shift", c_i),
        ("split_hint", c_i),
    ]


class ConvBwdParams(ctypes.Structure):
    """Mirror of ``e4s_conv_bwd_params`` (include/e4s_hip.h)."""
    _fields_ = [
        ("gz", c_p), ("wt", c_p), ("dx", c_p), ("x", c_p), ("ds", c_p), ("s", c_p), ("d", c_p), ("labels", c_p),
        ("Hm", c_i), ("Wm", c_i), ("R", c_i),
        ("B", c_i), ("Hx", c_i), ("Wx", c_i), ("Cx", c_i), ("Hy", c_i), ("Wy", c_i), ("Cy", c_i),
        ("ncls", c_i), ("ds_ws", c_p),
    ]


class ConvWgradParams(ctypes.Structure):
    """Mirror of ``e4s_conv_wgrad_params`` (include/e4s_hip.h)."""
    _fields_ = [
        ("gz", c_p), ("x", c_p), ("dw", c_p), ("ws", c_p), ("s", c_p), ("d", c_p), ("labels", c_p),
        ("Hm", c_i), ("Wm", c_i), ("R", c_i),
        ("B", c_i), ("Hi", c_i), ("Wi", c_i), ("Cin", c_i), ("Ha", c_i), ("Wa", c_i), ("Ho", c_i), ("Wo", c_i), ("Cout", c_i),
        ("istride", c_i), ("ostride", c_i), ("py", c_i), ("px", c_i), ("ntaps", c_i), ("tap_shift", c_i),
    ]


# name -> argtypes (all return int except the two info calls); keep in sync with include/e4s_hip.h
SIGNATURES = {
    "e4s_fused_bias_act_f32": [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_p],
    "e4s_upfirdn2d_f32": [c_p, c_p, c_p] + [c_i] * 14 + [c_p],
    "e4s_channel_sum_f32": [c_p, c_p, c_l, c_i, c_i, c_p],
    "e4s_rowdot_f32": [c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "e4s_weight_sqsum_f32": [c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_pack_taps_f32": [c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_polyphase_weights_f32": [c_p, c_p, c_p, c_i, c_i, c_p],
    "e4s_polyphase_fold_f32": [c_p, c_p, c_p, c_i, c_i, c_p],
    "e4s_rgb_weights_f32": [c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    "e4s_mask_labels": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_region_plan": [c_p] + [c_i] * 8 + [c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "e4s_conv_mfma_f32": [ctypes.POINTER(ConvParams), c_i, c_p],
    "e4s_upconv_mfma_f32": [ctypes.POINTER(ConvParams), c_p, c_p],
    "e4s_conv_bf16x3_f32": [ctypes.POINTER(ConvParams), c_p],
    "e4s_conv_bf16x3_ws_floats": [ctypes.POINTER(ConvParams)],
    "e4s_conv_mfma_ws_floats": [ctypes.POINTER(ConvParams), c_i],
    "e4s_split_bf16x2_f32": [c_p, c_p, c_l, c_i, c_p],
    "e4s_conv_region_bf16x3_f32": [ctypes.POINTER(ConvParams), c_p, c_p],
    "e4s_conv_region_ws_floats": [ctypes.POINTER(ConvParams)],
    "e4s_conv_region_path": [ctypes.POINTER(ConvParams)],
    "e4s_split16_bytes": [ctypes.c_int64, c_i, c_i],
    "e4s_split16_bf16x2_f32": [c_p, c_p, c_l, c_i, c_i, c_p],
    "e4s_upconv_blocks_per_cu": [],
    "e4s_instnorm_ws_doubles": [c_i, c_i, c_i],
    "e4s_conv_bwd_mfma_f32": [ctypes.POINTER(ConvBwdParams), c_p],
    "e4s_pack_taps_bwd_f32": [c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_demod_grad_f32": [c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_f, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_torgb_bwd_w_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_conv_bwd_ws_floats": [ctypes.POINTER(ConvBwdParams)],
    "e4s_seg_reduce_nsplit": [c_i, c_i, c_i, c_i],
    "e4s_grouped_linear_t_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_f, c_p],
    "e4s_grouped_linear_t_ws_floats": [c_i, c_i, c_i, c_i],
    "e4s_reduce_parts_f32": [c_p, c_p, c_i, c_l, c_f, c_p],
    "e4s_reduce_parts_ws_floats": [c_i, c_l],
    "e4s_grouped_outer_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "e4s_style_grad_multi_f32": [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_swap_styles_f32": [c_p, c_p, c_p, c_i, c_i, c_i, ctypes.c_uint, c_i, c_i, c_i, c_p],
    "e4s_batch_sum_f32": [c_p, c_p, c_i, c_l, c_p],
    "e4s_adam_step_f32": [c_p, c_p, c_p, c_p, c_l, c_d, c_d, c_d, c_d, c_d, c_i, c_p],
    "e4s_subpixel_weights_f32": [c_p, c_p, c_i, c_i, c_p],
    "e4s_upconv_bf16x3_f32": [c_p, c_p, c_p],
    "e4s_colsum_f32": [c_p, c_p, c_p, c_l, c_i, c_p],
    "e4s_colsum_ws_floats": [c_l, c_i],
    "e4s_scale_dot_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_p],
    "e4s_scale_dot_ws_floats": [c_i, c_l, c_i],
    "e4s_mask_to_u8": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_erode_u8": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_gaussian_blur_u8": [c_p, c_p, c_i, c_i, c_i, c_i, ctypes.POINTER(c_i), c_p],
    "e4s_alpha_composite_u8": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_pyrdown_u8": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_pyrdown_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_pyrup_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_u8_to_f32": [c_p, c_p, c_l, c_p],
    "e4s_lap_level_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_p],
    "e4s_clip_u8": [c_p, c_p, c_l, c_p],
    "e4s_conv_c32_bf16x3_f32": [c_p, c_p, c_p, c_p],
    "e4s_torgb_finish_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_rowdot_multi_f32": [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_ema_f32": [c_p, c_p, c_l, c_d, c_p],
    "e4s_adam_step_dev_f32": [c_p, c_p, c_p, c_p, c_l, c_d, c_p, c_d, c_d, c_d, c_d, c_p, c_p],
    "e4s_advance_i64": [c_p, c_l, c_p],
    "e4s_conv_smallcin_col2im_f32": [c_p, c_p] + [c_i] * 10 + [c_p],
    "e4s_region_scale_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_col2im_region_nsplit": [c_i, c_i, c_i, c_i],
    "e4s_col2im_region_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_act_bwd_demod_nsplit": [c_i, c_i, c_i, c_i],
    "e4s_act_bwd_demod_f32": [c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_f, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_pixel_unshuffle2_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_wino_weights_bytes": [c_i, c_i],
    "e4s_wino_weights_f32": [c_p, c_p, c_i, c_i, c_p],
    "e4s_conv_wino_covers": [c_p],
    "e4s_conv_wino_ws_floats": [c_p],
    "e4s_conv_wino_bf16x3_f32": [c_p, c_p],
    "e4s_adam_multi_dev_f32": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_d, c_p, c_d, c_d, c_d, c_d, c_p],
    "e4s_ema_multi_f32": [c_i, c_p, c_p, c_p, c_d, c_p],
    "e4s_torgb_bwd_x_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_shift_scale_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p] + [c_i] * 12 + [c_p],
    "e4s_torgb_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_mask_mul_add_f32": [c_p, c_p, c_p] + [c_i] * 10 + [c_p],
    "e4s_noise_bias_act_nhwc_f32": [c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p],
    "e4s_nchw_to_nhwc_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_nhwc_to_nchw_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_const_input_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_resize_bilinear_f32": [c_p, c_p] + [c_i] * 6 + [c_p],
    "e4s_conv3x3_small_f32": [c_p, c_p, c_p] + [c_i] * 5 + [c_p],
    "e4s_conv3x3_stem_f32": [c_p, c_p, c_p, c_p] + [c_i] * 5 + [c_p],
    "e4s_instnorm_stats_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "e4s_instnorm_finalize_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "e4s_instnorm_apply_stats_f32": [c_p] * 9 + [c_i] * 5 + [c_p],
    "e4s_instnorm_apply_f32": [c_p] * 7 + [c_i] * 5 + [c_p],
    "e4s_se_gate_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_instnorm_finalize_se_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "e4s_region_mean_f32": [c_p, c_p, c_i, c_i, c_p] + [c_i] * 7 + [c_p],
    "e4s_conv_wgrad_f32": [ctypes.POINTER(ConvWgradParams), c_p],
    "e4s_conv_wgrad_ws_floats": [ctypes.POINTER(ConvWgradParams)],
    "e4s_conv_wgrad_path": [ctypes.POINTER(ConvWgradParams)],
    "e4s_instnorm_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_instnorm_bwd_ws_doubles": [c_i, c_i, c_i],
    "e4s_prelu_f32": [c_p, c_p, c_p, c_l, c_i, c_p],
    "e4s_prelu_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_p],
    "e4s_prelu_bwd_ws_floats": [c_l, c_i],
    "e4s_strided_scatter_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_strided_place_f32": [c_p, c_p] + [c_i] * 9 + [c_p],
    "e4s_region_mean_bwd_f32": [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_conv1x1_small_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_f, c_p],
    "e4s_noise_half_f32": [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_f, c_f, c_p],
    "e4s_pixelnorm_f32": [c_p, c_p, c_i, c_i, c_p],
    "e4s_add_scale_f32": [c_p, c_p, c_p, c_f, c_l, c_p],
    "e4s_minibatch_stddev_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_onehot_u8_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_swap_head_mask_u8": [c_p, c_p, c_p, c_p, c_l, c_p],
    "e4s_foreground_mask_f32": [c_p, c_p, c_p, c_l, c_p],
    "e4s_morph_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_create_masks_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_tensor2im_u8": [c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_stream_copy_u8": [c_p, c_p, ctypes.c_int64, c_i, c_p],
    "e4s_paste_u8": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "e4s_grouped_linear_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_p],
    "e4s_adaptive_pool_f32": [c_p, c_p] + [c_i] * 11 + [c_p, c_p, c_p],
    "e4s_adaptive_pool_bwd_f32": [c_p, c_p] + [c_i] * 11 + [c_p, c_i, c_p],
    "e4s_conv_smallcin_f32": [c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_p],
    "e4s_conv_smallcin_bwd_f32": [c_p, c_p, c_p] + [c_i] * 10 + [c_p],
    "e4s_maxpool3s2_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_maxpool3s2_bwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_maxpool2_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_maxpool2_bwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_relu_bwd_f32": [c_p, c_p, c_p, c_l, c_i, c_p],
    "e4s_lpips_layer_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_lpips_layer_ws_doubles": [c_i, c_i],
    "e4s_lpips_layer_bwd_f32": [c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_instnorm_bwd_sums_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "e4s_norm_bwd_frozen_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "e4s_cosine_f32": [c_p, c_p, c_p, c_p, c_i, c_l, c_p],
    "e4s_cosine_ws_doubles": [c_i, c_l],
    "e4s_cosine_bwd_f32": [c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_l, c_i, c_p],
}

INT64_RETURN = {"e4s_split16_bytes", "e4s_instnorm_ws_doubles", "e4s_conv_bwd_ws_floats", "e4s_grouped_linear_t_ws_floats", "e4s_reduce_parts_ws_floats", "e4s_instnorm_bwd_ws_doubles", "e4s_prelu_bwd_ws_floats", "e4s_conv_wgrad_ws_floats", "e4s_conv_bf16x3_ws_floats", "e4s_conv_region_ws_floats", "e4s_lpips_layer_ws_doubles", "e4s_conv_mfma_ws_floats",
                "e4s_cosine_ws_doubles", "e4s_colsum_ws_floats", "e4s_scale_dot_ws_floats", "e4s_wino_weights_bytes", "e4s_conv_wino_ws_floats"}       # size queries: return a count, not an error code

_lib = None


def load():
    """dlopen libe4s_hip.so (built in-tree by e4s_amd.build / __graft_entry__.build). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -m e4s_amd.build` (hipcc, gfx950). "
                           "e4s_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.e4s_abi_version.restype = c_i
    lib.e4s_build_arch.restype = ctypes.c_char_p
    if lib.e4s_abi_version() != ABI_VERSION:
        raise RuntimeError("libe4s_hip.so ABI version mismatch: rebuild with `python -m e4s_amd.build --force`")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_l if name in INT64_RETURN else c_i
    _lib = lib
    return lib


def stream():
    return c_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous fp32/uint8/int32/f64 ROCm tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("e4s_amd kernels need tensors on a ROCm device (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("e4s_amd kernels need contiguous tensors")
    return c_p(t.data_ptr())


def fptr(t):
    if t is not None and t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    return ptr(t)


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed: hipError {code}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)
