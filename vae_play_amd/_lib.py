"""ctypes binding of libvaeplay_hip.so (C ABI: include/vaeplay_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the
product path raises.  PyTorch is used only for device memory (``data_ptr()``) and the
current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAEPLAY_HIP_LIB", os.path.join(_HERE, "libvaeplay_hip.so"))   # override: A/B of two builds

P = c_void_p  # device pointers / stream


class PackJob(ctypes.Structure):
    """vp_pack_job of include/vaeplay_hip.h (one weight tensor of vp_pack_w5_batch)."""
    _fields_ = [("w", c_void_p), ("p0", c_void_p), ("p1", c_void_p), ("Csmall", c_int), ("Cbig", c_int),
                ("Csmall_pad", c_int), ("split", c_int)]


# name -> (restype, argtypes); mirrors include/vaeplay_hip.h one to one
SIGNATURES = {
    "vp_abi_version": (c_int, []),
    "vp_last_error": (c_char_p, []),
    "vp_nchw_to_nhwc_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "vp_nhwc_to_nchw_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "vp_pack_w5_f32": (c_int, [P, P, P, c_int, c_int, P]),
    "vp_conv5_gather_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv5_scatter_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv5_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "vp_conv5_wgrad_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "vp_conv5_wgrad_f32_cus": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "vp_pack_w_f32": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "vp_conv_gather_f32": (c_int, [P, P, P, P] + [c_int] * 10 + [P]),
    "vp_conv_scatter_f32": (c_int, [P, P, P] + [c_int] * 9 + [P]),
    "vp_conv_wgrad_workspace_bytes": (c_size_t, [c_int] * 9),
    "vp_conv_wgrad_f32": (c_int, [P, P, P] + [c_int] * 9 + [P, c_size_t, P]),
    "vp_split_f32": (c_int, [P, P, c_size_t, P]),
    "vp_split_pad_f32": (c_int, [P, P, c_size_t, c_int, c_int, P]),
    "vp_pack_w5_split": (c_int, [P, P, P, c_int, c_int, P]),
    "vp_pack_w_split": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "vp_conv_gather_bf16x3": (c_int, [P, P, P, P] + [c_int] * 10 + [P]),
    "vp_conv_scatter_bf16x3": (c_int, [P, P, P] + [c_int] * 9 + [P]),
    "vp_conv_wgrad_bf16x3_workspace_bytes": (c_size_t, [c_int] * 9),
    "vp_conv_wgrad_bf16x3": (c_int, [P, P, P] + [c_int] * 9 + [P, c_size_t, P]),
    "vp_conv5_gather_bf16x3": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv5_scatter_bf16x3": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv5_smallout_bf16x3": (c_int, [P, P, P, P] + [c_int] * 6 + [P]),
    "vp_conv5_smallout_wgrad_bf16x3_workspace_bytes": (c_size_t, [c_int] * 5),
    "vp_conv5_smallout_wgrad_bf16x3": (c_int, [P, P, P] + [c_int] * 5 + [P, c_size_t, P]),
    "vp_conv5_smallout_wgrad_f32_workspace_bytes": (c_size_t, [c_int] * 5),
    "vp_conv5_smallout_wgrad_f32": (c_int, [P, P, P] + [c_int] * 5 + [P, c_size_t, P]),
    "vp_im2col5s2_cols": (c_int, [c_int]),
    "vp_im2col5s2_split_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_pack_w_im2col5_split": (c_int, [P, P, c_int, c_int, P]),
    "vp_unpack_dw_im2col5_f32": (c_int, [P, P, c_int, c_int, P]),
    "vp_im2col5s2_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_pack_w_im2col5_f32": (c_int, [P, P, c_int, c_int, P]),
    "vp_conv5_stats_workspace_bytes": (c_size_t, [c_int] * 7),
    "vp_conv5_gather_stats_bf16x3": (c_int, [P, P, P] + [c_int] * 6 + [c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "vp_conv5_gather_stats_f32": (c_int, [P, P, P] + [c_int] * 6 + [c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "vp_conv5_scatter_stats_f32": (c_int, [P, P, P] + [c_int] * 6 + [c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "vp_conv5_stats_f32_workspace_bytes": (c_size_t, [c_int] * 7),
    "vp_conv5_scatter_stats_bf16x3": (c_int, [P, P, P] + [c_int] * 6 + [c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "vp_split_fmt_f32": (c_int, [P, P, c_size_t, c_int, c_float, P]),
    "vp_nchw_to_nhwc_split_fmt_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_bn_act_fwd_split_fmt_f32": (c_int, [P] * 7 + [c_int, c_int, c_int, c_float, c_int, P]),
    "vp_bn_act_bwd_split_fmt_f32": (c_int, [P] * 10 + [c_int, c_int, c_int, c_float, c_int, c_int, c_float, P, c_size_t, P]),
    "vp_bn_act_bwd_split_fmt_sat_f32": (c_int, [P] * 10 + [c_int, c_int, c_int, c_float, c_int, c_int, c_float, P, P, c_size_t, P]),
    "vp_im2col5s2_split_fmt_f32": (c_int, [P, P] + [c_int] * 6 + [P]),
    "vp_pack_w_im2col5_split_fmt": (c_int, [P, P, c_int, c_int, c_int, P]),
    "vp_conv5_gather_f16": (c_int, [P, P, P, P] + [c_int] * 8 + [c_float, P]),
    "vp_conv_gather_f16": (c_int, [P, P, P, P] + [c_int] * 11 + [c_float, P]),
    "vp_conv5_scatter_f16": (c_int, [P, P, P] + [c_int] * 7 + [c_float, P]),
    "vp_conv_scatter_f16": (c_int, [P, P, P] + [c_int] * 10 + [c_float, P]),
    "vp_conv5_wgrad_f16x2": (c_int, [P, P, P] + [c_int] * 6 + [c_float, P, c_size_t, P]),
    "vp_conv5_wgrad_f16x2_cus": (c_int, [P, P, P] + [c_int] * 6 + [c_float, c_int, P, c_size_t, P]),
    "vp_conv_wgrad_f16x2": (c_int, [P, P, P] + [c_int] * 9 + [c_float, P, c_size_t, P]),
    "vp_conv5_stats_f16_workspace_bytes": (c_size_t, [c_int] * 7),
    "vp_conv5_gather_stats_f16": (c_int, [P, P, P] + [c_int] * 7 + [c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "vp_conv5_scatter_stats_f16": (c_int, [P, P, P] + [c_int] * 7 + [c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "vp_conv5_smallin_fwd_bf16x3": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv5_smallin_dgrad_bf16x3": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv5_smallin_dgrad_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv5_wgrad_bf16x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "vp_conv5_wgrad_bf16x3": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "vp_conv5_wgrad_bf16x3_cus": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "vp_wgrad_slab_reduce_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv3_small_wgrad_workspace_bytes": (c_size_t, [c_int] * 5),
    "vp_conv3_small_wgrad_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "vp_conv3_small_fwd_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_conv3_small_dgrad_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_bn_small_fwd_f32": (c_int, [P, c_int, c_int, c_float, c_float, P, P, P, P, P, P, P, c_int, c_float, P]),
    "vp_bn_small_bwd_f32": (c_int, [P] * 9 + [c_int, c_int, c_int, c_float, c_int, P]),
    "vp_bn_act_fwd_split_f32": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, P]),
    "vp_bn_act_bwd_split_f32": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, P, c_size_t, P]),
    "vp_nchw_to_nhwc_split_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "vp_pack_w5_p1_split_padded": (c_int, [P, P, c_int, c_int, c_int, P]),
    "vp_pack_w5_batch": (c_int, [ctypes.POINTER(PackJob), c_int, P]),
    "vp_bce_sigmoid_bwd_pad_split_f32": (c_int, [P, P, c_float, P, P, c_size_t, c_int, c_int, P]),
    "vp_gemm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vp_gemm_f32": (c_int, [P, c_long, c_long, P, c_long, c_long, P, c_int, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "vp_colsum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vp_colsum_f32": (c_int, [P, P, c_int, c_int, P, c_size_t, P]),
    "vp_bn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vp_bn_stats_f32": (c_int, [P, c_int, c_int, c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "vp_bn_act_fwd_f32": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_float, P]),
    "vp_bn_act_bwd_f32": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, P, c_size_t, P]),
    "vp_instnorm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vp_instnorm_act_fwd_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, c_int, c_float, P, c_size_t, P]),
    "vp_instnorm_act_bwd_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "vp_instnorm_act_fwd_split_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, c_float, P, c_size_t, P]),
    "vp_instnorm_act_bwd_split_f32": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P, c_size_t, P]),
    "vp_act_fwd_f32": (c_int, [P, P, c_size_t, c_int, c_float, P]),
    "vp_act_bwd_from_y_f32": (c_int, [P, P, P, c_size_t, c_int, c_float, P]),
    "vp_upsample2x_bilinear_fwd_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "vp_upsample2x_bilinear_bwd_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "vp_add_coords_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vp_slice_channels_f32": (c_int, [P, P, c_size_t, c_int, c_int, P]),
    "vp_latent_fwd_f32": (c_int, [P, P, P, P, P, c_int, c_int, P]),
    "vp_latent_bwd_f32": (c_int, [P, P, P, P, P, c_float, P, P, c_int, c_int, P]),
    "vp_reduce_workspace_bytes": (c_size_t, [c_size_t]),
    "vp_bce_sum_f32": (c_int, [P, P, c_size_t, P, P, c_size_t, P]),
    "vp_bce_bwd_f32": (c_int, [P, P, P, c_float, P, c_size_t, P]),
    "vp_bce_sigmoid_bwd_f32": (c_int, [P, P, c_float, P, c_size_t, P]),
    "vp_sum_f32": (c_int, [P, c_size_t, P, P, c_size_t, P]),
    "vp_vae_loss_f32": (c_int, [P, P, c_size_t, P, c_int, P, P, P, c_float, P, c_size_t, P]),
    "vp_add_f32": (c_int, [P, P, P, c_size_t, P]),
    "vp_global_avgpool_fwd_f32": (c_int, [P, P, c_int, c_int, c_int, P]),
    "vp_global_avgpool_bwd_f32": (c_int, [P, P, c_int, c_int, c_int, P]),
    "vp_softmax_rows_fwd_f32": (c_int, [P, P, c_int, c_int, P]),
    "vp_softmax_rows_bwd_f32": (c_int, [P, P, P, c_int, c_int, P]),
    "vp_cross_entropy_fwd_f32": (c_int, [P, P, P, P, c_int, c_int, P]),
    "vp_cross_entropy_bwd_f32": (c_int, [P, P, P, P, c_int, c_int, P]),
    "vp_l1_mean_f32": (c_int, [P, P, c_size_t, P, P, c_size_t, P]),
    "vp_l1_mean_bwd_f32": (c_int, [P, P, P, P, P, c_size_t, P]),
    "vp_be_loss_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vp_be_loss_fwd_f32": (c_int, [P, P, P, P, c_int, c_int, c_float, c_float, P, c_size_t, P]),
    "vp_be_loss_bwd_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_float, c_float, P]),
    "vp_dice_loss_fwd_f32": (c_int, [P, P, P, P, c_int, c_int, c_float, P, c_size_t, P]),
    "vp_dice_loss_bwd_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_float, P]),
    "vp_half_sqdiff_f32": (c_int, [P, P, P, c_size_t, P]),
    "vp_half_sqdiff_rowsum_f32": (c_int, [P, P, P, c_int, c_int, P]),
    "vp_half_sqdiff_bwd_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "vp_gan_head_f32": (c_int, [P, c_int, c_float, P, P, P, P]),
    "vp_smooth_l1_cat_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P, P, P, P]),
    "vp_adam_f32": (c_int, [P, P, P, P, c_size_t, c_float, c_float, c_float, c_float, c_int, c_float, P]),
    "vp_adam_outer_f32": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, c_float, P]),
    "vp_rmsprop_f32": (c_int, [P, P, P, c_size_t, c_float, c_float, c_float, c_float, P]),
}

_lib = None


class VaePlayHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library; raise (never fall back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VaePlayHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vae_play_amd/csrc`. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().vp_last_error()
        raise VaePlayHipError(f"{what or 'vaeplay_hip'} failed (status {rc}): {msg.decode() if msg else ''}")


TRACE = None     # benches only (tools/op_roofline.py): {"events": []} brackets every call with a HIP event pair on the launch stream


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    tr = TRACE
    if tr is None:
        check(getattr(load(), name)(*args), name)
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = getattr(load(), name)(*args)
    e1.record()
    tr["events"].append((name, args, e0, e1))
    check(rc, name)
