"""ctypes binding of libcrnerf_hip.so (the C ABI declared in include/crnerf.h).

The library is the product: if it is missing or fails to load, every call raises -- there is no
eager/CPU fallback behind these functions.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRNERF_LIB_PATH: a tuning build of the same library (tools/variants.py) -- never a different backend; a path that does not exist raises at load
LIB_PATH = os.environ.get("CRNERF_LIB_PATH") or os.path.join(_HERE, "libcrnerf_hip.so")

EXPORTS = [
    "crnerf_abi_version", "crnerf_last_error", "crnerf_packed_mlp_bytes", "crnerf_pack_mlp_weights",
    "crnerf_posenc_f32", "crnerf_embed_points_f32", "crnerf_mlp_forward_f32", "crnerf_composite_f32", "crnerf_composite_backward_f32", "crnerf_sample_pdf_merge_f32",
    "crnerf_render_rays_f32", "crnerf_render_rays_train_f32", "crnerf_rng_fill_f32", "crnerf_mlp_backward_ex_f32", "crnerf_packed_mlp_mixed_bytes",
    "crnerf_pack_mlp_weights_mixed", "crnerf_mlp_train_mixed_acts_bytes", "crnerf_mlp_train_mixed_scratch_bytes", "crnerf_mlp_forward_train_mixed_f32", "crnerf_mlp_backward_mixed_f32", "crnerf_mlp_backward_mixed_ex_f32", "crnerf_render_rays_train_bf16", "crnerf_crossray_workspace_bytes", "crnerf_crossray_chansum_f32",
    "crnerf_crossray_gram_f32", "crnerf_crossray_matrix_f32", "crnerf_crossray_fold_f32",
    "crnerf_crossray_apply_f32", "crnerf_crossray_decode_f32",
    "crnerf_packed_mlp_t_bytes", "crnerf_pack_mlp_weights_t", "crnerf_mlp_train_acts_bytes", "crnerf_mlp_train_scratch_bytes",
    "crnerf_mlp_forward_train_f32", "crnerf_mlp_backward_f32",
    "crnerf_ray_directions_f32", "crnerf_rays_from_directions_f32", "crnerf_generate_rays_f32",
    "crnerf_encoder_workspace_bytes", "crnerf_encoder_forward_f32",
    "crnerf_crossray_backward_workspace_bytes", "crnerf_crossray_decode_backward_f32", "crnerf_crossray_decode_sharded_f32",
    "crnerf_crossray_decode_backward_sharded_f32",
    "crnerf_packed_mlp_h2_bytes", "crnerf_pack_mlp_weights_h2", "crnerf_mlp_forward_f32h2", "crnerf_render_rays_f32h2",
    "crnerf_render_rays_f32x3_repair", "crnerf_mlp_forward_f32x3_repair",
    "crnerf_render_rays_train_f32h2", "crnerf_render_rays_train_f32x3_repair", "crnerf_packed_mlp_t_h2_bytes", "crnerf_pack_mlp_weights_t_h2", "crnerf_mlp_backward_h2_f32", "crnerf_pack_mlp_weights_h2_async", "crnerf_pack_h2_status",
    "crnerf_packed_mlp_x3_bytes", "crnerf_pack_mlp_weights_x3", "crnerf_mlp_forward_f32x3", "crnerf_render_rays_f32x3", "crnerf_render_rays_train_f32x3", "crnerf_packed_mlp_t_x3_bytes", "crnerf_pack_mlp_weights_t_x3", "crnerf_mlp_backward_x3_f32", "crnerf_packed_mlp_bf16_bytes", "crnerf_pack_mlp_weights_bf16", "crnerf_mlp_forward_bf16", "crnerf_render_rays_bf16", "crnerf_render_rays_bf16_fine",
    "crnerf_decoder_content_backward_workspace_bytes", "crnerf_decoder_content_backward_f32",
    "crnerf_encoder_train_saved_bytes", "crnerf_encoder_train_scratch_bytes", "crnerf_encoder_forward_train_f32", "crnerf_encoder_backward_f32",
    "crnerf_encoder_train_band_saved_bytes", "crnerf_encoder_train_band_scratch_bytes", "crnerf_encoder_forward_train_band_f32", "crnerf_encoder_backward_band_f32",
    "crnerf_loss_workspace_bytes", "crnerf_loss_f32", "crnerf_loss_backward_f32", "crnerf_grid_sample_batch_f32", "crnerf_adam_max_tensors", "crnerf_adam_step_f32",
    "crnerf_conv2d_f32", "crnerf_conv2d_backward_f32", "crnerf_bn_prelu_f32", "crnerf_bn_prelu_train_f32", "crnerf_bn_prelu_backward_f32", "crnerf_avgpool3s2_f32",
    "crnerf_fglo_f32", "crnerf_fglo_backward_f32", "crnerf_bilinear_gather_f32", "crnerf_bilinear_gather_backward_f32",
    "crnerf_cgnet_param_count", "crnerf_cgnet_bn_count", "crnerf_cgnet_arena_bytes", "crnerf_cgnet_forward_train_f32", "crnerf_cgnet_backward_f32",
    "crnerf_peer_window_bytes", "crnerf_peer_window_create", "crnerf_peer_window_open", "crnerf_peer_window_close", "crnerf_peer_window_destroy",
    "crnerf_peer_window_status", "crnerf_peer_allreduce_f32",
    "crnerf_cus_per_xcd", "crnerf_stream_create_cu_share", "crnerf_stream_destroy",
]

_c_fp = ctypes.c_void_p  # device float*


class RenderArgs(ctypes.Structure):
    """struct crnerf_render_args (include/crnerf.h)."""
    _fields_ = [
        ("packed_coarse", ctypes.c_void_p), ("packed_fine", ctypes.c_void_p),
        ("rays", _c_fp), ("view_dir", _c_fp), ("z_coarse", _c_fp), ("z_steps", _c_fp), ("u", _c_fp), ("u_stride", ctypes.c_int64),
        ("noise_coarse", _c_fp), ("noise_fine", _c_fp),
        ("noise_std", ctypes.c_float), ("use_disp", ctypes.c_int32),
        ("n_rays", ctypes.c_int64), ("n_samples", ctypes.c_int32), ("n_importance", ctypes.c_int32),
        ("weights_coarse", _c_fp), ("feature_coarse", _c_fp), ("depth_coarse", _c_fp),
        ("weights_fine", _c_fp), ("feature_fine", _c_fp), ("depth_fine", _c_fp), ("z_fine", _c_fp),
        ("rng_seed", ctypes.c_uint64), ("rng_ray_offset", ctypes.c_int64), ("rng_flags", ctypes.c_int32), ("perturb", ctypes.c_float),
        ("z_coarse_out", _c_fp), ("noise_coarse_out", _c_fp), ("noise_fine_out", _c_fp),
    ]


RNG_JITTER, RNG_U, RNG_NOISE = 1, 2, 4     # CRNERF_RNG_* (include/crnerf.h)
ERR_RANGE = -4                              # CRNERF_ERR_RANGE


class LossArgs(ctypes.Structure):
    """struct crnerf_loss_args (include/crnerf.h)."""
    _fields_ = [
        ("rgb_coarse", _c_fp), ("rgb_coarse_row_stride", ctypes.c_int64), ("rgb_coarse_chan_stride", ctypes.c_int64),
        ("rgb_fine", _c_fp), ("rgb_fine_row_stride", ctypes.c_int64), ("rgb_fine_chan_stride", ctypes.c_int64),
        ("targets", _c_fp), ("targets_row_stride", ctypes.c_int64), ("targets_chan_stride", ctypes.c_int64),
        ("mask", _c_fp),
        ("a_embedded", _c_fp), ("n_a", ctypes.c_int64),
        ("a_embedded_random", _c_fp), ("a_embedded_random_rec", _c_fp), ("n_rec", ctypes.c_int64),
        ("content_wo", _c_fp), ("content_with", _c_fp), ("n_content", ctypes.c_int64),
        ("n_rays", ctypes.c_int64),
        ("mse_on_appearance", ctypes.c_int32),
        ("coef", ctypes.c_float), ("weight_kl", ctypes.c_float), ("weight_rec_a", ctypes.c_float), ("weight_content", ctypes.c_float),
        ("mask_size_weight", ctypes.c_float), ("mask_digit_weight", ctypes.c_float),
    ]


class LossGrads(ctypes.Structure):
    """struct crnerf_loss_grads."""
    _fields_ = [(n, _c_fp) for n in ("d_rgb_coarse", "d_rgb_fine", "d_mask", "d_a_embedded", "d_a_embedded_random_rec",
                                     "d_content_wo", "d_content_with")]


class BatchArgs(ctypes.Structure):
    """struct crnerf_batch_args."""
    _fields_ = [
        ("all_rays", _c_fp), ("ray_stride", ctypes.c_int64), ("all_rgbs", _c_fp), ("row_offset", ctypes.c_int64),
        ("img_w", ctypes.c_int32), ("img_h", ctypes.c_int32), ("side", ctypes.c_int32),
        ("w_lin", _c_fp), ("h_lin", _c_fp),
        ("scale", ctypes.c_float), ("h_offset", ctypes.c_float), ("w_offset", ctypes.c_float),
        ("rays", _c_fp), ("ts", ctypes.c_void_p), ("rgbs", _c_fp), ("rgb_idx", ctypes.c_void_p), ("uv_sample", _c_fp),
    ]


class ConvGeom(ctypes.Structure):
    """struct crnerf_conv_geom."""
    _fields_ = [(n, ctypes.c_int32) for n in ("cin", "cout", "H", "W", "k", "stride", "pad", "dil", "depthwise")]


_lib = None
_lock = threading.Lock()


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "crnerf_amd: %s not found -- build it with `python cr-nerf-pytorch_amd/build.py` "
                "(or __graft_entry__.build()); there is no fallback path." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        i64, i32, vp, f32, f64 = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_double
        pp = ctypes.POINTER(ctypes.c_void_p)
        sig = {
            "crnerf_abi_version": (ctypes.c_int, []),
            "crnerf_last_error": (ctypes.c_char_p, []),
            "crnerf_packed_mlp_bytes": (ctypes.c_size_t, []),
            "crnerf_crossray_workspace_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_packed_mlp_t_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights_t": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_mlp_train_acts_bytes": (ctypes.c_size_t, [i64]),
            "crnerf_mlp_train_scratch_bytes": (ctypes.c_size_t, [i64]),
            "crnerf_mlp_forward_train_f32": (ctypes.c_int, [vp, vp, vp, vp, i64, vp]),
            "crnerf_mlp_backward_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, pp, i64, vp]),
            "crnerf_mlp_backward_ex_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, pp, i64, i32, vp]),
            "crnerf_packed_mlp_mixed_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights_mixed": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_mlp_train_mixed_acts_bytes": (ctypes.c_size_t, [i64]),
            "crnerf_mlp_train_mixed_scratch_bytes": (ctypes.c_size_t, [i64]),
            "crnerf_mlp_forward_train_mixed_f32": (ctypes.c_int, [pp, vp, vp, vp, vp, i64, vp]),
            "crnerf_mlp_backward_mixed_f32": (ctypes.c_int, [pp, vp, vp, vp, vp, vp, vp, pp, i64, vp]),
            "crnerf_mlp_backward_mixed_ex_f32": (ctypes.c_int, [pp, vp, vp, vp, vp, vp, pp, i64, ctypes.c_int, vp]),
            "crnerf_posenc_f32": (ctypes.c_int, [vp, vp, i64, i32, vp]),
            "crnerf_embed_points_f32": (ctypes.c_int, [vp, vp, vp, vp, i64, i32, vp]),
            "crnerf_encoder_workspace_bytes": (ctypes.c_size_t, [i32, i32]),
            "crnerf_encoder_forward_f32": (ctypes.c_int, [vp, i32, i32, pp, vp, vp, vp]),
            "crnerf_ray_directions_f32": (ctypes.c_int, [i32, i32, f32, f32, f32, f32, vp, vp]),
            "crnerf_rays_from_directions_f32": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_float), i64, vp, vp, vp]),
            "crnerf_generate_rays_f32": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), i32, i32, f32, f32, vp, vp]),
            "crnerf_mlp_forward_f32": (ctypes.c_int, [vp, vp, vp, i64, i32, vp]),
            "crnerf_composite_f32": (ctypes.c_int, [vp, vp, vp, f32, vp, vp, vp, i64, i32, vp]),
            "crnerf_composite_backward_f32": (ctypes.c_int, [vp, vp, vp, f32, vp, vp, vp, vp, i64, i32, vp]),
            "crnerf_sample_pdf_merge_f32": (ctypes.c_int, [vp, vp, vp, i64, vp, vp, i64, i32, i32, vp]),
            "crnerf_render_rays_f32": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp]),
            "crnerf_rng_fill_f32": (ctypes.c_int, [vp, i64, i32, ctypes.c_uint64, i32, i64, vp]),
            "crnerf_render_rays_train_f32": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp, vp, vp, vp, vp]),
            "crnerf_render_rays_train_bf16": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp, vp, vp, vp, vp]),
            "crnerf_render_rays_bf16": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp]),
            "crnerf_render_rays_bf16_fine": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp]),
            "crnerf_packed_mlp_h2_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights_h2": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_mlp_forward_f32h2": (ctypes.c_int, [vp, vp, vp, i64, i32, vp]),
            "crnerf_render_rays_f32h2": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp]),
            "crnerf_render_rays_f32x3_repair": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp]),
            "crnerf_mlp_forward_f32x3_repair": (ctypes.c_int, [vp, vp, vp, i64, i32, vp]),
            "crnerf_render_rays_train_f32h2": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp, vp, vp, vp, vp]),
            "crnerf_render_rays_train_f32x3_repair": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp, vp, vp, vp, vp]),
            "crnerf_packed_mlp_t_h2_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights_t_h2": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_mlp_backward_h2_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, pp, i64, ctypes.c_int, vp]),
            "crnerf_pack_mlp_weights_h2_async": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_pack_h2_status": (ctypes.c_int, [vp, vp]),
            "crnerf_packed_mlp_x3_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights_x3": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_mlp_forward_f32x3": (ctypes.c_int, [vp, vp, vp, i64, i32, vp]),
            "crnerf_render_rays_f32x3": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp]),
            "crnerf_render_rays_train_f32x3": (ctypes.c_int, [ctypes.POINTER(RenderArgs), vp, vp, vp, vp, vp]),
            "crnerf_packed_mlp_t_x3_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights_t_x3": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_mlp_backward_x3_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, pp, i64, ctypes.c_int, vp]),
            "crnerf_packed_mlp_bf16_bytes": (ctypes.c_size_t, []),
            "crnerf_pack_mlp_weights_bf16": (ctypes.c_int, [pp, vp, vp]),
            "crnerf_mlp_forward_bf16": (ctypes.c_int, [vp, vp, vp, i64, i32, vp]),
            "crnerf_crossray_chansum_f32": (ctypes.c_int, [vp, i64, vp, vp, vp]),
            "crnerf_crossray_gram_f32": (ctypes.c_int, [vp, i64, vp, pp, vp, vp, vp]),
            "crnerf_crossray_matrix_f32": (ctypes.c_int, [vp, f64, vp, vp, vp, vp]),
            "crnerf_crossray_fold_f32": (ctypes.c_int, [vp, vp, vp, vp, pp, vp, vp]),
            "crnerf_crossray_apply_f32": (ctypes.c_int, [vp, i64, vp, vp, i64, vp]),
            "crnerf_crossray_backward_workspace_bytes": (ctypes.c_size_t, [i64, i64]),
            "crnerf_crossray_decode_backward_f32": (ctypes.c_int, [vp, i64, vp, i64, pp, vp, i64, vp, vp, vp, pp, vp]),
            "crnerf_crossray_decode_sharded_f32": (ctypes.c_int, [vp, i64, vp, i64, pp, i32, vp, f64, vp, vp, i64, vp]),
            "crnerf_crossray_decode_backward_sharded_f32": (ctypes.c_int, [vp, i64, vp, i64, pp, vp, i64, vp, vp, vp, pp, i32, vp, f64, vp, vp]),
            "crnerf_crossray_decode_f32": (ctypes.c_int, [vp, i64, vp, i64, pp, vp, vp, i64, vp]),
            "crnerf_decoder_content_backward_workspace_bytes": (ctypes.c_size_t, [i64]),
            "crnerf_decoder_content_backward_f32": (ctypes.c_int, [vp, i64, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp]),
            "crnerf_encoder_train_saved_bytes": (ctypes.c_size_t, [i32, i32]),
            "crnerf_encoder_train_scratch_bytes": (ctypes.c_size_t, [i32, i32]),
            "crnerf_encoder_forward_train_f32": (ctypes.c_int, [vp, i32, i32, pp, vp, vp, vp]),
            "crnerf_encoder_backward_f32": (ctypes.c_int, [i32, i32, pp, vp, vp, vp, vp, pp, vp, vp]),
            "crnerf_encoder_train_band_saved_bytes": (ctypes.c_size_t, [i32, i32, i32]),
            "crnerf_encoder_train_band_scratch_bytes": (ctypes.c_size_t, [i32, i32, i32]),
            "crnerf_encoder_forward_train_band_f32": (ctypes.c_int, [vp, i32, i32, i32, i32, i32, i32, pp, vp, vp, vp]),
            "crnerf_encoder_backward_band_f32": (ctypes.c_int, [i32, i32, i32, i32, i32, i32, pp, vp, vp, vp, vp, pp, vp, vp]),
            "crnerf_loss_workspace_bytes": (ctypes.c_size_t, []),
            "crnerf_loss_f32": (ctypes.c_int, [ctypes.POINTER(LossArgs), vp, vp, vp]),
            "crnerf_loss_backward_f32": (ctypes.c_int, [ctypes.POINTER(LossArgs), vp, ctypes.POINTER(LossGrads), vp]),
            "crnerf_grid_sample_batch_f32": (ctypes.c_int, [ctypes.POINTER(BatchArgs), vp]),
            "crnerf_adam_max_tensors": (ctypes.c_int, []),
            "crnerf_adam_step_f32": (ctypes.c_int, [vp, vp, vp, vp, i32, pp, i32, f32, f32, f32, f32, f32, f32, vp]),
            "crnerf_conv2d_f32": (ctypes.c_int, [ctypes.POINTER(ConvGeom), vp, vp, vp, vp]),
            "crnerf_conv2d_backward_f32": (ctypes.c_int, [ctypes.POINTER(ConvGeom), vp, vp, vp, vp, vp, vp]),
            "crnerf_bn_prelu_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, f32, i32, vp]),
            "crnerf_bn_prelu_train_f32": (ctypes.c_int, [vp] * 11 + [f32, i32, i64, f32, vp]),
            "crnerf_bn_prelu_backward_f32": (ctypes.c_int, [vp] * 11 + [i32, i64, i32, vp]),
            "crnerf_avgpool3s2_f32": (ctypes.c_int, [vp, vp, i32, i32, i32, i32, vp]),
            "crnerf_fglo_f32": (ctypes.c_int, [vp] * 7 + [i32, i32, i64, vp]),
            "crnerf_fglo_backward_f32": (ctypes.c_int, [vp] * 11 + [i32, i32, i64, vp]),
            "crnerf_bilinear_gather_f32": (ctypes.c_int, [vp, i32, i32, i32, i32, vp, i64, i32, vp, vp]),
            "crnerf_bilinear_gather_backward_f32": (ctypes.c_int, [vp, vp, i32, i32, i32, i32, vp, i64, i32, vp, vp]),
            "crnerf_cgnet_param_count": (ctypes.c_int, []),
            "crnerf_cgnet_bn_count": (ctypes.c_int, []),
            "crnerf_cgnet_arena_bytes": (ctypes.c_size_t, [i32, i32, i32]),
            "crnerf_cgnet_forward_train_f32": (ctypes.c_int, [vp, i32, i32, i32, pp, pp, pp, pp, f32, f32, vp, vp, vp]),
            "crnerf_cgnet_backward_f32": (ctypes.c_int, [vp, i32, i32, i32, pp, vp, vp, vp, vp, pp, vp]),
            "crnerf_peer_window_bytes": (ctypes.c_size_t, []),
            "crnerf_peer_window_create": (ctypes.c_int, [pp, ctypes.c_char_p]),
            "crnerf_peer_window_open": (ctypes.c_int, [ctypes.c_char_p, pp]),
            "crnerf_peer_window_close": (ctypes.c_int, [vp]),
            "crnerf_peer_window_destroy": (ctypes.c_int, [vp]),
            "crnerf_peer_window_status": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int)]),
            "crnerf_peer_allreduce_f32": (ctypes.c_int, [vp, i32, pp, i32, i32, ctypes.c_uint32, i64, vp]),
            "crnerf_cus_per_xcd": (ctypes.c_int, []),
            "crnerf_stream_create_cu_share": (ctypes.c_int, [pp, i32, i32]),
            "crnerf_stream_destroy": (ctypes.c_int, [vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)  # AttributeError here = the library does not match include/crnerf.h
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().crnerf_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, code, msg.decode() if msg else "?"))


# Host cost matters at the reference's 1,024-ray training batch (~450 launches per step, the step is host-bound): these helpers run once per
# pointer / launch, so they use torch's raw accessors -- no Stream object per launch, no string formatting unless something is wrong.
_get_device = getattr(torch._C, "_cuda_getDevice", None)
_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _current_device():
    return _get_device() if _get_device is not None else torch.cuda.current_device()


def stream_ptr():
    if _get_raw_stream is not None:
        return ctypes.c_void_p(_get_raw_stream(_current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bad_tensor(t, name, dtype):
    if not t.is_cuda:
        return RuntimeError("crnerf_amd: %s must live on the GPU (got %s); the HIP path has no CPU fallback" % (name, t.device))
    if t.device.index != _current_device():
        # kernels are enqueued on the CURRENT device's stream (stream_ptr): a tensor of another GPU would be reached through peer
        # access, unordered with its producers -- one process drives one GPU (torch.cuda.set_device(LOCAL_RANK) first)
        return RuntimeError("crnerf_amd: %s lives on cuda:%d but the current device is cuda:%d; call torch.cuda.set_device(%d) "
                            "(or wrap the call in torch.cuda.device) before using the HIP path"
                            % (name, t.device.index, _current_device(), t.device.index))
    if t.dtype != dtype:
        return TypeError("crnerf_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    return ValueError("crnerf_amd: %s must be contiguous" % name)


def dev_ptr(t, name="tensor", dtype=torch.float32):
    """Pointer of a contiguous device tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous() or t.device.index != _current_device():
        raise _bad_tensor(t, name, dtype)
    return ctypes.c_void_p(t.data_ptr())


_PTR_ARRAYS = {}      # tuple of device addresses -> the ctypes array that holds them


def ptr_array(tensors, name):
    """HOST array of the tensors' device pointers (every `const float* const*` of the ABI).  A training step builds ~40 of them, mostly for the
    same parameter lists (their addresses are stable: optim.FlatAdam keeps p.data a view of one flat buffer) and for gradient lists the caching
    allocator hands back at the same addresses step after step: the arrays are kept by address tuple (round 6: 13 us -> 3 us per call; the
    1,024-ray step is host-bound).  The tensors are validated when an array is built; a hit means the same addresses passed that before."""
    key = tuple([t.data_ptr() for t in tensors])
    arr = _PTR_ARRAYS.get(key)
    if arr is not None:
        return arr
    arr = (ctypes.c_void_p * len(tensors))()
    dev = _current_device()
    for i, t in enumerate(tensors):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.device.index != dev:
            raise _bad_tensor(t, "%s[%d]" % (name, i), torch.float32)
        arr[i] = t.data_ptr()
    if len(_PTR_ARRAYS) >= 4096:
        _PTR_ARRAYS.clear()
    _PTR_ARRAYS[key] = arr
    return arr
