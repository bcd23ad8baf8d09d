"""MI355X-native CR-NeRF rendering hot path.

Host side = PyTorch-ROCm (tensors, streams, torch.distributed); arithmetic = hand-written HIP for
gfx950 behind the C ABI of include/crnerf.h (libcrnerf_hip.so, built in-tree by build.py).

    from crnerf_amd.models.rendering import render_rays_cross_ray
    from crnerf_amd.models.nerf import PosEmbedding, NeRF_sigma
    from crnerf_amd.models.linearStyleTransfer import style_net

mirror the reference's models/rendering.py, models/nerf.py, models/linearStyleTransfer.py.
There is no CPU or eager-PyTorch fallback: without the HIP library every compute call raises.
"""
from . import _lib  # noqa: F401  (does not load the shared object until first use)

__all__ = ["_lib", "set_precision", "get_precision"]

_precision = "f32"


def set_precision(precision):
    """Default arithmetic of NeRF_sigma inside the inference paths: "f32" (exact fp32 MFMA, the reference's
    numerics), "bf16" (bf16 matrix cores, fp32 accumulate -- include/crnerf.h "bf16 variants") or "f32x3" (fp32 on the bf16
    matrix cores: three-piece splits of every fp32 operand, six MFMAs per product -- meets the fp32 tolerances, 1.6x the fp32
    kernel's speed; include/crnerf.h "f32x3") or "f32h2" (fp32 on the fp16 matrix cores: two-piece splits, three MFMAs per product --
    meets the fp32 tolerances at 2.8x the fp32 kernel's speed WHILE weights and activations fit fp16's range: |w| < 255 is checked when the
    weights are packed, an activation >= 65,504 turns its point into NaN; include/crnerf.h "f32h2"), or "auto": f32h2 made safe -- a pack the h2
    core refuses runs on f32x3, a ray (a 128-point group) the h2 core poisoned is re-rendered on f32x3 inside the same call, so that no NaN of
    the h2 core's making reaches the caller (include/crnerf.h "auto"), or "bf16_hc": bf16 with an fp32-accurate COARSE pass ("auto" arithmetic on the
    coarse network's 25 % of the points, the fine network on the bf16 matrix cores: the fine depths then follow the fp32 reference;
    models/rendering.py::_render_bf16_accurate_coarse).  A `precision=` keyword to render_rays_cross_ray / batched_inference /
    NeRF_sigma.forward overrides it per call.
    Training (grad mode) does not read this setting: its default is "auto" for the forward and the data gradient (fp32-accurate two-piece fp16
    splits on the fp16 matrix cores, f32x3 where the h2 core refuses) and f16x2 for the weight gradients (the same two-piece form, ranged per
    delta tensor; bf16x3 -- three-piece bf16 splits -- inside the same launch where an operand leaves fp16's range) -- every
    product fp32-ACCURATE (float64 distance of the fp32 MFMA), none bit-for-bit the reference's fp32 arithmetic.  Opt out with
    autograd.set_training_forward_precision("f32") / CRNERF_TRAIN_FWD=f32 and autograd.set_wgrad_precision("f32") / CRNERF_WGRAD_F32=1
    (every product on the fp32 matrix cores); autograd.set_training_precision("bf16") is the opt-in mixed-precision mode."""
    global _precision
    from .ops import _is_auto, _is_bf16, _is_h2, _is_x3
    _precision = "bf16_hc" if precision in ("bf16_hc", "bf16+h2c") else "auto" if _is_auto(precision) else "f32h2" if _is_h2(precision) else ("f32x3" if _is_x3(precision) else ("bf16" if _is_bf16(precision) else "f32"))


def get_precision():
    return _precision
__version__ = "0.1.0"
