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

__all__ = ["_lib"]
__version__ = "0.1.0"
