"""get_ray_directions / get_rays with the reference's signatures (datasets/ray_utils.py:5-52) computed on
the GPU, plus generate_rays(): the whole per-frame ray build of datasets/PhototourismDataset.py:12-25 in
one launch, written straight into the renderer's rays[H*W,8] layout (no host build + H2D per frame)."""
import ctypes

import numpy as np
import torch

from .. import _lib


def _host_floats(a, n):
    a = np.ascontiguousarray(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float32).reshape(-1))
    if a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def get_ray_directions(H, W, K, device="cuda"):
    """(H, W, 3) ray directions in camera coordinates; K: (3,3) intrinsics (no +0.5 pixel centring)."""
    lib = _lib.load()
    K = np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float64)
    out = torch.empty(int(H), int(W), 3, dtype=torch.float32, device=device)
    _lib.check(lib.crnerf_ray_directions_f32(int(H), int(W), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                                             _lib.dev_ptr(out), _lib.stream_ptr()), "crnerf_ray_directions_f32")
    return out


def get_rays(directions, c2w):
    """directions (H,W,3) on the GPU, c2w (3,4) -> rays_o (H*W,3), rays_d (H*W,3), normalised, world frame."""
    lib = _lib.load()
    d = directions.reshape(-1, 3)
    d = d if (d.dtype == torch.float32 and d.is_contiguous()) else d.to(torch.float32).contiguous()
    keep, ptr = _host_floats(c2w, 12)
    o, r = torch.empty_like(d), torch.empty_like(d)
    _lib.check(lib.crnerf_rays_from_directions_f32(_lib.dev_ptr(d), ptr, d.shape[0], _lib.dev_ptr(o), _lib.dev_ptr(r), _lib.stream_ptr()),
               "crnerf_rays_from_directions_f32")
    return o, r


def generate_rays(H, W, K, c2w, near=0.0, far=5.0, device="cuda"):
    """rays[H*W,8] = cat[rays_o, rays_d, near, far] for one camera (video path: near=0, far=5)."""
    lib = _lib.load()
    K = np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float64)
    k1, kp = _host_floats([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], 4)
    k2, cp = _host_floats(c2w, 12)
    rays = torch.empty(int(H) * int(W), 8, dtype=torch.float32, device=device)
    _lib.check(lib.crnerf_generate_rays_f32(kp, cp, int(H), int(W), float(near), float(far), _lib.dev_ptr(rays), _lib.stream_ptr()),
               "crnerf_generate_rays_f32")
    return rays
