"""The procedural training scene behind tests/golden/g15_trained.npz: a few soft coloured balls, a ring of pinhole cameras, one
appearance (per-channel gain / offset) per image and an opaque transient rectangle in every third image.  Pure numpy + torch
tensors; shared by the fixture generator (tests/golden/make_golden_trained.py, which trains the REFERENCE on it) and
tests/test_gpu_train_converges.py (which trains the HIP path on the identical images)."""
import argparse
import math

import numpy as np
import torch

SIDE = 32                       # images are SIDE x SIDE; one training batch = one image's 1,024 rays as a 32 x 32 grid
N_IMAGES = 12


def hparams():
    """opt.py defaults with command/train.sh's overrides; sample counts cut to 32+32 for CPU training time."""
    return argparse.Namespace(
        N_emb_xyz=15, N_emb_dir=4, N_samples=32, N_importance=32, use_disp=False, pertubeCord=False, perturb=1.0, noise_std=1.0,
        N_vocab=N_IMAGES, encode_a=True, encode_c=True, encode_random=True, use_mask=True, N_a=48, nerf_out_dim=64, img_wh=[SIDE, SIDE],
        decoder="linearStyle", decoder_num_res_blocks=1, maskrs_max=5e-2, maskrs_min=6e-3, maskrs_k=1e-3, maskrd=0.0, weightKL=1e-5,
        weightRecA=1e-3, weightMS=1e-6, weightcontent=1e-4, mse_on_appearance=False, batch_size=SIDE * SIDE, chunk=1310720,
        optimizer="adam", lr=5e-4, lr_scheduler="cosine", weight_decay=0, num_epochs=20)


# ---------------------------------------------------------------- procedural scene ----------------------------------------------------------
BLOBS = np.array([  # centre xyz, radius, density, rgb
    [0.00, 0.00, 0.00, 0.45, 18.0, 0.85, 0.25, 0.20],
    [0.55, 0.15, -0.20, 0.28, 25.0, 0.20, 0.65, 0.90],
    [-0.45, -0.25, 0.30, 0.33, 14.0, 0.30, 0.80, 0.35],
    [0.10, 0.50, 0.40, 0.22, 30.0, 0.95, 0.85, 0.30]], dtype=np.float64)


def field(p):
    """density [..], colour [..,3] of a sum of soft balls with a striped albedo (so the scene has mid-frequency detail)."""
    sig = np.zeros(p.shape[:-1])
    col = np.zeros(p.shape)
    for b in BLOBS:
        r2 = ((p - b[0:3]) ** 2).sum(-1)
        s = b[4] * np.exp(-0.5 * r2 / (0.5 * b[3]) ** 2)
        stripe = 0.75 + 0.25 * np.sin(9.0 * (p[..., 0] + 0.7 * p[..., 1] - 0.4 * p[..., 2]))
        sig += s
        col += s[..., None] * b[5:8] * stripe[..., None]
    return sig, col / np.maximum(sig, 1e-9)[..., None]


def camera_rays(theta, phi, radius=2.4, near=1.2, far=3.6):
    """rays[SIDE*SIDE, 8] of a pinhole looking at the origin (datasets/ray_utils.py conventions: -z forward, unit directions)."""
    eye = radius * np.array([math.cos(phi) * math.sin(theta), math.sin(phi), math.cos(phi) * math.cos(theta)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.stack([right, up, -fwd], 1)
    focal = SIDE / 2 / math.tan(math.radians(22.0))
    j, i = np.meshgrid(np.arange(SIDE, dtype=np.float64), np.arange(SIDE, dtype=np.float64), indexing="ij")
    d = np.stack([(i - SIDE / 2) / focal, -(j - SIDE / 2) / focal, -np.ones_like(i)], -1).reshape(-1, 3) @ c2w.T
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(eye, d.shape)
    return np.concatenate([o, d, np.full((len(d), 1), near), np.full((len(d), 1), far)], -1).astype(np.float32)


def ground_truth(rays, n=384):
    """Quadrature of the emission-absorption integral on a grey background."""
    o, d, near, far = rays[:, 0:3].astype(np.float64), rays[:, 3:6].astype(np.float64), rays[:, 6:7], rays[:, 7:8]
    t = near + (far - near) * (np.arange(n) + 0.5) / n
    sig, col = field(o[:, None, :] + d[:, None, :] * t[..., None])
    alpha = 1 - np.exp(-sig * (far - near) / n)
    trans = np.cumprod(np.concatenate([np.ones_like(alpha[:, :1]), 1 - alpha[:, :-1]], 1), 1)
    w = alpha * trans
    return (w[..., None] * col).sum(1) + (1 - w.sum(1))[:, None] * 0.45


def make_dataset(rng):
    """N_IMAGES views; image k has its own appearance (per-channel gain / offset = the 'photo collection' variation the cross-ray
    transfer models) and, every third image, an opaque transient rectangle (what the mask network is for)."""
    data = []
    for k in range(N_IMAGES + 1):                                   # the last one is the held-out test view
        held_out = k == N_IMAGES
        theta = 2 * math.pi * (k + (0.37 if held_out else 0.0)) / N_IMAGES
        phi = 0.25 * math.sin(1.7 * k) + (0.1 if held_out else 0.0)
        rays = camera_rays(theta, phi)
        img = ground_truth(rays)
        gain = rng.uniform(0.6, 1.25, size=3)
        offs = rng.uniform(-0.08, 0.12, size=3)
        img = np.clip(img * gain + offs, 0, 1)
        if k % 3 == 1 and not held_out:
            y0, x0 = rng.integers(2, SIDE - 12, size=2)
            img = img.reshape(SIDE, SIDE, 3).copy()
            img[y0:y0 + 9, x0:x0 + 7] = rng.uniform(0, 1, size=3)
            img = img.reshape(-1, 3)
        data.append(dict(rays=torch.from_numpy(rays), rgbs=torch.from_numpy(img.astype(np.float32)), ts=torch.full((SIDE * SIDE,), k, dtype=torch.long)))
    return data



def whole_image(rgbs):
    return (rgbs.t().reshape(1, 3, SIDE, SIDE) * 2 - 1).contiguous()        # the dataset's normalize(mean .5, std .5)
