"""Test-time camera paths and the frame loop of the appearance-hallucination video (SURVEY 8f N2 / BASELINE configs[4]):
appearance_modification_video.py:104-189 (poses, intrinsics) and :224-262 (per-frame render + decode).

The two hard-coded fly-throughs are kept as DATA (start pose, per-axis key values, how the frame range is cut into
segments) and evaluated by one generic routine; frames are independent, so a multi-GPU run shards the frame list
across ranks with no collective ("video")."""
import math

import numpy as np
import torch

from . import pipeline

N_FRAMES = 30 * 8     # :122

# Each channel: list of (start, end) key values, one per segment; segments split the frame range as the reference does
# ("halves": N//2 and the rest, "quarters": three of N//4 and the rest, "whole": one segment).
PATHS = {
    "brandenburg_gate": {          # :120-147
        "pose_init": [[0.99702646, 0.00170214, -0.07704115, 0.03552477],
                      [0.01082206, -0.99294089, 0.11811554, 0.02343685],
                      [-0.07629626, -0.11859807, -0.99000676, 0.12162088]],
        "dx": ("whole", [(-0.25, 0.25)]),
        "dy": ("halves", [(0.05, -0.1), (-0.1, 0.05)]),
        "dz": ("halves", [(0.1, 0.3), (0.3, 0.1)]),
        "theta_x": ("halves", [(math.pi / 30, 0.0), (0.0, math.pi / 30)]),
        "theta_y": ("whole", [(math.pi / 10, -math.pi / 10)]),
        "theta_z": ("whole", [(0.0, 0.0)]),
    },
    "trevi_fountain": {            # :149-181
        "pose_init": [[9.99719757e-01, -4.88717623e-03, -2.31629550e-02, -2.66316808e-02],
                      [-6.52512819e-03, -9.97442504e-01, -7.11749546e-02, -6.68793042e-04],
                      [-2.27558713e-02, 7.13061496e-02, -9.97194867e-01, 7.93278041e-04]],
        "dx": ("whole", [(-0.8, 0.7)]),
        "dy": ("halves", [(-0.0, 0.05), (0.05, -0.0)]),
        "dz": ("quarters", [(0.4, 0.1), (0.1, 0.5), (0.5, 0.1), (0.1, 0.4)]),
        "theta_x": ("halves", [(-0.0, 0.0), (0.0, -0.0)]),
        "theta_y": ("whole", [(math.pi / 6, -math.pi / 6)]),
        "theta_z": ("whole", [(0.0, 0.0)]),
    },
}
_CUTS = {"whole": lambda n: [n], "halves": lambda n: [n // 2, n - n // 2], "quarters": lambda n: [n // 4, n // 4, n // 4, n - 3 * (n // 4)]}


def _channel(spec, n):
    kind, keys = spec
    return np.concatenate([np.linspace(a, b, m) for (a, b), m in zip(keys, _CUTS[kind](n))])


def euler_to_matrix(tx, ty, tz):
    """R = Rz(tz) @ Ry(ty) @ Rx(tx)  (eulerAnglesToRotationMatrix, :104-118)."""
    cx, sx, cy, sy, cz, sz = math.cos(tx), math.sin(tx), math.cos(ty), math.sin(ty), math.cos(tz), math.sin(tz)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ (ry @ rx)


def define_poses(scene, n_frames=N_FRAMES):
    """[n_frames, 3, 4] camera-to-world matrices of the named fly-through ('brandenburg_gate' / 'trevi_fountain')."""
    path = PATHS[scene]
    ch = {k: _channel(path[k], n_frames) for k in ("dx", "dy", "dz", "theta_x", "theta_y", "theta_z")}
    poses = np.tile(np.asarray(path["pose_init"], dtype=np.float64), (n_frames, 1, 1))
    for i in range(n_frames):
        poses[i, :, 3] += (ch["dx"][i], ch["dy"][i], ch["dz"][i])
        poses[i, :, :3] = euler_to_matrix(ch["theta_x"][i], ch["theta_y"][i], ch["theta_z"][i]) @ poses[i, :, :3]
    return poses


def define_camera(img_wh):
    """Pinhole intrinsics with a 60 degree horizontal field of view (:183-189)."""
    w, h = img_wh
    focal = w / 2 / np.tan(np.pi / 6)
    return np.array([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1]])


@torch.no_grad()
def render_video(models, embeddings, enc_a, style_img, hparams_, scene="brandenburg_gate", n_frames=N_FRAMES, rank=0, world_size=1,
                 chunk=32768, precision=None, near=0.0, far=5.0):
    """Frames rank, rank + world_size, ... of the fly-through as uint8 [H,W,3] arrays (the reference writes PNGs + a GIF,
    :255-262).  style_img: [1,3,h,w] in [0,1] (the 1/8-scale example image, :236-246)."""
    w, h = hparams_.img_wh
    K, poses = define_camera(hparams_.img_wh), define_poses(scene, n_frames)
    frames = {}
    a_emb = enc_a(style_img)            # once per style image, as the reference (appearance_modification_video.py:239)
    for i in range(rank, n_frames, world_size):
        img = pipeline.render_frame(models, embeddings, enc_a, style_img, h, w, K, poses[i].astype(np.float32), hparams_, near=near, far=far,
                                    chunk=chunk, precision=precision, a_emb=a_emb)
        frames[i] = (img.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
    return frames
