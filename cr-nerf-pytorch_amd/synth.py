"""Seeded synthetic inputs (no dataset or checkpoint ships with the reference): network weights and
rays.  Pure numpy so that the golden-vector generator, the tests and bench.py all regenerate the same
bits from a seed instead of committing multi-MB blobs.  """
import math

import numpy as np

from .ops import MLP_TENSOR_NAMES, MLP_TENSOR_SHAPES

DECODER_SHAPES = {}
for _net in ("snet", "cnet"):
    DECODER_SHAPES["multi_net.%s.convs.0.weight" % _net] = (128, 64, 1, 1)
    DECODER_SHAPES["multi_net.%s.convs.0.bias" % _net] = (128,)
    DECODER_SHAPES["multi_net.%s.convs.2.weight" % _net] = (64, 128, 1, 1)
    DECODER_SHAPES["multi_net.%s.convs.2.bias" % _net] = (64,)
    DECODER_SHAPES["multi_net.%s.convs.4.weight" % _net] = (32, 64, 1, 1)
    DECODER_SHAPES["multi_net.%s.convs.4.bias" % _net] = (32,)
    DECODER_SHAPES["multi_net.%s.fc.weight" % _net] = (1024, 1024)
    DECODER_SHAPES["multi_net.%s.fc.bias" % _net] = (1024,)
DECODER_SHAPES["multi_net.compress.weight"] = (32, 64, 1, 1)
DECODER_SHAPES["multi_net.compress.bias"] = (32,)
DECODER_SHAPES["multi_net.unzip.weight"] = (64, 32, 1, 1)
DECODER_SHAPES["multi_net.unzip.bias"] = (64,)
DECODER_SHAPES["decoder.rgb_upsample.1.f"] = (3,)
DECODER_SHAPES["decoder.feat_2_rgb_list.0.weight"] = (3, 64, 1, 1)
DECODER_SHAPES["decoder.feat_2_rgb_list.0.bias"] = (3,)


def _uniform_linear(rng, shape, fan_in, gain):
    bound = gain / math.sqrt(fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def mlp_state(seed, gain=1.0, sigma_bias=0.0, band_limit=None):
    """24 tensors of one NeRF_sigma, nn.Linear-style U(-g/sqrt(fan_in), g/sqrt(fan_in)).
    gain=3 gives the "peaky" variant (sigma spans 0..50, features saturate) of SURVEY 8c.
    band_limit=k0: the "smooth-density" variant -- the columns of the two layers that read the xyz embedding
    (xyz_encoding_1, and the first 93 columns of the skip layer xyz_encoding_5; nerf.py:137-141) that belong to frequency
    2^k are scaled by 2^-(k-k0) for k > k0, so the network is a smooth function of position (|d out/d x| = O(2^k0))
    instead of amplifying a 1-ulp depth change by 2^14.  On such a net the reference itself is well-conditioned
    (tests/golden/make_golden.py::smooth_goldens records its own 1-ulp sensitivity), so end-to-end parity can be held
    to SURVEY 8d's stated tolerances."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in zip(MLP_TENSOR_NAMES, MLP_TENSOR_SHAPES):
        fan_in = shape[1] if len(shape) == 2 else MLP_TENSOR_SHAPES[MLP_TENSOR_NAMES.index(name.replace("bias", "weight"))][1]
        out[name] = _uniform_linear(rng, shape, fan_in, gain)
    out["static_sigma.0.bias"] = out["static_sigma.0.bias"] + np.float32(sigma_bias)
    if band_limit is not None:
        for name in ("xyz_encoding_1.0.weight", "xyz_encoding_5.0.weight"):
            for k in range(15):                               # embedding columns [x | sin 2^0 x, cos 2^0 x | sin 2^1 x, ...], 3 wide each
                out[name][:, 3 + 6 * k:9 + 6 * k] *= np.float32(2.0 ** (-max(k - band_limit, 0)))
    return out


def decoder_state(seed, gain=1.0, contrast=1.0):
    """22 tensors of style_net (+ the unused blur kernel), nn.Conv2d / nn.Linear-style uniform init.
    contrast != 1: the "high-gain" decoder of the parity instruments -- with default-scale weights the pixel-specific part
    of the decode (unzip(T @ compress(x - mean)), linearStyleTransfer.py:86-88) is ~1e-2 of the style-mean part, so the image
    spans rgb in [0.55, 0.66] and a feature error is damped ~500x.  `contrast` multiplies multi_net.unzip.weight and centres
    the rgb bias for a 0.5-mean style; contrast=4000 makes the image span ~[0.14, 0.78] and a 1e-3 feature error move
    pixels by ~7e-3, so PSNR / max|d rgb| respond to feature-level errors."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in DECODER_SHAPES.items():
        if name.endswith(".f"):
            out[name] = np.array([1, 2, 1], dtype=np.float32)
            continue
        wshape = DECODER_SHAPES[name.replace("bias", "weight")]
        out[name] = _uniform_linear(rng, shape, wshape[1], gain)
    if contrast != 1.0:
        out["multi_net.unzip.weight"] = out["multi_net.unzip.weight"] * np.float32(contrast)
        w = out["decoder.feat_2_rgb_list.0.weight"].reshape(3, 64)
        out["decoder.feat_2_rgb_list.0.bias"] = (-0.5 * w.sum(1)).astype(np.float32)
    return out


ENCODER_SHAPES = {}
for _i, (_ci, _co, _k) in enumerate(((3, 3, 1), (3, 64, 3), (64, 64, 3), (64, 128, 3), (128, 128, 3), (128, 128, 3), (128, 64, 1)), start=1):
    ENCODER_SHAPES["conv%d.weight" % _i] = (_co, _ci, _k, _k)
    ENCODER_SHAPES["conv%d.bias" % _i] = (_co,)


def encoder_state(seed, gain=1.0):
    """14 tensors of encoder_sameoutputsize (models/linearStyleTransfer.py:208-276), nn.Conv2d-style uniform init."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in ENCODER_SHAPES.items():
        w = ENCODER_SHAPES[name.replace("bias", "weight")]
        out[name] = _uniform_linear(rng, shape, w[1] * w[2] * w[3], gain)
    return out


def rays(n_rays, seed=0, H=None, W=None, near=None, far=None):
    """rays[R,8]: 60-degree-fov pinhole at a jittered origin, unit directions (datasets/ray_utils.py:45),
    one (near, far) pair per image drawn from [0.3,1] x [3,5] unless given."""
    rng = np.random.default_rng(seed)
    if H is None or W is None:
        W = int(math.ceil(math.sqrt(n_rays)))
        H = int(math.ceil(n_rays / W))
    focal = W / 2 / math.tan(math.pi / 6)
    j, i = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    d = np.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -np.ones_like(i)], -1).reshape(-1, 3)[:n_rays]
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(rng.normal(0, 0.1, size=(1, 3)), d.shape)
    nr = rng.uniform(0.3, 1.0) if near is None else near
    fr = rng.uniform(3.0, 5.0) if far is None else far
    out = np.concatenate([o, d, np.full((n_rays, 1), nr), np.full((n_rays, 1), fr)], -1)
    return np.ascontiguousarray(out, dtype=np.float32)
