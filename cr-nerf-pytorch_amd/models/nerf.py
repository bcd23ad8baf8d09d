"""PosEmbedding and NeRF_sigma with the reference's constructor signatures and state_dict keys
(reference: models/nerf.py:4-30 and :115-182), backed by the HIP kernels.

Only the classes the reference instantiates are mirrored (NeRF and NeRF_sigma_tanh are never
constructed -- SURVEY 2, row 2).
"""
import torch
from torch import nn

from .. import ops

_XYZ_FREQS, _DIR_FREQS, _W, _OUT = 15, 4, 256, 64


class PosEmbedding(nn.Module):
    """(x, sin(2^k x), cos(2^k x), ...) -- reference models/nerf.py:4-30."""

    def __init__(self, max_logscale, N_freqs, logscale=True):
        super().__init__()
        if not logscale:
            raise NotImplementedError("crnerf_amd: only logscale=True frequencies are implemented in HIP "
                                      "(the reference never constructs logscale=False)")
        if max_logscale != N_freqs - 1:
            raise NotImplementedError("crnerf_amd: frequencies must be 2^0..2^(N_freqs-1), i.e. max_logscale == N_freqs-1 "
                                      "(how every reference call site builds it: eval.py:91-92)")
        self.N_freqs = int(N_freqs)
        self.max_logscale = max_logscale
        self.freqs = 2 ** torch.linspace(0, max_logscale, N_freqs)

    def forward(self, x):
        lead = x.shape[:-1]
        return ops.posenc(x.reshape(-1, 3), self.N_freqs).reshape(*lead, 6 * self.N_freqs + 3)


class NeRF_sigma(nn.Module):
    """8x256 density/feature MLP -- reference models/nerf.py:115-182 (same ctor, attributes, keys)."""

    def __init__(self, typ, args, D=8, W=256, skips=[4], in_channels_xyz=63, in_channels_dir=27,
                 encode_appearance=False, in_channels_a=48, encode_random=False):
        super().__init__()
        self.typ = typ
        self.D, self.W, self.skips = D, W, skips
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.encode_appearance = False if typ == 'coarse' else encode_appearance
        self.in_channels_a = in_channels_a if encode_appearance else 0
        self.encode_random = False if typ == 'coarse' else encode_random
        out_dim = args.nerf_out_dim
        if (D, W, list(skips), in_channels_xyz, in_channels_dir, out_dim) != (8, _W, [4], 6 * _XYZ_FREQS + 3, 6 * _DIR_FREQS + 3, _OUT):
            raise NotImplementedError(
                "crnerf_amd: the HIP kernels are specialised for the shipped network (D=8, W=256, skips=[4], "
                "in_channels_xyz=93, in_channels_dir=27, nerf_out_dim=64); got D=%r W=%r skips=%r xyz=%r dir=%r out=%r"
                % (D, W, skips, in_channels_xyz, in_channels_dir, out_dim))
        for i in range(D):
            fan_in = in_channels_xyz if i == 0 else (W + in_channels_xyz if i in skips else W)
            setattr(self, "xyz_encoding_%d" % (i + 1), nn.Sequential(nn.Linear(fan_in, W), nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.static_sigma = nn.Sequential(nn.Linear(W, 1), nn.Softplus())
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.static_rgb = nn.Sequential(nn.Linear(W // 2, out_dim), nn.Sigmoid())
        self._packed = None
        self._packed_key = None

    # ---- packed weights (kernel layout), rebuilt whenever a parameter changes
    def invalidate_packed(self):
        """Drop the cached kernel-layout copies of the weights.  Called automatically by train(), load_state_dict() and every
        grad-mode forward (a training step is about to change the parameters); call it by hand after updating parameters
        through `p.data` outside of those (p.data.* does not bump p._version, which is what the cache key watches --
        torch_optimizer's radam/ranger update that way)."""
        self._packed = None
        self._packed_key = None

    def train(self, mode=True):
        self.invalidate_packed()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def packed_weights(self, precision="f32"):
        """Packed buffer for the crnerf_*_f32 (default), crnerf_*_bf16, crnerf_*_f32x3 or crnerf_*_f32h2 entry points ("auto": an ops.AutoPack holding the
        h2 and the x3 pack); re-packed when a parameter changes."""
        bf16 = "auto" if ops._is_auto(precision) else "h2" if ops._is_h2(precision) else ("x3" if ops._is_x3(precision) else ops._is_bf16(precision))
        params = dict(zip(ops.MLP_TENSOR_NAMES, ops.mlp_params(self)))
        key = tuple((p.data_ptr(), p._version, p.device) for p in params.values())
        if self._packed is None or key != self._packed_key:
            self._packed = {}
            self._packed_key = key
        if bf16 not in self._packed:
            self._packed[bf16] = (ops.pack_mlp_weights_auto(params) if bf16 == "auto" else ops.pack_mlp_weights_h2(params) if bf16 == "h2" else ops.pack_mlp_weights_x3(params) if bf16 == "x3"
                                  else ops.pack_mlp_weights(params, out=None, precision="bf16" if bf16 else "f32"))
        return self._packed[bf16]

    def forward(self, x, sigma_only=False, output_random=True, precision=None):
        """precision: None -> crnerf_amd.get_precision(); an extension of the reference signature (nerf.py:157)."""
        if torch.is_grad_enabled() and (x.requires_grad or ops.any_requires_grad(self)):
            if sigma_only:
                raise NotImplementedError("crnerf_amd: sigma_only has no backward twin (no reference caller uses it)")
            from ..autograd import mlp_forward_with_grad   # training: HIP forward-with-save + HIP backward
            return mlp_forward_with_grad(self, x.to(torch.float32).contiguous())
        if precision is None:
            from .. import get_precision
            precision = get_precision()
        if ops._is_auto(precision):
            return ops.mlp_forward_auto(self.packed_weights(precision), x, sigma_only=sigma_only)
        if ops._is_h2(precision):
            return ops.mlp_forward_h2(self.packed_weights(precision), x, sigma_only=sigma_only)
        if ops._is_x3(precision):
            return ops.mlp_forward_x3(self.packed_weights(precision), x, sigma_only=sigma_only)
        return ops.mlp_forward(self.packed_weights(precision), x, sigma_only=sigma_only, precision=precision)
