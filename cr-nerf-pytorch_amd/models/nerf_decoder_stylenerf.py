"""NeuralRenderer at the only configuration the reference reaches (img_size == featmap_size, so
n_blocks == 0 and the module is sigmoid(Conv1x1_{64->3}) -- reference
models/nerf_decoder_stylenerf.py:227-291).  It owns the parameters (same state_dict keys, including
the unused Blur buffer 'rgb_upsample.1.f'); the arithmetic is folded into the cross-ray apply kernel."""
from math import log2

import torch
from torch import nn

from .. import ops


class _BlurStub(nn.Module):
    """Carries the reference's `f` buffer (nerf_decoder_stylenerf.py:105-110) so checkpoints load strictly."""

    def __init__(self):
        super().__init__()
        self.register_buffer('f', torch.Tensor([1, 2, 1]))


class NeuralRenderer(nn.Module):
    def __init__(self, bg_type="white", feat_nc=128, out_dim=3, final_actvn=True, min_feat=32, featmap_size=(32, 32),
                 img_size=(256, 256), **kwargs):
        super().__init__()
        self.bg_type, self.featmap_size, self.final_actvn = bg_type, featmap_size, final_actvn
        self.n_feat, self.out_dim, self.min_feat = feat_nc, out_dim, min_feat
        self.n_blocks = int(log2(img_size[0] / featmap_size[0]))
        if self.n_blocks != 0 or feat_nc != 64 or out_dim != 3 or not final_actvn:
            raise NotImplementedError("crnerf_amd: NeuralRenderer is implemented for the shipped configuration only "
                                      "(featmap_size == img_size, feat_nc=64, out_dim=3, final sigmoid)")
        self.feat_upsample_list = nn.ModuleList()
        self.rgb_upsample = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False), _BlurStub())
        self.feat_2_rgb_list = nn.ModuleList([nn.Conv2d(feat_nc, out_dim, 1, 1, padding=0)])
        self.feat_layers = nn.ModuleList()

    def rgb_tensors(self):
        conv = self.feat_2_rgb_list[0]
        return conv.weight.reshape(3, 64), conv.bias

    def forward(self, x):
        """x: [1,64,H,W] -> [1,3,H,W]."""
        from .linearStyleTransfer import _pixel_major
        xp, (H, W) = _pixel_major(x)
        w, b = self.rgb_tensors()
        zeros = [torch.zeros(1, device=x.device)] * 4
        affine = ops.crossray_fold(None, None, None, None, zeros + [w, b])
        return ops.crossray_apply(xp, affine).view(1, 3, H, W)
