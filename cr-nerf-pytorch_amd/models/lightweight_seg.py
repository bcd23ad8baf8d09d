"""Transient-mask network: Context_Guided_Network as CR-NeRF instantiates it (classes=1, M=2, N=2, input_channel=3;
train_mask_grid_sample.py:99-101), on the HIP operators of csrc/cgnet.hip.

Mirror of models/lightweight_seg.py:12-368: the same module tree, so the same parameter and buffer names
(`level2_0.F_glo.fc.0.weight`, `b1.bn.running_mean`, ...) and a reference checkpoint's `implicit_mask.*` entries load
unchanged; each module's forward is one fused HIP call (conv, BatchNorm+PReLU, FGlo, pooling, upsample+sigmoid), with
the backward in HIP too.  torch supplies only the parameter containers, the channel concatenations and the residual
add.  One image per call (batch 1) -- that is the only way the reference uses it.

Attribution: the architecture (module tree, layer hyper-parameters and therefore the attribute names a checkpoint's
state_dict fixes) is CGNet by Tianyi Wu et al. (wutianyi@ict.ac.cn, (c) 2018), as vendored in the reference's
models/lightweight_seg.py; only that schema is shared -- every forward here dispatches to this repo's HIP operators.
"""
import torch
from torch import nn

import os

from ..autograd import AvgPool3s2Fn, BilinearGatherFn, BNPReLUFn, CGNetFn, Conv2dFn, FGloFn

__all__ = ["Context_Guided_Network", "mask_at_pixels", "set_chain"]

_CHAIN = [os.environ.get("CRNERF_CGNET_CHAIN", "1") != "0"]


def set_chain(on):
    """Training-mode forward as one autograd node (default) or module by module (the path eval mode and other M / N always take)."""
    _CHAIN[0] = bool(on)


class _ConvParam(nn.Module):
    """Holds `weight` under the name nn.Conv2d would give it (the reference wraps nn.Conv2d as `.conv`)."""

    def __init__(self, n_in, n_out, k, stride=1, dilation=1, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in // groups, k, k))
        nn.init.kaiming_normal_(self.weight)            # lightweight_seg.py:318-321
        self.stride, self.padding, self.dilation, self.groups = stride, (k - 1) // 2 * dilation, dilation, groups

    def forward(self, x):
        return Conv2dFn.apply(x, self.weight, self.stride, self.padding, self.dilation, self.groups)


def _bn_prelu(x, bn, act):
    return BNPReLUFn.apply(x, bn.weight, bn.bias, act.weight, bn)


class ConvBNPReLU(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1):
        super().__init__()
        self.conv = _ConvParam(nIn, nOut, kSize, stride)
        self.bn = nn.BatchNorm2d(nOut, eps=1e-03)
        self.act = nn.PReLU(nOut)

    def forward(self, x):
        return _bn_prelu(self.conv(x), self.bn, self.act)


class BNPReLU(nn.Module):
    def __init__(self, nOut):
        super().__init__()
        self.bn = nn.BatchNorm2d(nOut, eps=1e-03)
        self.act = nn.PReLU(nOut)

    def forward(self, x):
        return _bn_prelu(x, self.bn, self.act)


class Conv(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1):
        super().__init__()
        self.conv = _ConvParam(nIn, nOut, kSize, stride)

    def forward(self, x):
        return self.conv(x)


class ChannelWiseConv(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1):
        super().__init__()
        self.conv = _ConvParam(nIn, nOut, kSize, stride, groups=nIn)

    def forward(self, x):
        return self.conv(x)


class ChannelWiseDilatedConv(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1, d=1):
        super().__init__()
        self.conv = _ConvParam(nIn, nOut, kSize, stride, dilation=d, groups=nIn)

    def forward(self, x):
        return self.conv(x)


class FGlo(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction), nn.ReLU(inplace=True), nn.Linear(channel // reduction, channel),
                                nn.Sigmoid())     # containers for fc.0 / fc.2; evaluated fused

    def forward(self, x):
        return FGloFn.apply(x, self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias)


class ContextGuidedBlock_Down(nn.Module):
    def __init__(self, nIn, nOut, dilation_rate=2, reduction=16):
        super().__init__()
        self.conv1x1 = ConvBNPReLU(nIn, nOut, 3, 2)
        self.F_loc = ChannelWiseConv(nOut, nOut, 3, 1)
        self.F_sur = ChannelWiseDilatedConv(nOut, nOut, 3, 1, dilation_rate)
        self.bn = nn.BatchNorm2d(2 * nOut, eps=1e-3)
        self.act = nn.PReLU(2 * nOut)
        self.reduce = Conv(2 * nOut, nOut, 1, 1)
        self.F_glo = FGlo(nOut, reduction)

    def forward(self, x):
        y = self.conv1x1(x)
        local_and_surround = torch.cat([self.F_loc(y), self.F_sur(y)], 1)
        return self.F_glo(self.reduce(_bn_prelu(local_and_surround, self.bn, self.act)))


class ContextGuidedBlock(nn.Module):
    def __init__(self, nIn, nOut, dilation_rate=2, reduction=16, add=True):
        super().__init__()
        n = int(nOut / 2)
        self.conv1x1 = ConvBNPReLU(nIn, n, 1, 1)
        self.F_loc = ChannelWiseConv(n, n, 3, 1)
        self.F_sur = ChannelWiseDilatedConv(n, n, 3, 1, dilation_rate)
        self.bn_prelu = BNPReLU(nOut)
        self.add = add
        self.F_glo = FGlo(nOut, reduction)

    def forward(self, x):
        y = self.conv1x1(x)
        y = self.F_glo(self.bn_prelu(torch.cat([self.F_loc(y), self.F_sur(y)], 1)))
        return x + y if self.add else y


class InputInjection(nn.Module):
    def __init__(self, downsamplingRatio):
        super().__init__()
        self.ratio = downsamplingRatio

    def forward(self, x):
        for _ in range(self.ratio):
            x = AvgPool3s2Fn.apply(x)
        return x


class Context_Guided_Network(nn.Module):
    """CGNet; forward(image[1,C,H,W]) -> mask[1,classes,H,W] in (0,1)  (lightweight_seg.py:274-368)."""

    def __init__(self, classes=19, M=3, N=21, input_channel=64, dropout_flag=False):
        super().__init__()
        if dropout_flag:
            raise NotImplementedError("crnerf_amd: dropout_flag is never set by CR-NeRF and is not built")
        if classes != 1:
            raise NotImplementedError("crnerf_amd: the upsample operator handles the single-channel mask CR-NeRF uses (classes=1)")
        self.level1_0 = ConvBNPReLU(input_channel, 32, 3, 2)
        self.level1_1 = ConvBNPReLU(32, 32, 3, 1)
        self.level1_2 = ConvBNPReLU(32, 32, 3, 1)
        self.sample1 = InputInjection(1)
        self.sample2 = InputInjection(2)
        self.b1 = BNPReLU(32 + input_channel)
        self.level2_0 = ContextGuidedBlock_Down(32 + input_channel, 64, dilation_rate=2, reduction=8)
        self.level2 = nn.ModuleList(ContextGuidedBlock(64, 64, dilation_rate=2, reduction=8) for _ in range(M - 1))
        self.bn_prelu_2 = BNPReLU(128 + input_channel)
        self.level3_0 = ContextGuidedBlock_Down(128 + input_channel, 128, dilation_rate=4, reduction=16)
        self.level3 = nn.ModuleList(ContextGuidedBlock(128, 128, dilation_rate=4, reduction=16) for _ in range(N - 1))
        self.bn_prelu_3 = BNPReLU(256)
        self.classifier = nn.Sequential(Conv(256, classes, 1, 1))

    # ---- the training step's fast path: the whole network as one autograd node (csrc/cgnet_chain.hip) ----
    def _chain_modules(self):
        """(parameter list, BatchNorm list) in the order csrc/cgnet_chain.hip documents -- the module tree's own order; kept per module."""
        from .. import ops as _ops
        return _ops.cached_list(self, "chain_modules", self._build_chain_modules)

    def _build_chain_modules(self):
        def cbp(m):
            return [m.conv.weight, m.bn.weight, m.bn.bias, m.act.weight], [m.bn]

        def down(m):
            p, b = cbp(m.conv1x1)
            return (p + [m.F_loc.conv.weight, m.F_sur.conv.weight, m.bn.weight, m.bn.bias, m.act.weight, m.reduce.conv.weight,
                         m.F_glo.fc[0].weight, m.F_glo.fc[0].bias, m.F_glo.fc[2].weight, m.F_glo.fc[2].bias], b + [m.bn])

        def block(m):
            p, b = cbp(m.conv1x1)
            return (p + [m.F_loc.conv.weight, m.F_sur.conv.weight, m.bn_prelu.bn.weight, m.bn_prelu.bn.bias, m.bn_prelu.act.weight,
                         m.F_glo.fc[0].weight, m.F_glo.fc[0].bias, m.F_glo.fc[2].weight, m.F_glo.fc[2].bias], b + [m.bn_prelu.bn])

        def bp(m):
            return [m.bn.weight, m.bn.bias, m.act.weight], [m.bn]

        params, bns = [], []
        for p, b in (cbp(self.level1_0), cbp(self.level1_1), cbp(self.level1_2), bp(self.b1), down(self.level2_0), block(self.level2[0]),
                     bp(self.bn_prelu_2), down(self.level3_0), block(self.level3[0]), bp(self.bn_prelu_3)):
            params += p
            bns += b
        return params + [self.classifier[0].conv.weight], bns

    def _chain_applies(self, image, modules=None):
        """One node for the whole network when this is the reference's training configuration: M = N = 2, train mode under autograd, every
        BatchNorm tracking running statistics with one momentum / eps, fp32 parameters, an image that needs no gradient."""
        if not (_CHAIN[0] and self.training and torch.is_grad_enabled() and len(self.level2) == 1 and len(self.level3) == 1):
            return False
        if image.requires_grad or image.dim() != 4 or image.shape[0] != 1:
            return False
        params, bns = modules if modules is not None else self._chain_modules()
        b0 = bns[0]
        if b0.momentum is None or any(b.running_mean is None or not b.track_running_stats or b.momentum != b0.momentum or b.eps != b0.eps
                                      or b.running_mean.dtype != torch.float32 or not b.running_mean.is_cuda for b in bns):
            return False
        return all(p.dtype == torch.float32 and p.is_cuda and p.is_contiguous() for p in params)

    def forward(self, image):
        if not image.is_cuda:
            raise RuntimeError("crnerf_amd: Context_Guided_Network runs on the HIP operators only; input is on %s" % image.device)
        modules = self._chain_modules() if _CHAIN[0] and self.training and len(self.level2) == 1 and len(self.level3) == 1 else None
        if modules is not None and self._chain_applies(image, modules):
            return CGNetFn.apply(image, modules[1], *modules[0])
        stage1 = self.level1_2(self.level1_1(self.level1_0(image)))                     # 1/2 scale, 32 channels
        half, quarter = self.sample1(image), self.sample2(image)                        # the image itself, re-injected at 1/2 and 1/4
        stage2_in = self.level2_0(self.b1(torch.cat([stage1, half], 1)))                # 1/4 scale, 64 channels
        stage2 = stage2_in
        for block in self.level2:
            stage2 = block(stage2)
        stage3_in = self.level3_0(self.bn_prelu_2(torch.cat([stage2, stage2_in, quarter], 1)))   # 1/8 scale, 128 channels
        stage3 = stage3_in
        for block in self.level3:
            stage3 = block(stage3)
        logits = self.classifier(self.bn_prelu_3(torch.cat([stage3_in, stage3], 1)))
        return BilinearGatherFn.apply(logits, tuple(image.shape[2:]), None, True)       # x8 bilinear upsample + sigmoid


def mask_at_pixels(pred_mask, hw_whole, rgb_idx):
    """interpolate(pred_mask, size=hw_whole)[rgb_idx] without forming the full-resolution mask
    (train_mask_grid_sample.py:172-175: interpolate -> rearrange('b c h w -> (b h w) c') -> [rgb_idx]).  -> [n,1]
    rgb_idx=None: every pixel, row-major (NeRFSystem.forward's val_mode, :174-175 without the gather)."""
    out = BilinearGatherFn.apply(pred_mask, (int(hw_whole[0]), int(hw_whole[1])), rgb_idx, False)
    return out.reshape(-1, 1)
