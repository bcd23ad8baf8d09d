"""losses.py of the reference (CRNeRFLoss :42-94, ColorLoss :6-17, the two annealing schedules :19-39, loss_dict :96-97)
on the fused HIP loss kernel: one launch produces the seven loss terms, one more their gradients.

    from crnerf_amd.losses import loss_dict
    criterion = loss_dict['crnerf'](hparams, coef=1)                 # train_mask_grid_sample.py:74
    loss_d, annealing = criterion(results, rgbs, hparams, global_step)   # :284
    loss = sum(l for l in loss_d.values())
"""
import math

import torch
from torch import nn

from . import ops


class CosineAnnealingWeight():
    def __init__(self, max, min, Tmax):
        self.max, self.min, self.Tmax = max, min, Tmax

    def getWeight(self, Tcur):
        return self.min + (self.max - self.min) * (1 + math.cos(math.pi * Tcur / self.Tmax)) / 2


class ExponentialAnnealingWeight():
    def __init__(self, max, min, k):
        self.max, self.min, self.k = max, min, k

    def getWeight(self, Tcur):
        return max(self.min, self.max * math.exp(-Tcur * self.k))


_GRAD_OF = {"rgb_coarse": "rgb_coarse", "rgb_fine": "rgb_fine", "out_mask": "mask", "a_embedded": "a_embedded",
            "a_embedded_random_rec": "a_embedded_random_rec", "content_wo_a_embed": "content_wo", "content_with_a_embed": "content_with"}


# what each of the kernel's seven terms (ops.LOSS_KEYS order) reads beside rgb_coarse / targets: a term whose inputs are absent is an exact zero
_TERM_NEEDS = {"kl_a": ("a_embedded",), "rec_a_random": ("a_embedded", "a_embedded_random_rec"), "c_l": (), "content_constraint": ("content_wo_a_embed", "content_with_a_embed"),
               "r_ms": ("out_mask", "rgb_fine"), "r_md": ("out_mask", "rgb_fine"), "f_l": ("rgb_fine",)}


class _LossFn(torch.autograd.Function):
    """losses[7] = crnerf_loss_f32(...); backward = crnerf_loss_backward_f32 with autograd's upstream[7]."""

    @staticmethod
    def forward(ctx, cfg, names, *tensors):
        inputs = dict(zip(names, tensors))
        args, keep = ops.loss_args(inputs["rgb_coarse"], inputs["targets"], rgb_fine=inputs.get("rgb_fine"), mask=inputs.get("out_mask"),
                                   a_embedded=inputs.get("a_embedded"), a_embedded_random=inputs.get("a_embedded_random"),
                                   a_embedded_random_rec=inputs.get("a_embedded_random_rec"), content_wo=inputs.get("content_wo_a_embed"),
                                   content_with=inputs.get("content_with_a_embed"), **cfg)
        ctx.args, ctx.keep, ctx.names, ctx.shapes = args, keep, names, [t.shape for t in tensors]
        return ops.loss_forward(args)

    @staticmethod
    def backward(ctx, upstream):
        want = {_GRAD_OF[n]: (s if n not in ("rgb_coarse", "rgb_fine") else (s[0], 3))
                for n, s, need in zip(ctx.names, ctx.shapes, ctx.needs_input_grad[2:]) if need and n in _GRAD_OF}
        order = getattr(ctx.args, "_memory_order", {})      # inputs the kernel read in memory order: their gradients get the same strides
        got = ops.loss_backward(ctx.args, upstream.contiguous(), want, {_GRAD_OF[n]: st for n, st in order.items() if n in _GRAD_OF}) if want else {}
        grads = []
        for n, s, need in zip(ctx.names, ctx.shapes, ctx.needs_input_grad[2:]):
            grads.append(got[_GRAD_OF[n]].view(s) if (need and n in _GRAD_OF) else None)
        return (None, None) + tuple(grads)


class LossTerms(dict):
    """The reference's loss dict (same keys, same insertion order, each value a 0-dim tensor) that also remembers the kernel's 7-vector it was cut
    from: total() is the reference's `sum(l for l in loss_d.values())` as ONE reduction (absent terms are exact zeros in the vector) instead of
    seven selects, seven adds and their twenty-odd backward launches."""
    vector = None

    def total(self):
        return self.vector.sum() if self.vector is not None else sum(l for l in self.values())


class CRNeRFLoss(nn.Module):
    """Same constructor, call signature and returned (dict, annealing weight) as the reference (losses.py:42-78); the dict
    holds exactly the keys the reference would produce for the given inputs, in its insertion order."""

    def __init__(self, hparams, coef=1, lambda_u=0.01):
        super().__init__()
        self.coef = coef
        self.lambda_u = lambda_u
        self.Annealing = ExponentialAnnealingWeight(max=hparams.maskrs_max, min=hparams.maskrs_min, k=hparams.maskrs_k)

    def forward(self, inputs, targets, hparams, global_step):
        ann = self.Annealing.getWeight(global_step)
        has = lambda k: k in inputs  # noqa: E731
        use = {"rgb_coarse": inputs["rgb_coarse"], "targets": targets}
        keys = []
        if has("a_embedded"):
            use["a_embedded"] = inputs["a_embedded"]
            keys.append("kl_a")
            if has("a_embedded_random_rec"):
                use["a_embedded_random"] = inputs["a_embedded_random"]
                use["a_embedded_random_rec"] = inputs["a_embedded_random_rec"]
                keys.append("rec_a_random")
        if has("out_mask"):
            use["out_mask"] = inputs["out_mask"]
        keys.append("c_l")
        if has("content_wo_a_embed") and has("content_with_a_embed"):
            use["content_wo_a_embed"], use["content_with_a_embed"] = inputs["content_wo_a_embed"], inputs["content_with_a_embed"]
            keys.append("content_constraint")
        if has("rgb_fine"):
            use["rgb_fine"] = inputs["rgb_fine"]
            keys += (["r_ms", "r_md"] if has("out_mask") else []) + ["f_l"]
        cfg = dict(mse_on_appearance=getattr(hparams, "mse_on_appearance", False), coef=self.coef, weight_kl=getattr(hparams, "weightKL", 0.0),
                   weight_rec_a=getattr(hparams, "weightRecA", 0.0), weight_content=getattr(hparams, "weightcontent", 0.0),
                   mask_size_weight=ann, mask_digit_weight=getattr(hparams, "maskrd", 0.0))
        names = tuple(use.keys())
        losses = _LossFn.apply(cfg, names, *[use[n] for n in names])
        out = LossTerms((k, losses[ops.LOSS_KEYS.index(k)]) for k in keys)
        if set(keys) >= {k for k in ops.LOSS_KEYS if all(n in use for n in _TERM_NEEDS[k])}:    # every term the kernel can have filled is a key
            out.vector = losses
        return out, ann


class ColorLoss(nn.Module):
    """losses.py:6-17: coef * (MSE(rgb_coarse) + MSE(rgb_fine)) = 2 coef (c_l + f_l) of the unmasked CRNeRF terms."""

    def __init__(self, coef=1):
        super().__init__()
        self.coef = coef

    def forward(self, inputs, targets):
        use = {"rgb_coarse": inputs["rgb_coarse"], "targets": targets}
        if "rgb_fine" in inputs:
            use["rgb_fine"] = inputs["rgb_fine"]
        names = tuple(use.keys())
        losses = _LossFn.apply(dict(coef=2.0 * self.coef), names, *[use[n] for n in names])
        return losses[2] + losses[6] if "rgb_fine" in inputs else losses[2]


loss_dict = {'color': ColorLoss, 'crnerf': CRNeRFLoss}
