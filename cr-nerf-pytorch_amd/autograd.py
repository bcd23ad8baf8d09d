"""torch.autograd glue for training: the reference gets its gradients from PyTorch autograd over eager
ops (train_mask_grid_sample.py:268-337 -> loss.backward()); here every differentiable piece of the
path is one autograd.Function whose backward calls the HIP backward twins.

    MlpFn        NeRF_sigma.forward           fwd: crnerf_mlp_forward_train_f32   bwd: crnerf_mlp_backward_f32
    CompositeFn  inference() compositing      fwd: crnerf_composite_f32           bwd: crnerf_composite_backward_f32
    DecoderFn    style_net.forward            fwd: crnerf_crossray_decode_f32     bwd: crnerf_crossray_decode_backward_f32
    EncoderFn    encoder_sameoutputsize.forward   fwd: crnerf_encoder_forward_train_f32   bwd: crnerf_encoder_backward_f32

Inputs that the reference does not train through this path (ray geometry, embeddings, hierarchical
depths -- the latter are .detach()ed at models/rendering.py:184) get no gradient.
"""
import torch

from . import ops


class MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        if module is not None and hasattr(module, "invalidate_packed"):
            module.invalidate_packed()       # an optimiser step follows; optimisers that write p.data do not bump p._version
        state = dict(zip(ops.MLP_TENSOR_NAMES, params))
        packed = ops.pack_mlp_weights(state)
        out, acts = ops.mlp_forward_train(packed, x)
        ctx.save_for_backward(x, out, acts, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out, acts, *params = ctx.saved_tensors
        packed_t = ops.pack_mlp_weights_t(dict(zip(ops.MLP_TENSOR_NAMES, params)))
        grads = ops.mlp_backward(packed_t, x, out, d_out.contiguous(), acts)
        return (None, None) + tuple(grads)


def mlp_forward_with_grad(module, x):
    params = [dict(module.named_parameters())[n] for n in ops.MLP_TENSOR_NAMES]
    if x.requires_grad:
        raise NotImplementedError("crnerf_amd: gradients w.r.t. the embedded input of NeRF_sigma are not implemented "
                                  "(the rendering path never needs them: embeddings are functions of fixed ray geometry)")
    return MlpFn.apply(x, module, *params)


class CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, noise, noise_std):
        w, f, d = ops.composite(raw, z, noise, noise_std)
        ctx.save_for_backward(raw, z, noise if noise is not None else torch.empty(0, device=raw.device))
        ctx.noise_std = noise_std
        ctx.has_noise = noise is not None
        return w, f, d

    @staticmethod
    def backward(ctx, d_w, d_f, d_d):
        raw, z, noise = ctx.saved_tensors
        if d_f is None:
            d_f = torch.zeros(raw.shape[0], 64, device=raw.device)
        d_raw = ops.composite_backward(raw, z, d_f.contiguous(), None if d_d is None else d_d.contiguous(),
                                       None if d_w is None else d_w.contiguous(), noise=noise if ctx.has_noise else None,
                                       noise_std=ctx.noise_std)
        return d_raw, None, None, None


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, sp, *w):
        out = ops.crossray_decode(xp, sp, w)
        ctx.save_for_backward(xp, sp, *w)
        return out

    @staticmethod
    def backward(ctx, d_rgb):
        xp, sp, *w = ctx.saved_tensors
        dx, ds, grads = ops.crossray_decode_backward(xp, sp, w, d_rgb.contiguous())
        return (dx, ds) + tuple(g.view_as(t) for g, t in zip(grads, w))


class EncoderFn(torch.autograd.Function):
    """encoder_sameoutputsize.forward with gradients: fwd crnerf_encoder_forward_train_f32, bwd crnerf_encoder_backward_f32."""

    @staticmethod
    def forward(ctx, image, *w):
        out, saved, hw = ops.encoder_forward_train(image, w)
        ctx.save_for_backward(out, saved, *w)
        ctx.hw, ctx.image_shape = hw, image.shape
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, saved, *w = ctx.saved_tensors
        grads, d_img = ops.encoder_backward(w, saved, ctx.hw, out, d_out.contiguous(), want_d_image=ctx.needs_input_grad[0])
        return (d_img.view(ctx.image_shape) if d_img is not None else None,) + tuple(g.view_as(t) for g, t in zip(grads, w))


class ContentDecoderFn(torch.autograd.Function):
    """style_net.forward(content, None, type='content'): fwd crnerf_crossray_decode_f32 (style == NULL),
    bwd crnerf_decoder_content_backward_f32."""

    @staticmethod
    def forward(ctx, xp, rgb_w, rgb_b, all_weights):
        out = ops.crossray_decode(xp, None, all_weights)          # planar [3,HW]
        ctx.save_for_backward(xp, rgb_w, out)
        ctx.w_shape = rgb_w.shape
        return out

    @staticmethod
    def backward(ctx, d_rgb):
        xp, rgb_w, out = ctx.saved_tensors
        dx, dw, db = ops.decoder_content_backward(xp, rgb_w, out, d_rgb.contiguous())
        return dx, dw.view(ctx.w_shape), db, None


# ---------------------------------------------------------------- transient-mask network (models/lightweight_seg.py)
class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d(..., bias=False) on one image: csrc/cgnet.hip forward, dgrad and wgrad."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation, groups):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, dilation, groups)
        return ops.conv2d(x, w, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, d_y):
        x, w = ctx.saved_tensors
        dx, dw = ops.conv2d_backward(x, w, d_y, *ctx.cfg, want_dx=ctx.needs_input_grad[0])
        return dx, dw, None, None, None, None


class BNPReLUFn(torch.autograd.Function):
    """BatchNorm2d(eps) -> PReLU in one pass; `bn` supplies mode, momentum and the running buffers (updated here)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, alpha, bn):
        training = bn.training or bn.running_mean is None
        y, mean, invstd, var_u = ops.bn_prelu(x, gamma, beta, alpha, bn.eps, training, bn.running_mean, bn.running_var)
        if training and bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked += 1
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1.0 - m).add_(var_u, alpha=m)
        ctx.save_for_backward(x, gamma, beta, alpha, mean, invstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, gamma, beta, alpha, mean, invstd = ctx.saved_tensors
        dx, dg, db, da = ops.bn_prelu_backward(x, gamma, beta, alpha, mean, invstd, d_y, ctx.training)
        return dx, dg, db, da, None


class AvgPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape[-3:])
        return ops.avgpool3s2(x)

    @staticmethod
    def backward(ctx, d_y):
        return ops.avgpool3s2_backward(d_y, ctx.shape)


class FGloFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        y, stats = ops.fglo(x, w1, b1, w2, b2)
        ctx.save_for_backward(x, w1, w2, stats)
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, w1, w2, stats = ctx.saved_tensors
        return ops.fglo_backward(x, w1, w2, stats, d_y)


class BilinearGatherFn(torch.autograd.Function):
    """F.interpolate(bilinear, align_corners=False) [+ sigmoid] read at `idx` (None = every pixel)."""

    @staticmethod
    def forward(ctx, x, size, idx, sigmoid):
        out = ops.bilinear_gather(x, size, idx, sigmoid)
        ctx.cfg = (tuple(x.shape[-2:]), tuple(size), sigmoid)
        ctx.save_for_backward(out if sigmoid else None, idx)
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, idx = ctx.saved_tensors
        in_hw, size, sigmoid = ctx.cfg
        return ops.bilinear_gather_backward(out, d_out, in_hw, size, idx, sigmoid), None, None, None
