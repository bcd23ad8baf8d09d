"""torch.autograd glue for training: the reference gets its gradients from PyTorch autograd over eager
ops (train_mask_grid_sample.py:268-337 -> loss.backward()); here every differentiable piece of the
path is one autograd.Function whose backward calls the HIP backward twins.

    MlpFn        NeRF_sigma.forward           fwd: crnerf_mlp_forward_train_f32   bwd: crnerf_mlp_backward_f32
    CompositeFn  inference() compositing      fwd: crnerf_composite_f32           bwd: crnerf_composite_backward_f32
    DecoderFn    style_net.forward            fwd: crnerf_crossray_decode_f32     bwd: INTERIM -- re-evaluates the same
                 math with torch ops on the GPU and differentiates that (0.016 % of the path's FLOPs; the
                 HIP backward twin of the decoder is the next training item)

Inputs that the reference does not train through this path (ray geometry, embeddings, hierarchical
depths -- the latter are .detach()ed at models/rendering.py:184) get no gradient.
"""
import torch
import torch.nn.functional as F

from . import ops


class MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
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


def _decoder_torch(xp, sp, w):
    """The decoder's math with differentiable torch ops (same decomposition as csrc/crossray.hip);
    xp [HW,64], sp [HWs,64], w = the 22 tensors in state_dict order.  Used ONLY inside DecoderFn.backward."""
    (s1, sb1, s2, sb2, s3, sb3, sfw, sfb, c1, cb1, c2, cb2, c3, cb3, cfw, cfb, comp_w, comp_b, unz_w, unz_b, rgb_w, rgb_b) = w

    def matrix(x, a1, b1, a2, b2, a3, b3, fw, fb):
        h = F.leaky_relu(x @ a1.reshape(128, 64).t() + b1, 0.2)
        h = F.leaky_relu(h @ a2.reshape(64, 128).t() + b2, 0.2)
        h = h @ a3.reshape(32, 64).t() + b3
        g = (h.t() @ h) / x.shape[0]
        return (fw @ g.reshape(-1) + fb).view(32, 32)

    c_mean, s_mean = xp.mean(0), sp.mean(0)
    xc, sc = xp - c_mean, sp - s_mean
    T = matrix(sc, s1, sb1, s2, sb2, s3, sb3, sfw, sfb) @ matrix(xc, c1, cb1, c2, cb2, c3, cb3, cfw, cfb)
    comp = xc @ comp_w.reshape(32, 64).t() + comp_b
    fused = (comp @ T.t()) @ unz_w.reshape(64, 32).t() + unz_b + s_mean
    return torch.sigmoid(fused @ rgb_w.reshape(3, 64).t() + rgb_b).t()       # [3, HW]


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, sp, *w):
        out = ops.crossray_decode(xp, sp, w)
        ctx.save_for_backward(xp, sp, *w)
        return out

    @staticmethod
    def backward(ctx, d_rgb):
        xp, sp, *w = ctx.saved_tensors
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_(True) for t in (xp, sp, *w)]
            rgb = _decoder_torch(leaves[0], leaves[1], leaves[2:])
            grads = torch.autograd.grad(rgb, leaves, d_rgb, allow_unused=True)
        return tuple(grads)
