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
        ctx.mixed = None
        if get_training_bf16():
            packed, tensors = ops.pack_mlp_weights_mixed(state)
            out, acts = ops.mlp_forward_train_mixed(packed, tensors, x)
            ctx.mixed = packed          # forward and transposed fragment streams: the backward needs no second pack
        else:
            out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(state), x)
        ctx.save_for_backward(x, out, acts, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out, acts, *params = ctx.saved_tensors
        if ctx.mixed is not None:
            tensors = [p.detach() for p in params]
            return (None, None) + tuple(ops.mlp_backward_mixed(ctx.mixed, tensors, x, out, d_out.contiguous(), acts))
        packed_t = ops.pack_mlp_weights_t(dict(zip(ops.MLP_TENSOR_NAMES, params)))
        grads = ops.mlp_backward(packed_t, x, out, d_out.contiguous(), acts, wgrad_bf16=get_wgrad_bf16())
        return (None, None) + tuple(grads)


def mlp_forward_with_grad(module, x):
    params = list(ops.mlp_params(module))
    if x.requires_grad:
        raise NotImplementedError("crnerf_amd: gradients w.r.t. the embedded input of NeRF_sigma are not implemented "
                                  "(the rendering path never needs them: embeddings are functions of fixed ray geometry)")
    return MlpFn.apply(x, module, *params)


_RECOMPUTE = [False]
_WGRAD_BF16 = [None]


def set_wgrad_precision(precision=None):
    """How the weight gradients of NeRF_sigma's 256 x 256 blocks are formed.  None: the default -- "f16x2" behind the h2 data gradient (training
    forward "auto" / "f32h2": the arithmetic of that core -- two fp16 pieces per operand, three piece products, 22 mantissa bits -- on the eight full
    256 x 256 blocks, the delta operands ranged by the largest |delta| of each tensor, which the data gradient leaves behind; a wave that meets an
    operand outside fp16's range redoes its chunk as bf16x3 would have.  tests/test_gpu_parity.py::test_f16x2_*: same distance to a float64
    evaluation as bf16x3; 65,536-ray train.sh step 210-214 -> 200 ms), "bf16x3" everywhere else (CRNERF_WGRAD_F32=1 / CRNERF_WGRAD_BF16=1 /
    CRNERF_WGRAD_BF16X3=1 in the environment choose the others).  Why bf16x3 took over from f32 in round 4: it is fp32-accurate -- against a float64 evaluation the
    split path and the fp32 matrix cores sit at the SAME distance on every tensor (tests/test_gpu_train_fused.py::
    test_wgrad_bf16x3_is_as_accurate_as_the_fp32_matrix_cores) -- and faster: the weight gradient is the third of the training MLP work whose
    operands come from HBM (2 KB per point and layer); the fp32-MFMA kernel sits at 75 % of its peak there for reasons that are NOT power (an
    all-zero-operand run takes the same time, tools/train_energy_probe.py), the split kernel runs the same products on the 16x faster bf16 pipe:
    65,536-ray train.sh step 372 -> 344 ms.  Its products are not the fp32 MFMA's bit for bit (they differ by fp32 summation noise, <= 2.3e-6 of
    a tensor's largest entry), which is why "f32" stays available.
    "f32": every gradient on the fp32 matrix cores, product for product the reference's autograd arithmetic.  "bf16": opt-in mixed precision for the weight
    gradients of every nn.Linear except static_sigma (CRNERF_BWD_WGRAD_BF16, include/crnerf.h: the full 256x256 blocks AND the
    narrow edge blocks of the 93/349/283-wide layers): same fp32 operands, rounded to bf16 in registers, bf16 MFMA with fp32
    accumulation -- that third of the MLP work then runs at HBM speed instead of fp32-MFMA speed.
    "bf16x3": fp32-ACCURATE weight gradients on the bf16 matrix cores (CRNERF_BWD_WGRAD_BF16X3, include/crnerf.h): every fp32 operand of
    the 256 x 256 blocks is split into three bf16 pieces in registers and a product is the sum of the six leading piece products (fp32
    accumulation; the dropped terms are one fp32 rounding), so that third of the MLP work runs at the rate of its operand reads with none
    of the bf16 mode's rounding noise.  Forward, loss, data gradients, biases and all other tensors are unchanged.
    "f16x2": CRNERF_BWD_WGRAD_F16X2 (include/crnerf.h) where the h2 data gradient runs, "bf16x3" elsewhere."""
    _WGRAD_BF16[0] = None if precision is None else (3 if str(precision).lower() in ("f16x2", "h2") else 2 if str(precision).lower() in ("bf16x3", "x3") else
                                                     (1 if ops._is_bf16(precision) else 0))


_TRAIN_BF16 = [False]
_TRAIN_FWD = [None]


def set_training_forward_precision(precision=None):
    """None: the default ("auto", see get_training_forward_mode).  "f32": the grad-mode forward of render_rays_cross_ray on the fp32 matrix cores (crnerf_render_rays_train_f32).  "f32x3": the same
    fp32 forward on the bf16 matrix cores (crnerf_render_rays_train_f32x3: three-piece bf16 splits of every fp32 operand, six MFMAs per
    product -- include/crnerf.h "f32x3"), and with it the data gradient on the same core (crnerf_mlp_backward_x3_f32): same saved state and
    scratch layouts, the weight gradients as set_wgrad_precision says; the stochastic draws in-kernel from the same Philox counters as the fp32 twin.
    "f32h2": forward and data gradient on the fp16 matrix cores from two-piece fp16 splits, three MFMAs per product
    (crnerf_render_rays_train_f32h2 / crnerf_mlp_backward_h2_f32) -- fp32-accurate within fp16's range: a weight >= 255 raises when packing, a
    point whose activations pass 65,504 comes out NaN; the gradient side rescales every point's deltas per layer and has no range limit.
    "auto": "f32h2" with the scale-free f32x3 twins as its safety net -- poisoned ray quads are rendered again (saved rows included) by
    crnerf_render_rays_train_f32x3_repair on the device; a weight >= 255 leaves a flag in the (asynchronous) pack that makes the h2 kernels hand
    the whole step to the f32x3 ones, also on the device: no host round trip, no NaN of the h2 core's making in the loss."""
    _TRAIN_FWD[0] = (None if precision is None else "f32x3" if ops._is_x3(precision) else "f32h2" if ops._is_h2(precision) else
                     "auto" if ops._is_auto(precision) else ("f32" if not ops._is_bf16(precision) else _bad_fwd_precision(precision)))


def _bad_fwd_precision(precision):
    raise ValueError("crnerf_amd: set_training_forward_precision takes 'f32', 'f32x3', 'f32h2' or 'auto' (mixed precision: set_training_precision('bf16')), got %r"
                     % (precision,))


def get_training_forward_mode():
    """"auto" (the default since round 4) | "f32" | "f32x3" | "f32h2" (environment: CRNERF_TRAIN_FWD=f32|x3|h2|auto, or the older CRNERF_TRAIN_FWD_X3=1).
    Why "auto" is the default: it is fp32-accurate (every saved row, output and gradient held to the fp32 twins' bars, tests/test_gpu_h2.py), it
    cannot fail on range (device-side fall-back to the scale-free f32x3 twins), and it is what the part is fast at -- the 65,536-ray train.sh
    step 343 -> 256 ms, the 1,024-ray one 9.1 -> 7.6-8.1 ms.  "f32" keeps every product on the fp32 matrix cores, the reference's arithmetic."""
    import os
    if _TRAIN_FWD[0] is not None:
        return _TRAIN_FWD[0]
    env = os.environ.get("CRNERF_TRAIN_FWD", "").lower()
    if env in ("x3", "f32x3"):
        return "f32x3"
    if env in ("h2", "f32h2"):
        return "f32h2"
    if env in ("f32", "fp32"):
        return "f32"
    if env == "auto":
        return "auto"
    return "f32x3" if os.environ.get("CRNERF_TRAIN_FWD_X3", "") not in ("", "0") else "auto"


def get_training_forward_x3():
    return get_training_forward_mode() == "f32x3"


def _pack_for_training(mode, state):
    return (ops.pack_mlp_weights_x3(state) if mode == "f32x3" else ops.pack_mlp_weights_h2(state) if mode == "f32h2" else
            ops.pack_mlp_weights_auto(state, check=False) if mode == "auto" else ops.pack_mlp_weights(state))


def _effective_mode(mode, packed):
    """What "auto" turned into for this set of packs: the h2 core when every pack was accepted, else f32x3."""
    if mode != "auto":
        return mode
    return "f32x3" if any(pk.h2 is None for pk in packed) else "auto"


def set_training_precision(precision="f32"):
    """"f32" (default): training through the fp32 twins -- the reference's arithmetic.  "bf16": opt-in mixed precision for NeRF_sigma
    in grad mode (include/crnerf.h): forward in the arithmetic of the bf16 inference kernels, data and weight gradients from
    bf16-rounded operands, fp32 accumulation; activations and deltas are kept as bf16 rows.  The forward is the fused bf16 renderer's
    training twin (crnerf_render_rays_train_bf16); the backward runs per-layer bf16 GEMMs (crnerf_mlp_backward_mixed_ex_f32)."""
    _TRAIN_BF16[0] = ops._is_bf16(precision)


def get_training_bf16():
    import os
    return _TRAIN_BF16[0] or os.environ.get("CRNERF_TRAIN_BF16", "") not in ("", "0")


def get_wgrad_bf16(h2=False):
    """0: exact fp32 MFMA; 1: bf16-rounded operands; 2: three-piece bf16 split (fp32-accurate); 3: two-piece fp16 split of the full blocks
    (fp32-accurate, h2 data gradient only: `h2` says whether the caller runs it -- without it 3 means 2)."""
    import os
    if _WGRAD_BF16[0] is not None:
        return int(_WGRAD_BF16[0]) if h2 else min(int(_WGRAD_BF16[0]), 2)
    if os.environ.get("CRNERF_WGRAD_F32", "") not in ("", "0"):
        return 0
    if os.environ.get("CRNERF_WGRAD_BF16X3", "") not in ("", "0"):
        return 2
    if os.environ.get("CRNERF_WGRAD_BF16", "") not in ("", "0"):
        return 1
    return 3 if h2 else 2


def set_training_recompute(flag=True):
    """Memory-bounded training (activation checkpointing per ray chunk): the grad-mode forward of render_rays_cross_ray runs the
    INFERENCE renderer and keeps only rays / depths / noise; backward re-runs the fused training forward of the chunk (one more
    MLP forward = +1/3 of the MLP work) and then the backward twins.  Saved state drops from ~10.5 KB per sample point to < 1 KB."""
    _RECOMPUTE[0] = bool(flag)


def get_training_recompute():
    import os
    return _RECOMPUTE[0] or os.environ.get("CRNERF_TRAIN_RECOMPUTE", "") not in ("", "0")


def _embed_points(rays, z, view_dir):
    """x[P,120] = cat(PosEmbedding_xyz(o + d z), PosEmbedding_dir(d)) for the wgrad of the layers that read it (rendering.py:108-114);
    rebuilt in backward instead of being stored between forward and backward."""
    demb = ops.posenc((view_dir if view_dir is not None else rays[:, 3:6]).contiguous(), 4)
    return ops.embed_points(rays, z, demb)       # one pass, one 480-byte row per point (the torch composition moved ~1.5 KB)


class FusedRenderFn(torch.autograd.Function):
    """render_rays_cross_ray's arithmetic (rendering.py:100-194) for one chunk of rays as ONE differentiable node:
    fwd crnerf_render_rays_train_f32 (posenc + both MLPs + activation save + compositing + sample_pdf/merge in one launch),
    bwd per pass crnerf_composite_backward_f32 -> crnerf_mlp_backward_f32.  weights_coarse -> sample_pdf carries no gradient
    (.detach() at rendering.py:184); rays, depths and noise are inputs, not differentiated."""

    @staticmethod
    def forward(ctx, cfg, rays, *params):
        ctx.set_materialize_grads(False)     # an unused pass (e.g. coarse when only feature_fine feeds the loss) gets None, not zeros
        Nc, Ni = cfg["Nc"], cfg["Ni"]
        n_models = 2 if Ni > 0 else 1
        states = [dict(zip(ops.MLP_TENSOR_NAMES, params[24 * m:24 * m + 24])) for m in range(n_models)]
        for mod in cfg["modules"]:
            if mod is not None and hasattr(mod, "invalidate_packed"):
                mod.invalidate_packed()      # an optimiser step follows (see MlpFn)
        mode = get_training_forward_mode()
        # one set of packs per render call, shared by its ray chunks (fused_render_with_grad; round 6: eight chunks of a 65,536-ray step packed
        # the same weights eight times, forward and transposed: ~130 launches and 0.7 ms)
        shared = cfg.get("shared")
        packed = shared.get(("fwd", mode)) if shared is not None else None
        if packed is None:
            packed = [_pack_for_training(mode, st) for st in states]
            if shared is not None:
                shared[("fwd", mode)] = packed
        recompute = get_training_recompute()
        ctx.mode = _effective_mode(mode, packed)   # the backward follows the forward's core
        out = ops.render_rays(packed[0], packed[1] if Ni > 0 else None, rays, Nc, Ni, use_disp=cfg["use_disp"], view_dir=cfg["view_dir"],
                              z_coarse=cfg["z_coarse"], u=cfg["u"], noise_coarse=cfg["noise_c"], noise_fine=cfg["noise_f"],
                              noise_std=cfg["noise_std"], want_z_fine=True, train=not recompute, rng=cfg.get("rng"), z_steps=cfg.get("z_steps"),
                              precision=mode)
        if cfg.get("rng") is not None:       # what the kernel drew is what the backward composites with (a few KB per ray chunk)
            cfg = dict(cfg, z_coarse_bwd=out["z_coarse_used"], noise_c_bwd=out.get("noise_coarse_used", cfg["noise_c"]),
                       noise_f_bwd=out.get("noise_fine_used", cfg["noise_f"]))
        ctx.cfg, ctx.recompute, ctx.n_models = cfg, recompute, n_models
        keep = [rays, out["z_fine"] if Ni > 0 else rays.new_empty(0)]
        if not recompute:
            keep += [out["acts_coarse"], out["raw_coarse"]] + ([out["acts_fine"], out["raw_fine"]] if Ni > 0 else [])
        ctx.n_keep = len(keep)
        ctx.save_for_backward(*keep, *params)
        # a batch of more rays than hparams.chunk runs this node once per chunk: with deferral on, the chunks' parameter gradients are summed by one
        # multi-tensor add per chunk at the end of backward instead of 48 AccumulateGrad adds per chunk (336 launches of a 65,536-ray step)
        ctx.defer, ctx.leaves = _DEFER_ON[0], ([_leaf_of(t) for t in params] if _DEFER_ON[0] else None)
        res = (out["weights_coarse"], out["feature_coarse"], out["depth_coarse"])
        if Ni > 0:
            res += (out["weights_fine"], out["feature_fine"], out["depth_fine"])
        return res

    @staticmethod
    def backward(ctx, *g):
        cfg = ctx.cfg
        Nc, Ni = cfg["Nc"], cfg["Ni"]
        saved = ctx.saved_tensors
        keep, params = saved[:ctx.n_keep], saved[ctx.n_keep:]
        rays, z_fine = keep[0], keep[1]
        states = [dict(zip(ops.MLP_TENSOR_NAMES, params[24 * m:24 * m + 24])) for m in range(ctx.n_models)]
        shared = cfg.get("shared")

        def cached(key, make):     # the transposed packs of a step's weights: made by the first chunk's backward, used by all of them
            if shared is None:
                return make()
            if key not in shared:
                shared[key] = make()
            return shared[key]
        if ctx.recompute:
            packed = cached(("fwd", ctx.mode), lambda: [_pack_for_training(ctx.mode, st) for st in states])
            out = ops.render_rays(packed[0], packed[1] if Ni > 0 else None, rays, Nc, Ni, use_disp=cfg["use_disp"], view_dir=cfg["view_dir"],
                                  z_coarse=cfg["z_coarse"], u=cfg["u"], noise_coarse=cfg["noise_c"], noise_fine=cfg["noise_f"],
                                  noise_std=cfg["noise_std"], train=True, rng=cfg.get("rng"), z_steps=cfg.get("z_steps"),
                                  precision=ctx.mode)   # same (seed, ray, sample) -> same draws
            per_pass = [(out["acts_coarse"], out["raw_coarse"])] + ([(out["acts_fine"], out["raw_fine"])] if Ni > 0 else [])
            z_fine = out["z_fine"] if Ni > 0 else z_fine
            del out
        else:
            per_pass = [(keep[2], keep[3])] + ([(keep[4], keep[5])] if Ni > 0 else [])
        grads = []
        for m, (acts, raw) in enumerate(per_pass):
            d_w, d_f, d_d = g[3 * m], g[3 * m + 1], g[3 * m + 2]
            z = z_fine if m == 1 else cfg.get("z_coarse_bwd", cfg["z_coarse"])
            noise = cfg.get("noise_f_bwd", cfg["noise_f"]) if m == 1 else cfg.get("noise_c_bwd", cfg["noise_c"])
            if d_w is None and d_f is None and d_d is None:
                grads += [None] * 24
                continue
            if d_f is None:
                d_f = torch.zeros(raw.shape[0], 64, device=raw.device)
            d_raw = ops.composite_backward(raw, z, d_f.contiguous(), None if d_d is None else d_d.contiguous(),
                                           None if d_w is None else d_w.contiguous(), noise=noise, noise_std=cfg["noise_std"])
            x = _embed_points(rays, z, cfg["view_dir"])
            mode = getattr(ctx, "mode", "f32")   # a split-core forward brings the data gradient on the same core with it (set_training_forward_precision)
            x3, h2 = mode == "f32x3", mode in ("f32h2", "auto")
            packed_t = cached(("t", "h2" if h2 else "x3" if x3 else "f32", m),
                              lambda: ops.pack_mlp_weights_t_h2(states[m]) if h2 else (ops.pack_mlp_weights_t_x3(states[m]) if x3 else ops.pack_mlp_weights_t(states[m])))
            net = cached(("t", "x3", m), lambda: ops.pack_mlp_weights_t_x3(states[m])) if mode == "auto" else None      # "auto": the f32x3 data gradient stands by (device-side range flag)
            grads += ops.mlp_backward(packed_t, x, raw.view(-1, 65), d_raw.view(-1, 65), acts, wgrad_bf16=get_wgrad_bf16(h2), dgrad_x3=x3, dgrad_h2=h2,
                                      fallback_t_x3=net)
            del x, d_raw
        if getattr(ctx, "defer", False):
            todo = [(leaf, gr.view(leaf.shape)) for leaf, gr in zip(ctx.leaves, grads) if leaf is not None and gr is not None]
            _defer([t[0] for t in todo], [t[1] for t in todo])
            grads = [None if (leaf is not None or gr is None) else gr for leaf, gr in zip(ctx.leaves, grads)]
        return (None, None) + tuple(grads)


class MixedRecomputeRenderFn(torch.autograd.Function):
    """set_training_precision("bf16") + set_training_recompute(True), one chunk of rays: the forward IS the fused bf16 inference
    renderer (crnerf_render_rays_bf16: ~0.22 ms per 1,024 rays, nothing kept but rays / depths / noise); backward re-runs the chunk through
    the renderer's training twin (crnerf_render_rays_train_bf16 -- bit-identical outputs, so the gradient is taken at exactly the activations
    the loss saw) and then runs MixedFusedRenderFn's backward.  CRNERF_TRAIN_BF16_UNFUSED=1: the round-2 rebuild (embedded points -> per-layer
    GEMM twins), whose outputs differ from the forward's by the bf16 kernels' mutual tolerance (mean 3e-5, tests/test_gpu_bf16.py)."""

    @staticmethod
    def forward(ctx, cfg, rays, *params):
        ctx.set_materialize_grads(False)
        Nc, Ni = cfg["Nc"], cfg["Ni"]
        n_models = 2 if Ni > 0 else 1
        states = [dict(zip(ops.MLP_TENSOR_NAMES, params[24 * m:24 * m + 24])) for m in range(n_models)]
        for mod in cfg["modules"]:
            if mod is not None and hasattr(mod, "invalidate_packed"):
                mod.invalidate_packed()
        packed = [ops.pack_mlp_weights(st, precision="bf16") for st in states]
        out = ops.render_rays(packed[0], packed[1] if Ni > 0 else None, rays, Nc, Ni, use_disp=cfg["use_disp"], view_dir=cfg["view_dir"],
                              z_coarse=cfg["z_coarse"], u=cfg["u"], noise_coarse=cfg["noise_c"], noise_fine=cfg["noise_f"],
                              noise_std=cfg["noise_std"], want_z_fine=True, precision="bf16")
        ctx.cfg, ctx.n_models = cfg, n_models
        ctx.save_for_backward(rays, out["z_fine"] if Ni > 0 else rays.new_empty(0), *params)
        res = (out["weights_coarse"], out["feature_coarse"], out["depth_coarse"])
        if Ni > 0:
            res += (out["weights_fine"], out["feature_fine"], out["depth_fine"])
        return res

    @staticmethod
    def backward(ctx, *g):
        cfg = ctx.cfg
        Nc, Ni = cfg["Nc"], cfg["Ni"]
        rays, z_fine, *params = ctx.saved_tensors
        states = [dict(zip(ops.MLP_TENSOR_NAMES, params[24 * m:24 * m + 24])) for m in range(ctx.n_models)]
        fused = get_training_bf16_fused()
        if fused:
            packed_b = [ops.pack_mlp_weights(st, precision="bf16") for st in states]
            trn = ops.render_rays(packed_b[0], packed_b[1] if Ni > 0 else None, rays, Nc, Ni, use_disp=cfg["use_disp"], view_dir=cfg["view_dir"],
                                  z_coarse=cfg["z_coarse"], u=cfg["u"], noise_coarse=cfg["noise_c"], noise_fine=cfg["noise_f"],
                                  noise_std=cfg["noise_std"], precision="bf16", train=True)
            z_fine = trn["z_fine"] if Ni > 0 else z_fine
        grads = []
        for m in range(ctx.n_models):
            d_w, d_f, d_d = g[3 * m], g[3 * m + 1], g[3 * m + 2]
            if d_w is None and d_f is None and d_d is None:
                grads += [None] * 24
                continue
            z = z_fine if m == 1 else cfg["z_coarse"]
            noise = cfg["noise_f"] if m == 1 else cfg["noise_c"]
            packed, tensors = ops.pack_mlp_weights_mixed(states[m])
            R, N = z.shape
            if fused:
                tag = "fine" if m == 1 else "coarse"
                raw, acts, x = trn["raw_" + tag].view(-1, 65), trn["acts_" + tag], None
            else:
                x = _embed_points(rays, z, cfg["view_dir"])
                raw, acts = ops.mlp_forward_train_mixed(packed, tensors, x)
            if d_f is None:
                d_f = torch.zeros(R, 64, device=raw.device)
            d_raw = ops.composite_backward(raw.view(R, N, 65), z, d_f.contiguous(), None if d_d is None else d_d.contiguous(),
                                           None if d_w is None else d_w.contiguous(), noise=noise, noise_std=cfg["noise_std"])
            grads += ops.mlp_backward_mixed(packed, tensors, x, raw, d_raw.view(-1, 65), acts, fused_acts=fused)
            del x, raw, acts, d_raw
        return (None, None) + tuple(grads)


class MixedFusedRenderFn(torch.autograd.Function):
    """set_training_precision("bf16"), one chunk of rays as ONE differentiable node: fwd crnerf_render_rays_train_bf16 -- the fused bf16
    renderer that also keeps, per pass, the bf16 activation rows / relu bits / embedded input (written from the registers they are born
    in) and the raw MLP outputs; bwd per pass crnerf_composite_backward_f32 -> crnerf_mlp_backward_mixed_ex_f32 (per-layer bf16 GEMMs for
    the data gradients, weight gradients from the stored rows).  The gradient is taken at exactly the activations the loss saw."""

    @staticmethod
    def forward(ctx, cfg, rays, *params):
        ctx.set_materialize_grads(False)
        Nc, Ni = cfg["Nc"], cfg["Ni"]
        n_models = 2 if Ni > 0 else 1
        states = [dict(zip(ops.MLP_TENSOR_NAMES, params[24 * m:24 * m + 24])) for m in range(n_models)]
        for mod in cfg["modules"]:
            if mod is not None and hasattr(mod, "invalidate_packed"):
                mod.invalidate_packed()
        packed = [ops.pack_mlp_weights(st, precision="bf16") for st in states]
        out = ops.render_rays(packed[0], packed[1] if Ni > 0 else None, rays, Nc, Ni, use_disp=cfg["use_disp"], view_dir=cfg["view_dir"],
                              z_coarse=cfg["z_coarse"], u=cfg["u"], noise_coarse=cfg["noise_c"], noise_fine=cfg["noise_f"],
                              noise_std=cfg["noise_std"], want_z_fine=True, precision="bf16", train=True)
        ctx.cfg, ctx.n_models = cfg, n_models
        keep = [out["z_fine"] if Ni > 0 else rays.new_empty(0), out["acts_coarse"], out["raw_coarse"]]
        if Ni > 0:
            keep += [out["acts_fine"], out["raw_fine"]]
        ctx.n_keep = len(keep)
        ctx.save_for_backward(*keep, *params)
        res = (out["weights_coarse"], out["feature_coarse"], out["depth_coarse"])
        if Ni > 0:
            res += (out["weights_fine"], out["feature_fine"], out["depth_fine"])
        return res

    @staticmethod
    def backward(ctx, *g):
        cfg = ctx.cfg
        saved = ctx.saved_tensors
        keep, params = saved[:ctx.n_keep], saved[ctx.n_keep:]
        z_fine = keep[0]
        grads = []
        for m in range(ctx.n_models):
            d_w, d_f, d_d = g[3 * m], g[3 * m + 1], g[3 * m + 2]
            if d_w is None and d_f is None and d_d is None:
                grads += [None] * 24
                continue
            acts, raw = keep[1 + 2 * m], keep[2 + 2 * m]
            z = z_fine if m == 1 else cfg["z_coarse"]
            noise = cfg["noise_f"] if m == 1 else cfg["noise_c"]
            if d_f is None:
                d_f = torch.zeros(raw.shape[0], 64, device=raw.device)
            d_raw = ops.composite_backward(raw, z, d_f.contiguous(), None if d_d is None else d_d.contiguous(),
                                           None if d_w is None else d_w.contiguous(), noise=noise, noise_std=cfg["noise_std"])
            packed, tensors = ops.pack_mlp_weights_mixed(dict(zip(ops.MLP_TENSOR_NAMES, params[24 * m:24 * m + 24])))
            grads += ops.mlp_backward_mixed(packed, tensors, None, raw.view(-1, 65), d_raw.view(-1, 65), acts, fused_acts=True)
            del d_raw
        return (None, None) + tuple(grads)


def get_training_bf16_fused():
    """The mixed-precision mode trains through the fused bf16 renderer (MixedFusedRenderFn) unless CRNERF_TRAIN_BF16_UNFUSED=1 asks for the
    round-2 path (embedded points -> per-layer GEMM twins -> compositing), kept for A/B runs and as the module-level twin."""
    import os
    return os.environ.get("CRNERF_TRAIN_BF16_UNFUSED", "") in ("", "0")


def fused_render_with_grad(coarse, fine, rays, Nc, Ni, use_disp, view_dir, z_coarse, u, noise_c, noise_f, noise_std, rng=None):
    """Grad-mode render of `rays` in ray chunks of <= 2^21 fine sample points (rays are independent; the chunk bounds the
    backward's scratch -- 10 KB of layer deltas per point -- and, in recompute mode, the live activations)."""
    from .models.rendering import _linspace_tables
    R = rays.shape[0]
    z_steps, u_steps = _linspace_tables(Nc, Ni, rays.device)
    jitter_in_kernel = rng is not None and rng.get("jitter")
    if z_coarse is None and not jitter_in_kernel:   # rendering.py:160-167, the reference's own un-fused arithmetic
        near, far = rays[:, 6:7], rays[:, 7:8]
        z_coarse = (near * (1 - z_steps) + far * z_steps) if not use_disp else 1 / (1 / near * (1 - z_steps) + 1 / far * z_steps)
        z_coarse = z_coarse.expand(R, Nc).contiguous()
    names = ops.MLP_TENSOR_NAMES
    params = list(ops.mlp_params(coarse)) + (list(ops.mlp_params(fine)) if Ni > 0 else [])
    import os
    # ray chunks of <= 2^21 fine-pass points (round 6; 2^20 before): half as many launches, partial-sum slabs and kernel tails -- the 65,536-ray
    # step 191.8 -> 186.9 / 188.3 ms, 16,384 rays 50.8 -> 49.5 ms, +11 GB of backward scratch (2.8 M points: no better; the kernels address rows
    # with 32-bit offsets, < 3.9 M points per call).  Recompute mode keeps 2^20: there the chunk IS the step's activation memory.
    max_pts = int(os.environ.get("CRNERF_TRAIN_CHUNK_POINTS", str(1 << (20 if get_training_recompute() else 21))))
    n_chunks = max(1, -(-R * (Nc + Ni) // max_pts))                 # equal chunks of at most max_pts fine-pass points
    step = max(4, -(-(-(-R // n_chunks)) // 4) * 4)
    parts = []
    # packs of this call's weights, made once and shared by the chunks' nodes (forward packs, and the transposed ones their backward makes); the
    # dict dies with the graph.  The weights cannot change between this forward and its backward without autograd objecting to the saved
    # parameters -- except through p.data writes that bypass the version counter (FlatAdam): do not step the optimiser between the two.
    shared = {} if R > step else None
    for lo in range(0, R, step):
        hi = min(R, lo + step)
        sl = lambda t: None if t is None else t[lo:hi].contiguous()  # noqa: E731
        cfg = {"Nc": Nc, "Ni": Ni, "use_disp": use_disp, "view_dir": sl(view_dir), "z_coarse": sl(z_coarse),
               "u": (u_steps if u is None else sl(u)) if Ni > 0 else None, "noise_c": sl(noise_c), "noise_f": sl(noise_f),
               "noise_std": float(noise_std), "modules": (coarse, fine), "shared": shared}
        if rng is not None:
            cfg["rng"] = dict(rng, ray_offset=int(rng.get("ray_offset", 0)) + lo)
            cfg["z_steps"] = z_steps          # the reference's torch.linspace table (rendering.py:160): the jitter is formed from it in-kernel
            if rng.get("u") and Ni > 0:
                cfg["u"] = None
        fn = FusedRenderFn
        if get_training_bf16():
            fn = MixedRecomputeRenderFn if get_training_recompute() else MixedFusedRenderFn
        parts.append(fn.apply(cfg, rays[lo:hi].contiguous(), *params))
    keys = ["weights_coarse", "feature_coarse", "depth_coarse"] + (["weights_fine", "feature_fine", "depth_fine"] if Ni > 0 else [])
    return {k: (parts[0][i] if len(parts) == 1 else torch.cat([p[i] for p in parts], 0)) for i, k in enumerate(keys)}


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


# ---- deferred accumulation of parameter gradients for modules that are called several times per step.  The reference calls its appearance
# encoder three times, its decoder three times and its content encoder twice per training step (train_mask_grid_sample.py:151-226); PyTorch's
# AccumulateGrad then adds every further use's gradient to .grad tensor by tensor -- 72 launches of ~2 us per step.  With deferral ON (read when the
# forward runs: pipeline.TrainingSystem switches it on around its forward unless torch DDP may be listening) the backward of these nodes hands the
# engine None for the parameters, keeps the gradients, and ONE callback at the end of the backward pass sums them with multi-tensor adds and
# writes / accumulates .grad.  Not for torch.autograd.grad() callers, parameter hooks or DDP's reducer: those need AccumulateGrad to run.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda dev: torch.cuda.current_stream(dev).cuda_stream)
_DEFER_ON = [False]
_DEFERRED = {}       # graph-task id -> {id(param): (param, [gradients])}: one entry per backward pass that has deferred something and not finished yet


class deferred_param_grads:
    """with deferred_param_grads(True): ... -- the nodes created inside defer their parameter gradients (see above)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _DEFER_ON[0] = _DEFER_ON[0], self.on
        if self.on and torch._C._current_graph_task_id() < 0:
            reset_deferred()                 # a forward is starting and no backward is running: whatever is pending belongs to a backward that died

    def __exit__(self, *exc):
        _DEFER_ON[0] = self.prev


def reset_deferred():
    """Drop gradients a backward pass left behind when it raised before its end-of-pass callback ran (the engine discards the callback with the
    failed pass).  Nothing of the failed pass reaches .grad -- what torch does with a node that raised."""
    _DEFERRED.clear()


def _defer(params, grads):
    # The callback belongs to ONE backward pass: the engine drops it when a later node of that pass raises (HIP error, OOM, KeyboardInterrupt).  The
    # queued state is therefore keyed to the graph task's id, one dict per pass: a nested or re-entrant backward (torch.utils.checkpoint with
    # use_reentrant=True, a .backward() inside a Function.backward) has its own id, its own dict and its own callback, and leaves the outer pass's
    # pending gradients alone (a single slot made the inner pass discard them).  What a dead pass left behind is dropped when the next
    # forward starts (deferred_param_grads.__enter__) -- never by another pass.
    task = torch._C._current_graph_task_id()
    pend = _DEFERRED.get(task)
    if pend is None:
        pend = _DEFERRED[task] = {}
        torch.autograd.Variable._execution_engine.queue_callback(lambda task=task: _flush_deferred(task))
    for p, g in zip(params, grads):
        pend.setdefault(id(p), (p, []))[1].append(g)
    if grads and grads[0].is_cuda:
        # the node may have run on a side stream (pipeline.TrainingSystem's branch streams: a node's backward runs on its forward's stream); the
        # end-of-pass sum runs on the stream backward() was called on and must wait for these gradients -- they never pass an AccumulateGrad
        # node, so the engine's own end-of-pass stream synchronisation does not know them.  One Stream object per distinct raw stream and pass
        # (the raw handle is a cheap lookup; torch.cuda.current_stream() is not): the flush waits for everything enqueued on it so far
        raw = _raw_stream(grads[0].device.index)
        streams = pend.setdefault("__streams__", {})
        if raw not in streams:
            streams[raw] = torch.cuda.current_stream(grads[0].device)


def _flush_deferred(task):
    pend = _DEFERRED.pop(task, None)
    if not pend:
        return
    side = pend.pop("__streams__", {})
    if side:
        cur = torch.cuda.current_stream()
        for st in side.values():
            if st != cur:
                cur.wait_stream(st)
    if not pend:
        return
    with torch.no_grad():
        total = {pid: gs[0] for pid, (_, gs) in pend.items()}
        for k in range(1, max(len(gs) for _, gs in pend.values())):       # use k + 1 of every parameter that has one: one multi-tensor add
            ids = [pid for pid, (_, gs) in pend.items() if len(gs) > k]
            torch._foreach_add_([total[i] for i in ids], [pend[i][1][k] for i in ids])
        had = [(p, total[pid]) for pid, (p, _) in pend.items() if p.grad is not None]
        for pid, (p, _) in pend.items():
            if p.grad is None:
                p.grad = total[pid]
        if had:                                                           # gradient accumulation over several backward passes
            torch._foreach_add_([p.grad for p, _ in had], [g for _, g in had])


def _leaf_of(t):
    """The Parameter a tensor handed to a node stands for: itself, or the leaf it is a same-size view of (conv weights travel as [cout, cin] views
    of their [cout, cin, 1, 1] parameters); None: anything else keeps going through the engine."""
    if t.is_leaf:
        return t if t.requires_grad else None          # a frozen Parameter is a leaf too: AccumulateGrad never gives it a .grad, neither may the deferral
    base = t._base if t._is_view() else None
    return base if (base is not None and base.is_leaf and base.requires_grad and base.numel() == t.numel() and base.is_contiguous() and t.is_contiguous()) else None


def _param_grads(ctx, params, grads):
    """What a multi-use node returns for its parameters: the gradients, or -- deferred -- None for every tensor whose leaf is known (ctx.leaves)."""
    grads = [g if g.shape == t.shape else g.view_as(t) for g, t in zip(grads, params)]
    if not getattr(ctx, "defer", False):
        return tuple(grads)
    keep = [(leaf, g if g.shape == leaf.shape else g.view(leaf.shape)) for leaf, g in zip(ctx.leaves, grads) if leaf is not None]
    _defer([k[0] for k in keep], [k[1] for k in keep])
    return tuple(None if leaf is not None else g for leaf, g in zip(ctx.leaves, grads))


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, sp, *w):
        out = ops.crossray_decode(xp, sp, w)
        ctx.save_for_backward(xp, sp, *w)
        ctx.defer, ctx.leaves = _DEFER_ON[0], ([_leaf_of(t) for t in w] if _DEFER_ON[0] else None)
        return out

    @staticmethod
    def backward(ctx, d_rgb):
        xp, sp, *w = ctx.saved_tensors
        dx, ds, grads = ops.crossray_decode_backward(xp, sp, w, d_rgb.contiguous())
        return (dx, ds) + _param_grads(ctx, w, grads)


class EncoderFn(torch.autograd.Function):
    """encoder_sameoutputsize.forward with gradients: fwd crnerf_encoder_forward_train_f32, bwd crnerf_encoder_backward_f32."""

    @staticmethod
    def forward(ctx, image, *w):
        out, saved, hw = ops.encoder_forward_train(image, w)
        ctx.save_for_backward(out, saved, *w)
        ctx.hw, ctx.image_shape = hw, image.shape
        ctx.defer, ctx.leaves = _DEFER_ON[0], ([_leaf_of(t) for t in w] if _DEFER_ON[0] else None)     # (the Parameter objects: .grad is theirs)
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, saved, *w = ctx.saved_tensors
        grads, d_img = ops.encoder_backward(w, saved, ctx.hw, out, d_out.contiguous(), want_d_image=ctx.needs_input_grad[0])
        return (d_img.view(ctx.image_shape) if d_img is not None else None,) + _param_grads(ctx, w, grads)


class DecodeShardedFn(torch.autograd.Function):
    """style_net.forward in ray-parallel training with the content grid SHARDED (round 6): every rank decodes its own block of pixels
    -- the forward is parallel.decode_sharded's three phases around two all-reduces (64 channel sums, 1,024 Gram sums) -- and the RGB planes are
    all-gathered (12 B per pixel), because what consumes the image (the loss, the encoder passes) is replicated.  Backward: every rank holds
    dL/d(image), takes its pixels' columns and runs crnerf_crossray_decode_backward_sharded_f32's three phases around two all-reduces (the
    gradient of the folded affine: 320 floats; the column sums of the centred chain's input gradient: 64 floats).  d_content stays local (it feeds
    this rank's renderer: no feature all-gather any more), d_style and every gradient except the content chain's three convolutions come out
    whole on every rank; those six tensors are this rank's part and leave multiplied by the group size, so that the AVERAGE that
    sync_ray_parallel_gradients takes over replicated modules is the sum of the parts (cf. BandEncoderFn)."""

    @staticmethod
    def forward(ctx, xp, sp, group, n_total, *w):
        import torch.distributed as dist
        from .parallel import _timed
        ws = dist.get_world_size(group)
        dev = xp.device
        xchg = torch.zeros(64 + 1024, dtype=torch.float32, device=dev)
        count = float(n_total)
        ops.crossray_decode_sharded(xp, sp, w, 0, xchg, count)
        _timed("allreduce_channel_sums_64f", lambda: dist.all_reduce(xchg[:64], group=group), dev)
        ops.crossray_decode_sharded(xp, sp, w, 1, xchg, count)
        _timed("allreduce_gram_1024f", lambda: dist.all_reduce(xchg[64:], group=group), dev)
        rgb_local = ops.crossray_decode_sharded(xp, sp, w, 2, xchg, count)
        full = torch.empty(ws * 3, xp.shape[0], dtype=torch.float32, device=dev)
        _timed("allgather_rgb_12B_per_pixel", lambda: dist.all_gather_into_tensor(full, rgb_local.contiguous(), group=group), dev)
        ctx.save_for_backward(xp, sp, xchg, *w)
        ctx.group, ctx.count, ctx.rank, ctx.ws = group, count, dist.get_rank(group), ws
        ctx.defer, ctx.leaves = _DEFER_ON[0], ([_leaf_of(t) for t in w] if _DEFER_ON[0] else None)
        return full.view(ws, 3, xp.shape[0]).permute(1, 0, 2).reshape(3, ws * xp.shape[0])

    @staticmethod
    def backward(ctx, d_rgb):
        import torch.distributed as dist
        from .parallel import _timed
        xp, sp, xchg, *w = ctx.saved_tensors
        n = xp.shape[0]
        d_local = d_rgb[:, ctx.rank * n:(ctx.rank + 1) * n].contiguous()
        xb = torch.zeros(384, dtype=torch.float32, device=xp.device)
        st = ops.crossray_decode_backward_sharded(xp, sp, w, d_local, 0, xchg, ctx.count, xb)
        _timed("allreduce_affine_gradient_320f", lambda: dist.all_reduce(xb[:320], group=ctx.group), xp.device)
        ops.crossray_decode_backward_sharded(xp, sp, w, d_local, 1, xchg, ctx.count, xb, st)
        _timed("allreduce_column_sums_64f", lambda: dist.all_reduce(xb[320:], group=ctx.group), xp.device)
        _, dx, ds, grads = ops.crossray_decode_backward_sharded(xp, sp, w, d_local, 2, xchg, ctx.count, xb, st)
        grads = [g * float(ctx.ws) if 8 <= i <= 13 else g for i, g in enumerate(grads)]
        return (dx, ds, None, None) + _param_grads(ctx, w, grads)


class ContentDecodeShardedFn(torch.autograd.Function):
    """style_net.forward(content, None, type="content") on a ray-sharded grid: per pixel, no statistics -- decode the local pixels, all-gather RGB;
    backward on the local pixels, the two parameter gradients (sums over pixels) leave multiplied by the group size (see DecodeShardedFn)."""

    @staticmethod
    def forward(ctx, xp, rgb_w, rgb_b, group, all_weights):
        import torch.distributed as dist
        from .parallel import _timed
        ws = dist.get_world_size(group)
        out = ops.crossray_decode(xp, None, all_weights)          # planar [3, n_local]
        full = torch.empty(ws * 3, xp.shape[0], dtype=torch.float32, device=xp.device)
        _timed("allgather_rgb_12B_per_pixel", lambda: dist.all_gather_into_tensor(full, out.contiguous(), group=group), xp.device)
        ctx.save_for_backward(xp, rgb_w, out)
        ctx.w_shape, ctx.rank, ctx.ws = rgb_w.shape, dist.get_rank(group), ws
        ctx.defer, ctx.leaves = _DEFER_ON[0], ([_leaf_of(rgb_w), _leaf_of(rgb_b)] if _DEFER_ON[0] else None)
        return full.view(ws, 3, xp.shape[0]).permute(1, 0, 2).reshape(3, ws * xp.shape[0])

    @staticmethod
    def backward(ctx, d_rgb):
        xp, rgb_w, out = ctx.saved_tensors
        n = xp.shape[0]
        dx, dw, db = ops.decoder_content_backward(xp, rgb_w, out, d_rgb[:, ctx.rank * n:(ctx.rank + 1) * n].contiguous())
        gw, gb = _param_grads(ctx, (rgb_w, db), (dw.view(ctx.w_shape) * float(ctx.ws), db * float(ctx.ws)))
        return dx, gw, gb, None, None


class BandEncoderFn(torch.autograd.Function):
    """encoder_sameoutputsize.forward on a re-rendered image in ray-parallel training (round 6): every rank of `group` holds the whole
    image (the decoder is replicated) but runs the seven layers on ITS band of rows only (+ a 12-row halo where the band is cut inside the image:
    crnerf_encoder_forward_train_band_f32) and produces its rows of the 32 x 32 style grid; the grid is all-gathered (32 KB per rank).  Backward:
    every rank holds dL/d(grid) (the consumers are replicated), takes its rows, runs the band backward, and the bands' image gradients -- they
    overlap in the halos -- are summed by one all-reduce of the image-sized buffer (12 B per pixel).  The weight gradients are this band's PART of
    the whole image's; they leave multiplied by the group size (exact: a power of two), because the encoders' other pass (the photo, replicated on
    every rank) yields whole gradients and sync_ray_parallel_gradients AVERAGES replicated modules: average(whole + ws * part_r) = whole + sum of
    the parts."""

    @staticmethod
    def forward(ctx, image, plan, group, *w):
        import torch.distributed as dist
        from .parallel import _timed
        H, row0, rows, o0, o1, ws = plan
        sub = image[0, :, row0:row0 + rows, :].contiguous()
        own, saved, hw = ops.encoder_forward_train_band(sub, H, row0, o0, o1, w)
        full = torch.empty(1024, 64, dtype=torch.float32, device=own.device)
        _timed("allgather_style_grid_rows", lambda: dist.all_gather_into_tensor(full, own, group=group), own.device)
        ctx.save_for_backward(own, saved, *w)
        ctx.hw, ctx.plan, ctx.group, ctx.image_shape = hw, plan, group, image.shape
        ctx.defer, ctx.leaves = _DEFER_ON[0], ([_leaf_of(t) for t in w] if _DEFER_ON[0] else None)
        return full

    @staticmethod
    def backward(ctx, d_out):
        import torch.distributed as dist
        from .parallel import _timed
        own, saved, *w = ctx.saved_tensors
        H, row0, rows, o0, o1, ws = ctx.plan
        want = ctx.needs_input_grad[0]
        grads, d_sub = ops.encoder_backward_band(w, saved, ctx.hw, H, row0, o0, o1, own, d_out[o0 * 32:o1 * 32].contiguous(), want_d_image=want)
        d_img = None
        if want:
            d_img = torch.zeros(ctx.image_shape, dtype=torch.float32, device=own.device)
            d_img[0, :, row0:row0 + rows, :] = d_sub
            _timed("allreduce_image_gradient_12B_per_pixel", lambda: dist.all_reduce(d_img, group=ctx.group), own.device)
        grads = [g * float(ws) for g in grads]
        return (d_img, None, None) + _param_grads(ctx, w, grads)


class ContentDecoderFn(torch.autograd.Function):
    """style_net.forward(content, None, type='content'): fwd crnerf_crossray_decode_f32 (style == NULL),
    bwd crnerf_decoder_content_backward_f32."""

    @staticmethod
    def forward(ctx, xp, rgb_w, rgb_b, all_weights):
        out = ops.crossray_decode(xp, None, all_weights)          # planar [3,HW]
        ctx.save_for_backward(xp, rgb_w, out)
        ctx.w_shape = rgb_w.shape
        ctx.defer, ctx.leaves = _DEFER_ON[0], ([_leaf_of(rgb_w), _leaf_of(rgb_b)] if _DEFER_ON[0] else None)   # the decoder's fourth use of these two
        return out

    @staticmethod
    def backward(ctx, d_rgb):
        xp, rgb_w, out = ctx.saved_tensors
        dx, dw, db = ops.decoder_content_backward(xp, rgb_w, out, d_rgb.contiguous())
        gw, gb = _param_grads(ctx, (rgb_w, db), (dw.view(ctx.w_shape), db))
        return dx, gw, gb, None


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
        fused_update = (training and bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None
                        and bn.running_mean.dtype == torch.float32 and bn.running_mean.is_cuda
                        and (bn.num_batches_tracked is None or (bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64)))
        y, mean, invstd, var_u = ops.bn_prelu(x, gamma, beta, alpha, bn.eps, training, bn.running_mean, bn.running_var,
                                              update=(bn.momentum, bn.num_batches_tracked) if fused_update else None)
        if training and bn.track_running_stats and bn.running_mean is not None and not fused_update:
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


class CGNetFn(torch.autograd.Function):
    """Context_Guided_Network(classes=1, M=2, N=2) in train mode as ONE autograd node: forward = crnerf_cgnet_forward_train_f32, backward =
    crnerf_cgnet_backward_f32 (csrc/cgnet_chain.hip: the operator kernels of the per-module path enqueued back to back, no torch.cat / add /
    gradient-sum launches, no autograd node per layer).  `bns`: the 14 BatchNorm2d modules (running buffers); the image receives no gradient."""

    @staticmethod
    def forward(ctx, image, bns, *params):
        mask, saved = ops.cgnet_forward_train(image, params, [b.running_mean for b in bns], [b.running_var for b in bns],
                                              [b.num_batches_tracked for b in bns], bns[0].momentum, bns[0].eps)
        ctx.save_for_backward(image, mask, saved, *params)
        return mask

    @staticmethod
    def backward(ctx, d_mask):
        image, mask, saved, *params = ctx.saved_tensors
        return (None, None) + tuple(ops.cgnet_backward(image, params, saved, mask, d_mask))
