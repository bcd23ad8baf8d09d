"""render_rays_cross_ray with the reference's signature (models/rendering.py:50-63) on the fused HIP
renderer.  Callers: train_mask_grid_sample.py:185-197, eval.py:39-52,
appearance_modification_video.py:82-95 -- all pass the first 11 arguments positionally."""
import torch

from .. import ops
from .nerf import NeRF_sigma, PosEmbedding

__all__ = ['render_rays_cross_ray']

_FUSED_MAX = 256


def _coarse_depths(rays, N_samples, use_disp, perturb):
    """Stratified coarse depths for perturb > 0 (models/rendering.py:161-176), drawn with torch's RNG."""
    near, far = rays[:, 6:7], rays[:, 7:8]
    s = torch.linspace(0, 1, N_samples, device=rays.device)
    z = near * (1 - s) + far * s if not use_disp else 1 / (1 / near * (1 - s) + 1 / far * s)
    z = z.expand(rays.shape[0], N_samples)
    mid = 0.5 * (z[:, :-1] + z[:, 1:])
    upper = torch.cat([mid, z[:, -1:]], -1)
    lower = torch.cat([z[:, :1], mid], -1)
    return (lower + (upper - lower) * (perturb * torch.rand_like(z))).contiguous()


_tables = {}


def _linspace_tables(n_samples, n_importance, device):
    """The reference's two tables, built the reference's way on the same device (rendering.py:160, :27)."""
    key = (n_samples, n_importance, str(device))
    if key not in _tables:
        _tables[key] = (torch.linspace(0, 1, n_samples, device=device),
                        torch.linspace(0, 1, n_importance, device=device) if n_importance > 0 else None)
    return _tables[key]


def _render_unfused(coarse, fine, rays, Nc, Ni, use_disp, view_dir, z_coarse, u, noise_c, noise_f, noise_std, jitter, chunk, train=False,
                    precision="f32"):
    R = rays.shape[0]
    o, d = rays[:, None, 0:3], rays[:, None, 3:6]
    z_steps, u_steps = _linspace_tables(Nc, Ni, rays.device)
    if z_coarse is None:
        near, far = rays[:, 6:7], rays[:, 7:8]
        z_coarse = (near * (1 - z_steps) + far * z_steps) if not use_disp else 1 / (1 / near * (1 - z_steps) + 1 / far * z_steps)
        z_coarse = z_coarse.expand(R, Nc).contiguous()
    demb = ops.posenc((view_dir if view_dir is not None else rays[:, 3:6]).contiguous(), 4)

    def run(model, z, noise):
        N = z.shape[1]
        # point chunks (rendering.py:110-114) only bound the reference's activation memory; the result is chunk-invariant
        # (SURVEY G7), and every MLP call here costs a weight re-pack (+ a wgrad reduction in training), so never go below 2^20 points
        step = max(int(chunk), 1 << 20)
        with torch.no_grad():                                                # embeddings are inputs, not trained
            if jitter:
                pts = (o + d * z[..., None]).reshape(-1, 3)
                pts = pts + 0.00001 * torch.rand_like(pts)                   # rendering.py:102-104
                dirs = demb[:, None, :].expand(R, N, demb.shape[-1]).reshape(R * N, -1)
                xs = [torch.cat([ops.posenc(pts[i:i + step].contiguous(), 15), dirs[i:i + step]], 1) for i in range(0, R * N, step)]
            else:   # the same rows in one pass per ray block (crnerf_embed_points_f32): points, embedding, repeat and cat fused
                rstep = max(step // N, 1)
                zc = z.contiguous()
                xs = [ops.embed_points(rays[i:i + rstep], zc[i:i + rstep], demb[i:i + rstep]) for i in range(0, R, rstep)]
        if train:   # autograd.Functions over the HIP forward/backward twins (autograd.py)
            from ..autograd import CompositeFn, mlp_forward_with_grad
            raw = torch.cat([mlp_forward_with_grad(model, x) for x in xs], 0)
            return CompositeFn.apply(raw.view(R, N, 65), z.contiguous(), noise, noise_std)
        raw = torch.cat([ops.mlp_forward(model.packed_weights(precision), x, precision=precision) for x in xs], 0)   # point chunks, rendering.py:110-114
        return ops.composite(raw.view(R, N, 65), z.contiguous(), noise, noise_std)

    out = {}
    out["weights_coarse"], out["feature_coarse"], out["depth_coarse"] = run(coarse, z_coarse, noise_c)
    if Ni > 0:
        z_fine = ops.sample_pdf_merge(z_coarse.contiguous(), out["weights_coarse"].detach(), Ni,   # .detach(): rendering.py:184
                                      u=u if u is not None else u_steps)
        out["weights_fine"], out["feature_fine"], out["depth_fine"] = run(fine, z_fine, noise_f)
    return out


def _render_bf16_accurate_coarse(coarse, fine, rays, Nc, Ni, use_disp, view_dir, noise_c, noise_f, noise_std, chunk, want_z_fine=False):
    """precision="bf16_hc" (inference, perturb = 0): the COARSE pass in fp32 accuracy on the fp16 matrix cores (precision "auto": f32h2 with the
    f32x3 safety net; 25 % of the points), the FINE pass on the bf16 matrix cores.  Why: bf16's end-to-end pixel error on a trained checkpoint is
    sampling sensitivity -- bf16 coarse weights move the fine depths (weights_fine rel-L2 2.8e-2, tests/test_gpu_trained_ckpt.py) -- so with
    fp32-accurate coarse weights the fine depths are the fp32 reference's and what is left is the fine network's own bf16 rounding.
    Two fused launches (+ the repair kernel that leaves at once): the coarse-only render on the h2 core, then crnerf_render_rays_bf16_fine --
    sample_pdf, merge, embedding, fine MLP and compositing in one kernel on the coarse weights (round 6; rounds 4-5 ran five un-fused calls with
    the [P,120] embeddings and [P,65] raw rows through HBM: 297 ms per 800 x 800 frame)."""
    z_steps, u_steps = _linspace_tables(Nc, Ni, rays.device)
    out = ops.render_rays(coarse.packed_weights("auto"), None, rays, Nc, 0, use_disp=use_disp, view_dir=view_dir, z_steps=z_steps,
                          noise_coarse=noise_c, noise_std=float(noise_std), precision="auto")
    out.update(ops.render_rays_bf16_fine(fine.packed_weights("bf16"), rays, out["weights_coarse"], Nc, Ni, use_disp=use_disp, view_dir=view_dir,
                                         z_steps=z_steps, u=u_steps, noise_fine=noise_f, noise_std=float(noise_std), want_z_fine=want_z_fine))
    return out


def _check_embedding(emb, n_freqs, what):
    if not isinstance(emb, PosEmbedding) or emb.N_freqs != n_freqs:
        raise NotImplementedError("crnerf_amd: embeddings['%s'] must be crnerf_amd PosEmbedding(%d, %d); the fused kernel computes "
                                  "the embedding in registers and cannot call an arbitrary Python callable" % (what, n_freqs - 1, n_freqs))


def render_rays_cross_ray(models, embeddings, rays, ts, N_samples=64, use_disp=False, perturb=0, noise_std=1,
                          N_importance=0, chunk=1024 * 32, white_back=False, test_time=False, **kwargs):
    """Same contract as the reference: returns weights_/feature_/depth_ for 'coarse' and, when
    N_importance > 0, 'fine' (+ 'feature_fine_random', the SAME tensor object as 'feature_fine',
    models/rendering.py:140-141,192).  ts / white_back / test_time / chunk are accepted and, as in the
    reference's arithmetic, do not influence the result (the MLP is point-wise, so chunking is invisible).
    One keyword beyond the reference's: precision="f32"|"bf16"|"f32x3"|"f32h2"|"auto"|"bf16_hc" (default crnerf_amd.get_precision()) selects the
    matrix-core arithmetic of NeRF_sigma at inference (include/crnerf.h).  Grad mode trains through the exact-fp32 twins unless
    the caller opted into mixed precision (autograd.set_training_precision("bf16") / CRNERF_TRAIN_BF16=1: bf16-operand GEMM twins,
    fp32 accumulation; autograd.set_wgrad_precision("bf16"): weight gradients only) -- neither has a counterpart in the reference."""
    args = kwargs['args']
    jitter = bool(getattr(args, 'pertubeCord', False))
    if getattr(args, 'nerf_out_dim', 64) != 64:
        raise NotImplementedError("crnerf_amd: nerf_out_dim must be 64")
    _check_embedding(embeddings['xyz'], 15, 'xyz')
    _check_embedding(embeddings['dir'], 4, 'dir')
    coarse = models['coarse']
    fine = models.get('fine') if N_importance > 0 else None
    for m in (coarse, fine):
        if m is not None and not isinstance(m, NeRF_sigma):
            raise NotImplementedError("crnerf_amd: models must be crnerf_amd NeRF_sigma instances")
    train = torch.is_grad_enabled() and any(ops.any_requires_grad(m) for m in (coarse, fine) if m is not None)
    precision = kwargs.get('precision', None)
    if precision is None:
        from .. import get_precision
        precision = get_precision()
    bf16_hc = precision in ("bf16_hc", "bf16+h2c") and not train
    precision = "f32" if train else ("bf16" if bf16_hc else "auto" if ops._is_auto(precision) else "f32h2" if ops._is_h2(precision) else "f32x3" if ops._is_x3(precision)
                                     else ("bf16" if ops._is_bf16(precision) else "f32"))

    rays = rays.to(torch.float32).contiguous()
    R = rays.shape[0]
    view_dir = kwargs.get('view_dir', None)
    z_coarse = u = noise_c = noise_f = None
    from ..autograd import get_training_bf16, get_training_recompute, get_training_bf16_fused, get_training_forward_x3
    fused_train = (train and not get_training_bf16() and not jitter and N_samples <= _FUSED_MAX and N_importance <= _FUSED_MAX
                   and (N_importance == 0 or N_samples >= 3))
    rng = None
    if fused_train and (perturb > 0 or noise_std != 0) and ops.in_kernel_rng():
        # the fused fp32 training kernel draws the jitter / sample_pdf uniforms / density noise itself (csrc/philox.h): no [R,N] random
        # tensors, no launches for them.  The seed comes from torch's CPU generator, so torch.manual_seed() governs the run.
        # rng_ray_offset (kwarg of this mirror only): the index of rays[0] in the caller's whole batch -- the draws are keyed on (seed, GLOBAL ray,
        # sample), so a batch rendered in ray chunks or sharded over ranks (pipeline.TrainingSystem) draws what one call over all of it would
        rng = {"seed": int(torch.randint(0, 2 ** 62, (1,), device="cpu")), "perturb": float(perturb), "jitter": perturb > 0, "u": perturb > 0,
               "noise": noise_std != 0, "ray_offset": int(kwargs.get("rng_ray_offset", 0))}
    if rng is None and perturb > 0:
        z_coarse = _coarse_depths(rays, N_samples, use_disp, perturb)
        if N_importance > 0:
            u = torch.rand(R, N_importance, device=rays.device)
    if noise_std != 0 and rng is None:
        noise_c = torch.randn(R, N_samples, device=rays.device)
        if N_importance > 0:
            noise_f = torch.randn(R, N_samples + N_importance, device=rays.device)

    if (train and (not get_training_bf16() or get_training_recompute() or get_training_bf16_fused()) and not jitter
            and N_samples <= _FUSED_MAX and N_importance <= _FUSED_MAX
            and (N_importance == 0 or N_samples >= 3)):
        # training: the fused renderer's training twin (one launch per ray chunk: posenc + MLPs + activation save + compositing +
        # sample_pdf/merge) as one autograd node whose backward runs the HIP backward twins (autograd.FusedRenderFn)
        from ..autograd import fused_render_with_grad
        out = fused_render_with_grad(coarse, fine, rays, N_samples, N_importance, use_disp, view_dir, z_coarse, u, noise_c, noise_f,
                                     float(noise_std), rng=rng)
    elif bf16_hc and N_importance > 0 and perturb == 0 and not jitter and 3 <= N_samples <= _FUSED_MAX:
        out = _render_bf16_accurate_coarse(coarse, fine, rays, N_samples, N_importance, use_disp, view_dir, noise_c, noise_f, float(noise_std), int(chunk))
    elif train or N_samples > _FUSED_MAX or N_importance > _FUSED_MAX or jitter:
        # general path: the same HIP kernels, un-fused (posenc -> MLP -> compositing -> sample_pdf/merge), for
        # sample counts beyond the fused kernel's LDS scratch and for args.pertubeCord (rendering.py:102-104)
        out = _render_unfused(coarse, fine, rays, N_samples, N_importance, use_disp, view_dir, z_coarse, u, noise_c, noise_f,
                              float(noise_std), jitter, int(chunk), train, precision)
    else:
        tables = _linspace_tables(N_samples, N_importance, rays.device)
        out = ops.render_rays(coarse.packed_weights(precision), fine.packed_weights(precision) if fine is not None else None, rays,
                              N_samples, N_importance, use_disp=use_disp, view_dir=view_dir, z_coarse=z_coarse,
                              z_steps=tables[0], u=u if u is not None else tables[1],
                              noise_coarse=noise_c, noise_fine=noise_f, noise_std=float(noise_std), precision=precision)

    typ_c = coarse.typ
    results = {'weights_%s' % typ_c: out['weights_coarse'], 'feature_%s' % typ_c: out['feature_coarse'],
               'depth_%s' % typ_c: out['depth_coarse']}
    if N_importance > 0:
        typ_f = fine.typ
        results['weights_%s' % typ_f] = out['weights_fine']
        results['feature_%s' % typ_f] = out['feature_fine']
        if kwargs.get('output_random', True) and fine.encode_random:
            results['feature_fine_random'] = out['feature_fine']
        results['depth_%s' % typ_f] = out['depth_fine']
    return results
