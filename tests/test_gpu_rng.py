"""In-kernel random draws of the fused fp32 renderer (include/crnerf.h CRNERF_RNG_*, csrc/philox.h): the stratified jitter of the
coarse depths (models/rendering.py:169-176), the sample_pdf uniforms of the non-deterministic branch (:30) and the density noise
(:125) are drawn inside render_rays16 / render_rays_train16 from Philox4x32-10 keyed on (seed, stream, global ray, sample).

The reference draws from torch's global generator, whose stream cannot be matched (it differs between the reference's own CPU and
CUDA runs); what must hold is (a) the distributions, (b) the reference's ARITHMETIC on the draws -- checked exactly: the same draws
fed through the tensor arguments (the round-1/2 path, itself held to the oracle by tests/test_gpu_parity.py) give bit-identical
outputs and gradients -- and (c) invariance to ray chunking and to the backward's recomputation.
"""
import numpy as np
import pytest
import torch

import crnerf_amd.synth as synth
from crnerf_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def C(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_draw_statistics_and_keying():
    R, N, seed = 4096, 192, 1234567
    u = ops.rng_fill(R, N, seed, 0).double()
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert torch.equal(u * 2 ** 24, (u * 2 ** 24).round())                       # the 2^-24 grid of torch.rand
    n = u.numel()
    assert abs(float(u.mean()) - 0.5) < 4 * (1 / 12) ** 0.5 / n ** 0.5 and abs(float(u.var()) - 1 / 12) < 1e-3
    g = ops.rng_fill(R, N, seed, 2).double()
    assert abs(float(g.mean())) < 4 / n ** 0.5 and abs(float(g.var()) - 1.0) < 1e-2
    assert abs(float((g ** 4).mean()) - 3.0) < 0.05 and abs(float((g ** 3).mean())) < 0.02   # kurtosis / skewness of a normal
    assert 4.0 < float(g.abs().max()) < 6.5                                        # tails exist and are sane for 786k draws
    # independence: neighbouring rays, neighbouring samples, different streams, different seeds
    cor = lambda a, b: abs(float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std())))  # noqa: E731
    assert cor(u[:-1], u[1:]) < 5e-3 and cor(u[:, :-1], u[:, 1:]) < 5e-3
    assert cor(g[:-1], g[1:]) < 5e-3 and cor(g[:, :-1], g[:, 1:]) < 5e-3
    assert cor(u, ops.rng_fill(R, N, seed, 1).double()) < 5e-3 and cor(g, ops.rng_fill(R, N, seed, 3).double()) < 5e-3
    assert cor(u, ops.rng_fill(R, N, seed + 1, 0).double()) < 5e-3
    # keyed on the GLOBAL ray index: a chunk that starts at ray 100 sees rows 100.. of the whole batch
    assert torch.equal(ops.rng_fill(R - 100, N, seed, 0, ray_offset=100).double(), u[100:])
    assert torch.equal(ops.rng_fill(R, N, seed, 0).double(), u)                   # and it is a pure function of its arguments


def _nets(precision="f32"):
    pack = ops.pack_mlp_weights_x3 if precision == "f32x3" else ops.pack_mlp_weights
    pk = lambda st: pack({k: C(v) for k, v in st.items()})  # noqa: E731
    return pk(synth.mlp_state(11, 2.0, 0.5)), pk(synth.mlp_state(12, 2.0, 0.5))


def _tensor_path_inputs(rays, Nc, Ni, seed, perturb, use_disp=False, ray_offset=0):
    """What models/rendering.py::_coarse_depths does with torch ops (the reference's expression, rendering.py:161-176), on the kernel's draws."""
    R = rays.shape[0]
    near, far = rays[:, 6:7], rays[:, 7:8]
    s = torch.linspace(0, 1, Nc, device=DEV)
    z = near * (1 - s) + far * s if not use_disp else 1 / (1 / near * (1 - s) + 1 / far * s)
    mid = 0.5 * (z[:, :-1] + z[:, 1:])
    upper, lower = torch.cat([mid, z[:, -1:]], -1), torch.cat([z[:, :1], mid], -1)
    z_j = lower + (upper - lower) * (perturb * ops.rng_fill(R, Nc, seed, 0, ray_offset))
    return z_j.contiguous(), lower, upper, ops.rng_fill(R, Ni, seed, 1, ray_offset), ops.rng_fill(R, Nc, seed, 2, ray_offset), ops.rng_fill(R, Nc + Ni, seed, 3, ray_offset)


@torch.no_grad()
@pytest.mark.parametrize("use_disp", [False, True])
@pytest.mark.parametrize("Nc,Ni", [(64, 128), (64, 64), (33, 0)])
@pytest.mark.parametrize("precision", ["f32", "f32x3"])
def test_in_kernel_draws_equal_the_tensor_path_bit_for_bit(Nc, Ni, use_disp, precision):
    """precision = f32x3: the same counters in render_rays_x3 / render_rays_train_x3 (csrc/render_fused_x3.hip) -- one seed, one set of draws."""
    import functools
    pc, pf = _nets(precision)
    render = functools.partial(ops.render_rays, precision=precision)
    rays = C(synth.rays(257, seed=5))
    seed, perturb, nstd = 987654321012, 1.0, 1.0
    z_j, lower, upper, u, n_c, n_f = _tensor_path_inputs(rays, Nc, Ni, seed, perturb, use_disp)
    z_steps = torch.linspace(0, 1, Nc, device=DEV)
    ref = render(pc, pf if Ni else None, rays, Nc, Ni, use_disp=use_disp, z_coarse=z_j, u=u if Ni else None, noise_coarse=n_c,
                          noise_fine=n_f if Ni else None, noise_std=nstd, want_z_fine=True)
    got = render(pc, pf if Ni else None, rays, Nc, Ni, use_disp=use_disp, z_steps=z_steps, noise_std=nstd, want_z_fine=True,
                          rng={"seed": seed, "perturb": perturb, "jitter": True, "u": True, "noise": True})
    assert torch.equal(got["z_coarse_used"], z_j)                                 # the reference's jitter expression, same rounding
    assert bool((got["z_coarse_used"] >= lower).all()) and bool((got["z_coarse_used"] <= upper).all())      # stratum bounds
    assert torch.equal(got["noise_coarse_used"], n_c) and (Ni == 0 or torch.equal(got["noise_fine_used"], n_f))
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    # chunking: the second half of the batch rendered on its own, with its offset, is the second half of the batch
    half = render(pc, pf if Ni else None, rays[128:].contiguous(), Nc, Ni, use_disp=use_disp, z_steps=z_steps, noise_std=nstd,
                           rng={"seed": seed, "perturb": perturb, "jitter": True, "u": True, "noise": True, "ray_offset": 128})
    for k in ("weights_coarse", "feature_coarse") + (("weights_fine", "feature_fine") if Ni else ()):
        assert torch.equal(half[k], got[k][128:]), k
    # noise only (perturb == 0): deterministic depths, in-kernel noise
    ref0 = render(pc, pf if Ni else None, rays, Nc, Ni, use_disp=use_disp, z_steps=z_steps, u=torch.linspace(0, 1, Ni, device=DEV) if Ni else None,
                           noise_coarse=n_c, noise_fine=n_f if Ni else None, noise_std=0.5)
    got0 = render(pc, pf if Ni else None, rays, Nc, Ni, use_disp=use_disp, z_steps=z_steps, u=torch.linspace(0, 1, Ni, device=DEV) if Ni else None,
                           noise_std=0.5, rng={"seed": seed, "noise": True})
    for k in ref0:
        assert torch.equal(got0[k], ref0[k]), k


class _Args:
    nerf_out_dim, img_wh, pertubeCord = 64, [16, 8], False


def _models():
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    m = {"coarse": NeRF_sigma("coarse", _Args(), in_channels_xyz=93, in_channels_dir=27).to(DEV),
         "fine": NeRF_sigma("fine", _Args(), in_channels_xyz=93, in_channels_dir=27, encode_random=True).to(DEV)}
    m["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(21, 2.0, 0.5).items()})
    m["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(22, 2.0, 0.5).items()})
    return m, {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}


@pytest.mark.parametrize("recompute", [False, True])
@pytest.mark.parametrize("forward", ["f32", "f32x3", "auto"])
def test_training_through_the_shim_in_kernel_vs_tensor_draws(recompute, forward):
    """render_rays_cross_ray in grad mode (perturb = 1, noise_std = 1, command/train.sh): the in-kernel path and the tensor path
    fed with the same draws give identical outputs and identical parameter gradients; the recomputing backward re-draws the same
    numbers."""
    from crnerf_amd import autograd as ag
    from crnerf_amd.models.rendering import render_rays_cross_ray
    m, emb = _models()
    rays = C(synth.rays(128, seed=9, H=8, W=16))
    tgt = torch.rand(128, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    params = [p for k in ("coarse", "fine") for p in m[k].parameters()]
    ag.set_training_recompute(recompute)
    ag.set_training_forward_precision(forward)      # f32x3: the x3 training twin draws from the same counters
    try:
        def run(in_kernel):
            ops.set_in_kernel_rng(in_kernel)
            torch.manual_seed(77)
            for p in params:
                p.grad = None
            if in_kernel:
                res = render_rays_cross_ray(m, emb, rays, None, 64, False, 1.0, 1.0, 64, 1 << 20, False, args=_Args())
            else:   # the tensor path on the draws the kernel path will make: same seed, streams 0..3
                seed = int(torch.randint(0, 2 ** 62, (1,), device="cpu"))
                z_j, _, _, u, n_c, n_f = _tensor_path_inputs(rays, 64, 64, seed, 1.0)
                res = ag.fused_render_with_grad(m["coarse"], m["fine"], rays, 64, 64, False, None, z_j, u, n_c, n_f, 1.0)
            loss = ((res["feature_fine"] - tgt) ** 2).mean() + ((res["feature_coarse"] - tgt) ** 2).mean() + res["depth_fine"].mean() * 1e-2
            loss.backward()
            return {k: v.detach().clone() for k, v in res.items()}, [p.grad.clone() for p in params]
        out_k, g_k = run(True)
        out_t, g_t = run(False)
    finally:
        ops.set_in_kernel_rng(True)
        ag.set_training_recompute(False)
        ag.set_training_forward_precision(None)
    for k in ("weights_coarse", "feature_coarse", "depth_coarse", "weights_fine", "feature_fine", "depth_fine"):
        assert torch.equal(out_k[k], out_t[k]), k
    for a, b in zip(g_k, g_t):
        assert torch.equal(a, b)
    assert float(out_k["weights_fine"].sum(-1).mean()) > 0.5 and not torch.equal(out_k["feature_fine"], out_k["feature_coarse"])


def test_a_sharded_batch_draws_what_the_whole_batch_draws():
    """rng_ray_offset: ray-parallel training seeds every rank alike, so the draws must be keyed on the ray's index in the WHOLE batch.
    Two shards of 64 rays rendered under the same seed with their offsets give, ray for ray, the outputs of one call over all 128 rays -- and
    without the offset the second shard would repeat the first one's jitter / noise."""
    from crnerf_amd.models.rendering import render_rays_cross_ray
    m, emb = _models()
    rays = C(synth.rays(128, seed=9, H=8, W=16))

    def run(part, **kw):
        torch.manual_seed(77)
        return render_rays_cross_ray(m, emb, part, None, 64, False, 1.0, 1.0, 64, 1 << 20, False, args=_Args(), **kw)
    whole = run(rays)
    a, b = run(rays[:64]), run(rays[64:], rng_ray_offset=64)
    b0 = run(rays[64:])
    for k in ("feature_fine", "weights_fine", "depth_coarse"):
        assert torch.equal(torch.cat([a[k], b[k]]), whole[k]), k
    assert not torch.equal(b0["feature_fine"], b["feature_fine"])
