"""The fused TRAINING renderer (crnerf_render_rays_train_f32 + autograd.FusedRenderFn): one launch does what the reference does
under autograd in render_rays_cross_ray (rendering.py:100-194) and keeps the activations the backward twins need.  Checked
against the inference renderer (same outputs, bit for bit), the stand-alone training twins (same saved activations), the
un-fused grad path, and -- through tests/test_gpu_parity.py::test_training_step_gradients_vs_autograd_oracle, which now runs
this path -- torch autograd through the oracle."""
import numpy as np
import pytest
import torch

import crnerf_amd.synth as synth
from crnerf_amd import autograd as AG
from crnerf_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def C(a):
    return T(np.ascontiguousarray(a)).to(DEV)


class _Args:
    nerf_out_dim, img_wh, pertubeCord = 64, [8, 8], False


def _inputs(R, Nc, Ni, seed=0):
    rng = np.random.default_rng(seed)
    rays = synth.rays(R, seed=seed)
    z = np.sort(rng.uniform(rays[:, 6:7], rays[:, 7:8], (R, Nc)).astype(np.float32), -1)
    u = rng.uniform(0, 1, (R, max(Ni, 1))).astype(np.float32)
    return C(rays), C(z), C(u), C(rng.normal(size=(R, Nc)).astype(np.float32)), C(rng.normal(size=(R, Nc + Ni)).astype(np.float32))


@torch.no_grad()
@pytest.mark.parametrize("R,Nc,Ni", [(37, 64, 64), (5, 33, 20), (16, 64, 0), (9, 256, 256)])
def test_train_forward_equals_inference_and_standalone_twins(R, Nc, Ni):
    st_c, st_f = {k: C(v) for k, v in synth.mlp_state(5, 2.0, 0.5).items()}, {k: C(v) for k, v in synth.mlp_state(6, 2.0, 0.5).items()}
    pc, pf = ops.pack_mlp_weights(st_c), ops.pack_mlp_weights(st_f)
    rays, z, u, nc, nf = _inputs(R, Nc, Ni)
    kw = dict(z_coarse=z, u=u if Ni else None, noise_coarse=nc, noise_fine=nf if Ni else None, noise_std=0.7)
    inf = ops.render_rays(pc, pf if Ni else None, rays, Nc, Ni, want_z_fine=True, **kw)
    trn = ops.render_rays(pc, pf if Ni else None, rays, Nc, Ni, train=True, **kw)
    for k in inf:
        assert torch.equal(inf[k], trn[k]), k                        # the hooks do not touch the arithmetic
    # what was saved == what the stand-alone training forward saves for the same points
    for tag, pk, zz, N in (("coarse", pc, z, Nc),) + ((("fine", pf, trn["z_fine"], Nc + Ni),) if Ni else ()):
        x = AG._embed_points(rays, zz, None)
        out, acts = ops.mlp_forward_train(pk, x)
        raw = trn["raw_" + tag].view(-1, 65)
        # embeddings: in-register sincosf vs the posenc kernel's -- same routine, same arguments
        assert float((raw - out).abs().max()) <= 2e-6, tag
        a_f, a_s = trn["acts_" + tag], acts
        n_act = 10 * R * N * 256
        fa, fs = a_f[:4 * n_act].view(torch.float32).view(10, R * N, 256), a_s[:4 * n_act].view(torch.float32).view(10, R * N, 256)
        assert float((fa[:9] - fs[:9]).abs().max()) <= 2e-5 * float(fs[:9].abs().max()) and float((fa[9, :, :128] - fs[9, :, :128]).abs().max()) <= 2e-5
        # relu bits are consistent with the saved activations of the same buffer
        bits = a_f[4 * n_act:].view(torch.int64)[:10 * R * N * 4].view(10, R * N, 4)      # (behind the bits: the 256-byte line of the range word, kernels.h)
        assert int(a_f[4 * n_act + 320 * R * N:][:4].view(torch.int32)) == 0                    # the fp32 twin tracks no operand range: the word is zeroed
        g = 2
        k = torch.arange(64, device=DEV)
        feat = 16 * (k // 4) + 4 * g + (k % 4)
        on = ((bits[3, :, g, None] >> k) & 1).bool()
        assert torch.equal(on, fa[3][:, feat] > 0)


def _modules(seed_c=41, seed_f=42, **net):
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    args = _Args()
    if net:
        models = {"coarse": NeRF_sigma("coarse", args, in_channels_xyz=93, in_channels_dir=27).to(DEV),
                  "fine": NeRF_sigma("fine", args, in_channels_xyz=93, in_channels_dir=27, encode_random=True).to(DEV)}
        models["coarse"].load_state_dict({k: T(v) for k, v in synth.mlp_state(seed_c, **net).items()})
        models["fine"].load_state_dict({k: T(v) for k, v in synth.mlp_state(seed_f, **net).items()})
        return models, {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}, args
    models = {"coarse": NeRF_sigma("coarse", args, in_channels_xyz=93, in_channels_dir=27).to(DEV),
              "fine": NeRF_sigma("fine", args, in_channels_xyz=93, in_channels_dir=27, encode_random=True).to(DEV)}
    models["coarse"].load_state_dict({k: T(v) for k, v in synth.mlp_state(seed_c, 2.0, 0.5).items()})
    models["fine"].load_state_dict({k: T(v) for k, v in synth.mlp_state(seed_f, 2.0, 0.5).items()})
    return models, {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}, args


def _grads(models, fn):
    for m in models.values():
        m.zero_grad(set_to_none=True)
    fn().backward()
    return {"%s.%s" % (t, n): p.grad.clone() for t, m in models.items() for n, p in m.named_parameters()}


@pytest.mark.parametrize("mode", ["f32", "auto"])
@pytest.mark.parametrize("R", [64, 8200])     # 8200 rays x 128 samples: two ray chunks of the fused path
def test_fused_grad_path_matches_unfused_twins_and_recompute_mode(R, mode, monkeypatch):
    """mode "f32": the fused training renderer against the un-fused fp32 twins -- the same arithmetic, 3e-4 of each gradient's largest entry.
    mode "auto" (the default forward / data-gradient mode since round 4: the h2 core with its f32x3 safety net): fp32-accurate, not the same
    products, and these gain-2 nets turn a 1e-7 difference of a coarse weight into another fine depth -- 3e-2 (measured 1.0e-2 at 64 rays, 4e-3 at
    8,200); recompute stays bit-identical -- over the same ray chunks (round 6: the default chunk is 2^21 points, recompute mode keeps 2^20; the
    chunking decides the summation order of the weight gradients, so both legs are pinned to 2^20 here: 8,200 rays are two chunks)."""
    from crnerf_amd.models import rendering
    monkeypatch.setenv("CRNERF_TRAIN_CHUNK_POINTS", str(1 << 20))
    AG.set_training_forward_precision(mode)
    try:
        _fused_vs_unfused(R, 3e-4 if mode == "f32" else 3e-2, rendering)
    finally:
        AG.set_training_forward_precision(None)


def _fused_vs_unfused(R, bar, rendering):
    models, emb, args = _modules()
    rays, z, u, nc, nf = _inputs(R, 64, 64, seed=3)
    gw = torch.randn(R, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    gd = torch.randn(R, generator=torch.Generator().manual_seed(2)).to(DEV)

    def loss_of(out):
        return ((out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum() + (out["depth_fine"] * gd).sum()
                + 0.1 * (out["weights_fine"] ** 2).sum())

    def fused():
        return loss_of(AG.fused_render_with_grad(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0))

    def unfused():
        return loss_of(rendering._render_unfused(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0, False, 1 << 20,
                                                 train=True))
    g_f = _grads(models, fused)
    g_u = _grads(models, unfused)
    for k in g_u:
        scale = float(g_u[k].abs().max()) + 1e-12
        assert float((g_f[k] - g_u[k]).abs().max()) <= bar * scale, (k, float((g_f[k] - g_u[k]).abs().max()), scale)
    AG.set_training_recompute(True)
    try:
        g_r = _grads(models, fused)
    finally:
        AG.set_training_recompute(False)
    for k in g_f:
        assert torch.equal(g_r[k], g_f[k]), k       # recompute re-runs the identical forward: identical gradients


def test_reference_signature_grad_mode_uses_fused_training_renderer():
    from crnerf_amd.models.rendering import render_rays_cross_ray
    models, emb, args = _modules()
    rays = C(synth.rays(32, seed=1, H=4, W=8))
    torch.manual_seed(0)
    res = render_rays_cross_ray(models, emb, rays, None, 64, False, 1.0, 1.0, 64, 4096, False, args=args)
    assert res["feature_fine"].requires_grad and res["feature_fine"].grad_fn.__class__.__name__ == "FusedRenderFnBackward"
    assert res["feature_fine_random"] is res["feature_fine"]
    res["feature_fine"].sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in models["fine"].parameters())
    assert all(p.grad is None for p in models["coarse"].parameters())          # weights_coarse -> sample_pdf is detached (rendering.py:184)


def test_wgrad_bf16_option_changes_only_the_weight_gradients():
    """CRNERF_BWD_WGRAD_BF16 (opt-in): dW of every nn.Linear except static_sigma from bf16-rounded operands with fp32 accumulation --
    close to the exact fp32 gradient (rounding noise 2^-8 per product, averaged over the points); bias gradients still sum the
    un-rounded deltas; static_sigma is bit-identical."""
    n = 8192
    st = synth.mlp_state(13, 2.0, 0.5)
    dev_state = {k: C(v) for k, v in st.items()}
    rng = np.random.default_rng(5)
    x = AG._embed_points(C(synth.rays(64, seed=4)), C(np.sort(rng.uniform(0.5, 4.5, (64, n // 64)).astype(np.float32), -1)), None)
    d_out = C(rng.normal(size=(n, 65)).astype(np.float32))
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x)
        pt = ops.pack_mlp_weights_t(dev_state)
        exact = ops.mlp_backward(pt, x, out, d_out, acts)
        mixed = ops.mlp_backward(pt, x, out, d_out, acts, wgrad_bf16=True)
    for name, ge, gm in zip(ops.MLP_TENSOR_NAMES, exact, mixed):
        if name.startswith("static_sigma"):
            assert torch.equal(ge, gm), name                        # the 1 x 256 head stays on the fp32 path
        elif name.endswith("weight"):
            scale = float(ge.abs().max())
            err = float((gm - ge).abs().max())
            rel = float((gm - ge).norm() / ge.norm())
            assert 0 < err <= 1e-2 * scale and rel <= 5e-3, (name, err, scale, rel)
        else:
            # bias gradients ride along in the same launches: summed from the un-rounded fp32 deltas, in another order
            assert float((ge - gm).abs().max()) <= 1e-5 * float(ge.abs().max()) + 1e-7, name


@pytest.mark.parametrize("n,gain", [(1, 1.0), (37, 1.0), (1000, 1.0), (4133, 2.0)])
def test_mixed_precision_training_twins_vs_bf16_training_oracle(n, gain):
    """crnerf_mlp_forward_train_mixed_f32 / crnerf_mlp_backward_mixed_f32 (opt-in): forward = the bf16 inference arithmetic
    (oracle mlp_forward_bf16), backward = torch autograd through the oracle's bf16-operand Linear (oracle mlp_forward_bf16_train)."""
    from oracle import cpu_ref as O
    st = synth.mlp_state(17, gain, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(T(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15), O.posenc(T(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    d_out = T(rng.normal(size=(n, 65)).astype(np.float32))
    w = {k: T(v).clone().requires_grad_(True) for k, v in st.items()}
    ref = O.mlp_forward_bf16_train(w, x)
    (ref * d_out).sum().backward()
    with torch.no_grad():
        packed, tensors = ops.pack_mlp_weights_mixed({k: C(v) for k, v in st.items()})
        out, acts = ops.mlp_forward_train_mixed(packed, tensors, x.to(DEV))
        # forward: summation order + the rare bf16 rounding flip of an intermediate activation (tests/test_gpu_bf16.py bounds)
        err = (out.cpu() - ref.detach()).abs()
        print("forward max %.3e mean %.3e" % (float(err.max()), float(err.mean())))
        assert float(err.max()) <= (1e-3 if gain == 1.0 else 6e-2) and float(err.mean()) <= (2e-6 if gain == 1.0 else 1e-4), (float(err.max()), float(err.mean()))
        grads = ops.mlp_backward_mixed(packed, tensors, x.to(DEV), out, d_out.to(DEV), acts)
    for name, g in zip(ops.MLP_TENSOR_NAMES, grads):
        r = w[name].grad
        scale = float(r.abs().max()) + 1e-12
        rel = float((g.cpu() - r).norm() / (r.norm() + 1e-30))
        print("%-28s max|d| %.3e of %.3e  rel-L2 %.3e" % (name, float((g.cpu() - r).abs().max()), scale, rel))
        # weight gradients: rounding noise of the operands (the narrow blocks are multiplied un-rounded here, rounded in the oracle)
        assert float((g.cpu() - r).abs().max()) <= 2e-2 * scale and rel <= 2e-2, (name, float((g.cpu() - r).abs().max()), scale, rel)


def test_mixed_precision_modes_end_to_end_gradients_agree_with_fp32():
    """set_training_precision("bf16") (un-fused GEMM twins) and its recompute combination (fused bf16 inference forward, everything
    rebuilt in backward) against the exact fp32 training renderer on the same rays / depths / noise: same gradient up to bf16 noise.
    Band-limited nets (synth band_limit): on the gain-2/3 nets the hierarchical depths themselves move with the coarse pass's bf16
    noise and the 2^14-frequency embedding turns that into a different fine-network input, so the fine gradients would be compared
    at different sample positions."""
    models, emb, args = _modules(gain=2.45, sigma_bias=-1.0, band_limit=4)
    R = 256
    rays, z, u, nc, nf = _inputs(R, 64, 64, seed=5)
    gw = torch.randn(R, 64, generator=torch.Generator().manual_seed(1)).to(DEV)

    def run():
        out = AG.fused_render_with_grad(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0)
        return (out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum()
    g32 = _grads(models, run)
    AG.set_training_precision("bf16")
    AG.set_training_recompute(True)
    try:
        g_rc = _grads(models, run)
    finally:
        AG.set_training_recompute(False)
    try:
        from crnerf_amd.models import rendering

        def run_unfused():
            out = rendering._render_unfused(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0, False, 1 << 20, train=True)
            return (out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum()
        g_mx = _grads(models, run_unfused)
    finally:
        AG.set_training_precision("f32")
    worst = {}
    for k in g32:
        n32 = float(g32[k].norm()) + 1e-20
        for tag, gq in (("mixed", g_mx), ("mixed+recompute", g_rc)):
            rel = float((gq[k] - g32[k]).norm()) / n32
            cos = float((gq[k] * g32[k]).sum() / (gq[k].norm() * g32[k].norm() + 1e-30))
            worst[tag] = max(worst.get(tag, 0.0), rel)
            # bf16 operand rounding (2^-9 per operand and product, relu masks flipping with it) through up to eleven layers: a noisy
            # but well-aligned gradient -- measured on this 49 k-point batch: rel-L2 0.13 / cosine 0.991 on xyz_encoding_1 (the end of
            # the chain), a few per cent on the late layers
            assert rel <= 0.25 and cos >= 0.97, (tag, k, rel, cos)
    print("worst rel-L2 vs fp32 gradient:", worst)


@pytest.mark.parametrize("recompute", [False, True])
def test_mixed_precision_training_reduces_the_loss(recompute):
    from crnerf_amd.models.rendering import render_rays_cross_ray
    models, emb, args = _modules(gain=2.45, sigma_bias=-1.0, band_limit=4)
    rays = C(synth.rays(256, seed=2, H=16, W=16))
    target = torch.rand(256, 64, generator=torch.Generator().manual_seed(4)).to(DEV)
    opt = torch.optim.Adam([p for m in models.values() for p in m.parameters()], lr=5e-4, fused=True)
    AG.set_training_precision("bf16")
    AG.set_training_recompute(recompute)
    losses = []
    try:
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            res = render_rays_cross_ray(models, emb, rays, None, 64, False, 0, 0, 64, 4096, False, args=args)
            loss = ((res["feature_fine"] - target) ** 2).mean() + ((res["feature_coarse"] - target) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        AG.set_training_precision("f32")
        AG.set_training_recompute(False)
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses


@pytest.mark.parametrize("mixed", [False, True])
def test_training_twins_cold_l2_are_deterministic(mixed):
    """The training twins under the treatment of tests/test_gpu_bf16.py::test_render_cold_l2_is_deterministic: forward-with-save and
    backward of 200,000 points, L2 thrashed before every call; outputs and all 24 gradients bit-identical across passes (the weight
    rings certify what they read, the partial-sum reductions have a fixed order)."""
    st = {k: C(v) for k, v in synth.mlp_state(9, 2.0, 0.5).items()}
    n = 200_000
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, 120, generator=g).to(DEV)
    d_out = torch.randn(n, 65, generator=g).to(DEV)
    junk_a, junk_b = torch.empty(1 << 28, device=DEV), torch.zeros(1 << 28, device=DEV)

    def run():
        junk_a.copy_(junk_b)
        if mixed:
            packed, tensors = ops.pack_mlp_weights_mixed(st)
            out, acts = ops.mlp_forward_train_mixed(packed, tensors, x)
            junk_a.copy_(junk_b)
            return [out] + ops.mlp_backward_mixed(packed, tensors, x, out, d_out, acts)
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(st), x)
        junk_a.copy_(junk_b)
        return [out] + ops.mlp_backward(ops.pack_mlp_weights_t(st), x, out, d_out, acts)
    with torch.no_grad():
        ref = run()
        for it in range(3):
            for i, (a, b) in enumerate(zip(run(), ref)):
                assert torch.equal(a, b), "pass %d: tensor %d differs, max |d| %.3e" % (it, i, float((a - b).abs().max()))


@torch.no_grad()
@pytest.mark.parametrize("R,N", [(1, 1), (37, 65), (512, 192)])
def test_embed_points_equals_the_torch_composition_bitwise(R, N):
    """crnerf_embed_points_f32 (rendering.py:100-114 in one pass) against the composition it replaces: points by torch broadcasting,
    crnerf_posenc_f32 on them, the direction embedding repeated, torch.cat -- same arithmetic, so the same bits."""
    rays = C(synth.rays(R, seed=R, H=1, W=R))
    z = torch.sort(torch.rand(R, N, generator=torch.Generator().manual_seed(N)) * 4 + 0.5, dim=1).values.to(DEV)
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=torch.Generator().manual_seed(1)), dim=1).to(DEV)
    for view_dir in (None, vd):
        pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)
        demb = ops.posenc((view_dir if view_dir is not None else rays[:, 3:6]).contiguous(), 4)
        want = torch.cat([ops.posenc(pts, 15), demb[:, None, :].expand(R, N, 27).reshape(R * N, 27)], 1)
        got = AG._embed_points(rays, z, view_dir)
        assert got.shape == (R * N, 120) and torch.equal(got, want)


@pytest.mark.parametrize("n", [4111, 50000, 300000])      # a 15-point ragged last chunk / batched launch (<= 2^18 points) / per-layer launches (whole k-step pairs only)
def test_wgrad_bf16x3_is_as_accurate_as_the_fp32_matrix_cores(n):
    """CRNERF_BWD_WGRAD_BF16X3 (opt-in): the 256 x 256 weight-gradient blocks from three-piece bf16 splits of the fp32 operands, six bf16
    MFMAs per product.  Held against (i) the exact fp32-MFMA path on the same deltas / activations -- the two may differ by fp32 summation
    noise only -- and (ii) a float64 evaluation of the same sums through torch autograd on the GPU (the oracle's MLP in float64 on the
    kernel's own inputs): the split path must not be further from it than the fp32 matrix cores are."""
    from oracle import cpu_ref as O
    st = synth.mlp_state(23, 1.0, 0.5)
    g = torch.Generator().manual_seed(n)
    x = torch.cat([O.posenc(torch.rand(n, 3, generator=g) * 4 - 2, 15), O.posenc(torch.rand(n, 3, generator=g) * 2 - 1, 4)], 1).to(DEV)
    d_out = torch.randn(n, 65, generator=g).to(DEV)
    dev = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev), x)
        pt = ops.pack_mlp_weights_t(dev)
        exact = ops.mlp_backward(pt, x, out, d_out, acts)
        split = ops.mlp_backward(pt, x, out, d_out, acts, wgrad_bf16="x3")
    w64 = {k: v.double().clone().requires_grad_(True) for k, v in dev.items()}
    (O.mlp_forward(w64, x.double()) * d_out.double()).sum().backward()
    worst = 0.0
    for name, ge, gs in zip(ops.MLP_TENSOR_NAMES, exact, split):
        ref = w64[name].grad
        scale = float(ref.abs().max())
        e_exact, e_split = float((ge.double() - ref).abs().max()) / scale, float((gs.double() - ref).abs().max()) / scale
        diff = float((gs - ge).abs().max()) / scale
        worst = max(worst, diff)
        print("%-28s vs float64: fp32 MFMA %.2e  bf16x3 %.2e   |bf16x3 - fp32 MFMA| %.2e" % (name, e_exact, e_split, diff))
        # weights: fp32 summation noise; biases: fp32 column sums in a different order
        assert diff <= 5e-6 and e_split <= 1.5 * e_exact + 2e-7, (name, e_exact, e_split, diff)
    assert worst > 0.0                                             # the split path did run
