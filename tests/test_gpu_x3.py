"""The "f32x3" entry points (include/crnerf.h): NeRF_sigma.forward (models/nerf.py:157-182) in fp32 on the bf16 matrix cores -- every fp32
operand of the eleven nn.Linear split into three bf16 pieces, a product = the six leading piece products, fp32 accumulation.  Held to the SAME
goldens and tolerances as the fp32 entry points (tests/test_gpu_parity.py), and against a float64 evaluation: the split path must sit where the
fp32 matrix cores sit."""
import numpy as np
import pytest
import torch

import crnerf_amd.synth as synth
from crnerf_amd import ops
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def C(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(got, want, atol, rtol=0.0):
    torch.testing.assert_close(got.detach().float().cpu(), torch.as_tensor(want).float(), atol=atol, rtol=rtol)


def test_mlp_x3_golden(golden):
    g = golden("g2_mlp")
    x = C(g["x"])
    for tag, atol, rtol in (("default", 1e-6, 0.0), ("peaky", 3e-5, 1e-5)):      # the fp32 entry point's bars (test_mlp_golden)
        pk = ops.pack_mlp_weights_x3({k: C(v) for k, v in synth.mlp_state(int(g["seed_" + tag]), float(g["gain_" + tag])).items()})
        close(ops.mlp_forward_x3(pk, x), g["out_" + tag], atol=atol, rtol=rtol)
        close(ops.mlp_forward_x3(pk, x[:, :93].contiguous(), sigma_only=True), g["sigma_" + tag], atol=atol, rtol=rtol)


@pytest.mark.parametrize("n", [1, 31, 33, 127, 129, 4099])
def test_mlp_x3_vs_oracle_ragged_sizes(n):
    st = synth.mlp_state(11, 2.0, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-3, 3, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    w = {k: torch.from_numpy(v) for k, v in st.items()}
    close(ops.mlp_forward_x3(ops.pack_mlp_weights_x3({k: C(v) for k, v in st.items()}), x.to(DEV)), O.mlp_forward(w, x), atol=2e-5, rtol=1e-5)


def test_mlp_x3_detects_transposed_or_permuted_packing():
    """One-hot input rows reproduce single columns of W1 / of the dir layer: catches any row / column / slot / piece-order error in fragX."""
    st = synth.mlp_state(3, 1.0)
    w = {k: torch.from_numpy(v) for k, v in st.items()}
    x = torch.zeros(120, 120)
    x[torch.arange(120), torch.arange(120)] = 1.0
    close(ops.mlp_forward_x3(ops.pack_mlp_weights_x3({k: C(v) for k, v in st.items()}), x.to(DEV)), O.mlp_forward(w, x), atol=1e-6)


@pytest.mark.parametrize("gain", [1.0, 2.0, 3.0])
def test_mlp_x3_is_as_accurate_as_the_fp32_matrix_cores(gain):
    """Against the oracle's MLP in float64 on the same fp32 inputs and weights: max / mean error of the x3 path vs the fp32-MFMA path."""
    n = 20000
    st = synth.mlp_state(29, gain, 0.5)
    g = torch.Generator().manual_seed(int(gain * 10))
    x = torch.cat([O.posenc(torch.rand(n, 3, generator=g) * 4 - 2, 15), O.posenc(torch.rand(n, 3, generator=g) * 2 - 1, 4)], 1).to(DEV)
    dev = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        ref = O.mlp_forward({k: v.double() for k, v in dev.items()}, x.double())
        o32 = ops.mlp_forward(ops.pack_mlp_weights(dev), x).double()
        ox3 = ops.mlp_forward_x3(ops.pack_mlp_weights_x3(dev), x).double()
    e32, ex3 = (o32 - ref).abs(), (ox3 - ref).abs()
    print("gain %.1f: fp32 MFMA max %.3e mean %.3e | x3 max %.3e mean %.3e | x3 vs fp32 MFMA max %.3e" %
          (gain, float(e32.max()), float(e32.mean()), float(ex3.max()), float(ex3.mean()), float((ox3 - o32).abs().max())))
    assert float(ex3.mean()) <= 1.5 * float(e32.mean()) + 1e-9 and float(ex3.max()) <= 2.0 * float(e32.max()) + 1e-7


# ------------------------------------------------------------------ the fused renderer on the x3 core (crnerf_render_rays_f32x3)
def _packx(st):
    return ops.pack_mlp_weights_x3({k: C(v) for k, v in st.items()})


@torch.no_grad()
@pytest.mark.parametrize("tag,ni,disp", [("c64", 0, False), ("c64_f128", 128, False), ("c64_f128_disp", 128, True), ("c64_f64", 64, False)])
def test_render_x3_golden(golden, tag, ni, disp):
    """tests/test_gpu_parity.py::test_render_golden, assertion for assertion, on the f32x3 renderer."""
    from test_gpu_parity import _models, assert_depths
    g = golden("g5_render")
    st_c, st_f = _models(g)
    out = ops.render_rays(_packx(st_c), _packx(st_f) if ni else None, C(g["rays"]), 64, ni, use_disp=disp, z_steps=C(g["z_steps_64"]),
                          u=C(g["u_steps_%d" % ni]) if ni else None, want_z_fine=True, precision="f32x3")
    close(out["weights_coarse"], g[tag + "__weights_coarse"], atol=3e-6)
    close(out["feature_coarse"], g[tag + "__feature_coarse"], atol=1e-5)
    close(out["depth_coarse"], g[tag + "__depth_coarse"], atol=1e-5)
    if not ni:
        return
    z_coarse = O.coarse_depths(torch.from_numpy(g["rays"]), 64, disp, torch.from_numpy(g["z_steps_64"]))
    assert_depths(out["z_fine"], g[tag + "__z_fine"], z_coarse, g[tag + "__weights_coarse"])
    rays = torch.from_numpy(g["rays"])
    zf = out["z_fine"].cpu()
    raw = O._run_model(O.to_torch(st_f), rays, zf, O.posenc(rays[:, 3:6], 4), 32768)
    w2, f2, d2 = O.composite(raw, zf)
    close(out["weights_fine"], w2, atol=3e-6)
    close(out["feature_fine"], f2, atol=1e-5)
    close(out["depth_fine"], d2, atol=2e-5)
    close(out["feature_fine"], g[tag + "__feature_fine"], atol=0.15)
    ref = torch.from_numpy(g[tag + "__feature_fine"])
    assert float((out["feature_fine"].cpu() - ref).norm() / ref.norm()) < 3e-2
    assert float((out["weights_fine"].sum(-1).cpu() - 1).abs().max()) < 1e-5


@torch.no_grad()
@pytest.mark.parametrize("net", ["band", "gain1"])
@pytest.mark.parametrize("tag,disp", [("c64_f128", False), ("c64_f128_disp", True)])
def test_x3_end_to_end_meets_the_stated_fp32_tolerance(golden, net, tag, disp):
    """tests/test_gpu_e2e_parity.py::test_fp32_end_to_end_meets_stated_tolerance on the f32x3 renderer: SURVEY 8d's fp32 bars, END TO END
    against the reference's own outputs, through the high-contrast decoder."""
    import test_gpu_e2e_parity as E
    g = golden("g14_render_smooth")
    key = "%s__%s__" % (net, tag)
    st_c, st_f = synth.mlp_state(41, **E.SMOOTH_NETS[net]), synth.mlp_state(42, **E.SMOOTH_NETS[net])
    out = ops.render_rays(_packx(st_c), _packx(st_f), C(g["rays"]), 64, 128, use_disp=disp, z_steps=C(g["z_steps_64"]), u=C(g["u_steps_128"]),
                          want_z_fine=True, precision="f32x3")
    H, W = int(g["H"]), int(g["W"])
    rgb = E._decoder(g)(out["feature_fine"].t().reshape(1, 64, H, W), C(g["style"])).reshape(3, H * W).t()
    m = E._metrics(out, g, key, rgb)
    E.record("f32x3 %s %s" % (net, tag), m)
    far = float(g["rays"][:, 7].max())
    assert m["rgb"]["max_abs"] <= 2e-5, m
    assert m["feature_fine"]["rel_l2"] <= 1e-5 and m["feature_coarse"]["rel_l2"] <= 1e-5, m
    assert m["z_fine"]["max_abs"] <= 1e-5 * far, m
    assert m["feature_fine"]["max_abs"] <= 1e-5 and m["weights_fine"]["max_abs"] <= 1e-5 and m["depth_fine"]["max_abs"] <= 2e-5, m
    assert m["weights_coarse"]["max_abs"] <= 3e-6 and m["depth_coarse"]["max_abs"] <= 1e-5, m


@torch.no_grad()
@pytest.mark.parametrize("R,Nc,Ni", [(37, 64, 64), (5, 33, 20), (16, 64, 0)])
def test_x3_train_forward_equals_x3_inference_and_saves_what_the_fp32_twin_saves(R, Nc, Ni):
    """crnerf_render_rays_train_f32x3: outputs bit-identical to the f32x3 inference renderer; saved activations / relu bits / raw rows in the
    fp32 training twins' layout, equal to what crnerf_render_rays_train_f32 saves up to the two paths' fp32-level difference."""
    st_c, st_f = {k: C(v) for k, v in synth.mlp_state(5, 1.0, 0.5).items()}, {k: C(v) for k, v in synth.mlp_state(6, 1.0, 0.5).items()}
    px = [ops.pack_mlp_weights_x3(st_c), ops.pack_mlp_weights_x3(st_f)]
    p32 = [ops.pack_mlp_weights(st_c), ops.pack_mlp_weights(st_f)]
    rng = np.random.default_rng(R)
    rays = C(synth.rays(R, seed=R))
    z = C(np.sort(rng.uniform(2, 6, (R, Nc)).astype(np.float32), -1))
    u = C(rng.uniform(0, 1, (R, max(Ni, 1))).astype(np.float32))
    kw = dict(z_coarse=z, u=u if Ni else None, noise_std=0.0)
    inf = ops.render_rays(px[0], px[1] if Ni else None, rays, Nc, Ni, want_z_fine=True, precision="f32x3", **kw)
    trn = ops.render_rays(px[0], px[1] if Ni else None, rays, Nc, Ni, train=True, precision="f32x3", **kw)
    for k in inf:
        assert torch.equal(inf[k], trn[k]), k
    ref = ops.render_rays(p32[0], p32[1] if Ni else None, rays, Nc, Ni, train=True, **kw)
    for tag, N in (("coarse", Nc),) + ((("fine", Nc + Ni),) if Ni else ()):
        P = R * N
        if tag == "fine" and not torch.equal(trn["z_fine"], ref["z_fine"]):
            continue                                    # the two paths sampled (slightly) different depths: rows are not comparable point by point
        n_act = 10 * P * 256
        a, b = trn["acts_" + tag][:4 * n_act].view(torch.float32).view(10, P, 256), ref["acts_" + tag][:4 * n_act].view(torch.float32).view(10, P, 256)
        assert float((a[:9] - b[:9]).abs().max()) <= 2e-5 * float(b[:9].abs().max()) and float((a[9, :, :128] - b[9, :, :128]).abs().max()) <= 2e-5, tag
        assert float((trn["raw_" + tag] - ref["raw_" + tag]).abs().max()) <= 2e-6, tag
        ma, mb = trn["acts_" + tag][4 * n_act:4 * n_act + 320 * P].view(10, P, 32), ref["acts_" + tag][4 * n_act:4 * n_act + 320 * P].view(10, P, 32)
        for s in (0, 1, 2, 3, 4, 5, 6, 7):
            assert float((ma[s] == mb[s]).float().mean()) >= 0.999, (tag, s)         # bits differ only where a pre-activation is within fp32 noise of 0
        assert float((ma[9].view(P, 4, 8)[:, :, :4] == mb[9].view(P, 4, 8)[:, :, :4]).float().mean()) >= 0.999, tag
        # and they are the record of the saved rows themselves
        k = torch.arange(64, device=DEV)
        for s in (0, 5):
            want = a[s].view(P, 16, 4, 4) > 0                                        # [T][g][r]
            bits = trn["acts_" + tag][4 * n_act:].view(torch.int64)[:10 * P * 4].view(10, P, 4)[s]
            got = ((bits[:, :, None] >> k[None, None, :]) & 1).bool().view(P, 4, 16, 4).permute(0, 2, 1, 3)
            assert torch.equal(want, got), (tag, s)


def test_x3_training_forward_gradients_match_the_fp32_forward():
    """set_training_forward_precision("f32x3"): FusedRenderFn on the x3 forward + the fp32 backward twins, against the all-fp32 path on the same
    rays / depths / noise: the same gradient up to the two forwards' fp32-level difference (band-limited nets: identical sampling decisions)."""
    from crnerf_amd import autograd as AG
    from test_gpu_train_fused import _grads, _inputs, _modules
    models, emb, args = _modules(gain=2.45, sigma_bias=-1.0, band_limit=4)
    R = 128
    rays, z, u, nc, nf = _inputs(R, 64, 64, seed=5)
    gw = torch.randn(R, 64, generator=torch.Generator().manual_seed(1)).to(DEV)

    def run():
        out = AG.fused_render_with_grad(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0)
        return (out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum() + 0.1 * (out["weights_fine"] ** 2).sum()
    AG.set_training_forward_precision("f32")
    g32 = _grads(models, run)
    AG.set_training_forward_precision("f32x3")
    try:
        gx3 = _grads(models, run)
    finally:
        AG.set_training_forward_precision(None)
    for k in g32:
        rel = float((gx3[k] - g32[k]).norm() / (g32[k].norm() + 1e-30))
        assert rel <= 2e-3, (k, rel)     # measured 6e-4 on xyz_encoding_1 (a handful of relu masks flip with the forwards' fp32-level difference)


@torch.no_grad()
def test_x3_render_cold_l2_is_deterministic():
    """Guard for the x3 core's weight ring (mlp_core_x3.h WeightPipeX: six slots, fragments read through a six-deep queue that runs ahead into the
    next stage) with the weight stream evicted from L2 before every launch: 131,072 rays in 32,768-ray chunks, a 1 GiB copy before each chunk, four
    passes, every output bit-identical to the first pass (tests/test_gpu_bf16.py::test_render_cold_l2_is_deterministic is the same guard for the
    other cores); the training twin rides along on the last chunk."""
    st_c = {k: C(v) for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
    st_f = {k: C(v) for k, v in synth.mlp_state(2, 2.0, 0.5).items()}
    pc, pf = ops.pack_mlp_weights_x3(st_c), ops.pack_mlp_weights_x3(st_f)
    R = 131072
    rays = C(synth.rays(R, seed=0, H=512, W=256))
    z_steps, u = torch.linspace(0, 1, 64, device=DEV), torch.linspace(0, 1, 128, device=DEV)
    junk_a, junk_b = torch.empty(1 << 28, device=DEV), torch.zeros(1 << 28, device=DEV)

    def run():
        outs = []
        for i in range(0, R, 32768):
            junk_a.copy_(junk_b)
            outs.append(ops.render_rays(pc, pf, rays[i:i + 32768], 64, 128, z_steps=z_steps, u=u, precision="f32x3"))
        junk_a.copy_(junk_b)
        trn = ops.render_rays(pc, pf, rays[R - 4096:], 64, 128, z_steps=z_steps, u=u, precision="f32x3", train=True)
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}, trn
    ref, trn0 = run()
    for k in ("feature_fine", "weights_fine", "feature_coarse"):
        assert torch.equal(trn0[k], ref[k][R - 4096:]), k
    for it in range(3):
        out, trn = run()
        for k in ref:
            assert torch.equal(out[k], ref[k]), (it, k)
        assert torch.equal(trn["raw_fine"], trn0["raw_fine"]) and torch.equal(trn["raw_coarse"], trn0["raw_coarse"]), it


@torch.no_grad()
@pytest.mark.parametrize("n", [1, 33, 4099, 70000])
def test_mlp_backward_x3_matches_the_fp32_data_gradient(n):
    """crnerf_mlp_backward_x3_f32: the data gradient on the x3 core writes the same deltas the fp32 kernel writes (seen through every weight /
    bias gradient, computed by the SAME exact weight-gradient kernels from them) -- equal up to fp32 summation noise; and against torch autograd
    through the oracle at the fp32 twin's own tolerance (tests/test_gpu_parity.py::test_mlp_backward_vs_autograd_oracle)."""
    st = synth.mlp_state(13, 2.0 if n > 100 else 1.0, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    d_out = torch.from_numpy(rng.normal(size=(n, 65)).astype(np.float32))
    dev = {k: C(v) for k, v in st.items()}
    out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev), x.to(DEV))
    g32 = ops.mlp_backward(ops.pack_mlp_weights_t(dev), x.to(DEV), out, d_out.to(DEV), acts)
    gx3 = ops.mlp_backward(ops.pack_mlp_weights_t_x3(dev), x.to(DEV), out, d_out.to(DEV), acts, dgrad_x3=True)
    for name, a, b in zip(ops.MLP_TENSOR_NAMES, gx3, g32):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 3e-5 * scale + 1e-7, (name, float((a - b).abs().max()), scale)   # measured <= 2.0e-5 (n = 70000, chunked fp32 sums)
    if n <= 100:
        with torch.enable_grad():
            w = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st.items()}
            (O.mlp_forward(w, x) * d_out).sum().backward()
        for name, gq in zip(ops.MLP_TENSOR_NAMES, gx3):
            ref = w[name].grad
            assert float((gq.cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-5, name


@torch.no_grad()
@pytest.mark.parametrize("log2c", [40, -40, 100])
def test_mlp_x3_keeps_fp32_accuracy_across_the_exponent_range(log2c):
    """bf16 has fp32's exponent range, so the three-piece split is scale-free: xyz_encoding_1 scaled by c = 2^log2c and xyz_encoding_2's weights by
    1/c (relu is positively homogeneous: the same function) must give the outputs of the unscaled net to fp32 accuracy -- activations of 1e12 or
    1e-12 (1e30 for 2^100) in between are split as exactly as values near 1."""
    st = synth.mlp_state(31, 1.0, 0.5)
    c = float(2.0 ** log2c)
    st2 = dict(st)
    st2["xyz_encoding_1.0.weight"] = (st["xyz_encoding_1.0.weight"] * c).astype(np.float32)
    st2["xyz_encoding_1.0.bias"] = (st["xyz_encoding_1.0.bias"] * c).astype(np.float32)
    st2["xyz_encoding_2.0.weight"] = (st["xyz_encoding_2.0.weight"] / c).astype(np.float32)
    rng = np.random.default_rng(3)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (2000, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (2000, 3)).astype(np.float32)), 4)], 1).to(DEV)
    a = ops.mlp_forward_x3(ops.pack_mlp_weights_x3({k: C(v) for k, v in st.items()}), x)
    b = ops.mlp_forward_x3(ops.pack_mlp_weights_x3({k: C(v) for k, v in st2.items()}), x)
    assert torch.isfinite(b).all() and float((a - b).abs().max()) <= 1e-6, float((a - b).abs().max())
