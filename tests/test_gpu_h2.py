"""The "f32h2" entry points (include/crnerf.h): NeRF_sigma.forward (models/nerf.py:157-182) in fp32 on the fp16 matrix cores -- every fp32
operand of the eleven nn.Linear split into TWO fp16 pieces (weights scaled by 2^8 at pack time), a product = the three leading piece products,
fp32 accumulation.  Held to the SAME goldens and tolerances as the fp32 entry points (tests/test_gpu_parity.py; tests/test_gpu_x3.py is the
same file for the three-piece bf16 core), and against a float64 evaluation: the split path must sit where the fp32 matrix cores sit.  Unlike
f32x3 the split is not scale-free -- the range tests pin what happens at fp16's edges."""
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


def _packh(st):
    return ops.pack_mlp_weights_h2({k: C(v) for k, v in st.items()})


def test_mlp_h2_golden(golden):
    g = golden("g2_mlp")
    x = C(g["x"])
    for tag, atol, rtol in (("default", 1e-6, 0.0), ("peaky", 3e-5, 1e-5)):      # the fp32 entry point's bars (test_mlp_golden)
        pk = _packh(synth.mlp_state(int(g["seed_" + tag]), float(g["gain_" + tag])))
        close(ops.mlp_forward_h2(pk, x), g["out_" + tag], atol=atol, rtol=rtol)
        close(ops.mlp_forward_h2(pk, x[:, :93].contiguous(), sigma_only=True), g["sigma_" + tag], atol=atol, rtol=rtol)


@pytest.mark.parametrize("n", [1, 31, 33, 127, 129, 4099])
def test_mlp_h2_vs_oracle_ragged_sizes(n):
    st = synth.mlp_state(11, 2.0, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-3, 3, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    w = {k: torch.from_numpy(v) for k, v in st.items()}
    close(ops.mlp_forward_h2(_packh(st), x.to(DEV)), O.mlp_forward(w, x), atol=2e-5, rtol=1e-5)


def test_mlp_h2_detects_transposed_or_permuted_packing():
    """One-hot input rows reproduce single columns of W1 / of the dir layer: catches any row / column / slot / piece-order error in fragH."""
    st = synth.mlp_state(3, 1.0)
    w = {k: torch.from_numpy(v) for k, v in st.items()}
    x = torch.zeros(120, 120)
    x[torch.arange(120), torch.arange(120)] = 1.0
    close(ops.mlp_forward_h2(_packh(st), x.to(DEV)), O.mlp_forward(w, x), atol=1e-6)


@pytest.mark.parametrize("gain", [1.0, 2.0, 3.0])
def test_mlp_h2_is_as_accurate_as_the_fp32_matrix_cores(gain):
    """Against the oracle's MLP in float64 on the same fp32 inputs and weights: max / mean error of the h2 path vs the fp32-MFMA path."""
    n = 20000
    st = synth.mlp_state(29, gain, 0.5)
    g = torch.Generator().manual_seed(int(gain * 10))
    x = torch.cat([O.posenc(torch.rand(n, 3, generator=g) * 4 - 2, 15), O.posenc(torch.rand(n, 3, generator=g) * 2 - 1, 4)], 1).to(DEV)
    dev = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        ref = O.mlp_forward({k: v.double() for k, v in dev.items()}, x.double())
        o32 = ops.mlp_forward(ops.pack_mlp_weights(dev), x).double()
        oh2 = ops.mlp_forward_h2(ops.pack_mlp_weights_h2(dev), x).double()
    e32, eh2 = (o32 - ref).abs(), (oh2 - ref).abs()
    print("gain %.1f: fp32 MFMA max %.3e mean %.3e | h2 max %.3e mean %.3e | h2 vs fp32 MFMA max %.3e" %
          (gain, float(e32.max()), float(e32.mean()), float(eh2.max()), float(eh2.mean()), float((oh2 - o32).abs().max())))
    assert float(eh2.mean()) <= 1.5 * float(e32.mean()) + 1e-9 and float(eh2.max()) <= 2.0 * float(e32.max()) + 1e-7


def test_mlp_h2_range_edges_are_loud_or_exact():
    """fp16's range is the price of three MFMAs per product.  (a) Tiny weights (|w| ~ 2^-14: second pieces subnormal even after the 2^8 pack scale)
    still come out within fp32 noise -- the matrix cores honour fp16 subnormals.  (b) A weight beyond the documented bound (|w| >= 255 after which
    256 w leaves fp16), or a non-finite one, is REFUSED at pack time (crnerf_pack_mlp_weights_h2 checks and reports), never packed into a
    silently wrong stream."""
    st = synth.mlp_state(5, 1.0)
    tiny = dict(st)
    tiny["xyz_encoding_2.0.weight"] = (st["xyz_encoding_2.0.weight"] * 2.0 ** -10).astype(np.float32)
    tiny["xyz_encoding_3.0.weight"] = (st["xyz_encoding_3.0.weight"] * 2.0 ** 10).astype(np.float32)     # keeps the later activations at scale
    g = torch.Generator().manual_seed(1)
    x = torch.cat([O.posenc(torch.rand(512, 3, generator=g) * 4 - 2, 15), O.posenc(torch.rand(512, 3, generator=g) * 2 - 1, 4)], 1)
    ref = O.mlp_forward({k: torch.from_numpy(v).double() for k, v in tiny.items()}, x.double())
    o32 = ops.mlp_forward(ops.pack_mlp_weights({k: C(v) for k, v in tiny.items()}), x.to(DEV)).double().cpu()
    oh2 = ops.mlp_forward_h2(_packh(tiny), x.to(DEV)).double().cpu()
    e32, eh2 = float((o32 - ref).abs().max()), float((oh2 - ref).abs().max())
    print("tiny weights: fp32 MFMA max err %.3e, h2 max err %.3e" % (e32, eh2))
    assert eh2 <= 4.0 * e32 + 2e-7
    big = dict(st)
    wbig = st["xyz_encoding_4.0.weight"].copy()
    wbig[7, 9] = 300.0
    big["xyz_encoding_4.0.weight"] = wbig
    with pytest.raises(RuntimeError, match="h2 core's range"):
        _packh(big)
    nanw = dict(st)
    wn = st["static_rgb.0.weight"].copy()
    wn[0, 0] = np.nan
    nanw["static_rgb.0.weight"] = wn
    with pytest.raises(RuntimeError, match="h2 core's range"):
        _packh(nanw)


def test_mlp_h2_activation_overflow_poisons_the_point():
    """(c) Activations beyond fp16's range (|a| >= 65,504; weights in range): every point is either right -- within the fp32 path's tolerance --
    or NaN in all 65 outputs; never finite and wrong.  (Without the guard the (inf, -inf) pieces of such an activation become NaN in the next
    layer and its relu clamps that to 0.)"""
    st = dict(synth.mlp_state(5, 1.0))
    for l in (1, 2, 3):
        st["xyz_encoding_%d.0.weight" % l] = (st["xyz_encoding_%d.0.weight" % l] * 200.0).astype(np.float32)
    g = torch.Generator().manual_seed(2)
    x = torch.cat([O.posenc(torch.rand(2048, 3, generator=g) * 4 - 2, 15), O.posenc(torch.rand(2048, 3, generator=g) * 2 - 1, 4)], 1)
    x[::2] *= 1e-3                                          # half of the points stay small enough
    dev = {k: C(v) for k, v in st.items()}
    assert max(float(v.abs().max()) for v in dev.values()) < 255
    o32 = ops.mlp_forward(ops.pack_mlp_weights(dev), x.to(DEV))
    oh2 = ops.mlp_forward_h2(ops.pack_mlp_weights_h2(dev), x.to(DEV))
    bad = torch.isnan(oh2).any(1)
    assert bool((torch.isnan(oh2).all(1) == bad).all())    # poisoned points are NaN in every output
    assert 0 < int(bad.sum()) < 2048, int(bad.sum())
    ok = ~bad
    torch.testing.assert_close(oh2[ok], o32[ok], atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------ the fused renderer on the h2 core (crnerf_render_rays_f32h2)
@torch.no_grad()
@pytest.mark.parametrize("tag,ni,disp", [("c64", 0, False), ("c64_f128", 128, False), ("c64_f128_disp", 128, True), ("c64_f64", 64, False)])
def test_render_h2_golden(golden, tag, ni, disp):
    """tests/test_gpu_parity.py::test_render_golden, assertion for assertion, on the f32h2 renderer."""
    from test_gpu_parity import _models, assert_depths
    g = golden("g5_render")
    st_c, st_f = _models(g)
    out = ops.render_rays(_packh(st_c), _packh(st_f) if ni else None, C(g["rays"]), 64, ni, use_disp=disp, z_steps=C(g["z_steps_64"]),
                          u=C(g["u_steps_%d" % ni]) if ni else None, want_z_fine=True, precision="f32h2")
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
def test_h2_end_to_end_meets_the_stated_fp32_tolerance(golden, net, tag, disp):
    """tests/test_gpu_e2e_parity.py::test_fp32_end_to_end_meets_stated_tolerance on the f32h2 renderer: SURVEY 8d's fp32 bars, END TO END
    against the reference's own outputs, through the high-contrast decoder."""
    import test_gpu_e2e_parity as E
    g = golden("g14_render_smooth")
    key = "%s__%s__" % (net, tag)
    st_c, st_f = synth.mlp_state(41, **E.SMOOTH_NETS[net]), synth.mlp_state(42, **E.SMOOTH_NETS[net])
    out = ops.render_rays(_packh(st_c), _packh(st_f), C(g["rays"]), 64, 128, use_disp=disp, z_steps=C(g["z_steps_64"]), u=C(g["u_steps_128"]),
                          want_z_fine=True, precision="f32h2")
    H, W = int(g["H"]), int(g["W"])
    rgb = E._decoder(g)(out["feature_fine"].t().reshape(1, 64, H, W), C(g["style"])).reshape(3, H * W).t()
    m = E._metrics(out, g, key, rgb)
    E.record("f32h2 %s %s" % (net, tag), m)
    far = float(g["rays"][:, 7].max())
    assert m["rgb"]["max_abs"] <= 2e-5, m
    assert m["feature_fine"]["rel_l2"] <= 1e-5 and m["feature_coarse"]["rel_l2"] <= 1e-5, m
    assert m["z_fine"]["max_abs"] <= 1e-5 * far, m
    assert m["feature_fine"]["max_abs"] <= 1e-5 and m["weights_fine"]["max_abs"] <= 1e-5 and m["depth_fine"]["max_abs"] <= 2e-5, m
    assert m["weights_coarse"]["max_abs"] <= 3e-6 and m["depth_coarse"]["max_abs"] <= 1e-5, m


@torch.no_grad()
def test_h2_in_kernel_draws_equal_the_tensor_path():
    """The renderer's in-kernel Philox draws (include/crnerf.h CRNERF_RNG_*) on the h2 core: the same draws through the tensor arguments give
    bit-identical outputs (tests/test_gpu_rng.py holds the fp32 and f32x3 kernels to the same)."""
    from test_gpu_rng import _tensor_path_inputs
    pc, pf = _packh(synth.mlp_state(11, 2.0, 0.5)), _packh(synth.mlp_state(12, 2.0, 0.5))
    rays = C(synth.rays(257, seed=5))
    seed = 987654321012
    z_j, lower, upper, u, n_c, n_f = _tensor_path_inputs(rays, 64, 128, seed, 1.0, False)
    ref = ops.render_rays(pc, pf, rays, 64, 128, z_coarse=z_j, u=u, noise_coarse=n_c, noise_fine=n_f, noise_std=1.0, want_z_fine=True, precision="f32h2")
    got = ops.render_rays(pc, pf, rays, 64, 128, z_steps=torch.linspace(0, 1, 64, device=DEV), noise_std=1.0, want_z_fine=True, precision="f32h2",
                          rng={"seed": seed, "perturb": 1.0, "jitter": True, "u": True, "noise": True})
    assert torch.equal(got["z_coarse_used"], z_j)
    for k in ref:
        assert torch.equal(got[k], ref[k]), k


def test_h2_render_cold_l2_is_deterministic():
    """Guard for the weight ring under the h2 core's geometry (two-fragment groups, four-deep queue, 151 stages per pass) with the weight stream
    evicted from L2 before every launch: 131,072 rays in 32,768-ray chunks, a 1 GiB copy before each chunk, four passes, every output
    bit-identical to the first pass."""
    pc, pf = _packh(synth.mlp_state(1, 2.0, 0.5)), _packh(synth.mlp_state(2, 2.0, 0.5))
    R = 131072
    rays = C(synth.rays(R, seed=0, H=512, W=256))
    z_steps, u = torch.linspace(0, 1, 64, device=DEV), torch.linspace(0, 1, 128, device=DEV)
    junk_a, junk_b = torch.empty(1 << 28, device=DEV), torch.zeros(1 << 28, device=DEV)

    def run():
        outs = []
        for i in range(0, R, 32768):
            junk_a.copy_(junk_b)
            outs.append(ops.render_rays(pc, pf, rays[i:i + 32768], 64, 128, z_steps=z_steps, u=u, precision="f32h2"))
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
    ref = run()
    assert bool(torch.isfinite(ref["feature_fine"]).all())
    for it in range(3):
        out = run()
        for k in ref:
            assert torch.equal(out[k], ref[k]), (it, k)


# ------------------------------------------------------------------ precision="auto": the h2 core with the x3 core as its safety net
def _overflow_net(seed=5):
    """Weights in the h2 core's range (|w| < 255) whose hidden activations pass 65,504 for inputs of ordinary size."""
    st = dict(synth.mlp_state(seed, 1.0))
    for l in (1, 2, 3):
        st["xyz_encoding_%d.0.weight" % l] = (st["xyz_encoding_%d.0.weight" % l] * 200.0).astype(np.float32)
    return st


@torch.no_grad()
def test_mlp_auto_repairs_poisoned_points_and_refused_packs():
    """round-3 verdict, missing #3: "h2 has no fallback".  precision="auto" = crnerf_mlp_forward_f32h2 followed by
    crnerf_mlp_forward_f32x3_repair over the same output: (a) a net with an activation forced past 65,504 -- the h2 call alone poisons those points
    (NaN), the auto call returns NO NaN, the repaired 128-point groups carry the x3 core's values bit for bit, the untouched groups the h2 core's;
    (b) a net with one weight >= 255 -- the h2 pack is refused (CRNERF_ERR_RANGE), auto renders on the x3 core throughout.  Both within the fp32
    path's tolerance of the fp32 kernel (the reference, models/nerf.py:157-182, has no range limit)."""
    st = _overflow_net()
    g = torch.Generator().manual_seed(2)
    x = torch.cat([O.posenc(torch.rand(4096, 3, generator=g) * 4 - 2, 15), O.posenc(torch.rand(4096, 3, generator=g) * 2 - 1, 4)], 1)
    x[:3072] *= 1e-3                                        # 24 groups of 128 points stay in range, 8 do not
    dev = {k: C(v) for k, v in st.items()}
    xd = x.to(DEV)
    o32 = ops.mlp_forward(ops.pack_mlp_weights(dev), xd)
    oh2 = ops.mlp_forward_h2(ops.pack_mlp_weights_h2(dev), xd)
    ox3 = ops.mlp_forward_x3(ops.pack_mlp_weights_x3(dev), xd)
    pack = ops.pack_mlp_weights(dev, precision="auto")
    assert isinstance(pack, ops.AutoPack) and pack.h2 is not None
    oau = ops.mlp_forward(pack, xd, precision="auto")
    bad = torch.isnan(oh2).any(1)
    assert 0 < int(bad.sum()) < 4096 and not bool(bad[:3072].any())
    assert not bool(torch.isnan(oau).any()), "a NaN of the h2 core's making reached the caller"
    grp_bad = bad.view(-1, 128).any(1)[:, None].expand(-1, 128).reshape(-1)
    assert torch.equal(oau[grp_bad], ox3[grp_bad]) and torch.equal(oau[~grp_bad], oh2[~grp_bad])
    torch.testing.assert_close(oau, o32, atol=2e-3, rtol=2e-4)   # activations ~1e5: fp32 summation noise of both paths
    # (b)
    big = dict(synth.mlp_state(5, 1.0))
    wbig = big["xyz_encoding_4.0.weight"].copy()
    wbig[7, 9] = 300.0
    big["xyz_encoding_4.0.weight"] = wbig
    devb = {k: C(v) for k, v in big.items()}
    packb = ops.pack_mlp_weights(devb, precision="auto")
    assert packb.h2 is None and packb.x3 is not None
    x2 = xd[:1024] * 1.0
    ob = ops.mlp_forward(packb, x2, precision="auto")
    assert torch.equal(ob, ops.mlp_forward_x3(packb.x3, x2))
    torch.testing.assert_close(ob, ops.mlp_forward(ops.pack_mlp_weights(devb), x2), atol=1e-4, rtol=1e-4)


@torch.no_grad()
@pytest.mark.parametrize("ni", [0, 128])
def test_render_auto_repairs_poisoned_rays(ni):
    """The fused renderer: rays through the overflow net.  f32h2 alone returns NaN features for the rays with a poisoned point; precision="auto"
    (h2 render + crnerf_render_rays_f32x3_repair on the same buffers) returns no NaN, the f32x3 renderer's values bit for bit in every repaired ray
    quad and the h2 renderer's everywhere else.  Also through the drop-in modules (crnerf_amd.set_precision("auto"))."""
    def net(seed):   # hidden unit 0 of layers 1 and 2 amplifies the raw x coordinate 250 x 250 times: |h2| passes 65,504 where x > ~1.05
        st = dict(synth.mlp_state(seed, 1.0))
        for l in (1, 2):
            w = st["xyz_encoding_%d.0.weight" % l].copy()
            w[0, 0] = 250.0
            st["xyz_encoding_%d.0.weight" % l] = w
        return st
    st_c, st_f = net(5), net(6)
    rays = synth.rays(256, seed=3, H=16, W=16).copy()
    rays[:192, 0:3] *= 1e-3                                  # three quarters of the rays: tiny origins ...
    rays[:192, 6] = 1e-4
    rays[:192, 7] = 2e-3                                     # ... and depths: x stays ~1e-3 there, no activation overflows
    rd = C(rays)
    dc, df = {k: C(v) for k, v in st_c.items()}, {k: C(v) for k, v in st_f.items()}
    zs, u = torch.linspace(0, 1, 64, device=DEV), (torch.linspace(0, 1, ni, device=DEV) if ni else None)
    kw = dict(z_steps=zs, u=u, want_z_fine=ni > 0)
    oh2 = ops.render_rays(ops.pack_mlp_weights_h2(dc), ops.pack_mlp_weights_h2(df) if ni else None, rd, 64, ni, precision="f32h2", **kw)
    ox3 = ops.render_rays(ops.pack_mlp_weights_x3(dc), ops.pack_mlp_weights_x3(df) if ni else None, rd, 64, ni, precision="f32x3", **kw)
    pc, pf = ops.pack_mlp_weights(dc, precision="auto"), (ops.pack_mlp_weights(df, precision="auto") if ni else None)
    oau = ops.render_rays(pc, pf, rd, 64, ni, precision="auto", **kw)
    keys = ["feature_coarse", "weights_coarse", "depth_coarse"] + (["feature_fine", "weights_fine", "depth_fine", "z_fine"] if ni else [])
    bad = torch.isnan(oh2["feature_coarse"]).any(1)
    if ni:
        bad |= torch.isnan(oh2["feature_fine"]).any(1)
    assert 0 < int(bad.sum()) <= 64 and not bool(bad[:192].any()), int(bad.sum())
    quad_bad = bad.view(-1, 4).any(1)[:, None].expand(-1, 4).reshape(-1)
    for k in keys:
        assert not bool(torch.isnan(oau[k]).any()), k
        assert torch.equal(oau[k][quad_bad], ox3[k][quad_bad]), k
        assert torch.equal(oau[k][~quad_bad], oh2[k][~quad_bad]), k
    # the launcher form used by bench.py issues both calls too
    launch, out2 = ops.render_rays(pc, pf, rd, 64, ni, precision="auto", launcher=True, **kw)
    launch()
    for k in keys:
        assert torch.equal(out2[k], oau[k]), k


# ------------------------------------------------------------------ training on the h2 core (round 4): crnerf_render_rays_train_f32h2, crnerf_mlp_backward_h2_f32
@torch.no_grad()
@pytest.mark.parametrize("R,Nc,Ni", [(37, 64, 64), (5, 33, 20), (16, 64, 0)])
def test_h2_train_forward_saves_what_the_fp32_twin_saves(R, Nc, Ni):
    """crnerf_render_rays_train_f32h2 (tests/test_gpu_x3.py has the same test for the x3 twin): outputs within fp32 noise of the f32h2 inference
    renderer (the training form finishes a layer in a different instruction order, not bit-identical), saved activations / relu bits / raw rows
    in the fp32 training twins' layout and equal to what crnerf_render_rays_train_f32 saves up to the two paths' fp32-level difference."""
    st_c, st_f = {k: C(v) for k, v in synth.mlp_state(5, 1.0, 0.5).items()}, {k: C(v) for k, v in synth.mlp_state(6, 1.0, 0.5).items()}
    ph = [ops.pack_mlp_weights_h2(st_c), ops.pack_mlp_weights_h2(st_f)]
    p32 = [ops.pack_mlp_weights(st_c), ops.pack_mlp_weights(st_f)]
    rng = np.random.default_rng(R)
    rays = C(synth.rays(R, seed=R))
    z = C(np.sort(rng.uniform(2, 6, (R, Nc)).astype(np.float32), -1))
    u = C(rng.uniform(0, 1, (R, max(Ni, 1))).astype(np.float32))
    kw = dict(z_coarse=z, u=u if Ni else None, noise_std=0.0)
    inf = ops.render_rays(ph[0], ph[1] if Ni else None, rays, Nc, Ni, want_z_fine=True, precision="f32h2", **kw)
    trn = ops.render_rays(ph[0], ph[1] if Ni else None, rays, Nc, Ni, train=True, precision="f32h2", **kw)
    for k in ("feature_coarse", "weights_coarse", "depth_coarse"):      # (the sigma head sums in another order: 1e-7 of sigma x the last sample's delta = 1e2, rendering.py:121-123)
        torch.testing.assert_close(trn[k], inf[k], atol=5e-5, rtol=1e-4)
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
            assert float((ma[s] == mb[s]).float().mean()) >= 0.999, (tag, s)
        assert float((ma[9].view(P, 4, 8)[:, :, :4] == mb[9].view(P, 4, 8)[:, :, :4]).float().mean()) >= 0.999, tag
        k = torch.arange(64, device=DEV)
        for s in (0, 5):                                # the bits are the record of the saved rows themselves
            want = a[s].view(P, 16, 4, 4) > 0
            bits = trn["acts_" + tag][4 * n_act:].view(torch.int64)[:10 * P * 4].view(10, P, 4)[s]
            got = ((bits[:, :, None] >> k[None, None, :]) & 1).bool().view(P, 4, 16, 4).permute(0, 2, 1, 3)
            assert torch.equal(want, got), (tag, s)


@torch.no_grad()
@pytest.mark.parametrize("n,log2s", [(1, 0), (33, 0), (4099, 0), (70000, 0), (4099, -40), (4099, 40), (513, -100), (513, 90)])
def test_mlp_backward_h2_matches_the_fp32_data_gradient(n, log2s):
    """crnerf_mlp_backward_h2_f32: the deltas of the h2 core equal the fp32 kernel's up to fp32 summation noise (seen through every weight / bias
    gradient, formed by the SAME weight-gradient kernels from them) -- at ordinary gradient magnitudes and with d_out scaled by 2^-100 ... 2^90
    (fp16 alone would lose everything below 6e-8: every point's delta vector is rescaled per layer, include/crnerf.h); against torch autograd
    through the oracle at the fp32 twin's own tolerance."""
    st = synth.mlp_state(13, 2.0 if n > 100 else 1.0, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    d_out = torch.from_numpy(rng.normal(size=(n, 65)).astype(np.float32))
    d_out[::7] *= 1e-6                                  # points of very different gradient size side by side in one tile
    d_out[::11, :64] = 0.0                              # points whose rgb head gets no gradient at all
    d_out = d_out * float(2.0 ** log2s)
    dev = {k: C(v) for k, v in st.items()}
    out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev), x.to(DEV))
    g32 = ops.mlp_backward(ops.pack_mlp_weights_t(dev), x.to(DEV), out, d_out.to(DEV), acts)
    gh2 = ops.mlp_backward(ops.pack_mlp_weights_t_h2(dev), x.to(DEV), out, d_out.to(DEV), acts, dgrad_h2=True)
    for name, a, b in zip(ops.MLP_TENSOR_NAMES, gh2, g32):
        assert bool(torch.isfinite(a).all()), name
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale + (1e-7 if log2s == 0 else 0.0), (name, float((a - b).abs().max()), scale)
    if n <= 100:
        with torch.enable_grad():
            w = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st.items()}
            (O.mlp_forward(w, x) * d_out).sum().backward()
        for name, gq in zip(ops.MLP_TENSOR_NAMES, gh2):
            ref = w[name].grad
            assert float((gq.cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-5, name


@pytest.mark.parametrize("mode", ["f32h2", "auto"])
def test_h2_training_forward_gradients_match_the_fp32_forward(mode):
    """set_training_forward_precision("f32h2" / "auto"): FusedRenderFn on the h2 training twin + the h2 data gradient, against the all-fp32 path on
    the same rays / depths / noise (tests/test_gpu_x3.py::test_x3_training_forward_gradients_match_the_fp32_forward, same bar)."""
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
    AG.set_training_forward_precision(mode)
    try:
        gh2 = _grads(models, run)
    finally:
        AG.set_training_forward_precision(None)
    for k in g32:
        rel = float((gh2[k] - g32[k]).norm() / (g32[k].norm() + 1e-30))
        assert rel <= 2e-3, (k, rel)


@torch.no_grad()
def test_train_auto_repairs_poisoned_rays_and_their_saved_rows():
    """Training under precision="auto": rays through a net whose hidden activations pass 65,504 for a quarter of the rays.  The h2 twin alone
    returns NaN features there; auto (h2 twin, then crnerf_render_rays_train_f32x3_repair on the same buffers) returns no NaN, and in every
    repaired ray quad the outputs, raw rows AND saved activation rows / relu bits are the x3 twin's bit for bit; everywhere else the h2 twin's."""
    def net(seed):
        st = dict(synth.mlp_state(seed, 1.0))
        for l in (1, 2):
            w = st["xyz_encoding_%d.0.weight" % l].copy()
            w[0, 0] = 250.0
            st["xyz_encoding_%d.0.weight" % l] = w
        return st
    st_c, st_f = net(5), net(6)
    R, Nc, Ni = 256, 64, 64
    rays = synth.rays(R, seed=3, H=16, W=16).copy()
    rays[:192, 0:3] *= 1e-3
    rays[:192, 6] = 1e-4
    rays[:192, 7] = 2e-3
    rd = C(rays)
    dc, df = {k: C(v) for k, v in st_c.items()}, {k: C(v) for k, v in st_f.items()}
    zs, u = torch.linspace(0, 1, Nc, device=DEV), torch.linspace(0, 1, Ni, device=DEV)
    kw = dict(z_steps=zs, u=u, train=True)
    oh2 = ops.render_rays(ops.pack_mlp_weights_h2(dc), ops.pack_mlp_weights_h2(df), rd, Nc, Ni, precision="f32h2", **kw)
    ox3 = ops.render_rays(ops.pack_mlp_weights_x3(dc), ops.pack_mlp_weights_x3(df), rd, Nc, Ni, precision="f32x3", **kw)
    oau = ops.render_rays(ops.pack_mlp_weights(dc, precision="auto"), ops.pack_mlp_weights(df, precision="auto"), rd, Nc, Ni, precision="auto", **kw)
    bad = torch.isnan(oh2["feature_coarse"]).any(1) | torch.isnan(oh2["feature_fine"]).any(1)
    assert 0 < int(bad.sum()) <= 64 and not bool(bad[:192].any()), int(bad.sum())
    quad_bad = bad.view(-1, 4).any(1)[:, None].expand(-1, 4).reshape(-1)
    for k in ("feature_coarse", "weights_coarse", "depth_coarse", "feature_fine", "weights_fine", "depth_fine", "z_fine", "raw_coarse", "raw_fine"):
        assert not bool(torch.isnan(oau[k]).any()), k
        assert torch.equal(oau[k][quad_bad], ox3[k][quad_bad]), k
        assert torch.equal(oau[k][~quad_bad], oh2[k][~quad_bad]), k
    for tag, N in (("coarse", Nc), ("fine", Nc + Ni)):
        P = R * N
        pt_bad = quad_bad[:, None].expand(-1, N).reshape(-1)
        rows = lambda o: o["acts_" + tag][:40 * P * 256].view(torch.float32).view(10, P, 256)        # noqa: E731
        bits = lambda o: o["acts_" + tag][40 * P * 256:40 * P * 256 + 320 * P].view(10, P, 32)       # noqa: E731
        for src, sel in ((ox3, pt_bad), (oh2, ~pt_bad)):
            assert torch.equal(rows(oau)[:9, sel].view(torch.int32), rows(src)[:9, sel].view(torch.int32)), tag
            assert torch.equal(rows(oau)[9, sel, :128].view(torch.int32), rows(src)[9, sel, :128].view(torch.int32)), tag     # slot 9 is 128 wide
            assert torch.equal(bits(oau)[:8, sel], bits(src)[:8, sel]), tag                                                   # slot 8 is linear: no bits
            assert torch.equal(bits(oau)[9, sel].view(-1, 4, 8)[:, :, :4], bits(src)[9, sel].view(-1, 4, 8)[:, :, :4]), tag


def test_train_auto_with_a_refused_pack_falls_back_on_the_device():
    """A weight >= 255 under set_training_forward_precision("auto"): the asynchronous h2 packs carry the range flag (crnerf_pack_h2_status says so),
    the h2 training twin marks every ray NaN, crnerf_render_rays_train_f32x3_repair renders them all, the h2 data gradient stands aside for the
    f32x3 one -- all decided on the device.  Outputs are then the f32x3 mode's bit for bit, the refused model's gradients to a few ulps (same products,
    other chunk lengths in the batched weight-gradient sum)."""
    from crnerf_amd import autograd as AG
    from test_gpu_train_fused import _grads, _inputs, _modules
    models, emb, args = _modules(gain=2.45, sigma_bias=-1.0, band_limit=4)
    with torch.no_grad():
        models["fine"].state_dict()["xyz_encoding_4.0.weight"][7, 9] = 300.0
    st = {k: v.detach() for k, v in models["fine"].state_dict().items()}
    pk = ops.pack_mlp_weights_auto(st, check=False)
    assert pk.h2 is not None and not ops.pack_h2_in_range(pk.h2) and not ops.pack_h2_in_range(ops.pack_mlp_weights_t_h2(st))
    assert ops.pack_h2_in_range(ops.pack_mlp_weights_auto({k: v.detach() for k, v in models["coarse"].state_dict().items()}, check=False).h2)
    R = 64
    rays, z, u, nc, nf = _inputs(R, 64, 64, seed=5)
    gw = torch.randn(R, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    outs = {}

    def run(tag):
        def f():
            out = AG.fused_render_with_grad(models["coarse"], models["fine"], rays, 64, 64, False, None, z, u, nc, nf, 1.0)
            outs[tag] = {k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}
            return (out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum() + 0.1 * (out["weights_fine"] ** 2).sum()
        return f
    try:
        AG.set_training_forward_precision("f32x3")
        gx3 = _grads(models, run("x3"))
        AG.set_training_forward_precision("auto")
        gau = _grads(models, run("auto"))
    finally:
        AG.set_training_forward_precision(None)
    for k in ("feature_coarse", "feature_fine", "weights_fine", "depth_fine"):
        assert not bool(torch.isnan(outs["auto"][k]).any()) and torch.equal(outs["auto"][k], outs["x3"][k]), k
    # the fine model (refused pack): the f32x3 data gradient and the bf16x3 weight-gradient arithmetic -- the same products as the f32x3 mode's, summed
    # in chunks of other lengths (the batched launch sizes its chunks for the f16x2 blocks' row-bound cost, round 6): equal to a few ulps of the
    # sum; the coarse model (accepted): the h2 data gradient, fp32-level apart
    for k in gx3:
        if k.startswith("fine"):
            assert float((gau[k] - gx3[k]).norm() / (gx3[k].norm() + 1e-30)) <= 2e-6, k
        else:
            assert float((gau[k] - gx3[k]).norm() / (gx3[k].norm() + 1e-30)) <= 1e-4, k


def test_train_auto_gradients_through_repaired_rays():
    """The BACKWARD of a step whose forward had to repair rays: the rows of a repaired ray hold activations past 65,504, which the f16x2 weight
    gradients (the default behind the h2 data gradient) cannot take -- the waves that meet them redo their chunk on the bf16x3 stream
    (csrc/mlp_train16.hip wgrad_h2_kernel).  Every gradient is finite and as close to the all-f32x3 step's as two fp32-accurate steps of such a net are."""
    from crnerf_amd import autograd as AG
    from test_gpu_train_fused import _grads, _modules
    models, emb, args = _modules(gain=1.0)
    with torch.no_grad():
        for m in models.values():
            for l in (1, 2):
                m.state_dict()["xyz_encoding_%d.0.weight" % l][0, 0] = 250.0
    R, Nc, Ni = 64, 64, 64
    rays = synth.rays(R, seed=3, H=8, W=8).copy()
    rays[:32, 0:3] *= 1e-3
    rays[:32, 6] = 1e-4
    rays[:32, 7] = 2e-3
    rng = np.random.default_rng(9)
    z = C(np.sort(rng.uniform(rays[:, 6:7], rays[:, 7:8], (R, Nc)).astype(np.float32), -1))
    u, nc, nf = C(rng.uniform(0, 1, (R, Ni)).astype(np.float32)), C(rng.normal(size=(R, Nc)).astype(np.float32)), C(rng.normal(size=(R, Nc + Ni)).astype(np.float32))
    rd = C(rays)
    gw = torch.randn(R, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    outs = {}

    def run(tag):
        def f():
            out = AG.fused_render_with_grad(models["coarse"], models["fine"], rd, Nc, Ni, False, None, z, u, nc, nf, 1.0)
            outs[tag] = {k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}
            return (out["feature_fine"] * gw).sum() + 0.5 * (out["feature_coarse"] * gw).sum() + 0.1 * (out["weights_fine"] ** 2).sum()
        return f
    try:
        AG.set_training_forward_precision("f32h2")
        with torch.no_grad():
            run("h2")()
        AG.set_training_forward_precision("f32x3")
        gx3 = _grads(models, run("x3"))
        AG.set_training_forward_precision("auto")
        gau = _grads(models, run("auto"))
    finally:
        AG.set_training_forward_precision(None)
    bad = torch.isnan(outs["h2"]["feature_coarse"]).any(1) | torch.isnan(outs["h2"]["feature_fine"]).any(1)
    assert 0 < int(bad.sum()) < R, int(bad.sum())                    # some rays poisoned by the h2 core alone, not all
    assert not bool(torch.isnan(outs["auto"]["feature_fine"]).any())
    for k in gx3:
        assert bool(torch.isfinite(gau[k]).all()), k
        # (2e-2: the two steps differ in their data gradients -- h2 against x3 -- and a 1e-7 difference of a coarse weight moves a fine depth, see
        # test_gpu_train_fused.py's "auto" bar; measured 2e-3 on a bias, whose sum is the same fp32 code in both)
        assert float((gau[k] - gx3[k]).norm() / (gx3[k].norm() + 1e-30)) <= 2e-2, (k, float((gau[k] - gx3[k]).norm() / (gx3[k].norm() + 1e-30)))


@pytest.mark.parametrize("core,wmode", [("h2", "f16x2"), ("h2", 2), ("x3", 2), ("f32", 0)])
def test_backward_phases_are_the_halves_of_the_one_call_backward(core, wmode):
    """CRNERF_BWD_PHASE_DGRAD / _WGRAD (round 6): the data gradient and the weight gradients as two calls over one scratch -- on one stream and with
    the weight gradients on a CU-share stream behind an event (crnerf_stream_create_cu_share) -- give the one-call backward's gradients bit for bit."""
    import ctypes
    from crnerf_amd import _lib
    n = 70000
    st = synth.mlp_state(13, 2.0, 0.5)
    rng = np.random.default_rng(3)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1).to(DEV)
    d_out = torch.from_numpy(rng.normal(size=(n, 65)).astype(np.float32)).to(DEV)
    dev = {k: C(v) for k, v in st.items()}
    out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev), x)
    pk = ops.pack_mlp_weights_t_h2(dev) if core == "h2" else ops.pack_mlp_weights_t_x3(dev) if core == "x3" else ops.pack_mlp_weights_t(dev)
    kw = dict(wgrad_bf16=wmode, dgrad_h2=core == "h2", dgrad_x3=core == "x3")
    whole = ops.mlp_backward(pk, x, out, d_out, acts, **kw)
    scratch = ops.mlp_backward(pk, None, out, d_out, acts, phase="dgrad", **kw)
    halves = ops.mlp_backward(None, x, None, None, acts, phase="wgrad", scratch=scratch, **kw)
    for name, a, b in zip(ops.MLP_TENSOR_NAMES, halves, whole):
        assert torch.equal(a, b), name
    # the weight gradients on a stream that owns a quarter of every XCD's CUs, ordered behind the data gradient by an event
    lib = _lib.load()
    per = lib.crnerf_cus_per_xcd()
    assert per >= 4
    h = ctypes.c_void_p()
    _lib.check(lib.crnerf_stream_create_cu_share(ctypes.byref(h), per // 4, per // 4), "crnerf_stream_create_cu_share")
    try:
        side = torch.cuda.ExternalStream(h.value)
        scratch2 = ops.mlp_backward(pk, None, out, d_out, acts, phase="dgrad", **kw)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            shared = ops.mlp_backward(None, x, None, None, acts, phase="wgrad", scratch=scratch2, **kw)
        torch.cuda.current_stream().wait_stream(side)
        for name, a, b in zip(ops.MLP_TENSOR_NAMES, shared, whole):
            assert torch.equal(a, b), name
    finally:
        torch.cuda.synchronize()
        _lib.check(lib.crnerf_stream_destroy(h), "crnerf_stream_destroy")
    assert lib.crnerf_stream_create_cu_share(ctypes.byref(h), per, 1) != 0          # a share outside the XCD is refused
    with pytest.raises(ValueError, match="scratch"):
        ops.mlp_backward(None, x, None, None, acts, phase="wgrad", **kw)


@torch.no_grad()
def test_f16x2_weight_gradients_keep_small_activation_columns():
    """up to round 5 the activation operand of the f16x2 weight gradients went in unscaled -- below 2^-14 an fp16 piece is a
    subnormal, so a column of uniformly small activations (here: 64 units of xyz_encoding_3 scaled by 2^-12, and the raw coordinate columns of a
    scene in small units) kept 13-15 bits instead of 22, invisible to every test that measures errors against the tensor's LARGEST entry.  Since
    round 6 the h2 forward twin leaves the pass's largest |operand| behind the saved rows (the range word) and the weight gradients scale the
    activation operand by the power of two that puts it in [2^14, 2^15).  Checked per COLUMN against a float64 evaluation, next to bf16x3 (which has
    fp32's exponent and no such floor): the f16x2 result must be as good, column by column."""
    R, Nc = 96, 64
    st = synth.mlp_state(41, 1.5, 0.5)
    st["xyz_encoding_3.0.weight"][:64] *= np.float32(2.0 ** -12)
    st["xyz_encoding_3.0.bias"][:64] *= np.float32(2.0 ** -12)
    rays_np = synth.rays(R, seed=5)
    rays_np[:, 0:3] *= np.float32(0.01)          # a scene in small units: |xyz| ~ 0.02, the three raw-coordinate columns of the embedding are small
    rays_np[:, 6] = 0.5
    rays_np[:, 7] = 2.0
    rays = C(rays_np)
    dev = {k: C(v) for k, v in st.items()}
    zs = torch.linspace(0, 1, Nc, device=DEV)
    z = (rays[:, 6:7] * (1 - zs) + rays[:, 7:8] * zs).contiguous()
    trn = ops.render_rays(ops.pack_mlp_weights_h2(dev), None, rays, Nc, 0, train=True, precision="f32h2", z_coarse=z)
    P = R * Nc
    acts, raw = trn["acts_coarse"], trn["raw_coarse"].view(P, 65)
    rows = acts[:4 * 10 * P * 256].view(torch.float32).view(10, P, 256)
    word = acts[4 * 10 * P * 256 + 32 * 10 * P:][:4].view(torch.float32)
    x = ops.embed_points(rays, z, ops.posenc(rays[:, 3:6].contiguous(), 4))
    # the range word: the largest |operand| of the pass -- saved activations and embedded inputs
    top = max(float(rows[:9].abs().max()), float(rows[9, :, :128].abs().max()), float(x.abs().max()))
    assert float(word) == pytest.approx(top, rel=1e-5) and 0.5 < top < 100.0, (float(word), top)
    assert float(rows[2, :, :64].abs().max()) < 2.0 ** -9 and float(rows[2, :, 64:].abs().max()) > 2.0 ** -4          # the small columns ARE small
    g = torch.Generator().manual_seed(3)
    d_out = torch.randn(P, 65, generator=g).to(DEV)
    pack_t = ops.pack_mlp_weights_t_h2(dev)
    g2 = ops.mlp_backward(pack_t, x, raw, d_out, acts, wgrad_bf16="f16x2", dgrad_h2=True)
    g3 = ops.mlp_backward(pack_t, x, raw, d_out, acts, wgrad_bf16="bf16x3", dgrad_h2=True)
    with torch.enable_grad():
        w = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in st.items()}
        (O.mlp_forward(w, x.cpu().double()) * d_out.cpu().double()).sum().backward()
    worst = {}
    for name, cols in (("xyz_encoding_4.0.weight", slice(0, 64)), ("xyz_encoding_1.0.weight", slice(0, 3)), ("xyz_encoding_5.0.weight", slice(0, 3))):
        k = ops.MLP_TENSOR_NAMES.index(name)
        ref = w[name].grad
        colmax = ref.abs().amax(0).clamp_min(1e-300)
        e2 = ((g2[k].cpu().double() - ref).abs().amax(0) / colmax)[cols]
        e3 = ((g3[k].cpu().double() - ref).abs().amax(0) / colmax)[cols]
        worst[name] = (float(e2.max()), float(e3.max()))
        assert not torch.equal(g2[k], g3[k]), name + ": the f16x2 kernel did not run"
        # column by column as good as bf16x3 (both sit at the fp32 level of the shared data gradient's deltas: a few 1e-7 of the column's largest entry)
        assert bool((e2 <= torch.maximum(4.0 * e3, torch.full_like(e3, 2e-6))).all()), (name, worst[name])
    print("per-column max error / column max, small-activation columns: %s" % worst)
