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
