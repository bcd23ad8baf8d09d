"""GPU parity: every C-ABI entry point (through the ctypes shim) against the golden vectors produced
by the reference and against the CPU oracle on identical inputs.  All comparisons are fp32 with the
tolerance written next to them (north_star: "pixels within a stated fp32 tolerance").

Two properties of the REFERENCE ITSELF shape the end-to-end tolerances (measured in tests/_gpu_report.py):
  * N_emb_xyz = 15 puts 2^14 in front of x: a 1-ulp change of a sample depth (~2.4e-7 at z~3) moves
    the highest-frequency sin/cos argument by ~4e-3 rad, i.e. ~1e-4..1e-2 on rendered features;
  * sample_pdf divides by cdf gaps as small as eps=1e-5 and switches formula at `denom < eps`
    (rendering.py:41-42): in (near-)empty bins a 1-ulp cdf change moves a sample by up to a bin width.
So: kernels are held to ~1e-6 on IDENTICAL inputs (incl. the fine pass re-evaluated by the oracle at
the HIP path's own depths); the hierarchical depths are held tight where the reference is
well-conditioned and to "same bin" elsewhere; end-to-end fine outputs get a looser bound plus PSNR.
"""
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


def packed(state):
    return ops.pack_mlp_weights({k: C(v) for k, v in state.items()})


def close(got, want, atol, rtol=0.0):
    torch.testing.assert_close(got.detach().float().cpu(), torch.as_tensor(want).float(), atol=atol, rtol=rtol)


def assert_depths(z_got, z_ref, z_coarse, w_coarse, tight=3e-5):
    """Sorted fine depths: |dz| <= tight wherever the coarse pdf around the sample is well-conditioned;
    everywhere else the sample must stay inside the coarse interval the reference put it in."""
    z_got, z_ref = z_got.detach().cpu().double(), torch.as_tensor(z_ref).double()
    zc, w = torch.as_tensor(z_coarse).double(), torch.as_tensor(w_coarse).double()
    d = (z_got - z_ref).abs()
    mid = 0.5 * (zc[:, :-1] + zc[:, 1:])
    pdf = (w[:, 1:-1] + 1e-5) / (w[:, 1:-1] + 1e-5).sum(-1, keepdim=True)
    # interval index of every reference depth among the mid-points; its pdf mass
    idx = (torch.searchsorted(mid.contiguous(), z_ref.contiguous(), right=True) - 1).clamp(0, pdf.shape[1] - 1)
    mass = pdf.gather(1, idx)
    bin_w = (mid[:, 1:] - mid[:, :-1]).gather(1, idx)
    well = mass > 2e-3
    assert float(well.double().mean()) > 0.5
    assert float(d[well].max()) <= tight, "well-conditioned depths differ by %g" % float(d[well].max())
    assert bool((d[~well] <= bin_w[~well] + tight).all()), "ill-conditioned depth left its coarse interval"
    assert bool((z_got[:, 1:] >= z_got[:, :-1]).all()), "depths not ascending"


# ------------------------------------------------------------------ A1/A2 positional embedding
@torch.no_grad()
def test_posenc_golden(golden):
    g = golden("g1_posenc")
    close(ops.posenc(C(g["x"]), 15), g["xyz"], atol=2.5e-7)   # sin/cos at |arg| up to 8.2e4: <= 2 ulp of 1.0
    close(ops.posenc(C(g["x"]), 4), g["dir"], atol=2.5e-7)
    assert ops.posenc(torch.empty(0, 3, device=DEV), 15).shape == (0, 93)


# ------------------------------------------------------------------ A3/A4 MLP (fp32 MFMA core)
@torch.no_grad()
def test_mlp_golden(golden):
    g = golden("g2_mlp")
    x = C(g["x"])
    for tag, atol, rtol in (("default", 1e-6, 0.0), ("peaky", 3e-5, 1e-5)):
        pk = packed(synth.mlp_state(int(g["seed_" + tag]), float(g["gain_" + tag])))
        close(ops.mlp_forward(pk, x), g["out_" + tag], atol=atol, rtol=rtol)
        close(ops.mlp_forward(pk, x[:, :93].contiguous(), sigma_only=True), g["sigma_" + tag], atol=atol, rtol=rtol)


@torch.no_grad()
@pytest.mark.parametrize("n", [1, 31, 33, 127, 129, 1000])
def test_mlp_ragged_sizes_vs_oracle(n):
    st = synth.mlp_state(7, 2.0)
    pk, w = packed(st), O.to_torch(st)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    close(ops.mlp_forward(pk, x.to(DEV)), O.mlp_forward(w, x), atol=2e-5, rtol=1e-5)


@torch.no_grad()
def test_mlp_detects_transposed_or_permuted_packing():
    """Asymmetric weights: a one-hot input column must reproduce exactly that column of W1 (+bias)
    through layer 1 -- catches any row/column or slot-permutation error in the fragment packing."""
    st = synth.mlp_state(3, 1.0)
    w = O.to_torch(st)
    x = torch.zeros(120, 120)
    x[torch.arange(120), torch.arange(120)] = 1.0
    close(ops.mlp_forward(packed(st), x.to(DEV)), O.mlp_forward(w, x), atol=1e-6)


# ------------------------------------------------------------------ A5 compositing
@torch.no_grad()
def test_composite_golden(golden):
    g = golden("g3_composite")
    for tag, nstd in (("det", 0.0), ("noisy", 1.0)):
        for lvl, zk in (("coarse", "z_coarse"), ("fine", "z_fine_" + tag)):
            w, f, d = ops.composite(C(g["raw_" + lvl]), C(g[zk]), C(g["noise_" + lvl]), nstd)
            close(w, g["%s__weights_%s" % (tag, lvl)], atol=2e-6)
            close(f, g["%s__feature_%s" % (tag, lvl)], atol=3e-6)
            close(d, g["%s__depth_%s" % (tag, lvl)], atol=1e-5)


@torch.no_grad()
@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 300])
def test_composite_ragged_vs_oracle(N):
    rng = np.random.default_rng(N)
    raw = rng.uniform(0, 1, (9, N, 65)).astype(np.float32)
    raw[..., 64] *= 20
    z = np.sort(rng.uniform(0.2, 5, (9, N)).astype(np.float32), -1)
    w, f, d = ops.composite(C(raw), C(z))
    wo, fo, do = O.composite(torch.from_numpy(raw), torch.from_numpy(z))
    close(w, wo, atol=2e-6), close(f, fo, atol=3e-6), close(d, do, atol=1e-5)


# ------------------------------------------------------------------ A6/A7 sample_pdf + merge
@torch.no_grad()
def test_sample_pdf_golden(golden):
    g, tab = golden("g4_sample_pdf"), golden("g5_render")
    zc = g["z_coarse"]
    wfull = np.zeros((64, 64), np.float32)
    wfull[:, 1:-1] = g["weights"]
    for ni, key, u in ((64, "det_64", tab["u_steps_64"]), (128, "det_128", tab["u_steps_128"]), (128, "rand_128", g["u_128"])):
        zs, smp = ops.sample_pdf_merge(C(zc), C(wfull), ni, u=C(u), return_samples=True)
        ref = torch.from_numpy(g[key]).double()
        d = (smp.cpu().double() - ref).abs()
        # conditioning of each reference sample: the cdf gap it was interpolated in
        w = torch.from_numpy(g["weights"]).double() + 1e-5
        pdf = w / w.sum(-1, keepdim=True)
        bins = torch.from_numpy(g["bins"]).double()
        idx = (torch.searchsorted(bins.contiguous(), ref.contiguous(), right=True) - 1).clamp(0, 61)
        well = pdf.gather(1, idx) > 2e-3
        assert float(d[well].max()) <= 3e-5      # well-conditioned samples: 3e-5 absolute on depths <= 5
        width = (bins[:, 1:] - bins[:, :-1]).gather(1, idx)
        assert bool((d[~well] <= width[~well] + 1e-5).all())
        want = torch.sort(torch.cat([torch.from_numpy(zc), smp.cpu()], -1), -1)[0]
        assert torch.equal(zs.cpu(), want)       # merge is exact: a permutation of its inputs, ascending


@torch.no_grad()
def test_sample_pdf_in_kernel_linspace_matches_table(golden):
    g = golden("g4_sample_pdf")
    wfull = np.zeros((64, 64), np.float32)
    wfull[:, 1:-1] = g["weights"]
    a = ops.sample_pdf_merge(C(g["z_coarse"]), C(wfull), 128)
    b = ops.sample_pdf_merge(C(g["z_coarse"]), C(wfull), 128, u=torch.linspace(0, 1, 128).to(DEV))
    assert float((a - b).abs().max()) <= 5e-4    # tables differ by <= 1 ulp of u; see module docstring


# ------------------------------------------------------------------ A0 fused renderer
def _models(g):
    st_c = synth.mlp_state(int(g["seed_coarse"]), float(g["gain"]), float(g["sigma_bias"]))
    st_f = synth.mlp_state(int(g["seed_fine"]), float(g["gain"]), float(g["sigma_bias"]))
    return st_c, st_f


@torch.no_grad()
@pytest.mark.parametrize("tag,ni,disp", [("c64", 0, False), ("c64_f128", 128, False), ("c64_f128_disp", 128, True), ("c64_f64", 64, False)])
def test_render_golden(golden, tag, ni, disp):
    g = golden("g5_render")
    st_c, st_f = _models(g)
    out = ops.render_rays(packed(st_c), packed(st_f) if ni else None, C(g["rays"]), 64, ni, use_disp=disp,
                          z_steps=C(g["z_steps_64"]), u=C(g["u_steps_%d" % ni]) if ni else None, want_z_fine=True)
    # coarse pass: identical inputs all the way (same linspace table) -> kernel-level tolerance
    close(out["weights_coarse"], g[tag + "__weights_coarse"], atol=3e-6)
    close(out["feature_coarse"], g[tag + "__feature_coarse"], atol=1e-5)
    close(out["depth_coarse"], g[tag + "__depth_coarse"], atol=1e-5)
    if not ni:
        return
    z_coarse = O.coarse_depths(torch.from_numpy(g["rays"]), 64, disp, torch.from_numpy(g["z_steps_64"]))
    assert_depths(out["z_fine"], g[tag + "__z_fine"], z_coarse, g[tag + "__weights_coarse"])
    # fine pass at IDENTICAL depths: the oracle re-evaluates the fine model at the HIP path's z_fine
    rays = torch.from_numpy(g["rays"])
    zf = out["z_fine"].cpu()
    raw = O._run_model(O.to_torch(st_f), rays, zf, O.posenc(rays[:, 3:6], 4), 32768)
    w2, f2, d2 = O.composite(raw, zf)
    close(out["weights_fine"], w2, atol=3e-6)
    close(out["feature_fine"], f2, atol=1e-5)
    close(out["depth_fine"], d2, atol=2e-5)
    # end to end against the reference's own fine outputs (includes the reference's ill-conditioning)
    close(out["feature_fine"], g[tag + "__feature_fine"], atol=0.15)
    rel = float((out["feature_fine"].cpu() - torch.from_numpy(g[tag + "__feature_fine"])).norm() / torch.from_numpy(g[tag + "__feature_fine"]).norm())
    assert rel < 3e-2
    s = out["weights_fine"].sum(-1).cpu()
    assert float((s - 1).abs().max()) < 1e-5     # last delta = 1e2 forces the weights to sum to 1


@torch.no_grad()
def test_render_view_dir_override(golden):
    g = golden("g5_render")
    st_c, st_f = _models(g)
    out = ops.render_rays(packed(st_c), packed(st_f), C(g["rays"]), 64, 128, view_dir=C(g["view_dir"]),
                          z_steps=C(g["z_steps_64"]), u=C(g["u_steps_128"]))
    ref = torch.from_numpy(g["viewdir__feature_fine"])
    assert float((out["feature_fine"].cpu() - ref).norm() / ref.norm()) < 3e-2


@torch.no_grad()
def test_render_fused_equals_unfused_hip_pipeline(golden):
    """The fused kernel against the same computation assembled from the stand-alone HIP entry points."""
    g = golden("g5_render")
    st_c, st_f = _models(g)
    pc, pf, rays = packed(st_c), packed(st_f), C(g["rays"])
    zt, ut = C(g["z_steps_64"]), C(g["u_steps_128"])
    fused = ops.render_rays(pc, pf, rays, 64, 128, z_steps=zt, u=ut, want_z_fine=True)
    near, far = rays[:, 6:7], rays[:, 7:8]
    z = (near * (1 - zt) + far * zt).contiguous()
    demb = ops.posenc(rays[:, 3:6].contiguous(), 4)

    def run(pk, zz):
        R, N = zz.shape
        pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * zz[..., None]).reshape(-1, 3).contiguous()
        x = torch.cat([ops.posenc(pts, 15), demb[:, None, :].expand(R, N, 27).reshape(R * N, 27)], 1).contiguous()
        return ops.composite(ops.mlp_forward(pk, x).view(R, N, 65), zz)

    w, f, d = run(pc, z)
    close(fused["weights_coarse"], w.cpu(), atol=1e-6), close(fused["feature_coarse"], f.cpu(), atol=2e-6)
    zf = ops.sample_pdf_merge(z, fused["weights_coarse"], 128, u=ut)
    assert torch.equal(zf, fused["z_fine"])     # same device functions on the same inputs
    w, f, d = run(pf, zf)
    close(fused["weights_fine"], w.cpu(), atol=1e-6), close(fused["feature_fine"], f.cpu(), atol=2e-6)


@torch.no_grad()
def test_render_full_size_properties():
    """BASELINE config 2 (1024 rays x 64+128) and a ragged ray count: size-independent properties."""
    pc, pf = packed(synth.mlp_state(1, 3.0, 1.0)), packed(synth.mlp_state(2, 3.0, 1.0))
    rays = C(synth.rays(1024, seed=0))
    full = ops.render_rays(pc, pf, rays, 64, 128, want_z_fine=True)
    for k, v in full.items():
        assert torch.isfinite(v).all(), k
    s = full["weights_fine"].sum(-1)            # = 1 - prod(1 - alpha): at most 1, and 1 unless the last sigma ~ 0
    assert float(s.max()) <= 1 + 1e-5 and float(s.min()) > 0.99
    assert float(full["weights_fine"].min()) >= 0 and float(full["weights_coarse"].min()) >= 0
    assert bool((full["z_fine"][:, 1:] >= full["z_fine"][:, :-1]).all())
    near, far = rays[:, 6], rays[:, 7]
    assert bool((full["z_fine"][:, 0] >= near - 1e-6).all()) and bool((full["z_fine"][:, -1] <= far + 1e-6).all())
    assert bool((full["depth_fine"] >= near - 1e-4).all()) and bool((full["depth_fine"] <= far + 1e-4).all())
    # rays are independent: any split of the batch gives bit-identical rows (the reference's chunk invariance, G7)
    for lo, hi in ((0, 1), (1, 6), (6, 517), (517, 1024)):
        part = ops.render_rays(pc, pf, rays[lo:hi].contiguous(), 64, 128)
        for k in part:
            assert torch.equal(part[k], full[k][lo:hi]), (k, lo, hi)
    # coarse-only equals the coarse half of the hierarchical run
    c = ops.render_rays(pc, None, rays, 64, 0)
    assert torch.equal(c["feature_coarse"], full["feature_coarse"]) and torch.equal(c["weights_coarse"], full["weights_coarse"])
    assert ops.render_rays(pc, pf, rays[:0].contiguous(), 64, 128)["feature_fine"].shape == (0, 64)


@torch.no_grad()
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_render_dynamic_quad_scheduling_is_invisible(precision):
    """Launches with more ray quads than workgroups pull quads from a device counter (kernels.h "Dynamic work distribution");
    launches that fit one pass use the static map.  Rays are independent, so both must give the same bits, launch after launch
    (the counter slot is left zeroed by the last workgroup), for a ragged ray count too."""
    pk = lambda st: ops.pack_mlp_weights({k: C(v) for k, v in st.items()}, precision=precision)  # noqa: E731
    pc, pf = pk(synth.mlp_state(1, 3.0, 1.0)), pk(synth.mlp_state(2, 3.0, 1.0))
    rays = C(synth.rays(5003, seed=2))
    kw = dict(z_steps=torch.linspace(0, 1, 64, device=DEV), u=torch.linspace(0, 1, 32, device=DEV), precision=precision)
    big = [ops.render_rays(pc, pf, rays, 64, 32, **kw) for _ in range(3)]          # 1251 quads > 256 workgroups: dynamic
    parts = [ops.render_rays(pc, pf, rays[lo:lo + 1000].contiguous(), 64, 32, **kw) for lo in range(0, 5003, 1000)]   # <= 250 quads: static
    for k in big[0]:
        ref = torch.cat([q[k] for q in parts], 0)
        for b in big:
            assert torch.equal(b[k], ref), k


@torch.no_grad()
@pytest.mark.parametrize("nc,ni", [(64, 64), (48, 40), (256, 256), (33, 1), (3, 5)])
def test_render_other_sample_counts_vs_oracle(nc, ni):
    st_c, st_f = synth.mlp_state(5, 3.0, 1.0), synth.mlp_state(6, 3.0, 1.0)
    rays_np = synth.rays(8, seed=3)
    zt, ut = torch.linspace(0, 1, nc), torch.linspace(0, 1, ni)
    out = ops.render_rays(packed(st_c), packed(st_f), C(rays_np), nc, ni, z_steps=zt.to(DEV), u=ut.to(DEV), want_z_fine=True)
    rays = torch.from_numpy(rays_np)
    ref = O.render_rays(O.to_torch(st_c), O.to_torch(st_f), rays, nc, ni, z_steps=zt, u=ut)
    close(out["weights_coarse"], ref["weights_coarse"], atol=3e-6)
    close(out["feature_coarse"], ref["feature_coarse"], atol=1e-5)
    zf = out["z_fine"].cpu()
    raw = O._run_model(O.to_torch(st_f), rays, zf, O.posenc(rays[:, 3:6], 4), 32768)
    w2, f2, _ = O.composite(raw, zf)
    close(out["weights_fine"], w2, atol=3e-6), close(out["feature_fine"], f2, atol=1e-5)


@torch.no_grad()
def test_render_noise_and_external_depths_vs_oracle():
    """The training-time inputs (perturb > 0, noise_std > 0) enter as tensors: stratified depths, uniforms, normals."""
    st_c, st_f = synth.mlp_state(8, 3.0, 1.0), synth.mlp_state(9, 3.0, 1.0)
    rng = np.random.default_rng(4)
    rays_np = synth.rays(16, seed=2)
    z = np.sort(rng.uniform(rays_np[:, 6:7], rays_np[:, 7:8], (16, 64)).astype(np.float32), -1)
    u = rng.uniform(0, 1, (16, 128)).astype(np.float32)
    nc, nf = rng.normal(size=(16, 64)).astype(np.float32), rng.normal(size=(16, 192)).astype(np.float32)
    out = ops.render_rays(packed(st_c), packed(st_f), C(rays_np), 64, 128, z_coarse=C(z), u=C(u), noise_coarse=C(nc), noise_fine=C(nf),
                          noise_std=1.0, want_z_fine=True)
    T = torch.from_numpy
    ref = O.render_rays(O.to_torch(st_c), O.to_torch(st_f), T(rays_np), 64, 128, z_coarse=T(z), u=T(u), noise_coarse=T(nc),
                        noise_fine=T(nf), noise_std=1.0)
    close(out["weights_coarse"], ref["weights_coarse"], atol=3e-6)
    close(out["feature_coarse"], ref["feature_coarse"], atol=1e-5)
    zf = out["z_fine"].cpu()
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())
    raw = O._run_model(O.to_torch(st_f), T(rays_np), zf, O.posenc(T(rays_np)[:, 3:6], 4), 32768)
    w2, f2, _ = O.composite(raw, zf, T(nf), 1.0)
    close(out["weights_fine"], w2, atol=3e-6), close(out["feature_fine"], f2, atol=1e-5)


@torch.no_grad()
def test_reference_signature_general_path_large_sample_counts_and_jitter():
    """N_samples/N_importance beyond the fused kernel (eval-style 256+256 is fused; 320+300 is not) and
    args.pertubeCord go through the un-fused HIP pipeline; compared with the oracle at its own depths."""
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    from crnerf_amd.models.rendering import render_rays_cross_ray
    st_c, st_f = synth.mlp_state(5, 3.0, 1.0), synth.mlp_state(6, 3.0, 1.0)
    args = _Args()
    models = {"coarse": NeRF_sigma("coarse", args, in_channels_xyz=93, in_channels_dir=27).to(DEV),
              "fine": NeRF_sigma("fine", args, in_channels_xyz=93, in_channels_dir=27, encode_random=True).to(DEV)}
    models["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in st_c.items()})
    models["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in st_f.items()})
    emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
    rays_np = synth.rays(6, seed=11)
    res = render_rays_cross_ray(models, emb, C(rays_np), None, 320, False, 0, 0, 300, 4096, False, args=args)
    assert res["weights_fine"].shape == (6, 620) and res["feature_fine_random"] is res["feature_fine"]
    rays = torch.from_numpy(rays_np)
    ref = O.render_rays(O.to_torch(st_c), O.to_torch(st_f), rays, 320, 0, z_steps=torch.linspace(0, 1, 320))
    close(res["feature_coarse"], ref["feature_coarse"], atol=5e-4)          # device-built linspace table (1-ulp depths)
    assert float((res["weights_fine"].sum(-1) - 1).abs().max()) < 1e-4

    class Jit(_Args):
        pertubeCord = True
    a = render_rays_cross_ray(models, emb, C(rays_np), None, 64, False, 0, 0, 0, 4096, False, args=Jit())
    b = render_rays_cross_ray(models, emb, C(rays_np), None, 64, False, 0, 0, 0, 4096, False, args=args)
    d = float((a["feature_coarse"] - b["feature_coarse"]).abs().max())
    assert 0 < d < 0.2                                                      # 1e-5 jitter x 2^14 frequencies: visible, bounded


def test_render_rejects_unsupported_sizes():
    pc = packed(synth.mlp_state(1))
    rays = C(synth.rays(4))
    with pytest.raises(RuntimeError, match="N_samples"):
        ops.render_rays(pc, pc, rays, 300, 0)
    with pytest.raises(RuntimeError, match="N_importance"):
        ops.render_rays(pc, pc, rays, 64, 300)
    with pytest.raises(RuntimeError):
        ops.render_rays(pc, None, rays, 64, 16)


# ------------------------------------------------------------------ A8/A9 cross-ray decoder
class _Args:
    nerf_out_dim, img_wh, pertubeCord = 64, [40, 24], False


@torch.no_grad()
def test_decoder_golden(golden):
    from crnerf_amd.models.linearStyleTransfer import style_net
    g = golden("g6_decoder")
    net = style_net(_Args()).to(DEV)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(int(g["seed"])).items()})
    close(net(C(g["content"]), C(g["style"])), g["rgb"], atol=2e-6)
    close(net(C(g["content"]), None, type="content"), g["rgb_content"], atol=1e-6)
    # the reference's caller-side layout (eval.py:291-294): feature[HW,64] -> transposed view -> decoder
    feat = C(g["content"])[0].reshape(64, -1).t().contiguous()
    grid = feat.t().reshape(1, 64, 24, 40)
    close(net(grid, C(g["style"])), g["rgb"], atol=2e-6)


@torch.no_grad()
@pytest.mark.parametrize("H,W", [(1, 1), (7, 13), (100, 100)])
def test_decoder_ragged_grids_vs_oracle(H, W):
    from crnerf_amd.models.linearStyleTransfer import style_net
    st = synth.decoder_state(5, 2.0)
    net = style_net(_Args()).to(DEV)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    rng = np.random.default_rng(H * W)
    content = rng.uniform(0, 1, (1, 64, H, W)).astype(np.float32)
    style = rng.uniform(0, 1, (1, 64, 32, 32)).astype(np.float32)
    ref = O.crossray_decode(O.to_torch(st), torch.from_numpy(content), torch.from_numpy(style))
    close(net(C(content), C(style)), ref, atol=5e-6)


# ------------------------------------------------------------------ reference-signature modules end to end
@torch.no_grad()
def test_reference_call_signature_end_to_end(golden):
    """eval.py's sequence (batched_inference :29-59 + decode :288-295) on the drop-in modules."""
    from crnerf_amd.models.linearStyleTransfer import style_net
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    from crnerf_amd.models.rendering import render_rays_cross_ray
    g = golden("g5_render")
    st_c, st_f = _models(g)
    args = _Args()
    models = {"coarse": NeRF_sigma("coarse", args, in_channels_xyz=93, in_channels_dir=27).to(DEV),
              "fine": NeRF_sigma("fine", args, in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, in_channels_a=48,
                                 encode_random=True).to(DEV),
              "decoder": style_net(args).to(DEV)}
    models["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in st_c.items()})
    models["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in st_f.items()})
    dst = synth.decoder_state(31)
    models["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in dst.items()})
    emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
    rays, ts = C(g["rays"]), torch.zeros(64, dtype=torch.long, device=DEV)
    chunks = [render_rays_cross_ray(models, emb, rays[i:i + 24], ts[i:i + 24], 64, False, 0, 0, 128, 24, False, test_time=True, args=args)
              for i in range(0, 64, 24)]
    res = {k: torch.cat([c[k] for c in chunks], 0) for k in chunks[0]}
    assert list(res) == ["weights_coarse", "feature_coarse", "depth_coarse", "weights_fine", "feature_fine", "feature_fine_random", "depth_fine"]
    assert chunks[0]["feature_fine_random"] is chunks[0]["feature_fine"]
    close(res["feature_coarse"], g["c64_f128__feature_coarse"], atol=5e-4)   # device linspace table, not the fixture's
    # decode an 8x8 grid with the reference glue, compare images by PSNR against the oracle image
    feat = res["feature_fine"]
    grid = feat.t().reshape(1, 64, 8, 8)
    style = C(np.random.default_rng(1).uniform(0, 1, (1, 64, 32, 32)).astype(np.float32))
    img = models["decoder"](grid, style).cpu()
    ref_feat = torch.from_numpy(g["c64_f128__feature_fine"])
    ref_img = O.crossray_decode(O.to_torch(dst), O.feature_to_grid(ref_feat, 8, 8), style.cpu())
    target = ref_img + 0.05 * torch.from_numpy(np.random.default_rng(2).normal(size=ref_img.shape).astype(np.float32))
    assert abs(O.psnr(img, target) - O.psnr(ref_img, target)) < 0.05      # north_star: PSNR within 0.05 dB
    # module-level forwards
    x = torch.cat([emb["xyz"](rays[:, :3].contiguous()), emb["dir"](rays[:, 3:6].contiguous())], 1)
    assert x.shape == (64, 120)
    close(models["fine"](x), O.mlp_forward(O.to_torch(st_f), x.cpu()), atol=3e-5, rtol=1e-5)


# ------------------------------------------------------------------ backward twins (training)
@pytest.mark.parametrize("N,with_extras", [(64, False), (192, True), (1, False), (65, True), (300, True)])
def test_composite_backward_vs_autograd_oracle(N, with_extras):
    """crnerf_composite_backward_f32 against torch autograd through the oracle's compositing."""
    rng = np.random.default_rng(N)
    R = 7
    raw = rng.uniform(0, 1, (R, N, 65)).astype(np.float32)
    raw[..., 64] = (rng.normal(size=(R, N)) * rng.choice([0.3, 3, 30], size=(R, 1))).astype(np.float32)   # negatives -> relu gate
    raw[0, :, 64] = 0.0
    if R > 1:
        raw[1, :, 64] = 500.0                                    # alpha -> 1 after the first sample
    z = np.sort(rng.uniform(0.2, 5, (R, N)).astype(np.float32), -1)
    noise = rng.normal(size=(R, N)).astype(np.float32)
    gf = rng.normal(size=(R, 64)).astype(np.float32)
    gd = rng.normal(size=(R,)).astype(np.float32) if with_extras else None
    gw = rng.normal(size=(R, N)).astype(np.float32) if with_extras else None
    nstd = 0.5 if with_extras else 0.0
    rt = torch.from_numpy(raw).requires_grad_(True)
    w, f, d = O.composite(rt, torch.from_numpy(z), torch.from_numpy(noise), nstd)
    loss = (f * torch.from_numpy(gf)).sum()
    if with_extras:
        loss = loss + (d * torch.from_numpy(gd)).sum() + (w * torch.from_numpy(gw)).sum()
    loss.backward()
    with torch.no_grad():
        got = ops.composite_backward(C(raw), C(z), C(gf), None if gd is None else C(gd), None if gw is None else C(gw),
                                     noise=C(noise), noise_std=nstd)
    ref = rt.grad
    scale = float(ref.abs().max()) + 1e-6
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * scale + 1e-6


@pytest.mark.parametrize("n,gain", [(37, 1.0), (400, 2.0)])
def test_mlp_backward_vs_autograd_oracle(n, gain):
    """crnerf_mlp_forward_train_f32 + crnerf_mlp_backward_f32 against torch autograd through the oracle MLP."""
    st = synth.mlp_state(13, gain, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    d_out = torch.from_numpy(rng.normal(size=(n, 65)).astype(np.float32))
    w = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st.items()}
    ref_out = O.mlp_forward(w, x)
    (ref_out * d_out).sum().backward()
    dev_state = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
        close(out, ref_out.detach(), atol=3e-5, rtol=1e-5)
        grads = ops.mlp_backward(ops.pack_mlp_weights_t(dev_state), x.to(DEV), out, d_out.to(DEV), acts)
    for name, gq in zip(ops.MLP_TENSOR_NAMES, grads):
        ref = w[name].grad
        err = float((gq.cpu() - ref).abs().max())
        tol = 2e-4 * float(ref.abs().max()) + 1e-5
        assert err <= tol, "%s: max|d| %.3e > %.3e (|ref|max %.3e)" % (name, err, tol, float(ref.abs().max()))


@pytest.mark.parametrize("core", ["h2", "x3"])
def test_split_core_backward_vs_autograd_oracle(core):
    """crnerf_mlp_backward_{h2,x3}_f32 (data gradient on the fp16 / bf16 matrix cores) + the bf16x3 weight gradients -- the kernels of the
    training default -- held DIRECTLY to torch autograd through oracle.cpu_ref.mlp_forward (n = 400, gain 2), at the fp32 twin's own bar
    (test_mlp_backward_vs_autograd_oracle), not to the fp32 HIP twin."""
    n, gain = 400, 2.0
    st = synth.mlp_state(13, gain, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    d_out = torch.from_numpy(rng.normal(size=(n, 65)).astype(np.float32))
    d_out[::7] *= 1e-6
    w = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st.items()}
    ref_out = O.mlp_forward(w, x)
    (ref_out * d_out).sum().backward()
    dev_state = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
        close(out, ref_out.detach(), atol=3e-5, rtol=1e-5)
        pack_t = ops.pack_mlp_weights_t_h2(dev_state) if core == "h2" else ops.pack_mlp_weights_t_x3(dev_state)
        grads = ops.mlp_backward(pack_t, x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="bf16x3", dgrad_h2=core == "h2", dgrad_x3=core == "x3")
    for name, gq in zip(ops.MLP_TENSOR_NAMES, grads):
        ref = w[name].grad
        err = float((gq.cpu() - ref).abs().max())
        tol = 2e-4 * float(ref.abs().max()) + 1e-5
        assert bool(torch.isfinite(gq).all()) and err <= tol, "%s: max|d| %.3e > %.3e (|ref|max %.3e)" % (name, err, tol, float(ref.abs().max()))


def _backward_case(n, seed=13, gain=2.0):
    st = synth.mlp_state(seed, gain, 0.5)
    rng = np.random.default_rng(n)
    x = torch.cat([O.posenc(torch.from_numpy(rng.uniform(-2, 2, (n, 3)).astype(np.float32)), 15),
                   O.posenc(torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)), 4)], 1)
    d_out = torch.from_numpy(rng.normal(size=(n, 65)).astype(np.float32))
    d_out[::7] *= 1e-6
    return st, x, d_out


def test_f16x2_weight_gradients_vs_autograd_oracle():
    """CRNERF_BWD_WGRAD_F16X2 (the weight gradients of the training default: two-piece fp16 splits of the full 256 x 256 blocks, ranged by the
    largest |delta| the h2 data gradient saw) against torch autograd through oracle.cpu_ref.mlp_forward at the fp32 twin's own bar, on deltas
    that span twelve decades (every seventh point 1e-6 of the rest, the whole batch then scaled 1e-6 / 1 / 1e+6: the range word has to follow)."""
    n = 1184                                   # whole 32-point pairs of k-steps in every chunk but a ragged last one
    st, x, d_out = _backward_case(n)
    dev_state = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
        pack_t = ops.pack_mlp_weights_t_h2(dev_state)
    for scale in (1.0, 1e-6, 1e6):
        w = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st.items()}
        (O.mlp_forward(w, x) * (d_out * scale)).sum().backward()
        with torch.no_grad():
            grads = ops.mlp_backward(pack_t, x.to(DEV), out, (d_out * scale).to(DEV), acts, wgrad_bf16="f16x2", dgrad_h2=True)
        for name, gq in zip(ops.MLP_TENSOR_NAMES, grads):
            ref = w[name].grad
            err = float((gq.cpu() - ref).abs().max())
            tol = 2e-4 * float(ref.abs().max()) + 1e-5 * scale
            assert bool(torch.isfinite(gq).all()) and err <= tol, "scale %g, %s: max|d| %.3e > %.3e" % (scale, name, err, tol)


def test_f16x2_weight_gradients_are_as_accurate_as_bf16x3():
    """Against a float64 evaluation of the same products (the device's own deltas and activations: D^T A per layer), the two-piece fp16 form and the
    three-piece bf16 form sit at the same distance -- both are fp32-accurate; fp16's eleven-bit pieces lose nothing a reader of the fp32 result
    would see."""
    n = 4096
    st, x, d_out = _backward_case(n, seed=29)
    dev_state = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
        pack_t = ops.pack_mlp_weights_t_h2(dev_state)
        g2 = ops.mlp_backward(pack_t, x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="f16x2", dgrad_h2=True)
        g3 = ops.mlp_backward(pack_t, x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="bf16x3", dgrad_h2=True)
    w = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in st.items()}
    (O.mlp_forward(w, x.double()) * d_out.double()).sum().backward()
    for k, name in enumerate(ops.MLP_TENSOR_NAMES):
        ref = w[name].grad
        top = float(ref.abs().max())
        e2, e3 = float((g2[k].cpu().double() - ref).abs().max()) / top, float((g3[k].cpu().double() - ref).abs().max()) / top
        assert e2 <= max(2.0 * e3, 2e-6), "%s: f16x2 %.3e vs bf16x3 %.3e of the largest entry" % (name, e2, e3)
        if name.startswith("xyz_encoding_") and name.endswith("weight") and name not in ("xyz_encoding_1.0.weight",):
            assert not torch.equal(g2[k], g3[k]), name + ": the f16x2 kernel did not run"


def test_f16x2_weight_gradients_fall_back_bit_for_bit():
    """Operands outside fp16's range -- what the saved rows of a ray the forward had to repair may hold (here: two activation columns set to 7e4 and
    1e5).  A wave that meets one redoes its chunk on the bf16x3 stream: every tensor equals the bf16x3 call's bit for bit, and nothing is Inf / NaN
    (fp16(7e4) is Inf: a result kept from the fp16 stream would be)."""
    n = 2048
    st, x, d_out = _backward_case(n, seed=31)
    dev_state = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
        pack_t = ops.pack_mlp_weights_t_h2(dev_state)
        rows = acts.view(torch.float32)[: 10 * n * 256].view(10, n, 256)
        rows[:, :, 5::64] = 7.0e4                   # a column in every 32-wide tile of every saved activation, all points: every wave of every job
        rows[:, :, 37::64] = 1.0e5                  # that reads activations meets a value fp16 cannot hold
        g2 = ops.mlp_backward(pack_t, x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="f16x2", dgrad_h2=True)
        g3 = ops.mlp_backward(pack_t, x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="bf16x3", dgrad_h2=True)
    kept = 0
    for k, name in enumerate(ops.MLP_TENSOR_NAMES):
        assert bool(torch.isfinite(g2[k]).all()), name
        a, b = g2[k], g3[k]
        # the blocks that multiply the embedded INPUT (in range by construction) keep their fp16 result: close to, not equal to, bf16x3's
        emb = {"xyz_encoding_1.0.weight": slice(0, 93), "xyz_encoding_5.0.weight": slice(0, 93), "dir_encoding.0.weight": slice(256, 283)}.get(name)
        if emb is not None:
            close(a[:, emb], b[:, emb].cpu(), atol=2e-5 * float(b.abs().max()), rtol=0)
            kept += int(not torch.equal(a[:, emb], b[:, emb]))
            keep = torch.ones(a.shape[1], dtype=torch.bool)
            keep[emb] = False
            a, b = a[:, keep], b[:, keep]
        assert torch.equal(a, b), "%s: max|d| %.3e" % (name, float((a - b).abs().max()))
    assert kept == 3, "the narrow blocks did not run on the fp16 form"


def test_f16x2_weight_gradients_per_layer_launches_match_the_batched_launch():
    """Beyond 2^18 points every weight-gradient job is its own launch (wgrad_h2_kernel, wgrad_h2_narrow_kernel for the two smallest) instead of one
    batched launch: the same arithmetic over other chunks of points.  Forced here at a small size (CRNERF_WGRAD_BATCH=0 is read once per process, so
    the per-layer leg runs in a child process) and compared tensor by tensor: fp32 summation-order distance, nothing more."""
    import os
    import subprocess
    import sys
    import tempfile
    n = 2500                                                       # ragged last chunk included
    st, x, d_out = _backward_case(n, seed=37)
    dev_state = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
        g_batched = ops.mlp_backward(ops.pack_mlp_weights_t_h2(dev_state), x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="f16x2", dgrad_h2=True)
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import crnerf_amd.synth as synth
from crnerf_amd import ops
from test_gpu_parity import _backward_case, C, DEV
st, x, d_out = _backward_case(%d, seed=37)
dev_state = {k: C(v) for k, v in st.items()}
with torch.no_grad():
    out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
    g = ops.mlp_backward(ops.pack_mlp_weights_t_h2(dev_state), x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="f16x2", dgrad_h2=True)
np.savez(sys.argv[1], *[t.cpu().numpy() for t in g])
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), n)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "g.npz")
        env = dict(os.environ, CRNERF_WGRAD_BATCH="0")
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=600)
        per_layer = np.load(path)
        for k, name in enumerate(ops.MLP_TENSOR_NAMES):
            a, b = torch.from_numpy(per_layer["arr_%d" % k]), g_batched[k].cpu()
            assert bool(torch.isfinite(a).all()), name
            assert float((a - b).abs().max()) <= 3e-6 * float(b.abs().max()) + 1e-9, "%s: max|d| %.3e of %.3e" % (name, float((a - b).abs().max()), float(b.abs().max()))


def test_f16x2_weight_gradients_need_the_h2_data_gradient():
    st, x, d_out = _backward_case(64)
    dev_state = {k: C(v) for k, v in st.items()}
    with torch.no_grad():
        out, acts = ops.mlp_forward_train(ops.pack_mlp_weights(dev_state), x.to(DEV))
        with pytest.raises(ValueError, match="h2 data gradient"):
            ops.mlp_backward(ops.pack_mlp_weights_t_x3(dev_state), x.to(DEV), out, d_out.to(DEV), acts, wgrad_bf16="f16x2", dgrad_x3=True)


@pytest.mark.parametrize("mode", ["f32", "auto"])
def test_training_step_gradients_vs_autograd_oracle(mode):
    """A training-style step on the reference-signature modules (grad mode, perturb/noise inputs fixed):
    render -> decode coarse+fine -> MSE; parameter gradients against torch autograd through the oracle -- in "f32" (every product on the
    fp32 matrix cores) and in "auto", the training default (h2 forward / data gradient, bf16x3 weight gradients): both are fp32-accurate
    and held to 2e-4 of each tensor's largest gradient entry (smoke() measures 3.5e-6 on the same kind of step)."""
    from crnerf_amd import autograd as AG
    AG.set_training_forward_precision(mode)
    AG.set_wgrad_precision("f32" if mode == "f32" else None)
    try:
        _training_step_gradients_vs_autograd_oracle()
    finally:
        AG.set_training_forward_precision(None)
        AG.set_wgrad_precision(None)


def _training_step_gradients_vs_autograd_oracle():
    from crnerf_amd.models.linearStyleTransfer import style_net
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    from crnerf_amd.models.rendering import render_rays_cross_ray
    R, H, W, Nc, Ni = 16, 4, 4, 64, 64
    st_c, st_f, dst = synth.mlp_state(41, 2.0, 0.5), synth.mlp_state(42, 2.0, 0.5), synth.decoder_state(43)
    args = _Args()
    models = {"coarse": NeRF_sigma("coarse", args, in_channels_xyz=93, in_channels_dir=27).to(DEV),
              "fine": NeRF_sigma("fine", args, in_channels_xyz=93, in_channels_dir=27, encode_random=True).to(DEV),
              "decoder": style_net(args).to(DEV)}
    models["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in st_c.items()})
    models["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in st_f.items()})
    models["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in dst.items()})
    emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
    rays_np = synth.rays(R, seed=21, H=H, W=W)
    rng = np.random.default_rng(5)
    style_np = rng.uniform(0, 1, (1, 64, 32, 32)).astype(np.float32)
    target_np = rng.uniform(0, 1, (R, 3)).astype(np.float32)
    style = C(style_np).requires_grad_(True)
    res = render_rays_cross_ray(models, emb, C(rays_np), None, Nc, False, 0, 0, Ni, 1 << 20, False, args=args)
    assert res["feature_fine"].requires_grad and res["feature_coarse"].requires_grad

    def decode(feat):
        return models["decoder"](feat.t().reshape(1, 64, H, W), style).reshape(3, R).t()
    loss = ((decode(res["feature_coarse"]) - C(target_np)) ** 2).mean() + ((decode(res["feature_fine"]) - C(target_np)) ** 2).mean()
    loss.backward()

    # oracle: same graph with torch autograd on the CPU (depths: device linspace tables would differ by 1 ulp -> feed the HIP z's)
    wc = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st_c.items()}
    wf = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st_f.items()}
    wd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in dst.items() if not k.endswith(".f")}
    rays = torch.from_numpy(rays_np)
    with torch.no_grad():
        zc = (rays[:, 6:7] * (1 - torch.linspace(0, 1, Nc)) + rays[:, 7:8] * torch.linspace(0, 1, Nc))
        zc = (C(rays_np)[:, 6:7] * (1 - torch.linspace(0, 1, Nc, device=DEV)) + C(rays_np)[:, 7:8] * torch.linspace(0, 1, Nc, device=DEV)).cpu()
        zf = ops.sample_pdf_merge(zc.to(DEV).contiguous(), res["weights_coarse"].detach(), Ni, u=torch.linspace(0, 1, Ni, device=DEV)).cpu()
    demb = O.posenc(rays[:, 3:6], 4)
    raw_c = O._run_model(wc, rays, zc, demb, 1 << 20)
    _, fc, _ = O.composite(raw_c, zc)
    raw_f = O._run_model(wf, rays, zf, demb, 1 << 20)
    _, ff, _ = O.composite(raw_f, zf)
    style_ref = torch.from_numpy(style_np).requires_grad_(True)

    def decode_ref(feat):
        return O.crossray_decode(wd, O.feature_to_grid(feat, H, W), style_ref).reshape(3, R).t()
    loss_ref = ((decode_ref(fc) - torch.from_numpy(target_np)) ** 2).mean() + ((decode_ref(ff) - torch.from_numpy(target_np)) ** 2).mean()
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5

    def check(named_params, ref, what):
        for name, p in named_params:
            if name.endswith(".f"):
                continue
            g, r_ = p.grad.cpu(), ref[name].grad
            r_ = r_.reshape(g.shape)
            tol = 2e-4 * float(r_.abs().max()) + 1e-7
            assert float((g - r_).abs().max()) <= tol, "%s %s: %.3e > %.3e" % (what, name, float((g - r_).abs().max()), tol)
    check(models["coarse"].named_parameters(), wc, "coarse")
    check(models["fine"].named_parameters(), wf, "fine")
    check(models["decoder"].named_parameters(), wd, "decoder")
    gs, rs = style.grad.cpu(), style_ref.grad
    assert float((gs - rs).abs().max()) <= 2e-4 * float(rs.abs().max()) + 1e-7


# ------------------------------------------------------------------ next rows
@torch.no_grad()
def test_ray_generation_golden(golden):
    from crnerf_amd.datasets.ray_utils import generate_rays, get_ray_directions, get_rays
    g = golden("g8_rays")
    H, W = int(g["H"]), int(g["W"])
    dirs = get_ray_directions(H, W, g["K"])
    close(dirs, g["directions"], atol=1e-6)
    o, d = get_rays(dirs, torch.from_numpy(g["c2w"]).float())
    close(o, g["rays_o"], atol=0), close(d, g["rays_d"], atol=1e-6)
    close(generate_rays(H, W, g["K"], g["c2w"], 0.0, 5.0), g["rays"], atol=1e-6)
    assert float((d.norm(dim=-1) - 1).abs().max()) < 1e-6


@torch.no_grad()
def test_appearance_encoder_golden(golden):
    from crnerf_amd.models.linearStyleTransfer import encoder_sameoutputsize, style_net
    g = golden("g9_encoder")
    enc = encoder_sameoutputsize(64).to(DEV)
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == synth.ENCODER_SHAPES
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(int(g["seed"]), float(g["gain"])).items()})
    for tag in ("a", "b"):
        feat = enc(C(g["img_" + tag]))
        assert feat.shape == (1, 64, 32, 32)
        ref = torch.from_numpy(g["feat_" + tag])
        assert float((feat.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
    # feeds the decoder without a layout copy, and the trainable (torch) branch agrees with the HIP branch
    net = style_net(_Args()).to(DEV)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(31).items()})
    content = C(np.random.default_rng(0).uniform(0, 1, (1, 64, 24, 40)).astype(np.float32))
    assert net(content, feat).shape == (1, 3, 24, 40)
    with torch.enable_grad():
        feat_t = enc(C(g["img_b"]))
    assert feat_t.requires_grad and float((feat_t.detach() - feat).abs().max()) < 1e-4


@torch.no_grad()
def test_orchestration_mirror_video_frame_and_checkpoint_prefixes(tmp_path):
    """get_model / load_ckpt (Lightning prefix convention) / batched_inference / decode_image / render_frame."""
    from crnerf_amd import pipeline

    class H(_Args):
        N_emb_xyz, N_emb_dir, N_samples, N_importance, use_disp, encode_a, encode_random, N_a = 15, 4, 64, 64, False, True, True, 48
        img_wh = [24, 16]
    hp = H()
    models, emb = pipeline.get_model(hp, DEV), pipeline.get_embeddings(hp)
    assert set(models) == {"coarse", "fine", "decoder"} and models["fine"].encode_random and not models["coarse"].encode_random
    enc_a = pipeline.encoder_sameoutputsize(64).to(DEV)
    # a Lightning-style checkpoint: {'state_dict': {'<attribute>.<key>': tensor}}  (utils/__init__.py:67-88)
    sd = {}
    for prefix, st in (("nerf_coarse", synth.mlp_state(61, 3.0, 1.0)), ("nerf_fine", synth.mlp_state(62, 3.0, 1.0)),
                       ("decoder", synth.decoder_state(63)), ("enc_a", synth.encoder_state(64, 2.0))):
        sd.update({"%s.%s" % (prefix, k): torch.from_numpy(v) for k, v in st.items()})
    path = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": sd, "epoch": 3}, path)
    pipeline.load_ckpt(models["coarse"], path, model_name="nerf_coarse")
    pipeline.load_ckpt(models["fine"], path, model_name="nerf_fine")
    pipeline.load_ckpt(models["decoder"], path, model_name="decoder")
    pipeline.load_ckpt(enc_a, path, model_name="enc_a")
    assert torch.equal(models["fine"].state_dict()["static_rgb.0.bias"].cpu(), sd["nerf_fine.static_rgb.0.bias"])

    Hh, Ww = 16, 24
    focal = Ww / 2 / np.tan(np.pi / 6)
    K = np.array([[focal, 0, Ww / 2], [0, focal, Hh / 2], [0, 0, 1]])
    c2w = np.array([[1, 0, 0, 0.05], [0, -1, 0, 0.02], [0, 0, -1, 0.1]], dtype=np.float32)
    style_img = C(np.random.default_rng(3).uniform(0, 1, (1, 3, 40, 56)).astype(np.float32))
    img = pipeline.render_frame(models, emb, enc_a, style_img, Hh, Ww, K, c2w, hp, chunk=100)   # ragged chunks
    assert img.shape == (Hh, Ww, 3) and torch.isfinite(img).all() and 0 <= float(img.min()) and float(img.max()) <= 1
    # same frame through the oracle, end to end (rays, encoder, render at the HIP depths' own table, decode): PSNR check
    _, rays = O.generate_rays(Hh, Ww, K, c2w, 0.0, 5.0)
    wc, wf = O.to_torch(synth.mlp_state(61, 3.0, 1.0)), O.to_torch(synth.mlp_state(62, 3.0, 1.0))
    ref = O.render_rays(wc, wf, rays, 64, 64)
    a_ref = O.encoder_forward(O.to_torch(synth.encoder_state(64, 2.0)), style_img.cpu())
    ref_img = O.crossray_decode(O.to_torch(synth.decoder_state(63)), O.feature_to_grid(ref["feature_fine"], Hh, Ww), a_ref)
    ref_img = ref_img.reshape(3, -1).t().reshape(Hh, Ww, 3)
    target = ref_img + 0.05 * torch.from_numpy(np.random.default_rng(2).normal(size=ref_img.shape).astype(np.float32))
    assert abs(O.psnr(img.cpu(), target) - O.psnr(ref_img, target)) < 0.05


@pytest.mark.parametrize("H,W,hs", [(8, 8, 32), (13, 21, 32), (40, 24, 16)])
def test_decoder_backward_vs_autograd_oracle(H, W, hs):
    """crnerf_crossray_decode_backward_f32 against torch autograd through the oracle decoder: gradients w.r.t. the
    content grid, the style grid and all 22 parameter tensors."""
    st = synth.decoder_state(7, 2.0)
    rng = np.random.default_rng(H * W)
    content = rng.uniform(0, 1, (H * W, 64)).astype(np.float32)
    style = rng.uniform(0, 1, (hs * hs, 64)).astype(np.float32)
    d_rgb = rng.normal(size=(3, H * W)).astype(np.float32)
    names = [k for k in st if not k.endswith(".f")]
    w = {k: torch.from_numpy(st[k]).clone().requires_grad_(True) for k in names}
    xc, xs = torch.from_numpy(content).requires_grad_(True), torch.from_numpy(style).requires_grad_(True)
    rgb = O.crossray_decode(w, O.feature_to_grid(xc, H, W), O.feature_to_grid(xs, hs, hs)).reshape(3, H * W)
    (rgb * torch.from_numpy(d_rgb)).sum().backward()
    with torch.no_grad():
        dx, ds, grads = ops.crossray_decode_backward(C(content), C(style), [C(st[k]) for k in names], C(d_rgb))

    def check(got, ref, what):
        ref = ref.reshape(got.shape)
        tol = 2e-3 * float(ref.abs().max()) + 1e-8
        err = float((got.cpu() - ref).abs().max())
        assert err <= tol, "%s: %.3e > %.3e" % (what, err, tol)
    check(dx, xc.grad, "d_content")
    check(ds, xs.grad, "d_style")
    for k, gq in zip(names, grads):
        check(gq, w[k].grad, k)


@torch.no_grad()
def test_sharded_decode_phases_match_fused_decode_single_rank():
    """parallel.decode_sharded (RCCL, world_size 1) == the single-call decode; also an uneven split of the grid
    driven by hand through the three phases with the sums added on the host (what the all-reduces do)."""
    import os
    import socket
    import torch.distributed as dist
    from crnerf_amd.models.linearStyleTransfer import style_net
    from crnerf_amd.parallel import decode_sharded
    net = style_net(_Args()).to(DEV)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(9, 2.0).items()})
    rng = np.random.default_rng(1)
    feat = C(rng.uniform(0, 1, (37 * 29, 64)).astype(np.float32))
    style = C(rng.uniform(0, 1, (1, 64, 32, 32)).astype(np.float32))
    ref = net(feat.t().reshape(1, 64, 37, 29), style).reshape(3, -1)
    # three phases by hand over an uneven 3-way split (one part empty)
    sp = style.permute(0, 2, 3, 1).reshape(-1, 64).contiguous()
    w = net.decoder_tensors()
    parts = [feat[:400].contiguous(), feat[400:400].contiguous(), feat[400:].contiguous()]
    xs = [torch.zeros(1088, device=DEV) for _ in parts]
    for x, part in zip(xs, parts):
        ops.crossray_decode_sharded(part, sp, w, 0, x, float(feat.shape[0]))
    tot = sum(x[:64] for x in xs)
    for x, part in zip(xs, parts):
        x[:64] = tot
        ops.crossray_decode_sharded(part, sp, w, 1, x, float(feat.shape[0]))
    tot = sum(x[64:] for x in xs)
    outs = []
    for x, part in zip(xs, parts):
        x[64:] = tot
        o = ops.crossray_decode_sharded(part, sp, w, 2, x, float(feat.shape[0]))
        if part.shape[0]:
            outs.append(o)
    close(torch.cat(outs, 1), ref.cpu(), atol=2e-6)
    # through torch.distributed / RCCL with one rank
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        close(decode_sharded(net, feat, style, equal_shards=True), ref.cpu(), atol=2e-6)
        close(decode_sharded(net, feat, style), ref.cpu(), atol=2e-6)
    finally:
        dist.destroy_process_group()
