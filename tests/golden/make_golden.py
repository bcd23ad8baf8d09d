"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (/root/reference) in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [main] [rays] [encoder] [smooth]

The fixtures hold data only: seeded synthetic inputs and the reference's outputs on them.  Network
weights are NOT stored -- both sides regenerate them from the seed with crnerf_amd.synth (numpy
default_rng); a float64 checksum per weight set is stored to catch RNG drift.  The reference never
travels to the GPU box; these files do.  
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

# models/nerf_decoder_stylenerf.py:104 imports kornia.filters.filter2d at module import; it is only used
# by Blur.forward, unreachable at n_blocks == 0.  Stub the import in THIS process only.
_k, _kf = types.ModuleType("kornia"), types.ModuleType("kornia.filters")
_kf.filter2d = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("filter2d is not reachable at n_blocks=0"))
_k.filters = _kf
sys.modules.setdefault("kornia", _k)
sys.modules.setdefault("kornia.filters", _kf)

from models.linearStyleTransfer import encoder_sameoutputsize, style_net  # noqa: E402  (reference)
from models.nerf import NeRF_sigma, PosEmbedding  # noqa: E402  (reference)
from models.rendering import render_rays_cross_ray, sample_pdf  # noqa: E402  (reference)

import crnerf_amd.synth as synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
torch.set_num_threads(8)


class Args:
    nerf_out_dim = 64
    pertubeCord = False
    img_wh = [40, 24]


def ref_mlp(typ, state):
    m = NeRF_sigma(typ, Args(), in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, in_channels_a=48,
                   encode_random=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return m.eval()


def checksum(state):
    return float(sum(float(np.asarray(v, dtype=np.float64).sum()) for v in state.values()))


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("%-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


@torch.no_grad()
def main():
    rng = np.random.default_rng(20240928)
    emb_xyz, emb_dir = PosEmbedding(14, 15), PosEmbedding(3, 4)

    # ---- G1 positional embedding
    x = rng.uniform(-5, 5, size=(257, 3)).astype(np.float32)
    special = [0.0, 5.0, -5.0, 1e-3, 4.999999] + [math_pi / 2 ** k for k in range(0, 15)] + [-math_pi / 2 ** k for k in range(0, 15)]
    x[: len(special), 0] = np.array(special, dtype=np.float32)
    xt = torch.from_numpy(x)
    save("g1_posenc", x=x, xyz=emb_xyz(xt), dir=emb_dir(xt))

    # ---- G2 MLP, default-like and peaky weights
    pts = rng.uniform(-3, 3, size=(512, 3)).astype(np.float32)
    dirs = rng.normal(size=(512, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    xin = torch.cat([emb_xyz(torch.from_numpy(pts)), emb_dir(torch.from_numpy(dirs))], 1)
    g2 = {"x": xin}
    for tag, seed, gain in (("default", 11, 1.0), ("peaky", 12, 3.0)):
        st = synth.mlp_state(seed, gain)
        m = ref_mlp("fine", st)
        g2["out_" + tag] = m(xin)
        g2["sigma_" + tag] = m(xin[:, :93], sigma_only=True)
        g2["seed_" + tag], g2["gain_" + tag], g2["wsum_" + tag] = seed, gain, checksum(st)
    save("g2_mlp", **g2)

    # ---- G3 compositing, computed by the reference's own inference() (rendering.py:82-145): a canned
    # "model" returns prescribed raw [P,65] rows, torch.sort / torch.randn_like are wrapped for the
    # duration of the call to capture the depths and to inject known noise.
    R3, NC3, NI3 = 12, 64, 128
    raw_c = rng.uniform(0, 1, size=(R3, NC3, 65)).astype(np.float32)
    raw_f = rng.uniform(0, 1, size=(R3, NC3 + NI3, 65)).astype(np.float32)
    scale = rng.choice([0.1, 1, 10, 100], size=(R3, 1)).astype(np.float32)
    raw_c[..., 64] = np.abs(rng.normal(size=(R3, NC3))).astype(np.float32) * scale
    raw_f[..., 64] = np.abs(rng.normal(size=(R3, NC3 + NI3))).astype(np.float32) * scale
    raw_c[0, :, 64] = 0.0; raw_f[0, :, 64] = 0.0          # empty ray
    raw_c[1, :, 64] = 1e4; raw_f[1, :, 64] = 1e4          # transmittance underflows after sample 0
    raw_f[2, :100, 64] = 0.0                              # mass only in the back
    raw_f[3, :, 64] = -1.0                                # negative raw sigma -> relu clamps
    rays3 = synth.rays(R3, seed=9)
    rays3[:, 6] = rng.uniform(0.3, 1.0, size=R3); rays3[:, 7] = rng.uniform(3.0, 5.0, size=R3)
    rays3[4, 7] = rays3[4, 6]                             # near == far: every delta is 0

    class Canned(torch.nn.Module):
        def __init__(self, typ, rows):
            super().__init__()
            self.typ, self.encode_random, self.rows = typ, False, torch.from_numpy(rows).reshape(-1, 65)

        def forward(self, xx, output_random=True):
            assert xx.shape[0] == self.rows.shape[0]
            return self.rows

    g3 = {"raw_coarse": raw_c, "raw_fine": raw_f, "rays": rays3}
    noise_c = rng.normal(size=(R3, NC3)).astype(np.float32)
    noise_f = rng.normal(size=(R3, NC3 + NI3)).astype(np.float32)
    g3["noise_coarse"], g3["noise_fine"] = noise_c, noise_f
    real_sort, real_randn_like = torch.sort, torch.randn_like
    for tag, nstd in (("det", 0.0), ("noisy", 1.0)):
        captured = {}

        def sort_spy(t, *a, **k):
            out = real_sort(t, *a, **k)
            captured["cat"], captured["sorted"] = t.clone(), out[0].clone()
            return out

        def randn_like_canned(t, *a, **k):
            return torch.from_numpy(noise_c if t.shape[1] == NC3 else noise_f)

        torch.sort, torch.randn_like = sort_spy, randn_like_canned
        try:
            res = render_rays_cross_ray({"coarse": Canned("coarse", raw_c), "fine": Canned("fine", raw_f)},
                                        {"xyz": emb_xyz, "dir": emb_dir}, torch.from_numpy(rays3), None, NC3, False, 0, nstd,
                                        NI3, 1 << 30, False, args=Args())
        finally:
            torch.sort, torch.randn_like = real_sort, real_randn_like
        g3["z_coarse"] = captured["cat"][:, :NC3]
        g3["z_fine_" + tag] = captured["sorted"]
        for k, v in res.items():
            g3["%s__%s" % (tag, k)] = v
    save("g3_composite", **g3)

    # ---- G4 sample_pdf (reference function, det=True and explicit-u via monkeypatched torch.rand)
    R4, M = 64, 62
    zc = np.sort(rng.uniform(0.3, 5.0, size=(R4, M + 2)).astype(np.float32), axis=-1)
    bins = 0.5 * (zc[:, :-1] + zc[:, 1:])
    w4 = (rng.uniform(0, 1, size=(R4, M)) ** 4).astype(np.float32)
    w4[0] = 0.0                                # all-zero -> uniform
    w4[1] = 0.0; w4[1, 17] = 1.0               # one-hot
    w4[2] = 0.0; w4[2, 0] = 1.0                # first bin only
    w4[3] = 0.0; w4[3, -1] = 1.0               # last bin only
    w4[4] = 1.0                                # exactly uniform
    w4[5, ::2] = 0.0                           # alternating empty bins
    g4 = {"z_coarse": zc, "bins": bins, "weights": w4}
    for ni in (64, 128):
        g4["det_%d" % ni] = sample_pdf(torch.from_numpy(bins), torch.from_numpy(w4), ni, det=True)
    u = rng.uniform(0, 1, size=(R4, 128)).astype(np.float32)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(u)
    try:
        g4["rand_128"] = sample_pdf(torch.from_numpy(bins), torch.from_numpy(w4), 128, det=False)
    finally:
        torch.rand = real_rand
    g4["u_128"] = u
    save("g4_sample_pdf", **g4)

    # ---- G5 end-to-end render_rays_cross_ray + G7 chunk invariance
    st_c, st_f = synth.mlp_state(21, 3.0, sigma_bias=1.0), synth.mlp_state(22, 3.0, sigma_bias=1.0)
    models = {"coarse": ref_mlp("coarse", st_c), "fine": ref_mlp("fine", st_f)}
    embeddings = {"xyz": emb_xyz, "dir": emb_dir}
    rays = torch.from_numpy(synth.rays(64, seed=5))
    g5 = {"rays": rays, "seed_coarse": 21, "seed_fine": 22, "gain": 3.0, "sigma_bias": 1.0,
          "wsum_coarse": checksum(st_c), "wsum_fine": checksum(st_f)}
    ts = torch.zeros(64, dtype=torch.long)
    # the linspace tables the reference builds on its device (rendering.py:160, :27); ATen's CPU linspace
    # differs in the last bit between vector widths, so the fixture carries the exact tables used
    g5["z_steps_64"] = torch.linspace(0, 1, 64)
    g5["u_steps_64"], g5["u_steps_128"] = torch.linspace(0, 1, 64), torch.linspace(0, 1, 128)
    real_sort = torch.sort
    for tag, ni, disp in (("c64", 0, False), ("c64_f128", 128, False), ("c64_f128_disp", 128, True), ("c64_f64", 64, False)):
        captured = {}

        def sort_spy(t, *a, **k):
            out = real_sort(t, *a, **k)
            captured["z_fine"] = out[0].clone()
            return out

        torch.sort = sort_spy
        try:
            res = render_rays_cross_ray(models, embeddings, rays, ts, 64, disp, 0, 0, ni, 32768, False, test_time=True, args=Args())
        finally:
            torch.sort = real_sort
        if ni > 0:
            assert res["feature_fine_random"] is res["feature_fine"]
            g5["%s__z_fine" % tag] = captured["z_fine"]
        for k, v in res.items():
            if k != "feature_fine_random":
                g5["%s__%s" % (tag, k)] = v
        res2 = render_rays_cross_ray(models, embeddings, rays, ts, 64, disp, 0, 0, ni, 2048, False, test_time=True, args=Args())
        for k in res:   # G7: chunk must not matter (bit-identical on the reference CPU path)
            assert torch.equal(res[k], res2[k]), (tag, k)
    # view_dir override (rendering.py:155)
    vd = torch.from_numpy(np.roll(synth.rays(64, seed=5)[:, 3:6], 7, axis=0).copy())
    res = render_rays_cross_ray(models, embeddings, rays, ts, 64, False, 0, 0, 128, 32768, False, test_time=True, args=Args(), view_dir=vd)
    g5["view_dir"] = vd
    g5["viewdir__feature_fine"] = res["feature_fine"]
    save("g5_render", **g5)

    # ---- G6 cross-ray decoder
    dst = synth.decoder_state(31, 1.0)
    net = style_net(Args()).eval()
    assert net.decoder.n_blocks == 0
    net.load_state_dict({k: torch.from_numpy(v) for k, v in dst.items()})
    content = torch.from_numpy(rng.uniform(0, 1, size=(1, 64, 24, 40)).astype(np.float32))
    style = torch.from_numpy(rng.uniform(0, 1, size=(1, 64, 32, 32)).astype(np.float32))
    save("g6_decoder", content=content, style=style, seed=31, wsum=checksum(dst),
         rgb=net(content.clone(), style.clone()), rgb_content=net(content.clone(), None, type="content"))


def ray_goldens():
    """G8 ray generation: the reference's datasets/ray_utils.py loaded straight from its file (the package
    __init__ drags in torchvision/pandas readers).  kornia.create_meshgrid is absent here; it is a pixel
    index grid ([...,0] = x = column, [...,1] = y = row), provided by the stub below."""
    import importlib.util

    def create_meshgrid(H, W, normalized_coordinates=True):
        assert not normalized_coordinates
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        return torch.stack([xs, ys], -1)[None]
    sys.modules["kornia"].create_meshgrid = create_meshgrid
    spec = importlib.util.spec_from_file_location("ref_ray_utils", "/root/reference/datasets/ray_utils.py")
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    H, W = 24, 40
    focal = W / 2 / np.tan(np.pi / 6)                       # appearance_modification_video.py:183-189
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
    c2w = np.array([[0.99702646, 0.00170214, -0.07704115, 0.03552477],   # pose_init, appearance_modification_video.py:123-125
                    [0.01082206, -0.99294089, 0.11811554, 0.02343685],
                    [-0.07629626, -0.11859807, -0.99000676, 0.12162088]])
    dirs = ru.get_ray_directions(H, W, K)
    o, d = ru.get_rays(dirs, torch.FloatTensor(c2w))
    rays = torch.cat([o, d, 0 * torch.ones_like(o[:, :1]), 5 * torch.ones_like(o[:, :1])], 1)   # PhototourismDataset.py:17-22
    save("g8_rays", H=H, W=W, K=K, c2w=c2w, directions=dirs, rays_o=o, rays_d=d, rays=rays)


@torch.no_grad()
def encoder_goldens():
    """G9 appearance encoder: reference encoder_sameoutputsize on a small photo-like input."""
    rng = np.random.default_rng(77)
    st = synth.encoder_state(51, 2.0)
    enc = encoder_sameoutputsize(64).eval()
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    out = {"seed": 51, "gain": 2.0, "wsum": checksum(st)}
    for tag, (h, w) in (("a", (52, 76)), ("b", (130, 128))):       # 13x19 and 32x32 maps before the adaptive pool
        img = torch.from_numpy(rng.uniform(0, 1, (1, 3, h, w)).astype(np.float32))
        out["img_" + tag] = img
        out["feat_" + tag] = enc(img.clone())
    save("g9_encoder", **out)


@torch.no_grad()
def smooth_goldens():
    """G14 end-to-end render on SMOOTH-DENSITY nets + a high-contrast decode (round-2 verdict item 1).

    G5's gain-3 nets sit where the reference itself is ill-conditioned (2^14 embedding gain x sample_pdf's denom<eps switch,
    rendering.py:41-45), so G5's end-to-end fine outputs can only be held loosely.  Here the nets are band-limited
    (crnerf_amd.synth.mlp_state(band_limit=4)) and every coarse bin carries mass, so the reference is well-conditioned and
    SURVEY 8d's stated tolerances (pixels 2e-5, features rel-L2 1e-5, fine z 1e-5*far) apply END TO END against the
    reference's outputs.  The fixture also records the reference's OWN sensitivity: its outputs when weights_coarse is
    perturbed by <= 1 ulp before sample_pdf (a wrapper around the reference's sample_pdf, this process only) -- the floor no
    implementation with a different summation order can beat.  A second variant uses the judge's "gain-1 nets with a sigma
    bias".  Decoded pixels: the reference's style_net on feature_fine with the high-contrast decoder of synth.decoder_state."""
    import models.rendering as ref_rendering
    emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
    H = W = 8
    rays = torch.from_numpy(synth.rays(64, seed=14, H=H, W=W))
    ts = torch.zeros(64, dtype=torch.long)
    out = {"rays": rays, "H": H, "W": W, "z_steps_64": torch.linspace(0, 1, 64), "u_steps_128": torch.linspace(0, 1, 128),
           "contrast": 4000.0, "seed_decoder": 43}
    dst = synth.decoder_state(43, 1.0, contrast=4000.0)
    out["wsum_decoder"] = checksum(dst)

    class A(Args):
        img_wh = [W, H]
    net = style_net(A()).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in dst.items()})
    style = torch.from_numpy(np.random.default_rng(1414).uniform(0, 1, (1, 64, 32, 32)).astype(np.float32))
    out["style"] = style
    real_sort, real_pdf = torch.sort, ref_rendering.sample_pdf
    for net_tag, kw in (("band", dict(gain=2.45, sigma_bias=-1.0, band_limit=4)), ("gain1", dict(gain=1.0, sigma_bias=-0.5))):
        st_c, st_f = synth.mlp_state(41, **kw), synth.mlp_state(42, **kw)
        models = {"coarse": ref_mlp("coarse", st_c), "fine": ref_mlp("fine", st_f)}
        out[net_tag + "__wsum_coarse"], out[net_tag + "__wsum_fine"] = checksum(st_c), checksum(st_f)
        for tag, disp in (("c64_f128", False), ("c64_f128_disp", True)):
            cap = {}

            def sort_spy(t, *a, **k):
                r = real_sort(t, *a, **k)
                cap["z_fine"] = r[0].clone()
                return r
            torch.sort = sort_spy
            try:
                res = render_rays_cross_ray(models, emb, rays, ts, 64, disp, 0, 0, 128, 32768, False, test_time=True, args=Args())
            finally:
                torch.sort = real_sort
            key = "%s__%s__" % (net_tag, tag)
            out[key + "z_fine"] = cap["z_fine"]
            for k, v in res.items():
                if k != "feature_fine_random":
                    out[key + k] = v
            grid = res["feature_fine"].t().reshape(1, 64, H, W)                       # eval.py:291-292
            out[key + "rgb"] = net(grid.clone(), style.clone()).reshape(3, H * W).t()   # eval.py:293-294 -> [HW,3]
            # the reference's own conditioning: <= 1 ulp on weights_coarse ahead of sample_pdf
            g = torch.Generator().manual_seed(7)

            def pdf_perturbed(bins, weights, n, det=False, eps=1e-5):
                return real_pdf(bins, weights * (1 + (torch.rand(weights.shape, generator=g) - 0.5) * 2.4e-7), n, det=det, eps=eps)
            ref_rendering.sample_pdf = pdf_perturbed
            torch.sort = sort_spy
            try:
                alt = render_rays_cross_ray(models, emb, rays, ts, 64, disp, 0, 0, 128, 32768, False, test_time=True, args=Args())
            finally:
                ref_rendering.sample_pdf, torch.sort = real_pdf, real_sort
            rgb_alt = net(alt["feature_fine"].t().reshape(1, 64, H, W).clone(), style.clone()).reshape(3, H * W).t()
            sens = {"z_fine": float((cap["z_fine"] - out[key + "z_fine"]).abs().max()),
                    "feature_fine_maxabs": float((alt["feature_fine"] - res["feature_fine"]).abs().max()),
                    "feature_fine_rel_l2": float((alt["feature_fine"] - res["feature_fine"]).norm() / res["feature_fine"].norm()),
                    "weights_fine_maxabs": float((alt["weights_fine"] - res["weights_fine"]).abs().max()),
                    "depth_fine_maxabs": float((alt["depth_fine"] - res["depth_fine"]).abs().max()),
                    "rgb_maxabs": float((rgb_alt - out[key + "rgb"]).abs().max())}
            for k, v in sens.items():
                out[key + "ref_1ulp_sensitivity__" + k] = v
            print(net_tag, tag, "rgb range [%.3f, %.3f]" % (float(out[key + "rgb"].min()), float(out[key + "rgb"].max())),
                  "reference 1-ulp self-sensitivity:", {k: "%.2e" % v for k, v in sens.items()})
    save("g14_render_smooth", **out)


if __name__ == "__main__":
    import math
    math_pi = np.float32(math.pi)
    which = sys.argv[1:] or ["main", "rays", "encoder", "smooth"]
    if "main" in which:
        main()
    if "rays" in which:
        ray_goldens()
    if "encoder" in which:
        encoder_goldens()
    if "smooth" in which:
        smooth_goldens()
