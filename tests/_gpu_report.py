"""Developer tool (GPU box): prints max-abs / rel-L2 errors of every HIP entry point against the golden
vectors and the oracle, to calibrate the tolerances written in tests/test_gpu_parity.py."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crnerf_amd  # noqa: E402
import crnerf_amd.synth as synth  # noqa: E402
from crnerf_amd import ops  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402

G = lambda n: dict(np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")))  # noqa: E731
dev = torch.device("cuda:0")
C = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def err(name, got, want):
    got, want = got.detach().float().cpu(), torch.as_tensor(want).float()
    d = (got - want).abs()
    rel = float((got - want).norm() / (want.norm() + 1e-30))
    print("%-44s max_abs %.3e  rel_l2 %.3e  max|want| %.3e  nan %d" % (name, float(d.max()), rel, float(want.abs().max()), int(torch.isnan(got).sum())), flush=True)


def packed(state):
    return ops.pack_mlp_weights({k: C(v) for k, v in state.items()})


@torch.no_grad()
def main():
    print(torch.cuda.get_device_name(0), flush=True)
    g = G("g1_posenc")
    err("posenc xyz", ops.posenc(C(g["x"]), 15), g["xyz"])
    err("posenc dir", ops.posenc(C(g["x"]), 4), g["dir"])

    g = G("g2_mlp")
    for tag in ("default", "peaky"):
        pk = packed(synth.mlp_state(int(g["seed_" + tag]), float(g["gain_" + tag])))
        out = ops.mlp_forward(pk, C(g["x"]))
        err("mlp %s feat" % tag, out[:, :64], g["out_" + tag][:, :64])
        err("mlp %s sigma" % tag, out[:, 64], g["out_" + tag][:, 64])
        err("mlp %s sigma_only" % tag, ops.mlp_forward(pk, C(g["x"][:, :93]), sigma_only=True), g["sigma_" + tag])

    g = G("g3_composite")
    for tag, nstd in (("det", 0.0), ("noisy", 1.0)):
        for lvl, zk in (("coarse", "z_coarse"), ("fine", "z_fine_" + tag)):
            w, f, d = ops.composite(C(g["raw_" + lvl]), C(g[zk]), C(g["noise_" + lvl]), nstd)
            err("composite %s %s weights" % (tag, lvl), w, g["%s__weights_%s" % (tag, lvl)])
            err("composite %s %s feature" % (tag, lvl), f, g["%s__feature_%s" % (tag, lvl)])
            err("composite %s %s depth" % (tag, lvl), d, g["%s__depth_%s" % (tag, lvl)])
        zs = ops.sample_pdf_merge(C(g["z_coarse"]), C(g[tag + "__weights_coarse"]), 128)
        err("sample_pdf_merge %s z_fine" % tag, zs, g["z_fine_" + tag])

    g = G("g4_sample_pdf")
    wfull = np.zeros((64, 64), np.float32)
    wfull[:, 1:-1] = g["weights"]
    for ni in (64, 128):
        _, smp = ops.sample_pdf_merge(C(g["z_coarse"]), C(wfull), ni, return_samples=True)
        err("sample_pdf det %d" % ni, smp, g["det_%d" % ni])
    zs, smp = ops.sample_pdf_merge(C(g["z_coarse"]), C(wfull), 128, u=C(g["u_128"]), return_samples=True)
    err("sample_pdf rand 128", smp, g["rand_128"])
    want = np.sort(np.concatenate([g["z_coarse"], g["rand_128"]], -1), -1)
    err("merge (unsorted samples)", zs, want)

    g = G("g5_render")
    st_c = synth.mlp_state(int(g["seed_coarse"]), float(g["gain"]), float(g["sigma_bias"]))
    st_f = synth.mlp_state(int(g["seed_fine"]), float(g["gain"]), float(g["sigma_bias"]))
    pc, pf = packed(st_c), packed(st_f)
    rays = C(g["rays"])
    for tag, ni, disp in (("c64", 0, False), ("c64_f128", 128, False), ("c64_f128_disp", 128, True), ("c64_f64", 64, False)):
        out = ops.render_rays(pc, pf if ni else None, rays, 64, ni, use_disp=disp, want_z_fine=True)
        for k in sorted(out):
            key = "%s__%s" % (tag, k)
            if key in g:
                err("render %s %s" % (tag, k), out[k], g[key])
        if ni:
            # the fine pass re-evaluated by the ORACLE at the HIP path's own z_fine: separates kernel
            # error from the sensitivity of the 2^14-frequency embedding to 1-ulp depth differences
            wc, wf = O.to_torch(st_c), O.to_torch(st_f)
            rc = torch.from_numpy(g["rays"])
            zf = out["z_fine"].cpu()
            raw = O._run_model(wf, rc, zf, O.posenc(rc[:, 3:6], 4), 32768)
            w2, f2, d2 = O.composite(raw, zf)
            err("render %s feature_fine @same z" % tag, out["feature_fine"], f2)
            err("render %s weights_fine @same z" % tag, out["weights_fine"], w2)
            gz = torch.from_numpy(g["%s__weights_fine" % tag])
            ref = O.render_rays(wc, wf, rc, 64, ni, use_disp=disp)
            err("render %s z_fine" % tag, out["z_fine"], ref["z_fine"])

    g = G("g6_decoder")
    from crnerf_amd.models.linearStyleTransfer import style_net

    class A:
        nerf_out_dim, img_wh = 64, [40, 24]
    net = style_net(A()).to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(int(g["seed"])).items()})
    err("decoder rgb", net(C(g["content"]), C(g["style"])), g["rgb"])
    err("decoder rgb content-only", net(C(g["content"]), None, type="content"), g["rgb_content"])

    # quick timing of the headline config
    st = synth.mlp_state(1, 3.0, 1.0)
    pc, pf = packed(st), packed(synth.mlp_state(2, 3.0, 1.0))
    rays = C(synth.rays(1024, seed=0))
    for _ in range(2):
        ops.render_rays(pc, pf, rays, 64, 128)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        ops.render_rays(pc, pf, rays, 64, 128)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print("render 1024 x (64+128): %.3f ms  -> %.1f k rays/s, %.1f TFLOP/s" % (dt * 1e3, 1024 / dt / 1e3, 1024 * 315.69e6 / dt / 1e12))
    x = torch.randn(262144, 120, device=dev)
    ops.mlp_forward(pc, x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        ops.mlp_forward(pc, x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print("mlp_forward 262144 pts: %.3f ms -> %.1f TFLOP/s" % (dt * 1e3, 262144 * 1.233152e6 / dt / 1e12))


if __name__ == "__main__":
    main()
