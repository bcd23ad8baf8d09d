"""bf16 path vs its oracle restatement and vs the fp32 path: error statistics + timing (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
from oracle import cpu_ref as O

DEV = "cuda:0"
C = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def stats(name, got, want):
    d = (got.double().cpu() - want.double()).abs()
    print("  %-28s max %.3e  mean %.3e  rel-L2 %.3e" % (name, float(d.max()), float(d.mean()), float(d.norm() / want.double().norm())))


def main():
    torch.manual_seed(0)
    for gain in (1.0, 3.0):
        st = synth.mlp_state(7, gain)
        w = O.to_torch(st)
        n = 1000
        pts = torch.rand(n, 3) * 6 - 3
        dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=1)
        x = torch.cat((O.posenc(pts, 15), O.posenc(dirs, 4)), 1)
        pb = ops.pack_mlp_weights({k: C(v) for k, v in st.items()}, precision="bf16")
        pf = ops.pack_mlp_weights({k: C(v) for k, v in st.items()})
        got = ops.mlp_forward(pb, x.to(DEV), precision="bf16")
        got32 = ops.mlp_forward(pf, x.to(DEV))
        ob, o32 = O.mlp_forward_bf16(w, x), O.mlp_forward(w, x)
        print("MLP gain %.0f, %d points" % (gain, n))
        stats("bf16 HIP vs bf16 oracle feat", got[:, :64], ob[:, :64])
        stats("bf16 HIP vs bf16 oracle sigma", got[:, 64], ob[:, 64])
        stats("bf16 oracle vs fp32 oracle feat", ob[:, :64], o32[:, :64])
        stats("bf16 oracle vs fp32 oracle sigma", ob[:, 64], o32[:, 64])
        stats("fp32 HIP vs fp32 oracle feat", got32[:, :64], o32[:, :64])
    # fused render
    R = 1024
    rays = C(synth.rays(R))
    sc, sf = synth.mlp_state(11, 3.0), synth.mlp_state(12, 3.0)
    pk = lambda s, p: ops.pack_mlp_weights({k: C(v) for k, v in s.items()}, precision=p)
    for prec in ("f32", "bf16"):
        pc, pf_ = pk(sc, prec), pk(sf, prec)
        out = ops.render_rays(pc, pf_, rays, 64, 128, precision=prec, want_z_fine=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.render_rays(pc, pf_, rays, 64, 128, precision=prec)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("render %s: %.3f ms / 1024 rays = %.0f rays/s" % (prec, dt * 1e3, R / dt))
        if prec == "f32":
            ref = out
        else:
            orc = O.render_rays(O.to_torch(sc), O.to_torch(sf), rays.cpu()[:128], 64, 128, precision="bf16", z_fine=out["z_fine"].cpu()[:128])
            print("bf16 HIP vs bf16 oracle (fine pass at the HIP depths), 128 rays")
            for k in ("weights_coarse", "feature_coarse", "depth_coarse", "weights_fine", "feature_fine", "depth_fine"):
                stats(k, out[k][:128], orc[k])
            print("bf16 HIP vs fp32 HIP, 1024 rays")
            for k in ("weights_coarse", "feature_coarse", "depth_coarse", "feature_fine", "depth_fine"):
                stats(k, out[k], ref[k].cpu())
            s = out["weights_fine"].sum(-1)
            print("  sum weights_fine in [%.7f, %.7f]; z sorted %s" % (float(s.min()), float(s.max()), bool((out["z_fine"][:, 1:] >= out["z_fine"][:, :-1]).all())))


if __name__ == "__main__":
    main()
