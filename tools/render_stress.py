"""GPU box: determinism stress of the fused renderers under dynamic quad scheduling -- the same rays N times, every output bit-compared
with the first run; mismatching rays are listed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
st_c = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
st_f = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(2, 2.0, 0.5).items()}
pc, pf = ops.pack_mlp_weights(st_c, precision=prec), ops.pack_mlp_weights(st_f, precision=prec)
rays = torch.from_numpy(synth.rays(R, seed=0, H=200, W=R // 200)).to(dev)
z_steps, u = torch.linspace(0, 1, 64, device=dev), torch.linspace(0, 1, 128, device=dev)
ref = None
bad = 0
for i in range(reps):
    # interleave other sizes so that slots rotate and launches of different shapes follow each other
    ops.render_rays(pc, pf, rays[: 32768 if i % 2 else 40000], 64, 128, z_steps=z_steps, u=u, precision=prec)
    out = ops.render_rays(pc, pf, rays, 64, 128, z_steps=z_steps, u=u, precision=prec, want_z_fine=True)
    torch.cuda.synchronize()
    if ref is None:
        ref = {k: v.clone() for k, v in out.items()}
        continue
    for k, v in out.items():
        if not torch.equal(v, ref[k]):
            d = (v != ref[k])
            rows = d.view(d.shape[0], -1).any(1).nonzero().flatten()
            print("run %d: %s differs in %d rays: first %s (quads %s)" % (i, k, rows.numel(), rows[:8].tolist(), sorted(set((rows[:64] // 4).tolist()))[:10]), flush=True)
            bad += 1
print("%s R=%d: %d runs, %d mismatching outputs" % (prec, R, reps, bad))
