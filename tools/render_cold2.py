"""GPU box: repro hunt for the intermittent first-pass mismatch of render_rays_bf16 seen in tests/test_gpu_fullsize.py::config2 --
the test's own sequence (encoder, chunked renders, decode, one-launch render), repeated in one process; every render compared with
the first iteration's one-launch result and with each other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops, pipeline
dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = 640000
st_c = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
st_f = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(2, 2.0, 0.5).items()}
pc, pf = ops.pack_mlp_weights(st_c, precision=prec), ops.pack_mlp_weights(st_f, precision=prec)
rays = torch.from_numpy(synth.rays(R, seed=0, H=800, W=800)).to(dev)
z_steps, u = torch.linspace(0, 1, 64, device=dev), torch.linspace(0, 1, 128, device=dev)
enc = pipeline.encoder_sameoutputsize(64).to(dev)
photo = torch.rand(1, 3, 100, 100, generator=torch.Generator().manual_seed(0)).to(dev)
class Args: nerf_out_dim, img_wh = 64, [800, 800]
from crnerf_amd.models.linearStyleTransfer import style_net
net = style_net(Args()).to(dev)
def run(chunk):
    outs = [ops.render_rays(pc, pf, rays[i:i + chunk], 64, 128, z_steps=z_steps, u=u, precision=prec) for i in range(0, R, chunk)]
    return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
def cmp(name, x, y):
    n = 0
    for k in x:
        if not torch.equal(x[k], y[k]):
            d = (x[k] != y[k]).view(R, -1).any(1).nonzero().flatten()
            print("%s: %s differs in %d rays: %s ... %s; max |d| %.3e" % (name, k, d.numel(), d[:8].tolist(), d[-3:].tolist(), float((x[k] - y[k]).abs().max())), flush=True)
            n += 1
    return n
ref = None
bad = 0
with torch.no_grad():
    for it in range(iters):
        a_emb = enc(photo)
        a = run(32768)
        grid = a["feature_fine"].t().reshape(1, 64, 800, 800)
        rgb = net(grid, a_emb)
        b = run(R)
        c = run(50000)
        torch.cuda.synchronize()
        if ref is None:
            ref = {k: v.clone() for k, v in b.items()}
        bad += cmp("iter %d chunks32768 vs ref" % it, a, ref) + cmp("iter %d one vs ref" % it, b, ref) + cmp("iter %d chunks50000 vs ref" % it, c, ref)
print("%s: %d iterations, %d mismatching outputs" % (prec, iters, bad))
