"""GPU box: cold-process chunk-invariance check of the fused renderer (what tests/test_gpu_fullsize.py::config2 does first):
the same rays in 32,768-ray chunks, as one launch and in 50,000-ray chunks, bit-compared; mismatching rays are listed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
R = 640000
st_c = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
st_f = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(2, 2.0, 0.5).items()}
pc, pf = ops.pack_mlp_weights(st_c, precision=prec), ops.pack_mlp_weights(st_f, precision=prec)
rays = torch.from_numpy(synth.rays(R, seed=0, H=800, W=800)).to(dev)
z_steps, u = torch.linspace(0, 1, 64, device=dev), torch.linspace(0, 1, 128, device=dev)
def run(chunk):
    outs = [ops.render_rays(pc, pf, rays[i:i + chunk], 64, 128, z_steps=z_steps, u=u, precision=prec) for i in range(0, R, chunk)]
    return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
a, b, c = run(32768), run(R), run(50000)
torch.cuda.synchronize()
bad = 0
for name, x, y in (("32768 vs one", a, b), ("32768 vs 50000", a, c), ("one vs 50000", b, c)):
    for k in x:
        if not torch.equal(x[k], y[k]):
            d = (x[k] != y[k]).view(R, -1).any(1).nonzero().flatten()
            print("%s: %s differs in %d rays: %s ... %s; max |d| %.3e" % (name, k, d.numel(), d[:12].tolist(), d[-4:].tolist(), float((x[k] - y[k]).abs().max())), flush=True)
            bad += 1
print("%s cold check: %d mismatching outputs" % (prec, bad))
