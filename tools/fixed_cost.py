"""GPU box: fused renderer time vs rays per launch (1024 rays = one ray quad per CU = one pass of the persistent grid), to split
the per-launch fixed cost (launch, ramp, prologue, drain) from the per-iteration cost.  usage: python tools/fixed_cost.py [bf16|f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
C = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
pc, pf = ops.pack_mlp_weights(C(synth.mlp_state(1, 3.0, 1.0)), precision=prec), ops.pack_mlp_weights(C(synth.mlp_state(2, 3.0, 1.0)), precision=prec)
xs, ys = [], []
for it in (1, 2, 3, 4, 6, 8, 16):
    R = 1024 * it
    rays = torch.from_numpy(synth.rays(R)).to(dev)
    for _ in range(5):
        ops.render_rays(pc, pf, rays, 64, 128, precision=prec)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for s, e in ev:
        s.record(); ops.render_rays(pc, pf, rays, 64, 128, precision=prec); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)[15] * 1e3
    xs.append(it); ys.append(t)
    print("%s %6d rays (%2d iterations per workgroup): %8.1f us  (%.1f us per iteration)" % (prec, R, it, t, t / it))
b, a = np.polyfit(xs, ys, 1)
print("fit: %.1f us fixed + %.1f us per 1024 rays" % (a, b))
