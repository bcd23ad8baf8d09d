"""Tuning tool (GPU box): the f32h2 renderer with bench weights vs ALL-ZERO weights (same instruction stream, no operand toggling in the matrix pipe):
how much of the kernel's time is the power governor's clock.  1,024 rays x (64+128)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rays = torch.from_numpy(synth.rays(R, seed=0)).to(dev)
zs, u = torch.linspace(0, 1, 64, device=dev), torch.linspace(0, 1, 128, device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for prec in ("f32h2", "f32x3", "f32"):
    for name, scale in (("bench weights", 1.0), ("all-zero weights", 0.0)):
        sts = [{k: torch.from_numpy(v).to(dev) * scale for k, v in synth.mlp_state(i, 2.0, 0.5).items()} for i in (1, 2)]
        pk = [ops.pack_mlp_weights(s, precision=prec) for s in sts]
        t = timed(lambda: ops.render_rays(pk[0], pk[1], rays, 64, 128, z_steps=zs, u=u, precision=prec))
        print("%-6s %-18s %7.3f ms per %d rays" % (prec, name, t, R))
