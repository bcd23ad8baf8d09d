"""Tuning tool (GPU box): NeRF_sigma forward at P points (default 2^20) -- the fp32-MFMA entry point, the bf16 one, and "f32x3" (fp32-accurate
products from three-piece bf16 splits, include/crnerf.h); with R given as second argument also the fused renderers at R rays x (64+128)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = os.environ.get("CRNERF_EXTRA_FLAGS")
if flags:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], stdout=subprocess.DEVNULL)
    print("flags:", flags)
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = "cuda:0"
st = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
x = torch.rand(P, 120, device=dev) * 2 - 1


def timed(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


flop = P * 1.233152e6
p32, pb, px = ops.pack_mlp_weights(st), ops.pack_mlp_weights(st, precision="bf16"), ops.pack_mlp_weights_x3(st)
ph = ops.pack_mlp_weights_h2(st)
for name, fn in (("fp32 MFMA (16x16x4 core)", lambda: ops.mlp_forward(p32, x)), ("bf16", lambda: ops.mlp_forward(pb, x, precision="bf16")),
                 ("f32x3", lambda: ops.mlp_forward_x3(px, x)), ("f32h2 (two fp16 pieces)", lambda: ops.mlp_forward_h2(ph, x))):
    t = timed(fn)
    print("mlp_forward %-26s %7.3f ms per %d points   %7.1f TFLOP/s (algorithmic fp32 FLOPs)" % (name, t, P, flop / t / 1e9))
if len(sys.argv) > 2 and hasattr(ops, "render_rays_x3"):
    R = int(sys.argv[2])
    st2 = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(2, 2.0, 0.5).items()}
    rays = torch.from_numpy(synth.rays(R, seed=0)).to(dev)
    p32b, pxb, phb = ops.pack_mlp_weights(st2), ops.pack_mlp_weights_x3(st2), ops.pack_mlp_weights_h2(st2)
    zs, u = torch.linspace(0, 1, 64, device=dev), torch.linspace(0, 1, 128, device=dev)
    for name, fn in (("fp32 MFMA", lambda: ops.render_rays(p32, p32b, rays, 64, 128, z_steps=zs, u=u)),
                     ("f32x3", lambda: ops.render_rays_x3(px, pxb, rays, 64, 128, z_steps=zs, u=u)),
                     ("f32h2", lambda: ops.render_rays(ph, phb, rays, 64, 128, z_steps=zs, u=u, precision="f32h2"))):
        t = timed(fn, 20)
        print("render_rays %-12s %7.3f ms per %d rays x (64+128)   %8.1f k rays/s" % (name, t, R, R / t))
