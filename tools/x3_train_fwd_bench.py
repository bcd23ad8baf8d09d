"""Tuning tool (GPU box): the f32x3 renderer, its training twin and the fp32 training twin on one ray chunk (5,460 rays x (64+64)).  Variant libraries from tools/variants.py are selected with
CRNERF_LIB_PATH (profiles/r5/row_store_experiments.txt)."""
import os, subprocess, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
dev = "cuda:0"
R = 5460
st = [{k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(s, 2.0, 0.5).items()} for s in (1, 2)]
px = [ops.pack_mlp_weights_x3(s) for s in st]
p32 = [ops.pack_mlp_weights(s) for s in st]
ph = [ops.pack_mlp_weights_h2(s) for s in st]
pa = [ops.pack_mlp_weights(s, precision="auto") for s in st]
rays = torch.from_numpy(synth.rays(R, seed=0)).to(dev)
rng = np.random.default_rng(0)
z = torch.from_numpy(np.sort(rng.uniform(2, 6, (R, 64)).astype(np.float32), -1)).to(dev)
u = torch.from_numpy(rng.uniform(0, 1, (R, 64)).astype(np.float32)).to(dev)
def timed(fn, n=5):
    k = [fn(), fn()]; del k
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
pts = R * 192
for name, fn in (("f32x3 inference", lambda: ops.render_rays(px[0], px[1], rays, 64, 64, z_coarse=z, u=u, precision="f32x3", want_z_fine=True)),
                 ("f32x3 training twin", lambda: ops.render_rays(px[0], px[1], rays, 64, 64, z_coarse=z, u=u, precision="f32x3", train=True)),
                 ("f32h2 inference", lambda: ops.render_rays(ph[0], ph[1], rays, 64, 64, z_coarse=z, u=u, precision="f32h2", want_z_fine=True)),
                 ("f32h2 training twin", lambda: ops.render_rays(ph[0], ph[1], rays, 64, 64, z_coarse=z, u=u, precision="f32h2", train=True)),
                 ("auto training twin", lambda: ops.render_rays(pa[0], pa[1], rays, 64, 64, z_coarse=z, u=u, precision="auto", train=True)),
                 ("fp32 training twin", lambda: ops.render_rays(p32[0], p32[1], rays, 64, 64, z_coarse=z, u=u, train=True))):
    t = timed(fn)
    print("%-22s %7.3f ms  (%.3f ms per 2^20 points)" % (name, t, t / pts * 2 ** 20))
