"""Tuning tool (GPU box): the forward of the mixed-precision training mode on one ray chunk (default 5,460 rays x (64+64) = 2^20 fine + 2^19
coarse points): bf16 inference renderer, its training twin (crnerf_render_rays_train_bf16), and the un-fused forward it replaces
(embed_points -> per-layer GEMM twins -> compositing), plus the backward from either buffer."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = os.environ.get("CRNERF_EXTRA_FLAGS")      # experiment macros (CRNERF_EXP_SAVE=n): rebuild first; FWD_ONLY=1 skips the rest
if flags:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], stdout=subprocess.DEVNULL)
    print("flags:", flags)
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops, autograd as AG

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5460
Nc, Ni = 64, 64
dev = "cuda:0"
C = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
st = [{k: C(v) for k, v in synth.mlp_state(s, 2.0, 0.5).items()} for s in (1, 2)]
pk = [ops.pack_mlp_weights(s, precision="bf16") for s in st]
rays = C(synth.rays(R, seed=0))
rng = np.random.default_rng(0)
z = C(np.sort(rng.uniform(2, 6, (R, Nc)).astype(np.float32), -1))
u = C(rng.uniform(0, 1, (R, Ni)).astype(np.float32))
kw = dict(z_coarse=z, u=u, noise_std=0.0, precision="bf16")


def timed(fn, n=5):
    keep = [fn(), fn()]
    del keep
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


pts = R * (Nc + Nc + Ni)
t_inf, _ = timed(lambda: ops.render_rays(pk[0], pk[1], rays, Nc, Ni, want_z_fine=True, **kw))
t_trn, trn = timed(lambda: ops.render_rays(pk[0], pk[1], rays, Nc, Ni, train=True, **kw))
print("points per step: %d (%.2f x 2^20)" % (pts, pts / 2 ** 20))
print("bf16 inference renderer          %7.3f ms  (%.3f ms per 2^20 points)" % (t_inf, t_inf / pts * 2 ** 20))
print("bf16 training twin (fused save)  %7.3f ms  (%.3f ms per 2^20 points; %.2f TB/s of saved state)" % (t_trn, t_trn / pts * 2 ** 20, pts * (5664 + 260) / t_trn / 1e9))
if os.environ.get("FWD_ONLY"):
    sys.exit(0)


def unfused():
    outs = []
    for m, zz in ((0, z), (1, trn["z_fine"])):
        x = AG._embed_points(rays, zz, None)
        packed, tensors = ops.pack_mlp_weights_mixed(st[m])
        raw, acts = ops.mlp_forward_train_mixed(packed, tensors, x)
        outs.append((ops.composite(raw.view(R, -1, 65), zz), acts))
    return outs


t_un, _ = timed(unfused)
print("un-fused mixed forward           %7.3f ms  (%.3f ms per 2^20 points)" % (t_un, t_un / pts * 2 ** 20))
d = [torch.randn(R * Nc, 65, device=dev), torch.randn(R * (Nc + Ni), 65, device=dev)]
packed = [ops.pack_mlp_weights_mixed(s) for s in st]


def bwd():
    return [ops.mlp_backward_mixed(packed[m][0], packed[m][1], None, trn["raw_" + t].view(-1, 65), d[m], trn["acts_" + t], fused_acts=True)
            for m, t in ((0, "coarse"), (1, "fine"))]


t_b, _ = timed(bwd)
print("mixed backward (fused buffers)   %7.3f ms  (%.3f ms per 2^20 points)" % (t_b, t_b / pts * 2 ** 20))
