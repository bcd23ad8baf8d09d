"""Tuning tool (GPU box): times one training-style step (render in grad mode + decode + MSE + backward)
on the drop-in modules: 1024 rays (32x32 grid), 64+64 samples, perturb=1, noise_std=1 (command/train.sh:19-24)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crnerf_amd.synth as synth
from crnerf_amd.models.linearStyleTransfer import style_net
from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
from crnerf_amd.models.rendering import render_rays_cross_ray

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = W = int(R ** 0.5)
dev = "cuda:0"


class A:
    nerf_out_dim, img_wh, pertubeCord = 64, [W, H], False


m = {"coarse": NeRF_sigma("coarse", A(), in_channels_xyz=93, in_channels_dir=27).to(dev),
     "fine": NeRF_sigma("fine", A(), in_channels_xyz=93, in_channels_dir=27, encode_random=True).to(dev),
     "decoder": style_net(A()).to(dev)}
for k, s in (("coarse", synth.mlp_state(1, 2.0, 0.5)), ("fine", synth.mlp_state(2, 2.0, 0.5)), ("decoder", synth.decoder_state(3))):
    m[k].load_state_dict({n: torch.from_numpy(v) for n, v in s.items()})
emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
rays = torch.from_numpy(synth.rays(R, H=H, W=W)).to(dev)
style = torch.rand(1, 64, 32, 32, device=dev)
target = torch.rand(R, 3, device=dev)
params = [p for mod in m.values() for p in mod.parameters()]
opt = torch.optim.Adam(params, lr=5e-4, fused=os.environ.get("CRNERF_BENCH_ADAM", "fused") == "fused")   # the reference: Adam(lr, eps=1e-8); fused = one multi-tensor launch
if os.environ.get("CRNERF_TRAIN_BF16"):
    from crnerf_amd import autograd as _ag
    _ag.set_training_precision("bf16")


class HPL:
    maskrs_max, maskrs_min, maskrs_k, maskrd, weightKL, weightRecA, weightcontent, mse_on_appearance = 5e-2, 6e-3, 1e-3, 0.0, 1e-5, 1e-3, 1e-4, False


from crnerf_amd.losses import loss_dict
crit = loss_dict["crnerf"](HPL, coef=1)


def step():
    opt.zero_grad(set_to_none=True)
    res = render_rays_cross_ray(m, emb, rays, None, 64, False, 1.0, 1.0, 64, 1 << 22, False, args=A())
    dec = lambda f: m["decoder"](f.t().reshape(1, 64, H, W), style).reshape(3, R).t()
    # CRNeRFLoss's colour terms c_l + f_l (losses.py:61-72) through the fused HIP loss (crnerf_amd.losses)
    loss_d, _ = crit({"rgb_coarse": dec(res["feature_coarse"]), "rgb_fine": dec(res["feature_fine"])}, target, HPL, 0)
    loss = sum(loss_d.values())
    loss.backward()
    opt.step()
    return float(loss.detach())


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
pts = R * (64 + 128)
print("train step %d rays x (64+64): %.2f ms -> %.1f k rays/s; fwd+bwd MLP FLOPs %.1f TFLOP/s; loss %.4f; peak mem %.2f GB"
      % (R, dt * 1e3, R / dt / 1e3, 3 * pts * 1.233152e6 / dt / 1e12, l, torch.cuda.max_memory_allocated() / 2 ** 30))
