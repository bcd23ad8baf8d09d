"""Tuning tool (GPU box): stand-alone NeRF_sigma forward, bf16 core vs fp32 core, on P embedded points.
CRNERF_EXTRA_FLAGS is honoured (rebuilds first), e.g. CRNERF_EXTRA_FLAGS=-DCRNERF_B_AHEAD=8."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("CRNERF_EXTRA_FLAGS"):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], stdout=subprocess.DEVNULL)
import torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
dev = torch.device("cuda:0")
P = int(os.environ.get("P", 262144))
st = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(1, 1.0).items()}
x = torch.rand(P, 120, device=dev) * 2 - 1
for prec in ("bf16", "f32"):
    pk = ops.pack_mlp_weights(st, precision=prec)
    for _ in range(3):
        ops.mlp_forward(pk, x, precision=prec)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        ops.mlp_forward(pk, x, precision=prec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%s: %.3f ms for %d points = %.1f TFLOP/s" % (prec, dt * 1e3, P, P * 1233152 / dt / 1e12))
if os.environ.get("CRNERF_EXTRA_FLAGS") and not os.environ.get("CRNERF_KEEP_BUILD"):
    env = dict(os.environ); env.pop("CRNERF_EXTRA_FLAGS")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], stdout=subprocess.DEVNULL, env=env)
