"""Tuning tool (GPU box): latency of the cross-ray decode of a small (32x32) and a large (800x800) grid, kernel by
kernel (torch profiler-free: rocprofv3 --kernel-trace --stats around this script gives the per-kernel split)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import crnerf_amd.synth as synth
from crnerf_amd.models.linearStyleTransfer import style_net
dev = torch.device("cuda:0")


class Args:
    nerf_out_dim, img_wh = 64, [32, 32]


net = style_net(Args()).to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
style = torch.rand(1024, 64, device=dev).view(1, 32, 32, 64).permute(0, 3, 1, 2)   # the encoder's pixel-major layout (zero-copy)
SIZES = [tuple(int(v) for v in a.split("x")) for a in os.environ.get("SIZES", "32x32,800x800").split(",")]
with torch.no_grad():
    for H, W in SIZES:
        feat = torch.rand(H * W, 64, device=dev)
        x = feat.t().reshape(1, 64, H, W)
        for _ in range(5):
            net(x, style)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            net(x, style)
        torch.cuda.synchronize()
        print("decode %4dx%-4d: %8.1f us" % (H, W, (time.perf_counter() - t0) / n * 1e6))
