"""Tuning tool (GPU box): the fp32 MLP training twins at P points with bench operands vs ALL-ZERO operands (weights, inputs, upstream gradient):
the same instruction streams without operand toggling.  If the zero run is much faster the kernel's clock is power-governed (the h2 / bf16
renderers: 20-30 %); if not, its time is its schedule (the fp32-MFMA renderer: 1 %)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = "cuda:0"


def timed(fn, n=3):
    keep = [fn(), fn()]
    del keep
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


for name, scale in (("bench operands", 1.0), ("all-zero operands", 0.0)):
    st = {k: torch.from_numpy(v).to(dev) * scale for k, v in synth.mlp_state(1, 2.0, 0.5).items()}
    packed, packed_t = ops.pack_mlp_weights(st), ops.pack_mlp_weights_t(st)
    x = (torch.rand(P, 120, device=dev) * 2 - 1) * scale
    d_out = torch.randn(P, 65, device=dev) * scale
    t_s, (out, acts) = timed(lambda: ops.mlp_forward_train(packed, x))
    t_b, _ = timed(lambda: ops.mlp_backward(packed_t, x, out, d_out, acts))
    t_b3, _ = timed(lambda: ops.mlp_backward(packed_t, x, out, d_out, acts, wgrad_bf16=2))
    print("%-18s P = %d: forward-with-save %6.2f ms | backward (fp32 dgrad + fp32 wgrad) %6.2f ms | backward (fp32 dgrad + bf16x3 wgrad) %6.2f ms"
          % (name, P, t_s, t_b, t_b3), flush=True)
