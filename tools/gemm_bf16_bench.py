"""GPU box: the mixed-precision training twins (per-layer bf16-MFMA GEMMs, csrc/mlp_gemm_bf16.hip) on 2^20 points: forward and
backward time, with CRNERF_GEMM_DBG switching parts off (1 = no epilogue, 2 = no operand loads, 4 = no MFMAs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
dev = "cuda:0"
P = 1 << 20
st = {k: torch.from_numpy(v).to(dev) for k, v in synth.mlp_state(3, 2.0, 0.5).items()}
x = torch.rand(P, 120, device=dev)
d_out = torch.randn(P, 65, device=dev)
packed, tensors = ops.pack_mlp_weights_mixed(st)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


with torch.no_grad():
    tf, (out, acts) = timed(lambda: ops.mlp_forward_train_mixed(packed, tensors, x))
    tb, _ = timed(lambda: ops.mlp_backward_mixed(packed, tensors, x, out, d_out, acts))
    pf = ops.pack_mlp_weights(st)
    t32, (o32, a32) = timed(lambda: ops.mlp_forward_train(pf, x))
    pt = ops.pack_mlp_weights_t(st)
    tb32, _ = timed(lambda: ops.mlp_backward(pt, x, o32, d_out, a32))
print("dbg %s: mixed forward %.2f ms, backward %.2f ms | fp32 twins forward %.2f ms, backward %.2f ms (2^20 points)"
      % (os.environ.get("CRNERF_GEMM_DBG", "0"), tf, tb, t32, tb32))
