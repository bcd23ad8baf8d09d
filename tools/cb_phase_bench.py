"""GPU box: composite_backward_kernel with phases switched off (CRNERF_CB_PHASES bit mask: 1 = row pass, 2 = alpha/T, 4 = reverse
scan, 8 = flat write), to see where its time goes.  Needs a library built with -DCRNERF_TIMING (the product build has no phase
switch).  usage: CRNERF_CB_PHASES=<mask> python tools/cb_phase_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crnerf_amd import ops
dev = "cuda:0"
R, N = 65536, 128
raw = torch.rand(R, N, 65, device=dev)
z = torch.sort(torch.rand(R, N, device=dev) * 4 + 0.5, dim=1)[0]
dF, dD = torch.randn(R, 64, device=dev), torch.randn(R, device=dev)
for _ in range(3):
    ops.composite_backward(raw, z, dF, dD)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for s, e in ev:
    s.record(); ops.composite_backward(raw, z, dF, dD); e.record()
torch.cuda.synchronize()
t = sorted(s.elapsed_time(e) for s, e in ev)
print("phases %s: %.1f us" % (os.environ.get("CRNERF_CB_PHASES", "15"), t[len(t) // 2] * 1e3))
