"""GPU box: the HBM-bound stand-alone entries against their algorithmic bytes (SURVEY 8d / DESIGN 3.3-3.4):
composite forward / backward, positional encoding, sample_pdf + merge, the decoder's channel sums and per-pixel apply,
and the whole decode at a full-image grid.  Prints achieved GB/s = algorithmic bytes / kernel time (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import crnerf_amd.synth as synth
from crnerf_amd import ops
from crnerf_amd.models.linearStyleTransfer import style_net

dev = "cuda:0"
PEAK = 8000.0


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2] * 1e-3


def report(name, nbytes, t):
    print("%-44s %8.1f MB  %8.1f us  %7.1f GB/s  %5.1f %% of 8 TB/s" % (name, nbytes / 1e6, t * 1e6, nbytes / t / 1e9, 100 * nbytes / t / 1e9 / PEAK))


R, N = 65536, 128
raw = torch.rand(R, N, 65, device=dev)
z = torch.sort(torch.rand(R, N, device=dev) * 4 + 0.5, dim=1)[0]
report("composite forward  (R=65536, N=128)", 4 * (67 * N + 65) * R, timed(lambda: ops.composite(raw, z)))
dF, dD = torch.randn(R, 64, device=dev), torch.randn(R, device=dev)
report("composite backward (reads raw, z; writes d_raw)", 4 * (2 * 65 * N + N + 65) * R, timed(lambda: ops.composite_backward(raw, z, dF, dD)))
del raw
P = 1 << 22
x = torch.rand(P, 3, device=dev) * 10 - 5
report("posenc xyz (P=4M, 3 -> 93)", 4 * (3 + 93) * P, timed(lambda: ops.posenc(x, 15)))
zc = torch.sort(torch.rand(R, 64, device=dev) * 4 + 0.5, dim=1)[0]
wc = torch.rand(R, 64, device=dev)
report("sample_pdf + merge (R=65536, 64 + 128)", 4 * (2 * 64 + 192) * R, timed(lambda: ops.sample_pdf_merge(zc, wc, 128)))
HW = 640000
feat = torch.rand(HW, 64, device=dev)
report("decoder channel sums (800x800 grid)", 256 * HW, timed(lambda: ops.crossray_chansum(feat)))
aff = torch.rand(3 * 64 + 3, device=dev)
try:
    report("decoder apply (800x800 grid)", (256 + 12) * HW, timed(lambda: ops.crossray_apply(feat, aff)))
except Exception as e:   # affine layout is internal; skip if the size differs
    print("apply skipped:", e)


class A:
    nerf_out_dim, img_wh = 64, [800, 800]


net = style_net(A()).to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
style = torch.rand(1, 64, 32, 32, device=dev)
with torch.no_grad():
    report("whole decode (800x800 grid; 780 B/pixel)", 780 * HW, timed(lambda: net(feat.t().reshape(1, 64, 800, 800), style)))
