"""Tuning tool (GPU box): rebuilds with -DCRNERF_TIMING and prints, for ONE launch of the bf16 fused renderer at the headline
batch, the spread of workgroup entry / exit times (s_memrealtime, 100 MHz) -- where a launch's fixed cost goes."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
env = dict(os.environ, CRNERF_EXTRA_FLAGS="-DCRNERF_TIMING")
subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], env=env, stdout=subprocess.DEVNULL)
import numpy as np, torch
import crnerf_amd.synth as synth
from crnerf_amd import ops, _lib
dev = torch.device("cuda:0")
C = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
pc, pf = ops.pack_mlp_weights(C(synth.mlp_state(1, 3.0, 1.0)), precision="bf16"), ops.pack_mlp_weights(C(synth.mlp_state(2, 3.0, 1.0)), precision="bf16")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rays = torch.from_numpy(synth.rays(R)).to(dev)
lib = _lib.load()
fn = lib.crnerf_debug_read_wgtimes_bf16 if os.environ.get("CRNERF_BF16_CORE") == "64" else lib.crnerf_debug_read_wgtimes_bf16p
fn.argtypes = [ctypes.c_void_p]
for rep in range(4):
    for _ in range(3):
        ops.render_rays(pc, pf, rays, 64, 128, precision="bf16")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.render_rays(pc, pf, rays, 64, 128, precision="bf16"); e.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 2048)()
    assert fn(buf) == 0
    t = np.array(buf[:512], dtype=np.float64).reshape(256, 2) / 100.0      # us
    t0 = t[:, 0].min()
    ent, ext, life = t[:, 0] - t0, t[:, 1] - t0, t[:, 1] - t[:, 0]
    print("event %.1f us | span (first entry -> last exit) %.1f us | entry spread %.1f us (p50 %.1f) | exit: min %.1f p50 %.1f max %.1f | lifetime: min %.1f p50 %.1f max %.1f"
          % (s.elapsed_time(e) * 1e3, ext.max(), ent.max(), np.median(ent), ext.min(), np.median(ext), ext.max(), life.min(), np.median(life), life.max()))
    if rep == 3:
        xcd = np.arange(256) % 8
        for x in range(8):
            print("  XCD %d: entry p50 %.1f, lifetime p50 %.1f max %.1f" % (x, np.median(ent[xcd == x]), np.median(life[xcd == x]), life[xcd == x].max()))
subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], stdout=subprocess.DEVNULL)
