"""Tuning tool (GPU box): rebuilds with -DCRNERF_TIMING, runs the headline config and prints the
per-phase cycle breakdown of wave 0 / block 0 (shader clock)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRAIN = os.environ.get("CRNERF_PHASE_TRAIN") == "1"      # the training twin of the kernel (train=True) instead of the inference kernel
if not os.environ.get("CRNERF_LIB_PATH"):                  # (a -DCRNERF_TIMING variant from tools/variants.py needs no rebuild on the box)
    env = dict(os.environ, CRNERF_EXTRA_FLAGS="-DCRNERF_TIMING " + os.environ.get("CRNERF_EXTRA_FLAGS", ""))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], env=env, stdout=subprocess.DEVNULL)
    print("flags:", env["CRNERF_EXTRA_FLAGS"])
import torch
import crnerf_amd.synth as synth
from crnerf_amd import ops, _lib
dev = torch.device("cuda:0")
C = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
PREC = os.environ.get("CRNERF_PRECISION", "f32")   # bf16 -> the bf16 fused kernel
pc, pf = ops.pack_mlp_weights(C(synth.mlp_state(1, 3.0, 1.0)), precision=PREC), ops.pack_mlp_weights(C(synth.mlp_state(2, 3.0, 1.0)), precision=PREC)
rays = torch.from_numpy(synth.rays(1024)).to(dev)
for _ in range(3):
    ops.render_rays(pc, pf, rays, 64, 128, precision=PREC, train=TRAIN)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
fn = (lib.crnerf_debug_read_timing_h2 if PREC == "f32h2" else lib.crnerf_debug_read_timing_x3 if PREC == "f32x3" else
      lib.crnerf_debug_read_timing_bf16p if PREC == "bf16" else lib.crnerf_debug_read_timing16)      # exported by -DCRNERF_TIMING builds only
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
names = ["prologue(posenc)", "mma", "epilogue+init", "sigma", "composite", "ray-level", "total"] + ["x%d" % i for i in range(8)] + ["real (100 MHz)"]
tot = buf[6]
for n, v in zip(names, buf):
    print("%-18s %12d cycles  %6.2f %%" % (n, v, 100.0 * v / tot))
if buf[15]:
    print("wave lifetime %.1f us -> shader clock %.3f GHz" % (buf[15] / 100.0, tot / (buf[15] * 10.0)))
if PREC == "f32h2":
    print("x0..x6 = xyz_encoding_1 | 2-4 | 5 | 6-8 | final | dir | rgb; ideal matrix-pipe cycles per tile: 4608 | 36864 | 16896 | 36864 | 12288 | 6912 | 1536 = 115968; x 8 tiles per ray = %d" % (8 * 115968))
elif PREC == "bf16":
    print("ideal matrix-pipe cycles per SIMD: %d (9664 MFMA x 32 cycles; pair core: two waves of 4 x 1208 each)" % (4 * 2416 * 32))
else:
    print("ideal matrix-pipe cycles per SIMD: %d (8 steps x 9664 MFMA x 64 cycles-equivalent)" % (8 * 9664 * 64))
if not os.environ.get("CRNERF_KEEP_BUILD") and not os.environ.get("CRNERF_LIB_PATH"):   # restore the production build
    subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], stdout=subprocess.DEVNULL)
