"""GPU box: where does the bf16 pair kernel's ENERGY go?  The kernel is power-limited (fewer cycles come back as a lower
shader clock), so each variant below removes one activity from the MLP stream -- results are garbage, only time and clock matter --
and reports the launch time, wave-0 cycles and the shader clock at the headline batch.
usage: python tools/bf16_energy_probe.py [variant ...]      variants: base mlponly nolds halflds noglds zero zeroact core64 core64zero"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLAGS = {"base": "", "nolds": "-DCRNERF_EXP_NOLDSREAD", "halflds": "-DCRNERF_EXP_HALFLDSREAD", "noglds": "-DCRNERF_EXP_NOGLDS", "noepi": "-DCRNERF_EXP_NOEPI", "mlponly": "-DCRNERF_EXP_MLP_ONLY",
         "zero": "", "zeroact": "", "core64": "", "core64zero": ""}


def child(variant):
    import numpy as np, torch
    import crnerf_amd.synth as synth
    from crnerf_amd import ops, _lib
    dev = torch.device("cuda:0")
    sc, sf = synth.mlp_state(1, 3.0, 1.0), synth.mlp_state(2, 3.0, 1.0)
    if variant in ("zero", "core64zero"):          # all-zero weights and biases: no operand toggling at all
        sc = {k: np.zeros_like(v) for k, v in sc.items()}
        sf = {k: np.zeros_like(v) for k, v in sf.items()}
    if variant == "zeroact":       # random weights, but every hidden activation is relu(negative) = 0: B operands of the hidden layers are zero
        for s in (sc, sf):
            for k in s:
                if k.endswith("bias") and k.startswith("xyz_encoding_") and "final" not in k:
                    s[k] = np.full_like(s[k], -1e4)
    C = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}  # noqa: E731
    pc, pf = ops.pack_mlp_weights(C(sc), precision="bf16"), ops.pack_mlp_weights(C(sf), precision="bf16")
    lib = _lib.load()
    out = []
    for R, n in ((1024, 50), (16384, 6)):
        rays = torch.from_numpy(synth.rays(R, seed=3)).to(dev)
        launch, _ = ops.render_rays(pc, pf, rays, 64, 128, precision="bf16", launcher=True)
        for _ in range(5):
            launch()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                launch()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        buf = (ctypes.c_ulonglong * 16)()
        fn = lib.crnerf_debug_read_timing_bf16 if variant.startswith("core64") else lib.crnerf_debug_read_timing_bf16p
        fn.argtypes = [ctypes.c_void_p]
        assert fn(buf) == 0
        out.append("%6d rays %8.1f us/launch (%6.1f per 1024) wave0 %7d cycles, mma %7d, clock %.3f GHz" %
                   (R, best * 1e3, best * 1e3 * 1024 / R, buf[6], buf[1] + sum(buf[8:12]), buf[6] / (buf[15] * 10.0) if buf[15] else 0))
    print("%-10s %s" % (variant, " | ".join(out)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        last = None
        for v in (sys.argv[1:] or ["base", "zero", "zeroact", "core64", "core64zero", "mlponly", "nolds", "halflds", "noglds"]):
            flags = "-DCRNERF_TIMING " + FLAGS[v]
            if flags != last:
                env = dict(os.environ, CRNERF_EXTRA_FLAGS=flags)
                subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], env=env, stdout=subprocess.DEVNULL)
                last = flags
            env = dict(os.environ)
            if v.startswith("core64"):
                env["CRNERF_BF16_CORE"] = "64"
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", v], env=env)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "cr-nerf-pytorch_amd", "build.py"), "--force"], stdout=subprocess.DEVNULL)
