"""GPU box: the bf16 fused renderer, pair core (default) vs the one-wave-per-SIMD core (CRNERF_BF16_CORE=64), same process layout as
bench.py's extra.bf16_kernel: 50 back-to-back C-ABI launches between one pair of events.  Also checks that the two cores agree.
usage: python tools/bf16_ab.py            (spawns one child per core; the core switch is read once per process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import crnerf_amd.synth as synth
    from crnerf_amd import ops
    dev = torch.device("cuda:0")
    C = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}  # noqa: E731
    pc, pf = ops.pack_mlp_weights(C(synth.mlp_state(1, 3.0, 1.0)), precision="bf16"), ops.pack_mlp_weights(C(synth.mlp_state(2, 3.0, 1.0)), precision="bf16")
    flop = 2 * 616576 * 256
    for R, n in ((1024, 50), (2048, 30), (16384, 10)):
        rays = torch.from_numpy(synth.rays(R, seed=3)).to(dev)
        launch, out = ops.render_rays(pc, pf, rays, 64, 128, precision="bf16", launcher=True)
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
        print("core %s  %6d rays: %8.1f us per launch = %6.1f us per 1024 rays, %.3f PFLOP/s = %.3f of 2.5" %
              (os.environ.get("CRNERF_BF16_CORE", "pair"), R, best * 1e3, best * 1e3 * 1024 / R, flop * R / best / 1e12, flop * R / best / 1e12 / 2.5), flush=True)
        if R == 1024:
            torch.save({k: v.cpu() for k, v in out.items()}, "/tmp/bf16_ab_%s.pt" % os.environ.get("CRNERF_BF16_CORE", "pair"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for core in ("64", None):
            env = dict(os.environ)
            env.pop("CRNERF_BF16_CORE", None)
            if core:
                env["CRNERF_BF16_CORE"] = core
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=env)
        import torch
        a, b = torch.load("/tmp/bf16_ab_64.pt"), torch.load("/tmp/bf16_ab_pair.pt")
        for k in a:
            d = (a[k].double() - b[k].double()).abs()
            print("pair vs 64  %-16s max |d| %.3e  (max |ref| %.3e)" % (k, float(d.max()), float(a[k].abs().max())))
