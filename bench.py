"""Headline benchmark: rays/sec through the CR-NeRF rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic rays already resident in HBM:
render_rays_cross_ray (coarse 64 + fine 64+128 samples, two 8x256 NeRF_sigma MLPs, compositing,
sample_pdf, merge -- ONE fused HIP launch) followed by the cross-ray decode of the batch's feature
grid.  Workload = BASELINE.json configs[1]: 1024 rays x (64+128), fp32, per GPU (weak scaling: every
rank renders its own 1024-ray shard; the decoder's two tiny reductions and the RGB all-gather are the
only collectives).  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is render_rays16_kernel (fp32 MFMA bound).  achieved = algorithmic MLP
FLOPs per launch (1,233,152 FLOP/point x 256 points/ray x rays) / its average duration,
measured live with HIP events on the launch stream.  peak = 157.3 TFLOP/s (fp32 MFMA, MI355X guide).
cpu_baseline: the CPU oracle (plain-PyTorch restatement of the reference, oracle/cpu_ref.py) timed on
this box's host cores on a bounded sample of the same workload; rank 0, N == 1 only.
parity (with cpu_baseline): BASELINE's "PSNR vs ref" -- the decoded image of the timed batch against the oracle's image of the
same rays and weights: max |d rgb|, PSNR, and the PSNR difference against a common noisy target (bar: 0.05 dB).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 2 * 616576          # SURVEY 8a A4 / BASELINE.md section 2
PEAK_F32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), --precision bf16 only
NC, NI = 64, 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=1024, help="rays per GPU per step (configs[1]: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (the headline, default); bf16 = the bf16 matrix-core kernel of configs[2]")
    return ap.parse_args()


def cpu_baseline(rays_np, st_c, st_f, dst, grid_hw, style_nchw=None, gpu_rgb=None):
    """Oracle (kind 'port') on the host cores: same rays/weights/sample counts, bounded to ~10-30 s.
    The thread count is calibrated first (torch's intra-op pool collapses when every SMT thread of a
    big host is used on 256-wide layers): the fastest of a few candidates on a 128-ray probe is used."""
    from oracle import cpu_ref as O
    ncpu = os.cpu_count() or 1
    wc, wf, d = O.to_torch(st_c), O.to_torch(st_f), O.to_torch(dst)
    rays = torch.from_numpy(rays_np)
    style = style_nchw if style_nchw is not None else torch.rand(1, 64, 32, 32, generator=torch.Generator().manual_seed(0))

    def step(r):
        with torch.no_grad():
            out = O.render_rays(wc, wf, r, NC, NI)
            if r.shape[0] == rays.shape[0]:
                return O.crossray_decode(d, O.feature_to_grid(out["feature_fine"], *grid_hw), style)

    probe = rays[:128].contiguous()
    best_t, best_n = None, None
    for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(n)
        step(probe)
        t0 = time.perf_counter()
        step(probe)
        t = time.perf_counter() - t0
        if best_t is None or t < best_t:
            best_t, best_n = t, n
        if t > 20.0:
            break
    torch.set_num_threads(best_n)
    ref_rgb = step(rays)  # warm-up; also the checker for the "parity" object
    parity = None
    if gpu_rgb is not None:
        ref, got = ref_rgb.reshape(3, -1).double(), gpu_rgb.reshape(3, -1).double()
        target = (ref + 0.05 * torch.randn(ref.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64))
        psnr = lambda a, b: float(-10.0 * torch.log10(((a - b) ** 2).mean()))  # noqa: E731  (metrics.py:12-13)
        parity = {"max_abs_rgb_vs_oracle": float((ref - got).abs().max()), "psnr_vs_oracle_db": psnr(got, ref),
                  "delta_psnr_db": psnr(got, target) - psnr(ref, target),
                  "note": "decoded 32x32 image of the timed batch vs the CPU oracle on the same rays/weights; delta_psnr against a target = oracle image + N(0, 0.05^2)"}
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_all < 10.0 and len(times) < 20):
        t0 = time.perf_counter()
        step(rays)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 60.0:
            break
    times.sort()
    med = times[len(times) // 2]
    return parity, {"value": rays.shape[0] / med, "unit": "rays/s", "cores": best_n, "kind": "port",
            "sample": "%d reps (median) of the full step on %d rays x (%d+%d) samples + %dx%d cross-ray decode, fp32, torch %s CPU, "
                      "no_grad, %d threads (fastest of a 128-ray probe over 8..128 threads; host has %d logical CPUs)"
                      % (len(times), rays.shape[0], NC, NI, grid_hw[0], grid_hw[1], torch.__version__, best_n, ncpu)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)" % a.gpus)
        a.gpus = world
    import torch.distributed as dist
    # test hook: CRNERF_BENCH_TEST_BACKEND=gloo runs all ranks on cuda:0 over gloo, to exercise the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    test_backend = os.environ.get("CRNERF_BENCH_TEST_BACKEND")
    if test_backend:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ      # launched by torch.distributed.run (also with --nproc-per-node 1)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if test_backend:
            dist.init_process_group(test_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import crnerf_amd.synth as synth
    from crnerf_amd import ops
    from crnerf_amd.models.linearStyleTransfer import style_net
    from crnerf_amd.parallel import decode_sharded

    R = a.rays
    W = int(R ** 0.5)
    while R % W:
        W -= 1
    grid_hw = (R // W, W)
    st_c, st_f, dst = synth.mlp_state(1, 3.0, 1.0), synth.mlp_state(2, 3.0, 1.0), synth.decoder_state(3)
    rays_np = synth.rays(R, seed=rank, H=grid_hw[0], W=grid_hw[1])
    to_dev = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}  # noqa: E731

    class Args:
        nerf_out_dim, img_wh = 64, [grid_hw[1], grid_hw[0]]

    with torch.no_grad():
        pc, pf = ops.pack_mlp_weights(to_dev(st_c), precision=a.precision), ops.pack_mlp_weights(to_dev(st_f), precision=a.precision)
        net = style_net(Args()).to(dev)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in dst.items()})
        rays = torch.from_numpy(rays_np).to(dev)
        # the appearance encoder hands the decoder a pixel-major [1024,64] grid viewed as NCHW (zero-copy, models/linearStyleTransfer.py)
        style = torch.rand(1024, 64, generator=torch.Generator().manual_seed(0)).to(dev).view(1, 32, 32, 64).permute(0, 3, 1, 2)
        z_steps, u_steps = torch.linspace(0, 1, NC, device=dev), torch.linspace(0, 1, NI, device=dev)  # rendering.py:160, :27

        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

        def step(i=None):
            if i is not None:
                ev[i][0].record()
            out = ops.render_rays(pc, pf, rays, NC, NI, z_steps=z_steps, u=u_steps, precision=a.precision)
            if i is not None:
                ev[i][1].record()
            feat = out["feature_fine"]
            if use_dist:
                return decode_sharded(net, feat, style, gather=True, equal_shards=True)
            return net(feat.t().reshape(1, 64, *grid_hw), style)

        def fence():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(a.warmup):
            last_rgb = step()
        fence()
        t0 = time.perf_counter()
        for i in range(a.steps):
            last_rgb = step(i)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        kern_ms = sum(s.elapsed_time(e) for s, e in ev) / a.steps

    if rank == 0:
        flops = FLOP_PER_POINT * (NC + NC + NI) * R
        achieved = flops / (kern_ms * 1e-3) / 1e12
        bf16 = a.precision == "bf16"
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        kernel = "render_rays_bf16_kernel" if bf16 else ("render_rays_kernel" if os.environ.get("CRNERF_CORE") == "32" else "render_rays16_kernel")
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "render_rays_hbm_bytes.json")   # written from a rocprofv3 --pmc pass (profiles/README.md)
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get(a.precision, {}).get("hbm_bytes_per_launch_%d_rays" % R)
        line = {
            "metric": "rays/sec (64+128 samples, 8-layer W=256 MLP)", "value": world * R * a.steps / dt, "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]%s: %d rays x (%d coarse + %d fine) per GPU, NeRF_sigma 8x256 coarse+fine, "
                                   "fused render_rays + cross-ray decode of the %dx%d feature grid"
                                   % (2 if bf16 else 1, " arithmetic (bf16 MFMA operands, fp32 accumulate) on the configs[1] ray batch" if bf16 else "",
                                      R, NC, NI, grid_hw[0], grid_hw[1]),
                       "rays_per_gpu": R, "n_samples": NC, "n_importance": NI,
                       "parallelism": "rays sharded %d-way, weights replicated" % world},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "kernel_ms": kern_ms,
                         "flops_per_launch": flops},
        }
        if world == 1 and not a.no_cpu_baseline:
            parity, line["cpu_baseline"] = cpu_baseline(rays_np, st_c, st_f, dst, grid_hw, style.cpu().contiguous(), last_rgb.float().cpu())
            if parity:
                line["parity"] = parity
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
