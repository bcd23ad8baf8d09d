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
parity (with cpu_baseline, untimed): BASELINE's "PSNR vs ref" on the timed batch and on well-conditioned nets -- max-abs / rel-L2 of
feature_fine, weights_fine, depth_fine, z_fine end to end against the oracle and at identical depths, and the image decoded
through a HIGH-CONTRAST decoder (rgb spans most of [0,1], so PSNR responds to feature errors): max |d rgb|, PSNR, delta-PSNR.
extra (same run): the bf16 kernel's live kernel time / roofline fractions / parity vs the fp32 oracle, BASELINE configs[0]
(coarse only) and configs[2] (800x800 full image, bf16 and fp32) rates.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HIP runtime starts: RCCL / IPC need dmabuf handles on this pool

import torch  # noqa: E402

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
    ap.add_argument("--precision", choices=["f32", "bf16", "f32x3", "f32h2"], default="f32",
                    help="f32 = BASELINE configs[1] on the fp32 MFMA (the headline, default); bf16 = the bf16 matrix-core kernel of configs[2]; "
                         "f32x3 = configs[1] in fp32 on the bf16 matrix cores (three-piece splits, six MFMAs per product; DESIGN 3.4b); "
                         "f32h2 = the same on the fp16 matrix cores (two-piece splits, three MFMAs per product; DESIGN 3.4c)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): every rank renders its own --rays batch.  strong: ONE unit of --workload is "
                         "split over the ranks (the configs BASELINE.json names for 8 GPUs)")
    ap.add_argument("--workload", choices=["configs2", "configs3", "configs4"], default="configs2",
                    help="--scaling strong only.  configs2: one 800x800 frame, 64+128, cross-ray decoder on, --precision (default there: bf16); "
                         "configs3: one 65,536-ray training batch with grid-sample masking (fwd + bwd + Adam, --train-precision); "
                         "configs4: the appearance-hallucination fly-through -- --frames frames of 320x240 at --samples, style-image conditioned, "
                         "poses and rays made on the device, --video-split frames (no collective) or rays (decoder exchange per frame)")
    ap.add_argument("--frames", type=int, default=240, help="--workload configs4: frames of the fly-through (the script's 30 fps x 8 s; BASELINE says 120)")
    ap.add_argument("--samples", default="256+256", help="--workload configs4: N_samples+N_importance (the script's default 256+256; 64+128 = the training setting)")
    ap.add_argument("--video-split", choices=["frames", "rays"], default="frames",
                    help="--workload configs4: what the ranks split -- whole frames round-robin (no data-path collective) or every frame's rays")
    ap.add_argument("--frame", default="800x800", help="--workload configs2: frame size HxW")
    ap.add_argument("--train-rays", type=int, default=65536, help="--workload configs3: rays of the ONE batch that is split over the ranks")
    ap.add_argument("--train-precision", choices=["auto", "f32"], default="auto",
                    help="--workload configs3.  auto = the training default (crnerf_amd.autograd.get_training_forward_mode): forward and data gradient in fp32 "
                         "accuracy on the fp16 matrix cores with the f32x3 safety net, weight gradients from two-piece fp16 splits (f16x2) with the "
                         "three-piece bf16 splits (bf16x3) behind them; f32 = every product on the fp32 matrix cores (the reference's arithmetic)")
    ap.add_argument("--peer-exchange", action="store_true",
                    help="N > 1: carry the decoder's two reductions through HIP-IPC peer windows (parallel.PeerExchange) instead of RCCL")
    return ap.parse_args()


# which parity bar each arithmetic mode meets on the TRAINED checkpoint (tests/test_gpu_trained_ckpt.py): north_star's
# "PSNR within 0.05 dB of reference", SURVEY 8d's fp32 bars (features rel-L2 <= 1e-5, pixels <= 2e-5 resp. twice the reference's own
# fp64 - fp32 distance) and SURVEY 8d's bf16 pixel bar (max-abs <= 4e-3)
_FP32_BARS = "north_star 0.05 dB; SURVEY 8d fp32 bars (features rel-L2 1e-5, pixels 2e-5)"
MEETS = {"f32": _FP32_BARS, "f32x3": _FP32_BARS, "f32h2": _FP32_BARS, "auto": _FP32_BARS,
         "bf16": "north_star 0.05 dB (measured 0.004 dB); NOT SURVEY 8d's 4e-3 pixel bar on the trained checkpoint (6.1e-3: bf16 coarse weights move the fine depths)",
         "bf16_hc": "north_star 0.05 dB; SURVEY 8d 4e-3 pixel bar (measured 3.2e-4)"}

SMOOTH = dict(gain=2.45, sigma_bias=-1.0, band_limit=4)     # the well-conditioned nets of tests/golden g14 (synth.mlp_state)
CONTRAST = 4000.0                                            # high-contrast decoder: the image responds to feature errors (synth.decoder_state)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _diff(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return {"max_abs": float((got - ref).abs().max()), "rel_l2": float((got - ref).norm() / ref.norm().clamp_min(1e-30))}


def _image_metrics(got_rgb, ref_rgb):
    """BASELINE's "PSNR vs ref" (metrics.py:12-13) + SURVEY 8d's delta-PSNR against a common noisy target."""
    ref, got = ref_rgb.reshape(3, -1).double().cpu(), got_rgb.reshape(3, -1).double().cpu()
    target = ref + 0.05 * torch.randn(ref.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    psnr = lambda a, b: float(-10.0 * torch.log10(((a - b) ** 2).mean().clamp_min(1e-30)))  # noqa: E731
    return {"rgb_range": [float(ref.min()), float(ref.max())], "max_abs_rgb": float((ref - got).abs().max()),
            "psnr_vs_oracle_db": psnr(got, ref), "delta_psnr_db": psnr(got, target) - psnr(ref, target)}


def parity_block(O, gpu, rays, wc, wf, dec_hi, grid_hw, style, z_steps, u_steps, precision="f32"):
    """Feature-level and image-level parity of one GPU render (dict with z_fine and rgb_hi = its decode through the
    high-contrast decoder) against the oracle on the same rays / weights / linspace tables.
    end_to_end: the oracle samples its own fine depths (differences include the reference's own conditioning);
    identical_depths: the oracle re-evaluates the fine pass at the GPU path's z_fine (kernel arithmetic only)."""
    with torch.no_grad():
        e2e = O.render_rays(wc, wf, rays, NC, NI, z_steps=z_steps, u=u_steps)
        same = O.render_rays(wc, wf, rays, NC, NI, z_steps=z_steps, u=u_steps, z_fine=gpu["z_fine"].cpu(), precision=precision)
        ref_rgb = O.crossray_decode(dec_hi, O.feature_to_grid(e2e["feature_fine"], *grid_hw), style)
    keys = ("feature_fine", "weights_fine", "depth_fine")
    out = {"end_to_end": {k: _diff(gpu[k], e2e[k]) for k in keys + ("z_fine", "feature_coarse", "weights_coarse")},
           "identical_depths": {k: _diff(gpu[k], same[k]) for k in keys},
           "image_high_contrast": _image_metrics(gpu["rgb_hi"], ref_rgb)}
    out["end_to_end"]["z_fine"]["far"] = float(rays[:, 7].max())
    return out, e2e


def cpu_baseline(rays_np, st_c, st_f, dst, grid_hw, style_nchw):
    """Oracle (kind 'port'; tools/ab_oracle_vs_reference.py shows it costs what the imported reference costs, +-5 %) on the host
    cores: same rays / weights / sample counts as the timed step, bounded to ~10-30 s.  The thread count is calibrated on the
    FULL batch (torch's intra-op pool collapses when every SMT thread of a big host is used on 256-wide layers): the fastest of
    a few candidates is used for the timed repetitions.  Also times BASELINE configs[0] (1024 rays x 64 coarse only)."""
    from oracle import cpu_ref as O
    ncpu = os.cpu_count() or 1
    wc, wf, d = O.to_torch(st_c), O.to_torch(st_f), O.to_torch(dst)
    rays = torch.from_numpy(rays_np)

    def step(ni=NI):
        with torch.no_grad():
            out = O.render_rays(wc, wf, rays, NC, ni)
            if ni:
                return O.crossray_decode(d, O.feature_to_grid(out["feature_fine"], *grid_hw), style_nchw)

    best_t, best_n, probe = None, None, {}
    for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(n)
        if best_t is None:
            step()                      # first touch: page in MKL, allocator warm-up
        t0 = time.perf_counter()
        step()
        t = time.perf_counter() - t0
        probe[n] = round(t, 3)
        if best_t is None or t < best_t:
            best_t, best_n = t, n
        if t > 8.0:
            break
    torch.set_num_threads(best_n)

    def timed(fn, min_reps, budget):
        times, t_all = [], time.perf_counter()
        while len(times) < min_reps or (time.perf_counter() - t_all < budget and len(times) < 20):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > 4 * budget:
                break
        times.sort()
        return times[len(times) // 2], len(times)

    med, reps = timed(step, 3, 8.0)
    med0, reps0 = timed(lambda: step(0), 3, 4.0)
    whole_box = None
    try:
        whole_box = cpu_whole_box(best_n, ncpu, rays.shape[0])
    except Exception as e:   # noqa: BLE001
        whole_box = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"value": rays.shape[0] / med, "unit": "rays/s", "cores": best_n, "kind": "port", "cpu_model": _cpu_model(),
            "logical_cpus": ncpu, "thread_probe_s": probe, "whole_box_multi_process": whole_box,
            "configs0_coarse_only": {"value": rays.shape[0] / med0, "unit": "rays/s", "reps": reps0,
                                     "sample": "%d rays x %d coarse samples, render only" % (rays.shape[0], NC)},
            "sample": "%d reps (median) of the full step on %d rays x (%d+%d) samples + %dx%d cross-ray decode, fp32, torch %s CPU, "
                      "no_grad, %d threads (fastest of a FULL-batch probe over %s threads; host has %d logical CPUs)"
                      % (reps, rays.shape[0], NC, NI, grid_hw[0], grid_hw[1], torch.__version__, best_n, sorted(probe), ncpu)}


def cpu_whole_box(threads, ncpu, R):
    """Round-2 verdict, weak #12: one torch process stops scaling at ~16 threads on a 256-CPU host, so the single-process figure uses
    a sixteenth of the box.  This leg runs ncpu / (2 * threads) worker processes (one per disjoint CPU range, `threads` intra-op
    threads each: every physical core of an SMT-2 host busy once), each rendering the SAME 1,024-ray step repeatedly for a few
    seconds; the aggregate rays/s is the fairest "host cores of the same box" number for anyone who quotes a GPU / CPU ratio."""
    import subprocess
    nproc = max(1, min(ncpu // (2 * threads), 16))
    if nproc == 1:
        return {"processes": 1, "note": "host has no room for a second %d-thread worker" % threads}
    span = ncpu // nproc
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d,%d,%d,%d" % (threads, k * span, span, R)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(nproc)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    recs = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    t = max(r["elapsed_s"] for r in recs)
    steps = sum(r["reps"] for r in recs)
    return {"processes": nproc, "threads_each": threads, "cores": nproc * threads, "value": steps * R / t, "unit": "rays/s",
            "sample": "%d workers x %d threads, CPU ranges of %d logical CPUs each, every worker repeats the full %d-ray step for >= 6 s "
                      "(%d steps in all, slowest worker %.1f s)" % (nproc, threads, span, R, steps, t)}


def cpu_worker(spec):
    """`bench.py --cpu-worker threads,first_cpu,n_cpus,rays`: one worker of cpu_whole_box (no GPU, no torch.distributed)."""
    threads, first, span, R = (int(v) for v in spec.split(","))
    try:
        os.sched_setaffinity(0, range(first, first + span))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(threads)
    import crnerf_amd.synth as synth
    from oracle import cpu_ref as O
    W = int(R ** 0.5)
    while R % W:
        W -= 1
    grid_hw = (R // W, W)
    wc, wf, d = O.to_torch(synth.mlp_state(1, 3.0, 1.0)), O.to_torch(synth.mlp_state(2, 3.0, 1.0)), O.to_torch(synth.decoder_state(3))
    rays = torch.from_numpy(synth.rays(R, seed=0, H=grid_hw[0], W=grid_hw[1]))
    style = torch.rand(1024, 64, generator=torch.Generator().manual_seed(0)).view(1, 32, 32, 64).permute(0, 3, 1, 2).contiguous()

    def step():
        with torch.no_grad():
            out = O.render_rays(wc, wf, rays, NC, NI)
            return O.crossray_decode(d, O.feature_to_grid(out["feature_fine"], *grid_hw), style)
    step()
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < 6.0 or reps < 2:
        step()
        reps += 1
    print(json.dumps({"reps": reps, "elapsed_s": time.perf_counter() - t0}), flush=True)


def evidence(a, dev, rays, rays_np, st_c, st_f, grid_hw, style, z_steps, u_steps, Args):
    """Untimed, rank 0, N == 1: the `parity` object (feature- and image-level, fp32 kernel vs the oracle) and the `extra` object
    (bf16 kernel roofline + parity vs the fp32 oracle, BASELINE configs[0] and configs[2] rates) measured in this same run."""
    import numpy as np
    import crnerf_amd.synth as synth
    from crnerf_amd import ops, pipeline
    from crnerf_amd.models.linearStyleTransfer import style_net
    from oracle import cpu_ref as O
    R = rays.shape[0]
    to_dev = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}  # noqa: E731
    dst_hi = synth.decoder_state(3, 1.0, contrast=CONTRAST)
    net_hi = style_net(Args()).to(dev)
    net_hi.load_state_dict({k: torch.from_numpy(v) for k, v in dst_hi.items()})
    style_cpu, rays_cpu = style.cpu().contiguous(), torch.from_numpy(rays_np)
    zt, ut = z_steps.cpu(), u_steps.cpu()
    sm_c, sm_f = synth.mlp_state(1, **SMOOTH), synth.mlp_state(2, **SMOOTH)

    def gpu_render(sc, sf, precision):
        with torch.no_grad():
            out = ops.render_rays(ops.pack_mlp_weights(to_dev(sc), precision=precision), ops.pack_mlp_weights(to_dev(sf), precision=precision),
                                  rays, NC, NI, z_steps=z_steps, u=u_steps, want_z_fine=True, precision=precision)
            out["rgb_hi"] = net_hi(out["feature_fine"].t().reshape(1, 64, *grid_hw), style)
        return out

    parity = {"tolerance_stated": "SURVEY 8d, fp32, where the reference is well-conditioned (smooth_nets.end_to_end): pixels max-abs <= 2e-5, "
                                  "features rel-L2 <= 1e-5, fine z max-abs <= 1e-5*far; bf16: pixels <= 4e-3, |delta PSNR| <= 0.05 dB",
              "decoder": "synth.decoder_state(3, contrast=%g): image spans rgb_range, a 1e-3 feature error moves pixels by ~7e-3" % CONTRAST}
    args_hi = O.to_torch(dst_hi)
    parity["timed_batch_peaky_nets"], _ = parity_block(O, gpu_render(st_c, st_f, "f32"), rays_cpu, O.to_torch(st_c), O.to_torch(st_f), args_hi,
                                                       grid_hw, style_cpu, zt, ut)
    parity["timed_batch_peaky_nets"]["note"] = ("gain-3 nets of the timed step: end_to_end includes the reference's own ill-conditioning "
                                                "(2^14 embedding gain x sample_pdf's denom<eps switch, DESIGN section 5); identical_depths is the kernel's arithmetic")
    parity["smooth_nets"], _ = parity_block(O, gpu_render(sm_c, sm_f, "f32"), rays_cpu, O.to_torch(sm_c), O.to_torch(sm_f), args_hi,
                                            grid_hw, style_cpu, zt, ut)
    e2e = parity["smooth_nets"]["end_to_end"]
    parity["meets_stated_tolerance"] = bool(parity["smooth_nets"]["image_high_contrast"]["max_abs_rgb"] <= 2e-5
                                            and e2e["feature_fine"]["rel_l2"] <= 1e-5 and e2e["z_fine"]["max_abs"] <= 1e-5 * e2e["z_fine"]["far"])

    extra = {}
    # ---- bf16 matrix-core kernel (BASELINE configs[2] arithmetic) on the timed ray batch: live HIP-event kernel time + parity
    with torch.no_grad():
        pcb, pfb = ops.pack_mlp_weights(to_dev(st_c), precision="bf16"), ops.pack_mlp_weights(to_dev(st_f), precision="bf16")
        n = max(a.steps, 50)
        # n launches back to back between ONE pair of events, nothing but the C call on the host side between them: a per-launch
        # event pair (and ~100 us of Python per ops.render_rays call) would be charged to a 0.24 ms kernel; the fp32 kernel's
        # 2.26 ms hides both.  This average is what rocprofv3 --kernel-trace reports as the kernel's duration.
        launch, _ = ops.render_rays(pcb, pfb, rays, NC, NI, z_steps=z_steps, u=u_steps, precision="bf16", launcher=True)
        for _ in range(5):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        # the same kernel on 16 x the batch in ONE launch (16 passes of the persistent grid, quads pulled dynamically): what a
        # full-image render sees per 1,024 rays
        rays16 = torch.from_numpy(synth.rays(16 * R, seed=7)).to(dev)
        launch16, _ = ops.render_rays(pcb, pfb, rays16, NC, NI, z_steps=z_steps, u=u_steps, precision="bf16", launcher=True)
        for _ in range(2):
            launch16()
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e2.record()
        for _ in range(10):
            launch16()
        e3.record()
        torch.cuda.synchronize()
        ms16 = e2.elapsed_time(e3) / 10
    flops = FLOP_PER_POINT * (NC + NC + NI) * R
    tf = flops / (ms * 1e-3) / 1e12
    tf16 = 16 * flops / (ms16 * 1e-3) / 1e12
    bf = parity_block(O, gpu_render(sm_c, sm_f, "bf16"), rays_cpu, O.to_torch(sm_c), O.to_torch(sm_f), args_hi, grid_hw, style_cpu, zt, ut,
                      precision="bf16")[0]
    extra["bf16_kernel"] = {"kernel": "render_rays_bf16p_kernel", "meets": MEETS["bf16"], "kernel_ms": ms, "achieved_tflops": tf, "frac_nominal_2500": tf / PEAK_BF16_MFMA_TFLOPS,
                            "frac_attainable_1890": tf / 1890.0, "rays_per_s_kernel_only": R / (ms * 1e-3),
                            "multi_pass_16x": {"rays": 16 * R, "launch_ms": ms16, "ms_per_%d_rays" % R: ms16 / 16, "achieved_tflops": tf16,
                                               "frac_nominal_2500": tf16 / PEAK_BF16_MFMA_TFLOPS, "frac_attainable_1890": tf16 / 1890.0},
                            "note": "kernel_ms = launch-to-launch time of 50 back-to-back C-ABI launches of the pair core (render_rays_bf16p_kernel: one ray per wave "
                                    "pair, two waves per SIMD).  The kernel is POWER-governed on this part "
                                    "(profiles/r3/energy_probe.txt, DESIGN 3.7): the pair core spends 372k cycles per ray pair against 426k for the round-2 kernel "
                                    "(matrix pipe busy 88 %% vs 72 %%) and the shader clock drops from 1.85 to 1.65 GHz, same wall time; with all-zero operands the SAME "
                                    "binary runs at 2.4 GHz = 0.78 (single launch) / 0.82 (multi-pass) of the nominal 2.5 PFLOP/s; the MLP stream alone (no per-ray "
                                    "work, same ring) caps at 0.57 / 0.59 with random operands.  1890 TFLOP/s = bare v_mfma_f32_32x32x16_bf16 stream under sustained "
                                    "load (profiles/r1/ubench_mfma_stream.txt); same %d-ray batch and weights as the timed fp32 step" % R,
                            "parity_smooth_nets": {"vs_fp32_oracle_end_to_end": bf["end_to_end"], "vs_bf16_oracle_identical_depths": bf["identical_depths"],
                                                   "image_high_contrast_vs_fp32_oracle": bf["image_high_contrast"]}}

    # ---- "f32x3" / "f32h2": the same fp32 path evaluated on the bf16 / fp16 matrix cores (three-piece bf16 splits, six MFMAs per product; two-piece
    # fp16 splits, three MFMAs per product; fp32 accumulation; include/crnerf.h) on the timed ray batch.  NOT the headline: the
    # timed step above is the fp32-MFMA kernel.
    def split_kernel(prec, pack, pieces_products):
        with torch.no_grad():
            pcx, pfx = pack(to_dev(st_c)), pack(to_dev(st_f))
            launchx, _ = ops.render_rays(pcx, pfx, rays, NC, NI, z_steps=z_steps, u=u_steps, precision=prec, launcher=True)
            for _ in range(5):
                launchx()
            e4, e5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e4.record()
            for _ in range(n):
                launchx()
            e5.record()
            torch.cuda.synchronize()
            msx = e4.elapsed_time(e5) / n

            def gpu_render_split(sc, sf):
                out = ops.render_rays(pack(to_dev(sc)), pack(to_dev(sf)), rays, NC, NI, z_steps=z_steps, u=u_steps, want_z_fine=True, precision=prec)
                out["rgb_hi"] = net_hi(out["feature_fine"].t().reshape(1, 64, *grid_hw), style)
                return out
            px3 = parity_block(O, gpu_render_split(sm_c, sm_f), rays_cpu, O.to_torch(sm_c), O.to_torch(sm_f), args_hi, grid_hw, style_cpu, zt, ut)[0]
        issued = pieces_products * (7296.0 / 7248.0) * flops / (msx * 1e-3) / 1e12
        return {"kernel_ms": msx, "rays_per_s_kernel_only": R / (msx * 1e-3), "fp32_work_tflops": flops / (msx * 1e-3) / 1e12,
                "issued_mfma_tflops": issued, "frac_nominal_2500_of_issued_mfma": issued / PEAK_BF16_MFMA_TFLOPS, "speedup_vs_fp32_mfma_kernel": None,
                "parity_smooth_nets": {"end_to_end": px3["end_to_end"], "identical_depths": px3["identical_depths"],
                                       "image_high_contrast": px3["image_high_contrast"],
                                       "meets_stated_fp32_tolerance": bool(px3["image_high_contrast"]["max_abs_rgb"] <= 2e-5 and
                                                                           px3["end_to_end"]["feature_fine"]["rel_l2"] <= 1e-5 and
                                                                           px3["end_to_end"]["z_fine"]["max_abs"] <= 1e-5 * px3["end_to_end"]["z_fine"]["far"])}}
    extra["f32x3_kernel"] = dict(split_kernel("f32x3", ops.pack_mlp_weights_x3, 6), kernel="render_rays_x3_kernel",
                                 note="fp32 inputs / outputs / biases / activations / embeddings; each product of the eleven nn.Linear = the six leading products of "
                                      "three-piece bf16 splits of both fp32 operands (w = w1 + w2 + w3, 24 mantissa bits), exact in fp32, accumulated in fp32; dropped "
                                      "terms <= 3 x 2^-24 of a product.  Meets the fp32 entry points' goldens and SURVEY 8d's fp32 bars (tests/test_gpu_x3.py) and sits at "
                                      "the fp32 MFMA's distance from a float64 evaluation.  fp32_work_tflops counts the ALGORITHMIC fp32 FLOPs (it exceeds the fp32 "
                                      "MFMA peak of 157.3: the work runs on the bf16 pipe); the issued bf16 MFMA work is 6x that, at the power-limited rate the bf16 "
                                      "renderer reaches (extra.bf16_kernel)")
    extra["f32x3_kernel"]["bf16_mfma_tflops"] = extra["f32x3_kernel"]["issued_mfma_tflops"]
    extra["f32h2_kernel"] = dict(split_kernel("f32h2", ops.pack_mlp_weights_h2, 3), kernel="render_rays_h2_kernel",
                                 note="as f32x3 with TWO fp16 pieces per operand (x = h1 + h2: 11 + 11 mantissa bits and the sign of h2) and the THREE leading piece "
                                      "products per product (dropped: h2 h2 <= 2^-24): half the MFMAs, two thirds of the weight stream.  Weights are scaled by 2^8 at "
                                      "pack time (undone exactly in registers) so that their second pieces stay normal fp16 numbers; the matrix cores honour fp16 "
                                      "subnormals.  NOT scale-free: |w| < 255 is checked when packing, an activation >= 65,504 turns its point's outputs into NaN "
                                      "(tests/test_gpu_h2.py).  Meets the same fp32 goldens / SURVEY 8d bars and sits at the fp32 MFMA's distance from float64")

    # "auto" = the h2 launch + the x3 repair launch that finds nothing to repair on these weights: what the safety net costs
    with torch.no_grad():
        pca, pfa = ops.pack_mlp_weights(to_dev(st_c), precision="auto"), ops.pack_mlp_weights(to_dev(st_f), precision="auto")
        launcha, _ = ops.render_rays(pca, pfa, rays, NC, NI, z_steps=z_steps, u=u_steps, precision="auto", launcher=True)
        for _ in range(5):
            launcha()
        e6, e7 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e6.record()
        for _ in range(n):
            launcha()
        e7.record()
        torch.cuda.synchronize()
    extra["auto_kernels"] = {"kernels": "render_rays_h2_kernel + render_rays_x3_kernel (repair: one workgroup per ray quad, all of them leave at once)",
                             "kernel_ms": e6.elapsed_time(e7) / n, "f32h2_alone_ms": extra["f32h2_kernel"]["kernel_ms"],
                             "note": "precision='auto' (include/crnerf.h): f32h2 with the scale-free f32x3 core as its safety net -- no NaN of the h2 core's making "
                                     "reaches the caller (tests/test_gpu_h2.py::test_render_auto_repairs_poisoned_rays)"}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # ---- BASELINE configs[0]: 1024 rays x 64 coarse only (fp32, HIP)
    with torch.no_grad():
        pc = ops.pack_mlp_weights(to_dev(st_c))
        t0 = timed(lambda: ops.render_rays(pc, None, rays, NC, 0, z_steps=z_steps), 50)
    extra["configs0_coarse_only_f32"] = {"rays_per_s": R / t0, "ms": t0 * 1e3, "tflops": FLOP_PER_POINT * NC * R / t0 / 1e12}

    # ---- BASELINE configs[2]: 800x800 image, 32,768-ray chunks, 64+128, cross-ray decoder on, bf16, through the drop-in modules
    class HP:
        nerf_out_dim, pertubeCord, N_emb_xyz, N_emb_dir, use_disp, encode_a, encode_random, N_a = 64, False, 15, 4, False, True, True, 48
        img_wh, N_samples, N_importance = [800, 800], NC, NI
    hp = HP()
    m, emb = pipeline.get_model(hp, dev), pipeline.get_embeddings(hp)
    m["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in st_c.items()})
    m["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in st_f.items()})
    m["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
    enc = pipeline.encoder_sameoutputsize(64).to(dev)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(4, 2.0).items()})
    focal = 800 / 2 / np.tan(np.pi / 6)
    K = np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]])
    c2w = np.array([[1, 0, 0, 0.05], [0, -1, 0, 0.02], [0, 0, -1, 0.1]], dtype=np.float32)
    photo = torch.rand(1, 3, 100, 100, device=dev)
    for prec, reps in (("bf16", 3), ("bf16_hc", 2), ("auto", 2), ("f32h2", 2), ("f32x3", 2), ("f32", 1)):
        t = timed(lambda: pipeline.render_frame(m, emb, enc, photo, 800, 800, K, c2w, hp, chunk=32768, precision=prec), reps)
        extra["configs2_full_image_%s" % prec] = {"meets": MEETS[prec], "rays_per_s": 640000 / t, "ms_per_frame": t * 1e3, "tflops": FLOP_PER_POINT * (NC + NC + NI) * 640000 / t / 1e12,
                                                  "workload": "800x800 rays in 32,768-ray chunks x (64+128), appearance encoder + on-device rays + "
                                                              "render + cross-ray decode of the 640k-pixel grid"}
    return parity, extra


def collectives_record(dist, dev, world, rank, test_backend):
    """What the collective layer saw (every rank contributes, rank 0 keeps it): backend, world size, the device each rank ran on
    (name, PCI bus id, uuid), the RCCL version torch was built against.  Per-step collective times are added by the caller."""
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": dev.index, "device_name": props.name,
            "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
            "uuid": str(getattr(props, "uuid", "")), "host": os.uname().nodename}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:   # noqa: BLE001
        rccl = None
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": rccl, "ranks": everyone,
            "distinct_devices": len({(r["host"], r["pci_bus_id"]) for r in everyone}),
            "test_hook_backend": test_backend,   # null in a real run; "gloo" = all ranks on one GPU (tests only)
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}


def timed_steps(a, step, use_dist, dist, dev):
    """The contract's timed region: W untimed steps, then EXACTLY K steps between barrier + synchronize, MAX over ranks."""
    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    last = None
    for _ in range(a.warmup):
        last = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def strong_configs2(a, dev, world, rank, use_dist, dist, exchange):
    """ONE configs[2] frame (default 800x800 = 640,000 rays, 64+128, cross-ray decoder on) split over the ranks: contiguous ray blocks,
    32,768-ray render chunks per rank, the decoder's two all-reduces + the RGB all-gather (parallel.decode_sharded)."""
    import numpy as np
    import crnerf_amd.synth as synth
    from crnerf_amd import ops, parallel
    from crnerf_amd.models.linearStyleTransfer import style_net
    H, W = (int(v) for v in a.frame.lower().split("x"))
    R = H * W
    lo, hi = parallel.shard_bounds(R, world, rank)
    prec = a.precision
    to_dev = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}  # noqa: E731

    class Args:
        nerf_out_dim, img_wh = 64, [W, H]

    with torch.no_grad():
        pc, pf = (ops.pack_mlp_weights(to_dev(synth.mlp_state(sd, 3.0, 1.0)), precision=prec) for sd in (1, 2))
        net = style_net(Args()).to(dev)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
        rays = torch.from_numpy(synth.rays(R, seed=0, H=H, W=W, near=0.0, far=5.0)[lo:hi]).to(dev)     # this rank's rows of the frame
        style = torch.rand(1024, 64, generator=torch.Generator().manual_seed(0)).to(dev).view(1, 32, 32, 64).permute(0, 3, 1, 2)
        z_steps, u_steps = torch.linspace(0, 1, NC, device=dev), torch.linspace(0, 1, NI, device=dev)
        ev = []

        def step():
            feats = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(0, hi - lo, 32768):
                feats.append(ops.render_rays(pc, pf, rays[i:i + 32768], NC, NI, z_steps=z_steps, u=u_steps, precision=prec)["feature_fine"])
            e1.record()
            ev.append((e0, e1))
            feat = torch.cat(feats) if len(feats) != 1 else feats[0]
            if use_dist:
                return parallel.decode_sharded(net, feat, style, gather=True, equal_shards=(R % world == 0), exchange=exchange, check_exchange=False)
            return net(feat.t().reshape(1, 64, H, W), style)

        dt, last = timed_steps(a, step, use_dist, dist, dev)
        torch.cuda.synchronize()
        render_ms = sum(s.elapsed_time(e) for s, e in ev[-a.steps:]) / a.steps
    flops_local = FLOP_PER_POINT * (NC + NC + NI) * (hi - lo)
    peak = PEAK_BF16_MFMA_TFLOPS if prec in ("bf16", "f32x3", "f32h2") else PEAK_F32_MFMA_TFLOPS
    issued = {"f32x3": 6.0 * 7296.0 / 7248.0, "f32h2": 3.0 * 7296.0 / 7248.0}.get(prec, 1.0)   # split modes: the piece MFMAs the kernel ISSUES
    achieved = issued * flops_local / (render_ms * 1e-3) / 1e12
    kernel = {"bf16": "render_rays_bf16p_kernel", "f32x3": "render_rays_x3_kernel", "f32h2": "render_rays_h2_kernel"}.get(prec, "render_rays16_kernel")
    return {"metric": "rays/sec (64+128 samples, 8-layer W=256 MLP)", "value": R * a.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": prec,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: ONE %dx%d frame (%d rays) x (%d coarse + %d fine) split over %d rank(s) in contiguous ray blocks, "
                                   "32,768-ray render chunks, cross-ray decode of the whole frame (two all-reduces + RGB all-gather)" % (H, W, R, NC, NI, world),
                       "rays_total": R, "rays_this_rank": hi - lo, "n_samples": NC, "n_importance": NI,
                       "parallelism": "one frame's rays sharded %d-way, weights replicated" % world,
                       "reductions": "none" if world == 1 else ("peer windows (HIP IPC)" if exchange is not None else "RCCL all-reduce")},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "traffic_source": None, "kernel_ms": render_ms,
                         "flops_per_launch": issued * flops_local, "note": "rank 0's render chunks of one frame (HIP events around the chunk loop)"},
            "meets": MEETS.get(prec), "image_checksum": float(last.double().sum())}


def strong_configs4(a, dev, world, rank, use_dist, dist, exchange):
    """BASELINE configs[4]: the appearance-hallucination video path (appearance_modification_video.py:121-189 poses, :224-262 frame loop) --
    ONE fly-through of --frames frames of 320x240 at --samples, the style image encoded once, every frame = camera pose -> rays on the device ->
    render_rays_cross_ray -> cross-ray decode -> uint8.  The unit that is split over the ranks: --video-split frames = frames rank, rank + N, ...
    (independent: no collective on the data path, SURVEY 8e "video"), rays = every frame's pixel rows in contiguous blocks (the decoder's two
    all-reduces + the RGB all-gather per frame).  value = rays of the WHOLE fly-through per second."""
    import numpy as np
    import crnerf_amd.synth as synth
    from crnerf_amd import parallel, pipeline, video
    nc, ni = (int(v) for v in a.samples.split("+"))
    prec = a.precision
    W, H = 320, 240

    class HP:
        nerf_out_dim, pertubeCord, N_emb_xyz, N_emb_dir, use_disp, encode_a, encode_random, N_a = 64, False, 15, 4, False, True, True, 48
        img_wh, N_samples, N_importance = [W, H], nc, ni
    hp = HP()
    with torch.no_grad():
        m, emb = pipeline.get_model(hp, dev), pipeline.get_embeddings(hp)
        m["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(1, 3.0, 1.0).items()})
        m["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(2, 3.0, 1.0).items()})
        m["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
        enc = pipeline.encoder_sameoutputsize(64).to(dev)
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(4, 2.0).items()})
        style = torch.rand(1, 3, H // 4, W // 4, generator=torch.Generator().manual_seed(0)).to(dev)
        K, poses = video.define_camera(hp.img_wh), video.define_poses("brandenburg_gate", a.frames).astype(np.float32)
        by_rays = a.video_split == "rays" and world > 1
        lo, hi = parallel.shard_bounds(H * W, world, rank) if by_rays else (0, H * W)
        mine = range(a.frames) if by_rays else range(rank, a.frames, world)
        ev = []

        def step():
            a_emb = enc(style)                                   # once per style image (appearance_modification_video.py:239)
            checksum = torch.zeros((), dtype=torch.float64, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            render_ms = 0.0
            for i in mine:
                rays = pipeline.generate_rays(H, W, K, poses[i], 0.0, 5.0, device=dev)[lo:hi]
                res = pipeline.batched_inference(m, emb, rays, None, nc, ni, False, 76800, False, args=hp, a_embedded_from_img=a_emb, precision=prec)
                if by_rays:
                    rgb = parallel.decode_sharded(m["decoder"], res["feature_fine"], a_emb, gather=True, equal_shards=((H * W) % world == 0),
                                                  exchange=exchange, check_exchange=False)
                    img = rgb.reshape(3, H * W).t().reshape(H, W, 3)
                else:
                    img = pipeline.decode_image(m, res, H, W, a_emb).reshape(H, W, 3)
                frame = (img.clamp(0, 1) * 255).to(torch.uint8)  # what the script writes out (:255-262); stays on the device here
                checksum += frame.double().sum()
            return checksum

        dt, last = timed_steps(a, step, use_dist, dist, dev)
        total = last.clone()
        if use_dist and not by_rays:
            dist.all_reduce(total)                               # (untimed) the whole fly-through's checksum: every rank holds only its frames
    R = a.frames * H * W
    pts = (nc + nc + ni)
    flops_local = FLOP_PER_POINT * pts * len(mine) * (hi - lo)
    peak = PEAK_BF16_MFMA_TFLOPS if prec in ("bf16", "f32x3", "f32h2") else PEAK_F32_MFMA_TFLOPS
    issued = {"f32x3": 6.0 * 7296.0 / 7248.0, "f32h2": 3.0 * 7296.0 / 7248.0}.get(prec, 1.0)
    achieved = issued * flops_local / (dt / a.steps) / 1e12
    kernel = {"bf16": "render_rays_bf16p_kernel", "f32x3": "render_rays_x3_kernel", "f32h2": "render_rays_h2_kernel"}.get(prec, "render_rays16_kernel")
    return {"metric": "rays/sec (%d+%d samples, 8-layer W=256 MLP)" % (nc, ni), "value": R * a.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": prec,
            "data": "synthetic", "frames_per_s": a.frames * a.steps / dt,
            "config": {"workload": "BASELINE configs[4]: appearance-hallucination fly-through, %d frames of %dx%d x (%d coarse + %d fine), style-image "
                                   "conditioned decoder, poses + rays made on the device, split over %d rank(s) by %s"
                                   % (a.frames, W, H, nc, ni, world, "rays of every frame (decoder exchange per frame)" if by_rays else "whole frames (no collective)"),
                       "rays_total": R, "frames": a.frames, "frames_this_rank": len(mine), "rays_per_frame_this_rank": hi - lo, "n_samples": nc, "n_importance": ni,
                       "parallelism": ("every frame's rays sharded %d-way" if by_rays else "frames round-robin over %d rank(s), no data-path collective") % world,
                       "reductions": "none" if not by_rays else ("peer windows (HIP IPC)" if exchange is not None else "RCCL all-reduce")},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                         "traffic_source": None, "flops_per_launch": issued * FLOP_PER_POINT * pts * min(76800, hi - lo),
                         "note": "this rank's share of the fly-through over the WHOLE step time (ray generation, encoder, decoder, uint8 conversion included)"},
            "image_checksum": float(last), "frames_checksum_all_ranks": float(total)}


def strong_configs3(a, dev, world, rank, use_dist, dist):
    """ONE configs[3] training batch (default 65,536 rays = a 256x256 grid-sample batch with transient masking, 64+64 samples as
    command/train.sh trains) split over the ranks: ray-parallel TrainingSystem (parallel.GatherRays + sync_gradients), forward + backward +
    Adam per step, in --train-precision (auto: fp32-accurate split-operand products on the fp16 / bf16 matrix cores; f32: the fp32 matrix cores)."""
    import numpy as np
    import crnerf_amd.synth as synth
    from crnerf_amd import autograd as AG, optim as crnerf_optim, pipeline
    exact = a.train_precision == "f32"
    AG.set_training_forward_precision("f32" if exact else "auto")
    AG.set_wgrad_precision("f32" if exact else None)   # None: the default behind the h2 data gradient = f16x2 (bf16x3 as its in-launch fallback)
    from crnerf_amd.datasets.phototourism_mask_grid_sample import GridSampleBatcher
    R = a.train_rays
    side = int(R ** 0.5)
    nc, ni = 64, 64

    class HP:
        maskrs_max, maskrs_min, maskrs_k, maskrd = 5e-2, 6e-3, 1e-3, 0.0
        weightKL, weightRecA, weightcontent, mse_on_appearance = 1e-5, 1e-3, 1e-4, False
        nerf_out_dim, pertubeCord, N_emb_xyz, N_emb_dir, use_disp, encode_a, encode_random, N_a = 64, False, 15, 4, False, True, True, 48
        img_wh, N_samples, N_importance, perturb, noise_std, chunk, N_vocab = [side, side], nc, ni, 1.0, 1.0, 8 * 1024, 1500
        use_mask, encode_c = True, True

    hp = HP()
    torch.manual_seed(0)
    sysm = pipeline.TrainingSystem(hp, device=dev, ray_parallel_group=None if use_dist else False)   # (also with ONE rank under the launcher: the collectives are then issued and timed over RCCL)
    sysm.models["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(1, 2.0, 0.5).items()})
    sysm.models["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(2, 2.0, 0.5).items()})
    sysm.models["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
    sysm.enc_a.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(4, 2.0).items()})
    sysm.enc_cont.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(5, 2.0).items()})
    n_img, iw, ih = 8, 512, 384
    rays = torch.cat([torch.cat([torch.from_numpy(synth.rays(iw * ih, seed=i, H=ih, W=iw)), torch.full((iw * ih, 1), float(i))], 1) for i in range(n_img)]).to(dev)
    rgbs = torch.rand(n_img * iw * ih, 3, device=dev)
    imgs = [torch.rand(1, 3, ih // 8, iw // 8, device=dev) * 2 - 1 for _ in range(n_img)]
    batcher = GridSampleBatcher(rays, rgbs, np.array([[iw, ih]] * n_img), batch_size=R, all_imgs=imgs)
    opt = crnerf_optim.FlatAdam(sysm.parameters(), lr=5e-4, eps=1e-8)      # utils/__init__.py:31 Adam(lr, eps=1e-8), one HIP launch (crnerf_adam_step_f32)
    counter = [0]

    def step():
        batch = batcher.__getitem__(counter[0], 0)
        counter[0] += 1
        opt.zero_grad(set_to_none=True)
        loss, _, _ = sysm.training_step(batch)
        loss.backward()
        if use_dist:
            sysm.sync_gradients()
        opt.step()
        return loss.detach()

    dt, last = timed_steps(a, step, use_dist, dist, dev)
    # One more step, instrumented: where the ray-SHARDED part (the two MLPs' render, forward and backward: 1 / N of it per rank) ends and the rest
    # begins.  Events: step start | features ready (render + all-gather done) | forward done | gradient arrives back at the features (everything
    # downstream of the render has been differentiated) | backward done | gradients synchronised + Adam done.
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    fired = []

    def probe(results):
        ev[1].record()
        for k in ("feature_coarse", "feature_fine"):
            if k in results and results[k].requires_grad:
                results[k].register_hook(lambda g, k=k: (fired.append(k), ev[3].record())[0] and None)     # the last one to fire leaves the stamp
    sysm.after_render = probe
    batch = batcher.__getitem__(counter[0], 0)
    opt.zero_grad(set_to_none=True)
    ev[0].record()
    loss, _, _ = sysm.training_step(batch)
    ev[2].record()
    loss.backward()
    ev[4].record()
    if use_dist:
        sysm.sync_gradients()
    opt.step()
    ev[5].record()
    torch.cuda.synchronize()
    sysm.after_render = None
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    sections = {"render_forward_ms": t[0], "rest_forward_ms": t[1], "rest_backward_ms": t[2], "render_backward_ms": t[3], "sync_and_adam_ms": t[4],
                "sharded_ms": t[0] + t[3], "not_sharded_ms": t[1] + t[2] + t[4],
                "note": "one instrumented step after the timed ones, HIP events on the launch stream.  sharded = the renderer's forward and backward over this "
                        "rank's rays (1 / N of the batch).  not_sharded = decodes, mask network, loss, optimiser and -- replicated at N = 1, row bands over "
                        "the ranks at N > 1 (parallel.encode_banded) -- the three encoder passes over the re-rendered images: the part that does not "
                        "shrink as 1 / N (DESIGN 4)"}
    pts = R * (nc + nc + ni)
    achieved = 3 * pts * FLOP_PER_POINT / dt * a.steps / 1e12 / world      # per GPU: algorithmic fp32 FLOPs of forward + data gradient + weight gradient
    # auto: what the matrix cores are ISSUED -- forward 3.02 x, data gradient 3 x, weight gradient 3 x (two fp16 pieces per operand, three MFMAs per
    # product, all three) the algorithmic FLOPs of their third of the step
    issued = achieved if exact else achieved * (3.02 + 3.0 + 3.0) / 3.0
    peak = PEAK_F32_MFMA_TFLOPS if exact else PEAK_BF16_MFMA_TFLOPS
    line = {"metric": "rays/sec (64+64 samples, training step: fwd + bwd + Adam)", "value": R * a.steps / dt, "unit": "rays/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if exact else "f32h2 / f16x2 (fp32 operands split into two fp16 pieces, three MFMAs per product, fp32 accumulation)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: ONE %d-ray grid-sample training batch (%dx%d grid, transient mask, encode_a / encode_c / "
                                   "encode_random) x (%d+%d) split over %d rank(s): rays sharded, feature rows all-gathered, decoder / encoders / mask "
                                   "network replicated, one flat gradient all-reduce; --train-precision %s" % (R, side, side, nc, ni, world, a.train_precision),
                       "rays_total": R, "n_samples": nc, "n_importance": ni, "parallelism": "one batch's rays sharded %d-way" % world,
                       "train_precision": a.train_precision},
            "roofline": {"bound": "mfma",
                         "kernel": ("render_rays_train16_kernel + mlp_backward16_kernel + wgrad_kernel" if exact else
                                    "render_rays_train_h2_kernel + mlp_backward_h2_kernel + wgrad_h2_kernel"),
                         "achieved": issued, "peak": peak, "unit": "TFLOP/s", "frac": issued / peak, "traffic": None, "traffic_source": None,
                         "fp32_work_tflops": achieved,
                         "note": "whole step per GPU: the MLP work of this rank's rays / step time (decoder, encoders, mask network and Adam included in the "
                                 "time).  f32: 3 x the forward's algorithmic FLOPs on the fp32 MFMA.  auto: the ISSUED fp16 MFMA work (9.02 x the "
                                 "forward's algorithmic FLOPs) against the nominal 2.5 PFLOP/s -- the three big kernels of that step are bound by their activation / "
                                 "delta rows in HBM, not by the matrix cores (DESIGN 3.5); fp32_work_tflops = the algorithmic fp32 work"},
            "loss": float(last), "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30, "sections": sections}
    if use_dist and world > 1:
        chk = torch.tensor([float(last), float(sum(p.detach().double().sum() for p in sysm.parameters()))], dtype=torch.float64, device=dev)
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        line["replicas_identical"] = bool(all(torch.equal(allc[0], c) for c in allc))
    return line


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2])
    a = parse()
    # stdout carries the ONE JSON line and nothing else: libraries that print to fd 1 (RCCL's version banner at communicator creation in this image)
    # go to stderr for the life of the process; the line itself is written to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
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
    from crnerf_amd import ops, parallel
    from crnerf_amd.models.linearStyleTransfer import style_net
    from crnerf_amd.parallel import PeerExchange, decode_sharded
    exchange = PeerExchange() if (use_dist and world > 1 and a.peer_exchange) else None
    coll = collectives_record(dist, dev, world, rank, test_backend) if use_dist else None
    if use_dist:
        parallel.collect_collective_times(True)     # event pairs around every collective of the timed steps

    def finish(line):
        """rank 0 prints the ONE JSON line; every rank leaves the process group."""
        if use_dist:
            times = parallel.collective_times_ms()
            if rank == 0:
                coll["per_call_ms"] = times
                coll["note"] = ("HIP events on the launch stream around each collective, averaged over warm-up + timed steps of rank 0; "
                                "expected on xGMI (DESIGN section 4): ~10-25 us per tiny all-reduce, all-gather ~ 12 B/pixel / link rate + ~20 us")
                line["collectives"] = coll
        if rank == 0:
            json_out.write(json.dumps(line) + "\n")
            json_out.flush()
        if exchange is not None:
            exchange.check()
            exchange.close()
        if use_dist:
            dist.destroy_process_group()

    if a.scaling == "strong":
        if a.workload == "configs2":
            if "--precision" not in sys.argv:
                a.precision = "bf16"                 # BASELINE configs[2]: "1x MI355X bf16"
            return finish(strong_configs2(a, dev, world, rank, use_dist, dist, exchange))
        if a.workload == "configs4":
            return finish(strong_configs4(a, dev, world, rank, use_dist, dist, exchange))
        return finish(strong_configs3(a, dev, world, rank, use_dist, dist))

    R = a.rays
    W = int(R ** 0.5)
    while R % W:
        W -= 1
    grid_hw = (R // W, W)
    st_c, st_f, dst = synth.mlp_state(1, 3.0, 1.0), synth.mlp_state(2, 3.0, 1.0), synth.decoder_state(3)
    rays_np = synth.rays(R, seed=rank, H=grid_hw[0], W=grid_hw[1])
    to_dev = lambda s: {k: torch.from_numpy(v).to(dev) for k, v in s.items()}  # noqa: E731

    class Args:
        nerf_out_dim, img_wh, pertubeCord = 64, [grid_hw[1], grid_hw[0]], False

    # The timed step calls the DROP-IN function -- crnerf_amd.models.rendering.render_rays_cross_ray with the reference's positional signature
    # (models/rendering.py:50-63, the call eval.py:39-52 makes) on NeRF_sigma modules -- not the C-ABI wrapper under it: the
    # number includes the mirror's own host work (type checks, packed-weight cache lookup, result dict).  `timing.direct_c_abi` times the wrapper
    # alone afterwards, untimed by the contract, so the line shows what the mirror costs.
    from crnerf_amd.models.nerf import NeRF_sigma, PosEmbedding
    from crnerf_amd.models.rendering import render_rays_cross_ray
    with torch.no_grad():
        models = {"coarse": NeRF_sigma("coarse", Args(), in_channels_xyz=93, in_channels_dir=27).to(dev),
                  "fine": NeRF_sigma("fine", Args(), in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, in_channels_a=48, encode_random=True).to(dev)}
        models["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in st_c.items()})
        models["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in st_f.items()})
        for m in models.values():
            m.eval().requires_grad_(False)
        embeddings = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
        pc, pf = models["coarse"].packed_weights(a.precision), models["fine"].packed_weights(a.precision)   # the direct-ABI leg's packs (the modules cache the same)
        net = style_net(Args()).to(dev)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in dst.items()})
        rays = torch.from_numpy(rays_np).to(dev)
        # the appearance encoder hands the decoder a pixel-major [1024,64] grid viewed as NCHW (zero-copy, models/linearStyleTransfer.py)
        style = torch.rand(1024, 64, generator=torch.Generator().manual_seed(0)).to(dev).view(1, 32, 32, 64).permute(0, 3, 1, 2)
        z_steps, u_steps = torch.linspace(0, 1, NC, device=dev), torch.linspace(0, 1, NI, device=dev)  # rendering.py:160, :27

        # event pairs around the render launch of every step (warm-up steps included: the warm-up runs the IDENTICAL host path, so the
        # first hipEventCreate / hipEventRecord, their code pages and torch's lazy event pool are paid before the timed region -- on a
        # cold box the round-4 line lost 11 ms = 0.56 ms per step to exactly that), one more event at the end of every
        # step, and a host stamp after every enqueue: the line carries the per-step device and host times it was computed from
        n_ev = a.warmup + a.steps
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        host_stamp = [0.0] * n_ev

        r4_warmup = os.environ.get("CRNERF_BENCH_R4_WARMUP") == "1"   # diagnosis only: warm-up steps record no events, as bench.py did up to round 4

        args_obj = Args()

        def step(i, direct=False):
            quiet = (r4_warmup and i < a.warmup) or direct      # (the direct leg records no events: it must not re-stamp the timed steps')
            if not quiet:
                ev[i][0].record()
            if direct:
                out = ops.render_rays(pc, pf, rays, NC, NI, z_steps=z_steps, u=u_steps, precision=a.precision)
            else:     # eval.py:39-52: rays, ts, N_samples, use_disp, perturb = 0, noise_std = 0, N_importance, chunk, white_back, test_time=True
                out = render_rays_cross_ray(models, embeddings, rays, None, NC, False, 0, 0, NI, 32 * 1024, False, test_time=True, args=args_obj,
                                            precision=a.precision)
            if not quiet:
                ev[i][1].record()
            feat = out["feature_fine"]
            if use_dist:
                rgb = decode_sharded(net, feat, style, gather=True, equal_shards=True, exchange=exchange, check_exchange=False)   # checked once after the timed region
            else:
                rgb = net(feat.t().reshape(1, 64, *grid_hw), style)
            if not quiet:
                ev[i][2].record()
            if not direct:
                host_stamp[i] = time.perf_counter()
            return rgb

        def fence():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        import gc
        for i in range(a.warmup):
            last_rgb = step(i)
        gc.collect()
        gc.disable()        # a generation-2 collection of torch's heap is a ~10 ms host pause: exposed if it lands on the first steps, before work is queued
        try:
            fence()
            t0 = time.perf_counter()
            for i in range(a.warmup, n_ev):
                last_rgb = step(i)
            t_enq = time.perf_counter()
            fence()
            dt = time.perf_counter() - t0
            # the same steps through the C-ABI wrapper alone (ops.render_rays: what rounds 1-5 timed), outside the contract's timed region
            td0 = time.perf_counter()
            if not use_dist:                       # (N = 1 only: under torch.distributed these steps would add collectives to the record)
                for i in range(a.warmup, n_ev):
                    step(i, direct=True)
            td_enq = time.perf_counter()
            fence()
            dt_direct = time.perf_counter() - td0
        finally:
            gc.enable()
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        tev = ev[a.warmup:]
        kern_each = [s.elapsed_time(e) for s, e, _ in tev]
        step_each = [s.elapsed_time(d) for s, _, d in tev]                      # render launch .. end of the decode, on the device
        gap_each = [tev[i][2].elapsed_time(tev[i + 1][0]) for i in range(len(tev) - 1)]   # device idle between two steps (host-starved if > ~0.01)
        hs = [t0] + host_stamp[a.warmup:]
        host_each = [(hs[i + 1] - hs[i]) * 1e3 for i in range(len(hs) - 1)]    # host time to enqueue one step

        def _stats(v):
            v2 = sorted(v)
            return {"min": v2[0], "median": v2[len(v2) // 2], "max": v2[-1]} if v2 else None
        timing = {"step_ms_each": [round(x, 4) for x in step_each], "kernel_ms_each": [round(x, 4) for x in kern_each],
                  "device_gap_ms_each": [round(x, 4) for x in gap_each], "host_enqueue_ms_each": [round(x, 4) for x in host_each],
                  "step_ms": _stats(step_each), "kernel_ms": _stats(kern_each), "device_gap_ms": _stats(gap_each), "host_enqueue_ms": _stats(host_each),
                  "host_enqueue_total_ms": (t_enq - t0) * 1e3, "wall_ms": dt * 1e3, "device_span_ms": tev[0][0].elapsed_time(tev[-1][2]),
                  "timed_through": "crnerf_amd.models.rendering.render_rays_cross_ray (the reference's signature, eval.py:39-52) + style_net.forward",
                  "direct_c_abi": None if use_dist else {"ms_per_step": dt_direct / a.steps * 1e3, "host_enqueue_ms_per_step": (td_enq - td0) / a.steps * 1e3,
                                   "note": "the same steps through ops.render_rays (the ctypes wrapper of crnerf_render_rays_*: what rounds 1-5 timed), "
                                           "run after the timed region"},
                  "gc": "gc.collect() + gc.disable() around the timed region (a generation-2 collection is a ~10 ms host pause; a training loop that "
                        "does not do the same pays it now and then -- the device queue hides it once a few steps are in flight)",
                  "note": "HIP events on the launch stream (render start / render end / decode end of every timed step) and perf_counter stamps after "
                          "every enqueue; wall_ms = the contract's barrier-to-barrier time; wall_ms - device_span_ms = launch latency of the first "
                          "step + the final synchronize"}
        kern_ms = sum(kern_each) / a.steps
        rgb_sums = None
        if use_dist and test_backend:      # test hook only: every rank must hold the same gathered image
            mine = torch.tensor([float(last_rgb.double().sum()), float(last_rgb.shape[-1])], dtype=torch.float64, device=dev)
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            rgb_sums = [p.tolist() for p in parts]

    if rank == 0:
        flops = FLOP_PER_POINT * (NC + NC + NI) * R
        achieved = flops / (kern_ms * 1e-3) / 1e12
        bf16 = a.precision == "bf16"
        x3 = a.precision in ("f32x3", "f32h2")
        h2 = a.precision == "f32h2"
        peak = PEAK_BF16_MFMA_TFLOPS if (bf16 or x3) else PEAK_F32_MFMA_TFLOPS
        kernel = "render_rays_bf16p_kernel" if bf16 else "render_rays16_kernel"
        if x3:   # the roofline of this mode is the bf16 pipe, priced with the bf16 MFMA work the kernel ISSUES: six piece products per fp32 product
            kernel = "render_rays_h2_kernel" if h2 else "render_rays_x3_kernel"
            flops = (3.0 if h2 else 6.0) * (7296.0 / 7248.0) * flops      # (the fp16 MFMA's dense peak is the bf16 MFMA's)
            achieved = flops / (kern_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "render_rays_hbm_bytes.json")   # written from a rocprofv3 --pmc pass (profiles/README.md)
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get(a.precision, {}).get("hbm_bytes_per_launch_%d_rays" % R)
        line = {
            "metric": "rays/sec (64+128 samples, 8-layer W=256 MLP)", "value": world * R * a.steps / dt, "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic", "meets": MEETS[a.precision],
            "config": {"workload": "BASELINE configs[%d]%s: %d rays x (%d coarse + %d fine) per GPU, NeRF_sigma 8x256 coarse+fine, "
                                   "fused render_rays + cross-ray decode of the %dx%d feature grid"
                                   % (2 if bf16 else 1, " arithmetic (bf16 MFMA operands, fp32 accumulate) on the configs[1] ray batch" if bf16 else
                                      (" in fp32 on the fp16 matrix cores (two-piece fp16 splits of every fp32 operand, three MFMAs per product, fp32 accumulation; "
                                       "roofline.achieved counts the ISSUED fp16 MFMA work = 3.02 x the algorithmic fp32 FLOPs)" if h2 else
                                       " in fp32 on the bf16 matrix cores (three-piece bf16 splits of every fp32 operand, six MFMAs per product, fp32 accumulation; "
                                       "roofline.achieved counts the ISSUED bf16 MFMA work = 6.04 x the algorithmic fp32 FLOPs)" if x3 else ""),
                                      R, NC, NI, grid_hw[0], grid_hw[1]),
                       "rays_per_gpu": R, "n_samples": NC, "n_importance": NI,
                       "parallelism": "rays sharded %d-way, weights replicated" % world,
                       "reductions": "none" if world == 1 else ("peer windows (HIP IPC)" if exchange is not None else "RCCL all-reduce")},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_source": (None if traffic is None else "profiles/render_rays_hbm_bytes.json: a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                                          "pass over this command (profiles/README.md), not measured in this run"),
                         "kernel_ms": kern_ms, "flops_per_launch": flops},
            "timing": timing,
        }
        if world == 1 and not a.no_cpu_baseline:
            # untimed legs; the headline line above must be printed whatever happens in them
            try:
                line["cpu_baseline"] = cpu_baseline(rays_np, st_c, st_f, dst, grid_hw, style.cpu().contiguous())
            except Exception as e:   # noqa: BLE001
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                line["parity"], line["extra"] = evidence(a, dev, rays, rays_np, st_c, st_f, grid_hw, style, z_steps, u_steps, Args)

                line["extra"]["step_ms_each"], line["extra"]["host_enqueue_ms_each"] = timing["step_ms_each"], timing["host_enqueue_ms_each"]
                for kx in ("f32x3_kernel", "f32h2_kernel"):
                    if not bf16 and kx in line["extra"]:
                        line["extra"][kx]["speedup_vs_fp32_mfma_kernel"] = kern_ms / line["extra"][kx]["kernel_ms"]
            except Exception as e:   # noqa: BLE001
                line["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                # configs[3]'s training step at the reference's own batch (command/train.sh:24: 1,024 rays) and at 16,384 rays: the whole
                # train.sh configuration (batcher, encoders, mask network, grad-mode render, decodes, loss, backward, Adam), training defaults
                import argparse
                import gc
                gc.collect()
                gc.freeze()     # what the legs above left on the heap (oracle tensors, fixtures) is not this step's garbage: without this every
                                # generation-2 collection during the host-bound 1,024-ray steps walks it (6.3-7.0 ms per step against 5.3-5.8 in a process that only trains)
                for tr, st in ((1024, 30), (16384, 6)):
                    ta = argparse.Namespace(**dict(vars(a), train_rays=tr, steps=st, warmup=5, train_precision="auto"))
                    tl = strong_configs3(ta, dev, 1, 0, False, None)
                    line.setdefault("extra", {})["train_step_%d" % tr] = {"ms_per_step": tl["ms_per_step"], "rays_per_s": tl["value"], "dtype": tl["dtype"], "steps": st,
                                                           "fp32_work_tflops": tl["roofline"]["fp32_work_tflops"], "workload": tl["config"]["workload"]}
                    if tr == 1024:
                        # the same step with its two host threads (caller + autograd worker) inside one L3 domain (crnerf_amd.hostpin, opt-in):
                        # this size is bound by those two handing the GIL over, and where they run decides 5.3 against 6 ms
                        from crnerf_amd import hostpin
                        pinned = hostpin.pin_step_threads(dev)
                        try:
                            tp = strong_configs3(ta, dev, 1, 0, False, None)
                        finally:
                            hostpin.unpin_host_threads(pinned)
                        line["extra"]["train_step_1024"]["step_threads_in_one_l3_domain"] = {
                            "ms_per_step": tp["ms_per_step"], "rays_per_s": tp["value"], "cpus": sorted(pinned["cpus"]) if pinned else None,
                            "note": "crnerf_amd.hostpin.pin_step_threads(device) for this leg only (restored afterwards)"}
                gc.unfreeze()
                from crnerf_amd import autograd as _AG
                _AG.set_training_forward_precision(None)
                _AG.set_wgrad_precision(None)
            except Exception as e:   # noqa: BLE001
                line.setdefault("extra", {})["train_step_error"] = "%s: %s" % (type(e).__name__, e)
        if rgb_sums is not None:
            line["test_rgb_checksum_per_rank"] = rgb_sums
    finish(line if rank == 0 else None)


if __name__ == "__main__":
    main()
