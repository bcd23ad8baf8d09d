"""Measurement tool (GPU box): where does training END in each precision mode?  (VERDICT r5 "do this" #6: north_star's bar for a precision mode
is PSNR within 0.05 dB of the reference; the mixed-precision training mode halves the 65,536-ray step and had no convergence evidence.)

The recipe is the one that produced tests/golden/g16_trained.npz with the REFERENCE's modules (tests/golden/make_golden_trained.py --steps 5000:
command/train.sh's configuration on the procedural scene of tests/_procedural_scene.py, Adam lr 5e-4, cosine schedule stepped per "epoch" of
steps / 20), run here through pipeline.TrainingSystem in each mode with several seeds.  Reported per run: mean training PSNR of the last 500 steps and
the PSNR of the HELD-OUT view (fixture rays, appearance of training image 0, fp32 inference at 64+128 -- eval.py's recipe) against the scene; the
fixture holds the reference-trained checkpoint's own held-out render, i.e. the number the reference's training reaches.  Training is stochastic
(initialisation, jitter, noise, image order), so modes are compared through the spread over seeds, not run against run.

    python tools/train_precision_study.py [steps=5000] [seeds=3] [modes=auto,bf16,bf16+recompute]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _procedural_scene as S
from crnerf_amd import autograd as AG, optim, pipeline

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
MODES = (sys.argv[3] if len(sys.argv) > 3 else "auto,bf16,bf16+recompute").split(",")
DEV = "cuda:0"
fix = np.load(os.path.join(ROOT, "tests", "golden", "g16_trained.npz"))
gt = torch.from_numpy(fix["gt"])
ref_psnr = float(-10 * torch.log10(((torch.from_numpy(fix["ref__64_128__rgb_fine"]) - gt) ** 2).mean()))
ref_train = float(fix["train_log"][-500:, 2].mean())


def psnr(a, b):
    return float(-10 * torch.log10(((a - b) ** 2).mean()))


def run(mode, seed):
    base, _, opt_ = mode.partition("+")
    mixed = base == "bf16"
    AG.set_training_precision("bf16" if mixed else "f32")
    AG.set_training_recompute(opt_ == "recompute")
    AG.set_training_forward_precision(None if mixed else base)
    AG.set_wgrad_precision(None if (mixed or base == "auto") else "f32")
    torch.manual_seed(1000 + seed)
    rng = np.random.default_rng(7)                                  # the fixture's generator state: identical images
    data = S.make_dataset(rng)
    train, test = data[:S.N_IMAGES], data[S.N_IMAGES]
    order = np.random.default_rng(100 + seed)                       # the image order is part of what a seed changes
    hp = S.hparams()
    sysm = pipeline.TrainingSystem(hp, device=DEV)
    opt = optim.FlatAdam(optim.get_parameters(sysm.models_to_train), lr=hp.lr, eps=1e-8, weight_decay=hp.weight_decay)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=hp.num_epochs, eta_min=1e-8)
    per_epoch = max(STEPS // hp.num_epochs, 1)
    idx = torch.arange(S.SIDE * S.SIDE, device=DEV)
    batches = [dict(rays=b["rays"].to(DEV), ts=b["ts"].to(DEV), rgbs=b["rgbs"].to(DEV), whole_img=S.whole_image(b["rgbs"]).to(DEV), rgb_idx=idx,
                    img_wh=(S.SIDE, S.SIDE), image_id=k) for k, b in enumerate(train)]
    tail = []
    t0 = time.perf_counter()
    for step in range(STEPS):
        b = batches[int(order.integers(S.N_IMAGES))]
        opt.zero_grad(set_to_none=True)
        loss, _, res = sysm.training_step(b)
        loss.backward()
        opt.step()
        if (step + 1) % per_epoch == 0:
            sched.step()
        if step >= STEPS - 500:
            tail.append(-10 * torch.log10(((res["rgb_fine"].detach() - b["rgbs"]) ** 2).mean()))
    train_psnr = float(torch.stack(tail).mean())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # held-out view: eval.py's recipe in fp32 (perturb = 0, noise_std = 0, 64+128), appearance of training image 0
    sysm.eval()
    with torch.no_grad():
        a = sysm.enc_a((S.whole_image(train[0]["rgbs"]).to(DEV) + 1) / 2)
        res = pipeline.batched_inference(sysm.models, sysm.embeddings, test["rays"].to(DEV), test["ts"].to(DEV), 64, 128, False, 1 << 15, False,
                                         args=hp, a_embedded_from_img=a, precision="f32")
        rgb = pipeline.decode_image(sysm.models, res, S.SIDE, S.SIDE, a).cpu()
        # the training views, deterministically (perturb = 0, noise_std = 0), each with its OWN appearance: what the model has fitted, without the
        # jitter / noise of the training-time PSNR and without the held-out view's unseen appearance
        fit = []
        for k, b in enumerate(train):
            ak = sysm.enc_a((S.whole_image(b["rgbs"]).to(DEV) + 1) / 2)
            rk = pipeline.batched_inference(sysm.models, sysm.embeddings, b["rays"].to(DEV), b["ts"].to(DEV), 64, 128, False, 1 << 15, False,
                                            args=hp, a_embedded_from_img=ak, precision="f32")
            if k % 3 != 1:                                        # (every third image carries a transient rectangle the model is meant to ignore)
                fit.append(psnr(pipeline.decode_image(sysm.models, rk, S.SIDE, S.SIDE, ak).cpu(), b["rgbs"]))
    return train_psnr, psnr(rgb, test["rgbs"]), dt, float(np.mean(fit))


print("recipe: %d steps; reference-trained checkpoint (tests/golden/g16_trained.npz, 5000 steps on the CPU): training PSNR of its last 500 steps %.2f dB, "
      "held-out view %.2f dB" % (STEPS, ref_train, ref_psnr), flush=True)
summary = {}
for mode in MODES:
    rows = []
    for seed in range(SEEDS):
        tr, ho, dt, fit = run(mode, seed)
        rows.append((tr, ho, fit))
        print("mode %-16s seed %d: training PSNR (last 500 steps) %6.2f dB   fitted views (deterministic eval) %6.2f dB   held-out %6.2f dB   %5.1f s = %.2f ms per step"
              % (mode, seed, tr, fit, ho, dt, dt / STEPS * 1e3), flush=True)
    r = np.array(rows)
    summary[mode] = r
    print("mode %-16s mean over %d seeds: training %6.2f +- %.2f dB   fitted views %6.2f +- %.2f dB   held-out %6.2f +- %.2f dB (min %.2f, max %.2f)"
          % (mode, SEEDS, r[:, 0].mean(), r[:, 0].std(), r[:, 2].mean(), r[:, 2].std(), r[:, 1].mean(), r[:, 1].std(), r[:, 1].min(), r[:, 1].max()), flush=True)
AG.set_training_precision("f32")
AG.set_training_recompute(False)
AG.set_training_forward_precision(None)
AG.set_wgrad_precision(None)
base = summary.get("auto")
if base is not None:
    for mode, r in summary.items():
        if mode == "auto":
            continue
        d_ho, d_tr, d_fit = r[:, 1].mean() - base[:, 1].mean(), r[:, 0].mean() - base[:, 0].mean(), r[:, 2].mean() - base[:, 2].mean()
        for what, d, col in (("held-out", d_ho, 1), ("fitted views", d_fit, 2)):
            se = float(np.sqrt(base[:, col].var(ddof=1) / len(base) + r[:, col].var(ddof=1) / len(r))) if len(r) > 1 else float("nan")
            print("%-16s vs auto, %-12s: %+.3f dB (standard error of the difference %.3f dB over %d + %d seeds) -> %s"
                  % (mode, what, d, se, len(base), len(r), "within north_star's 0.05 dB" if abs(d) <= 0.05 else
                     ("not resolved from zero (< 2 standard errors), outside 0.05 dB" if abs(d) <= 2 * se else "DIFFERENT (> 2 standard errors)")), flush=True)
        print("%-16s vs auto, training PSNR of the last 500 steps: %+.3f dB" % (mode, d_tr), flush=True)
