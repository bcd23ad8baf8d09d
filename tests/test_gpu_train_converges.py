"""Does the drop-in TRAIN like the reference?  tests/golden/g15_trained.npz holds the loss / PSNR curve of the reference's own
modules trained on the procedural scene of tests/_procedural_scene.py (CPU, 1,500 steps, command/train.sh's configuration: encode_a,
encode_c, encode_random, use_mask, Adam lr 5e-4, perturb = 1, noise_std = 1, 32+32 samples).  Here pipeline.TrainingSystem -- the
NeRFSystem mirror on the HIP twins: fused training renderer with in-kernel jitter / noise, four decodes, three encoder passes, the
CGNet mask, the fused CRNeRF loss, backward through every twin, torch's fused Adam -- is trained on the IDENTICAL images from its own
default initialisation.  Random streams differ (initial weights, jitter, noise, image order), so the curves are compared as curves:
the window means of the fine-image PSNR must track the reference's within 1.5 dB, and the loss must fall like the reference's."""
import numpy as np
import pytest
import torch

import _procedural_scene as S
from crnerf_amd import optim, pipeline

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("mode", ["f32", "f32x3", "auto", "auto+FlatAdam", "bf16", "bf16+recompute"])
def test_training_system_learns_the_scene_like_the_reference(golden, mode):
    """mode f32x3: forward, data gradient and weight gradients on the bf16 matrix cores at fp32 accuracy (three-piece splits; DESIGN 3.4b / 3.5:
    set_training_forward_precision("f32x3") + set_wgrad_precision("bf16x3")) -- held to the same curve.  bf16 / bf16+recompute (round 6, VERDICT r5
    #6): the opt-in MIXED-precision twins (set_training_precision("bf16"): bf16 operands and bf16 saved rows, fp32 accumulation; not fp32-accurate)
    held to the same curve and bars -- whether that mode also ends where the fp32-accurate modes end is tools/train_precision_study.py's question
    (profiles/r6/mixed_precision_convergence.txt)."""
    from crnerf_amd import autograd as AG
    mode, _, opt = mode.partition("+")          # "+FlatAdam": the optimiser step as one HIP launch (crnerf_amd/optim.py) instead of torch's multi-tensor Adam
    mixed = mode == "bf16"
    AG.set_training_precision("bf16" if mixed else "f32")
    AG.set_training_recompute(mixed and opt == "recompute")
    AG.set_training_forward_precision(None if mixed else mode)
    AG.set_wgrad_precision(None if mixed else "f32" if mode == "f32" else ("bf16x3" if mode == "f32x3" else None))   # "auto": the defaults -- h2 forward / data gradient with the x3 safety net, f16x2 weight gradients
    try:
        _run(golden, opt == "FlatAdam")
    finally:
        AG.set_training_precision("f32")
        AG.set_training_recompute(False)
        AG.set_training_forward_precision(None)
        AG.set_wgrad_precision(None)


def _run(golden, flat_adam=False):
    ref = golden("g15_trained")["train_log"]                    # [1500, 3]: step, loss, psnr_fine of the batch
    torch.manual_seed(11)
    rng = np.random.default_rng(7)
    data = S.make_dataset(rng)[:S.N_IMAGES]                     # the same generator state as the fixture's run: identical images
    hp = S.hparams()
    sysm = pipeline.TrainingSystem(hp, device=DEV)
    if flat_adam:
        # utils/__init__.py:24-33, the 'adam' branch, over everything the system trains
        opt = optim.FlatAdam(optim.get_parameters(sysm.models_to_train), lr=hp.lr, eps=1e-8, weight_decay=hp.weight_decay)
    else:
        opt = torch.optim.Adam(sysm.parameters(), lr=hp.lr, eps=1e-8, fused=True)
    idx = torch.arange(S.SIDE * S.SIDE, device=DEV)
    batches = [dict(rays=b["rays"].to(DEV), ts=b["ts"].to(DEV), rgbs=b["rgbs"].to(DEV), whole_img=S.whole_image(b["rgbs"]).to(DEV), rgb_idx=idx,
                    img_wh=(S.SIDE, S.SIDE)) for b in data]
    steps = 600
    log = np.zeros((steps, 2))
    for step in range(steps):
        b = batches[int(rng.integers(S.N_IMAGES))]
        opt.zero_grad(set_to_none=True)
        loss, loss_d, res = sysm.training_step(b)
        loss.backward()
        opt.step()
        with torch.no_grad():
            log[step] = float(loss), float(-10 * torch.log10(((res["rgb_fine"] - b["rgbs"]) ** 2).mean()))
    assert np.isfinite(log).all()
    windows = ((0, 50), (200, 300), (300, 400), (400, 600))
    ours = [log[a:b, 1].mean() for a, b in windows]
    theirs = [ref[a:b, 2].mean() for a, b in windows]
    print("PSNR window means  HIP %s  reference %s" % (np.round(ours, 2), np.round(theirs, 2)))
    print("loss window means  HIP %s  reference %s" % (np.round([log[a:b, 0].mean() for a, b in windows], 5), np.round([ref[a:b, 1].mean() for a, b in windows], 5)))
    for o, t, w in zip(ours, theirs, windows):
        assert abs(o - t) <= 1.5, (w, o, t)
    assert ours[-1] - ours[0] >= 0.6 * (theirs[-1] - theirs[0]) > 1.0          # it learns, at the reference's pace
    assert log[400:, 0].mean() <= 1.25 * ref[400:600, 1].mean()                 # and the total loss (all seven terms) falls like the reference's
