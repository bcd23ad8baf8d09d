"""Generates tests/golden/g10_loss.npz and g11_batcher.npz by IMPORTING THE REFERENCE in the build container
(CRNeRFLoss losses.py:42-94 and the grid-sample batcher
datasets/phototourism_mask_grid_sample.py:241-275).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py

Data only: seeded inputs, the reference's outputs and (for the loss) the gradients torch autograd derives from the
reference's expression.  The dataset module imports torchvision / kornia / pandas-backed readers at import time; those
imports are stubbed in THIS process only, and the reference's own __getitem__ is run on a stub `self` that carries
exactly the attributes its train branch reads.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote", name)


class HP:   # opt.py defaults of the fields CRNeRFLoss reads (:15-22, :96-106), maskrd made non-zero so r_md is exercised
    maskrs_max, maskrs_min, maskrs_k, maskrd = 5e-2, 6e-3, 1e-3, 1e-3
    weightKL, weightRecA, weightcontent, mse_on_appearance = 1e-5, 1e-3, 1e-4, False


def loss_goldens():
    from losses import CRNeRFLoss   # reference
    g = torch.Generator().manual_seed(5)
    R = 4096
    out = {}
    for tag, mse_a, with_mask, with_fine in (("full", False, True, True), ("mse_a", True, True, True), ("nomask", False, False, True),
                                            ("coarse_only", False, False, False)):
        hp = HP()
        hp.mse_on_appearance = mse_a
        rc = torch.rand(R, 3, generator=g, requires_grad=True)
        rf = torch.rand(R, 3, generator=g, requires_grad=True)
        tg = torch.rand(R, 3, generator=g)
        mask = torch.rand(R, 1, generator=g, requires_grad=True)
        a = (torch.randn(1, 64, 8, 8, generator=g) * 0.3).requires_grad_()
        ar = torch.randn(1, 64, 8, 8, generator=g) * 0.3
        arr = (torch.randn(1, 64, 8, 8, generator=g) * 0.3).requires_grad_()
        cw = torch.randn(1, 16, 8, 8, generator=g).requires_grad_()
        cwith = torch.randn(1, 16, 8, 8, generator=g).requires_grad_()
        inputs = {"rgb_coarse": rc, "a_embedded": a, "a_embedded_random": ar, "a_embedded_random_rec": arr,
                  "content_wo_a_embed": cw, "content_with_a_embed": cwith}
        if with_fine:
            inputs["rgb_fine"] = rf
        if with_mask:
            inputs["out_mask"] = mask
        step = 1234
        crit = CRNeRFLoss(hp, coef=1)
        ret, ann = crit(inputs, tg, hp, step)
        total = sum(v for v in ret.values())
        total.backward()
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad  # noqa: E731
        out.update({tag + "__" + k: v for k, v in dict(
            rgb_coarse=rc, rgb_fine=rf, targets=tg, mask=mask, a=a, a_rand=ar, a_rand_rec=arr, c_wo=cw, c_with=cwith, step=step, ann=ann,
            d_rgb_coarse=z(rc), d_rgb_fine=z(rf), d_mask=z(mask), d_a=z(a), d_a_rand_rec=z(arr), d_c_wo=z(cw), d_c_with=z(cwith),
            **{"loss_" + k: v for k, v in ret.items()}).items()})
        out[tag + "__keys"] = np.array(list(ret.keys()))
    save("g10_loss", **out)


def batcher_goldens():
    # stub the import-time dependencies of datasets/ (never called by the train branch of __getitem__)
    for name in ("torchvision", "torchvision.transforms", "kornia", "pandas"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["kornia"].create_meshgrid = None
    pkg = types.ModuleType("datasets")
    pkg.__path__ = ["/root/reference/datasets"]
    sys.modules["datasets"] = pkg                 # bypass datasets/__init__.py (it imports every dataset flavour)
    from datasets import global_val
    spec = importlib.util.spec_from_file_location("datasets.phototourism_mask_grid_sample",
                                                  "/root/reference/datasets/phototourism_mask_grid_sample.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    DS = mod.PhototourismDataset

    rng = np.random.default_rng(3)
    wh = np.array([[37, 23], [64, 48], [51, 80], [29, 31]], dtype=np.int64)     # (w, h) of four "images"
    n = int((wh[:, 0] * wh[:, 1]).sum())
    all_rays = torch.from_numpy(rng.standard_normal((n, 9)).astype(np.float32))
    all_rays[:, 8] = torch.from_numpy(np.repeat(np.arange(4), wh[:, 0] * wh[:, 1]).astype(np.float32))
    all_rgbs = torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    out = {"wh": wh, "all_rays": all_rays, "all_rgbs": all_rgbs}
    for tag, batch, anneal, min_scale, epoch, idx in (("a", 1024, -1, 0.25, 0, 7), ("b", 256, 1e-3, 0.1, 3, 11), ("c", 4096, -1, 0.9, 1, 0)):
        stub = types.SimpleNamespace(split="train", iterations=50, all_imgs=[None] * 4, all_imgs_wh=torch.from_numpy(wh),
                                     batch_size=batch, scale_anneal=anneal, min_scale=min_scale, all_rays=all_rays, all_rgbs=all_rgbs)
        global_val.current_epoch = epoch
        torch.manual_seed(100 + idx)              # the reference draws scale / offsets from torch's global CPU generator
        s = DS.__getitem__(stub, idx)
        out.update({tag + "__" + k: v for k, v in dict(
            batch=batch, anneal=anneal, min_scale=min_scale, epoch=epoch, idx=idx, torch_seed=100 + idx, iterations=50,
            rays=s["rays"], ts=s["ts"], rgbs=s["rgbs"], rgb_idx=s["rgb_idx"], uv_sample=s["uv_sample"], img_wh=s["img_wh"],
            min_scale_cur=s["min_scale_cur"]).items()})
    save("g11_batcher", **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    loss_goldens()
    batcher_goldens()
