"""Generates tests/golden/g15_trained.npz: a TRAINED CR-NeRF checkpoint and the reference's outputs on it.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trained.py [--steps 1500] [--threads 4]

north_star asks for parity "on identical rays/checkpoint"; every other fixture uses seeded random networks.  This script
IMPORTS THE REFERENCE in the build container and trains its own modules -- NeRF_sigma x2 (models/nerf.py), style_net
(models/linearStyleTransfer.py), encoder_sameoutputsize x2 (enc_a, enc_cont), Context_Guided_Network (implicit_mask) -- with
its own render_rays_cross_ray (models/rendering.py) and CRNeRFLoss (losses.py) on a small procedural scene, under
command/train.sh's configuration (encode_a, encode_c, encode_random, use_mask, adam lr 5e-4, 1,024-ray grid batches,
perturb = 1, noise_std = 1).  pytorch_lightning is not installed here, so the three Lightning hooks the run needs
(NeRFSystem.forward / decode / training_step, train_mask_grid_sample.py:127-226,268-290) are driven by the small loop below;
every arithmetic step is the reference's code.

The result is saved the way the reference saves it -- a Lightning checkpoint {'state_dict': {'nerf_coarse.*', 'nerf_fine.*',
'decoder.*', 'enc_a.*', 'enc_cont.*', 'implicit_mask.*'}, 'optimizer_states', 'hyper_parameters', ...} -- read back with the
reference's own utils.load_ckpt (utils/__init__.py:67-88) into fresh reference modules, and rendered by the reference on a
held-out view at eval.py's settings.  The fixture holds DATA only: the state_dict arrays, the held-out rays / style image, and the
reference's outputs (style feature, render dict at 64+128 and 256+256, decoded image) plus the reference's own fp64 - fp32
difference on the same inputs (its conditioning on trained weights).
"""
import argparse
import math
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
OUT = os.path.dirname(os.path.abspath(__file__))

_k, _kf = types.ModuleType("kornia"), types.ModuleType("kornia.filters")
_kf.filter2d = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("filter2d is not reachable at n_blocks=0"))
_k.filters = _kf
sys.modules.setdefault("kornia", _k)
sys.modules.setdefault("kornia.filters", _kf)
for _name in ("torch_optimizer", "torchvision", "torchvision.transforms", "matplotlib", "matplotlib.pyplot", "cv2", "PIL", "PIL.Image"):
    sys.modules.setdefault(_name, types.ModuleType(_name))    # import-time dependencies of utils/ that load_ckpt never calls
sys.modules["cv2"].COLORMAP_JET = 2
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["PIL"].Image = sys.modules["PIL.Image"]

from losses import CRNeRFLoss  # noqa: E402  (reference)
from models.lightweight_seg import Context_Guided_Network  # noqa: E402  (reference)
from models.linearStyleTransfer import encoder_sameoutputsize, style_net  # noqa: E402  (reference)
from models.nerf import NeRF_sigma, PosEmbedding  # noqa: E402  (reference)
from models.rendering import render_rays_cross_ray  # noqa: E402  (reference)

sys.path.insert(0, os.path.join(ROOT, 'tests'))
from _procedural_scene import N_IMAGES, SIDE, hparams, make_dataset, whole_image  # noqa: E402


# ---------------------------------------------------------------- the reference's system, without Lightning ---------------------------------
class System(torch.nn.Module):
    """Attribute names = the state_dict prefixes of NeRFSystem (train_mask_grid_sample.py:77-115)."""

    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        self.enc_cont = encoder_sameoutputsize(out_channel=hp.nerf_out_dim)
        self.enc_a = encoder_sameoutputsize(out_channel=hp.nerf_out_dim)
        self.nerf_coarse = NeRF_sigma(typ="coarse", args=hp, in_channels_xyz=93, in_channels_dir=27)
        self.decoder = style_net(args=hp, residual_blocks=hp.decoder_num_res_blocks)
        self.nerf_fine = NeRF_sigma("fine", args=hp, in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, in_channels_a=hp.N_a,
                                    encode_random=hp.encode_random)
        self.implicit_mask = Context_Guided_Network(classes=1, M=2, N=2, input_channel=3)
        self.models = {"coarse": self.nerf_coarse, "decoder": self.decoder, "fine": self.nerf_fine}
        self.embeddings = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
        self.embedding_a_list = [None] * hp.N_vocab
        self.loss = CRNeRFLoss(hp, coef=1)

    def decode(self, res, typ, H, W, style):
        feat = res["feature_fine" if typ == "content" else "feature_" + typ]
        grid = feat.t().reshape(1, feat.shape[-1], H, W)
        if typ == "content":
            res["rgb_content_img"] = self.decoder(grid, None, type="content")
            return
        img = self.decoder(grid, style)
        if typ == "fine":
            res["rgb_fine_img"] = img
        res["rgb_" + typ] = img if typ == "fine_random" else img.reshape(3, H * W).t()

    def forward(self, rays, ts, whole_img, rng_choice):
        hp, H, W = self.hp, SIDE, SIDE
        whole_img = (whole_img + 1) / 2
        a_img = self.enc_a(whole_img)
        seen = [k for k, v in enumerate(self.embedding_a_list) if v is not None]
        a_rand = a_img if not seen else self.embedding_a_list[seen[rng_choice(len(seen))]]
        mask = self.implicit_mask(whole_img)
        mask = torch.nn.functional.interpolate(mask, size=(H, W), mode="bilinear", align_corners=False).reshape(1, H * W).t()
        kw = dict(args=hp, a_embedded_from_img=a_img, a_embedded_random=a_rand, mask_embedded_from_img=mask, H=H, W=W)
        res = dict(render_rays_cross_ray(self.models, self.embeddings, rays, ts, hp.N_samples, hp.use_disp, hp.perturb, hp.noise_std,
                                         hp.N_importance, hp.chunk, False, **kw))
        self.decode(res, "coarse", H, W, a_img)
        self.decode(res, "fine", H, W, a_img)
        self.decode(res, "content", H, W, None)
        res.update(out_mask=mask, a_embedded=a_img, whole_img=whole_img, a_embedded_random=a_rand)
        self.decode(res, "fine_random", H, W, a_rand)
        res["a_embedded_random_rec"] = self.enc_a(res["rgb_fine_random"])
        res["rgb_fine_random"] = res["rgb_fine_random"].reshape(3, H * W).t()
        self.embedding_a_list[int(ts[0])] = a_img.clone().detach()
        res["content_with_a_embed"] = self.enc_cont(res["rgb_fine_img"])
        res["content_wo_a_embed"] = self.enc_cont(res["rgb_content_img"])
        return res


def save_lightning_ckpt(path, system, optim, sched, hp, step):
    sd = {k: v for k, v in system.state_dict().items()}
    torch.save({"epoch": step // N_IMAGES, "global_step": step, "pytorch-lightning_version": "1.1.5", "state_dict": sd,
                "optimizer_states": [optim.state_dict()], "lr_schedulers": [sched.state_dict()],
                "callbacks": {"ModelCheckpoint": {"best_model_score": torch.tensor(0.0), "best_model_path": path}},
                "hparams_name": "hparams_", "hyper_parameters": {"hparams_": hp}}, path)


def reference_eval(hp, ckpt, test, style_rgbs, dtype=torch.float32):
    """eval.py's recipe on fresh reference modules: load_ckpt per prefix (eval.py:190-216), batched render at perturb = 0 / noise = 0
    (eval.py:29-59), style from enc_a(whole_img) (eval.py:262-270), decode (eval.py:288-295)."""
    from utils import load_ckpt  # reference
    coarse = NeRF_sigma(typ="coarse", args=hp, in_channels_xyz=93, in_channels_dir=27)
    fine = NeRF_sigma("fine", args=hp, in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, in_channels_a=hp.N_a, encode_random=True)
    dec, enc_a = style_net(args=hp, residual_blocks=1), encoder_sameoutputsize(out_channel=64)
    for m, name in ((coarse, "nerf_coarse"), (fine, "nerf_fine"), (dec, "decoder"), (enc_a, "enc_a")):
        with torch.serialization.safe_globals([argparse.Namespace]):   # torch 1.13 (requirements.txt:160) had no weights_only default
            load_ckpt(m, ckpt, model_name=name)
        m.to(dtype).eval()
    models = {"coarse": coarse, "fine": fine, "decoder": dec}
    emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
    out = {}
    with torch.no_grad():
        a = enc_a(((whole_image(style_rgbs) + 1) / 2).to(dtype))
        out["a_embedded"] = a
        for tag, nc, ni in (("64_128", 64, 128), ("256_256", 256, 256)):
            r = render_rays_cross_ray(models, emb, test["rays"].to(dtype), test["ts"], nc, False, 0, 0, ni, 2048, False, test_time=True,
                                      args=hp, a_embedded_from_img=a)
            grid = r["feature_fine"].t().reshape(1, 64, SIDE, SIDE)
            r["rgb_fine"] = dec(grid, a).reshape(3, SIDE * SIDE).t()
            r["rgb_coarse"] = dec(r["feature_coarse"].t().reshape(1, 64, SIDE, SIDE), a).reshape(3, SIDE * SIDE).t()
            r["rgb_content"] = dec(grid, None, type="content").reshape(3, SIDE * SIDE).t()
            r.pop("feature_fine_random", None)
            out.update({tag + "__" + k: v for k, v in r.items()})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--ckpt", default="/tmp/crnerf_trained/last.ckpt")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--out", default="g15_trained.npz", help="fixture file name under tests/golden/ (g16_trained.npz: the long run, --steps 8000)")
    ap.add_argument("--eval-modules-only", action="store_true",
                    help="keep only the state_dict prefixes eval.py loads (nerf_coarse / nerf_fine / decoder / enc_a): a smaller fixture")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    torch.manual_seed(7)
    rng = np.random.default_rng(7)
    os.makedirs(os.path.dirname(a.ckpt), exist_ok=True)
    hp = hparams()
    data = make_dataset(rng)
    train, test = data[:N_IMAGES], data[N_IMAGES]
    log = []
    if not a.eval_only:
        system = System(hp)
        params = [p for p in system.parameters()]
        optim = torch.optim.Adam(params, lr=hp.lr, eps=1e-8, weight_decay=hp.weight_decay)          # utils/__init__.py:31-33
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=hp.num_epochs, eta_min=1e-8)  # :49 (per epoch)
        steps_per_epoch = max(a.steps // hp.num_epochs, 1)
        t0 = time.time()
        for step in range(a.steps):
            b = train[int(rng.integers(N_IMAGES))]
            res = system(b["rays"], b["ts"], whole_image(b["rgbs"]), lambda n: int(rng.integers(n)))
            loss_d, _ = system.loss(res, b["rgbs"], hp, step)
            loss = sum(v for v in loss_d.values())
            optim.zero_grad()
            loss.backward()
            optim.step()
            if (step + 1) % steps_per_epoch == 0:
                sched.step()
            psnr = float(-10 * torch.log10(((res["rgb_fine"].detach() - b["rgbs"]) ** 2).mean()))
            log.append((step, float(loss.detach()), psnr))
            if step % 25 == 0 or step == a.steps - 1:
                print("step %5d  loss %.5f  psnr_fine %.2f  (%.1f s)" % (step, float(loss), psnr, time.time() - t0), flush=True)
            if (step + 1) % 250 == 0 or step == a.steps - 1:
                save_lightning_ckpt(a.ckpt, system, optim, sched, hp, step + 1)
        del system
    # ---- the reference reads its own checkpoint back and renders the held-out view with image 0's appearance
    style = train[0]["rgbs"]
    ref32 = reference_eval(hp, a.ckpt, test, style)
    ref64 = reference_eval(hp, a.ckpt, test, style, torch.float64)
    cond = {}
    for k in ref32:
        d = (ref64[k].double() - ref32[k].double()).abs()
        cond[k] = (float(d.max()), float(d.norm() / ref64[k].double().norm().clamp_min(1e-30)))
        print("reference fp64 - fp32  %-28s max %.3e  rel-L2 %.3e" % (k, *cond[k]))
    psnr = float(-10 * torch.log10(((ref32["64_128__rgb_fine"] - test["rgbs"]) ** 2).mean()))
    print("held-out PSNR of the reference's render vs the scene (own appearance not applied): %.2f dB" % psnr)
    ck = torch.load(a.ckpt, map_location="cpu", weights_only=False)
    keep = ("nerf_coarse.", "nerf_fine.", "decoder.", "enc_a.") if a.eval_modules_only else ("",)
    arrays = {"sd__" + k: v.numpy() for k, v in ck["state_dict"].items() if k.startswith(keep)}
    arrays.update({"ref__" + k: v.numpy() for k, v in ref32.items()})
    arrays.update({"cond__" + k: np.array(v) for k, v in cond.items()})
    arrays.update(rays=test["rays"].numpy(), ts=test["ts"].numpy(), gt=test["rgbs"].numpy(), style_rgbs=style.numpy(),
                  global_step=ck["global_step"], train_log=np.array(log, dtype=np.float32), side=SIDE)
    path = os.path.join(OUT, a.out)
    np.savez_compressed(path, **arrays)
    print("wrote %s  %.1f MiB" % (path, os.path.getsize(path) / 2 ** 20))


if __name__ == "__main__":
    main()
