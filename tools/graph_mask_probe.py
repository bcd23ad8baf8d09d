"""Experiment (GPU box): the transient-mask network's forward and backward as two hipGraphs (torch.cuda.make_graphed_callables) inside the
train.sh-configuration step at 1,024 rays: what the step costs when those ~112 launches stop costing host time."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crnerf_amd.synth as synth
from crnerf_amd import pipeline
from crnerf_amd.datasets.phototourism_mask_grid_sample import GridSampleBatcher
R, NC, NI = 1024, 64, 64
side, dev = 32, "cuda:0"


class HP:
    maskrs_max, maskrs_min, maskrs_k, maskrd = 5e-2, 6e-3, 1e-3, 0.0
    weightKL, weightRecA, weightcontent, mse_on_appearance = 1e-5, 1e-3, 1e-4, False
    nerf_out_dim, pertubeCord, N_emb_xyz, N_emb_dir, use_disp, encode_a, encode_random, N_a = 64, False, 15, 4, False, True, True, 48
    img_wh, N_samples, N_importance, perturb, noise_std, chunk, N_vocab = [side, side], NC, NI, 1.0, 1.0, 8 * 1024, 1500
    use_mask, encode_c = True, True


def build():
    torch.manual_seed(0)
    sysm = pipeline.TrainingSystem(HP(), device=dev)
    sysm.models["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(1, 2.0, 0.5).items()})
    sysm.models["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(2, 2.0, 0.5).items()})
    sysm.models["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
    sysm.enc_a.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(4, 2.0).items()})
    sysm.enc_cont.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(5, 2.0).items()})
    return sysm


n_img, iw, ih = 8, 512, 384
rays = torch.cat([torch.cat([torch.from_numpy(synth.rays(iw * ih, seed=i, H=ih, W=iw)), torch.full((iw * ih, 1), float(i))], 1) for i in range(n_img)]).to(dev)
rgbs = torch.rand(n_img * iw * ih, 3, device=dev)
imgs = [torch.rand(1, 3, ih // 8, iw // 8, device=dev) * 2 - 1 for _ in range(n_img)]
batcher = GridSampleBatcher(rays, rgbs, np.array([[iw, ih]] * n_img), batch_size=R, all_imgs=imgs)


def run(sysm, tag, n_warm=5, n=40):
    opt = torch.optim.Adam(sysm.parameters(), lr=5e-4, fused=True)

    def step(i):
        batch = batcher.__getitem__(i, 0)
        opt.zero_grad(set_to_none=True)
        loss, _, _ = sysm.training_step(batch)
        loss.backward()
        opt.step()
        return loss
    for i in range(n_warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        l = step(n_warm + i)
    torch.cuda.synchronize()
    print("%-28s %.2f ms per step, loss %.4f" % (tag, (time.perf_counter() - t0) / n * 1e3, float(l)), flush=True)


sysm = build()
run(sysm, "eager")
run(sysm, "eager (again)")
sample = ((imgs[0] + 1) / 2).clone()
sysm.implicit_mask = torch.cuda.make_graphed_callables(sysm.implicit_mask, (sample,))
run(sysm, "mask network graphed")
run(sysm, "mask network graphed (again)")
