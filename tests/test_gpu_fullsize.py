"""BASELINE configs[2], [3], [4] at their FULL sizes on one MI355X (round-1 verdict: the full sizes lived only in builder-run
tools), plus the validation branch of NeRFSystem (val_mode / validation_step).  Parity at these sizes goes through
size-independent properties, chunk invariance, oracle checks on ray subsamples at identical depths, and a CPU-oracle decode of
the full feature grid (cheap).  Timings and peak memory are printed (pytest -s) and written to gpurun_out/fullsize_r2.json.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

import crnerf_amd.synth as synth
from crnerf_amd import ops, pipeline
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy


def record(name, values):
    print(name, values, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(d):
        return
    path = os.path.join(d, "fullsize_r2.json")
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[name] = values
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


class HPBase:
    maskrs_max, maskrs_min, maskrs_k, maskrd = 5e-2, 6e-3, 1e-3, 1e-3
    weightKL, weightRecA, weightcontent, mse_on_appearance = 1e-5, 1e-3, 1e-4, False
    nerf_out_dim, pertubeCord, N_emb_xyz, N_emb_dir, use_disp, encode_a, encode_random, N_a = 64, False, 15, 4, False, True, True, 48


def _load(models, enc, gain=3.0, bias=1.0, **kw):
    models["coarse"].load_state_dict({k: T(v) for k, v in synth.mlp_state(1, gain, bias, **kw).items()})
    models["fine"].load_state_dict({k: T(v) for k, v in synth.mlp_state(2, gain, bias, **kw).items()})
    models["decoder"].load_state_dict({k: T(v) for k, v in synth.decoder_state(3).items()})
    if enc is not None:
        enc.load_state_dict({k: T(v) for k, v in synth.encoder_state(4, 2.0).items()})


def _camera(W, H):
    focal = W / 2 / np.tan(np.pi / 6)
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
    c2w = np.array([[1, 0, 0, 0.05], [0, -1, 0, 0.02], [0, 0, -1, 0.1]], dtype=np.float32)
    return K, c2w


# ------------------------------------------------------------------ configs[2]: 800x800 image, 32,768-ray chunks, decoder on, bf16
@torch.no_grad()
def test_config2_full_image_bf16_800x800():
    class HP(HPBase):
        img_wh, N_samples, N_importance = [800, 800], 64, 128
    hp = HP()
    Wd = Ht = 800
    R = Wd * Ht
    models, emb = pipeline.get_model(hp, DEV), pipeline.get_embeddings(hp)
    enc = pipeline.encoder_sameoutputsize(64).to(DEV)
    _load(models, enc)
    K, c2w = _camera(Wd, Ht)
    from crnerf_amd.datasets.ray_utils import generate_rays
    rays = generate_rays(Ht, Wd, K, c2w, 0.0, 5.0, device=torch.device(DEV))
    assert rays.shape == (R, 8)
    photo = torch.rand(1, 3, 100, 100, generator=torch.Generator().manual_seed(0)).to(DEV)
    a_emb = enc(photo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pipeline.batched_inference(models, emb, rays, None, 64, 128, False, 32768, False, args=hp, a_embedded_from_img=a_emb, precision="bf16")
    rgb = pipeline.decode_image(models, res, Ht, Wd, a_emb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # properties at full size
    for k, v in res.items():
        assert torch.isfinite(v).all(), k
    assert res["weights_fine"].shape == (R, 192) and res["feature_fine"].shape == (R, 64) and rgb.shape == (R, 3)
    s = res["weights_fine"].sum(-1)
    assert float(s.max()) <= 1 + 1e-5 and float(s.min()) > 0.99 and float(res["weights_fine"].min()) >= 0
    assert bool((res["depth_fine"] >= -1e-4).all()) and bool((res["depth_fine"] <= 5 + 1e-3).all())
    assert 0 <= float(rgb.min()) and float(rgb.max()) <= 1
    # chunk invariance: 20 launches of 32,768 rays == one launch of 640,000 == ragged 50,000-ray chunks (rays are independent, G7)
    one = pipeline.batched_inference(models, emb, rays, None, 64, 128, False, R, False, args=hp, a_embedded_from_img=a_emb, precision="bf16")
    rag = pipeline.batched_inference(models, emb, rays, None, 64, 128, False, 50000, False, args=hp, a_embedded_from_img=a_emb, precision="bf16")
    for k in res:
        for name, other in (("one launch", one), ("50,000-ray chunks", rag)):
            if not torch.equal(res[k], other[k]):
                d = (res[k] != other[k]).view(R, -1).any(1).nonzero().flatten()
                raise AssertionError("%s: 32,768-ray chunks vs %s differ in %d rays, first %s last %s, max |d| %.3e" % (
                    k, name, d.numel(), d[:16].tolist(), d[-4:].tolist(), float((res[k] - other[k]).abs().max())))
    # a 4,096-ray slice (8 scattered blocks of 512 rays) against the bf16 oracle at IDENTICAL depths
    idx = torch.cat([torch.arange(b, b + 512) for b in (0, 99_840, 200_192, 319_744, 320_256, 450_048, 560_128, R - 512)])
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sl = rays[idx.to(DEV)].contiguous()
    pk = lambda m: m.packed_weights("bf16")  # noqa: E731
    zt, ut = torch.linspace(0, 1, 64, device=DEV), torch.linspace(0, 1, 128, device=DEV)
    got = ops.render_rays(pk(models["coarse"]), pk(models["fine"]), sl, 64, 128, z_steps=zt, u=ut, want_z_fine=True, precision="bf16")
    for k in ("feature_fine", "weights_fine", "feature_coarse"):
        assert torch.equal(got[k], res[k][idx.to(DEV)]), k                   # the slice is the same arithmetic as the full image
    wc, wf = O.to_torch(synth.mlp_state(1, 3.0, 1.0)), O.to_torch(synth.mlp_state(2, 3.0, 1.0))
    ref = O.render_rays(wc, wf, sl.cpu(), 64, 128, z_steps=zt.cpu(), u=ut.cpu(), z_fine=got["z_fine"].cpu(), precision="bf16")
    dc = float((got["feature_coarse"].cpu() - ref["feature_coarse"]).abs().max())
    df = float((got["feature_fine"].cpu() - ref["feature_fine"]).abs().max())
    dw = float((got["weights_fine"].cpu() - ref["weights_fine"]).abs().max())
    # summation order + the rare bf16 rounding flip of an activation on the gain-3 nets: the MEAN is the tight bound (measured
    # 3e-5); over 1M points a few flips land on a heavily weighted sample (measured max 9e-3 coarse / 6e-3 fine / 1.5e-3 weights)
    assert dc <= 3e-2 and df <= 3e-2 and dw <= 5e-3, (dc, df, dw)
    mean_f = float((got["feature_fine"].cpu() - ref["feature_fine"]).abs().mean())
    mean_c = float((got["feature_coarse"].cpu() - ref["feature_coarse"]).abs().mean())
    assert mean_f <= 1e-4 and mean_c <= 1e-4, (mean_f, mean_c)
    # the 640k-pixel cross-ray decode against the CPU oracle on the GPU path's own feature grid
    ref_rgb = O.crossray_decode(O.to_torch(synth.decoder_state(3)), O.feature_to_grid(res["feature_fine"].cpu(), Ht, Wd),
                                a_emb.cpu().contiguous()).reshape(3, R).t()
    d_rgb = float((rgb.cpu() - ref_rgb).abs().max())
    assert d_rgb <= 5e-6, d_rgb
    record("configs2_800x800_bf16", {"ms_render_plus_decode": dt * 1e3, "rays_per_s": R / dt, "slice_max_abs_feature_fine_vs_bf16_oracle": df,
                                     "slice_mean_abs_feature_fine": mean_f, "decode_640k_max_abs_rgb_vs_oracle": d_rgb})


# ------------------------------------------------------------------ configs[3]: one 65,536-ray training step, grid-sample masking
@pytest.mark.parametrize("mode", ["auto", "auto+recompute", "f32", "bf16+recompute"])
def test_config3_training_step_65536_rays_use_mask(mode):
    """mode auto: the training DEFAULT, the one every headline training number quotes (forward / data gradient on the h2 core with the f32x3
    safety net, f16x2 weight gradients with bf16x3 behind them: every product fp32-accurate).  auto+recompute: the same arithmetic with
    activation checkpointing per ray chunk (set_training_recompute: the saved rows of one chunk at a time -- the step must fit 40 GiB where the
    default keeps 138).  f32: every product on the fp32 matrix cores (the reference's arithmetic; up to round 4 this leg did not pin the
    forward mode and silently ran the default).  bf16+recompute: the opt-in mixed-precision twins with the fused bf16 renderer as forward
    (DESIGN 3.5): the same checks, and the step must fit 60 GB."""
    from crnerf_amd import autograd as AG
    from crnerf_amd.datasets.phototourism_mask_grid_sample import GridSampleBatcher
    mixed = mode == "bf16+recompute"
    AG.set_training_precision("bf16" if mixed else "f32")
    AG.set_training_recompute(mode.endswith("+recompute"))
    AG.set_training_forward_precision("f32" if mode == "f32" else None)
    AG.set_wgrad_precision("f32" if mode == "f32" else None)
    try:
        if mode.startswith("auto"):
            # (unless the environment overrides the defaults) the weight gradients behind the h2 data gradient are f16x2 (3); a caller without it gets bf16x3 (2)
            assert AG.get_training_forward_mode() == "auto" and AG.get_wgrad_bf16(h2=True) == 3 and AG.get_wgrad_bf16() == 2
        _config3_step(mode, GridSampleBatcher)
    finally:
        AG.set_training_precision("f32")
        AG.set_training_recompute(False)
        AG.set_training_forward_precision(None)
        AG.set_wgrad_precision(None)


def _config3_step(mode, GridSampleBatcher):

    class HP(HPBase):
        img_wh, N_samples, N_importance, perturb, noise_std, chunk, N_vocab = [256, 256], 64, 64, 1.0, 1.0, 8 * 1024, 1500
        use_mask, encode_c = True, True                              # command/train.sh:24
    hp = HP()
    R = 65536
    torch.manual_seed(0)
    sysm = pipeline.TrainingSystem(hp, device=DEV)
    _load(sysm.models, sysm.enc_a, 2.0, 0.5)
    sysm.enc_cont.load_state_dict({k: T(v) for k, v in synth.encoder_state(5, 2.0).items()})
    n_img, iw, ih = 2, 512, 384
    rays = torch.cat([torch.cat([T(synth.rays(iw * ih, seed=i, H=ih, W=iw)), torch.full((iw * ih, 1), float(i))], 1) for i in range(n_img)]).to(DEV)
    rgbs = torch.rand(n_img * iw * ih, 3, device=DEV)
    imgs = [torch.rand(1, 3, ih // 8, iw // 8, device=DEV) * 2 - 1 for _ in range(n_img)]
    batcher = GridSampleBatcher(rays, rgbs, np.array([[iw, ih]] * n_img), batch_size=R, all_imgs=imgs)
    batch = batcher.__getitem__(0, 0)
    assert batch["rays"].shape == (R, 8) and batch["rgb_idx"].shape[0] == R
    opt = torch.optim.Adam(sysm.parameters(), lr=5e-4)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    times = []
    for it in range(2):
        opt.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        loss, loss_d, results = sysm.training_step(batch)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        if it == 0:
            assert list(loss_d.keys()) == ["kl_a", "rec_a_random", "c_l", "content_constraint", "r_ms", "r_md", "f_l"]
            ref, _ = O.crnerf_loss({k: v.detach().cpu() for k, v in results.items() if torch.is_tensor(v)}, batch["rgbs"].cpu(), hp, 0)
            for k in loss_d:
                assert abs(float(loss_d[k]) - float(ref[k])) <= 2e-5 * abs(float(ref[k])) + 1e-10, (k, float(loss_d[k]), float(ref[k]))
            assert results["out_mask"].shape == (R, 1) and results["rgb_fine"].shape == (R, 3) and results["weights_fine"].shape == (R, 128)
            for name, mod in (("coarse", sysm.models["coarse"]), ("fine", sysm.models["fine"]), ("decoder", sysm.models["decoder"]),
                              ("enc_a", sysm.enc_a), ("enc_cont", sysm.enc_cont), ("implicit_mask", sysm.implicit_mask)):
                for pn, p in mod.named_parameters():
                    assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0, (name, pn)
            first = float(loss.detach())
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert np.isfinite(first) and float(loss.detach()) < first * 1.5
    record("configs3_train_step_65536_%s" % mode, {"ms_step": min(times) * 1e3, "rays_per_s": R / min(times), "peak_mem_GiB": peak, "loss": float(loss.detach())})
    # fp32-accurate modes: fit one 288 GB MI355X with margin; with recompute the saved rows of ONE ray chunk are alive at a time
    assert peak < {"bf16+recompute": 60.0, "auto+recompute": 40.0}.get(mode, 200.0), peak


# ------------------------------------------------------------------ configs[4]: appearance-hallucination video frames, 320x240, 256+256
@torch.no_grad()
def test_config4_video_frames_320x240_256p256_style_conditioned():
    from crnerf_amd import video
    from crnerf_amd.datasets.ray_utils import generate_rays

    class HP(HPBase):
        img_wh, N_samples, N_importance = [320, 240], 256, 256        # appearance_modification_video.py:47-50, :31
    hp = HP()
    models, emb = pipeline.get_model(hp, DEV), pipeline.get_embeddings(hp)
    enc = pipeline.encoder_sameoutputsize(64).to(DEV)
    _load(models, enc, 2.45, -1.0, band_limit=4)                     # well-conditioned nets: the frame can be compared END TO END
    dst = synth.decoder_state(3, contrast=4000.0)                    # high-contrast decoder: the frames show the scene, not a flat colour
    models["decoder"].load_state_dict({k: T(v) for k, v in dst.items()})
    style_img = torch.rand(1, 3, 60, 80, generator=torch.Generator().manual_seed(3)).to(DEV)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = video.render_video(models, emb, enc, style_img, hp, scene="brandenburg_gate", n_frames=4, precision="bf16")
    torch.cuda.synchronize()
    dt_bf16 = (time.perf_counter() - t0) / 4
    assert sorted(frames) == [0, 1, 2, 3] and all(f.shape == (240, 320, 3) and f.dtype == np.uint8 for f in frames.values())
    assert not np.array_equal(frames[0], frames[3])                   # the camera moves
    # frame 2 in fp32 against the oracle: features on a ray subsample END TO END (stated fp32 tolerance; smooth nets), pixels of
    # the whole frame through the oracle's decode of the GPU feature grid, and against the uint8 bf16 frame
    K, poses = video.define_camera(hp.img_wh), video.define_poses("brandenburg_gate", 4)
    a_emb = enc(style_img)
    rays = generate_rays(240, 320, K, poses[2].astype(np.float32), 0.0, 5.0, device=torch.device(DEV))
    t0 = time.perf_counter()
    res = pipeline.batched_inference(models, emb, rays, None, 256, 256, False, 4096, False, args=hp, a_embedded_from_img=a_emb, precision="f32")
    rgb = pipeline.decode_image(models, res, 240, 320, a_emb)
    torch.cuda.synchronize()
    dt_f32 = time.perf_counter() - t0
    idx = torch.arange(0, 76800, 301)                                 # 256 rays across the frame
    wc, wf = O.to_torch(synth.mlp_state(1, 2.45, -1.0, band_limit=4)), O.to_torch(synth.mlp_state(2, 2.45, -1.0, band_limit=4))
    zt, ut = torch.linspace(0, 1, 256, device=DEV).cpu(), torch.linspace(0, 1, 256, device=DEV).cpu()
    ref = O.render_rays(wc, wf, rays[idx.to(DEV)].cpu(), 256, 256, z_steps=zt, u=ut)
    f_got, f_ref = res["feature_fine"][idx.to(DEV)].cpu(), ref["feature_fine"]
    rel = float((f_got - f_ref).norm() / f_ref.norm())
    assert rel <= 1e-5 and float((f_got - f_ref).abs().max()) <= 2e-5, (rel, float((f_got - f_ref).abs().max()))
    assert float((res["weights_fine"][idx.to(DEV)].cpu() - ref["weights_fine"]).abs().max()) <= 1e-5
    ref_rgb = O.crossray_decode(O.to_torch(dst), O.feature_to_grid(res["feature_fine"].cpu(), 240, 320), a_emb.cpu().contiguous())
    d_rgb = float((rgb.cpu() - ref_rgb.reshape(3, -1).t()).abs().max())
    assert d_rgb <= 2e-5, d_rgb                                       # SURVEY 8d pixel tolerance, high-contrast decoder
    assert float(rgb.max() - rgb.min()) > 0.3                         # the frame shows structure
    f32_u8 = (rgb.reshape(240, 320, 3).clamp(0, 1) * 255).to(torch.uint8).cpu().numpy().astype(np.int32)
    assert int(np.abs(f32_u8 - frames[2].astype(np.int32)).max()) <= 4   # bf16 frame within 4/255 (1.5e-2: bf16 through the x7 decoder) of fp32
    record("configs4_video_320x240_256p256", {"ms_per_frame_bf16": dt_bf16 * 1e3, "fps_bf16": 1 / dt_bf16, "ms_per_frame_f32": dt_f32 * 1e3,
                                              "subsample_feature_rel_l2_vs_oracle_end_to_end": rel, "decode_max_abs_rgb_vs_oracle": d_rgb})


# ------------------------------------------------------------------ N3: validation branch (train_mask_grid_sample.py:151,174-182,339-402)
@pytest.mark.parametrize("use_mask", [False, True])
def test_validation_step_val_mode(use_mask):
    class HP(HPBase):
        img_wh, N_samples, N_importance, perturb, noise_std, chunk, N_vocab = [40, 24], 32, 32, 1.0, 1.0, 2048, 8
        encode_c = True
    hp = HP()
    hp.use_mask = use_mask
    torch.manual_seed(0)
    sysm = pipeline.TrainingSystem(hp, device=DEV)
    _load(sysm.models, sysm.enc_a, 2.0, 0.5)
    sysm.enc_cont.load_state_dict({k: T(v) for k, v in synth.encoder_state(5, 2.0).items()})
    Wd, Ht = 40, 24
    R = Wd * Ht
    # a validation sample as PhototourismDataset(split='val') hands it to the DataLoader (batch dimension 1, :341-347)
    batch = {"rays": T(synth.rays(R, H=Ht, W=Wd)).to(DEV)[None], "ts": torch.full((1, R), 5, dtype=torch.int64, device=DEV),
             "rgbs": torch.rand(1, R, 3, device=DEV), "whole_img": torch.rand(1, 3, Ht, Wd, device=DEV) * 2 - 1,
             "img_wh": torch.tensor([[Wd, Ht]]), "rgb_idx": None}
    log = sysm.validation_step(batch, 0)
    res = log["results"]
    want = ["kl_a", "rec_a_random", "c_l", "content_constraint"] + (["r_ms", "r_md"] if use_mask else []) + ["f_l"]
    assert [k for k in log if k not in ("val_loss", "val_psnr", "results")] == want
    assert res["rgb_fine"].shape == (R, 3) and res["rgb_fine_random"].shape == (R, 3) and not res["rgb_fine"].requires_grad
    if use_mask:                                                      # the WHOLE interpolated mask, no rgb_idx gather (:174-175)
        assert res["out_mask"].shape == (R, 1)
        assert 0 < float(res["out_mask"].min()) and float(res["out_mask"].max()) < 1
    ref, _ = O.crnerf_loss({k: v.detach().cpu() for k, v in res.items() if torch.is_tensor(v)}, batch["rgbs"][0].cpu(), hp, 0)
    for k in want:
        assert abs(float(log[k]) - float(ref[k])) <= 2e-5 * abs(float(ref[k])) + 1e-10, k
    assert abs(float(log["val_psnr"]) - O.psnr(res["rgb_fine"].cpu(), batch["rgbs"][0].cpu())) < 1e-3
    assert abs(float(log["val_loss"]) - sum(float(log[k]) for k in want)) < 1e-6
    # val_mode renders with perturb / noise as configured (the reference does not switch them off, :186-197) under no_grad -> the
    # fused inference kernel; the same image through forward(val_mode=False)+rgb_idx=all pixels gives the same mask values
    assert sysm.training and sysm.implicit_mask.training if use_mask else sysm.training      # validation_step restored train mode
    if use_mask:
        sysm.eval()
        with torch.no_grad():
            r2 = sysm.forward(batch["rays"][0], batch["ts"][0], batch["whole_img"], Wd, Ht, torch.arange(R, device=DEV), hw_whole=(Ht, Wd))
        assert torch.equal(r2["out_mask"], res["out_mask"])
