"""Measurement tool (GPU box): the non-headline BASELINE configs on one MI355X, fp32 and bf16, through the drop-in
modules (pipeline.render_frame / batched_inference): full-image render 800x800 (config 3 shape) and the video
frame 320x240 at 256+256 samples (config 5 shape), plus coarse-only config 1."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crnerf_amd.synth as synth
from crnerf_amd import pipeline

dev = "cuda:0"


def setup(W, H, Nc, Ni):
    class HP:
        nerf_out_dim, pertubeCord, N_emb_xyz, N_emb_dir, use_disp, encode_a, encode_random, N_a = 64, False, 15, 4, False, True, True, 48
        img_wh, N_samples, N_importance = [W, H], Nc, Ni
    hp = HP()
    m, emb = pipeline.get_model(hp, dev), pipeline.get_embeddings(hp)
    m["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(1, 3.0, 1.0).items()})
    if "fine" in m:
        m["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(2, 3.0, 1.0).items()})
    m["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
    enc = pipeline.encoder_sameoutputsize(64).to(dev)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(4, 2.0).items()})
    return hp, m, emb, enc


def frame_bench(tag, W, H, Nc, Ni, chunk, reps, precision="f32"):
    hp, m, emb, enc = setup(W, H, Nc, Ni)
    focal = W / 2 / np.tan(np.pi / 6)
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]])
    c2w = np.array([[1, 0, 0, 0.05], [0, -1, 0, 0.02], [0, 0, -1, 0.1]], dtype=np.float32)
    style = torch.rand(1, 3, 60, 80, device=dev)
    pipeline.render_frame(m, emb, enc, style, H, W, K, c2w, hp, chunk=chunk, precision=precision)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        img = pipeline.render_frame(m, emb, enc, style, H, W, K, c2w, hp, chunk=chunk, precision=precision)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pts = Nc + (Nc + Ni if Ni else 0)
    print("%-34s %4s %4dx%-4d %3d+%-3d chunk %6d: %8.1f ms/frame  %7.1f k rays/s  %6.1f TFLOP/s (MLP)"
          % (tag, precision, W, H, Nc, Ni, chunk, dt * 1e3, W * H / dt / 1e3, W * H * pts * 1.233152e6 / dt / 1e12), flush=True)


frame_bench("config1 coarse-only (1024 rays)", 32, 32, 64, 0, 1024, 20)
frame_bench("config3 shape full image fp32", 800, 800, 64, 128, 32768, 2)
frame_bench("config3 shape, one launch", 800, 800, 64, 128, 640000, 2)
frame_bench("config5 video frame (script dflt)", 320, 240, 256, 256, 4096, 3)
frame_bench("config5 video frame, one launch", 320, 240, 256, 256, 76800, 3)
frame_bench("config5 video frame 64+128", 320, 240, 64, 128, 76800, 5)
# BASELINE configs[2] / [4] arithmetic: bf16 matrix cores
frame_bench("config3 full image", 800, 800, 64, 128, 32768, 3, "bf16")
frame_bench("config3 full image, one launch", 800, 800, 64, 128, 640000, 3, "bf16")
frame_bench("config3 eval setting 256+256", 800, 800, 256, 256, 640000, 2, "bf16")
frame_bench("config5 video frame (script dflt)", 320, 240, 256, 256, 4096, 5, "bf16")
frame_bench("config5 video frame, one launch", 320, 240, 256, 256, 76800, 5, "bf16")
frame_bench("config5 video frame 64+128", 320, 240, 64, 128, 76800, 10, "bf16")
