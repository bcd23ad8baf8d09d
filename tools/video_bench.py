"""Measurement tool (GPU box): BASELINE configs[4] -- the appearance-hallucination fly-through (240 frames of 320x240,
style-image conditioned decoder) on ONE MI355X; frames shard across GPUs without any exchange, so N GPUs render N x
as many frames per second.  usage: python tools/video_bench.py [frames=240]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crnerf_amd.synth as synth
from crnerf_amd import pipeline, video

N = int(sys.argv[1]) if len(sys.argv) > 1 else 240
dev = "cuda:0"
for prec, nc, ni in (("bf16", 64, 128), ("bf16", 256, 256), ("f32", 64, 128)):
    class HP:
        nerf_out_dim, pertubeCord, N_emb_xyz, N_emb_dir, use_disp, encode_a, encode_random, N_a = 64, False, 15, 4, False, True, True, 48
        img_wh, N_samples, N_importance = [320, 240], nc, ni
    hp = HP()
    m, emb = pipeline.get_model(hp, dev), pipeline.get_embeddings(hp)
    m["coarse"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(1, 3.0, 1.0).items()})
    m["fine"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.mlp_state(2, 3.0, 1.0).items()})
    m["decoder"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state(3).items()})
    enc = pipeline.encoder_sameoutputsize(64).to(dev)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.encoder_state(4, 2.0).items()})
    style = torch.rand(1, 3, 60, 80, device=dev)
    n = N if prec == "bf16" else max(N // 8, 4)
    video.render_video(m, emb, enc, style, hp, "brandenburg_gate", n_frames=2, chunk=76800, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr = video.render_video(m, emb, enc, style, hp, "brandenburg_gate", n_frames=n, chunk=76800, precision=prec)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("video %s %d+%d: %d frames of 320x240 in %.2f s = %.1f frames/s (%.0f k rays/s) incl. uint8 frames copied to the host"
          % (prec, nc, ni, len(fr), dt, len(fr) / dt, len(fr) * 76800 / dt / 1e3), flush=True)
