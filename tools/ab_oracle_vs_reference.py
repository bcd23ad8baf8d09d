"""A/B timing of the CPU oracle against the imported reference -- BUILD CONTAINER ONLY (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tools/ab_oracle_vs_reference.py [--rays 1024] [--threads 8] [--reps 5]

BASELINE.md section 3 / SURVEY 8d: bench.py's `cpu_baseline` times oracle/cpu_ref.py (the reference's Python cannot
travel to the GPU box), so the oracle has to cost what the reference costs.  This script runs both on the same rays
and weights, interleaved (ref, oracle, ref, oracle, ...), for BASELINE configs[0] (64 coarse only) and configs[1]
(64+128), checks the outputs are bit-identical and prints the median times and their ratio (bar: within +-5 %).
The result of the last run is kept in profiles/r2/ab_oracle_vs_reference.txt.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from models.nerf import NeRF_sigma, PosEmbedding            # noqa: E402  (reference)
from models.rendering import render_rays_cross_ray           # noqa: E402  (reference)

import crnerf_amd.synth as synth                             # noqa: E402
from oracle import cpu_ref as O                              # noqa: E402


class Args:
    nerf_out_dim = 64
    pertubeCord = False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chunk", type=int, default=32768)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    st_c, st_f = synth.mlp_state(1, 3.0, 1.0), synth.mlp_state(2, 3.0, 1.0)
    rays = torch.from_numpy(synth.rays(a.rays, seed=0))
    ts = torch.zeros(a.rays, dtype=torch.long)

    def ref_model(typ, st):
        m = NeRF_sigma(typ, Args(), in_channels_xyz=93, in_channels_dir=27, encode_appearance=True, in_channels_a=48, encode_random=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
        return m.eval()

    models = {"coarse": ref_model("coarse", st_c), "fine": ref_model("fine", st_f)}
    emb = {"xyz": PosEmbedding(14, 15), "dir": PosEmbedding(3, 4)}
    wc, wf = O.to_torch(st_c), O.to_torch(st_f)
    lines = ["threads %d, %d rays, chunk %d, torch %s, %d logical CPUs" % (a.threads, a.rays, a.chunk, torch.__version__, os.cpu_count())]
    with torch.no_grad():
        for name, ni in (("configs[0] 64 coarse only", 0), ("configs[1] 64+128", 128)):
            def run_ref():
                return render_rays_cross_ray(models, emb, rays, ts, 64, False, 0, 0, ni, a.chunk, False, test_time=True, args=Args())

            def run_orc():
                return O.render_rays(wc, wf, rays, 64, ni, chunk=a.chunk)

            r, o = run_ref(), run_orc()       # warm-up + identity check
            for k in r:
                if k != "feature_fine_random":
                    assert torch.equal(r[k], o[k]), k
            tr, to = [], []
            for _ in range(a.reps):
                t0 = time.perf_counter(); run_ref(); tr.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); run_orc(); to.append(time.perf_counter() - t0)
            tr.sort(); to.sort()
            mr, mo = tr[len(tr) // 2], to[len(to) // 2]
            lines.append("%-28s reference %.3f s (%.0f rays/s)   oracle %.3f s (%.0f rays/s)   oracle/reference = %.3f   outputs bit-identical"
                         % (name, mr, a.rays / mr, mo, a.rays / mo, mo / mr))
    print("\n".join(lines))
    out = os.path.join(ROOT, "profiles", "r2")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "ab_oracle_vs_reference.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
