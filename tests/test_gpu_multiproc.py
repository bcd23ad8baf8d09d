"""Multi-GPU launch readiness on a ONE-GPU box: the exact command lines the driver uses for the N > 1 scaling bench
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`)
and the ray-parallel configs[3] training bench, with two ranks sharing cuda:0 over gloo through the CRNERF_BENCH_TEST_BACKEND
hook (RCCL refuses two ranks on one device).  Everything but the collective transport is the production path: rendezvous from
the environment, ray sharding, the sharded decode's two all-reduces + RGB all-gather, the barrier / max-over-ranks timing, the
JSON line.  No multi-GPU scaling NUMBER comes out of this -- that needs the driver's 8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, nproc=2, extra_env=None, timeout=900):
    env = dict(os.environ, CRNERF_BENCH_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("n", [2, 8])
def test_bench_n_ranks_prints_one_valid_json_line(n):
    """The driver's N = 2, 4, 8 command line (8 ranks time-slice the one GPU here)."""
    out = _launch(["bench.py", "--gpus", str(n), "--steps", "4", "--warmup", "2"], nproc=n)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out                                         # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["steps"] == 4 and j["warmup"] == 2 and j["scaling"] == "weak" and j["unit"] == "rays/s"
    assert j["value"] > 0 and abs(j["value"] - n * 1024 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]   # whole-job aggregate
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] < 1
    assert "cpu_baseline" not in j                                      # N == 1 only
    sums = j["test_rgb_checksum_per_rank"]
    assert len(sums) == n and all(s == sums[0] for s in sums) and sums[0][1] == n * 1024   # every rank holds the same gathered n x 1024-pixel image


def test_train_config3_two_ranks_ray_parallel_replicas_agree():
    out = _launch(["tools/train_config4_bench.py", "4096", "32", "32"], extra_env={"CRNERF_TRAIN_BENCH_STEPS": "1,2"})
    assert "identical on every rank: True" in out, out
    assert "config-4 training step, 4096 rays" in out, out


def test_bench_two_ranks_peer_exchange_matches_the_collective_path():
    """--peer-exchange: the decoder's two reductions through HIP-IPC windows; with two ranks a + b is the same sum in either
    carrier, so the gathered image must be the same one, on both ranks."""
    base = json.loads([ln for ln in _launch(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"]).splitlines() if ln.startswith("{")][0])
    for env in ({}, {"CRNERF_PEER_WINDOW_COARSE": "1"}):
        out = _launch(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--peer-exchange"], extra_env=env)
        j = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
        assert j["config"]["reductions"] == "peer windows (HIP IPC)"
        sums = j["test_rgb_checksum_per_rank"]
        assert sums[0] == sums[1] == base["test_rgb_checksum_per_rank"][0], (sums, base["test_rgb_checksum_per_rank"])


def test_peer_exchange_sums_are_exact_and_identical_on_three_ranks():
    out = _launch(["tests/_peer_worker.py", "sums"], nproc=3, timeout=300)
    assert out.count("peer sums exact: True") == 3, out


def test_peer_exchange_missing_rank_gives_nan_and_a_name_not_a_hang():
    out = _launch(["tests/_peer_worker.py", "timeout"], nproc=2, timeout=120)
    assert "rank 0 timeout detected: nan=True msg=True" in out, out
