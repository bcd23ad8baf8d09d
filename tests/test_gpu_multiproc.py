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
    # the self-proving record of what the collective layer saw (round-2 verdict, next #5)
    c = j["collectives"]
    assert c["backend"] == "gloo" and c["test_hook_backend"] == "gloo" and c["world_size"] == n and len(c["ranks"]) == n
    assert [r["rank"] for r in c["ranks"]] == list(range(n)) and all(r["device_name"] and r["pci_bus_id"] for r in c["ranks"])
    assert c["distinct_devices"] == 1                                  # here: n ranks time-slice ONE GPU; a real run must show n
    times = c["per_call_ms"]
    for name in ("allreduce_channel_sums_64f", "allreduce_gram_1024f", "allgather_rgb_12B_per_pixel"):
        assert times[name]["calls"] == 6 and times[name]["ms_per_call"] > 0, times      # 2 warm-up + 4 timed steps
    assert j["roofline"]["traffic_source"] is None or "not measured in this run" in j["roofline"]["traffic_source"]
    # round 5: the line shows where a step's time went -- per timed step the device span, the render kernel, the idle gap to the next step and
    # the host's enqueue time (a slow driver-side line can be told from a slow kernel)
    t = j["timing"]
    assert len(t["step_ms_each"]) == 4 and len(t["kernel_ms_each"]) == 4 and len(t["host_enqueue_ms_each"]) == 4 and len(t["device_gap_ms_each"]) == 3
    assert all(k <= s_ for k, s_ in zip(t["kernel_ms_each"], t["step_ms_each"])) and t["device_span_ms"] <= t["wall_ms"] + 1e-3
    assert abs(sum(t["kernel_ms_each"]) / 4 - j["roofline"]["kernel_ms"]) < 1e-3 and j["meets"].startswith("north_star")


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_one_rank_over_rccl_leaves_exactly_the_json_line_on_stdout():
    """The driver's launcher with ONE rank and the production backend (nccl = RCCL; two ranks cannot share this box's one GPU): the process group,
    the collectives of the step and their timers run over RCCL, and stdout holds exactly one line -- the JSON -- although RCCL prints its version
    banner to fd 1 at communicator creation (bench.py sends everything but the line to stderr)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CRNERF_BENCH_TEST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:2000]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["collectives"]["backend"].startswith("nccl") and j["collectives"]["world_size"] == 1
    assert j["collectives"]["per_call_ms"], j["collectives"]


def test_bench_strong_scaling_one_frame_split_over_two_ranks():
    """--scaling strong --workload configs2: ONE frame's rays split over the ranks; the gathered image is the single-process image."""
    args = ["bench.py", "--scaling", "strong", "--workload", "configs2", "--frame", "120x160", "--steps", "2", "--warmup", "1"]
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    one = _json_line(r.stdout)
    two = _json_line(_launch(args + ["--gpus", "2"]))
    for j, n in ((one, 1), (two, 2)):
        assert j["scaling"] == "strong" and j["n_gpus"] == n and j["dtype"] == "bf16" and j["config"]["rays_total"] == 120 * 160
        assert abs(j["value"] - 120 * 160 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]          # whole-job rays/s of the ONE frame
    assert two["config"]["rays_this_rank"] == 120 * 160 // 2 and "collectives" in two and "collectives" not in one
    assert abs(two["image_checksum"] - one["image_checksum"]) <= 1e-5 * abs(one["image_checksum"])        # same frame (reduction order differs)


@pytest.mark.parametrize("split,n", [("frames", 2), ("rays", 2), ("frames", 8)])
def test_bench_strong_scaling_video_fly_through_split_over_ranks(split, n):
    """--scaling strong --workload configs4 (BASELINE configs[4], appearance_modification_video.py:121-189,224-262): ONE fly-through split over the
    ranks by whole frames (no data-path collective) or by every frame's rays (decoder exchange per frame).  The frames' checksum is the
    single-process one."""
    args = ["bench.py", "--scaling", "strong", "--workload", "configs4", "--frames", "8", "--samples", "32+32", "--steps", "1", "--warmup", "1"]
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    one = _json_line(r.stdout)
    many = _json_line(_launch(args + ["--gpus", str(n), "--video-split", split], nproc=n))
    for j, k in ((one, 1), (many, n)):
        assert j["scaling"] == "strong" and j["n_gpus"] == k and j["dtype"] == "f32" and j["config"]["rays_total"] == 8 * 320 * 240
        assert abs(j["value"] - 8 * 320 * 240 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"] and j["frames_per_s"] > 0
    if split == "frames":
        assert many["config"]["frames_this_rank"] == 8 // n and many["config"]["reductions"] == "none"
        assert "allreduce_gram_1024f" not in many["collectives"]["per_call_ms"], many["collectives"]["per_call_ms"]   # no data-path collective
        # rank 0's frames only: every rank renders its own, the checksum of the whole fly-through is the sum over ranks
    else:
        assert many["config"]["rays_per_frame_this_rank"] == 320 * 240 // n
        assert many["collectives"]["per_call_ms"]["allreduce_gram_1024f"]["calls"] == 2 * 8, many["collectives"]["per_call_ms"]
        assert abs(many["image_checksum"] - one["image_checksum"]) <= 2e-5 * abs(one["image_checksum"])   # uint8 frames: a reduction-order flip of a pixel is +-1
    assert many["frames_checksum_all_ranks"] is not None and abs(many["frames_checksum_all_ranks"] - one["image_checksum"]) <= 2e-5 * abs(one["image_checksum"])


@pytest.mark.parametrize("precision", ["auto", "f32"])
def test_bench_strong_scaling_one_training_batch_split_over_two_ranks(precision):
    """--scaling strong --workload configs3: ONE grid-sample training batch split over the ranks (ray-parallel TrainingSystem), in the training
    default (auto: split-operand products on the fp16 / bf16 matrix cores) and on the fp32 matrix cores."""
    j = _json_line(_launch(["bench.py", "--gpus", "2", "--scaling", "strong", "--workload", "configs3", "--train-rays", "4096", "--steps", "2", "--warmup", "1",
                            "--train-precision", precision]))
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["rays_total"] == 4096 and j["config"]["train_precision"] == precision
    assert j["dtype"] == "f32" if precision == "f32" else j["dtype"].startswith("f32h2"), j["dtype"]
    assert j["roofline"]["peak"] == (157.3 if precision == "f32" else 2500.0) and 0 < j["roofline"]["frac"] < 1
    assert j["replicas_identical"] is True and j["loss"] > 0
    times = j["collectives"]["per_call_ms"]
    assert any(k.startswith("allreduce_gradients_flat") for k in times), times
    # round 6: the decodes stay sharded (three with statistics + the content decode per step: forward all-reduces and the RGB all-gather, backward
    # the affine-gradient and column-sum all-reduces), the encoder passes run as row bands; the feature rows are no longer all-gathered.
    # 4 steps in all (1 warm-up + 2 timed + the instrumented one)
    assert "allgather_feature_rows" not in times, times
    assert times["allreduce_gram_1024f"]["calls"] >= 3 * 3 and times["allgather_rgb_12B_per_pixel"]["calls"] >= 4 * 3, times
    assert times["allreduce_affine_gradient_320f"]["calls"] >= 3 * 3 and times["allreduce_column_sums_64f"]["calls"] >= 3 * 3, times
    assert times["allgather_style_grid_rows"]["calls"] >= 3 * 3 and times["allreduce_image_gradient_12B_per_pixel"]["calls"] >= 3 * 3, times
    sec = j["sections"]
    assert sec["sharded_ms"] > 0 and sec["not_sharded_ms"] > 0 and abs(sec["sharded_ms"] + sec["not_sharded_ms"] - sum(sec[k] for k in
               ("render_forward_ms", "rest_forward_ms", "rest_backward_ms", "render_backward_ms", "sync_and_adam_ms"))) < 1e-6


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


def test_ddp_wrapped_training_system_matches_allreduce_gradients():
    """"or wrap the modules in torch.nn.parallel.DistributedDataParallel" -- the reference's strategy
    (train_mask_grid_sample.py:445-446).  Two ranks, each with its own batch: DDP's bucketed gradient averaging through the HIP autograd twins
    leaves the gradients parallel.allreduce_gradients leaves (tests/_ddp_worker.py)."""
    out = _launch(["tests/_ddp_worker.py"], nproc=2, timeout=600)
    assert out.count("ddp gradients match allreduce_gradients: True") == 2, out


def test_peer_exchange_sums_are_exact_and_identical_on_three_ranks():
    out = _launch(["tests/_peer_worker.py", "sums"], nproc=3, timeout=300)
    assert out.count("peer sums exact: True") == 3, out


def test_peer_exchange_missing_rank_gives_nan_and_a_name_not_a_hang():
    out = _launch(["tests/_peer_worker.py", "timeout"], nproc=2, timeout=120)
    assert "rank 0 timeout detected: nan=True msg=True" in out, out
