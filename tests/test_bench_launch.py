"""bench.py's own process layout: `python bench.py --gpus N` must spawn its N ranks itself.

CPU container: `--dry-run` takes the launcher path all the way to rank start-up -- self-launch under
torch.distributed.run on 127.0.0.1, rendezvous, one SUM all-reduce (gloo here, RCCL on a GPU node), one JSON line from
rank 0 -- and the real command must fail loudly (not hang, not fall back) when there is no GPU.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("workload", ["render", "train"])
def test_gpus2_self_launch_reaches_rank_startup_and_prints_one_json_line(workload):
    p = _run(["--gpus", "2", "--dry-run", "--workload", workload])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["self_launched"] is True
    assert d["ranks_seen"] == [0.0, 1.0]
    assert d["workload"] == workload


@pytest.mark.parametrize("workload", ["render", "train"])
def test_gpus8_dry_run_keeps_the_books_of_configs3(workload):
    """BASELINE configs[3] (batch 256 over 8 GPUs) cannot run here; its BOOK-KEEPING can: eight self-launched ranks over gloo,
    every rank joins the collective, per-rank seeds rank * 10^6 + index, 32 faces per rank in the training workload (8 in the render
    workload), and the nominal ray-step accounting of one step x 8."""
    p = _run(["--gpus", "8", "--dry-run", "--workload", workload], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and d["self_launched"] is True
    assert d["ranks_seen"] == [float(r) for r in range(8)]
    faces = 32 if workload == "train" else 8
    assert d["faces_per_rank"] == faces and d["global_batch"] == 8 * faces
    assert d["seed0_per_rank"] == [r * 1_000_000 for r in range(8)]
    assert d["nominal_ray_steps_per_step"] == 8 * faces * 256 * 256 * 160
    assert d["parallelism"] == "dp8"
    # the end-to-end relight rate of the whole job is part of the N-rank render line (every rank relights its own photographs)
    assert d["relight_e2e_ranks_runs"] is (workload == "render") and d["relight_e2e_leg_runs"] is False


def test_gpus8_dry_run_at_configs4_shape_keeps_every_light_of_a_face_on_one_rank():
    """BASELINE configs[4] (512 x 512 faces, 18 light directions per image, 320 march steps, 8 GPUs): one face per rank, all 18
    lights of that face on its rank (no light is split across ranks, no data-path collective) -- the launch line the driver would use
    on an 8-GPU node is `python3 bench.py --gpus 8 --size 512 --lights 18 --samples 320 --faces 1`."""
    p = _run(["--gpus", "8", "--dry-run", "--size", "512", "--lights", "18", "--samples", "320", "--faces", "1"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and d["ranks_seen"] == [float(r) for r in range(8)]
    assert d["faces_per_rank"] == 1 and d["global_batch"] == 8 and d["lights_per_face"] == 18
    assert d["face_lights_per_rank"] == 18 and d["lights_split_across_ranks"] is False
    assert d["nominal_ray_steps_per_step"] == 8 * 18 * 512 * 512 * 320
    assert d["relight_e2e_leg_runs"] is False                      # the end-to-end leg belongs to the 1-GPU headline line
    assert d["relight_e2e_ranks_runs"] is False                    # ... and its N-rank form to the headline shape (256 x 256, one light)


def test_single_rank_needs_no_launcher():
    p = _run(["--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["self_launched"] is False
    assert d["relight_e2e_leg_runs"] is True and d["lights_per_face"] == 1


def test_without_a_gpu_the_real_command_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert p.returncode != 0
    assert "needs a GPU" in (p.stderr + p.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra", [("render", ["--steps", "8", "--warmup", "2", "--regions", "2"]),
                                            ("train", ["--steps", "2", "--faces", "4"])])
def test_two_self_launched_ranks_run_the_real_workload_on_this_box(workload, extra):
    """`python bench.py --gpus 2` end to end on whatever this box has: with two GPUs the ranks talk RCCL, with one they share it
    over gloo (--oversubscribe: a rehearsal of the launch / fence / max-over-ranks path, flagged as such).  Both ranks ran the
    kernels, took part in the collectives, and rank 0 printed ONE line."""
    import torch
    n_dev = torch.cuda.device_count()
    args = ["--gpus", "2", "--workload", workload, "--no-cpu-baseline", "--no-worst-case"] + extra
    if n_dev < 2:
        args.append("--oversubscribe")
    p = _run(args, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["process_layout"]["ranks"] == 2 and d["process_layout"]["self_launched"] is True
    assert len(d["per_rank"]) == 2 and all(r["seconds"] > 0 for r in d["per_rank"])
    assert d["rccl_ranks"] == (2 if n_dev >= 2 else None)
    assert d["value"] > 0 and d["config"]["parallelism"] == "dp2"
