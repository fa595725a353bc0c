"""bench.py's own launch logic on the CPU: `--gpus N` without a launcher starts N rank processes (gloo rendezvous on
127.0.0.1), the parent relays rank 0's JSON line and fails when a rank fails.  BENCH_DRY_RUN=1 replaces the GPU work of
a rank by a stand-in, so what runs here is the spawn / rendezvous / max-over-ranks / gather-by-frame path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update({"BENCH_DRY_RUN": "1"}, **(env_extra or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("config,scaling,frames", [("c2", "weak", 512), ("c4", "strong", 4096)])
def test_gpus_2_spawns_two_ranks(config, scaling, frames):
    p = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--config", config)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout  # ONE JSON line and nothing else (gloo's chatter goes to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == scaling and out["config"]["dry_run"] is True
    # max over ranks of the stand-in times (rank r reports (r+1) ms per step): rank 1's
    assert abs(out["ms_per_step"] - 2.0) < 1e-6
    assert abs(out["value"] - frames * 3 / 6e-3) < 1e-3 * out["value"]


def test_single_rank_runs_in_process():
    p = run_bench("--gpus", "1", "--steps", "2", "--warmup", "0")
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["metric"].startswith("frames/sec at 1920x1200 BGR8")


def test_under_a_launcher_nothing_is_spawned():
    """With RANK / WORLD_SIZE set (torch.distributed.run's contract) bench.py is one rank: world 1 here."""
    p = run_bench("--gpus", "1", "--steps", "2", "--warmup", "0", env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_failed_rank_fails_the_parent():
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "c2", env_extra={"BENCH_DRY_RUN": "0", "HIP_VISIBLE_DEVICES": "",
                                                                                               "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode != 0
