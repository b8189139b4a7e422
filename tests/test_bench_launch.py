"""bench.py's rank launcher: `python bench.py --gpus N` must run N ranks and say so.

The reference's launch contract is env:// + one process per GPU (train.py:116-120, trainer.py:14-21).  Without a GPU
the step is empty (NRF_BENCH_DRY_RUN=1): what is checked here is the spawning, the rendezvous on 127.0.0.1, the
max-over-ranks timing collective and `n_gpus` = the process group's world size.  The GPU form of the same check is
tests/test_gpu_bench_launch.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(cmd, extra_env, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("n", [1, 2, 4])
def test_plain_start_spawns_n_ranks(n):
    p = run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "0"], {"NRF_BENCH_DRY_RUN": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout                       # rank 0 alone prints
    assert lines[0]["n_gpus"] == n and lines[0]["dry_run"] is True and lines[0]["value"] is None


@pytest.mark.parametrize("mode", ["render", "train", "testset"])
def test_eight_rank_dry_run_of_every_scaled_mode(mode):
    """The driver's SCALE run is `--gpus 8`: eight ranks rendezvous, barrier and reduce their clocks for each mode that
    has a multi-GPU form (configs[1] ray-sharded frame, configs[2] data-parallel step, configs[4] test set)."""
    p = run([sys.executable, BENCH, "--gpus", "8", "--mode", mode, "--steps", "2", "--warmup", "1"], {"NRF_BENCH_DRY_RUN": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 8 and lines[0]["mode"] == mode and lines[0]["backend"] == "gloo"


def test_torchrun_start_uses_its_world():
    p = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", "29631", BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], {"NRF_BENCH_DRY_RUN": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_world_mismatch_is_an_error():
    p = run([sys.executable, BENCH, "--gpus", "2"], {"NRF_BENCH_DRY_RUN": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("host has a GPU")
    p = run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], {})
    assert p.returncode != 0
    assert not json_lines(p.stdout)
