"""`python bench.py --gpus 2` on the GPU box: two ranks spawned by bench.py itself, one view ray-sharded over them.

The box has ONE GPU, so the ranks share it and the collectives run over gloo (NRF_BENCH_BACKEND=gloo, the rehearsal
switch; RCCL refuses two ranks on one device).  What this pins: the launcher, `n_gpus` = the group's world size, the
sharded headline + `weak_scaling` keys, and that the N = 1 line keeps the driver's contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def bench(*flags, env=None, timeout=420):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *flags], env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return lines[0]


def test_two_ranks_one_sharded_view():
    line = bench("--gpus", "2", "--dtype", "bf16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                 env={"NRF_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["backend"].startswith("gloo")
    assert line["config"]["rays_per_rank"] == 320000 and line["config"]["rays_per_frame"] == 640000
    assert line["value"] > 0 and abs(line["value"] - 640000 / (line["ms_per_frame"] * 1e-3)) < 1e-6 * line["value"]
    assert line["weak_scaling"]["value"] > 0 and line["weak_scaling"]["scaling"] == "weak"
    assert 0.0 < line["roofline"]["frac"] < 1.0


def test_one_rank_line_keeps_the_contract():
    line = bench("--steps", "1", "--warmup", "1", "--dtype", "bf16", "--cpu-seconds", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["dtype"] == "bf16"
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
