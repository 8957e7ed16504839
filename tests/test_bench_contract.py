"""bench.py prints ONE JSON line with the driver's contract keys (GPU box only)."""

from __future__ import annotations

import json
import subprocess
import sys

import pytest

from helpers import REPO_ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


@pytest.mark.gpu
def test_bench_json_line_contract():
    proc = subprocess.run(
        [sys.executable, str(REPO_ROOT / "bench.py"), "--pairs", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
        capture_output=True, text=True, timeout=600,
    )
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert REQUIRED <= set(line), REQUIRED - set(line)
    assert line["unit"] == "pairs/s" and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"] and line["config"]["outputs_finite"] is True
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(line["value"] - 8 * 2 / (line["ms_per_step"] * 2 / 1e3)) / line["value"] < 1e-6
    # single GPU, row-stationary model: the batch ran as two launch sequences, and the one-sequence figure is beside it
    assert "two independent" in line["config"]["parallelism"]
    assert line["one_pipeline"]["value"] > 0 and line["one_pipeline"]["unit"] == "pairs/s"


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_on_a_one_rank_group():
    """The N > 1 branch of bench.py (RCCL process group, ShardPlan partition, gather inside the timed step, MAX
    all-reduce of the elapsed time) executed on hardware with a one-rank group: same contract, same value formula."""

    proc = subprocess.run(
        [sys.executable, str(REPO_ROOT / "bench.py"), "--pairs", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
         "--no-long", "--exercise-gather"],
        capture_output=True, text=True, timeout=600,
    )
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 1 and line["config"]["outputs_finite"] is True and line["value"] > 0
    assert "one_pipeline" not in line  # a rank of a multi-GPU run executes one launch sequence
