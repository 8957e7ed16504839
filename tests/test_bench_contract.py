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
    # the headline is the fp32-valued checkpoint on the fp16 + e4m3 kernel set (one launch sequence); the same command
    # times the bf16-rounded weights as a sub-record, and reports the shader clock it held
    assert line["config"]["checkpoint_dtype"] == "fp32" and line["config"]["policy"]["kernel_set"] == "f16-f8-w"
    assert line["config"]["parallelism"] == "single GPU" and "one_pipeline" not in line
    other = line["bf16_checkpoint"]
    assert other["value"] > 0 and other["policy"]["kernel_set"] == "f16-f8" and other["checkpoint_dtype"] == "bf16"
    # the label says what ran (fp16 + e4m3 split operands), not "bf16"; every frac has the clock it was measured at beside
    # it, taken in a SEPARATE probed pass whose own step time is on the record
    assert "fp16" in line["dtype"] and "e4m3" in line["dtype"] and "fp16" in other["dtype"]
    assert 0.5 < line["shader_clock_ghz"]["value"] < 3.0 and line["shader_clock_ghz"]["ms_per_step_with_probe"] > 0
    assert 0.5 < roof["shader_clock_ghz"] < 3.0 and 0.5 < other["shader_clock_ghz"] < 3.0
    assert 0.5 < line["seq_len_2048"]["shader_clock_ghz"] < 3.0
    for rec in (line["config"], other, line["seq_len_2048"]):
        assert {"prune_sum", "prune_abs_sum", "rank_sum"} <= set(rec["output_checksum"])
        # (8 pairs: no stored checksum for this workload; the default workloads compare with tests/golden/bench_checksums.json)
        assert rec["output_checksum"]["stored"].startswith("none")
    assert roof["avg_launch_ms_source"] == "event_bracketed" and roof["frac_in_step"] > 0
    # the panel path (base dims) as a sub-record, both checkpoint dtypes: default flags = the (hi, lo) bf16 kernel sets
    base = line["base_model"]
    # (the Wi GEMM of an fp32-valued checkpoint takes the fp16 + e4m3 format by default: kernel set "bf16x3+wi-f16-f8-w")
    assert base["model"] == "base" and base["fp32_checkpoint"]["kernel_set"] == "bf16x3+wi-f16-f8-w" and base["bf16_checkpoint"]["kernel_set"] == "bf16-weights"
    assert base["fp32_checkpoint"]["value"] > 0 and base["bf16_checkpoint"]["value"] > 0
    assert "bf16" in base["fp32_checkpoint"]["dtype"] and 0.5 < base["fp32_checkpoint"]["shader_clock_ghz"] < 3.0


@pytest.mark.gpu
def test_bench_two_launch_sequences_with_the_bf16_kernel_sets():
    """OPEN_PROVENCE_NO_F8=1 (the (hi, lo) bf16 kernels): the batch runs as two launch sequences on CU-partitioned
    streams and the one-sequence figure is reported beside it."""

    import os

    proc = subprocess.run(
        [sys.executable, str(REPO_ROOT / "bench.py"), "--pairs", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
         "--no-long", "--no-base", "--no-other-dtype", "--weights", "bf16"],
        capture_output=True, text=True, timeout=600, env={**os.environ, "OPEN_PROVENCE_NO_F8": "1"},
    )
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert line["config"]["policy"]["kernel_set"] == "bf16-weights"
    assert "two independent" in line["config"]["parallelism"]
    assert line["one_pipeline"]["value"] > 0 and line["one_pipeline"]["unit"] == "pairs/s"


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_on_a_one_rank_group():
    """The N > 1 branch of bench.py (RCCL process group, ShardPlan partition, gather inside the timed step, MAX
    all-reduce of the elapsed time) executed on hardware with a one-rank group: same contract, same value formula."""

    proc = subprocess.run(
        [sys.executable, str(REPO_ROOT / "bench.py"), "--pairs", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
         "--no-long", "--no-base", "--exercise-gather"],
        capture_output=True, text=True, timeout=600,
    )
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 1 and line["config"]["outputs_finite"] is True and line["value"] > 0
    assert "one_pipeline" not in line  # a rank of a multi-GPU run executes one launch sequence


def test_bench_refuses_outputs_that_differ_from_the_stored_checksum(tmp_path, monkeypatch):
    """bench.py compares every record's output checksum with tests/golden/bench_checksums.json (CPU: the function alone)."""

    import importlib.util

    import torch

    spec = importlib.util.spec_from_file_location("bench_under_test", REPO_ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stored = tmp_path / "checksums.json"
    monkeypatch.setattr(bench, "CHECKSUM_FILE", stored)
    monkeypatch.delenv("OPEN_PROVENCE_BENCH_NO_CHECKSUM", raising=False)
    torch.manual_seed(0)
    prune, rank = torch.randn(4096, 2), torch.randn(8, 1)
    assert bench.output_checksum(prune, rank, "w|8x512|fp32")["stored"].startswith("none")
    plain = bench.output_checksum(prune, rank)
    stored.write_text(json.dumps({"w|8x512|fp32": plain}))
    assert bench.output_checksum(prune, rank, "w|8x512|fp32")["stored"] == "match"
    assert bench.output_checksum(prune + 1e-7, rank, "w|8x512|fp32")["stored"] == "match"  # another kernel set's rounding
    with pytest.raises(SystemExit, match="stored checksum"):
        bench.output_checksum(prune * 1.01, rank, "w|8x512|fp32")
    with pytest.raises(SystemExit, match="stored checksum"):
        bench.output_checksum(prune, rank + 0.01, "w|8x512|fp32")
    monkeypatch.setattr(bench, "_checksum_write_mode", True)
    assert bench.output_checksum(prune * 2, rank, "w|8x512|fp32")["stored"] == "written by this run"
    assert bench._checksums_seen["w|8x512|fp32"]["prune_abs_sum"] == pytest.approx(2 * plain["prune_abs_sum"])
