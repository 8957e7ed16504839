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
    # Round 5: the headline is the fp32-valued checkpoint in the REFERENCE'S INITIALISATION (SURVEY.md 8d's recipe), on the
    # kernel set the load-time calibration chose for it -- "f16", single-pass fp16 operands; the same command times the
    # bf16-rounded weights and the O(1) worst-case weights (the headline of rounds 1-4: the all-terms sets) as sub-records
    cfg = line["config"]
    assert cfg["checkpoint_dtype"] == "fp32" and cfg["weights_init"] == "refinit" and cfg["policy"]["kernel_set"] == "f16"
    cal = cfg["calibration"]
    assert cal["chosen_set"] == "f16" and cal["default_set"] == "f16-f8-w" and cal["reference_set"] == "bf16x3"
    assert cal["candidates"]["f16"] <= cal["tolerance"] < cal["candidates"]["bf16"]
    assert "two independent" in cfg["parallelism"] and line["one_pipeline"]["value"] > 0
    other = line["bf16_checkpoint"]
    assert other["value"] > 0 and other["policy"]["kernel_set"] == "f16" and other["checkpoint_dtype"] == "bf16"
    worst = line["worst_case_o1_weights"]
    assert worst["fp32"]["policy"]["kernel_set"] == "f16-f8-w" and worst["bf16"]["policy"]["kernel_set"] == "f16-f8"
    assert worst["fp32"]["calibration"]["chosen_set"] == "f16-f8-w" and worst["fp32"]["value"] > 0 and worst["fp32"]["roofline"]["frac"] > 0
    # the label says what ran, not "bf16"; every frac has the clock it was measured at beside it, taken in a SEPARATE probed
    # pass whose own step time is on the record
    assert "fp16 single pass" in line["dtype"] and "e4m3" in worst["fp32"]["dtype"] and "fp16" in other["dtype"]
    assert 0.5 < line["shader_clock_ghz"]["value"] < 3.0 and line["shader_clock_ghz"]["ms_per_step_with_probe"] > 0
    assert 0.5 < roof["shader_clock_ghz"] < 3.0 and 0.5 < other["shader_clock_ghz"] < 3.0
    assert 0.5 < line["seq_len_2048"]["shader_clock_ghz"] < 3.0
    for rec in (cfg, other, line["seq_len_2048"], worst["fp32"], worst["bf16"]):
        assert {"prune_sum", "prune_abs_sum", "rank_sum"} <= set(rec["output_checksum"])
        # (8 pairs: no stored checksum for this workload; the default workloads compare with tests/golden/bench_checksums.json)
        assert rec["output_checksum"]["stored"].startswith("none")
    # round 6: `frac` is the in-step figure (the bracketed one is secondary), the config says what was REQUESTED under that name,
    # and the untimed clock-settling steps behind the W warm-up steps are on the record
    assert roof["avg_launch_ms_source"] == "in_step" and roof["frac_in_step"] == roof["frac"] and roof["frac_bracketed"] > 0
    assert roof["avg_launch_ms_bracketed"] > 0 and "requested_policy" in line["config"] and "precision" not in line["config"]
    assert line["config"]["clock_settling"]["extra_steps"] >= 3
    # the panel path (base dims) as a sub-record, both checkpoint dtypes: calibrated (default), what op_weights_ready selects
    # without calibration, and the calibration at 2e-4 (which takes the single-pass fp16 set at this depth)
    base = line["base_model"]
    assert base["model"] == "base" and base["weights_init"] == "refinit"
    for wdt, uncal in (("fp32_checkpoint", "bf16x3+wi-f16-f8-w"), ("bf16_checkpoint", "bf16-weights")):
        rec = base[wdt]
        assert rec["value"] > 0 and rec["calibration"]["chosen_set"] == rec["kernel_set"] and 0.5 < rec["shader_clock_ghz"] < 3.0
        assert rec["uncalibrated"]["kernel_set"] == uncal and rec["uncalibrated"]["calibration"] is None
        assert rec["calibrate_2e-4"]["kernel_set"] == "f16" and rec["calibrate_2e-4"]["value"] > rec["uncalibrated"]["value"]


@pytest.mark.gpu
def test_bench_two_launch_sequences_with_the_bf16_kernel_sets():
    """OPEN_PROVENCE_NO_F8=1 (the (hi, lo) bf16 kernels): the batch runs as two launch sequences on CU-partitioned
    streams and the one-sequence figure is reported beside it."""

    import os

    proc = subprocess.run(
        [sys.executable, str(REPO_ROOT / "bench.py"), "--pairs", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
         "--no-long", "--no-base", "--no-other-dtype", "--no-worst-case", "--init", "o1", "--weights", "bf16"],
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
         "--no-long", "--no-base", "--no-worst-case", "--exercise-gather"],
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


def test_bench_help_prints():
    """``--help`` goes through argparse's %-formatting of every help string (a bare ``%`` in one raised instead of printing)."""

    proc = subprocess.run([sys.executable, str(REPO_ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and "--steps" in proc.stdout and "--varlen" in proc.stdout, proc.stderr[-1500:]
