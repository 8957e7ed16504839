"""Parity of the HIP path (through the C ABI) against the reference's own outputs (golden fixtures).

Bar (BASELINE.json north_star): per-token logits and rerank scores within 1e-3 of the fp32 CPU
reference -- asserted for the default precision (bf16x3).  The single-pass bf16 mode is what the
reference's GPU default (bf16) would itself give; it is checked to be finite and within the ~1e-1 band
measured for HF-bf16-vs-fp32 (SURVEY.md headline fact 5)."""

import numpy as np
import pytest
import torch

from parity_utils import run_fixture_on_gpu

pytestmark = pytest.mark.gpu

FIXTURES = ["g0b_hd64_refinit", "g0c_hd64_synth", "g1m_meanpool", "g1_xsmall", "g2_gte_varlen",
            "g7_xsmall_refinit", "g8_base_refinit", "g12_prenorm_tf4"]
TOL = 1e-3  # tolerance of the path (north_star): logits within 1e-3 of the CPU reference


def regression_bound(measured: float) -> float:
    """What a fixture's error may grow to before a test fails: 1.3 x the value measured when the kernels' arithmetic last
    changed (tests/golden/parity_bounds.json, written by scripts/parity_bounds.py on the GPU box) plus a small absolute
    floor for the fixtures whose error is fp32 noise, never above 8e-4 -- so the next change of operand format cannot eat
    the rest of the margin under the 1e-3 bar unnoticed (round 3 halved it; every assertion was the bare 1e-3)."""

    return min(8e-4, 1.3 * measured + 2e-5)


def _measured(name: str) -> dict:
    import json

    from helpers import GOLDEN_DIR

    return json.loads((GOLDEN_DIR / "parity_bounds.json").read_text())[name]


@pytest.mark.parametrize("name", FIXTURES)
def test_bf16x3_matches_reference_within_1e3(name):
    rep = run_fixture_on_gpu(name, "bf16x3")
    assert rep["finite"]
    assert rep["prune_max_err"] < TOL, rep
    assert rep["rank_max_err"] < TOL, rep
    assert rep["keep_prob_max_err"] < TOL, rep
    # ... and within the regression bound of this fixture on the kernel set it was measured with
    was = _measured(name)
    assert rep["kernel_set"] == was["kernel_set"], (rep["kernel_set"], was["kernel_set"])
    assert rep["prune_max_err"] < regression_bound(was["prune"]), (rep["prune_max_err"], was["prune"])
    assert rep["rank_max_err"] < regression_bound(was["rank"]), (rep["rank_max_err"], was["rank"])
    assert rep["keep_prob_max_err"] < regression_bound(was["keep_prob"]), (rep["keep_prob_max_err"], was["keep_prob"])
    for i, err in enumerate(rep.get("hidden_max_err", [])):
        assert err < 2e-3, (i, err)  # residual stream grows to |x|~14; 2e-3 abs = 1.5e-4 rel


@pytest.mark.parametrize("name", ["g0c_hd64_synth", "g1_xsmall"])
def test_bf16_single_pass_is_finite_and_in_the_bf16_band(name):
    rep = run_fixture_on_gpu(name, "bf16")
    assert rep["finite"]
    assert rep["prune_max_err"] < 0.5, rep
    assert rep["keep_prob_max_err"] < 0.15, rep


@pytest.mark.parametrize("chunk_rows", [256, 1024])
def test_chunking_does_not_change_results(chunk_rows):
    a = run_fixture_on_gpu("g1_xsmall", "bf16x3", chunk_rows=None, capture=False)
    b = run_fixture_on_gpu("g1_xsmall", "bf16x3", chunk_rows=chunk_rows, capture=False)
    assert b["prune_max_err"] < TOL and b["rank_max_err"] < TOL
    assert abs(a["prune_max_err"] - b["prune_max_err"]) < 1e-4


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x2", {"qk": 2, "pv": 1, "wi": 2}])
def test_generic_tiled_kernels_agree(precision):
    """Three GEMM families exist: row-stationary fused kernels (hidden <= 256, default), k-streamed panel kernels
    (hidden % 256 == 0: base / large / en-gte; the H=768 fixture takes them by default) and the generic 128x128
    tiles (any other shape; forced here through OP_FLAG_FORCE_TILED).  The tiled kernels must give the same answers
    as the defaults: within the bar for bf16x3; for bf16x2 (weights rounded to bf16 -- a deterministic rounding of
    static data) within 3e-4 of the default kernels (same arithmetic, different accumulation order).  A policy that
    drops an ACTIVATION's lo plane is not comparable across kernel families (an fp32 difference of one ulp upstream
    flips bf16 roundings that nothing compensates any more): there only the error band against the reference is
    checked."""

    for name in ["g1_xsmall", "g0c_hd64_synth", "g2_gte_varlen"]:
        tiled = run_fixture_on_gpu(name, precision, flags=1, capture=False, return_outputs=True)
        assert tiled["finite"]
        if precision == "bf16x3":
            assert tiled["prune_max_err"] < TOL and tiled["rank_max_err"] < TOL, tiled
            continue
        default = run_fixture_on_gpu(name, precision, capture=False, return_outputs=True)
        assert tiled["terms"] == default["terms"]
        if precision == "bf16x2":
            scale = max(1.0, float(np.abs(default["prune"]).max()))
            assert np.abs(tiled["prune"] - default["prune"]).max() < 3e-4 * scale, (name, precision)
            assert np.abs(tiled["rank"] - default["rank"]).max() < 3e-4 * scale, (name, precision)
        else:
            assert tiled["prune_max_err"] < 0.3 and default["prune_max_err"] < 0.3, (tiled["prune_max_err"], default["prune_max_err"])
