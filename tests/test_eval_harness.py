"""Row f1 of SURVEY.md section 8: the evaluation loop over process() (reference scripts/eval_datasets.py:247-486),
pinned by the golden run of the reference's own function (tests/golden/g4_eval_dataset.json, stub forward)."""

from __future__ import annotations

import json
import math

import pytest

from helpers import GOLDEN_DIR, CharTokenizer, golden_stub_forward, host_only_model
from open_provence_amd.eval_harness import evaluate_dataset, kept_flags, relevance_mask, sentences_from_spans


def _close(got, exp, tol=1e-9):
    if exp is None or got is None:
        return got is exp
    return math.isclose(float(got), float(exp), rel_tol=0.0, abs_tol=tol)


def test_evaluate_dataset_matches_reference_golden():
    meta = json.loads((GOLDEN_DIR / "g4_eval_dataset.json").read_text(encoding="utf-8"))
    model = host_only_model(tokenizer=CharTokenizer(), max_length=meta["max_length"], forward=golden_stub_forward)
    for run in meta["runs"]:
        got = evaluate_dataset(model, meta["dataset"], threshold=run["threshold"], batch_size=4, dataset_label="synthetic")
        exp = run["expected"]
        for key in ("span_total", "span_correct", "span_skipped", "contexts", "confusion_matrix"):
            assert got[key] == exp[key], (run["threshold"], key, got[key], exp[key])
        for key in ("span_accuracy", "mean_compression", "precision", "recall", "f2"):
            assert _close(got[key], exp[key]), (run["threshold"], key, got[key], exp[key])
        assert got["roc_data"]["labels"] == exp["roc_data"]["labels"]
        assert got["roc_data"]["predictions"] == exp["roc_data"]["predictions"]
        assert len(got["roc_data"]["scores"]) == len(exp["roc_data"]["scores"])
        for a, b in zip(got["roc_data"]["scores"], exp["roc_data"]["scores"]):
            assert _close(a, b, 1e-6)
        assert got["process_time_seconds"] >= 0.0 and "inference_seconds" in got["timing"]


def test_span_helpers_known_answers():
    assert relevance_mask([1, 0, 2], 3) == [1, 0, 1]            # mask form: any non-zero is relevant
    assert relevance_mask([0, 2, 7], 4) == [1, 0, 1, 0]         # index form; out-of-range index dropped
    assert relevance_mask(None, 2) == [0, 0] and relevance_mask([1], 0) == []
    with pytest.raises(TypeError):
        relevance_mask(3, 2)
    text = "ab. cd. ef"
    assert sentences_from_spans(text, [[0, 4], [4, 8], [8, 99], [5, 2]]) == ["ab. ", "cd. ", "ef", ""]
    assert sentences_from_spans(text, []) == [text] and sentences_from_spans("", []) == []
    # the cursor only advances over kept sentences, so a later duplicate of a dropped sentence can still match
    assert kept_flags(["ab. ", "cd. ", "ab. "], "ab. ab. ", 3) == [1, 0, 1]
    assert kept_flags(["ab. ", "", "cd. "], "cd. ", 3) == [0, 0, 1]
    assert kept_flags(["x"], "x", 0) == []


def test_empty_dataset_and_missing_queries():
    model = host_only_model(tokenizer=CharTokenizer(), max_length=96, forward=golden_stub_forward)
    out = evaluate_dataset(model, [{"query": None, "texts": ["a."]}], threshold=0.5, batch_size=2)
    assert out["span_total"] == 0 and out["contexts"] == 0 and out["span_accuracy"] is None and out["f2"] is None
    assert out["mean_compression"] is None and out["roc_data"] == {"scores": [], "labels": [], "predictions": []}


def _same(got, exp, path=""):
    if isinstance(exp, float) or isinstance(got, float):
        if exp is None or got is None:
            assert got is exp, path
        elif isinstance(exp, float) and math.isnan(exp):
            assert math.isnan(got), path
        else:
            assert math.isclose(float(got), float(exp), rel_tol=0.0, abs_tol=1e-6), (path, got, exp)
    elif isinstance(exp, dict):
        assert isinstance(got, dict) and set(got) == set(exp), (path, got, exp)
        for key in exp:
            _same(got[key], exp[key], f"{path}.{key}")
    elif isinstance(exp, (list, tuple)):
        assert isinstance(got, (list, tuple)) and len(got) == len(exp), (path, got, exp)
        for i, (g, e) in enumerate(zip(got, exp)):
            _same(g, e, f"{path}[{i}]")
    else:
        assert got == exp, (path, got, exp)


def test_mldr_records_match_reference_golden():
    """Reference scripts/eval_mldr.py:238-524 (build_records): explicit per-passage titles, positive / negative
    statistics, both reranker-score policies, and the single-query single-passage case where process() un-nests."""

    import inspect

    from helpers import period_splitter
    from open_provence_amd.eval_harness import build_mldr_records, clean_title, per_query_lists

    meta = json.loads((GOLDEN_DIR / "g5_mldr_records.json").read_text(encoding="utf-8"))
    model = host_only_model(tokenizer=CharTokenizer(), max_length=meta["max_length"], forward=golden_stub_forward)

    def process_fn(**kwargs):
        return model.process(sentence_splitter=period_splitter, **kwargs)

    process_fn.__signature__ = inspect.signature(model.process)
    for run in meta["runs"]:
        records, stats, n_queries = build_mldr_records(
            process_fn, run["rows"], threshold=meta["threshold"], batch_size=4, log_timing=False,
            use_best_reranker_score=run["use_best_reranker_score"], show_progress=False,
        )
        exp = run["expected"]
        assert n_queries == exp["n_queries"], run["label"]
        _same(records, exp["records"], run["label"] + ".records")
        _same(stats, exp["stats"], run["label"] + ".stats")

    assert clean_title(["A", " ", None, " b "]) == "A b" and clean_title("  ") is None and clean_title(7) == "7"
    assert per_query_lists(None, [2, 1], "x", lambda: 0.0) == [[0.0, 0.0], [0.0]]
    assert per_query_lists("s", [1], "x", str) == [["s"]] and per_query_lists(["a", "b"], [2], "x", str) == [["a", "b"]]
    assert per_query_lists([["a"], "b"], [1, 1], "x", str) == [["a"], ["b"]]
    with pytest.raises(ValueError):
        per_query_lists(["a"], [2], "x", str)
    with pytest.raises(ValueError):
        per_query_lists([["a"], ["b"]], [1, 1, 1], "x", str)
