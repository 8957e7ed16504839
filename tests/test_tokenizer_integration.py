"""Row f2 of SURVEY.md section 8: a REAL Hugging Face fast tokenizer through process() (reference call sites
standalone.py:1501-1538 `_requires_manual_special_tokens`, :2123-2135 manual CLS/SEP, :862-866 batch_decode), pinned by
goldens of the reference's own process() with the same tokenizer (tests/golden/g11_process_wordpiece_*.json)."""

from __future__ import annotations

import json

import pytest

from helpers import (
    GOLDEN_DIR,
    WORDPIECE_CASES,
    assert_process_result_matches,
    build_wordpiece_tokenizer,
    golden_stub_forward,
    host_only_model,
    period_splitter,
)


def _run_cases(model, variant, prob_tol, score_tol):
    for case, ref in zip(WORDPIECE_CASES, variant["cases"]):
        assert case["case"] == ref["case"]
        result = model.process(question=case["question"], context=case["context"], sentence_splitter=period_splitter,
                               show_progress=False, return_sentence_metrics=True, return_sentence_texts=True, batch_size=4,
                               **ref["kwargs"])
        assert_process_result_matches(result, ref["expected"], prob_tol=prob_tol, score_tol=score_tol)


@pytest.mark.parametrize("emit", [True, False])
def test_wordpiece_tokenizer_host_pipeline_matches_reference(emit):
    """Stub forward: block layout, manual special tokens, token ranges, fragment decoding and scores are the
    reference's, exactly."""

    meta = json.loads((GOLDEN_DIR / "g11_process_wordpiece_stub.json").read_text(encoding="utf-8"))
    variant = next(v for v in meta["variants"] if v["emit_specials"] == emit)
    model = host_only_model(tokenizer=build_wordpiece_tokenizer(emit), max_length=meta["max_length"], forward=golden_stub_forward)
    model._update_tokenizer_runtime()
    assert model._manual_special_tokens_required == variant["manual_special_tokens_required"] == (not emit)
    if not emit:
        assert (model._manual_cls_token_id, model._manual_sep_token_id) == (1, 2)
    _run_cases(model, variant, prob_tol=1e-6, score_tol=1e-6)


def test_transformers5_fast_tokenizer_without_build_inputs_takes_the_manual_path():
    """The plain transformers >= 5 PreTrainedTokenizerFast has no build_inputs_with_special_tokens at all; the 4.x
    generic fast tokenizer returned the bare concatenation for it.  Same results as that variant."""

    meta = json.loads((GOLDEN_DIR / "g11_process_wordpiece_stub.json").read_text(encoding="utf-8"))
    variant = next(v for v in meta["variants"] if not v["emit_specials"])
    raw = build_wordpiece_tokenizer(False, legacy_methods=False)
    assert not hasattr(raw, "build_inputs_with_special_tokens")
    model = host_only_model(tokenizer=raw, max_length=meta["max_length"], forward=golden_stub_forward)
    model._update_tokenizer_runtime()
    assert model._manual_special_tokens_required
    _run_cases(model, variant, prob_tol=1e-6, score_tol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("emit", [True, False])
def test_wordpiece_tokenizer_process_on_gpu_matches_reference(emit):
    """Real forward on the HIP path: identical kept / removed sentences, probabilities and scores within 1e-3."""

    from open_provence_amd.config import OpenProvenceConfig
    from open_provence_amd.modeling import OpenProvenceModel
    from open_provence_amd.synthetic import synth_state_dict
    from helpers import dims_from_meta

    meta = json.loads((GOLDEN_DIR / "g11_process_wordpiece_model.json").read_text(encoding="utf-8"))
    variant = next(v for v in meta["variants"] if v["emit_specials"] == emit)
    cfg = OpenProvenceConfig(base_model_config=meta["base_model_config"], tokenizer_name_or_path="wordpiece",
                             pruning_config={"hidden_size": meta["base_model_config"]["hidden_size"]},
                             max_length=meta["max_length"], num_labels=1)
    model = OpenProvenceModel(cfg, device="cuda:0", tokenizer=build_wordpiece_tokenizer(emit),
                              state_dict=synth_state_dict(dims_from_meta(meta), meta["weight_seed"]))
    assert model._manual_special_tokens_required == (not emit)
    _run_cases(model, variant, prob_tol=1e-3, score_tol=1e-3)
