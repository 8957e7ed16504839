"""Host pipeline of process() against golden outputs of the REAL reference (tests/golden/g3_process_stub*.json).

The forward is replaced by the same deterministic stub on both sides, so every difference would be a
difference in host semantics: input-shape dispatch, titles, sentence collection, fragmenting, block
assembly, special tokens, ranges (incl. the title-offset quirk), sentence means, thresholding,
compression, zero-score rule, reordering/top-k and output un-nesting."""

import json

import pytest

from helpers import (
    GOLDEN_DIR,
    CharTokenizer,
    assert_process_result_matches,
    golden_stub_forward,
    host_only_model,
    period_splitter,
)


def _cases(name):
    with open(GOLDEN_DIR / f"{name}.json", "r", encoding="utf-8") as handle:
        meta = json.load(handle)
    return meta, [pytest.param(meta, case, id=case["case"]) for case in meta["cases"]]


_META_STD, _CASES_STD = _cases("g3_process_stub")
_META_MAN, _CASES_MAN = _cases("g3_process_stub_manual_specials")


def _run(meta, case):
    model = host_only_model(
        CharTokenizer(emit_specials=meta["emit_specials"]), max_length=meta["max_length"], forward=golden_stub_forward
    )
    return model.process(
        question=case["question"],
        context=case["context"],
        sentence_splitter=period_splitter,
        show_progress=False,
        return_sentence_metrics=True,
        return_sentence_texts=True,
        batch_size=4,
        **case["kwargs"],
    )


@pytest.mark.parametrize("meta,case", _CASES_STD)
def test_process_matches_reference_with_stub_forward(meta, case):
    result = _run(meta, case)
    assert_process_result_matches(result, case["expected"], prob_tol=1e-6, score_tol=1e-6)
    assert set(result["timing"]) == {
        "preprocess_seconds", "assembly_seconds", "inference_seconds", "postprocess_seconds", "total_seconds",
        "sentence_collect_seconds", "sentence_normalize_seconds", "tokenize_seconds", "fragment_split_seconds",
        "fragment_decode_seconds",
    }
    assert result["performance_trace"].as_dict() == result["timing"]


@pytest.mark.parametrize("meta,case", _CASES_MAN)
def test_process_manual_special_tokens_path(meta, case):
    model = host_only_model(CharTokenizer(emit_specials=False), max_length=meta["max_length"], forward=golden_stub_forward)
    assert model._manual_special_tokens_required is True
    result = _run(meta, case)
    assert_process_result_matches(result, case["expected"], prob_tol=1e-6, score_tol=1e-6)


def test_optional_keys_follow_flags():
    model = host_only_model(forward=golden_stub_forward)
    plain = model.process("q?", "One. Two.", sentence_splitter=period_splitter, show_progress=False)
    assert "kept_sentences" not in plain and "sentence_probabilities" not in plain
    full = model.process(
        "q?", "One. Two.", sentence_splitter=period_splitter, show_progress=False,
        return_sentence_texts=True, return_sentence_metrics=True,
    )
    assert {"kept_sentences", "removed_sentences", "sentence_probabilities"} <= set(full)


def test_batch_size_and_preprocess_batch_do_not_change_results():
    model = host_only_model(forward=golden_stub_forward)
    case = _META_STD["cases"][4]  # long document, several blocks
    kwargs = dict(sentence_splitter=period_splitter, show_progress=False, return_sentence_metrics=True, threshold=0.5)
    a = model.process(case["question"], case["context"], batch_size=1, **kwargs)
    b = model.process(case["question"], case["context"], batch_size=64, preprocess_batch_size=1, **kwargs)
    assert a["pruned_context"] == b["pruned_context"]
    assert a["sentence_probabilities"] == b["sentence_probabilities"]
    assert a["reranking_score"] == b["reranking_score"]


def test_mismatched_lengths_raise():
    model = host_only_model(forward=golden_stub_forward)
    with pytest.raises(ValueError):
        model.process(["a", "b"], ["only one"], sentence_splitter=period_splitter, show_progress=False)
    with pytest.raises(ValueError):
        model.process("q", "ctx", title="T", first_line_as_title=True, sentence_splitter=period_splitter, show_progress=False)


def test_preprocess_workers_give_identical_results_in_order():
    """``preprocess_workers=N`` (reference: DataLoader worker processes, standalone.py:3589) runs the split / tokenize stage
    on N worker threads with a bounded look-ahead: same jobs, same order, same result as the single-thread pipeline."""

    import json
    import threading

    from helpers import GOLDEN_DIR, assert_process_result_matches

    meta = json.loads((GOLDEN_DIR / "g3_process_stub.json").read_text(encoding="utf-8"))
    seen_threads = set()

    def splitter(text):
        seen_threads.add(threading.current_thread().name)
        return period_splitter(text)

    model = host_only_model(tokenizer=CharTokenizer(), max_length=meta["max_length"], forward=golden_stub_forward)
    for case in meta["cases"]:
        kwargs = dict(question=case["question"], context=case["context"], sentence_splitter=splitter, show_progress=False,
                      return_sentence_metrics=True, return_sentence_texts=True, batch_size=4, **case["kwargs"])
        res = model.process(preprocess_workers=3, **kwargs)
        assert_process_result_matches(res, case["expected"], prob_tol=1e-6, score_tol=1e-6)
        res = model.process(torch_dataloader_kwargs={"num_workers": 2}, **kwargs)
        assert_process_result_matches(res, case["expected"], prob_tol=1e-6, score_tol=1e-6)
    assert any(name.startswith("open-provence-prep") for name in seen_threads), seen_threads


def test_fast_tokenizer_requests_take_worker_threads_by_default_and_change_nothing(monkeypatch):
    """From 128 jobs on, a request with a Hugging Face fast tokenizer AND one of the package's own sentence splitters
    prepares its groups of contexts (split, one ``encode_batch``, one ``decode_batch`` per group) on four worker threads
    without being asked to -- same result as ``preprocess_workers=0``, field for field.  A caller-supplied splitter is
    never called from several threads unless workers were requested (the reference only ever ran splitters in separate
    processes, standalone.py:3589): it may not be thread-safe."""

    import threading

    from helpers import build_wordpiece_tokenizer
    from open_provence_amd import pipeline as pl
    from open_provence_amd.splitters import is_builtin_splitter, simple_sentence_splitter

    words = "the tower is tall boats carry fish and salt to north city harbour many years ago it was new".split()
    sentences = [
        " ".join(f"{words[(i * 7 + k) % len(words)]}" for k in range(5 + i % 6)).capitalize() + "!"
        for i in range(140)
    ]
    contexts = ["\n".join(sentences[(i + d) % len(sentences)] for d in range(1 + i % 4)) for i in range(140)]
    tokenizing_threads = set()
    real_tokenize = pl.tokenize_sentence_groups

    def recording_tokenize(tokenizer, groups):
        tokenizing_threads.add(threading.current_thread().name)
        return real_tokenize(tokenizer, groups)

    monkeypatch.setattr(pl, "tokenize_sentence_groups", recording_tokenize)
    model = host_only_model(tokenizer=build_wordpiece_tokenizer(True), max_length=64, forward=golden_stub_forward)
    kwargs = dict(question="which boats carry salt?", context=contexts, show_progress=False,
                  return_sentence_metrics=True, return_sentence_texts=True, batch_size=16, threshold=0.4)
    assert is_builtin_splitter(simple_sentence_splitter)
    default = model.process(sentence_splitter=simple_sentence_splitter, **kwargs)
    assert any(name.startswith("open-provence-prep") for name in tokenizing_threads), tokenizing_threads
    tokenizing_threads.clear()
    single = model.process(sentence_splitter=simple_sentence_splitter, preprocess_workers=0, **kwargs)
    assert not any(name.startswith("open-provence-prep") for name in tokenizing_threads), tokenizing_threads
    for key in ("pruned_context", "reranking_score", "compression_rate", "kept_sentences", "removed_sentences", "sentence_probabilities", "title"):
        assert default[key] == single[key], key
    assert sum(len(k) for k in default["kept_sentences"]) > 0 and sum(len(r) for r in default["removed_sentences"]) > 0
    # the stage timers are wall-clock seconds of the stage (thread seconds scaled by 1 / workers): never more than the call
    timing = default["timing"]
    assert timing["preprocess_seconds"] <= timing["total_seconds"] + 1e-6

    # a caller's own splitter: one thread unless workers are requested
    seen = set()

    def splitter(text):
        seen.add(threading.current_thread().name)
        return simple_sentence_splitter(text)

    assert not is_builtin_splitter(splitter)
    own = model.process(sentence_splitter=splitter, **kwargs)
    assert seen and not any(name.startswith("open-provence-prep") for name in seen), seen
    seen.clear()
    asked = model.process(sentence_splitter=splitter, preprocess_workers=3, **kwargs)
    assert any(name.startswith("open-provence-prep") for name in seen), seen
    for key in ("pruned_context", "reranking_score", "kept_sentences", "removed_sentences", "sentence_probabilities"):
        assert own[key] == single[key] and asked[key] == single[key], key


def test_ranking_scores_of_a_batch_equal_the_reference_row_by_row_sigmoid():
    """The reference scores one row at a time (torch.sigmoid of a one-row tensor: ATen's scalar path, standalone.py:2913-2916);
    a vectorized sigmoid over the whole batch differs from it by 1 ulp in a few per cent of the rows, and in WHICH rows
    depends on their position -- the batch form must give every row the reference's value wherever it stands."""

    import torch

    from open_provence_amd import pipeline as pl

    model = host_only_model(forward=golden_stub_forward)
    torch.manual_seed(3)
    rank = torch.randn(257, 1) * 4
    states = {(0, i): pl.ContextState(sentences=[], fragments=[], blocks=[[]], prefix_length=0, prefix_sentences=[],
                                      prefix_token_counts=[], title_is_first_sentence=False, original_text="") for i in range(257)}
    chunk = [{"query_idx": 0, "context_idx": i, "block_idx": 0, "texts": []} for i in range(257)]
    keeps = [torch.zeros(1).numpy() for _ in range(257)]
    model._store_raw_predictions(chunk, [[] for _ in range(257)], ["q"], states, rank, keeps)
    got = [states[(0, i)].raw_blocks[0][1].ranking_score for i in range(257)]
    want = [model._ranking_score(rank[i]) for i in range(257)]
    assert got == want  # (torch.sigmoid(rank.reshape(-1)) differs from `want` in ~4 % of the rows on an AVX2 / AVX-512 host)
