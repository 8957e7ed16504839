"""Known-answer unit tests of the host helpers.  The input/expected vectors are the ones the reference's own
tests hold for this path (/root/reference/tests/test_modeling_open_provence.py -- line numbers cited per
test); they are re-stated here against this build's functions, reached through the reference's private
names where the reference's tests do so."""

import numpy as np
import pytest
import torch

from helpers import CharTokenizer, host_only_model

from open_provence_amd import modeling
from open_provence_amd.config import DEFAULT_PROCESS_THRESHOLD, OpenProvenceConfig
from open_provence_amd.modeling import (
    OpenProvenceModel,
    OpenProvenceRawPrediction,
    _collect_candidate_sentences,
    _fragmentize_example,
    _FragmentRecord,
    _normalize_sentences,
    _split_token_lists,
    _tokenize_sentences_with_context,
)
from open_provence_amd.splitters import is_japanese_fast, simple_sentence_splitter


class BlankDecodeTokenizer(CharTokenizer):
    """batch_decode yields whitespace only, decode yields a marker (ref tests :112-119)."""

    def batch_decode(self, batch, **_):
        return ["   " for _ in batch]

    def decode(self, tokens, **_):
        return "fallback"


class DoubleSepTokenizer(CharTokenizer):
    """[CLS] a [SEP] [SEP] b [SEP] (ref tests :122-140)."""

    def build_inputs_with_special_tokens(self, a, b=None):
        b = list(b or [])
        head = [self.cls_token_id, *a, self.sep_token_id, self.sep_token_id]
        return head + b + [self.sep_token_id] if b else head

    def create_token_type_ids_from_sequences(self, a, b=None):
        b = list(b or [])
        return [0] * (len(a) + 3) + ([1] * (len(b) + 1) if b else [])


# ---- config / thresholds (ref tests :198-229) -------------------------------------------------
def test_config_threshold_spellings():
    cfg = OpenProvenceConfig(default_threadshold=0.25)
    assert cfg.default_threadshold == pytest.approx(0.25) and cfg.default_threshold == pytest.approx(0.25)
    with pytest.warns(RuntimeWarning, match="default_threshold"):
        cfg = OpenProvenceConfig(default_threshold=0.3)
    assert cfg.default_threadshold == pytest.approx(0.3)
    with pytest.raises(TypeError):
        OpenProvenceConfig(default_threadshold="abc")


def test_process_threshold_resolution():
    model = OpenProvenceModel.__new__(OpenProvenceModel)
    model.default_threshold = 0.45
    assert model._resolve_process_threshold(None) == pytest.approx(0.45)
    assert model._resolve_process_threshold(0.2) == pytest.approx(0.2)
    del model.default_threshold
    assert model._resolve_process_threshold(None) == pytest.approx(DEFAULT_PROCESS_THRESHOLD)


# ---- input shapes (ref tests :348-377) ----------------------------------------------------------
def test_normalize_inputs_shapes():
    model = OpenProvenceModel.__new__(OpenProvenceModel)
    assert model._normalize_inputs("question", [["s1", "s2"], ["s3"]]) == (["question"], [[["s1", "s2"], ["s3"]]], "list")
    q, c, s = model._normalize_inputs(["q1", "q2"], [[["a1", "a2"], ["b1"]], [["c1"], ["d1", "d2"]]])
    assert (q, s) == (["q1", "q2"], "nested") and c == [[["a1", "a2"], ["b1"]], [["c1"], ["d1", "d2"]]]
    assert model._normalize_inputs("q", "ctx") == (["q"], [["ctx"]], "str")
    assert model._normalize_inputs(["a", "b"], ["x", "y"]) == (["a", "b"], [["x"], ["y"]], "aligned")
    with pytest.raises(ValueError):
        model._normalize_inputs(["a", "b"], ["x"])
    with pytest.raises(ValueError):
        model._normalize_inputs("q", 42)


# ---- titles (ref tests :380-397) ----------------------------------------------------------------
def test_extract_first_line_titles_mixed():
    model = OpenProvenceModel.__new__(OpenProvenceModel)
    contexts = [["Title line\nBody line one\nBody line two", ["", "List Title", "Item A", "Item B"]]]
    updated, titles = model._extract_first_line_titles(contexts)
    assert updated == [["Body line one\nBody line two", ["Item A", "Item B"]]]
    assert titles == [["Title line", "List Title"]]


def test_prefix_sentences_get_single_trailing_newline():
    model = OpenProvenceModel.__new__(OpenProvenceModel)
    assert model._resolve_prefix_sentences("first_sentence", 0) == ([], True)
    assert model._resolve_prefix_sentences(["  T1 ", "T2\n\n"], 1) == (["T2\n"], False)
    assert model._resolve_prefix_sentences([["a", " ", "b "]], 0) == (["a", "b\n"], False)
    assert model._resolve_prefix_sentences(None, 0) == ([], False)


# ---- reordering (ref tests :400-507) ------------------------------------------------------------
def test_apply_reordering_sorts_limits_and_handles_none():
    model = OpenProvenceModel.__new__(OpenProvenceModel)
    out = model._apply_reordering(
        [["docA", "docB", "docC"]], [[0.1, None, 0.9]], [[10.0, 5.0, 1.0]], [[["ka"], ["kb"], ["kc"]]],
        [[["ra"], ["rb"], ["rc"]]], [[None, "Title B", "Title C"]], [[[0.1], [0.2], [0.3]]], top_k=2,
    )
    assert out == (
        [["docC", "docA"]], [[0.9, 0.1]], [[1.0, 10.0]], [[["kc"], ["ka"]]], [[["rc"], ["ra"]]],
        [["Title C", None]], [[[0.3], [0.1]]],
    )
    empty = model._apply_reordering(
        [["docA", "docB"]], [[0.4, 0.2]], [[3.0, 1.0]], [[["ka"], ["kb"]]], [[["ra"], ["rb"]]], [[None, None]], None, top_k=0
    )
    assert empty == ([[]], [[]], [[]], [[]], [[]], [[]], None)
    same = model._apply_reordering([["docA", "docB"]], [[0.9, 0.8]], [[5.0, 2.0]], None, None, [[None, None]], None, top_k=None)
    assert same == ([["docA", "docB"]], [[0.9, 0.8]], [[5.0, 2.0]], None, None, [[None, None]], None)


# ---- post-processing (ref tests :510-657) -------------------------------------------------------
def _post_inputs(prob: float, score: float):
    frag = _FragmentRecord("Sentence 1", 0, 0, 0, 1, [1])
    raw = OpenProvenceRawPrediction("query", ["Sentence 1"], score, np.array([prob], dtype=np.float32), [(0, 1)])
    info = {
        (0, 0): {
            "sentences": ["Sentence 1"], "fragments": [frag], "blocks": [[frag]], "prefix_length": 0,
            "prefix_sentences": [], "prefix_token_counts": [], "title_is_first_sentence": False,
            "original_text": "Sentence 1", "raw_blocks": [(0, raw)],
        }
    }
    return ["query"], [["Sentence 1"]], info


@pytest.mark.parametrize(
    "prob,score,zero_rule,exp_text,exp_score,exp_comp",
    [(0.0, 0.87, True, "", 0.0, 100.0), (0.0, 0.73, False, "", 0.73, 100.0), (1.0, 0.42, True, "Sentence 1", 0.42, 0.0)],
)
def test_postprocess_zero_score_rule(prob, score, zero_rule, exp_text, exp_score, exp_comp):
    model = OpenProvenceModel.__new__(OpenProvenceModel)
    queries, contexts, info = _post_inputs(prob, score)
    pruned, scores, comp, kept, removed, titles, probs, _ = model._postprocess_contexts(
        queries, contexts, info, threshold=0.5, always_select_title=False, use_best_reranker_score=True,
        sentence_probability_groups_requested=False, collect_sentence_texts=True, first_line_as_title=False,
        zero_score_when_empty=zero_rule,
    )
    assert pruned == [[exp_text]] and scores == [[exp_score]] and comp[0][0] == pytest.approx(exp_comp)
    assert kept == [[["Sentence 1"] if exp_text else []]] and removed == [[[] if exp_text else ["Sentence 1"]]]
    assert titles == [[None]] and probs is None


def test_postprocess_title_offset_quirk():
    """SURVEY.md section 8a-P3 known answer: QQ | aaa.bbb.ccc. with keep-logit = position - 8, threshold 0.5;
    with an explicit title every context sentence is scored from positions shifted LEFT by the title length."""

    tok = CharTokenizer()

    def stub(input_ids=None, attention_mask=None, **_):
        pos = torch.arange(input_ids.shape[1], dtype=torch.float32)[None, :].expand(input_ids.shape[0], -1)
        return {"ranking_logits": torch.zeros(input_ids.shape[0], 1),
                "pruning_logits": torch.stack([torch.zeros_like(pos), pos - 8.0], dim=-1)}

    model = host_only_model(tok, max_length=64, forward=stub)
    split = lambda t: [s + "." for s in t.split(".") if s]  # noqa: E731
    plain = model.process("QQ", "aaa.bbb.ccc.", title=None, sentence_splitter=split, show_progress=False,
                          threshold=0.5, return_sentence_metrics=True)
    assert plain["sentence_probabilities"] == pytest.approx([0.113, 0.766, 0.993], abs=2e-3)
    titled = model.process("QQ", "aaa.bbb.ccc.", title="TITLE", sentence_splitter=split, show_progress=False,
                           threshold=0.5, return_sentence_metrics=True)
    assert titled["sentence_probabilities"] == pytest.approx([0.281, 0.113, 0.766, 0.993], abs=2e-3)
    assert titled["title"] == "TITLE\n" and titled["pruned_context"] == "bbb.ccc."


# ---- sentences / fragments (ref tests :660-772) -------------------------------------------------
def test_collect_and_normalize_sentences():
    example = {"context_text": "ignored", "prefix_sentences": ["prefix"], "manual_sentences": ["manual", None]}
    assert _collect_candidate_sentences(example, lambda t: ["split-1", "split-2"]) == ["prefix", "manual"]
    assert _normalize_sentences(["  hello  ", "", "\n"], " context ", True) == ["hello"]
    assert _normalize_sentences([], " context ", True) == ["context"]
    assert _normalize_sentences([], " context ", False) == [" context "]


def test_split_token_lists_known_answer():
    assert _split_token_lists([[1, 2, 3, 4, 5]], max_fragment_tokens=2) == [([1, 2], 0, 0, 0), ([3, 4], 0, 1, 1), ([5], 0, 2, 2)]
    assert _split_token_lists([[1, 2], [], [3, 4, 5]], 2, keep_sentence_boundaries=True) == [
        ([1, 2], 0, 0, 0), ([3, 4], 2, 0, 1), ([5], 2, 1, 2)
    ]


def test_fragmentize_example_variants():
    tok = CharTokenizer()
    res = _fragmentize_example({"context_text": " foo bar baz ", "prefix_sentences": ["  prefix  "]}, tok, 3,
                               lambda t: [" foo ", "bar", " ", "baz"], True)
    assert res["sentences"] == ["prefix", "foo", "bar", "baz"]
    assert (res["fragment_texts"][0], res["fragment_sentence_index"][0], res["fragment_fragment_index"][0]) == ("pre", 0, 0)

    res = _fragmentize_example({"context_text": "context"}, BlankDecodeTokenizer(), 5, lambda t: ["context"], True)
    assert res["fragment_texts"] == ["fallback"] and res["fragment_token_ids"] == [[99, 111, 110, 116, 101]]

    res = _fragmentize_example({"context_text": "こんにちは、可愛いですね"}, tok, 50, lambda t: ["こんにちは、", "可愛いですね"], False,
                               respect_sentence_boundaries=True)
    assert res["fragment_texts"] == ["こんにちは、", "可愛いですね"] and res["fragment_fragment_index"] == [0, 0]

    res = _fragmentize_example({"context_text": "ABCDEFG"}, tok, 3, lambda t: ["ABCDEFG"], False, respect_sentence_boundaries=True)
    assert res["fragment_texts"] == ["ABC", "DEF", "G"] and res["fragment_fragment_index"] == [0, 1, 2]


def test_tokenize_sentences_with_context():
    tok = CharTokenizer()
    assert _tokenize_sentences_with_context(tok, ["abc", "def"], prefix_count=0, context_text="abcdef", strip_sentences=False) == [
        tok.encode("abc"), tok.encode("def")
    ]


# ---- block inputs (ref tests :794-847) ----------------------------------------------------------
@pytest.mark.parametrize("tokenizer,expected_ranges", [(CharTokenizer(), [(3, 6), (6, 9)]), (DoubleSepTokenizer(), [(4, 7), (7, 10)])])
def test_prepare_block_inputs(tokenizer, expected_ranges):
    model = host_only_model(tokenizer, max_length=128)
    q = tokenizer.encode("Q")
    frags = [_FragmentRecord("abc", 0, 0, 0, 3, tokenizer.encode("abc")), _FragmentRecord("def", 1, 0, 1, 3, tokenizer.encode("def"))]
    ids, mask, types, ranges = model._prepare_block_inputs(q, frags)
    ctx = tokenizer.encode("abcdef")
    assert ids == tokenizer.build_inputs_with_special_tokens(q, ctx)
    assert mask == [1] * len(ids) and ranges == expected_ranges
    assert types == tokenizer.create_token_type_ids_from_sequences(q, ctx)


def test_manual_special_tokens_layout():
    tok = CharTokenizer(emit_specials=False)
    model = host_only_model(tok, max_length=128)
    assert model._manual_special_tokens_required and (model._manual_cls_token_id, model._manual_sep_token_id) == (1, 2)
    frags = [_FragmentRecord("ab", 0, 0, 0, 2, tok.encode("ab"))]
    ids, _, types, ranges = model._prepare_block_inputs(tok.encode("Q"), frags)
    assert ids == [1, ord("Q"), 2, ord("a"), ord("b"), 2] and ranges == [(3, 5)]
    assert len(types) == len(ids)


def test_block_assembly_budget_and_truncation():
    tok = CharTokenizer()
    model = host_only_model(tok, max_length=20)  # budget 18; query 4 + sep 1 -> 13 tokens of context per block
    frags = [_FragmentRecord("x" * n, i, 0, i, n, tok.encode("x" * n)) for i, n in enumerate([6, 6, 6, 30])]
    blocks = model._assemble_blocks_from_fragments(4, 1, frags)
    assert [[f.token_length for f in b] for b in blocks] == [[6, 6], [6], [13]]
    assert blocks[2][0].text == "x" * 13 and blocks[2][0].global_index == 3


# ---- preprocess-batch heuristics (ref tests :1066-1172; standalone.py:2567-2623) ----------------
def test_auto_tune_small_jobs_disable_workers_and_cap_batch(monkeypatch):
    model = host_only_model()
    monkeypatch.setattr(modeling.pl, "default_preprocess_workers", lambda: 8)
    monkeypatch.setattr(OpenProvenceModel, "_estimate_device_memory_bytes", lambda self: None)
    w, b, p = model._auto_tune_preprocess_loader(
        total_jobs=100, inference_batch_size=256, current_workers=8, current_preprocess_batch=256, current_prefetch=None,
        workers_explicit=False, batch_explicit=False, prefetch_explicit=False,
    )
    assert (w, b, p) == (0, 96, None)
    monkeypatch.setattr(OpenProvenceModel, "_estimate_device_memory_bytes", lambda self: 288 * 1024**3)
    w, b, p = model._auto_tune_preprocess_loader(
        total_jobs=5000, inference_batch_size=256, current_workers=0, current_preprocess_batch=256, current_prefetch=None,
        workers_explicit=False, batch_explicit=False, prefetch_explicit=False,
    )
    assert (w, b, p) == (8, 192, 8)
    w, b, p = model._auto_tune_preprocess_loader(
        total_jobs=5000, inference_batch_size=256, current_workers=2, current_preprocess_batch=512, current_prefetch=3,
        workers_explicit=True, batch_explicit=True, prefetch_explicit=True,
    )
    assert (w, b, p) == (2, 512, 3)


# ---- portable splitters -------------------------------------------------------------------------
def test_portable_splitters():
    assert simple_sentence_splitter("寿司が好きです。ラーメンも好きです。") == ["寿司が好きです。", "ラーメンも好きです。"]
    assert simple_sentence_splitter("") == [] and simple_sentence_splitter("no end") == ["no end"]
    assert is_japanese_fast("寿司が好きです。") and not is_japanese_fast("Sushi is tasty.") and not is_japanese_fast("漢字")
    model = OpenProvenceModel.__new__(OpenProvenceModel)
    with pytest.raises(ValueError):
        model._resolve_sentence_splitter(None, "es")
    with pytest.raises(ValueError):
        model._resolve_sentence_splitter({"en": simple_sentence_splitter}, None)
    assert model._resolve_sentence_splitter({"ja": simple_sentence_splitter}, "ja") is simple_sentence_splitter


def test_forward_boundary_accepts_mapping_and_tuple_protocol():
    """The boundary accepts any Mapping with the two logits (ref tests :940-949, standalone.py:1540-1555)."""

    model = OpenProvenceModel.__new__(OpenProvenceModel)
    out = {"logits": torch.ones(2, 1), "pruning_logits": torch.zeros(2, 3, 2)}
    assert model._extract_model_output(out, "ranking_logits") is out["logits"]
    with pytest.raises(KeyError):
        model._extract_model_output({}, "pruning_logits")


def test_fast_means_are_bit_identical_to_numpy_mean():
    """The post-processing averages go through ``ndarray.mean`` / ``np.mean`` in the reference (standalone.py:3075-3082,
    3116-3120); the drop-in's cheaper formulations must return the same bits (sentence keep/drop is a strict ``>``)."""

    import numpy as np

    from open_provence_amd.pipeline import mean_f32_slice, mean_of_floats

    rng = np.random.default_rng(0)
    probs = rng.random(4096).astype(np.float32)
    tiny = (rng.random(4096) * 1e-6).astype(np.float32)
    for _ in range(3000):
        n = int(rng.integers(1, 700))
        start = int(rng.integers(0, 4096 - n))
        for arr in (probs, tiny):
            seg = arr[start : start + n]
            assert mean_f32_slice(seg) == float(seg.mean())
    assert mean_f32_slice(probs[:5].astype(np.float64)) == float(probs[:5].astype(np.float64).mean())
    for _ in range(2000):
        n = int(rng.integers(1, 12))
        values = [float(v) for v in rng.random(n)]
        assert mean_of_floats(values) == float(np.mean(values))


def test_device_side_fragment_means_path_equals_the_token_path():
    """process() on the native forward hands ``score_fragments`` per-fragment means reduced on the device over
    ``fragment_token_ranges`` (op_segment_means); the ranges must be exactly the ones the token path averages over
    (ref :3075-3081), including the title-offset shift, clamping and empty ranges."""

    import numpy as np

    from open_provence_amd.pipeline import ContextState, RawPrediction, fragment_token_ranges, score_fragments

    rng = np.random.default_rng(3)
    for prefix_counts in ([], [4], [3, 5]):
        n_prefix = len(prefix_counts)
        blocks, raws_tokens, raws_means = [], [], []
        gidx = 0
        for b in range(3):
            n_tokens = int(rng.integers(20, 60))
            probs = rng.random(n_tokens).astype(np.float32)
            block, ranges = [], []
            pos = 3
            for s in range(int(rng.integers(2, 6))):
                length = int(rng.integers(0, 12))
                sent = n_prefix + len(block) if rng.random() < 0.8 else int(rng.integers(0, n_prefix + 1))
                block.append(_FragmentRecord(f"s{gidx}", sent, 0, gidx, n_prefix + 1 + gidx, [1]))
                start = pos + sum(prefix_counts) if n_prefix and sent >= n_prefix else pos
                ranges.append((start, start + length + (40 if s == 0 and b == 2 else 0)))  # one range runs off the row
                pos += length
                gidx += 1
            blocks.append(block)
            raws_tokens.append((b, RawPrediction("q", ["c"], 0.1 * b, probs, ranges)))
        state = ContextState.__new__(ContextState)
        state.blocks = blocks
        state.prefix_token_counts = list(prefix_counts)
        state.raw_blocks = raws_tokens
        want, want_rank = score_fragments(state, True)
        for (b, raw), block in zip(raws_tokens, blocks):
            segs = fragment_token_ranges(state, block, raw.context_ranges, len(raw.pruning_probs))
            means = [1.0 if e <= s else float(raw.pruning_probs[s:e].mean()) for s, e in segs]
            raws_means.append((b, RawPrediction("q", ["c"], raw.ranking_score, np.zeros(0, np.float32), raw.context_ranges, means)))
        state.raw_blocks = raws_means
        got, got_rank = score_fragments(state, True)
        assert dict(got) == dict(want) and got_rank == want_rank


@pytest.mark.parametrize("kind", ["char", "wordpiece"])
def test_grouped_tokenize_and_decode_equal_the_per_context_calls(kind):
    """``tokenize_sentence_groups`` / ``fragmentize_many`` (one tokenizer call per GROUP of contexts) return exactly what
    the per-context calls of the reference's stage return (standalone.py:2198-2259): same ids, same fragment records --
    including a context whose sentences decode to nothing (first fragment resurrected) and an empty group."""

    from helpers import build_wordpiece_tokenizer
    from open_provence_amd import pipeline as pl

    tok = CharTokenizer() if kind == "char" else build_wordpiece_tokenizer(True)
    groups = [
        ["alpha beta gamma. ", "delta epsilon zeta eta theta iota kappa lambda mu. ", "nu"],
        ["   "],
        [],
        ["omega " * 40, "alpha."],
        ["beta"],
    ]
    grouped = pl.tokenize_sentence_groups(tok, groups)
    single = [pl.tokenize_sentences(tok, g) for g in groups]
    assert grouped == single and [len(g) for g in grouped] == [len(g) for g in groups]
    contexts = [(ids, "".join(g)) for ids, g in zip(grouped, groups)]
    for strip in (False, True):
        for respect in (False, True):
            many = pl.fragmentize_many(tok, contexts, 7, strip_sentences=strip, respect_sentence_boundaries=respect)
            one = [pl.fragmentize(tok, ids, text, 7, strip_sentences=strip, respect_sentence_boundaries=respect) for ids, text in contexts]
            assert many == one
            assert all(len(records) >= 1 for records in many)


def test_grouped_tokenize_falls_back_when_the_tokenizer_drops_rows():
    """A tokenizer that does not answer one row per sentence for a flat batch is called per group, as before."""

    from open_provence_amd import pipeline as pl

    class Lossy(CharTokenizer):
        def __call__(self, texts, **kw):
            out = super().__call__(texts, **kw)
            if len(texts) > 2:
                out["input_ids"] = out["input_ids"][:-1]
            return out

    tok = Lossy()
    groups = [["ab", "cd"], ["ef"], ["gh", "ij"]]
    assert pl.tokenize_sentence_groups(tok, groups) == [pl.tokenize_sentences(tok, g) for g in groups]


def test_fast_decode_backend_only_for_stock_tokenizers_and_equal_to_batch_decode():
    """``decode_fragment_texts`` takes the Rust ``decode_batch`` shortcut for a stock fast tokenizer only, after checking
    it against ``batch_decode`` on the first batch; a subclass that overrides decoding keeps its own ``batch_decode``."""

    from helpers import build_wordpiece_tokenizer
    from open_provence_amd import pipeline as pl

    tok = build_wordpiece_tokenizer(True)
    seqs = tok(["The tower is tall.", "It was built long ago!", "boats carry fish and salt", "zzz qqq"], add_special_tokens=False)["input_ids"]
    seqs.append([1] + seqs[0] + [2])  # special tokens are skipped
    expected = tok.batch_decode(seqs, skip_special_tokens=True, clean_up_tokenization_spaces=False)
    assert pl.decode_fragment_texts(tok, seqs) == expected
    assert getattr(tok, pl._FAST_DECODE_ATTR) is tok._tokenizer  # the shortcut was accepted
    assert pl.decode_fragment_texts(tok, []) == []

    class Shouting(type(tok)):
        def batch_decode(self, sequences, **kw):
            return [t.upper() for t in super().batch_decode(sequences, **kw)]

    loud = build_wordpiece_tokenizer(True)
    loud.__class__ = Shouting
    assert pl.decode_fragment_texts(loud, seqs) == [t.upper() for t in expected]
    assert getattr(loud, pl._FAST_DECODE_ATTR) is False
    assert pl.decode_fragment_texts(CharTokenizer(), [[5, 6, 7]]) == CharTokenizer().batch_decode([[5, 6, 7]], skip_special_tokens=True, clean_up_tokenization_spaces=False)


def test_fast_encode_backend_equals_the_public_call_and_respects_backend_truncation():
    from helpers import build_wordpiece_tokenizer
    from open_provence_amd import pipeline as pl

    tok = build_wordpiece_tokenizer(True)
    groups = [["The tower is tall.", "It was built long ago!"], ["boats carry fish and salt " * 30], ["zzz qqq", "a"]]
    expected = [tok(list(g), add_special_tokens=False)["input_ids"] for g in groups]
    assert pl.tokenize_sentence_groups(tok, groups) == expected
    assert getattr(tok, pl._FAST_ENCODE_ATTR) is tok._tokenizer
    # a truncation left enabled on the backend (someone called the tokenizer with truncation=True in between) switches
    # the shortcut off for that call: the public call resets it, as the reference's per-context call does
    tok._tokenizer.enable_truncation(max_length=4)
    assert pl.tokenize_sentence_groups(tok, groups) == expected
    assert tok._tokenizer.truncation is None
