"""Text front-end: the English splitter around the Punkt model and the auto splitter's
language routing, pinned by the reference's own functions run with the same deterministic Punkt stand-in
(tests/golden/g6_english_splitter.json; reference standalone.py:481-612, 1032-1143)."""

from __future__ import annotations

import json
import re

import pytest

from helpers import GOLDEN_DIR
from open_provence_amd import splitters


class RegexPunkt:
    """Same stand-in as tests/golden/make_golden.py:RegexPunkt (contract of nltk's span_tokenize)."""

    def span_tokenize(self, text: str):
        start = None
        for m in re.finditer(r"\S+", text):
            if start is None:
                start = m.start()
            if re.search(r"[.!?]+[\"')\]]*$", m.group(0)):
                yield (start, m.end())
                start = None
        if start is not None:
            yield (start, len(text.rstrip()))


@pytest.fixture()
def punkt(monkeypatch):
    monkeypatch.setattr(splitters, "_ENGLISH_SENTENCE_TOKENIZER", RegexPunkt())


def test_english_splitter_matches_reference_golden(punkt):
    meta = json.loads((GOLDEN_DIR / "g6_english_splitter.json").read_text(encoding="utf-8"))
    assert splitters.DEFAULT_ENGLISH_SENTENCE_MAX_CHARS == meta["default_max_chars"]
    for run in meta["runs"]:
        split = splitters.create_english_sentence_splitter(run["max_chars"])
        for text, expected in zip(meta["texts"], run["outputs"]):
            got = split(text)
            assert got == expected, (run["max_chars"], text[:40], got, expected)
            assert all(len(piece) <= run["max_chars"] for piece in got)
    with pytest.raises(ValueError):
        splitters.create_english_sentence_splitter(0)


def test_auto_splitter_routes_by_kana_like_the_reference(punkt):
    meta = json.loads((GOLDEN_DIR / "g6_english_splitter.json").read_text(encoding="utf-8"))
    auto = splitters.create_auto_sentence_splitter(japanese_splitter=splitters.simple_sentence_splitter,
                                                   english_splitter=splitters.create_english_sentence_splitter())
    for text, expected in zip(meta["auto_texts"], meta["auto_outputs"]):
        assert auto(text) == expected, (text, auto(text), expected)


def test_missing_nltk_is_a_clear_error(monkeypatch):
    import sys

    monkeypatch.setattr(splitters, "_ENGLISH_SENTENCE_TOKENIZER", None)
    monkeypatch.setitem(sys.modules, "nltk", None)  # import nltk -> ImportError
    with pytest.raises(RuntimeError, match="nltk"):
        splitters.english_sentence_splitter("One. Two.")
