"""Sentence splitters of the text front-end.

Portable pieces of the reference's front-end (``standalone.py:121-155, 1018-1029, 1129-1143``): the kana
detector, the regex splitter and the auto dispatcher.  The English splitter (NLTK Punkt + the
bullet/overlong heuristics, ``standalone.py:466-612, 1032-1126``) and the Japanese ``fast-bunkai`` splitter
need third-party packages that are not in this image; they are resolved lazily and raise a clear error
when missing -- callers can always pass ``sentence_splitter=<callable>`` or pre-split sentences.
"""

from __future__ import annotations

import math
import re
from typing import Any, Callable, Iterable, Mapping

SentenceSplitter = Callable[[str], list[str]]

SUPPORTED_SPLITTER_LANGUAGES = {"ja", "en", "auto"}
DEFAULT_ENGLISH_SENTENCE_MAX_CHARS = 1200

_BUNKAI = None
_ENGLISH_SENTENCE_TOKENIZER: Any = None  # nltk Punkt model (or any object with span_tokenize), cached
_SIMPLE_PATTERN = re.compile(r".+?(?:。|！|？|!|\?|\n|$)", re.S)
_BULLET_PREFIX_RE = re.compile(r"""^\s*(?:[\-\*••]+|\d{1,4}[:.)]|[A-Za-z]{1}[:.)])\s+""", re.UNICODE)


def _is_kana(cp: int) -> bool:
    return (
        0x3041 <= cp <= 0x3096  # hiragana
        or 0x30A1 <= cp <= 0x30FA  # katakana
        or 0x31F0 <= cp <= 0x31FF  # katakana phonetic extensions
        or 0xFF71 <= cp <= 0xFF9D  # half-width katakana
    )


def is_japanese_fast(text: str, window: int = 500, min_kana_per_window: int = 1) -> bool:
    """Kana-density heuristic: at least ``min_kana_per_window`` kana per ``window`` characters."""

    if not text or text.isascii():
        return False
    required = math.ceil(len(text) / window) * min_kana_per_window
    if required <= 0:
        return False
    seen = 0
    for ch in text:
        cp = ord(ch)
        if cp > 0x7F and _is_kana(cp):
            seen += 1
            if seen >= required:
                return True
    return False


def simple_sentence_splitter(text: str) -> list[str]:
    """Regex splitter: a sentence ends at 。！？!? or a newline; nothing is stripped."""

    if not text:
        return []
    found = [m for m in _SIMPLE_PATTERN.findall(text) if m]
    return found or [text]


def fast_bunkai_sentence_splitter(text: str) -> list[str]:
    try:
        from fast_bunkai import FastBunkai
    except ImportError as exc:
        raise RuntimeError(
            "fast-bunkai is not installed. Install `fast-bunkai` or provide a custom sentence_splitter "
            "(e.g. `simple_sentence_splitter`)."
        ) from exc
    global _BUNKAI
    if _BUNKAI is None:
        _BUNKAI = FastBunkai()
    sentences = [s for s in _BUNKAI(text) if s]
    return sentences or ([text] if text else [])


def _iter_english_blocks(text: str) -> Iterable[tuple[str, int, int]]:
    """Group lines into blocks, starting a new block at every bullet-looking line; yields spans."""

    if not text:
        return
    lines = text.splitlines(keepends=True)
    if not lines:
        yield text, 0, len(text)
        return
    pos = 0
    start = 0
    parts: list[str] = []
    for line in lines:
        line_start = pos
        pos += len(line)
        if _BULLET_PREFIX_RE.match(line.rstrip("\r\n")) and parts:
            block = "".join(parts)
            if block:
                yield block, start, start + len(block)
            parts = [line]
            start = line_start
        else:
            if not parts:
                start = line_start
            parts.append(line)
    if parts:
        block = "".join(parts)
        if block:
            yield block, start, start + len(block)
    if pos < len(text):
        yield text[pos:], pos, len(text)


def split_overlong_sentence(sentence: str, max_chars: int = DEFAULT_ENGLISH_SENTENCE_MAX_CHARS, *, preserve_whitespace: bool = False) -> list[str]:
    """Cut a sentence longer than ``max_chars`` at the last newline, else the last of ``.?!;:``, else hard."""

    working = sentence if preserve_whitespace else sentence.strip()
    if not working:
        return []
    if len(working) <= max_chars:
        return [working]
    chunks: list[str] = []
    start, length = 0, len(working)
    while start < length:
        target = min(start + max_chars, length)
        boundary = None
        newline = working.rfind("\n", start + 1, target)
        if newline >= start + 1:
            boundary = newline + 1
        if boundary is None or boundary <= start:
            for idx in range(target, start, -1):
                if working[idx - 1] in ".?!;:\n":
                    boundary = idx
                    break
        if boundary is None or boundary <= start:
            boundary = target
        chunk = working[start:boundary]
        if not preserve_whitespace:
            chunk = chunk.strip()
        if chunk:
            chunks.append(chunk)
        start = boundary
    return chunks or [working]


def create_english_sentence_splitter(max_chars: int = DEFAULT_ENGLISH_SENTENCE_MAX_CHARS) -> SentenceSplitter:
    """Punkt spans per bullet-delimited block, each span stretched over trailing whitespace, each piece
    clipped to ``max_chars``; whitespace and newlines of the source are preserved."""

    if max_chars <= 0:
        raise ValueError("max_chars must be positive")

    def _punkt():
        global _ENGLISH_SENTENCE_TOKENIZER
        if _ENGLISH_SENTENCE_TOKENIZER is not None:  # loaded once per process, as the reference does (:466-478)
            return _ENGLISH_SENTENCE_TOKENIZER
        try:
            import nltk
            from nltk.tokenize import PunktSentenceTokenizer  # noqa: F401
        except ImportError as exc:
            raise RuntimeError(
                "The English splitter needs `nltk` (+ punkt data), which is not installed; pass "
                "sentence_splitter=<callable>, language='ja', or pre-split sentences instead."
            ) from exc
        try:
            _ENGLISH_SENTENCE_TOKENIZER = nltk.data.load("tokenizers/punkt/english.pickle")
            return _ENGLISH_SENTENCE_TOKENIZER
        except LookupError as exc:
            raise LookupError("Missing NLTK punkt tokenizer data. Run `python -m nltk.downloader punkt`.") from exc

    def split(text: str) -> list[str]:
        if not text:
            return []
        tokenizer = _punkt()
        out: list[str] = []
        for block, b_start, b_end in _iter_english_blocks(text):
            if not block:
                continue
            spans = list(tokenizer.span_tokenize(block))
            if not spans:
                segment = text[b_start:b_end]
                if segment.strip():
                    out.extend(split_overlong_sentence(segment, max_chars, preserve_whitespace=True))
                continue
            for s, e in spans:
                end = b_start + e
                while end < b_end and text[end].isspace():
                    end += 1
                segment = text[b_start + s : end]
                if segment and segment.strip():
                    out.extend(split_overlong_sentence(segment, max_chars, preserve_whitespace=True))
        if out:
            return out
        stripped = text.strip()
        return [stripped] if stripped else []

    return split


def english_sentence_splitter(text: str) -> list[str]:
    return create_english_sentence_splitter()(text)


def create_auto_sentence_splitter(
    *,
    japanese_splitter: SentenceSplitter = fast_bunkai_sentence_splitter,
    english_splitter: SentenceSplitter = english_sentence_splitter,
    kana_window: int = 500,
    min_kana_per_window: int = 1,
) -> SentenceSplitter:
    def split(text: str) -> list[str]:
        if is_japanese_fast(text, window=kana_window, min_kana_per_window=min_kana_per_window):
            return japanese_splitter(text)
        return english_splitter(text)

    return split


def resolve_sentence_splitter(
    splitter: SentenceSplitter | Mapping[str, SentenceSplitter] | None, language: str | None, default_language: str | None = "auto"
) -> SentenceSplitter:
    """callable | {language: callable} | None -> callable (ref: _resolve_sentence_splitter, standalone.py:2007-2039;
    the error messages are the reference's: its callers match on them)."""

    if callable(splitter) and not isinstance(splitter, Mapping):
        return splitter
    if isinstance(splitter, Mapping):  # the caller's own registry, keyed by language
        if language is None:
            raise ValueError("language must be provided when sentence_splitter is a mapping")
        try:
            return splitter[language]
        except KeyError:
            raise ValueError(f"No sentence splitter registered for language '{language}'") from None
    wanted = next((v for v in (language, default_language) if v is not None), "auto")
    builtin = {"auto": create_auto_sentence_splitter, "ja": lambda: fast_bunkai_sentence_splitter, "en": lambda: english_sentence_splitter}
    make = builtin.get(str(wanted).lower())
    if make is None:
        raise ValueError(
            f"Unsupported language code for sentence splitting: '{wanted}'. Supported values are 'auto', 'en', and 'ja'."
        )
    return make()
