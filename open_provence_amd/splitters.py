"""Sentence splitters of the text front-end.

Portable pieces of the reference's front-end (``standalone.py:121-155, 1018-1029, 1129-1143``): the kana
detector, the regex splitter and the auto dispatcher.  The English splitter (NLTK Punkt + the
bullet/overlong heuristics, ``standalone.py:466-612, 1032-1126``) and the Japanese ``fast-bunkai`` splitter
need third-party packages that are not in this image; they are resolved lazily and raise a clear error
when missing -- callers can always pass ``sentence_splitter=<callable>`` or pre-split sentences.
"""

from __future__ import annotations

import itertools
import math
import re
import threading
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


def _builtin(fn: SentenceSplitter) -> SentenceSplitter:
    """Marks one of this module's splitters: stateless apart from lazily created, lock-protected singletons, hence safe
    to call from the worker threads ``process()`` starts on its own (``is_builtin_splitter``)."""

    fn._open_provence_builtin = True  # type: ignore[attr-defined]
    return fn


def is_builtin_splitter(fn: Any) -> bool:
    """True for the splitters this module builds (``resolve_sentence_splitter`` with ``None``, ``simple_sentence_splitter``,
    ...).  A caller-supplied callable is not: ``process()`` never calls it from several threads unless asked to."""

    return bool(getattr(fn, "_open_provence_builtin", False))


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
    with _LAZY_INIT_LOCK:
        if _BUNKAI is None:
            _BUNKAI = FastBunkai()
    sentences = [s for s in _BUNKAI(text) if s]
    return sentences or ([text] if text else [])


_builtin(simple_sentence_splitter)
_builtin(fast_bunkai_sentence_splitter)


def _iter_english_blocks(text: str) -> Iterable[tuple[str, int, int]]:
    """Spans ``(block, start, end)`` of the bullet-delimited blocks of ``text``: a block begins at the start of the text
    and at every later LINE that opens with a bullet / enumeration marker (``standalone.py:485-530`` groups the lines the
    same way).  Formulated over offsets: the line lengths give every line's start, the bullet test picks the cut points,
    consecutive cut points delimit the blocks."""

    if not text:
        return
    lines = text.splitlines(keepends=True)
    offsets = [0, *itertools.accumulate(len(line) for line in lines)]  # offsets[i] = start of line i; [-1] = len(text)
    cuts = [0]
    cuts.extend(offsets[i] for i in range(1, len(lines)) if _BULLET_PREFIX_RE.match(lines[i].rstrip("\r\n")))
    cuts.append(offsets[-1])
    for begin, end in zip(cuts, cuts[1:]):
        if end > begin:
            yield text[begin:end], begin, end


_CUT_MARKS = ".?!;:\n"


def _cut_length(window: str) -> int:
    """How much of ``window`` (the next ``max_chars`` characters of an overlong sentence) the next piece takes: up to and
    including the last newline if there is one past the first character, else up to and including the last of ``.?!;:``
    (or a newline in first position), else all of it (``standalone.py:532-577`` searches backwards for the same places)."""

    newline = window.rfind("\n", 1)
    if newline >= 1:
        return newline + 1
    mark = max(window.rfind(ch) for ch in _CUT_MARKS)
    return mark + 1 if mark >= 0 else len(window)


def split_overlong_sentence(sentence: str, max_chars: int = DEFAULT_ENGLISH_SENTENCE_MAX_CHARS, *, preserve_whitespace: bool = False) -> list[str]:
    """Pieces of at most ``max_chars`` characters, cut behind the last newline of each window, else behind the last of
    ``.?!;:``, else hard at the limit; pieces are stripped (and empty ones dropped) unless ``preserve_whitespace``."""

    rest = sentence if preserve_whitespace else sentence.strip()
    if not rest:
        return []
    if len(rest) <= max_chars:
        return [rest]
    whole, pieces = rest, []
    while rest:
        take = _cut_length(rest[:max_chars])
        piece, rest = rest[:take], rest[take:]
        if not preserve_whitespace:
            piece = piece.strip()
        if piece:
            pieces.append(piece)
    return pieces or [whole]


_SPACE_RUN = re.compile(r"\s*")
_LAZY_INIT_LOCK = threading.Lock()  # the Punkt model / FastBunkai instance are created once, whichever thread comes first


def _load_punkt():
    global _ENGLISH_SENTENCE_TOKENIZER
    with _LAZY_INIT_LOCK:
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


def _english_segments(text: str, tokenizer: Any) -> Iterable[str]:
    """The raw sentence segments of ``text``: Punkt's spans inside every bullet-delimited block, each stretched over the
    whitespace that follows it (never past its block), or the whole block where Punkt finds nothing; whitespace-only
    segments are dropped (``standalone.py:1057-1126``)."""

    for block, begin, end in _iter_english_blocks(text):
        spans = list(tokenizer.span_tokenize(block))
        candidates = [text[begin + s : _SPACE_RUN.match(text, begin + e, end).end()] for s, e in spans] if spans else [block]
        yield from (segment for segment in candidates if segment.strip())


def create_english_sentence_splitter(max_chars: int = DEFAULT_ENGLISH_SENTENCE_MAX_CHARS) -> SentenceSplitter:
    """Punkt spans per bullet-delimited block, each span stretched over trailing whitespace, each piece
    clipped to ``max_chars``; whitespace and newlines of the source are preserved."""

    if max_chars <= 0:
        raise ValueError("max_chars must be positive")

    def split(text: str) -> list[str]:
        if not text:
            return []
        out = [piece for segment in _english_segments(text, _load_punkt())
               for piece in split_overlong_sentence(segment, max_chars, preserve_whitespace=True)]
        if out:
            return out
        stripped = text.strip()
        return [stripped] if stripped else []

    return _builtin(split)


def english_sentence_splitter(text: str) -> list[str]:
    return create_english_sentence_splitter()(text)


_builtin(english_sentence_splitter)


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

    return _builtin(split) if (is_builtin_splitter(japanese_splitter) and is_builtin_splitter(english_splitter)) else split


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
