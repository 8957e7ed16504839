"""Host pipeline of ``process()``: everything either side of the forward that is integer/string work.

Restates the behaviour of the reference's host code (``open_provence/modeling_open_provence_standalone.py``;
"ref:" citations below give the lines whose behaviour each function reproduces) as free functions over
plain Python data so that they can be unit-tested against the reference's own known-answer tests and
the G3 golden fixtures without a GPU.  No arithmetic of the encoder lives here.
"""

from __future__ import annotations

import functools
import math
import os
import heapq
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Callable, Mapping, Sequence

import numpy as np

SentenceSplitter = Callable[[str], list[str]]

DEFAULT_ENGLISH_SENTENCE_MAX_CHARS = 1200  # ref: standalone.py:100


# ---------------------------------------------------------------------------------------------
# data carried between stages
# ---------------------------------------------------------------------------------------------
@dataclass
class FragmentRecord:
    """A run of tokens of one sentence that travels as a unit (ref: _FragmentRecord :990-999)."""

    text: str
    sentence_index: int
    fragment_index: int
    global_index: int
    token_length: int
    token_ids: list[int]


@dataclass
class RawPrediction:
    """Per-block model output (ref: OpenProvenceRawPrediction :451-459)."""

    query: str
    contexts: list[str]
    ranking_score: float | None
    pruning_probs: np.ndarray
    context_ranges: list[tuple[int, int]]
    # process() on the native forward: the per-fragment means of `pruning_probs` over `fragment_token_ranges`, already
    # reduced on the device (op_segment_means: numpy's float32 pairwise order, bit for bit); pruning_probs is then empty
    fragment_means: list[float] | None = None


@dataclass
class ContextState:
    """Everything known about one (query, context) pair while it moves through the pipeline."""

    sentences: list[str]
    fragments: list[FragmentRecord]
    blocks: list[list[FragmentRecord]]
    prefix_length: int
    prefix_sentences: list[str]
    prefix_token_counts: list[int]
    title_is_first_sentence: bool
    original_text: str
    raw_blocks: list[tuple[int, RawPrediction]] = field(default_factory=list)


# ---------------------------------------------------------------------------------------------
# input normalisation (ref: _normalize_inputs :2261-2323)
# ---------------------------------------------------------------------------------------------
def _is_sequence(value: Any) -> bool:
    if isinstance(value, (str, bytes, bytearray)):  # (first: typing.Sequence's instance check costs microseconds per text)
        return False
    return isinstance(value, (list, tuple)) or isinstance(value, Sequence)


def _normalize_collection(values: Sequence[Any]) -> list[Any]:
    return [[str(e) for e in item] if _is_sequence(item) else str(item) for item in values]


def normalize_inputs(question: str | Sequence[str], context: Any) -> tuple[list[str], list[list[Any]], str]:
    """Return (queries, contexts[q][d], structure) with structure in {str, list, aligned, nested}."""

    queries = [question] if isinstance(question, str) else [str(q) for q in question]

    if isinstance(context, str):
        structure = "str"
        contexts: list[list[Any]] = [[context]]
    elif not _is_sequence(context):
        raise ValueError("Unsupported context format")
    elif len(queries) == 1:
        structure = "list"
        contexts = [_normalize_collection(context)]
    else:
        entries = list(context)
        if all(not _is_sequence(entry) for entry in entries):
            if len(entries) != len(queries):
                raise ValueError("Number of contexts must match number of queries")
            structure = "aligned"
            contexts = [[str(entry)] for entry in entries]
        else:
            structure = "nested"
            contexts = []
            for entry in entries:
                if not _is_sequence(entry):
                    raise ValueError("Number of context lists must match number of queries")
                contexts.append(_normalize_collection(entry))

    if structure == "list" and len(queries) != 1:
        raise ValueError("Single list of contexts requires a single query")
    if structure == "nested" and len(contexts) != len(queries):
        raise ValueError("Number of context lists must match number of queries")
    if structure == "str" and len(queries) != 1:
        raise ValueError("Single context string requires a single query")
    if structure in {"str", "list"}:
        contexts = [contexts[0]]
    return queries, contexts, structure


# ---------------------------------------------------------------------------------------------
# titles (ref: _prepare_titles :2325-2360, _extract_first_line_titles :2362-2410,
#         _resolve_titles :2412-2434, _resolve_prefix_sentences :1971-2005)
# ---------------------------------------------------------------------------------------------
def prepare_titles(title: Any, queries: list[str], contexts: list[list[Any]]) -> list[Any]:
    """The ``title=`` argument of ``process()`` -> one entry per query: ``None`` (no titles), the marker
    ``"first_sentence"``, or one title string per context of that query.  Accepted shapes (the reference's contract,
    standalone.py:2325-2360): a single string for every context; for ONE query a flat list with a title per context; for
    several queries either a title per query (shared by its contexts) or a list of per-context titles per query."""

    count = len(queries)
    if title is None:
        return [None] * count
    if isinstance(title, str):
        if title == "first_sentence":
            return [title] * count
        return [[title] * len(per_query) for per_query in contexts]
    if not isinstance(title, Sequence):
        raise ValueError("Unsupported title format")

    def is_nested(entry: Any) -> bool:
        return isinstance(entry, Sequence) and not isinstance(entry, str)

    entries = [[str(v) for v in entry] if is_nested(entry) else str(entry) for entry in title]
    nested = [isinstance(entry, list) for entry in entries]
    if count == 1 and not any(nested):
        return [entries]  # one query: the flat list is its per-context titles
    if len(entries) == count and all(nested):
        return entries
    if len(entries) == count and not any(nested):
        return [[entries[q]] * len(contexts[q]) for q in range(count)]
    raise ValueError("Unsupported title format")


def extract_first_line_titles(contexts: list[list[Any]]) -> tuple[list[list[Any]], list[list[str]]]:
    """Peel the first non-blank line (or pre-split sentence) off every context as its title."""

    new_contexts: list[list[Any]] = []
    titles: list[list[str]] = []
    for group in contexts:
        group_titles: list[str] = []
        new_group: list[Any] = []
        for entry in group:
            if isinstance(entry, list):
                pieces = [str(v) for v in entry]
                keep_from = next((i for i, seg in enumerate(pieces) if seg.strip()), None)
                if keep_from is None:
                    group_titles.append("")
                    new_group.append(pieces)
                else:
                    group_titles.append(pieces[keep_from].rstrip("\r\n"))
                    new_group.append(pieces[keep_from + 1 :])
            else:
                text = str(entry)
                candidate, remainder = "", ""
                if text:
                    lines = text.splitlines(keepends=True)
                    keep_from = next((i for i, line in enumerate(lines) if line.strip()), None)
                    if keep_from is None:
                        remainder = "".join(lines)
                    else:
                        candidate = lines[keep_from].rstrip("\r\n")
                        remainder = "".join(lines[keep_from + 1 :])
                group_titles.append(candidate)
                new_group.append(remainder)
        titles.append(group_titles)
        new_contexts.append(new_group)
    return new_contexts, titles


def resolve_titles(
    queries: list[str], contexts: list[list[Any]], title: Any, *, first_line_as_title: bool
) -> tuple[list[list[Any]], list[Any]]:
    if first_line_as_title:
        if title not in (None, "first_sentence"):
            raise ValueError("first_line_as_title=True cannot be combined with an explicit title override.")
        contexts, payload = extract_first_line_titles(contexts)
    else:
        payload = title
    return contexts, prepare_titles(payload, queries, contexts)


def resolve_prefix_sentences(title_spec: Any, context_idx: int) -> tuple[list[str], bool]:
    """(prefix sentences, title_is_first_sentence).  Explicit titles become prefix sentences whose last
    element ends with exactly one newline; the legacy "first_sentence" marker adds no tokens."""

    prefix: list[str] = []
    first_sentence_is_title = False
    if title_spec == "first_sentence":
        first_sentence_is_title = True
    elif isinstance(title_spec, list):
        raw = title_spec[context_idx] if context_idx < len(title_spec) else None
        if title_spec and isinstance(title_spec[0], list):
            if raw:
                prefix.extend(t.strip() for t in raw if isinstance(t, str) and t.strip())
        elif isinstance(raw, str) and raw.strip():
            prefix.append(raw.strip())
    elif isinstance(title_spec, str) and title_spec.strip():
        prefix.append(title_spec.strip())
    if prefix:
        prefix[-1] = prefix[-1].rstrip("\n") + "\n"
    return prefix, first_sentence_is_title


# ---------------------------------------------------------------------------------------------
# sentences (ref: _collect_candidate_sentences :615-630, _normalize_sentences :640-661,
#            _split_multiline_sentence :582-612, _fallback_sentence :633-637)
# ---------------------------------------------------------------------------------------------
def fallback_sentence(context_text: str, strip_sentences: bool) -> str:
    if not strip_sentences:
        return context_text
    return context_text.strip() or context_text


def split_multiline_sentence(text: str, strip_sentences: bool) -> list[str]:
    whole = [text.strip() if strip_sentences else text]
    if "\n" not in text:
        return whole
    segments = text.splitlines(keepends=not strip_sentences)
    meaningful = [seg for seg in segments if seg.strip()]
    if len(meaningful) <= 1:
        return whole
    if sum(1 for ch in text if ch in ".?!") >= len(meaningful):
        return whole  # already punctuated line by line
    if any(len(seg.strip()) > DEFAULT_ENGLISH_SENTENCE_MAX_CHARS for seg in meaningful):
        return whole
    pieces = [seg.strip() for seg in meaningful] if strip_sentences else list(meaningful)
    pieces = [p for p in pieces if p]
    return pieces or whole


def collect_candidate_sentences(example: Mapping[str, Any], splitter: SentenceSplitter) -> list[str]:
    sentences = [str(s) for s in (example.get("prefix_sentences") or []) if s is not None]
    manual = example.get("manual_sentences")
    if manual is not None:
        sentences.extend(str(s) for s in manual if s is not None)
    else:
        sentences.extend(str(s) for s in splitter(str(example.get("context_text", ""))) if s is not None)
    return sentences


def assign_jobs(contexts: Sequence[Sequence[Any]], world: int) -> list[list[int]]:
    """Owner rank of every (query, context) job of a request, the same on every rank: longest-processing-time-first
    over the contexts' character counts (the cost of splitting, tokenizing and running a context grows with its
    text), ties and equal loads broken by index."""

    world = max(1, int(world))
    costs = []
    for q_idx, per_query in enumerate(contexts):
        for c_idx, entry in enumerate(per_query):
            size = sum(len(str(s)) for s in entry) if isinstance(entry, (list, tuple)) else len(str(entry))
            costs.append((size + 64, q_idx, c_idx))  # + a per-context constant: many tiny contexts are not free
    owner = [[0] * len(per_query) for per_query in contexts]
    loads = [(0, rank) for rank in range(world)]  # a heap of (load, rank): the least loaded rank, the lowest on a tie
    for size, q_idx, c_idx in sorted(costs, key=lambda t: (-t[0], t[1], t[2])):
        load, rank = loads[0]
        heapq.heapreplace(loads, (load + size, rank))
        owner[q_idx][c_idx] = rank
    return owner


def normalize_sentences(raw_sentences: Sequence[str], context_text: str, strip_sentences: bool) -> list[str]:
    out: list[str] = []
    for entry in raw_sentences:
        text = str(entry)
        if not text:
            continue
        out.extend(seg for seg in split_multiline_sentence(text, strip_sentences) if seg)
    return out or [fallback_sentence(context_text, strip_sentences)]


def tokenize_sentences(tokenizer: Any, sentences: Sequence[str]) -> list[list[int]]:
    if not sentences:
        return []
    encoded = tokenizer(list(sentences), add_special_tokens=False, return_attention_mask=False)
    if not isinstance(encoded, Mapping):
        try:
            encoded = dict(encoded)
        except Exception:
            return []
    return [_int_list(ids) for ids in encoded.get("input_ids", [])]


_FAST_ENCODE_ATTR = "_open_provence_fast_encode"


def _fast_encode_backend(tokenizer: Any, probe: Sequence[str]):
    """The Rust ``encode_batch`` of a STOCK Hugging Face fast tokenizer, or None.

    For the stock class ``tokenizer(texts, add_special_tokens=False)`` is ``backend.encode_batch(texts,
    add_special_tokens=False)`` (no truncation, no padding) followed by a Python conversion of every encoding into a
    dict of lists, of which only ``input_ids`` is read here (~30 % of the call).  Taken only when ``__call__`` /
    ``_encode_plus`` are the stock methods, and after the first batch encoded both ways has compared equal."""

    verdict = getattr(tokenizer, _FAST_ENCODE_ATTR, None)
    if verdict is not None:
        return verdict or None
    backend = None
    try:
        from transformers import PreTrainedTokenizerFast as Stock

        cls = type(tokenizer)
        candidate = getattr(tokenizer, "_tokenizer", None)
        if (isinstance(tokenizer, Stock) and hasattr(candidate, "encode_batch") and cls.__call__ is Stock.__call__
                and cls._encode_plus is Stock._encode_plus and probe):
            sample = list(probe[:8])
            public = tokenizer(sample, add_special_tokens=False, return_attention_mask=False)["input_ids"]  # (resets truncation / padding)
            want = [list(ids) for ids in public]
            if [list(e.ids) for e in candidate.encode_batch(sample, add_special_tokens=False)] == want:
                backend = candidate
                # tokenizers >= 0.20: the same encoding without the character offsets nothing here reads (-28 % of the
                # call on the WordPiece tokenizer of the tests); taken when it returns the same ids on the sample
                fast = getattr(candidate, "encode_batch_fast", None)
                use_fast = fast is not None and [list(e.ids) for e in fast(sample, add_special_tokens=False)] == want
                try:
                    setattr(tokenizer, _FAST_ENCODE_ATTR + "_no_offsets", bool(use_fast))
                except Exception:
                    pass
    except Exception:
        backend = None
    try:
        setattr(tokenizer, _FAST_ENCODE_ATTR, backend if backend is not None else False)
    except Exception:
        pass
    return backend


def tokenize_sentence_groups(tokenizer: Any, groups: Sequence[Sequence[str]]) -> list[list[list[int]]]:
    """``[tokenize_sentences(tokenizer, g) for g in groups]`` through ONE tokenizer call over all sentences of all
    groups: a sentence's ids do not depend on its batch companions (no padding, no truncation, no special tokens), and a
    Hugging Face fast tokenizer encodes one large batch on all cores (Rust ``encode_batch``) where a call per context
    pays the Python-side call overhead every time (measured on 256 contexts x 11 sentences: 114 ms -> 31 ms)."""

    flat = [s for g in groups for s in g]
    if len(groups) <= 1 or not flat:
        return [tokenize_sentences(tokenizer, g) for g in groups]
    backend = _fast_encode_backend(tokenizer, flat)
    if backend is not None and backend.truncation is None and backend.padding is None:
        encode = backend.encode_batch_fast if getattr(tokenizer, _FAST_ENCODE_ATTR + "_no_offsets", False) else backend.encode_batch
        ids = [e.ids for e in encode(flat, add_special_tokens=False)]
    else:
        ids = tokenize_sentences(tokenizer, flat)
    if len(ids) != len(flat):  # a tokenizer that does not return one row per sentence: per-group calls, as the reference
        return [tokenize_sentences(tokenizer, g) for g in groups]
    out, at = [], 0
    for g in groups:
        out.append(ids[at : at + len(g)])
        at += len(g)
    return out


def _int_list(ids: Any) -> list[int]:
    """``[int(t) for t in ids]`` without the per-token call when ``ids`` already is a list of Python ints (what HF
    fast tokenizers and this module's own stages hand over): the hot loops below copy ~500 ids per context."""

    # First / last element only: a full scan costs more than the conversion it avoids (measured: +20 % on the preprocess
    # stage).  A stray numpy integer inside such a list is harmless: rows enter the device through pack_rows, which
    # converts through numpy.  The caller's list is returned, NOT a copy: callers below never mutate it and copy where a
    # row leaves this module.
    if type(ids) is list and (not ids or (type(ids[0]) is int and type(ids[-1]) is int)):
        return ids
    return [int(t) for t in ids]


# ---------------------------------------------------------------------------------------------
# fragments and blocks (ref: _split_token_lists :686-713, _decode_and_filter_fragments :846-894,
#   _build_fragment_payload :795-843, _truncate_fragment :2082-2102,
#   _assemble_blocks_from_fragments :2222-2259)
# ---------------------------------------------------------------------------------------------
def split_token_lists(
    token_lists: Sequence[Sequence[int]], max_fragment_tokens: int, *, keep_sentence_boundaries: bool = False
) -> list[tuple[list[int], int, int, int]]:
    """(tokens, sentence_index, fragment_index, global_index) in reading order; empty sentences vanish."""

    step = max(1, int(max_fragment_tokens))
    out: list[tuple[list[int], int, int, int]] = []
    for sentence_index, ids in enumerate(token_lists):
        tokens = ids if type(ids) is list else list(ids)
        if not tokens:
            continue
        if keep_sentence_boundaries and len(tokens) <= max_fragment_tokens:
            out.append((tokens[:], sentence_index, 0, len(out)))
            continue
        for fragment_index, start in enumerate(range(0, len(tokens), step)):
            out.append((tokens[start : start + step], sentence_index, fragment_index, len(out)))
    return out


_FAST_DECODE_ATTR = "_open_provence_fast_decode"


def _fast_decode_backend(tokenizer: Any, probe: Sequence[Sequence[int]]):
    """The Rust ``decode_batch`` of a STOCK Hugging Face fast tokenizer, or None.

    ``tokenizer.batch_decode(seqs, skip_special_tokens=True, clean_up_tokenization_spaces=False)`` is, for the stock
    class, a Python loop of ``backend.decode(ids, skip_special_tokens=True)`` (transformers tokenization_utils_tokenizers
    ``_decode``); ``backend.decode_batch`` returns the same strings from one call on all cores (11 k fragments: 236 ms ->
    79 ms).  Used only when the class has not overridden ``decode`` / ``_decode`` / ``batch_decode``, and only after the
    first batch decoded both ways has compared equal (the verdict is remembered on the tokenizer object)."""

    verdict = getattr(tokenizer, _FAST_DECODE_ATTR, None)
    if verdict is not None:
        return verdict or None
    backend = None
    try:
        from transformers import PreTrainedTokenizerFast as Stock

        cls = type(tokenizer)
        candidate = getattr(tokenizer, "_tokenizer", None)
        if (isinstance(tokenizer, Stock) and hasattr(candidate, "decode_batch") and cls._decode is Stock._decode
                and cls.decode is Stock.decode and cls.batch_decode is Stock.batch_decode and probe):
            sample = [list(ids) for ids in probe[:8]]
            if candidate.decode_batch(sample, skip_special_tokens=True) == list(
                tokenizer.batch_decode(sample, skip_special_tokens=True, clean_up_tokenization_spaces=False)
            ):
                backend = candidate
    except Exception:
        backend = None
    try:
        setattr(tokenizer, _FAST_DECODE_ATTR, backend if backend is not None else False)
    except Exception:
        pass
    return backend


def decode_fragment_texts(tokenizer: Any, sequences: Sequence[Sequence[int]]) -> list[str]:
    """Texts of token-id sequences as the reference decodes fragments (standalone.py:846-894:
    ``skip_special_tokens=True, clean_up_tokenization_spaces=False``)."""

    if not sequences:
        return []
    backend = _fast_decode_backend(tokenizer, sequences)
    if backend is not None:
        return list(backend.decode_batch([ids if type(ids) is list else list(ids) for ids in sequences], skip_special_tokens=True))
    return list(tokenizer.batch_decode(list(sequences), skip_special_tokens=True, clean_up_tokenization_spaces=False))


def _fragment_pieces(tokenizer, token_lists, context_text, max_fragment_tokens, strip_sentences, respect_sentence_boundaries):
    pieces = split_token_lists(
        [_int_list(ids) for ids in token_lists],
        max_fragment_tokens,
        keep_sentence_boundaries=respect_sentence_boundaries,
    )
    if not pieces:
        fallback = tokenizer.encode(fallback_sentence(context_text, strip_sentences), add_special_tokens=False)
        pieces = [(list(fallback), 0, 0, 0)]
    return pieces


def _fragment_records(tokenizer, pieces, texts, strip_sentences) -> list[FragmentRecord]:
    records: list[FragmentRecord] = []
    for text, (tokens, s_idx, f_idx, g_idx) in zip(texts, pieces):
        shown = text.strip() if strip_sentences else text
        if not (shown if strip_sentences else text):
            continue
        records.append(FragmentRecord(shown, s_idx, f_idx, g_idx, len(tokens), tokens))  # `tokens` is this piece's own slice
    if not records:
        tokens, s_idx, f_idx, g_idx = pieces[0]
        text = tokenizer.decode(tokens, skip_special_tokens=True, clean_up_tokenization_spaces=False)
        records.append(
            FragmentRecord(text.strip() if strip_sentences else text, s_idx, f_idx, g_idx, len(tokens), list(tokens))
        )
    return records


def fragmentize(
    tokenizer: Any,
    token_lists: Sequence[Sequence[int]],
    context_text: str,
    max_fragment_tokens: int,
    *,
    strip_sentences: bool,
    respect_sentence_boundaries: bool,
) -> list[FragmentRecord]:
    """Token lists -> fragment records with decoded text; fragments whose text decodes to nothing are
    dropped, and if that empties the context the first fragment is resurrected."""

    pieces = _fragment_pieces(tokenizer, token_lists, context_text, max_fragment_tokens, strip_sentences, respect_sentence_boundaries)
    texts = decode_fragment_texts(tokenizer, [tokens for tokens, _, _, _ in pieces])
    return _fragment_records(tokenizer, pieces, texts, strip_sentences)


def fragmentize_many(
    tokenizer: Any,
    contexts: Sequence[tuple[Sequence[Sequence[int]], str]],
    max_fragment_tokens: int,
    *,
    strip_sentences: bool,
    respect_sentence_boundaries: bool,
) -> list[list[FragmentRecord]]:
    """:func:`fragmentize` of several contexts ``(token_lists, context_text)`` with ONE ``batch_decode`` over the
    fragments of all of them (a fragment's text depends on its own tokens only)."""

    all_pieces = [
        _fragment_pieces(tokenizer, token_lists, text, max_fragment_tokens, strip_sentences, respect_sentence_boundaries)
        for token_lists, text in contexts
    ]
    flat = [tokens for pieces in all_pieces for tokens, _, _, _ in pieces]
    texts = decode_fragment_texts(tokenizer, flat)
    if len(texts) != len(flat):
        return [
            fragmentize(tokenizer, token_lists, text, max_fragment_tokens, strip_sentences=strip_sentences,
                        respect_sentence_boundaries=respect_sentence_boundaries)
            for token_lists, text in contexts
        ]
    out, at = [], 0
    for pieces in all_pieces:
        out.append(_fragment_records(tokenizer, pieces, texts[at : at + len(pieces)], strip_sentences))
        at += len(pieces)
    return out


def truncate_fragment(tokenizer: Any, fragment: FragmentRecord, max_tokens: int) -> FragmentRecord:
    max_tokens = max(1, max_tokens)
    if fragment.token_length <= max_tokens:
        return fragment
    kept = list(fragment.token_ids[:max_tokens])
    text = tokenizer.decode(kept, skip_special_tokens=True, clean_up_tokenization_spaces=False)
    return FragmentRecord(text, fragment.sentence_index, fragment.fragment_index, fragment.global_index, len(kept), kept)


def assemble_blocks(
    tokenizer: Any, fragments: list[FragmentRecord], query_len: int, sep_len: int, max_length: int
) -> list[list[FragmentRecord]]:
    """Greedy, order-preserving packing: len(query)+len(sep)+sum(fragments) <= max_length-2 per block; a
    fragment that cannot fit even alone opens a new block and is truncated to the remaining room."""

    if not fragments:
        return []
    budget = max_length - 2
    base = query_len + sep_len
    solo_capacity = max(1, budget - base)
    blocks: list[list[FragmentRecord]] = []
    current: list[FragmentRecord] = []
    used = base
    for fragment in fragments:
        if used + fragment.token_length <= budget:
            current.append(fragment)
            used += fragment.token_length
            continue
        if current:
            blocks.append(current)
        clipped = truncate_fragment(tokenizer, fragment, solo_capacity)
        current = [clipped]
        used = base + clipped.token_length
    if current:
        blocks.append(current)
    return blocks


# ---------------------------------------------------------------------------------------------
# block -> model input (ref: _prepare_block_inputs :2104-2196, _requires_manual_special_tokens :1501-1538)
# ---------------------------------------------------------------------------------------------
def _first_int(*candidates: Any) -> int | None:
    for c in candidates:
        if isinstance(c, int):
            return c
    return None


def special_token_candidates(tokenizer: Any) -> tuple[list[int], list[int]]:
    special_map = getattr(tokenizer, "special_tokens_map", {}) or {}
    cls_c = [
        getattr(tokenizer, "cls_token_id", None),
        special_map.get("cls_token_id"),
        getattr(tokenizer, "bos_token_id", None),
        special_map.get("bos_token_id"),
    ]
    sep_c = [
        getattr(tokenizer, "sep_token_id", None),
        special_map.get("sep_token_id"),
        getattr(tokenizer, "eos_token_id", None),
        special_map.get("eos_token_id"),
    ]
    return [v for v in cls_c if isinstance(v, int)], [v for v in sep_c if isinstance(v, int)]


def requires_manual_special_tokens(tokenizer: Any) -> bool:
    """True for tokenizers (gte-ModernBERT) whose build_inputs_with_special_tokens drops CLS/SEP."""

    try:
        q = tokenizer.encode("open provence query", add_special_tokens=False)
        c = tokenizer.encode("open provence document", add_special_tokens=False)
    except Exception:
        return False
    if not q or not c:
        return False
    built = [int(t) for t in build_inputs_with_special_tokens(tokenizer, q, c)]
    cls_c, sep_c = special_token_candidates(tokenizer)
    missing_cls = bool(cls_c) and not any(t in cls_c for t in built)
    missing_sep = bool(sep_c) and not any(t in sep_c for t in built)
    return missing_cls or missing_sep


def build_inputs_with_special_tokens(tokenizer: Any, first: Sequence[int], second: Sequence[int]) -> Sequence[int]:
    """``tokenizer.build_inputs_with_special_tokens(first, second)`` as the reference calls it (standalone.py:1513,
    2114).  transformers >= 5 removed that method from the fast-tokenizer class; what the generic
    ``PreTrainedTokenizerFast`` of the 4.x line (the reference's pin) returned for it is the plain concatenation with NO
    special tokens -- the very case the reference's manual CLS/SEP path exists for (:1501-1538) -- so that is the
    answer for a tokenizer without the method."""

    method = getattr(tokenizer, "build_inputs_with_special_tokens", None)
    if callable(method):
        return method(first, second)
    return list(first) + list(second or [])


def _find_subsequence(haystack: Sequence[int], needle: Sequence[int]) -> int:
    n = len(needle)
    if n == 0:
        return -1
    needle = list(needle)
    first = needle[0]
    for idx in range(0, len(haystack) - n + 1):
        if haystack[idx] == first and list(haystack[idx : idx + n]) == needle:
            return idx
    return -1


def prepare_block_inputs(
    tokenizer: Any,
    query_tokens: Sequence[int],
    fragments: Sequence[FragmentRecord],
    *,
    manual_specials: bool = False,
    manual_cls: int | None = None,
    manual_sep: int | None = None,
) -> tuple[list[int], list[int], list[int], list[tuple[int, int]]]:
    """(input_ids, attention_mask, token_type_ids, token range of every fragment inside input_ids)."""

    query = _int_list(query_tokens)
    ctx: list[int] = []
    for frag in fragments:
        ctx.extend(_int_list(frag.token_ids))
    built = _int_list(build_inputs_with_special_tokens(tokenizer, query, ctx))

    if manual_specials:
        ids: list[int] = []
        if manual_cls is not None:
            ids.append(manual_cls)
        ids.extend(query)
        if manual_sep is not None:
            ids.append(manual_sep)
        ids.extend(ctx)
        if manual_sep is not None and ctx:
            ids.append(manual_sep)
    else:
        ids = list(built) if built else query + ctx  # a fresh list: the row is handed to the batch, `built` may alias the tokenizer's

    try:
        type_ids = tokenizer.create_token_type_ids_from_sequences(query, ctx)
        type_ids = _int_list(type_ids) if type_ids is not None else None
    except Exception:
        type_ids = None

    ranges: list[tuple[int, int]] = []
    if ctx:
        start = _find_subsequence(ids, ctx)
        if start < 0:  # (ref :2176-2178; cannot happen on the manual path, whose layout is known)
            if manual_specials:
                start = (1 if manual_cls is not None else 0) + len(query) + (1 if manual_sep is not None else 0)
            else:
                start = len(build_inputs_with_special_tokens(tokenizer, query, []))
        cursor = start
        for frag in fragments:
            ranges.append((cursor, cursor + len(frag.token_ids)))
            cursor += len(frag.token_ids)

    if type_ids is not None and len(type_ids) < len(ids):
        pad = type_ids[-1] if type_ids else 0
        type_ids = type_ids + [pad] * (len(ids) - len(type_ids))
    if type_ids is None:
        ctx_start = ranges[0][0] if ctx else len(ids)
        type_ids = [0] * ctx_start + [1] * (len(ids) - ctx_start)
    return ids, [1] * len(ids), type_ids, ranges


# ---------------------------------------------------------------------------------------------
# preprocess-batch heuristics (ref: _default_preprocess_workers :82-96,
#   _auto_tune_preprocess_loader :2567-2623).  The reference runs one inference pass per DataLoader
#   batch, so these numbers cap the effective forward batch.
# ---------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=1)
def default_preprocess_workers() -> int:
    # cached: psutil walks /sys/devices/system/cpu (~4 ms per call on a 256-thread host) and the answer is static
    try:
        import psutil

        total = psutil.cpu_count(logical=False) or psutil.cpu_count(logical=True)
    except Exception:
        total = os.cpu_count()
    return 0 if total is None else max(0, int(total) - 1)


# GiB of device memory below which a preprocess batch is capped at the paired size (ref :2600-2607); above the last: 192
_DEVICE_BATCH_CAPS = ((12.0, 64), (20.0, 128))
_SMALL_REQUEST_JOBS = 2_000  # below this many jobs the reference starts no worker at all (ref :2591-2592)


def _implicit_worker_count(jobs: int, requested: int) -> int:
    """Workers when the caller gave none: nothing for a small request, else the configured count clipped to the host
    (all cores but one; the whole host when nothing is configured), never more workers than jobs."""

    host = max(0, default_preprocess_workers())
    if jobs < _SMALL_REQUEST_JOBS:
        return 0
    count = min(requested or host, host)
    return min(count, jobs) if jobs else count


def _implicit_preprocess_batch(jobs: int, batch: int, inference_batch: int, device_memory_bytes: int | None) -> int:
    """Preprocess batch when the caller gave none: bounded by the device-memory table (or 32..96 around the inference
    batch without a device figure), by the inference batch and by the number of jobs."""

    if device_memory_bytes:
        gib = device_memory_bytes / float(1024**3)
        cap = next((size for limit, size in _DEVICE_BATCH_CAPS if gib < limit), 192)
    else:
        cap = min(96, max(32, inference_batch))
    bounds = [batch, cap, max(1, inference_batch)]
    if jobs:
        bounds.append(jobs)
    return min(bounds)


def auto_tune_preprocess_loader(
    *,
    total_jobs: int,
    inference_batch_size: int,
    current_workers: int,
    current_preprocess_batch: int,
    current_prefetch: int | None,
    workers_explicit: bool,
    batch_explicit: bool,
    prefetch_explicit: bool,
    device_memory_bytes: int | None,
) -> tuple[int, int, int | None]:
    """(workers, preprocess batch, prefetch factor) as the reference settles them (ref :2567-2623; keyword interface kept:
    the reference's tests call it by keyword).  Explicit values pass through; the implicit ones come from the two rules
    above, and the prefetch factor is ceil(batch / workers) clipped to 2..8 whenever workers run."""

    jobs = max(0, int(total_jobs))
    workers = max(0, int(current_workers))
    batch = max(1, int(current_preprocess_batch))
    if not workers_explicit:
        workers = _implicit_worker_count(jobs, workers)
    if not batch_explicit:
        batch = _implicit_preprocess_batch(jobs, batch, inference_batch_size, device_memory_bytes)
    if prefetch_explicit:
        prefetch = current_prefetch
    else:
        prefetch = max(2, min(8, math.ceil(batch / workers))) if workers > 0 else None
    return workers, batch, prefetch


# ---------------------------------------------------------------------------------------------
# post-processing (ref: _postprocess_contexts :2962-3202)
# ---------------------------------------------------------------------------------------------
@dataclass
class PostprocessResult:
    pruned_contexts: list[list[str]]
    reranking_scores: list[list[float | None]]
    compression_rates: list[list[float]]
    kept_sentences: list[list[list[str]]] | None
    removed_sentences: list[list[list[str]]] | None
    titles: list[list[Any]]
    sentence_probabilities: list[list[list[float]]] | None


def mean_f32_slice(values: np.ndarray) -> float:
    """``float(values.mean())`` for a non-empty 1-D float32 array, bit for bit, without ``ndarray.mean``'s Python-level
    wrapper (numpy/_core/_methods.py:_mean: pairwise float32 sum, divided by the count as ``intp`` -- i.e. in float64
    -- and cast back to float32).  ``process()`` averages thousands of short probability slices per request."""

    if values.dtype != np.float32:
        return float(values.mean())
    return float(np.float32(np.add.reduce(values) / np.intp(values.shape[0])))


def mean_of_floats(values: Sequence[float]) -> float:
    """``float(np.mean(values))`` for a non-empty list of Python floats (float64 pairwise sum / count), bit for bit."""

    if len(values) == 1:
        return float(values[0])
    return float(np.add.reduce(np.asarray(values, dtype=np.float64)) / np.intp(len(values)))


def fragment_token_ranges(state: ContextState, block: Sequence[Any], ranges: Sequence[tuple[int, int]], n_tokens: int) -> list[tuple[int, int]]:
    """The token range each fragment of a block is averaged over, within a row of ``n_tokens`` keep-probabilities:
    the fragment's range shifted LEFT by the token count of the prefix (title) sentences that precede its sentence index
    and clamped -- exactly the arithmetic of :func:`score_fragments` (ref :3075-3081).  ``end <= start`` = empty."""

    counts = state.prefix_token_counts
    out: list[tuple[int, int]] = []
    if not counts:
        for _, (start, end) in zip(block, ranges):
            if start < 0:
                start = 0
            out.append((start, end if end <= n_tokens else n_tokens))
        return out
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + c)
    last = len(counts)
    for fragment, (start, end) in zip(block, ranges):
        k = fragment.sentence_index
        if k > 0:
            offset = offsets[k if k < last else last]
            start -= offset
            end -= offset
        if start < 0:
            start = 0
        out.append((start, end if end <= n_tokens else n_tokens))
    return out


def score_fragments(state: ContextState, use_best_reranker_score: bool) -> tuple[dict[int, list[float]], float | None]:
    """Mean keep-probability of every fragment in every block, and the context's rerank score.

    The fragment's token range is shifted LEFT by the token count of the prefix (title) sentences that
    precede its sentence index -- the reference's behaviour (ref :3075-3082)."""

    per_fragment: dict[int, list[float]] = defaultdict(list)
    ranking: float | None = None
    ordered = sorted(state.raw_blocks, key=lambda item: item[0])
    # offsets[k] = sum(prefix_token_counts[:k]) (a slice past the end sums everything), computed once per context
    counts = state.prefix_token_counts
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + c)
    last = len(counts)
    reduce_f32, f32, intp = np.add.reduce, np.float32, np.intp
    for (_, raw), block in zip(ordered, state.blocks):
        if raw.fragment_means is not None:  # reduced on the device over fragment_token_ranges()
            for fragment, value in zip(block, raw.fragment_means):
                per_fragment[fragment.global_index].append(value)
            if raw.ranking_score is not None:
                if ranking is None:
                    ranking = raw.ranking_score
                elif use_best_reranker_score:
                    ranking = max(ranking, raw.ranking_score)
            continue
        probs = raw.pruning_probs
        n = len(probs)
        fast = isinstance(probs, np.ndarray) and probs.dtype == np.float32  # mean_f32_slice, inlined (thousands of calls)
        for fragment, (start, end) in zip(block, raw.context_ranges):
            k = fragment.sentence_index
            if k > 0:
                offset = offsets[k if k < last else last]
                start = start - offset
                end = end - offset
            if start < 0:
                start = 0
            if end > n:
                end = n
            if end <= start:
                value = 1.0
            elif fast:
                value = float(f32(reduce_f32(probs[start:end]) / intp(end - start)))
            else:
                value = mean_f32_slice(probs[start:end])
            per_fragment[fragment.global_index].append(value)
        if raw.ranking_score is not None:
            if ranking is None:
                ranking = raw.ranking_score
            elif use_best_reranker_score:
                ranking = max(ranking, raw.ranking_score)
    return per_fragment, ranking


def postprocess_contexts(
    queries: list[str],
    contexts: list[list[Any]],
    states: Mapping[tuple[int, int], ContextState],
    *,
    threshold: float,
    always_select_title: bool,
    use_best_reranker_score: bool,
    want_sentence_probabilities: bool,
    want_sentence_texts: bool,
    first_line_as_title: bool,
    zero_score_when_empty: bool,
) -> PostprocessResult:
    out = PostprocessResult([], [], [], [] if want_sentence_texts else None, [] if want_sentence_texts else None, [],
                            [] if want_sentence_probabilities else None)
    for q_idx in range(len(queries)):
        q_pruned: list[str] = []
        q_scores: list[float | None] = []
        q_comp: list[float] = []
        q_kept: list[list[str]] = []
        q_removed: list[list[str]] = []
        q_titles: list[Any] = []
        q_probs: list[list[float]] = []
        for c_idx, entry in enumerate(contexts[q_idx]):
            state = states.get((q_idx, c_idx))
            prefix = tuple(str(p) for p in state.prefix_sentences) if state else ()
            fallback_title: Any = None
            if first_line_as_title and prefix:
                fallback_title = prefix[0] if len(prefix) == 1 else list(prefix)

            if state is None or not state.fragments:
                q_pruned.append(entry)
                q_scores.append(None)
                q_comp.append(0.0)
                q_kept.append([entry] if entry else [])
                q_removed.append([])
                q_titles.append(fallback_title)
                q_probs.append([])
                continue
            if not state.blocks or not state.raw_blocks:
                q_pruned.append(entry)
                q_scores.append(None)
                q_comp.append(0.0)
                q_kept.append(state.sentences)
                q_removed.append([])
                q_titles.append(fallback_title)
                q_probs.append([1.0] * len(state.sentences))
                continue

            per_fragment, ranking = score_fragments(state, use_best_reranker_score)
            per_sentence: dict[int, list[float]] = defaultdict(list)
            for fragment in state.fragments:
                if fragment.global_index in per_fragment:
                    per_sentence[fragment.sentence_index].extend(per_fragment[fragment.global_index])

            sentences = state.sentences
            n_prefix = state.prefix_length
            title_index: int | None = None
            if always_select_title:
                if n_prefix > 0:
                    title_index = 0
                elif state.title_is_first_sentence and len(sentences) > n_prefix:
                    title_index = n_prefix

            averages: list[float] = []
            for s_idx in range(len(sentences)):
                values = per_sentence.get(s_idx)
                avg = mean_of_floats(values) if values else 0.0
                averages.append(max(0.0, min(avg, 1.0)))
            any_above = any(a > threshold for a in averages)
            keep = [a > threshold for a in averages]
            if title_index is not None and any_above:
                keep[title_index] = True

            kept = [s for s, k in zip(sentences, keep) if k]
            removed = [s for s, k in zip(sentences, keep) if not k]
            pruned_text = "".join(s for i, (s, k) in enumerate(zip(sentences, keep)) if k and i >= n_prefix)
            original = state.original_text
            compression = (len(original) - len(pruned_text)) / max(len(original), 1) * 100.0
            if zero_score_when_empty and not pruned_text.strip():
                ranking = 0.0

            if state.prefix_sentences:
                title_value: Any = (
                    state.prefix_sentences[0] if len(state.prefix_sentences) == 1 else list(state.prefix_sentences)
                )
            else:
                title_value = None

            q_pruned.append(pruned_text)
            q_scores.append(ranking)
            q_comp.append(compression)
            q_kept.append(kept)
            q_removed.append(removed)
            q_titles.append(title_value)
            q_probs.append(averages)

        out.pruned_contexts.append(q_pruned)
        out.reranking_scores.append(q_scores)
        out.compression_rates.append(q_comp)
        if out.kept_sentences is not None:
            out.kept_sentences.append(q_kept)
        if out.removed_sentences is not None:
            out.removed_sentences.append(q_removed)
        out.titles.append(q_titles)
        if out.sentence_probabilities is not None:
            out.sentence_probabilities.append(q_probs)
    return out


# ---------------------------------------------------------------------------------------------
# reordering (ref: _apply_reordering :3204-3312)
# ---------------------------------------------------------------------------------------------
def apply_reordering(result: PostprocessResult, top_k: int | None) -> PostprocessResult:
    """Per query: stable sort by score descending (None = -inf), then keep the first top_k."""

    if not result.pruned_contexts:
        return result
    limit = None if top_k is None else max(0, int(top_k))

    def pick(rows: list[Any] | None, order: list[int]) -> list[Any] | None:
        return None if rows is None else [rows[i] for i in order]

    new = PostprocessResult([], [], [], [] if result.kept_sentences is not None else None,
                            [] if result.removed_sentences is not None else None, [],
                            [] if result.sentence_probabilities is not None else None)
    for q_idx, scores in enumerate(result.reranking_scores):
        if scores:
            order = sorted(
                range(len(scores)),
                key=lambda i: float("-inf") if scores[i] is None else float(scores[i]),
                reverse=True,
            )
            if limit is not None:
                order = order[:limit]
        else:
            order = list(range(len(result.pruned_contexts[q_idx])))
        new.pruned_contexts.append(pick(result.pruned_contexts[q_idx], order))
        new.reranking_scores.append(pick(scores, order) if scores else scores)
        new.compression_rates.append(pick(result.compression_rates[q_idx], order))
        if new.kept_sentences is not None:
            new.kept_sentences.append(pick(result.kept_sentences[q_idx], order))
        if new.removed_sentences is not None:
            new.removed_sentences.append(pick(result.removed_sentences[q_idx], order))
        new.titles.append(pick(result.titles[q_idx], order))
        if new.sentence_probabilities is not None:
            new.sentence_probabilities.append(pick(result.sentence_probabilities[q_idx], order))
    return new
