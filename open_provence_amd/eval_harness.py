"""Span-level evaluation loop over ``process()`` -- counterpart of the reference's dataset evaluator.

Reference: ``scripts/eval_datasets.py:247-486`` (``evaluate_dataset``) with its helpers ``_normalize_relevance``
(``:132-146``), ``_extract_sentences`` (``:149-161``) and ``_infer_predictions`` (``:171-184``).  The reference feeds
the annotated spans of every passage to ``process()`` as pre-split sentences (no sentence splitter), decides which
spans were kept by walking the pruned text, and scores them against the gold relevance labels.  This module keeps
the dataset schema (``query``, ``texts``, ``context_spans``, ``context_spans_relevance``), the ``process()`` call and
the result keys, so the reference's report builder can consume its output unchanged.  Pinned by
``tests/golden/g4_eval_dataset.json`` (the reference's function run in the build container).

The model only has to provide ``process(...)`` -- ``OpenProvenceModel`` here, or anything duck-typed like it.
"""

from __future__ import annotations

import inspect
from dataclasses import dataclass, field
from time import perf_counter
from typing import Any, Callable, Iterable, Mapping, Sequence

TIMING_KEYS = ("preprocess_seconds", "assembly_seconds", "inference_seconds", "postprocess_seconds")
OPTIONAL_TIMING_KEYS = (
    "sentence_collect_seconds",
    "sentence_normalize_seconds",
    "tokenize_seconds",
    "fragment_split_seconds",
    "fragment_decode_seconds",
)


def relevance_mask(values: Any, span_count: int) -> list[int]:
    """Gold labels of one passage as a 0/1 list of ``span_count`` entries.  Two encodings exist in the datasets: a
    mask with one entry per span (any non-zero = relevant), or a list of relevant span indices (out-of-range indices
    are dropped).  ``None`` = nothing relevant.  (ref ``:132-146``)"""

    if span_count <= 0:
        return []
    if values is None:
        return [0] * span_count
    if not isinstance(values, Sequence):
        raise TypeError(f"context_spans_relevance must be a sequence, got {type(values)}")
    if len(values) == span_count:
        return [int(int(v) != 0) for v in values]
    mask = [0] * span_count
    for raw in values:
        index = int(raw)
        if 0 <= index < span_count:
            mask[index] = 1
    return mask


def sentences_from_spans(text: str, spans: Sequence[Sequence[int]]) -> list[str]:
    """Character spans -> sentences, clamped to the text; an inverted or empty span becomes ``""`` so that sentence
    and span indices stay aligned.  A passage without spans is one sentence (none if it is empty).  (ref ``:149-161``)"""

    if not spans:
        return [text] if text else []
    size = len(text)
    out = []
    for span in spans:
        lo, hi = max(0, int(span[0])), min(size, int(span[1]))
        out.append(text[lo:hi] if hi > lo else "")
    return out


def kept_flags(sentences: Sequence[str], pruned_text: str, span_count: int) -> list[int]:
    """Which of the first ``span_count`` sentences survived pruning: the pruned text is the concatenation of the kept
    sentences in order, so a cursor walks it and a sentence counts as kept iff it sits at the cursor.  Empty sentences
    are never kept.  (ref ``:171-184``)"""

    if span_count <= 0:
        return []
    flags = []
    cursor = 0
    for sentence in sentences[:span_count]:
        text = sentence or ""
        if text and pruned_text.startswith(text, cursor):
            flags.append(1)
            cursor += len(text)
        else:
            flags.append(0)
    return flags


@dataclass
class SpanCounts:
    total: int = 0
    correct: int = 0
    skipped: int = 0
    tp: int = 0
    fp: int = 0
    tn: int = 0
    fn: int = 0
    scores: list[float] = field(default_factory=list)
    labels: list[int] = field(default_factory=list)
    predictions: list[int] = field(default_factory=list)

    def add(self, gold: Sequence[int], predicted: Sequence[int], probabilities: Sequence[float] | None) -> None:
        self.total += len(gold)
        for index, (g, p) in enumerate(zip(gold, predicted)):
            self.correct += int(g == p)
            if g == 1:
                self.tp += p == 1
                self.fn += p == 0
            else:
                self.fp += p == 1
                self.tn += p == 0
            if probabilities is not None:
                self.scores.append(float(probabilities[index]))
                self.labels.append(int(g))
                self.predictions.append(int(p))


def _at(seq: Any, index: int, default: Any) -> Any:
    return seq[index] if isinstance(seq, Sequence) and not isinstance(seq, (str, bytes)) and index < len(seq) else default


def _timing_summary(outputs: Mapping[str, Any], fallback_total: float) -> dict[str, float]:
    trace = outputs.get("performance_trace")
    if trace is not None and hasattr(trace, "as_dict") and hasattr(trace, "total_seconds"):
        return dict(trace.as_dict())
    payload = outputs.get("timing") or {}
    if not isinstance(payload, Mapping):
        return {}
    summary = {key: float(payload.get(key, 0.0)) for key in TIMING_KEYS}
    summary["total_seconds"] = float(payload.get("total_seconds", fallback_total))
    for key in OPTIONAL_TIMING_KEYS:
        if key in payload:
            summary[key] = float(payload.get(key, 0.0))
    return summary


def evaluate_dataset(
    model: Any,
    dataset: Iterable[Mapping[str, Any]],
    *,
    threshold: float,
    batch_size: int,
    dataset_label: str = "dataset",
    show_progress: bool = False,
    debug_messages: bool = False,
    print_timing_summary: bool = False,
    silent: bool = True,
) -> dict[str, Any]:
    """Run ``model.process`` once over every (query, passages) example of ``dataset`` and score the kept spans.

    Returns the reference's keys: ``span_total, span_correct, span_accuracy, span_skipped, contexts,
    mean_compression, process_time_seconds, precision, recall, f2, confusion_matrix{tp,fp,tn,fn},
    roc_data{scores,labels,predictions}, timing``."""

    questions: list[str] = []
    passages: list[list[list[str]]] = []   # query -> passage -> span sentences
    span_counts: list[list[int]] = []
    gold_raw: list[list[Any]] = []
    for example in dataset:
        question = example.get("query")
        if question is None:
            continue
        texts = example.get("texts") or []
        spans_all = example.get("context_spans") or []
        relevance_all = example.get("context_spans_relevance") or []
        sentences_q, counts_q, gold_q = [], [], []
        for index, text in enumerate(texts):
            spans = spans_all[index] if index < len(spans_all) else []
            sentences_q.append(sentences_from_spans(text, spans))
            counts_q.append(len(spans))
            gold_q.append(relevance_all[index] if index < len(relevance_all) else [])
        questions.append(str(question))
        passages.append(sentences_q)
        span_counts.append(counts_q)
        gold_raw.append(gold_q)

    debug_hook: bool | Callable[[str], None] = False
    if debug_messages and not silent:

        def debug_hook(message: str) -> None:  # type: ignore[misc]
            print(f"[process:{dataset_label}] {message}")

    counts = SpanCounts()
    compression_sum = 0.0
    n_contexts = 0
    process_time = 0.0
    timing: dict[str, float] = {}

    if questions:
        started = perf_counter()
        outputs = model.process(
            question=questions,
            context=passages,
            title=None,
            batch_size=batch_size,
            threshold=threshold,
            sentence_splitter=None,
            show_progress=show_progress,
            debug_messages=debug_hook,
            return_sentence_metrics=True,
            show_inference_progress=show_progress and not silent,
        )
        process_time = perf_counter() - started
        timing = _timing_summary(outputs, process_time)
        if timing:
            process_time = timing.get("total_seconds", process_time)
            if print_timing_summary or not silent:
                print(
                    f"[process:{dataset_label}] total={process_time:.2f}s (pre={timing.get('preprocess_seconds', 0.0):.2f}s "
                    f"asm={timing.get('assembly_seconds', 0.0):.2f}s inf={timing.get('inference_seconds', 0.0):.2f}s "
                    f"post={timing.get('postprocess_seconds', 0.0):.2f}s)"
                )

        pruned_all = outputs["pruned_context"]
        compression_all = outputs["compression_rate"]
        probabilities_all = outputs.get("sentence_probabilities") or []
        for q, sentences_q in enumerate(passages):
            pruned_q = _at(pruned_all, q, [])
            compression_q = _at(compression_all, q, [])
            probabilities_q = _at(probabilities_all, q, [])
            for c, sentences in enumerate(sentences_q):
                n_spans = _at(span_counts[q], c, 0)
                if n_spans > 0:
                    gold = relevance_mask(_at(gold_raw[q], c, []), n_spans)
                    predicted = kept_flags(sentences, _at(pruned_q, c, ""), n_spans)
                    if len(gold) != n_spans or len(predicted) != n_spans:
                        counts.skipped += n_spans
                    else:
                        probabilities = _at(probabilities_q, c, [])
                        have_probabilities = isinstance(probabilities, Sequence) and len(probabilities) >= n_spans
                        counts.add(gold, predicted, probabilities if have_probabilities else None)
                if c < len(compression_q):
                    compression_sum += float(compression_q[c])
                n_contexts += 1

    precision = counts.tp / (counts.tp + counts.fp) if (counts.tp + counts.fp) else None
    recall = counts.tp / (counts.tp + counts.fn) if (counts.tp + counts.fn) else None
    f2 = None
    if precision is not None and recall is not None and (4 * precision + recall) > 0:
        f2 = (5 * precision * recall) / (4 * precision + recall)
    return {
        "span_total": counts.total,
        "span_correct": counts.correct,
        "span_accuracy": counts.correct / counts.total if counts.total else None,
        "span_skipped": counts.skipped,
        "contexts": n_contexts,
        "mean_compression": compression_sum / n_contexts if n_contexts else None,
        "process_time_seconds": process_time,
        "precision": precision,
        "recall": recall,
        "f2": f2,
        "confusion_matrix": {"tp": int(counts.tp), "fp": int(counts.fp), "tn": int(counts.tn), "fn": int(counts.fn)},
        "roc_data": {"scores": counts.scores, "labels": counts.labels, "predictions": counts.predictions},
        "timing": timing,
    }


# ---------------------------------------------------------------------------------------------
# MLDR long-document records (reference: scripts/eval_mldr.py:238-524 ``build_records``)
# ---------------------------------------------------------------------------------------------
def clean_title(value: Any) -> str | None:
    """A title as one stripped string: ``None`` / blank -> None, a sequence of parts -> the non-blank parts joined by
    one space.  (ref ``:254-274``)"""

    if value is None:
        return None
    if isinstance(value, str):
        return value.strip() or None
    if isinstance(value, Sequence):
        parts = [str(item).strip() for item in value if item is not None]
        parts = [part for part in parts if part]
        return " ".join(parts) if parts else None
    return str(value).strip() or None


def _to_lists(obj: Any) -> Any:
    if hasattr(obj, "tolist"):
        try:
            obj = obj.tolist()
        except Exception:  # pragma: no cover - defensive, as the reference
            pass
    if isinstance(obj, (list, tuple)):
        return [_to_lists(item) for item in obj]
    return obj


def per_query_lists(value: Any, docs_per_query: Sequence[int], name: str, fill: Callable[[], Any]) -> list[list[Any]]:
    """Bring one ``process()`` output field to ``[query][doc]`` shape whatever un-nesting ``process()`` applied (scalar
    for one query with one doc, flat list for one query), filling a missing field with ``fill()`` and raising on a
    count mismatch.  (ref ``:330-381``)"""

    if value is None:
        return [[fill() for _ in range(n)] for n in docs_per_query]
    value = _to_lists(value)
    single = len(docs_per_query) == 1
    if not isinstance(value, list):
        if single:
            if docs_per_query[0] != 1:
                raise ValueError(f"process() returned a scalar for '{name}' but expected {docs_per_query[0]} docs.")
            return [[value]]
        return [[fill() for _ in range(n)] for n in docs_per_query]
    if single and (not value or not isinstance(value[0], list)):
        if len(value) != docs_per_query[0]:
            raise ValueError(f"process() returned {len(value)} items for '{name}' but expected {docs_per_query[0]}.")
        return [value]
    if len(value) != len(docs_per_query):
        raise ValueError(f"process() returned {len(value)} query batches for '{name}' but expected {len(docs_per_query)}.")
    out: list[list[Any]] = []
    for q, n in enumerate(docs_per_query):
        item = value[q]
        if isinstance(item, list):
            if len(item) != n:
                raise ValueError(f"process() returned {len(item)} docs for query #{q} in '{name}' but expected {n}.")
            out.append(item)
        elif n == 1:
            out.append([item])
        else:
            raise ValueError(f"process() returned a scalar for query #{q} in '{name}' but expected {n} docs.")
    return out


def build_mldr_records(
    process_fn: Callable[..., Mapping[str, Any]],
    dataset: Iterable[Mapping[str, Any]],
    *,
    threshold: float,
    batch_size: int,
    log_timing: bool = False,
    use_best_reranker_score: bool = True,
    show_progress: bool = False,
) -> tuple[list[dict[str, Any]], dict[str, list[float]], int]:
    """One ``process()`` call over every MLDR row (``query_id``, ``query``, ``positive_passages`` / ``negative_passages``
    of ``{docid, title, text}``) with the passages' own titles, then one record per (query, passage) and the
    positive / negative score and compression samples.  Returns ``(records, stats, n_queries)`` like the reference."""

    stats: dict[str, list[float]] = {"pos_scores": [], "neg_scores": [], "pos_compression": [], "neg_compression": []}
    rows = []
    for row in dataset:
        docs = [(p, 1) for p in row["positive_passages"]] + [(p, 0) for p in row["negative_passages"]]
        if not docs:
            continue
        rows.append({
            "query_id": row["query_id"],
            "query": row["query"],
            "texts": [p["text"] for p, _ in docs],
            "titles": [clean_title(p.get("title") if isinstance(p, Mapping) else None) for p, _ in docs],
            "docids": [p["docid"] for p, _ in docs],
            "labels": [label for _, label in docs],
        })
    if not rows:
        return [], stats, 0
    docs_per_query = [len(r["texts"]) for r in rows]

    kwargs: dict[str, Any] = {
        "question": [r["query"] for r in rows],
        "context": [r["texts"] for r in rows],
        "title": [r["titles"] for r in rows],
        "threshold": threshold,
        "batch_size": batch_size,
        "log_timing": log_timing,
        "use_best_reranker_score": use_best_reranker_score,
        "show_progress": show_progress,
        "return_sentence_texts": True,
    }
    try:  # pass only what this process() understands (the reference also serves third-party models here)
        accepted = set(inspect.signature(process_fn).parameters)
    except (ValueError, TypeError):  # pragma: no cover
        accepted = set(kwargs)
    result = process_fn(**{k: v for k, v in kwargs.items() if k in accepted})
    if "pruned_context" not in result:
        raise KeyError("process() result must include 'pruned_context'.")

    pruned = per_query_lists(result.get("pruned_context"), docs_per_query, "pruned_context", lambda: "")
    scores = per_query_lists(result.get("reranking_score"), docs_per_query, "reranking_score", lambda: None)
    compression = per_query_lists(result.get("compression_rate"), docs_per_query, "compression_rate", lambda: 0.0)
    kept = per_query_lists(result.get("kept_sentences"), docs_per_query, "kept_sentences", lambda: [])
    removed = per_query_lists(result.get("removed_sentences"), docs_per_query, "removed_sentences", lambda: [])
    model_titles = result.get("title")

    records: list[dict[str, Any]] = []
    for q, row in enumerate(rows):
        titles_q = model_titles[q] if isinstance(model_titles, list) and q < len(model_titles) else None
        for d in range(docs_per_query[q]):
            title = clean_title(row["titles"][d])
            if title is None and isinstance(titles_q, list) and d < len(titles_q):
                title = clean_title(titles_q[d])
            score, rate, label = scores[q][d], compression[q][d], row["labels"][d]
            records.append({
                "query_id": row["query_id"],
                "query": row["query"],
                "docid": row["docids"][d],
                "label": label,
                "title": title,
                "original_text": row["texts"][d],
                "pruned_text": pruned[q][d],
                "reranking_score": score,
                "compression_rate": rate,
                "kept_sentences": kept[q][d],
                "removed_sentences": removed[q][d],
            })
            side = "pos" if label == 1 else "neg"
            stats[f"{side}_scores"].append(score if score is not None else float("nan"))
            stats[f"{side}_compression"].append(rate)
    return records, stats, len(rows)
