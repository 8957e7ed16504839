#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Build-container only (needs /root/reference and the third-party ``transformers`` ModernBERT); the
outputs (``*.npz`` + ``*.json``) are committed, this script documents how they were made.  Recipe =
SURVEY.md Appendix B:

  1. stub ``nltk`` so ``modeling_open_provence_standalone.py:44-50`` imports;
  2. load that single file by path (never ``import open_provence``);
  3. replace ``AutoTokenizer.from_pretrained`` by a char-code tokenizer (tests/helpers.py), as the
     reference's own tests do (tests/test_modeling_open_provence.py:143-183);
  4. build ``OpenProvenceConfig(base_model_config={modernbert dims})`` -> ``OpenProvenceModel``;
  5. either keep the reference's own random init (weights stored in the fixture) or overwrite the
     weights with ``open_provence_amd.synthetic.synth_state_dict(dims, seed)`` (only the seed is stored);
  6. run ``model(input_ids, attention_mask)`` / ``model.process(...)`` on CPU fp32 and store inputs + outputs.

Usage:  python tests/golden/make_golden.py [--only NAME ...]
"""

from __future__ import annotations

import argparse
import importlib.util
import json
import sys
import types
from pathlib import Path
from typing import Any

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

from helpers import WORDPIECE_CASES, CharTokenizer, build_wordpiece_tokenizer, period_splitter, wordpiece_vocab  # noqa: E402

from open_provence_amd.config import EncoderDims  # noqa: E402
from open_provence_amd.synthetic import refinit_state_dict, synth_state_dict  # noqa: E402

REFERENCE_FILE = Path("/root/reference/open_provence/modeling_open_provence_standalone.py")


def load_reference(emit_specials: bool = True):
    if "nltk" not in sys.modules:
        nltk = types.ModuleType("nltk")
        nltk_tok = types.ModuleType("nltk.tokenize")

        class _Punkt:  # never used: every call passes an explicit splitter / pre-split sentences
            pass

        nltk_tok.PunktSentenceTokenizer = _Punkt
        nltk.tokenize = nltk_tok
        nltk.data = types.SimpleNamespace(load=lambda *_a, **_k: (_ for _ in ()).throw(LookupError("stub")))
        sys.modules["nltk"] = nltk
        sys.modules["nltk.tokenize"] = nltk_tok
    spec = importlib.util.spec_from_file_location("reference_standalone", REFERENCE_FILE)
    module = importlib.util.module_from_spec(spec)
    sys.modules["reference_standalone"] = module
    spec.loader.exec_module(module)
    module.AutoTokenizer.from_pretrained = staticmethod(lambda *_a, **_k: CharTokenizer(emit_specials=emit_specials))
    return module


def versions() -> dict[str, str]:
    import transformers

    return {
        "torch": torch.__version__,
        "transformers": transformers.__version__,
        "reference_pins_transformers": "4.57.1 (uv.lock:3640-3641)",
        "numpy": np.__version__,
    }


def build_model(ref, base_cfg: dict[str, Any], *, max_length: int, seed: int, weight_seed: int | None, weight_init: str = "synth"):
    torch.manual_seed(seed)
    cfg = ref.OpenProvenceConfig(
        base_model_config=dict(base_cfg),
        tokenizer_name_or_path="char-tokenizer",
        pruning_config={"hidden_size": base_cfg["hidden_size"]},
        max_length=max_length,
        num_labels=1,
    )
    model = ref.OpenProvenceModel(cfg)
    dims = EncoderDims.from_base_model_config(base_cfg, num_labels=1)
    if weight_seed is not None:
        state = (refinit_state_dict if weight_init == "refinit" else synth_state_dict)(dims, weight_seed)
        missing, unexpected = model.load_state_dict(state, strict=False)
        bad = [k for k in missing if "inv_freq" not in k]
        assert not bad and not unexpected, (bad, unexpected)
    model.eval()
    return model, dims


class emulate_transformers4_hidden_states:
    """The reference takes ``outputs.hidden_states[-1]`` (standalone.py:1695).  transformers >= 5 ties that entry to
    ``last_hidden_state`` (utils/output_capturing.py:269-277); the 4.x line the reference pins (uv.lock: 4.57.1)
    appended the last layer's output BEFORE ``final_norm`` (source not available offline; recalled).
    This context reproduces the 4.x tuple around the reference's unmodified forward: a pre-hook on ``final_norm``
    records its input and the backbone's output tuple gets that tensor as its last entry."""

    def __init__(self, model) -> None:
        self.model = model
        self.captured: list[torch.Tensor] = []

    def __enter__(self):
        backbone = self.model.ranking_model
        self._hook = backbone.model.final_norm.register_forward_pre_hook(lambda _m, args: self.captured.append(args[0]))
        self._orig = backbone.forward

        def patched(*args, **kwargs):
            self.captured.clear()
            out = self._orig(*args, **kwargs)
            if getattr(out, "hidden_states", None) is not None and self.captured:
                out.hidden_states = tuple(out.hidden_states[:-1]) + (self.captured[-1],)
            return out

        backbone.forward = patched
        return self

    def __exit__(self, *exc):
        self.model.ranking_model.forward = self._orig
        self._hook.remove()
        return False


def make_rows(dims: EncoderDims, lengths: list[int], seed: int) -> tuple[torch.Tensor, torch.Tensor]:
    from open_provence_amd.synthetic import pad_rows, synth_pair_batch

    rows = synth_pair_batch(dims, len(lengths), lengths, seed=seed, query_tokens=8 if min(lengths) < 40 else 24)
    return pad_rows(rows, pad_id=dims.pad_token_id or 0)


def base_cfg(**kw: Any) -> dict[str, Any]:
    cfg = {
        "model_type": "modernbert",
        "max_position_embeddings": 8192,
        "local_attention": 128,
        "global_attn_every_n_layers": 3,
        "global_rope_theta": 160000.0,
        "local_rope_theta": 10000.0,
        "pad_token_id": 0,
        "cls_token_id": 1,
        "sep_token_id": 2,
        "bos_token_id": 1,
        "eos_token_id": 2,
    }
    cfg.update(kw)
    return cfg


def forward_fixture(
    name: str,
    cfg: dict[str, Any],
    lengths: list[int],
    *,
    weight_seed: int | None,
    hidden_stride: int | None,
    init_seed: int = 0,
    input_seed: int = 1234,
    weight_init: str = "synth",
    transformers4_hidden_states: bool = False,
) -> None:
    ref = load_reference()
    model, dims = build_model(ref, cfg, max_length=max(lengths), seed=init_seed, weight_seed=weight_seed, weight_init=weight_init)
    ids, mask = make_rows(dims, lengths, input_seed)
    with torch.no_grad():
        if transformers4_hidden_states:
            with emulate_transformers4_hidden_states(model):
                out = model(input_ids=ids, attention_mask=mask)
        else:
            out = model(input_ids=ids, attention_mask=mask)
    arrays: dict[str, np.ndarray] = {
        "input_ids": ids.numpy(),
        "attention_mask": mask.numpy(),
        "ranking_logits": out.ranking_logits.float().numpy(),
        "pruning_logits": out.pruning_logits.float().numpy(),
    }
    hs = out.hidden_states
    if hidden_stride:
        for i, h in enumerate(hs):
            arrays[f"hidden_{i}"] = h[:, ::hidden_stride, :].float().numpy()
    if weight_seed is None:
        for k, v in model.state_dict().items():
            if "inv_freq" in k:
                continue
            arrays[f"w::{k}"] = v.float().numpy()
    meta = {
        "name": name,
        "base_model_config": cfg,
        "num_labels": 1,
        "lengths": lengths,
        "input_seed": input_seed,
        "hidden_stride": hidden_stride,
        "n_hidden_states": len(hs),
        "attn_implementation": getattr(model.ranking_model.config, "_attn_implementation", None),
        "layer_types": list(model.ranking_model.config.layer_types),
        "generator": "tests/golden/make_golden.py (reference OpenProvenceModel.forward, CPU fp32)",
        "versions": versions(),
    }
    if weight_seed is not None:
        meta["weight_seed"] = weight_seed
        meta["weight_init"] = weight_init
    if transformers4_hidden_states:
        meta["prune_pre_final_norm"] = True
        meta["note"] = ("hidden_states[-1] patched to the final_norm INPUT (emulate_transformers4_hidden_states): what "
                        "transformers 4.x ModernBertModel.forward returned, hence what the reference's pruning head "
                        "sees under its own uv.lock pin (4.57.1); the reference's forward code is otherwise untouched")
    np.savez_compressed(HERE / f"{name}.npz", **arrays)
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=2, sort_keys=True))
    print(f"[golden] {name}: rank={arrays['ranking_logits'].ravel()[:4]} prune-absmax={np.abs(arrays['pruning_logits']).max():.3f}")


PROCESS_CASES: list[dict[str, Any]] = [
    dict(
        case="str_single",
        question="what is a cat?",
        context="Cats are small animals. They purr a lot! Dogs bark. The sky is blue. Cats like fish.",
        kwargs=dict(threshold=0.5),
    ),
    dict(
        case="list_docs_reorder",
        question="how tall is the tower?",
        context=[
            "The tower is 300 m tall. It was built in 1889. Many people visit it.",
            "Bread is made from flour. Water is wet.",
            ["Pre split one. ", "Pre split two about the tower. ", "   ", "Pre split three."],
        ],
        kwargs=dict(threshold=0.45, reorder=True, top_k=2),
    ),
    dict(
        case="aligned_titles",
        question=["first question here?", "second question about rivers?"],
        context=[
            "Alpha sentence one. Alpha sentence two is longer than one. Alpha three.",
            "Rivers flow to the sea. Mountains are tall. Rivers carry water and silt.",
        ],
        kwargs=dict(threshold=0.5, title=["Alpha Title", "River Title"], always_select_title=True),
    ),
    dict(
        case="nested_explicit_titles",
        question=["q one?", "q two?"],
        context=[
            ["Doc a first. Doc a second. Doc a third.", ["Sent x. ", "Sent y. "]],
            ["Doc b only sentence here."],
        ],
        kwargs=dict(threshold=0.5, title=[["T-a", "T-x"], ["T-b"]], use_best_reranker_score=False),
    ),
    dict(
        case="long_document_multiblock",
        question="find the needle?",
        context=" ".join(f"Sentence number {i} talks about topic {i % 7} in some detail." for i in range(40)),
        kwargs=dict(threshold=0.5),
    ),
    dict(
        case="first_line_title",
        question="what about the header?",
        context="Header Line\nBody sentence one. Body sentence two. Body three is here.",
        kwargs=dict(threshold=0.5, first_line_as_title=True, always_select_title=True),
    ),
    dict(
        case="respect_boundaries_strip",
        question="boundaries?",
        context="  A short one.   Another sentence that is a bit longer than the first one.  End. ",
        kwargs=dict(threshold=0.5, respect_sentence_boundaries=True, strip_sentences=True),
    ),
    dict(
        case="zero_score_disabled",
        question="nothing relevant?",
        context="abc. def. ghi.",
        kwargs=dict(threshold=0.999, zero_score_when_empty=False),
    ),
]


def _jsonable(value: Any) -> Any:
    if isinstance(value, dict):
        return {k: _jsonable(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_jsonable(v) for v in value]
    if isinstance(value, (np.floating, np.integer)):
        return value.item()
    if isinstance(value, np.ndarray):
        return [_jsonable(v) for v in value.tolist()]
    return value


def process_fixture(name: str, cfg: dict[str, Any], *, weight_seed: int, max_length: int, emit_specials: bool, stub: bool) -> None:
    """G3: full process() through the reference.  ``stub=True`` replaces forward by a deterministic
    position-dependent logit pattern so host semantics are pinned independently of encoder numerics
    (same trick as the reference's tests/test_modeling_open_provence.py:940-949)."""

    ref = load_reference(emit_specials=emit_specials)
    model, _dims = build_model(ref, cfg, max_length=max_length, seed=0, weight_seed=weight_seed)
    if stub:

        def stub_forward(input_ids=None, attention_mask=None, **_kw):
            b, length = input_ids.shape
            pos = torch.arange(length, dtype=torch.float32)[None, :].expand(b, length)
            tok = input_ids.to(torch.float32)
            keep = torch.sin(0.37 * pos + 0.011 * tok) * 3.0
            prune = torch.stack([torch.zeros_like(keep), keep], dim=-1)
            rank = (input_ids.sum(dim=1, keepdim=True).to(torch.float32) % 17.0) / 4.0 - 2.0
            return {"ranking_logits": rank, "pruning_logits": prune}

        model.forward = stub_forward  # type: ignore[method-assign]
    cases_out = []
    for case in PROCESS_CASES:
        kwargs = dict(case["kwargs"])
        with torch.no_grad():
            result = model.process(
                question=case["question"],
                context=case["context"],
                sentence_splitter=period_splitter,
                show_progress=False,
                return_sentence_metrics=True,
                return_sentence_texts=True,
                batch_size=4,
                **kwargs,
            )
        result.pop("timing")
        result.pop("performance_trace")
        cases_out.append({"case": case["case"], "question": case["question"], "context": case["context"], "kwargs": kwargs, "expected": _jsonable(result)})
    meta = {
        "name": name,
        "base_model_config": cfg,
        "num_labels": 1,
        "weight_seed": weight_seed,
        "max_length": max_length,
        "emit_specials": emit_specials,
        "stub_forward": stub,
        "cases": cases_out,
        "generator": "tests/golden/make_golden.py (reference OpenProvenceModel.process, CPU fp32)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: {len(cases_out)} process() cases")


def eval_dataset_examples() -> list[dict[str, Any]]:
    """Small span-annotated dataset in the schema scripts/eval_datasets.py reads (query, texts, context_spans,
    context_spans_relevance), exercising both relevance encodings (0/1 mask and index list), missing spans, empty and
    out-of-range spans, a missing query and a context without relevance labels."""

    def spans_of(text: str) -> list[list[int]]:
        out, start = [], 0
        for i, ch in enumerate(text):
            if ch == ".":
                end = i + 1
                while end < len(text) and text[end] == " ":
                    end += 1
                out.append([start, end])
                start = end
        if start < len(text):
            out.append([start, len(text)])
        return out

    t1 = "The tower is tall. It was built long ago. Many people visit it. Bread is made from flour."
    t2 = "Rivers flow to the sea. Mountains are tall. Rivers carry water and silt."
    t3 = "A single unsplit passage about nothing in particular"
    t4 = "Alpha one. Alpha two is longer than one. Alpha three. Alpha four closes the passage."
    t5 = "Short. Tiny. Brief one here. And the last sentence of this passage is the longest of them all."
    return [
        {"query": "how tall is the tower?", "texts": [t1, t2], "context_spans": [spans_of(t1), spans_of(t2)],
         "context_spans_relevance": [[1, 1, 0, 0], [1]]},
        {"query": "what do rivers carry?", "texts": [t2, t3, t4], "context_spans": [spans_of(t2), [], spans_of(t4)],
         "context_spans_relevance": [[0, 2], [], [0, 0, 0, 1]]},
        {"query": None, "texts": [t1], "context_spans": [spans_of(t1)], "context_spans_relevance": [[1, 0, 0, 0]]},
        {"query": "which sentence is longest?", "texts": [t5, t1],
         "context_spans": [spans_of(t5) + [[200, 300], [10, 5]], spans_of(t1)[:2]],
         "context_spans_relevance": [[3, 9], None]},
        {"query": "alpha?", "texts": [t4], "context_spans": [spans_of(t4)]},
    ]


def eval_fixture(name: str, cfg: dict[str, Any], *, max_length: int) -> None:
    """G4: the reference's evaluation loop (scripts/eval_datasets.py:247-486 ``evaluate_dataset``) run on the dataset
    above with the stub forward of G3, two thresholds.  The script needs Python 3.11's ``datetime.UTC`` and the
    ``open_provence`` package path: both are shimmed here (alias + module registration), nothing else is touched."""

    import datetime as _dt

    if not hasattr(_dt, "UTC"):
        _dt.UTC = _dt.timezone.utc  # type: ignore[attr-defined]
    ref = load_reference(emit_specials=True)
    pkg = types.ModuleType("open_provence")
    pkg.__path__ = []  # type: ignore[attr-defined]
    sys.modules["open_provence"] = pkg
    sys.modules["open_provence.modeling_open_provence_standalone"] = ref
    spec = importlib.util.spec_from_file_location("reference_eval_datasets", "/root/reference/scripts/eval_datasets.py")
    script = importlib.util.module_from_spec(spec)
    sys.modules["reference_eval_datasets"] = script  # dataclasses resolve their module through sys.modules
    spec.loader.exec_module(script)

    model, _dims = build_model(ref, cfg, max_length=max_length, seed=0, weight_seed=5)

    def stub_forward(input_ids=None, attention_mask=None, **_kw):
        b, length = input_ids.shape
        pos = torch.arange(length, dtype=torch.float32)[None, :].expand(b, length)
        tok = input_ids.to(torch.float32)
        keep = torch.sin(0.37 * pos + 0.011 * tok) * 3.0
        prune = torch.stack([torch.zeros_like(keep), keep], dim=-1)
        rank = (input_ids.sum(dim=1, keepdim=True).to(torch.float32) % 17.0) / 4.0 - 2.0
        return {"ranking_logits": rank, "pruning_logits": prune}

    model.forward = stub_forward  # type: ignore[method-assign]
    dataset = eval_dataset_examples()
    runs = []
    for threshold in (0.5, 0.1):
        result = script.evaluate_dataset(model, dataset, threshold=threshold, batch_size=4, dataset_label="synthetic",
                                         show_progress=False, debug_messages=False, print_timing_summary=False, silent=True)
        result.pop("process_time_seconds")
        result.pop("timing")
        runs.append({"threshold": threshold, "expected": _jsonable(result)})
    meta = {
        "name": name,
        "base_model_config": cfg,
        "max_length": max_length,
        "dataset": dataset,
        "runs": runs,
        "generator": "tests/golden/make_golden.py (reference scripts/eval_datasets.py evaluate_dataset, stub forward)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: {[(r['threshold'], r['expected']['confusion_matrix']) for r in runs]}")


def mldr_rows() -> list[dict[str, Any]]:
    """MLDR-shaped rows (query_id, query, positive_passages / negative_passages of {docid, title, text}) with string,
    list, blank and missing titles, a row without passages and a single-query / single-doc variant."""

    doc_a = "The tower is 300 m tall. It was built in 1889. Many people visit it every year."
    doc_b = "Bread is made from flour. Water is wet. Salt is salty."
    doc_c = "Rivers flow to the sea. Mountains are tall. Rivers carry water and silt to the delta."
    doc_d = " ".join(f"Sentence number {i} talks about topic {i % 5} in some detail." for i in range(14))
    return [
        {"query_id": "q1", "query": "how tall is the tower?",
         "positive_passages": [{"docid": "d1", "title": "Tower", "text": doc_a}],
         "negative_passages": [{"docid": "d2", "title": ["Bread", " and ", None, "water"], "text": doc_b},
                               {"docid": "d3", "title": "   ", "text": doc_c}]},
        {"query_id": "q2", "query": "nothing here?", "positive_passages": [], "negative_passages": []},
        {"query_id": "q3", "query": "what do rivers carry?",
         "positive_passages": [{"docid": "d4", "text": doc_c}, {"docid": "d5", "title": None, "text": doc_d}],
         "negative_passages": [{"docid": "d6", "title": "Loaf", "text": doc_b}]},
    ]


def mldr_fixture(name: str, cfg: dict[str, Any], *, max_length: int) -> None:
    """G5: the reference's MLDR record builder (scripts/eval_mldr.py:238-524 ``build_records``) on the rows above with
    the stub forward, for the multi-query set and for a single query with a single passage (where process() un-nests
    its outputs).  Shims for the import only: ``datetime.UTC`` is not needed here, ``litellm`` (absent, used by the
    LLM-judge half of the script) is an empty stub module, ``open_provence`` resolves to the standalone file."""

    ref = load_reference(emit_specials=True)
    pkg = types.ModuleType("open_provence")
    pkg.__path__ = []  # type: ignore[attr-defined]
    sys.modules["open_provence"] = pkg
    sys.modules["open_provence.modeling_open_provence_standalone"] = ref
    sys.modules.setdefault("litellm", types.ModuleType("litellm"))
    spec = importlib.util.spec_from_file_location("reference_eval_mldr", "/root/reference/scripts/eval_mldr.py")
    script = importlib.util.module_from_spec(spec)
    sys.modules["reference_eval_mldr"] = script
    spec.loader.exec_module(script)

    model, _dims = build_model(ref, cfg, max_length=max_length, seed=0, weight_seed=5)

    def stub_forward(input_ids=None, attention_mask=None, **_kw):
        b, length = input_ids.shape
        pos = torch.arange(length, dtype=torch.float32)[None, :].expand(b, length)
        tok = input_ids.to(torch.float32)
        keep = torch.sin(0.37 * pos + 0.011 * tok) * 3.0
        prune = torch.stack([torch.zeros_like(keep), keep], dim=-1)
        rank = (input_ids.sum(dim=1, keepdim=True).to(torch.float32) % 17.0) / 4.0 - 2.0
        return {"ranking_logits": rank, "pruning_logits": prune}

    model.forward = stub_forward  # type: ignore[method-assign]

    def process_fn(**kwargs):  # the reference resolves its splitter by language; pin the test splitter instead
        return model.process(sentence_splitter=period_splitter, **kwargs)

    import inspect as _inspect

    process_fn.__signature__ = _inspect.signature(model.process)  # type: ignore[attr-defined]
    rows = mldr_rows()
    single = [{"query_id": "s1", "query": "how tall is the tower?",
               "positive_passages": [rows[0]["positive_passages"][0]], "negative_passages": []}]
    runs = []
    for label, data, best in (("multi", rows, True), ("multi_last_block_score", rows, False), ("single", single, True)):
        records, stats, n_queries = script.build_records(process_fn, data, threshold=0.4, batch_size=4, log_timing=False,
                                                         use_best_reranker_score=best, show_progress=False)
        runs.append({"label": label, "rows": data, "use_best_reranker_score": best,
                     "expected": _jsonable({"records": records, "stats": stats, "n_queries": n_queries})})
    meta = {
        "name": name,
        "base_model_config": cfg,
        "max_length": max_length,
        "threshold": 0.4,
        "runs": runs,
        "generator": "tests/golden/make_golden.py (reference scripts/eval_mldr.py build_records, stub forward)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: {[(r['label'], len(r['expected']['records'])) for r in runs]}")


def wordpiece_fixture(name: str, cfg: dict[str, Any], *, weight_seed: int, max_length: int, stub: bool) -> None:
    """G11: process() through the reference with a REAL Hugging Face fast tokenizer (WordPiece over a fixed vocabulary,
    tests/helpers.py:build_wordpiece_tokenizer), once with a tokenizer that emits [CLS]/[SEP] itself and once with one
    that does not -- the reference's manual special-token path (standalone.py:1501-1538, 2123-2135).  ``stub`` as G3."""

    variants = []
    for emit in (True, False):
        ref = load_reference(emit_specials=True)
        ref.AutoTokenizer.from_pretrained = staticmethod(lambda *_a, _emit=emit, **_k: build_wordpiece_tokenizer(_emit))
        model, _dims = build_model(ref, cfg, max_length=max_length, seed=0, weight_seed=weight_seed)
        assert bool(model._manual_special_tokens_required) == (not emit)
        if stub:

            def stub_forward(input_ids=None, attention_mask=None, **_kw):
                b, length = input_ids.shape
                pos = torch.arange(length, dtype=torch.float32)[None, :].expand(b, length)
                tok = input_ids.to(torch.float32)
                keep = torch.sin(0.37 * pos + 0.011 * tok) * 3.0
                prune = torch.stack([torch.zeros_like(keep), keep], dim=-1)
                rank = (input_ids.sum(dim=1, keepdim=True).to(torch.float32) % 17.0) / 4.0 - 2.0
                return {"ranking_logits": rank, "pruning_logits": prune}

            model.forward = stub_forward  # type: ignore[method-assign]
        cases_out = []
        for case in WORDPIECE_CASES:
            kwargs = dict(case["kwargs"])
            with torch.no_grad():
                result = model.process(question=case["question"], context=case["context"], sentence_splitter=period_splitter,
                                       show_progress=False, return_sentence_metrics=True, return_sentence_texts=True,
                                       batch_size=4, **kwargs)
            result.pop("timing")
            result.pop("performance_trace")
            if not stub:
                margin = _sentence_margin(_jsonable(result["sentence_probabilities"]), kwargs["threshold"])
                assert margin > 5e-3, (case["case"], emit, margin)
            cases_out.append({"case": case["case"], "kwargs": kwargs, "expected": _jsonable(result)})
        variants.append({"emit_specials": emit, "manual_special_tokens_required": not emit, "cases": cases_out})
    meta = {
        "name": name, "base_model_config": cfg, "num_labels": 1, "weight_seed": weight_seed, "weight_init": "synth",
        "max_length": max_length, "stub_forward": stub, "vocab_size_tokenizer": len(wordpiece_vocab()), "variants": variants,
        "generator": "tests/golden/make_golden.py (reference OpenProvenceModel.process with a PreTrainedTokenizerFast built "
                     "offline; 4.x build_inputs_with_special_tokens semantics restored by tests/helpers.py)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: {[(v['emit_specials'], len(v['cases'])) for v in variants]}")


def _load_script(ref, name: str):
    """Import one of the reference's scripts (scripts/eval_datasets.py, scripts/eval_mldr.py) with `open_provence`
    resolving to the standalone module, `litellm` (absent; LLM-judge half only) stubbed and `datetime.UTC` aliased."""

    import datetime as _dt

    if not hasattr(_dt, "UTC"):
        _dt.UTC = _dt.timezone.utc  # type: ignore[attr-defined]
    pkg = types.ModuleType("open_provence")
    pkg.__path__ = []  # type: ignore[attr-defined]
    sys.modules["open_provence"] = pkg
    sys.modules["open_provence.modeling_open_provence_standalone"] = ref
    sys.modules.setdefault("litellm", types.ModuleType("litellm"))
    spec = importlib.util.spec_from_file_location(f"reference_{name}", f"/root/reference/scripts/{name}.py")
    script = importlib.util.module_from_spec(spec)
    sys.modules[f"reference_{name}"] = script
    spec.loader.exec_module(script)
    return script


def _sentence_margin(value: Any, threshold: float) -> float:
    """Smallest |probability - threshold| over every sentence probability in a (nested) process() result."""

    if isinstance(value, (list, tuple)):
        return min((_sentence_margin(v, threshold) for v in value), default=1.0)
    if isinstance(value, float):
        return abs(value - threshold)
    return 1.0


def long_mldr_rows() -> list[dict[str, Any]]:
    """MLDR-shaped rows whose passages are LONG documents (3 - 5 k characters = 2 - 3 blocks at max_length 2048 with
    the one-token-per-character tokenizer), all with explicit titles as scripts/eval_mldr.py passes them (:388)."""

    def doc(seed: int, n: int) -> str:
        topics = ["rivers", "towers", "bread", "mountains", "harbours", "bridges", "orchards"]
        parts = []
        for i in range(n):
            t = topics[(seed + i * 3) % len(topics)]
            parts.append(f"Section {seed}.{i} explains how {t} shape the region and why item {i * 7 + seed} matters for {t}.")
        return " ".join(parts)

    return [
        {"query_id": "L1", "query": "how do rivers shape the region?",
         "positive_passages": [{"docid": "p1", "title": "Rivers of the north", "text": doc(1, 48)}],
         "negative_passages": [{"docid": "n1", "title": "Bread and ovens", "text": doc(2, 36)}]},
        {"query_id": "L2", "query": "which bridges matter most?",
         "positive_passages": [{"docid": "p2", "title": ["Bridges", " ", "and harbours"], "text": doc(3, 40)}],
         "negative_passages": []},
    ]


def mldr_model_fixture(name: str, cfg: dict[str, Any], *, max_length: int, weight_seed: int, threshold: float) -> None:
    """G9 (BASELINE.json configs[3]): the reference's MLDR path -- scripts/eval_mldr.py build_records (:238-524) over
    process() with explicit per-passage titles, multi-block documents at max_length 2048 -- with the REAL forward of a
    large-shaped model (H = 768, I = 3072, 12 heads, full depth; synthetic seed weights), both score conventions."""

    import inspect as _inspect

    from open_provence_amd.eval_harness import clean_title  # pinned against the reference's normalize_title by G5

    ref = load_reference(emit_specials=True)
    script = _load_script(ref, "eval_mldr")
    model, _dims = build_model(ref, cfg, max_length=max_length, seed=0, weight_seed=weight_seed)

    def process_fn(**kwargs):
        return model.process(sentence_splitter=period_splitter, **kwargs)

    process_fn.__signature__ = _inspect.signature(model.process)  # type: ignore[attr-defined]
    rows = long_mldr_rows()
    runs = []
    with torch.no_grad():
        # the same process() call the script makes, kept whole: sentence probabilities do not depend on the threshold,
        # so the fixture's threshold is the candidate FARTHEST from every sentence probability (a kept / removed
        # decision within the 1e-3 numerical bar of the threshold would make "identical kept sets" a coin toss)
        probe = model.process(question=[r["query"] for r in rows],
                              context=[[p["text"] for p in r["positive_passages"] + r["negative_passages"]] for r in rows],
                              title=[[clean_title(p.get("title")) for p in r["positive_passages"] + r["negative_passages"]] for r in rows],
                              sentence_splitter=period_splitter, threshold=threshold, batch_size=4, show_progress=False,
                              return_sentence_metrics=True, return_sentence_texts=False)
        probs = _jsonable(probe["sentence_probabilities"])
        threshold = max((round(threshold + 0.01 * k, 2) for k in range(-10, 11)), key=lambda t: _sentence_margin(probs, t))
        margin = _sentence_margin(probs, threshold)
        assert margin > 5e-3, f"no candidate threshold is farther than {margin:.2e} from every sentence probability"
        for label, best in (("best_block_score", True), ("last_block_score", False)):
            records, stats, n_queries = script.build_records(process_fn, rows, threshold=threshold, batch_size=4, log_timing=False,
                                                             use_best_reranker_score=best, show_progress=False)
            runs.append({"label": label, "use_best_reranker_score": best,
                         "expected": _jsonable({"records": records, "stats": stats, "n_queries": n_queries})})
    n_blocks = sum(len(str(p["text"])) // (max_length - 64) + 1 for r in rows for p in r["positive_passages"] + r["negative_passages"])
    meta = {
        "name": name,
        "base_model_config": cfg,
        "max_length": max_length,
        "weight_seed": weight_seed,
        "weight_init": "synth",
        "threshold": threshold,
        "rows": rows,
        "runs": runs,
        "min_margin_to_threshold": margin,
        "approx_blocks": n_blocks,
        "generator": "tests/golden/make_golden.py (reference scripts/eval_mldr.py build_records over the REAL forward, CPU fp32)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: {[(r['label'], len(r['expected']['records'])) for r in runs]} margin {margin:.3e}")


def eval_model_fixture(name: str, cfg: dict[str, Any], *, max_length: int, weight_seed: int) -> None:
    """G4m: the reference's dataset evaluator (scripts/eval_datasets.py:247-486) over the REAL forward (small model)."""

    ref = load_reference(emit_specials=True)
    script = _load_script(ref, "eval_datasets")
    model, _dims = build_model(ref, cfg, max_length=max_length, seed=0, weight_seed=weight_seed)
    dataset = eval_dataset_examples()
    runs = []
    with torch.no_grad():
        for threshold in (0.5, 0.3):
            result = script.evaluate_dataset(model, dataset, threshold=threshold, batch_size=4, dataset_label="synthetic",
                                             show_progress=False, debug_messages=False, print_timing_summary=False, silent=True)
            result.pop("process_time_seconds")
            result.pop("timing")
            margin = min((abs(float(sc) - threshold) for sc in result["roc_data"]["scores"]), default=1.0)
            assert margin > 5e-3, (threshold, margin)
            runs.append({"threshold": threshold, "expected": _jsonable(result), "min_margin_to_threshold": margin})
    meta = {
        "name": name, "base_model_config": cfg, "max_length": max_length, "weight_seed": weight_seed, "weight_init": "synth",
        "dataset": dataset, "runs": runs,
        "generator": "tests/golden/make_golden.py (reference scripts/eval_datasets.py evaluate_dataset over the REAL forward)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: {[(r['threshold'], r['expected']['confusion_matrix']) for r in runs]}")


def raw_predictions_fixture(name: str, cfg: dict[str, Any], *, max_length: int, weight_seed: int) -> None:
    """G10: the single-block API with the REAL forward -- get_raw_predictions_batch (shared and per-sample queries),
    get_raw_predictions and predict_with_thresholds (mean and majority rule) (standalone.py:1742-1881)."""

    ref = load_reference(emit_specials=True)
    model, _dims = build_model(ref, cfg, max_length=max_length, seed=0, weight_seed=weight_seed)
    contexts_batch = [
        ["Cats purr a lot. ", "Dogs bark loudly! ", "The sky is blue."],
        ["A single context sentence that is somewhat longer than the others in this batch."],
        ["Short. ", "Tiny. ", "Brief one here. ", "And the last."],
    ]
    queries = ["what do cats do?", "single?", "which is brief?"]
    thresholds = [0.2, 0.5, 0.8]

    def dump(pred) -> dict[str, Any]:
        return {"ranking_score": float(pred.ranking_score), "pruning_probs": [float(v) for v in np.asarray(pred.pruning_probs)],
                "context_ranges": [[int(a), int(b)] for a, b in pred.context_ranges]}

    with torch.no_grad():
        shared = [dump(p) for p in model.get_raw_predictions_batch(queries[0], contexts_batch)]
        per_sample = [dump(p) for p in model.get_raw_predictions_batch(queries, contexts_batch, batch_size=2)]
        single = dump(model.get_raw_predictions(queries[2], contexts_batch[2]))
        thr_mean = model.predict_with_thresholds(queries[0], contexts_batch[0], thresholds)
        thr_major = model.predict_with_thresholds(queries[2], contexts_batch[2], thresholds, use_majority=True)
    meta = {
        "name": name, "base_model_config": cfg, "max_length": max_length, "weight_seed": weight_seed, "weight_init": "synth",
        "queries": queries, "contexts_batch": contexts_batch, "thresholds": thresholds,
        "expected": _jsonable({"shared_query": shared, "per_sample_queries": per_sample, "single": single,
                               "predict_with_thresholds_mean": {k: (v if not isinstance(v, dict) else {str(t): x for t, x in v.items()}) for k, v in thr_mean.items()},
                               "predict_with_thresholds_majority": {k: (v if not isinstance(v, dict) else {str(t): x for t, x in v.items()}) for k, v in thr_major.items()}}),
        "generator": "tests/golden/make_golden.py (reference get_raw_predictions(_batch) / predict_with_thresholds, REAL forward)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: scores {[round(p['ranking_score'], 4) for p in shared]}")


class RegexPunkt:
    """Deterministic stand-in for nltk's Punkt model (absent here): spans of text up to and including a run of
    ``.!?`` that is followed by whitespace or the end, trailing whitespace excluded -- the contract
    ``span_tokenize`` has.  Used on BOTH sides (reference and this repo) so the fixture pins everything around the
    model: block iteration, bullet grouping, whitespace stretching, over-long clipping."""

    def span_tokenize(self, text: str):
        import re as _re

        start = None
        for m in _re.finditer(r"\S+", text):
            if start is None:
                start = m.start()
            if _re.search(r"[.!?]+[\"')\]]*$", m.group(0)):
                yield (start, m.end())
                start = None
        if start is not None:
            yield (start, len(text.rstrip()))


def splitter_texts() -> list[str]:
    long_sentence = "This sentence goes on and on " * 40 + "until it finally stops."
    return [
        "Cats purr. Dogs bark!  Birds sing?\nA new line starts here. And ends.",
        "Intro paragraph before the list:\n- first bullet item. It has two sentences.\n- second bullet\n* star bullet here.\nTrailing text after list.",
        "1. Numbered heading\nBody under heading one. More body.\n2) Second heading\n\nBody two.",
        "   Leading spaces and trailing newline.\n\n",
        "No terminal punctuation at all",
        "",
        "   \n  ",
        long_sentence,
        "Short one. " + long_sentence + " Tail sentence.",
        "Semi; colon: and a very long clause " + "word " * 120 + "; then the end.",
        "Line one without period\nLine two without period\n\u2022 unicode bullet item one.\n\u2022 unicode bullet two",
        "Windows line endings.\r\nSecond line here.\r\n- bullet after CRLF.",
    ]


def splitter_fixture(name: str) -> None:
    """G6: the reference's English splitter (standalone.py:1032-1126, helpers :481-612) and the auto splitter's language
    routing (:1129-1143) with the Punkt model replaced by RegexPunkt on both sides."""

    ref = load_reference(emit_specials=True)
    ref._ENGLISH_SENTENCE_TOKENIZER = RegexPunkt()
    runs = []
    for max_chars in (ref.DEFAULT_ENGLISH_SENTENCE_MAX_CHARS, 120, 40):
        split = ref.create_english_sentence_splitter(max_chars)
        runs.append({"max_chars": max_chars, "outputs": [split(t) for t in splitter_texts()]})
    auto = ref.create_auto_sentence_splitter(japanese_splitter=ref.simple_sentence_splitter,
                                             english_splitter=ref.create_english_sentence_splitter())
    auto_texts = ["寿司が好きです。ラーメンも好きです。", "Sushi is tasty. Ramen too.", "漢字だけ。Mixed kana が here.", ""]
    meta = {
        "name": name,
        "texts": splitter_texts(),
        "runs": runs,
        "auto_texts": auto_texts,
        "auto_outputs": [auto(t) for t in auto_texts],
        "default_max_chars": ref.DEFAULT_ENGLISH_SENTENCE_MAX_CHARS,
        "generator": "tests/golden/make_golden.py (reference create_english_sentence_splitter / create_auto_sentence_splitter, RegexPunkt stand-in)",
        "versions": versions(),
    }
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1, sort_keys=True, ensure_ascii=False))
    print(f"[golden] {name}: {[len(o) for o in runs[0]['outputs']]}")


def check_writer(report_name: str = "check_writer_report") -> None:
    """Row f4: a checkpoint written by ``open_provence_amd``'s ``save_pretrained`` is read back by the REFERENCE --
    its OpenProvenceConfig parses config.json, its OpenProvenceModel takes model.safetensors with strict key matching,
    and its forward on those weights equals the oracle.  Also records what the reference's own ``from_pretrained`` does
    with the directory under this image's transformers (it fails: an incompatibility of the reference with
    transformers >= 5, not of the checkpoint).  Writes tests/golden/<report_name>.json."""

    import tempfile

    from safetensors.torch import load_file

    from open_provence_amd import modeling
    from open_provence_amd.config import OpenProvenceConfig as NativeConfig
    from oracle.modernbert_oracle import oracle_forward

    cfg = base_cfg(vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2, local_attention=32)
    dims = EncoderDims.from_base_model_config(cfg, num_labels=1)
    state = synth_state_dict(dims, 41)
    native = NativeConfig(base_model_config=cfg, tokenizer_name_or_path="char-tokenizer", pruning_config={"hidden_size": 128},
                          max_length=96, default_threadshold=0.2)
    writer = modeling.OpenProvenceModel.__new__(modeling.OpenProvenceModel)  # writer only: no GPU in this container
    writer.config, writer.max_length, writer.num_labels, writer.dims = native, 96, 1, dims
    writer.tokenizer, writer._weights, writer.pruning_hidden_state = CharTokenizer(), dict(state), "post_final_norm"
    report: dict[str, Any] = {"generator": "tests/golden/make_golden.py --check-writer", "versions": versions()}
    with tempfile.TemporaryDirectory() as tmp:
        writer.save_pretrained(tmp)
        ref = load_reference(emit_specials=True)
        ref_cfg = ref.OpenProvenceConfig.from_pretrained(tmp)
        report["reference_config_reads"] = {"max_length": ref_cfg.max_length, "num_labels": ref_cfg.num_labels,
                                            "hidden_size": ref_cfg.base_model_config["hidden_size"],
                                            "default_threadshold": getattr(ref_cfg, "default_threadshold", None)}
        torch.manual_seed(0)
        model = ref.OpenProvenceModel(ref_cfg)
        saved = load_file(str(Path(tmp) / "model.safetensors"))
        missing, unexpected = model.load_state_dict(saved, strict=False)
        report["strict_load"] = {"missing": [k for k in missing if "inv_freq" not in k], "unexpected": list(unexpected)}
        model.eval()
        ids, mask = make_rows(dims, [96, 40, 77], 1234)
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=mask)
        orc = oracle_forward(state, dims, ids, mask)
        m = mask.bool()
        report["reference_on_written_checkpoint_vs_oracle"] = {
            "max_abs_prune": float((out.pruning_logits - orc.pruning_logits)[m].abs().max()),
            "max_abs_rank": float((out.ranking_logits - orc.ranking_logits).abs().max()),
        }
        try:
            ref.OpenProvenceModel.from_pretrained(tmp, device="cpu")
            report["reference_from_pretrained"] = "ok"
        except Exception as exc:  # noqa: BLE001
            report["reference_from_pretrained"] = f"{type(exc).__name__}: {str(exc)[:300]}"
    (HERE / f"{report_name}.json").write_text(json.dumps(report, indent=1, sort_keys=True))
    print(json.dumps(report, indent=1))
    assert not report["strict_load"]["missing"] and not report["strict_load"]["unexpected"]
    assert report["reference_on_written_checkpoint_vs_oracle"]["max_abs_prune"] < 5e-5


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--only", nargs="*", default=None)
    parser.add_argument("--check-writer", action="store_true")
    args = parser.parse_args()
    if args.check_writer:
        check_writer()
        return

    def want(name: str) -> bool:
        return args.only is None or name in args.only

    # G0: the survey's probe config (head_dim 16) -- oracle-only (the HIP kernels need head_dim 64).
    if want("g0_tiny_hd16"):
        forward_fixture(
            "g0_tiny_hd16",
            base_cfg(vocab_size=512, hidden_size=64, intermediate_size=96, num_hidden_layers=4, num_attention_heads=4, local_attention=16),
            [128, 100, 128, 77],
            weight_seed=None,
            hidden_stride=5,
        )
    # G0b: head_dim 64, the reference's OWN random init (weights stored), window 16 so that the
    # sliding mask bites at L=128.
    if want("g0b_hd64_refinit"):
        forward_fixture(
            "g0b_hd64_refinit",
            base_cfg(vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=4, num_attention_heads=2, local_attention=16),
            [128, 100, 128, 33],
            weight_seed=None,
            hidden_stride=5,
        )
    # G0c: same shape, synthetic O(1) weights (seed only).
    if want("g0c_hd64_synth"):
        forward_fixture(
            "g0c_hd64_synth",
            base_cfg(vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=4, num_attention_heads=2, local_attention=16),
            [128, 100, 128, 33, 17, 64],
            weight_seed=11,
            hidden_stride=5,
        )
    # G1: xsmall shape (config C2 dims), ragged rows, synthetic weights.
    if want("g1_xsmall"):
        forward_fixture(
            "g1_xsmall",
            base_cfg(vocab_size=102400, hidden_size=256, intermediate_size=1024, num_hidden_layers=10, num_attention_heads=4),
            [512, 512, 400, 512, 129, 512, 64, 333],
            weight_seed=21,
            hidden_stride=61,
        )
    # G1m: mean pooling + 2 labels variant on a short stack.
    if want("g1m_meanpool"):
        forward_fixture(
            "g1m_meanpool",
            base_cfg(vocab_size=4096, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4, classifier_pooling="mean"),
            [200, 256, 31],
            weight_seed=23,
            hidden_stride=None,
        )
    # G2: en-gte width (H=768, I=1152, 12 heads) truncated to 3 layers, mixed lengths up to 2048.
    if want("g2_gte_varlen"):
        forward_fixture(
            "g2_gte_varlen",
            base_cfg(vocab_size=50368, hidden_size=768, intermediate_size=1152, num_hidden_layers=3, num_attention_heads=12),
            [2048, 128, 1024, 384, 1536, 256],
            weight_seed=31,
            hidden_stride=None,
        )
    # G7 / G8: xsmall and base at FULL depth with weights drawn from the reference's own initialisation
    # distributions (regenerated from the seed: open_provence_amd.synthetic.refinit_state_dict) -- the regime a freshly
    # constructed reference model is in, beside the O(1) regime of G1 / G2.
    if want("g7_xsmall_refinit"):
        forward_fixture(
            "g7_xsmall_refinit",
            base_cfg(vocab_size=2048, hidden_size=256, intermediate_size=1024, num_hidden_layers=10, num_attention_heads=4),
            [512, 300, 512, 129, 64, 411],
            weight_seed=51, weight_init="refinit", hidden_stride=61,
        )
    if want("g8_base_refinit"):
        forward_fixture(
            "g8_base_refinit",
            base_cfg(vocab_size=2048, hidden_size=512, intermediate_size=2048, num_hidden_layers=19, num_attention_heads=8),
            [512, 200, 384, 77],
            weight_seed=52, weight_init="refinit", hidden_stride=None,
        )
    # G12: the pruning head on the PRE-final_norm state (transformers 4.x hidden_states[-1]); xsmall shape, 4 layers.
    if want("g12_prenorm_tf4"):
        forward_fixture(
            "g12_prenorm_tf4",
            base_cfg(vocab_size=2048, hidden_size=256, intermediate_size=1024, num_hidden_layers=4, num_attention_heads=4),
            [256, 100, 192, 33],
            weight_seed=61, hidden_stride=None, transformers4_hidden_states=True,
        )
    g3_cfg = base_cfg(vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=4, num_attention_heads=2, local_attention=32)
    if want("g3_process_stub"):
        process_fixture("g3_process_stub", g3_cfg, weight_seed=41, max_length=96, emit_specials=True, stub=True)
    if want("g3_process_stub_manual_specials"):
        process_fixture("g3_process_stub_manual_specials", g3_cfg, weight_seed=41, max_length=96, emit_specials=False, stub=True)
    if want("g3_process_model"):
        process_fixture("g3_process_model", g3_cfg, weight_seed=41, max_length=96, emit_specials=True, stub=False)
    if want("g4_eval_dataset"):
        eval_fixture("g4_eval_dataset", g3_cfg, max_length=96)
    if want("g5_mldr_records"):
        mldr_fixture("g5_mldr_records", g3_cfg, max_length=96)
    if want("g6_english_splitter"):
        splitter_fixture("g6_english_splitter")
    wp_cfg = base_cfg(vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=4, num_attention_heads=2, local_attention=32)
    if want("g11_process_wordpiece_stub"):
        wordpiece_fixture("g11_process_wordpiece_stub", wp_cfg, weight_seed=43, max_length=64, stub=True)
    if want("g11_process_wordpiece_model"):
        wordpiece_fixture("g11_process_wordpiece_model", wp_cfg, weight_seed=43, max_length=64, stub=False)
    if want("g4m_eval_dataset_model"):
        eval_model_fixture("g4m_eval_dataset_model", g3_cfg, max_length=96, weight_seed=41)
    if want("g10_raw_predictions"):
        raw_predictions_fixture("g10_raw_predictions", g3_cfg, max_length=96, weight_seed=41)
    # G9 = BASELINE.json configs[3]: large dims at FULL depth, max_length 2048, the MLDR caller
    if want("g9_large_mldr"):
        mldr_model_fixture(
            "g9_large_mldr",
            base_cfg(vocab_size=256, hidden_size=768, intermediate_size=3072, num_hidden_layers=25, num_attention_heads=12),
            max_length=2048, weight_seed=71, threshold=0.4,
        )


if __name__ == "__main__":
    main()
