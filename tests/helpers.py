"""Shared test helpers (test infrastructure, not product code)."""

from __future__ import annotations

import json
import sys
from pathlib import Path
from typing import Any

import numpy as np
import torch

REPO_ROOT = Path(__file__).resolve().parents[1]
GOLDEN_DIR = REPO_ROOT / "tests" / "golden"
if str(REPO_ROOT) not in sys.path:
    sys.path.insert(0, str(REPO_ROOT))


class CharTokenizer:
    """One token per character (id = code point); implements exactly the tokenizer surface the
    ``process()`` pipeline touches (reference call sites: standalone.py:664-672, 862-866, 2114-2152,
    2462, 3494).  ``emit_specials=False`` mimics tokenizers such as gte-ModernBERT whose
    ``build_inputs_with_special_tokens`` drops CLS/SEP, which forces the manual special-token path
    (standalone.py:1501-1538, 2123-2135)."""

    sep_token = "|"
    pad_token_id = 0
    cls_token_id = 1
    sep_token_id = 2
    model_max_length = 512

    def __init__(self, emit_specials: bool = True) -> None:
        self.emit_specials = emit_specials
        self.special_tokens_map: dict[str, Any] = {}

    def _ids(self, text: str) -> list[int]:
        try:  # one C-level pass for Latin-1 text (ids = code points either way)
            return list(text.encode("latin-1"))
        except UnicodeEncodeError:
            return [ord(ch) for ch in text]

    def __call__(
        self,
        text,
        add_special_tokens: bool = True,
        return_attention_mask: bool = True,
        padding: bool = False,
        truncation: bool = False,
        max_length: int | None = None,
        return_tensors: str | None = None,
        **_: Any,
    ):
        single = isinstance(text, str)
        items = [text] if single else list(text)
        rows = [self._ids(item) for item in items]
        if add_special_tokens:
            rows = [[self.cls_token_id, *row, self.sep_token_id] for row in rows]
        if truncation and max_length is not None:
            rows = [row[:max_length] for row in rows]
        masks = [[1] * len(row) for row in rows]
        if padding and rows:
            width = max(len(row) for row in rows)
            masks = [m + [0] * (width - len(m)) for m in masks]
            rows = [row + [self.pad_token_id] * (width - len(row)) for row in rows]
        out: dict[str, Any] = {"input_ids": rows}
        if return_attention_mask:
            out["attention_mask"] = masks
        if return_tensors == "pt":
            out = {k: torch.tensor(v, dtype=torch.long) for k, v in out.items()}
        return out

    def encode(self, text: str, add_special_tokens: bool = False) -> list[int]:
        ids = self._ids(text)
        return [self.cls_token_id, *ids, self.sep_token_id] if add_special_tokens else ids

    _SPECIALS_LATIN1 = bytes([pad_token_id, cls_token_id, sep_token_id])

    def decode(self, tokens, skip_special_tokens: bool = True, clean_up_tokenization_spaces: bool = False) -> str:
        try:  # Latin-1 ids: one C-level pass (bytes(...) raises for ids >= 256)
            raw = bytes(tokens)
            if skip_special_tokens:
                raw = raw.translate(None, self._SPECIALS_LATIN1)
            return raw.decode("latin-1")
        except (ValueError, TypeError):
            specials = {self.pad_token_id, self.cls_token_id, self.sep_token_id} if skip_special_tokens else set()
            return "".join(chr(int(t)) for t in tokens if int(t) not in specials)

    def batch_decode(self, batch, skip_special_tokens: bool = True, clean_up_tokenization_spaces: bool = False):
        return [self.decode(tokens, skip_special_tokens=skip_special_tokens) for tokens in batch]

    def build_inputs_with_special_tokens(self, tokens_a, tokens_b=None):
        tokens_b = list(tokens_b or [])
        if not self.emit_specials:
            return [*tokens_a, *tokens_b]
        if tokens_b:
            return [self.cls_token_id, *tokens_a, self.sep_token_id, *tokens_b, self.sep_token_id]
        return [self.cls_token_id, *tokens_a, self.sep_token_id]

    def create_token_type_ids_from_sequences(self, tokens_a, tokens_b=None):
        tokens_b = list(tokens_b or [])
        if not self.emit_specials:
            return [0] * len(tokens_a) + [1] * len(tokens_b)
        if tokens_b:
            return [0] * (len(tokens_a) + 2) + [1] * (len(tokens_b) + 1)
        return [0] * (len(tokens_a) + 2)


def period_splitter(text: str) -> list[str]:
    """Sentence = run of characters up to and including '.', '!' or '?' plus trailing spaces/newlines."""

    out: list[str] = []
    start = 0
    i = 0
    n = len(text)
    while i < n:
        if text[i] in ".!?":
            j = i + 1
            while j < n and text[j] in " \n":
                j += 1
            out.append(text[start:j])
            start = j
            i = j
        else:
            i += 1
    if start < n:
        out.append(text[start:])
    return out


def load_golden(name: str) -> tuple[dict[str, np.ndarray], dict[str, Any]]:
    """Return (arrays, meta) of fixture ``tests/golden/<name>.npz`` + ``<name>.json``."""

    arrays = dict(np.load(GOLDEN_DIR / f"{name}.npz", allow_pickle=False))
    with open(GOLDEN_DIR / f"{name}.json", "r", encoding="utf-8") as handle:
        meta = json.load(handle)
    return arrays, meta


def dims_from_meta(meta: dict[str, Any]):
    from open_provence_amd.config import EncoderDims

    return EncoderDims.from_base_model_config(meta["base_model_config"], num_labels=meta.get("num_labels", 1))


def state_from_fixture(arrays: dict[str, np.ndarray], meta: dict[str, Any]) -> dict[str, torch.Tensor]:
    """Stored weights (keys prefixed ``w::``) or regenerated from ``meta['weight_seed']``."""

    if "weight_seed" in meta:
        from open_provence_amd.synthetic import refinit_state_dict, synth_state_dict

        make = refinit_state_dict if meta.get("weight_init") == "refinit" else synth_state_dict
        return make(dims_from_meta(meta), int(meta["weight_seed"]))
    return {k[3:]: torch.from_numpy(v.copy()) for k, v in arrays.items() if k.startswith("w::")}


def rows_from_fixture(arrays: dict[str, np.ndarray]) -> list[list[int]]:
    ids = arrays["input_ids"]
    mask = arrays["attention_mask"]
    return [ids[i, : int(mask[i].sum())].tolist() for i in range(ids.shape[0])]


def host_only_model(tokenizer=None, max_length: int = 96, forward=None, cls=None):
    """An ``OpenProvenceModel`` built WITHOUT touching a GPU (the reference's tests use the same
    ``__new__`` + hand-set attributes idiom: tests/test_modeling_open_provence.py:214,349,...) whose
    ``forward`` is a caller-supplied stub -- used to pin host semantics on the CPU box."""

    from open_provence_amd import modeling

    klass = cls or modeling.OpenProvenceModel
    model = klass.__new__(klass)
    model.tokenizer = tokenizer or CharTokenizer()
    model.max_length = max_length
    model._runtime_device = torch.device("cpu")
    model.default_threshold = 0.1
    model.default_splitter_language = "auto"
    model._manual_special_tokens_required = False
    model._manual_cls_token_id = None
    model._manual_sep_token_id = None
    model._update_runtime_defaults()
    if forward is not None:
        model.forward = forward
    return model


def golden_stub_forward(input_ids=None, attention_mask=None, **_kw):
    """The deterministic forward replacement used to generate g3_process_stub*.json
    (tests/golden/make_golden.py: stub_forward)."""

    b, length = input_ids.shape
    pos = torch.arange(length, dtype=torch.float32)[None, :].expand(b, length)
    tok = input_ids.to(torch.float32)
    keep = torch.sin(0.37 * pos + 0.011 * tok) * 3.0
    prune = torch.stack([torch.zeros_like(keep), keep], dim=-1)
    rank = (input_ids.sum(dim=1, keepdim=True).to(torch.float32) % 17.0) / 4.0 - 2.0
    return {"ranking_logits": rank, "pruning_logits": prune}


def assert_process_result_matches(result, expected, *, prob_tol: float, score_tol: float):
    """Structural equality for text/indices, tolerance for the floating-point fields."""

    import math

    def walk(got, exp, path, tol):
        if isinstance(exp, float) or isinstance(got, float):
            if exp is None or got is None:
                assert got == exp, path
            else:
                assert math.isclose(float(got), float(exp), rel_tol=0.0, abs_tol=tol), (path, got, exp)
        elif isinstance(exp, (list, tuple)):
            assert isinstance(got, (list, tuple)) and len(got) == len(exp), (path, got, exp)
            for i, (g, e) in enumerate(zip(got, exp)):
                walk(g, e, f"{path}[{i}]", tol)
        else:
            assert got == exp, (path, got, exp)

    for key in ("pruned_context", "title", "kept_sentences", "removed_sentences"):
        assert key in result, key
        walk(result[key], expected[key], key, 0.0)
    walk(result["compression_rate"], expected["compression_rate"], "compression_rate", 1e-9)
    walk(result["sentence_probabilities"], expected["sentence_probabilities"], "sentence_probabilities", prob_tol)
    walk(result["reranking_score"], expected["reranking_score"], "reranking_score", score_tol)


# ---------------------------------------------------------------------------------------------------------------
# A REAL Hugging Face fast tokenizer, built offline: WordPiece over a fixed vocabulary (no training, so the same
# object is rebuilt bit for bit wherever this file runs).  Two variants, as the reference distinguishes them
# (standalone.py:1501-1538): ``emit_specials=True`` = a BertTokenizerFast-like tokenizer whose
# build_inputs_with_special_tokens adds [CLS] / [SEP]; ``False`` = the generic PreTrainedTokenizerFast of transformers
# 4.x (e.g. gte-ModernBERT's), whose build_inputs_with_special_tokens is the bare concatenation and forces the
# reference's manual special-token path.  transformers >= 5 dropped both methods from the fast-tokenizer class; the
# subclass below restores the 4.x behaviour the reference was written against (its uv.lock pins 4.57.1).
# ---------------------------------------------------------------------------------------------------------------
_WP_WORDS = ("the tower is tall river rivers flow to sea bread made from flour it was built long ago many people visit what "
             "how do carry water and silt a cat cats purr dogs bark mountains are which in of for on with that this there "
             "high old new small large city north south bridge harbour boats fish salt stone wood king year years").split()
_WP_PIECES = ["##" + c for c in "setanroildmpchgbfkwyvxzqju"] + ["##ed", "##ing", "##er", "##ly", "##es"]
_WP_CHARS = list("abcdefghijklmnopqrstuvwxyz0123456789.,!?;:'\"-()")


def wordpiece_vocab() -> dict[str, int]:
    vocab: dict[str, int] = {}
    for tok in ["[PAD]", "[CLS]", "[SEP]", "[UNK]", "[MASK]"] + _WP_CHARS + _WP_PIECES + _WP_WORDS:
        vocab.setdefault(tok, len(vocab))
    return vocab


def build_wordpiece_tokenizer(emit_specials: bool, legacy_methods: bool = True):
    """``legacy_methods=False`` returns the plain transformers >= 5 object (no build_inputs_with_special_tokens)."""

    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast

    tok = Tokenizer(models.WordPiece(vocab=wordpiece_vocab(), unk_token="[UNK]", continuing_subword_prefix="##"))
    tok.normalizer = normalizers.Lowercase()
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.decoder = decoders.WordPiece(prefix="##", cleanup=False)
    if emit_specials:
        tok.post_processor = processors.TemplateProcessing(
            single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1", special_tokens=[("[CLS]", 1), ("[SEP]", 2)]
        )

    class WordPieceFast4x(PreTrainedTokenizerFast):
        def build_inputs_with_special_tokens(self, token_ids_0, token_ids_1=None):
            second = list(token_ids_1 or [])
            if not emit_specials:  # transformers 4.x PreTrainedTokenizerBase default
                return list(token_ids_0) + second
            out = [self.cls_token_id, *token_ids_0, self.sep_token_id]  # 4.x BertTokenizerFast
            return out + second + [self.sep_token_id] if second else out

        def create_token_type_ids_from_sequences(self, token_ids_0, token_ids_1=None):
            second = list(token_ids_1 or [])
            if not emit_specials:
                return [0] * len(token_ids_0) + [1] * len(second)
            first = [0] * (len(token_ids_0) + 2)
            return first + [1] * (len(second) + 1) if second else first

    cls = WordPieceFast4x if legacy_methods else PreTrainedTokenizerFast
    return cls(tokenizer_object=tok, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]",
               mask_token="[MASK]", model_max_length=512)


def wordpiece_tokenizer_for_workers():
    """Module-level factory (HostFrontEnd's replicas import it by name)."""

    return build_wordpiece_tokenizer(True)


WORDPIECE_CASES = [
    dict(case="wp_single", question="How tall is the tower?",
         context="The tower is tall. It was built long ago! Many people visit it. Bread is made from flour.",
         kwargs=dict(threshold=0.5)),
    dict(case="wp_docs_titles", question=["what do rivers carry?", "which city is old?"],
         context=[["Rivers flow to the sea. Mountains are tall. Rivers carry water and silt.",
                   "The old bridge is made of stone. Boats carry fish and salt."],
                  ["There is a small city in the north. The king built a harbour. Many years ago it was new."]],
         kwargs=dict(threshold=0.45, title=[["Rivers", "The bridge"], ["North city"]])),
    dict(case="wp_long_multiblock", question="Which boats carry salt?",
         context=" ".join(f"In year {i} the boats carry fish and salt to the harbour of the north city." for i in range(30)),
         kwargs=dict(threshold=0.5)),
]


def frontend_stub_model():
    """Model factory of tests/test_frontend.py (module-level: the front-end's worker processes unpickle it by name)."""

    return host_only_model(tokenizer=CharTokenizer(), max_length=96, forward=golden_stub_forward)
