"""Deterministic synthetic weights and inputs (no network: there are no checkpoints or datasets here).

* :func:`synth_state_dict` fills a checkpoint-shaped state dict (key names = the reference's
  ``model.safetensors`` layout, SURVEY.md section 8b / ``standalone.py:1452-1464``) from a counter-based
  integer hash, so that the golden-vector generator (which runs the real reference in the build
  container) and the GPU box regenerate bit-identical fp32 weights from ``(dims, seed)`` alone.  The
  per-tensor scales are chosen so that activations, attention scores and logits are O(1): a far more
  demanding numerical test than the 0.02-std Hugging Face initialisation, where every residual
  branch is tiny.
* :func:`synth_pair_batch` builds the BASELINE.json workload: one query shared by N contexts,
  ``[CLS] + 24 query ids + [SEP] + (L-27) context ids + [SEP]`` per row.
"""

from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch

from .config import EncoderDims

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _tensor_stream(seed: int, tensor_tag: int, count: int) -> np.ndarray:
    """``count`` floats uniform in [-1, 1), a pure function of (seed, tensor_tag, index)."""

    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x1000003) + np.uint64(tensor_tag)], dtype=np.uint64))[0]
        idx = np.arange(count, dtype=np.uint64) + base
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits -> exactly representable in fp32
    u = bits.astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u * np.float32(2.0) - np.float32(1.0)


def _filled(seed: int, tag: int, shape: Sequence[int], scale: float, offset: float = 0.0) -> torch.Tensor:
    count = int(np.prod(shape))
    values = _tensor_stream(seed, tag, count) * np.float32(scale) + np.float32(offset)
    return torch.from_numpy(values.reshape(tuple(shape)).astype(np.float32, copy=False)).clone()


def state_dict_keys(dims: EncoderDims) -> list[tuple[str, tuple[int, ...]]]:
    """Checkpoint tensor names and shapes, in the order the loader expects them."""

    H, I, V, nl = dims.hidden_size, dims.intermediate_size, dims.vocab_size, dims.num_labels
    keys: list[tuple[str, tuple[int, ...]]] = [
        ("ranking_model.model.embeddings.tok_embeddings.weight", (V, H)),
        ("ranking_model.model.embeddings.norm.weight", (H,)),
    ]
    for i in range(dims.num_layers):
        p = f"ranking_model.model.layers.{i}"
        if i != 0:
            keys.append((f"{p}.attn_norm.weight", (H,)))
        keys += [
            (f"{p}.attn.Wqkv.weight", (3 * H, H)),
            (f"{p}.attn.Wo.weight", (H, H)),
            (f"{p}.mlp_norm.weight", (H,)),
            (f"{p}.mlp.Wi.weight", (2 * I, H)),
            (f"{p}.mlp.Wo.weight", (H, I)),
        ]
    keys += [
        ("ranking_model.model.final_norm.weight", (H,)),
        ("ranking_model.head.dense.weight", (H, H)),
        ("ranking_model.head.norm.weight", (H,)),
        ("ranking_model.classifier.weight", (nl, H)),
        ("ranking_model.classifier.bias", (nl,)),
        ("pruning_head.classifier.weight", (2, H)),
        ("pruning_head.classifier.bias", (2,)),
    ]
    return keys


def synth_state_dict(dims: EncoderDims, seed: int) -> dict[str, torch.Tensor]:
    """fp32 CPU state dict with reference key names; identical bits for identical (dims, seed)."""

    H, I = dims.hidden_size, dims.intermediate_size
    s3 = math.sqrt(3.0)  # uniform(-a, a) has std a/sqrt(3)
    out: dict[str, torch.Tensor] = {}
    for tag, (name, shape) in enumerate(state_dict_keys(dims)):
        if name.endswith("norm.weight"):
            tensor = _filled(seed, tag, shape, 0.25, 1.0)
        elif name.endswith("tok_embeddings.weight"):
            tensor = _filled(seed, tag, shape, 1.0)
        elif name.endswith("Wqkv.weight"):
            tensor = _filled(seed, tag, shape, 1.5 * s3 / math.sqrt(H))
        elif name.endswith("attn.Wo.weight") or name.endswith("Wi.weight") or name.endswith("dense.weight"):
            tensor = _filled(seed, tag, shape, s3 / math.sqrt(H))
        elif name.endswith("mlp.Wo.weight"):
            tensor = _filled(seed, tag, shape, s3 / math.sqrt(I))
        elif name.endswith("classifier.weight"):
            tensor = _filled(seed, tag, shape, s3 / math.sqrt(H))
        elif name.endswith("classifier.bias"):
            tensor = _filled(seed, tag, shape, 0.5)
        else:  # pragma: no cover - key list and branches are kept in sync
            raise KeyError(name)
        out[name] = tensor
    return out


def _truncated_normal(seed: int, tag: int, shape: Sequence[int], std: float, cutoff: float) -> torch.Tensor:
    """N(0, std^2) truncated to +-cutoff * std (resampling, like ``nn.init.trunc_normal_``'s distribution), from the
    same counter-based stream: Box-Muller on float64 uniforms, so the values depend on (seed, tag, index) only."""

    count = int(np.prod(shape))
    out = np.empty(count, dtype=np.float64)
    todo = np.arange(count)
    attempt = 0
    while todo.size:
        with np.errstate(over="ignore"):
            base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x1000003) + np.uint64(tag) + (np.uint64(attempt) << np.uint64(40))], dtype=np.uint64))[0]
            idx = todo.astype(np.uint64) * np.uint64(2) + base
        b1 = _splitmix64(idx) >> np.uint64(11)
        b2 = _splitmix64(idx + np.uint64(1)) >> np.uint64(11)
        u1 = (b1.astype(np.float64) + 1.0) * (1.0 / (1 << 53))  # (0, 1]
        u2 = b2.astype(np.float64) * (1.0 / (1 << 53))
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        ok = np.abs(z) <= cutoff
        out[todo[ok]] = z[ok]
        todo = todo[~ok]
        attempt += 1
    values = (out * std).astype(np.float32)
    return torch.from_numpy(values.reshape(tuple(shape))).clone()


def refinit_state_dict(dims: EncoderDims, seed: int, *, initializer_range: float = 0.02, cutoff_factor: float = 2.0) -> dict[str, torch.Tensor]:
    """Checkpoint-shaped state dict drawn from the distributions the reference's OWN initialisation uses -- HF
    ``ModernBertPreTrainedModel._init_weights`` (modeling_modernbert.py:353-400: truncated normal, std 0.02 for
    embeddings / Wqkv / Wi, 0.02 / sqrt(2 N) for the output projections and the head's dense, H^-0.5 for the
    classifier, norm weights 1, biases 0) and ``OpenProvenceHead._init_weights`` (standalone.py:427-432: Xavier-uniform
    pruning classifier, zero bias) -- but regenerated from ``(dims, seed)`` so that full-depth fixtures need not store
    10^7..10^8 weights.  The numerical regime (tiny residual branches, near-uniform attention, logits of order
    0.1) is the one a freshly constructed reference model is in; :func:`synth_state_dict` is the hard regime."""

    H, nl = dims.hidden_size, dims.num_labels
    std_in = initializer_range
    std_out = initializer_range / math.sqrt(2.0 * dims.num_layers)
    out: dict[str, torch.Tensor] = {}
    for tag, (name, shape) in enumerate(state_dict_keys(dims)):
        if name.endswith("norm.weight"):
            tensor = torch.ones(shape, dtype=torch.float32)
        elif name.endswith("tok_embeddings.weight") or name.endswith("Wqkv.weight") or name.endswith("Wi.weight"):
            tensor = _truncated_normal(seed, tag, shape, std_in, cutoff_factor)
        elif name.endswith("attn.Wo.weight") or name.endswith("mlp.Wo.weight") or name.endswith("dense.weight"):
            tensor = _truncated_normal(seed, tag, shape, std_out, cutoff_factor)
        elif name == "ranking_model.classifier.weight":
            tensor = _truncated_normal(seed, tag, shape, H ** -0.5, cutoff_factor)
        elif name == "pruning_head.classifier.weight":
            tensor = _filled(seed, tag, shape, math.sqrt(6.0 / (H + 2)))
        elif name.endswith("classifier.bias"):
            tensor = torch.zeros(shape, dtype=torch.float32)
        else:  # pragma: no cover - key list and branches are kept in sync
            raise KeyError(name)
        out[name] = tensor
    return out


def _unit_stream(seed: int, tag: int, count: int) -> np.ndarray:
    """float64 uniforms in (0, 1) from the counter-based stream (values depend on (seed, tag, index) only)."""

    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x2000005) + np.uint64(tag)], dtype=np.uint64))[0]
        bits = _splitmix64(np.arange(count, dtype=np.uint64) + base) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def trained_like_state_dict(dims: EncoderDims, seed: int, *, row_sigma: float = 0.5, gain_range: tuple[float, float] = (0.1, 10.0),
                            n_outlier_channels: int = 4, outlier_range: tuple[float, float] = (30.0, 100.0),
                            zipf_alpha: float = 0.35) -> dict[str, torch.Tensor]:
    """A PROXY for a trained checkpoint (none can be downloaded here): :func:`refinit_state_dict`, then the statistics that
    set a trained encoder apart from a freshly initialised one and that decide which arithmetic a checkpoint can take
    (what is mirrored is the load-time choice of standalone.py:219-244, 1631-1642):

    * heavy-tailed weights: every row of a GEMM weight scaled by a log-normal factor exp(row_sigma * z), z ~ N(0, 1);
    * LayerNorm gains log-uniform in ``gain_range`` instead of 1;
    * outlier hidden channels: ``n_outlier_channels`` of the hidden size carry activations 30-100x the others -- the rows of
      every output projection (attention Wo, MLP Wo) and the embedding columns of those channels are scaled by a factor
      log-uniform in ``outlier_range`` (the outlier-feature pattern of trained transformer encoders);
    * token embeddings with Zipf-distributed norms: row r of a random permutation scaled by (1 + r) ** -zipf_alpha
      (frequent tokens have large embeddings), renormalised to the initialisation's mean square.

    Deterministic in ``(dims, seed)``.  Not a claim about any published checkpoint: a stress proxy between the reference
    initialisation (:func:`refinit_state_dict`) and the O(1) worst case (:func:`synth_state_dict`)."""

    H = dims.hidden_size
    out = refinit_state_dict(dims, seed)
    u = _unit_stream(seed, 9001, n_outlier_channels * 2)
    channels = sorted({int(u[i] * H) % H for i in range(n_outlier_channels)})
    factors = {c: float(np.exp(np.log(outlier_range[0]) + u[n_outlier_channels + i] * np.log(outlier_range[1] / outlier_range[0])))
               for i, c in enumerate(channels)}
    for tag, (name, tensor) in enumerate(list(out.items())):
        t = tensor.clone()
        if name.endswith("norm.weight"):
            g = _unit_stream(seed, 7000 + tag, t.numel())
            t = torch.from_numpy(np.exp(np.log(gain_range[0]) + g * np.log(gain_range[1] / gain_range[0])).astype(np.float32)).reshape(t.shape)
        elif t.ndim == 2 and name.endswith(("Wqkv.weight", "Wi.weight", "attn.Wo.weight", "mlp.Wo.weight", "dense.weight")):
            z = _unit_stream(seed, 8000 + tag, 2 * t.shape[0])
            normal = np.sqrt(-2.0 * np.log(z[: t.shape[0]])) * np.cos(2.0 * np.pi * z[t.shape[0]:])
            t = t * torch.from_numpy(np.exp(row_sigma * normal).astype(np.float32))[:, None]
            if name.endswith(("attn.Wo.weight", "mlp.Wo.weight")):  # output feature c = row c
                for c, f in factors.items():
                    t[c] *= f
        elif name.endswith("tok_embeddings.weight"):
            V = t.shape[0]
            order = np.argsort(_unit_stream(seed, 8500, V))  # a random permutation: rank of every token
            scale = (1.0 + np.arange(V, dtype=np.float64)) ** (-zipf_alpha)
            scale *= 1.0 / np.sqrt(np.mean(scale ** 2))
            per_row = np.empty(V, dtype=np.float32)
            per_row[order] = scale.astype(np.float32)
            t = t * torch.from_numpy(per_row)[:, None]
            for c, f in factors.items():
                t[:, c] *= f
        out[name] = t.contiguous()
    return out


def zipf_token_rows(dims: EncoderDims, n_rows: int, seq_len: int, seed: int, *, alpha: float = 1.1) -> list[list[int]]:
    """Rows shaped like :func:`synth_pair_batch`'s ([CLS] query [SEP] context [SEP]) whose token ids follow a Zipf law over a
    random permutation of the vocabulary (a few ids make up most of the text) instead of the uniform ids of the other
    synthetic batches: calibration / audit rows for :func:`trained_like_state_dict`."""

    V = dims.vocab_size
    margin = 1000 if V > 4000 else max(4, V // 8)
    lo, n_vocab = margin, V - 2 * margin
    weights = (1.0 + np.arange(n_vocab, dtype=np.float64)) ** (-alpha)
    cdf = np.cumsum(weights / weights.sum())
    perm = np.argsort(_unit_stream(seed, 8600, n_vocab))

    def draw(tag: int, count: int) -> list[int]:
        u = _unit_stream(seed, tag, count)
        return [int(t) for t in lo + perm[np.minimum(np.searchsorted(cdf, u), n_vocab - 1)]]

    cls_id = dims.cls_token_id if dims.cls_token_id is not None else 1
    sep_id = dims.sep_token_id if dims.sep_token_id is not None else 2
    query_tokens = 24
    if seq_len < query_tokens + 4:
        raise ValueError("seq_len too short")
    query = draw(8699, query_tokens)  # ONE query shared by all contexts
    return [[cls_id] + query + [sep_id] + draw(8700 + r, seq_len - query_tokens - 3) + [sep_id] for r in range(n_rows)]


def synth_pair_batch(
    dims: EncoderDims,
    n_pairs: int,
    seq_len: int | Sequence[int],
    *,
    seed: int = 1234,
    query_tokens: int = 24,
) -> list[list[int]]:
    """Token-id rows for ``n_pairs`` (query, context) pairs sharing ONE query."""

    rng = np.random.default_rng(seed)
    V = dims.vocab_size
    margin = 1000 if V > 4000 else max(4, V // 8)
    lo, hi = margin, V - margin
    cls_id = dims.cls_token_id if dims.cls_token_id is not None else 1
    sep_id = dims.sep_token_id if dims.sep_token_id is not None else 2
    lengths = [int(seq_len)] * n_pairs if isinstance(seq_len, (int, np.integer)) else [int(v) for v in seq_len]
    if len(lengths) != n_pairs:
        raise ValueError("seq_len sequence must have n_pairs entries")
    query = rng.integers(lo, hi, size=query_tokens).tolist()
    rows: list[list[int]] = []
    for length in lengths:
        n_ctx = length - (query_tokens + 3)
        if n_ctx < 1:
            raise ValueError(f"seq_len {length} too short for {query_tokens} query tokens + 3 specials")
        ctx = rng.integers(lo, hi, size=n_ctx).tolist()
        rows.append([cls_id] + query + [sep_id] + ctx + [sep_id])
    return rows


def synth_varlen_lengths(n_tokens_target: int, *, seed: int = 1234) -> list[int]:
    """Config C5: lengths from {128..2048} with probability proportional to 1/L until T ~ target."""

    choices = np.array([128, 256, 384, 512, 768, 1024, 1536, 2048])
    probs = (1.0 / choices) / (1.0 / choices).sum()
    rng = np.random.default_rng(seed)
    lengths: list[int] = []
    total = 0
    while total < n_tokens_target:
        length = int(rng.choice(choices, p=probs))
        lengths.append(length)
        total += length
    return lengths


def pad_rows(rows: Sequence[Sequence[int]], pad_id: int = 0) -> tuple[torch.Tensor, torch.Tensor]:
    """Right-pad id rows to a dense ``[B, Lmax]`` pair (input_ids, attention_mask), int64 like the reference."""

    max_len = max((len(r) for r in rows), default=0)
    ids = torch.full((len(rows), max_len), int(pad_id), dtype=torch.long)
    mask = torch.zeros((len(rows), max_len), dtype=torch.long)
    for i, row in enumerate(rows):
        ids[i, : len(row)] = torch.tensor(list(row), dtype=torch.long)
        mask[i, : len(row)] = 1
    return ids, mask


XSMALL = dict(vocab_size=102400, hidden_size=256, intermediate_size=1024, num_hidden_layers=10, num_attention_heads=4)
BASE = dict(vocab_size=102400, hidden_size=512, intermediate_size=2048, num_hidden_layers=19, num_attention_heads=8)
LARGE = dict(vocab_size=102400, hidden_size=768, intermediate_size=3072, num_hidden_layers=25, num_attention_heads=12)
EN_GTE = dict(vocab_size=50368, hidden_size=768, intermediate_size=1152, num_hidden_layers=22, num_attention_heads=12)


def named_dims(name: str, **overrides: int) -> EncoderDims:
    """Model-card dimensions of the four published checkpoints (not in the reference tree)."""

    table = {"xsmall": XSMALL, "base": BASE, "large": LARGE, "en-gte": EN_GTE}
    cfg = dict(table[name])
    cfg.update(
        model_type="modernbert",
        local_attention=128,
        global_attn_every_n_layers=3,
        global_rope_theta=160000.0,
        local_rope_theta=10000.0,
        max_position_embeddings=8192,
        pad_token_id=3,
        cls_token_id=1,
        sep_token_id=2,
    )
    if "num_layers" in overrides:  # the EncoderDims name of the HF key
        overrides["num_hidden_layers"] = overrides.pop("num_layers")
    unknown = sorted(set(overrides) - set(cfg))
    if unknown:  # (an ignored override runs a different model than the caller asked for)
        raise TypeError(f"named_dims: unknown override(s) {unknown}; known: {sorted(cfg)}")
    cfg.update(overrides)
    return EncoderDims.from_base_model_config(cfg, num_labels=1)
