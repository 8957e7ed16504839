"""Configuration objects for the MI355X-native OpenProvence forward path.

Two things live here:

* :class:`OpenProvenceConfig` -- the checkpoint-level ``config.json`` schema of the reference
  (``open_provence/modeling_open_provence_standalone.py:1246-1302``): ``base_model_config``,
  ``pruning_config``, ``max_length``, ``num_labels``, ``default_threadshold`` (legacy spelling kept,
  ``default_threshold`` accepted with a warning).
* :class:`EncoderDims` -- the subset of the ModernBERT backbone configuration the HIP kernels need
  (third-party ``transformers/models/modernbert/configuration_modernbert.py:77-162``): dims, layer
  types, the two RoPE thetas, the half window of the local layers, eps, pooling.

Neither class imports ``transformers``; they are plain Python so that the C-ABI layer can be fed from
a JSON file alone.
"""

from __future__ import annotations

import json
import warnings
from copy import deepcopy
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Mapping

DEFAULT_PROCESS_THRESHOLD = 0.1  # reference: standalone.py:57

_DEFAULT_GLOBAL_THETA = 160_000.0  # configuration_modernbert.py:77
_DEFAULT_LOCAL_THETA = 10_000.0


class UnsupportedModelError(ValueError):
    """Raised when a checkpoint asks for arithmetic the HIP path does not implement."""


@dataclass(frozen=True)
class EncoderDims:
    """Backbone dimensions and constants consumed by ``op_create`` (see include/open_provence_hip.h)."""

    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_layers: int
    num_heads: int
    num_labels: int = 1
    local_attention: int = 128
    layer_is_global: tuple[bool, ...] = ()
    global_rope_theta: float = _DEFAULT_GLOBAL_THETA
    local_rope_theta: float = _DEFAULT_LOCAL_THETA
    norm_eps: float = 1e-5
    classifier_pooling: str = "cls"
    max_position_embeddings: int = 8192
    pad_token_id: int | None = None
    cls_token_id: int | None = None
    sep_token_id: int | None = None

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def half_window(self) -> int:
        # HF: sliding_window = local_attention // 2, mask is |q-k| <= sliding_window
        # (configuration_modernbert.py:159-162, masking_utils.py:141-150).
        return self.local_attention // 2

    @classmethod
    def from_base_model_config(cls, raw: Mapping[str, Any], *, num_labels: int | None = None) -> "EncoderDims":
        cfg = dict(raw)
        model_type = cfg.get("model_type", "modernbert")
        if model_type != "modernbert":
            raise UnsupportedModelError(
                f"The MI355X path implements the ModernBERT backbone only; got model_type={model_type!r}."
            )
        for flag in ("attention_bias", "mlp_bias", "norm_bias", "classifier_bias"):
            if cfg.get(flag, False):
                raise UnsupportedModelError(f"ModernBERT option {flag}=True is not supported by the HIP kernels.")
        for key in ("hidden_activation", "classifier_activation"):
            if cfg.get(key, "gelu") != "gelu":
                raise UnsupportedModelError(f"{key}={cfg.get(key)!r}: only exact-erf 'gelu' is implemented.")

        hidden = int(cfg["hidden_size"])
        heads = int(cfg["num_attention_heads"])
        n_layers = int(cfg["num_hidden_layers"])
        if hidden % heads:
            raise UnsupportedModelError("hidden_size must be divisible by num_attention_heads")

        layer_types = cfg.get("layer_types")
        if layer_types is None:
            every = int(cfg.get("global_attn_every_n_layers", 3))
            is_global = tuple((i % every) == 0 for i in range(n_layers))
        else:
            if len(layer_types) != n_layers:
                raise UnsupportedModelError("layer_types length does not match num_hidden_layers")
            is_global = tuple(str(t) == "full_attention" for t in layer_types)

        g_theta = cfg.get("global_rope_theta")
        l_theta = cfg.get("local_rope_theta")
        rope_params = cfg.get("rope_parameters") or {}
        if isinstance(rope_params, Mapping):
            full = rope_params.get("full_attention") or {}
            slid = rope_params.get("sliding_attention") or {}
            for block in (full, slid):
                if block.get("rope_type", "default") != "default":
                    raise UnsupportedModelError("only rope_type='default' is implemented")
            if g_theta is None:
                g_theta = full.get("rope_theta")
            if l_theta is None:
                l_theta = slid.get("rope_theta")
        if g_theta is None:
            g_theta = cfg.get("rope_theta", _DEFAULT_GLOBAL_THETA)
        if l_theta is None:
            l_theta = _DEFAULT_LOCAL_THETA

        pooling = str(cfg.get("classifier_pooling", "cls"))
        if pooling not in {"cls", "mean"}:
            raise UnsupportedModelError(f"classifier_pooling={pooling!r} is not supported")

        labels = num_labels if num_labels is not None else cfg.get("num_labels")
        if labels is None:
            id2label = cfg.get("id2label")
            labels = len(id2label) if id2label else 1

        return cls(
            vocab_size=int(cfg["vocab_size"]),
            hidden_size=hidden,
            intermediate_size=int(cfg["intermediate_size"]),
            num_layers=n_layers,
            num_heads=heads,
            num_labels=int(labels),
            local_attention=int(cfg.get("local_attention", 128)),
            layer_is_global=is_global,
            global_rope_theta=float(g_theta),
            local_rope_theta=float(l_theta),
            norm_eps=float(cfg.get("norm_eps", 1e-5)),
            classifier_pooling=pooling,
            max_position_embeddings=int(cfg.get("max_position_embeddings", 8192)),
            pad_token_id=cfg.get("pad_token_id"),
            cls_token_id=cfg.get("cls_token_id", cfg.get("bos_token_id")),
            sep_token_id=cfg.get("sep_token_id", cfg.get("eos_token_id")),
        )

    def to_base_model_config(self) -> dict[str, Any]:
        """Inverse of :meth:`from_base_model_config` (HF-style keys, 4.x spelling for the thetas)."""

        return {
            "model_type": "modernbert",
            "vocab_size": self.vocab_size,
            "hidden_size": self.hidden_size,
            "intermediate_size": self.intermediate_size,
            "num_hidden_layers": self.num_layers,
            "num_attention_heads": self.num_heads,
            "local_attention": self.local_attention,
            "layer_types": ["full_attention" if g else "sliding_attention" for g in self.layer_is_global],
            "global_rope_theta": self.global_rope_theta,
            "local_rope_theta": self.local_rope_theta,
            "norm_eps": self.norm_eps,
            "classifier_pooling": self.classifier_pooling,
            "max_position_embeddings": self.max_position_embeddings,
            "pad_token_id": self.pad_token_id,
            "cls_token_id": self.cls_token_id,
            "sep_token_id": self.sep_token_id,
            "bos_token_id": self.cls_token_id,
            "eos_token_id": self.sep_token_id,
        }


class OpenProvenceConfig:
    """Checkpoint configuration; field names follow the reference (standalone.py:1246-1302)."""

    model_type = "open_provence"

    def __init__(
        self,
        mode: str = "reranking_pruning",
        base_model_name_or_path: str | None = None,
        base_model_config: Mapping[str, Any] | None = None,
        tokenizer_name_or_path: str | None = None,
        pruning_config: Mapping[str, Any] | None = None,
        max_length: int = 512,
        num_labels: int | None = None,
        num_pruning_labels: int | None = None,
        encoder_architecture: str | None = None,
        **kwargs: Any,
    ) -> None:
        raw_threadshold = kwargs.pop("default_threadshold", None)
        alt_threshold = kwargs.pop("default_threshold", None)
        kwargs.pop("splitter_default_language", None)
        kwargs.pop("standalone_process_default_language", None)
        self._name_or_path = kwargs.pop("_name_or_path", None) or kwargs.pop("name_or_path", None)
        self.mode = mode
        self.base_model_name_or_path = base_model_name_or_path
        if base_model_config is not None and hasattr(base_model_config, "to_dict"):
            base_model_config = base_model_config.to_dict()  # PretrainedConfig-like
        self.base_model_config = deepcopy(dict(base_model_config)) if base_model_config is not None else None
        self.tokenizer_name_or_path = tokenizer_name_or_path
        self.pruning_config = dict(pruning_config or {})
        self.max_length = int(max_length)
        self.encoder_architecture = encoder_architecture
        self.num_labels = 1 if num_labels is None else int(num_labels)
        self.num_pruning_labels = 2 if num_pruning_labels is None else int(num_pruning_labels)
        self.default_threadshold = None
        if raw_threadshold is not None:
            try:
                self.default_threadshold = float(raw_threadshold)
            except (TypeError, ValueError) as exc:
                raise TypeError(
                    "Config value 'default_threadshold' must be a numeric type convertible to float."
                ) from exc
        elif alt_threshold is not None:
            warnings.warn(
                "Config key 'default_threshold' detected. Did you intend 'default_threadshold'? "
                "Using the provided value for backwards compatibility.",
                RuntimeWarning,
                stacklevel=2,
            )
            try:
                self.default_threadshold = float(alt_threshold)
            except (TypeError, ValueError) as exc:
                raise TypeError(
                    "Config value 'default_threshold' must be a numeric type convertible to float."
                ) from exc
        self.extra = dict(kwargs)

    # The reference mirrors the value under the corrected spelling too (standalone.py:1302).
    @property
    def default_threshold(self) -> float | None:
        return self.default_threadshold

    def encoder_dims(self) -> EncoderDims:
        if not self.base_model_config:
            raise ValueError(
                "OpenProvenceConfig.base_model_config is required: the MI355X path rebuilds the backbone "
                "from the checkpoint's own config and never downloads a base model."
            )
        return EncoderDims.from_base_model_config(self.base_model_config, num_labels=self.num_labels)

    def to_dict(self) -> dict[str, Any]:
        payload: dict[str, Any] = {
            "model_type": self.model_type,
            "mode": self.mode,
            "base_model_name_or_path": self.base_model_name_or_path,
            "base_model_config": deepcopy(self.base_model_config),
            "tokenizer_name_or_path": self.tokenizer_name_or_path,
            "pruning_config": dict(self.pruning_config),
            "max_length": self.max_length,
            "num_labels": self.num_labels,
            "num_pruning_labels": self.num_pruning_labels,
            "encoder_architecture": self.encoder_architecture,
        }
        if self.default_threadshold is not None:
            payload["default_threadshold"] = self.default_threadshold
        payload.update(self.extra)
        return payload

    @classmethod
    def from_dict(cls, payload: Mapping[str, Any]) -> "OpenProvenceConfig":
        data = dict(payload)
        data.pop("model_type", None)
        return cls(**data)

    @classmethod
    def from_json_file(cls, path: str | Path) -> "OpenProvenceConfig":
        with open(path, "r", encoding="utf-8") as handle:
            payload = json.load(handle)
        cfg = cls.from_dict(payload)
        if cfg._name_or_path is None:
            cfg._name_or_path = str(Path(path).parent)
        cfg._from_file = True  # (resolve_pruning_hidden_state: a checkpoint's config, not one built in code)
        return cfg

    def save_json(self, path: str | Path) -> None:
        with open(path, "w", encoding="utf-8") as handle:
            json.dump(self.to_dict(), handle, indent=2, sort_keys=True)


__all__ = [
    "DEFAULT_PROCESS_THRESHOLD",
    "EncoderDims",
    "OpenProvenceConfig",
    "UnsupportedModelError",
]
