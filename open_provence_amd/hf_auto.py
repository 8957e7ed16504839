"""``AutoModel`` entry of the reference (README.md:53-74, standalone.py:3814-3906), landing on the HIP classes.

The reference is reached as ``AutoModel.from_pretrained(id, trust_remote_code=True)``: its checkpoints carry an
``auto_map`` naming ``modeling_open_provence_standalone.<Class>`` and a copy of that module next to the weights.  Two
routes are provided here, both optional (``transformers`` is imported only inside this module):

* :func:`register_auto_classes` -- ``AutoConfig.register("open_provence", ...)`` + ``AutoModel*.register(...)``: any
  checkpoint directory in the reference's format then resolves to the HIP classes without remote code;
* :func:`write_remote_code_shim` -- ``save_pretrained`` drops a freshly written
  ``modeling_open_provence_standalone.py`` next to the weights that re-exports the HIP classes under the names the
  ``auto_map`` points to, so ``AutoModel.from_pretrained(dir, trust_remote_code=True)`` loads THIS implementation
  (the reference copies its own 3.9 k-line module there: encoder.py:1040-1094).
"""

from __future__ import annotations

from pathlib import Path
from typing import Any

from transformers import PretrainedConfig

from .config import OpenProvenceConfig as NativeConfig

REMOTE_MODULE = "modeling_open_provence_standalone"

SHIM_SOURCE = '''"""Remote-code entry written by open_provence_amd.save_pretrained (MI355X-native OpenProvence).

``AutoModel.from_pretrained(<this directory>, trust_remote_code=True)`` resolves the ``auto_map`` of config.json to the
names below; they are the hand-written-HIP implementation of the package ``open_provence_amd`` (which must be
importable), not a copy of the reference's PyTorch module.
"""

from open_provence_amd.hf_auto import OpenProvenceHFConfig as OpenProvenceConfig  # noqa: F401
from open_provence_amd.modeling import (  # noqa: F401
    OpenProvenceForSequenceClassification,
    OpenProvenceForTokenClassification,
    OpenProvenceModel,
)
'''


class OpenProvenceConfig(PretrainedConfig):  # same NAME as the native class: AutoModel.register compares names
    """``PretrainedConfig`` face of :class:`open_provence_amd.config.OpenProvenceConfig` (same keys as the reference's
    config, standalone.py:1246-1302) -- what ``AutoConfig`` instantiates; :meth:`to_native` hands the model its own."""

    model_type = "open_provence"

    def __init__(self, **kwargs: Any) -> None:
        native = {k: kwargs.pop(k) for k in list(kwargs) if k in _NATIVE_FIELDS}
        # whether the source config.json carried num_labels (PretrainedConfig always has the attribute: default 2)
        explicit_labels = "num_labels" in kwargs or "id2label" in kwargs
        super().__init__(**kwargs)
        for key, value in native.items():
            setattr(self, key, value)
        if not explicit_labels:
            self.num_labels = 1  # the reference's default (standalone.py:1279)

    def to_native(self) -> NativeConfig:
        payload = {k: getattr(self, k) for k in _NATIVE_FIELDS if hasattr(self, k)}
        payload["num_labels"] = int(self.num_labels)  # ranking head width: a checkpoint with num_labels != 1 must keep it
        for extra in ("transformers_version",):
            if getattr(self, extra, None) is not None:
                payload[extra] = getattr(self, extra)
        cfg = NativeConfig.from_dict(payload)
        cfg._name_or_path = getattr(self, "_name_or_path", None) or cfg._name_or_path
        return cfg


OpenProvenceHFConfig = OpenProvenceConfig


_NATIVE_FIELDS = (
    "mode", "base_model_name_or_path", "base_model_config", "tokenizer_name_or_path", "pruning_config", "max_length",
    "num_pruning_labels", "encoder_architecture", "default_threadshold", "default_threshold",
    "splitter_default_language", "standalone_process_default_language", "pruning_hidden_state",
)


def register_auto_classes() -> None:
    """Idempotent: ``AutoConfig`` / ``AutoModel`` / ``AutoModelForSequenceClassification`` /
    ``AutoModelForTokenClassification`` resolve ``model_type == "open_provence"`` to the HIP classes."""

    from transformers import AutoConfig, AutoModel, AutoModelForSequenceClassification, AutoModelForTokenClassification

    from .modeling import OpenProvenceForSequenceClassification, OpenProvenceForTokenClassification

    AutoConfig.register("open_provence", OpenProvenceHFConfig, exist_ok=True)
    AutoModel.register(OpenProvenceHFConfig, OpenProvenceForSequenceClassification, exist_ok=True)
    AutoModelForSequenceClassification.register(OpenProvenceHFConfig, OpenProvenceForSequenceClassification, exist_ok=True)
    AutoModelForTokenClassification.register(OpenProvenceHFConfig, OpenProvenceForTokenClassification, exist_ok=True)


def write_remote_code_shim(directory: str | Path) -> Path:
    path = Path(directory) / f"{REMOTE_MODULE}.py"
    path.write_text(SHIM_SOURCE, encoding="utf-8")
    return path
