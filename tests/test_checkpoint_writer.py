"""Checkpoint writer (reference writer encoder.py:1040-1094, reader standalone.py:1557-1664).
CPU part: the files, keys and values ``save_pretrained`` produces.  (Checked once in the build container, where the
reference is importable: the reference's own OpenProvenceConfig.from_pretrained reads this config.json, its model
loads this model.safetensors with no missing / unexpected keys and reproduces the oracle's logits to 1.5e-6.)"""

from __future__ import annotations

import json

import pytest
import torch
from safetensors.torch import load_file

from helpers import CharTokenizer
from open_provence_amd import modeling
from open_provence_amd.config import EncoderDims, OpenProvenceConfig
from open_provence_amd.synthetic import state_dict_keys, synth_state_dict

BASE = dict(model_type="modernbert", vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=3,
            num_attention_heads=2, local_attention=32, global_attn_every_n_layers=3, global_rope_theta=160000.0,
            local_rope_theta=10000.0, max_position_embeddings=512, pad_token_id=0, cls_token_id=1, sep_token_id=2)


class SavingTokenizer(CharTokenizer):
    def save_pretrained(self, directory: str) -> None:
        with open(f"{directory}/tokenizer_config.json", "w", encoding="utf-8") as handle:
            json.dump({"tokenizer_class": "CharTokenizer"}, handle)


def _writer_only_model(state, config, tokenizer):
    model = modeling.OpenProvenceModel.__new__(modeling.OpenProvenceModel)  # no GPU: only the writer is exercised
    model.config = config
    model.max_length = 80
    model.num_labels = 1
    model.dims = EncoderDims.from_base_model_config(BASE, num_labels=1)
    model.tokenizer = tokenizer
    model._weights = dict(state)
    model.pruning_hidden_state = "post_final_norm"
    return model


def test_save_pretrained_writes_the_reference_format(tmp_path):
    dims = EncoderDims.from_base_model_config(BASE, num_labels=1)
    state = synth_state_dict(dims, 41)
    config = OpenProvenceConfig(base_model_config=BASE, tokenizer_name_or_path="char", pruning_config={"hidden_size": 128},
                                max_length=96, default_threadshold=0.2, some_future_key="kept")
    model = _writer_only_model(state, config, SavingTokenizer())
    model.save_pretrained(tmp_path)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["config.json", "model.safetensors",
                                                          "modeling_open_provence_standalone.py", "tokenizer_config.json"]

    payload = json.loads((tmp_path / "config.json").read_text(encoding="utf-8"))
    assert payload["model_type"] == "open_provence" and payload["mode"] == "reranking_pruning"
    assert payload["max_length"] == 80 and payload["num_labels"] == 1 and payload["num_pruning_labels"] == 2
    assert payload["encoder_architecture"] == "modernbert" and payload["vocab_size"] == 256 and payload["hidden_size"] == 128
    assert payload["architectures"] == ["OpenProvenceForSequenceClassification"]
    assert payload["auto_map"]["AutoModelForTokenClassification"].endswith("OpenProvenceForTokenClassification")
    assert payload["default_threadshold"] == 0.2 and payload["some_future_key"] == "kept"
    assert payload["base_model_config"]["num_hidden_layers"] == 3

    saved = load_file(str(tmp_path / "model.safetensors"))
    assert set(saved) == set(state) == {name for name, _shape in state_dict_keys(dims)}
    assert all(k.startswith(("ranking_model.", "pruning_head.")) for k in saved)
    assert all(torch.equal(saved[k], state[k]) for k in saved)

    again = OpenProvenceConfig.from_json_file(tmp_path / "config.json")
    assert again.max_length == 80 and again.default_threshold == 0.2 and again.encoder_dims() == dims


def test_state_dict_requires_loaded_weights_and_bin_format(tmp_path):
    dims = EncoderDims.from_base_model_config(BASE, num_labels=1)
    config = OpenProvenceConfig(base_model_config=BASE, tokenizer_name_or_path="char", pruning_config={"hidden_size": 128})
    model = _writer_only_model({}, config, CharTokenizer())
    model._weights = None
    with pytest.raises(RuntimeError):
        model.state_dict()
    model._weights = dict(synth_state_dict(dims, 3))
    model.save_pretrained(tmp_path, safe_serialization=False)
    loaded = torch.load(str(tmp_path / "pytorch_model.bin"), map_location="cpu", weights_only=True)
    assert set(loaded) == set(model._weights)


def test_auto_classes_resolve_to_the_hip_classes(tmp_path, monkeypatch):
    """The reference's documented entry is AutoModel.from_pretrained(id, trust_remote_code=True) through `auto_map`
    (README.md:53-74, standalone.py:3814-3906).  Both routes must land on the HIP classes: registered Auto* classes,
    and the remote-code module save_pretrained writes under the name the auto_map points to.  (CPU: the classes'
    from_pretrained is intercepted -- constructing the model needs a GPU; tests/test_gpu_model.py loads for real.)"""

    from transformers import AutoConfig, AutoModel, AutoModelForTokenClassification

    from open_provence_amd import hf_auto

    monkeypatch.setenv("HF_MODULES_CACHE", str(tmp_path / "hf_modules"))
    import transformers.dynamic_module_utils as dmu

    monkeypatch.setattr(dmu, "HF_MODULES_CACHE", str(tmp_path / "hf_modules"))
    dims = EncoderDims.from_base_model_config(BASE, num_labels=1)
    config = OpenProvenceConfig(base_model_config=BASE, tokenizer_name_or_path="char", pruning_config={"hidden_size": 128},
                                max_length=96, default_threadshold=0.2)
    ckpt = tmp_path / "ckpt"
    _writer_only_model(synth_state_dict(dims, 41), config, SavingTokenizer()).save_pretrained(ckpt)

    calls = []

    def fake_from_pretrained(cls, path, *args, **kwargs):
        calls.append((cls.__name__, hasattr(kwargs.get("config"), "to_native"), kwargs["config"].to_native().max_length))
        return cls.__name__

    monkeypatch.setattr(modeling.OpenProvenceModel, "from_pretrained", classmethod(fake_from_pretrained))

    # route 1: registered classes, no remote code
    hf_auto.register_auto_classes()
    cfg = AutoConfig.from_pretrained(str(ckpt))
    assert type(cfg) is hf_auto.OpenProvenceHFConfig and cfg.to_native().encoder_dims() == dims
    assert cfg.to_native().default_threshold == 0.2 and cfg.to_native().extra.get("pruning_hidden_state") == "post_final_norm"
    assert AutoModel.from_pretrained(str(ckpt)) == "OpenProvenceForSequenceClassification"
    assert AutoModelForTokenClassification.from_pretrained(str(ckpt)) == "OpenProvenceForTokenClassification"

    # route 2: trust_remote_code=True through the auto_map -> the module written next to the weights
    assert AutoModel.from_pretrained(str(ckpt), trust_remote_code=True) == "OpenProvenceForSequenceClassification"
    assert [c[0] for c in calls] == ["OpenProvenceForSequenceClassification", "OpenProvenceForTokenClassification",
                                     "OpenProvenceForSequenceClassification"]
    assert all(c[1] and c[2] == 80 for c in calls)


def test_reference_reads_the_written_checkpoint_report():
    """tests/golden/check_writer_report.json is produced by `make_golden.py --check-writer` in the build container (the
    reference is importable there): the reference's config class parses the written config.json, its model takes the
    written model.safetensors with strict key matching and reproduces the oracle.  (Its own from_pretrained raises
    under transformers 5.15 -- `all_tied_weights_keys` -- for ANY checkpoint directory: recorded, not a format issue.)"""

    from helpers import GOLDEN_DIR

    report = json.loads((GOLDEN_DIR / "check_writer_report.json").read_text(encoding="utf-8"))
    assert report["strict_load"] == {"missing": [], "unexpected": []}
    assert report["reference_config_reads"] == {"max_length": 96, "num_labels": 1, "hidden_size": 128, "default_threadshold": 0.2}
    assert report["reference_on_written_checkpoint_vs_oracle"]["max_abs_prune"] < 5e-5
    assert report["reference_on_written_checkpoint_vs_oracle"]["max_abs_rank"] < 5e-5
    assert "all_tied_weights_keys" in report["reference_from_pretrained"]


def test_hf_config_round_trips_num_labels_and_serialises_no_bookkeeping():
    """AutoConfig route: a checkpoint with num_labels != 1 keeps its ranking-head width in the native config, the
    default stays the reference's 1 (standalone.py:1279), and config.json gets no private bookkeeping keys."""

    from open_provence_amd.hf_auto import OpenProvenceHFConfig

    three = OpenProvenceHFConfig(base_model_config=BASE, pruning_config={"hidden_size": 128}, max_length=96, num_labels=3)
    assert three.to_native().num_labels == 3
    assert three.to_native().encoder_dims().num_labels == 3
    plain = OpenProvenceHFConfig(base_model_config=BASE, pruning_config={"hidden_size": 128}, max_length=96)
    assert plain.to_native().num_labels == 1
    payload = three.to_dict()
    assert not any(key.startswith("_native") for key in payload), sorted(payload)
    again = OpenProvenceHFConfig(**{k: v for k, v in payload.items() if k != "model_type"})
    assert again.to_native().num_labels == 3 and again.to_native().max_length == 96


def test_pruning_hidden_state_auto_is_logged_and_defaults_by_origin(caplog):
    """'auto' is a heuristic that changes pruning logits: it is logged, a stamped config follows its stamp, an unstamped
    CHECKPOINT config takes the reference lock's convention (transformers 4.57.1 = pre-norm), an explicit value wins."""

    import logging

    from open_provence_amd.modeling import resolve_pruning_hidden_state

    base = dict(base_model_config=BASE, pruning_config={"hidden_size": 128})
    with caplog.at_level(logging.WARNING, logger="open_provence_amd.modeling"):
        assert resolve_pruning_hidden_state(OpenProvenceConfig(**base, transformers_version="4.57.1")) == "pre_final_norm"
        assert resolve_pruning_hidden_state(OpenProvenceConfig(**base, transformers_version="5.15.0")) == "post_final_norm"
        from_file = OpenProvenceConfig(**base)
        from_file._from_file = True
        assert resolve_pruning_hidden_state(from_file) == "pre_final_norm"
    assert sum("pruning_hidden_state='auto'" in r.getMessage() for r in caplog.records) == 3
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="open_provence_amd.modeling"):
        assert resolve_pruning_hidden_state(OpenProvenceConfig(**base, pruning_hidden_state="post_final_norm")) == "post_final_norm"
        assert resolve_pruning_hidden_state(from_file, "post_final_norm") == "post_final_norm"
    assert not caplog.records
