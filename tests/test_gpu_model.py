"""GPU tests of the drop-in class: process() end to end on the HIP path against the real reference's
outputs (g3_process_model), the padded forward boundary, checkpoint loading, and size-independent
properties at the BASELINE.json batch size."""

import json

import numpy as np
import pytest
import torch

from helpers import (
    GOLDEN_DIR,
    CharTokenizer,
    assert_process_result_matches,
    dims_from_meta,
    load_golden,
    period_splitter,
    rows_from_fixture,
    state_from_fixture,
)

pytestmark = pytest.mark.gpu


def _g3_model(precision="bf16x3", cls=None):
    from open_provence_amd.config import OpenProvenceConfig
    from open_provence_amd.modeling import OpenProvenceModel
    from open_provence_amd.synthetic import synth_state_dict

    with open(GOLDEN_DIR / "g3_process_model.json", "r", encoding="utf-8") as handle:
        meta = json.load(handle)
    cfg = OpenProvenceConfig(
        base_model_config=meta["base_model_config"],
        tokenizer_name_or_path="char-tokenizer",
        pruning_config={"hidden_size": meta["base_model_config"]["hidden_size"]},
        max_length=meta["max_length"],
        num_labels=1,
    )
    state = synth_state_dict(dims_from_meta(meta), meta["weight_seed"])
    klass = cls or OpenProvenceModel
    model = klass(cfg, device="cuda:0", tokenizer=CharTokenizer(), state_dict=state, precision=precision)
    return model, meta


def test_process_end_to_end_matches_reference():
    """Identical kept/removed sentence sets and pruned text; probabilities / scores within 1e-3."""

    model, meta = _g3_model()
    for case in meta["cases"]:
        result = model.process(
            question=case["question"],
            context=case["context"],
            sentence_splitter=period_splitter,
            show_progress=False,
            return_sentence_metrics=True,
            return_sentence_texts=True,
            batch_size=4,
            **case["kwargs"],
        )
        assert_process_result_matches(result, case["expected"], prob_tol=1e-3, score_tol=1e-3)
        assert result["timing"]["inference_seconds"] > 0.0


@pytest.mark.timeout(600)
def test_host_front_end_on_the_gpu_equals_the_plain_call(monkeypatch):
    """``HostFrontEnd``: host-stage replicas without a GPU, every forward batch merged and run by this process -- the G3
    cases against the reference, and a 90-context request against the plain call (probabilities and scores bit for
    bit: a row's outputs do not depend on its batch companions, the fragment means are taken on the device either way)."""

    from open_provence_amd.frontend import HostFrontEnd

    model, meta = _g3_model()
    words = "the tower is tall boats carry fish and salt to north city harbour many years ago it was new".split()
    contexts = [
        " ".join(" ".join(words[(i * 5 + s * 3 + k) % len(words)] for k in range(4 + (i + s) % 5)).capitalize() + "." for s in range(1 + i % 6))
        for i in range(90)
    ]
    big = dict(question="which boats carry salt?", context=contexts, sentence_splitter=period_splitter, show_progress=False,
               return_sentence_metrics=True, return_sentence_texts=True, batch_size=16, threshold=0.4)
    want = model.process(**big)
    want_top = model.process(reorder=True, top_k=7, **big)
    with HostFrontEnd(model, workers=3) as front:
        for case in meta["cases"]:
            result = front.process(question=case["question"], context=case["context"], sentence_splitter=period_splitter,
                                   show_progress=False, return_sentence_metrics=True, return_sentence_texts=True, batch_size=4,
                                   **case["kwargs"])
            assert_process_result_matches(result, case["expected"], prob_tol=1e-3, score_tol=1e-3)
        got = front.process(**big)
        got_top = front.process(reorder=True, top_k=7, **big)
    for key in want:
        if key in ("timing", "performance_trace"):
            continue
        assert got[key] == want[key], key
        assert got_top[key] == want_top[key], key
    assert model.process(**big)["pruned_context"] == want["pruned_context"]  # the model is a plain model again
    # the same through the environment: process() itself keeps the front-end
    monkeypatch.setenv("OPEN_PROVENCE_HOST_REPLICAS", "3")
    try:
        routed = model.process(**big)
        assert model.__dict__["_host_front_end"].last_trace["rows"] > 0
    finally:
        model.__dict__["_host_front_end"].close()
    for key in want:
        if key not in ("timing", "performance_trace"):
            assert routed[key] == want[key], key


def test_process_starts_host_replicas_by_itself(monkeypatch):
    """Round 5: no environment variable needed.  ``preprocess_workers=N`` on a request of >= 256 contexts, and ANY request of
    >= 2000 contexts (the reference's auto rule, standalone.py:2588-2596), run their host stages on replicas that do not
    import the caller's ``__main__``; the result equals the in-process call's field for field, ``timing`` says how it ran."""

    from open_provence_amd import frontend

    monkeypatch.delenv("OPEN_PROVENCE_HOST_REPLICAS", raising=False)
    monkeypatch.setattr(frontend, "default_host_workers", lambda: 3)  # (the auto count of the box would start 31 processes)
    model, _meta = _g3_model()
    words = "the tower is tall boats carry fish and salt to north city harbour many years ago it was new".split()
    contexts = [
        " ".join(" ".join(words[(i * 5 + s * 3 + k) % len(words)] for k in range(4 + (i + s) % 5)).capitalize() + "." for s in range(1 + i % 6))
        for i in range(2100)
    ]
    call = dict(question="which boats carry salt?", sentence_splitter=period_splitter, show_progress=False,
                return_sentence_metrics=True, return_sentence_texts=True, threshold=0.4)
    monkeypatch.setenv("OPEN_PROVENCE_HOST_REPLICAS", "0")
    want_small, want_big = model.process(context=contexts[:300], **call), model.process(context=contexts, **call)
    assert "host_replicas" not in want_big["timing"] and want_big["performance_trace"].runtime["kernel_set"] == "f16-f8-w"
    assert all(isinstance(v, (int, float)) for v in want_big["timing"].values())  # callers sum / float() these
    monkeypatch.delenv("OPEN_PROVENCE_HOST_REPLICAS")
    try:
        got_small = model.process(context=contexts[:300], preprocess_workers=2, **call)   # asked for: two worker processes
        assert got_small["timing"]["host_replicas"] == 2 and got_small["timing"]["fallback_from_f8"] == 0
        assert model.process(context=contexts[:100], preprocess_workers=2, **call)["timing"].get("host_replicas") is None  # threads
        got_big = model.process(context=contexts, **call)                                  # by size alone
        assert got_big["timing"]["host_replicas"] == 3
        assert got_big["performance_trace"].as_dict() == got_big["timing"]
    finally:
        front = model.__dict__.get("_host_front_end")
        if front is not None:
            front.close()
    for got, want in ((got_small, want_small), (got_big, want_big)):
        for key in want:
            if key not in ("timing", "performance_trace"):
                assert got[key] == want[key], key


def test_padded_forward_boundary():
    from open_provence_amd.config import OpenProvenceConfig
    from open_provence_amd.modeling import OpenProvenceForTokenClassification, OpenProvenceModel

    arrays, meta = load_golden("g0c_hd64_synth")
    cfg = OpenProvenceConfig(
        base_model_config=meta["base_model_config"], tokenizer_name_or_path="x",
        pruning_config={"hidden_size": meta["base_model_config"]["hidden_size"]}, max_length=128,
    )
    state = state_from_fixture(arrays, meta)
    model = OpenProvenceModel(cfg, device="cuda", tokenizer=CharTokenizer(), state_dict=state)
    ids = torch.from_numpy(arrays["input_ids"])
    mask = torch.from_numpy(arrays["attention_mask"])
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids))
    assert out.logits is out.ranking_logits and out["pruning_logits"].shape == (ids.shape[0], ids.shape[1], 2)
    prune = out.pruning_logits.cpu().numpy()
    m = mask.bool().numpy()
    assert np.abs(prune - arrays["pruning_logits"])[m].max() < 1e-3
    assert np.all(prune[~m] == 0.0)
    assert np.abs(out.ranking_logits.cpu().numpy() - arrays["ranking_logits"]).max() < 1e-3
    rank_t, prune_t = model(input_ids=ids.cuda(), attention_mask=mask.cuda(), return_dict=False)
    assert torch.equal(rank_t, out.ranking_logits) and torch.equal(prune_t, out.pruning_logits)
    with pytest.raises(ValueError):
        model(input_ids=None)
    with pytest.raises(NotImplementedError):  # left padding is not what process() produces
        model(input_ids=ids, attention_mask=torch.flip(mask, dims=[1]))

    tok_model = OpenProvenceForTokenClassification(cfg, device="cuda", tokenizer=CharTokenizer(), state_dict=state)
    tout = tok_model(input_ids=ids, attention_mask=mask)
    assert tout.logits.shape[-1] == 2 and torch.equal(tout.logits, out.pruning_logits)
    assert torch.equal(tout.ranking_logits, out.ranking_logits)


def test_from_pretrained_roundtrip_including_legacy_keys_and_bf16_weights(tmp_path):
    from safetensors.torch import save_file

    from open_provence_amd.config import OpenProvenceConfig
    from open_provence_amd.modeling import OpenProvenceModel

    arrays, meta = load_golden("g0c_hd64_synth")
    state = state_from_fixture(arrays, meta)
    cfg = OpenProvenceConfig(
        base_model_config=meta["base_model_config"], tokenizer_name_or_path="x",
        pruning_config={"hidden_size": 128}, max_length=128, default_threadshold=0.2,
        pruning_hidden_state="post_final_norm",  # pinned: an unstamped config.json would otherwise resolve to the 4.x convention
    )
    rows = rows_from_fixture(arrays)
    ref_model = OpenProvenceModel(cfg, device="cuda", tokenizer=CharTokenizer(), state_dict=state)
    ref_prune, ref_rank, _ = ref_model.encoder.forward_rows(rows)

    # modern layout
    d1 = tmp_path / "ckpt"
    d1.mkdir()
    cfg.save_json(d1 / "config.json")
    save_file({k: v.contiguous() for k, v in state.items()}, str(d1 / "model.safetensors"))
    m1 = OpenProvenceModel.from_pretrained(d1, device="cuda", tokenizer=CharTokenizer(), max_length=64)
    assert m1.max_length == 64 and m1.default_threshold == pytest.approx(0.2)
    p1, r1, _ = m1.encoder.forward_rows(rows)
    assert torch.equal(p1, ref_prune) and torch.equal(r1, ref_rank)

    # legacy layout: no "ranking_model." prefix (standalone.py:1452-1464)
    d2 = tmp_path / "legacy"
    d2.mkdir()
    cfg.save_json(d2 / "config.json")
    legacy = {(k[len("ranking_model."):] if k.startswith("ranking_model.") else k): v.contiguous() for k, v in state.items()}
    save_file(legacy, str(d2 / "model.safetensors"))
    m2 = OpenProvenceModel.from_pretrained(d2, device="cuda", tokenizer=CharTokenizer())
    p2, _, _ = m2.encoder.forward_rows(rows)
    assert torch.equal(p2, ref_prune)

    # bf16 checkpoint tensors are widened on load (not bit-identical to fp32 weights, but close)
    d3 = tmp_path / "bf16"
    d3.mkdir()
    cfg.save_json(d3 / "config.json")
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in state.items()}, str(d3 / "model.safetensors"))
    m3 = OpenProvenceModel.from_pretrained(d3, device="cuda", tokenizer=CharTokenizer())
    p3, _, _ = m3.encoder.forward_rows(rows)
    assert torch.isfinite(p3).all() and (p3 - ref_prune).abs().max() < 0.3

    # writer -> reader: save_pretrained of a loaded model (legacy keys come back prefixed) reloads bit-identically
    d4 = tmp_path / "resaved"
    m2.save_pretrained(d4)
    assert set(m2.state_dict()) == set(state)
    m4 = OpenProvenceModel.from_pretrained(d4, device="cuda", tokenizer=CharTokenizer())
    p4, r4, _ = m4.encoder.forward_rows(rows)
    assert torch.equal(p4, ref_prune) and torch.equal(r4, ref_rank)
    assert m4.default_threshold == pytest.approx(0.2)

    with pytest.raises(FileNotFoundError):
        OpenProvenceModel.from_pretrained(tmp_path / "missing", device="cuda")


def test_missing_weight_is_reported():
    from open_provence_amd import _lib
    from open_provence_amd.engine import HipEncoder

    arrays, meta = load_golden("g0c_hd64_synth")
    state = state_from_fixture(arrays, meta)
    state.pop("ranking_model.model.layers.2.mlp.Wo.weight")
    enc = HipEncoder(dims_from_meta(meta), device="cuda")
    with pytest.raises(_lib.HipLibraryError, match="layers.2.mlp.Wo.weight"):
        enc.load_state_dict(state)
    with pytest.raises(_lib.HipLibraryError, match="shape"):
        enc.load_weight("ranking_model.model.final_norm.weight", torch.zeros(7))


def test_single_block_api():
    model, _ = _g3_model()
    contexts = ["First part. ", "Second part is here. ", "Third."]
    raw = model.get_raw_predictions("a query?", contexts)
    # the reference counts the specials of the encoded prefix ([CLS] + 9 chars + [SEP] = 11): standalone.py:1955-1969
    assert raw.context_ranges == [(11, 23), (23, 44), (44, 50)]
    assert raw.pruning_probs.dtype == np.float32 and len(raw.pruning_probs) == 50
    assert 0.0 < raw.ranking_score < 1.0
    res = model.predict_with_thresholds("a query?", contexts, [0.1, 0.9])
    assert set(res["predictions"]) == {0.1, 0.9} and all(len(v) == 3 for v in res["predictions"].values())
    batch = model.get_raw_predictions_batch(["q1?", "q2?"], [contexts, contexts[:1]], batch_size=1)
    assert len(batch) == 2 and len(batch[1].context_ranges) == 1
    with pytest.raises(ValueError):
        model.get_raw_predictions_batch(["q1?"], [contexts, contexts])


def test_edge_cases_empty_single_token_and_long_sequence():
    from open_provence_amd.engine import HipEncoder
    from oracle.modernbert_oracle import oracle_forward

    arrays, meta = load_golden("g0c_hd64_synth")
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    enc = HipEncoder(dims, device="cuda")
    enc.load_state_dict(state)
    prune, rank, _ = enc.forward_rows([])
    assert prune.shape == (0, 2) and rank.shape == (0, 1)
    rng = np.random.default_rng(5)
    rows = [[1], [1, 7, 2], rng.integers(3, 250, size=2050).tolist(), [], [9, 9]]
    prune, rank, cu = enc.forward_rows(rows)
    assert prune.shape == (2056, 2) and torch.isfinite(prune).all() and torch.isfinite(rank).all()
    assert rank[3].abs().max() == 0  # empty row: defined as zeros
    for i in (0, 1, 2, 4):
        ids = torch.tensor([rows[i]], dtype=torch.long)
        ref = oracle_forward(state, dims, ids, torch.ones_like(ids))
        got = prune[cu[i] : cu[i + 1]].cpu()
        assert (got - ref.pruning_logits[0]).abs().max() < 1e-3, i
        assert (rank[i].cpu() - ref.ranking_logits[0]).abs().max() < 1e-3, i


def test_panel_path_ragged_edge_cases():
    """hidden % 256 == 0 models (base / large / en-gte shapes) take the k-streamed panel kernels: a 3-layer H=512
    model (global + two local layers) on a ragged batch with an empty row, one-token rows, lengths that are not
    multiples of anything and one row longer than a 256-query attention block, against the oracle."""

    from open_provence_amd.config import EncoderDims
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import synth_state_dict
    from oracle.modernbert_oracle import oracle_forward

    dims = EncoderDims.from_base_model_config(
        dict(model_type="modernbert", vocab_size=512, hidden_size=512, intermediate_size=384, num_hidden_layers=3,
             num_attention_heads=8, local_attention=128, global_attn_every_n_layers=3, global_rope_theta=160000.0,
             local_rope_theta=10000.0, max_position_embeddings=1024, pad_token_id=0, cls_token_id=1, sep_token_id=2),
        num_labels=1,
    )
    state = synth_state_dict(dims, 11)
    enc = HipEncoder(dims, device="cuda")
    enc.load_state_dict(state)
    rng = np.random.default_rng(9)
    lengths = [1, 0, 33, 257, 700, 1, 130]
    rows = [rng.integers(3, 500, size=n).tolist() for n in lengths]
    prune, rank, cu = enc.forward_rows(rows)
    assert prune.shape == (sum(lengths), 2) and torch.isfinite(prune).all() and torch.isfinite(rank).all()
    assert rank[1].abs().max() == 0
    for i, n in enumerate(lengths):
        if n == 0:
            continue
        ids = torch.tensor([rows[i]], dtype=torch.long)
        ref = oracle_forward(state, dims, ids, torch.ones_like(ids))
        got = prune[cu[i] : cu[i + 1]].cpu()
        assert (got - ref.pruning_logits[0]).abs().max() < 1e-3, (i, n)
        assert (rank[i].cpu() - ref.ranking_logits[0]).abs().max() < 1e-3, (i, n)


@pytest.mark.parametrize("model_name,lengths", [("base", [512, 300, 77]), ("large", [512, 129]), ("large", [2048]),
                                                ("en-gte", [640, 64])])
def test_full_depth_published_shapes_match_oracle(model_name, lengths):
    """BASELINE.json configs 3-5 at their real depth (19 / 25 / 22 layers, H = 512 / 768): error accumulation through
    the whole stack of the panel kernels stays inside 1e-3 of the fp32 oracle (the H=768 golden fixture has 3 layers)."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict
    from oracle.modernbert_oracle import oracle_forward

    dims = named_dims(model_name, vocab_size=4096)
    state = synth_state_dict(dims, 17)
    rows = [synth_pair_batch(dims, 1, n, seed=100 + n)[0] for n in lengths]
    enc = HipEncoder(dims, device="cuda")
    enc.load_state_dict(state)
    prune, rank, cu = enc.forward_rows(rows)
    for i, row in enumerate(rows):
        ids = torch.tensor([row], dtype=torch.long)
        ref = oracle_forward(state, dims, ids, torch.ones_like(ids))
        err_p = (prune[cu[i] : cu[i + 1]].cpu() - ref.pruning_logits[0]).abs().max().item()
        err_r = (rank[i].cpu() - ref.ranking_logits[0]).abs().max().item()
        assert err_p < 1e-3 and err_r < 1e-3, (model_name, lengths[i], err_p, err_r)


SHAPE_SWEEP = [
    # hidden (multiple of 128, head_dim 64), intermediate (multiple of 64), heads, layers, pooling, labels, GEMM family
    (128, 64, 2, 2, "cls", 1, "row"),
    (128, 192, 2, 3, "mean", 2, "row"),
    (256, 320, 4, 2, "cls", 3, "row"),
    (512, 128, 8, 2, "mean", 1, "panel"),
    (768, 384, 12, 1, "cls", 2, "panel"),
    (1024, 256, 16, 1, "cls", 1, "panel"),
    (384, 192, 6, 2, "cls", 1, "tiled"),
    (640, 320, 10, 1, "mean", 1, "tiled"),
    (512, 192, 8, 1, "cls", 1, "tiled"),  # intermediate not a multiple of 128 -> falls back to the tiles
]


@pytest.mark.parametrize("hidden,inter,heads,layers,pooling,labels,path", SHAPE_SWEEP)
def test_shape_sweep_against_oracle(hidden, inter, heads, layers, pooling, labels, path):
    """Every GEMM family (row-stationary, panel, generic tiles) on shapes other than the published ones: odd chunk
    counts, one head, mean pooling, several ranking labels, ragged rows around the 16 / 32 / 64 / 128 boundaries."""

    from open_provence_amd.config import EncoderDims
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import synth_state_dict
    from oracle.modernbert_oracle import oracle_forward

    dims = EncoderDims.from_base_model_config(
        dict(model_type="modernbert", vocab_size=300, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
             num_attention_heads=heads, local_attention=128, global_attn_every_n_layers=2, global_rope_theta=160000.0,
             local_rope_theta=10000.0, max_position_embeddings=512, pad_token_id=0, cls_token_id=1, sep_token_id=2,
             classifier_pooling=pooling),
        num_labels=labels,
    )
    state = synth_state_dict(dims, 29)
    enc = HipEncoder(dims, device="cuda")
    enc.load_state_dict(state)
    rng = np.random.default_rng(hidden + inter)
    lengths = [1, 15, 16, 17, 31, 33, 63, 65, 127, 129, 200, 300]
    rows = [rng.integers(3, 299, size=n).tolist() for n in lengths]
    prune, rank, cu = enc.forward_rows(rows)
    assert prune.shape == (sum(lengths), 2) and rank.shape == (len(rows), labels)
    worst = 0.0
    for i, row in enumerate(rows):
        ids = torch.tensor([row], dtype=torch.long)
        ref = oracle_forward(state, dims, ids, torch.ones_like(ids))
        worst = max(worst, (prune[cu[i] : cu[i + 1]].cpu() - ref.pruning_logits[0]).abs().max().item(),
                    (rank[i].cpu() - ref.ranking_logits[0]).abs().max().item())
    assert worst < 1e-3, (path, worst)


@pytest.mark.parametrize("hidden,inter,heads", [(128, 64, 2), (512, 128, 8)])
def test_maximum_length_sequences(hidden, inter, heads):
    """max_position_embeddings-long rows (8192 tokens: 32 attention blocks, 128 key tiles per query block) next to a
    short one, on the row-stationary and the panel path."""

    from open_provence_amd.config import EncoderDims
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import synth_state_dict
    from oracle.modernbert_oracle import oracle_forward

    dims = EncoderDims.from_base_model_config(
        dict(model_type="modernbert", vocab_size=300, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=3,
             num_attention_heads=heads, local_attention=128, global_attn_every_n_layers=3, global_rope_theta=160000.0,
             local_rope_theta=10000.0, max_position_embeddings=8192, pad_token_id=0, cls_token_id=1, sep_token_id=2),
        num_labels=1,
    )
    state = synth_state_dict(dims, 5)
    enc = HipEncoder(dims, device="cuda")
    enc.load_state_dict(state)
    rng = np.random.default_rng(1)
    rows = [rng.integers(3, 299, size=n).tolist() for n in (8192, 257)]
    prune, rank, cu = enc.forward_rows(rows)
    for i, row in enumerate(rows):
        ids = torch.tensor([row], dtype=torch.long)
        ref = oracle_forward(state, dims, ids, torch.ones_like(ids))
        assert (prune[cu[i] : cu[i + 1]].cpu() - ref.pruning_logits[0]).abs().max() < 1e-3
        assert (rank[i].cpu() - ref.ranking_logits[0]).abs().max() < 1e-3
    with pytest.raises(Exception):  # longer than max_position_embeddings: refused, not truncated
        enc.forward_rows([rng.integers(3, 299, size=8193).tolist()])


def test_batch_composition_invariance_at_baseline_size():
    """C2 size (256 pairs x 512 tokens, xsmall dims): every pair's outputs are bit-identical whatever the
    batch order, the companions in the batch or the chunking -- pairs are independent."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict

    dims = named_dims("xsmall", vocab_size=8192)
    state = synth_state_dict(dims, 3)
    rows = synth_pair_batch(dims, 256, 512)
    enc = HipEncoder(dims, device="cuda")
    enc.load_state_dict(state)
    prune, rank, cu = enc.forward_rows(rows)
    assert torch.isfinite(prune).all() and torch.isfinite(rank).all()
    perm = np.random.default_rng(0).permutation(256)
    prune_p, rank_p, cu_p = enc.forward_rows([rows[i] for i in perm])
    assert torch.equal(rank_p, rank[torch.from_numpy(perm).cuda()])
    for j in (0, 17, 255):
        i = int(perm[j])
        assert torch.equal(prune_p[cu_p[j] : cu_p[j + 1]], prune[cu[i] : cu[i + 1]])
    small = HipEncoder(dims, device="cuda", chunk_rows=4096)
    small.load_state_dict(state)
    prune_s, rank_s, _ = small.forward_rows(rows[:40])
    assert torch.equal(rank_s, rank[:40]) and torch.equal(prune_s, prune[: cu[40]])
    # spot-check three pairs against the CPU oracle at full length
    from oracle.modernbert_oracle import oracle_forward

    for i in (0, 100, 255):
        ids = torch.tensor([rows[i]], dtype=torch.long)
        ref = oracle_forward(state, dims, ids, torch.ones_like(ids))
        assert (prune[cu[i] : cu[i + 1]].cpu() - ref.pruning_logits[0]).abs().max() < 1e-3
        assert (rank[i].cpu() - ref.ranking_logits[0]).abs().max() < 1e-3


def _model_from_meta(meta, precision="bf16x3"):
    from open_provence_amd.config import OpenProvenceConfig
    from open_provence_amd.modeling import OpenProvenceModel
    from open_provence_amd.synthetic import synth_state_dict

    cfg = OpenProvenceConfig(
        base_model_config=meta["base_model_config"], tokenizer_name_or_path="char-tokenizer",
        pruning_config={"hidden_size": meta["base_model_config"]["hidden_size"]}, max_length=meta["max_length"], num_labels=1,
    )
    state = synth_state_dict(dims_from_meta(meta), meta["weight_seed"])
    return OpenProvenceModel(cfg, device="cuda:0", tokenizer=CharTokenizer(), state_dict=state, precision=precision)


def _same_records(got, exp, path="", tol=1e-3):
    """Integers, strings and structure exact; floats (scores, probabilities, compression) within `tol`."""

    import math

    if isinstance(exp, float) or isinstance(got, float):
        if exp is None or got is None:
            assert got is exp, path
        elif isinstance(exp, float) and math.isnan(exp):
            assert math.isnan(got), path
        else:
            assert math.isclose(float(got), float(exp), rel_tol=0.0, abs_tol=tol), (path, got, exp)
    elif isinstance(exp, dict):
        assert isinstance(got, dict) and set(got) == set(exp), (path, sorted(got), sorted(exp))
        for key in exp:
            _same_records(got[key], exp[key], f"{path}.{key}", tol)
    elif isinstance(exp, (list, tuple)):
        assert isinstance(got, (list, tuple)) and len(got) == len(exp), (path, got, exp)
        for i, (g, e) in enumerate(zip(got, exp)):
            _same_records(g, e, f"{path}[{i}]", tol)
    else:
        assert got == exp, (path, got, exp)


def test_mldr_records_large_model_max_length_2048_match_reference():
    """BASELINE.json configs[3] on the GPU: the MLDR caller (reference scripts/eval_mldr.py build_records) over process()
    with explicit per-passage titles and multi-block documents at max_length 2048, large dims (H=768, I=3072, 12 heads)
    at FULL depth (25 layers) -- golden produced by the reference's own script over its REAL forward
    (tests/golden/g9_large_mldr.json).  Records identical (texts, ids, kept/removed structure), floats within 1e-3."""

    import inspect

    from open_provence_amd.eval_harness import build_mldr_records

    meta = json.loads((GOLDEN_DIR / "g9_large_mldr.json").read_text(encoding="utf-8"))
    assert meta["max_length"] == 2048 and meta["base_model_config"]["num_hidden_layers"] == 25
    model = _model_from_meta(meta)

    def process_fn(**kwargs):
        return model.process(sentence_splitter=period_splitter, **kwargs)

    process_fn.__signature__ = inspect.signature(model.process)
    for run in meta["runs"]:
        records, stats, n_queries = build_mldr_records(
            process_fn, meta["rows"], threshold=meta["threshold"], batch_size=4, log_timing=False,
            use_best_reranker_score=run["use_best_reranker_score"], show_progress=False,
        )
        exp = run["expected"]
        assert n_queries == exp["n_queries"]
        _same_records(records, exp["records"], run["label"] + ".records")
        _same_records(stats, exp["stats"], run["label"] + ".stats", tol=2e-3)  # compression is in percent


def test_evaluate_dataset_through_hip_model_matches_reference():
    """The dataset evaluator (reference scripts/eval_datasets.py:247-486) over the HIP model: confusion matrix, kept
    spans and predictions identical to the reference's run over its REAL forward, floats within 1e-3."""

    from open_provence_amd.eval_harness import evaluate_dataset

    meta = json.loads((GOLDEN_DIR / "g4m_eval_dataset_model.json").read_text(encoding="utf-8"))
    model = _model_from_meta(meta)
    for run in meta["runs"]:
        got = evaluate_dataset(model, meta["dataset"], threshold=run["threshold"], batch_size=4, dataset_label="synthetic")
        exp = run["expected"]
        for key in ("span_total", "span_correct", "span_skipped", "contexts", "confusion_matrix"):
            assert got[key] == exp[key], (run["threshold"], key, got[key], exp[key])
        assert got["roc_data"]["labels"] == exp["roc_data"]["labels"]
        assert got["roc_data"]["predictions"] == exp["roc_data"]["predictions"]
        _same_records(got["roc_data"]["scores"], exp["roc_data"]["scores"], "scores")
        for key in ("span_accuracy", "mean_compression", "precision", "recall", "f2"):
            _same_records(got[key], exp[key], key, tol=2e-3)


def test_single_block_api_matches_reference_values():
    """get_raw_predictions(_batch) / predict_with_thresholds against the reference's values (REAL forward, g10)."""

    meta = json.loads((GOLDEN_DIR / "g10_raw_predictions.json").read_text(encoding="utf-8"))
    model = _model_from_meta(meta)
    exp = meta["expected"]
    queries, batch, thresholds = meta["queries"], meta["contexts_batch"], meta["thresholds"]

    def check(pred, ref, label):
        assert [list(r) for r in pred.context_ranges] == ref["context_ranges"], label
        assert abs(float(pred.ranking_score) - ref["ranking_score"]) < 1e-3, label
        probs = np.asarray(pred.pruning_probs, dtype=np.float64)
        want = np.asarray(ref["pruning_probs"], dtype=np.float64)
        # pruning_probs come at the padded batch width, as the reference's (standalone.py:1820-1823); the values are
        # compared over the row's tokens (the reference's entries past them are its model's output on pad tokens)
        assert len(probs) == len(want), label
        n_real = ref["context_ranges"][-1][1]
        assert np.abs(probs[:n_real] - want[:n_real]).max() < 1e-3, label

    for i, pred in enumerate(model.get_raw_predictions_batch(queries[0], batch)):
        check(pred, exp["shared_query"][i], f"shared[{i}]")
    for i, pred in enumerate(model.get_raw_predictions_batch(queries, batch, batch_size=2)):
        check(pred, exp["per_sample_queries"][i], f"per_sample[{i}]")
    check(model.get_raw_predictions(queries[2], batch[2]), exp["single"], "single")
    for key, q, ctx, kwargs in (("predict_with_thresholds_mean", queries[0], batch[0], {}),
                                ("predict_with_thresholds_majority", queries[2], batch[2], {"use_majority": True})):
        got = model.predict_with_thresholds(q, ctx, thresholds, **kwargs)
        ref = exp[key]
        assert {str(t): v for t, v in got["predictions"].items()} == ref["predictions"], key
        assert [list(r) for r in got["context_ranges"]] == ref["context_ranges"] and got["contexts"] == ref["contexts"]
        assert abs(got["ranking_score"] - ref["ranking_score"]) < 1e-3


def test_sharded_forward_over_a_one_rank_nccl_group_is_bit_identical():
    """The N > 1 code path with the real encoder and the real backend (RCCL; gpurun grants one GPU, so world_size 1):
    partition -> HipEncoder.forward_rows -> ShardPlan.gather over NCCL -> reorder, bit-identical to the plain call."""

    import os

    import torch.distributed as dist

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.sharding import sharded_forward

    arrays, meta = load_golden("g1_xsmall")
    dims = dims_from_meta(meta)
    enc = HipEncoder(dims, device="cuda:0")
    enc.load_state_dict(state_from_fixture(arrays, meta))
    rows = rows_from_fixture(arrays)
    prune, rank, cu = enc.forward_rows(rows)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        per_row, all_rank = sharded_forward(rows, enc.forward_rows, dst=0)
        torch.cuda.synchronize()
        assert torch.equal(all_rank, rank)
        for i in range(len(rows)):
            assert torch.equal(per_row[i], prune[cu[i] : cu[i + 1]]), i
        # process() with the group attached keeps its pipelined path (asynchronous launches, on-device fragment means,
        # a gather of 4 bytes per fragment over RCCL): identical to the single-GPU process(), field for field
        model, meta = _g3_model()
        plain = [model.process(question=c["question"], context=c["context"], sentence_splitter=period_splitter, show_progress=False,
                               return_sentence_metrics=True, return_sentence_texts=True, **c["kwargs"]) for c in meta["cases"]]
        # shard="rows": forward batches divided (ShardPlan.gather of fragment means); shard="jobs" (the default): contexts
        # divided, the per-context results moved by one gather_object -- pickled through tensors on the GPU under NCCL
        for shard in ("rows", "jobs"):
            model.attach_process_group(None, dst=0, single_rank_gather=True, shard=shard)
            assert model._can_pipeline() and model._dist_info() is not None
            for case, want in zip(meta["cases"], plain):
                got = model.process(question=case["question"], context=case["context"], sentence_splitter=period_splitter,
                                    show_progress=False, return_sentence_metrics=True, return_sentence_texts=True, **case["kwargs"])
                for key in ("pruned_context", "reranking_score", "compression_rate", "kept_sentences", "removed_sentences", "sentence_probabilities"):
                    assert got[key] == want[key], (shard, key)
                assert_process_result_matches(got, case["expected"], prob_tol=1e-3, score_tol=1e-3)
            assert model._dist_info() is not None  # the job-sharded call restored the group for row-sharded entry points
    finally:
        dist.destroy_process_group()


def test_automodel_from_pretrained_lands_on_the_hip_class(tmp_path, monkeypatch):
    """README.md:53-74 of the reference: AutoModel.from_pretrained(dir, trust_remote_code=True).  A checkpoint written
    by save_pretrained (auto_map + the remote-code module under the name it points to) loads through transformers'
    AutoModel into the HIP class, and gives the same logits as the instance that wrote it."""

    import transformers.dynamic_module_utils as dmu
    from transformers import AutoModel, AutoModelForTokenClassification

    from open_provence_amd.modeling import OpenProvenceForSequenceClassification, OpenProvenceForTokenClassification

    monkeypatch.setattr(dmu, "HF_MODULES_CACHE", str(tmp_path / "hf_modules"))
    model, _meta = _g3_model()
    model.tokenizer = None  # the char tokenizer has no files; AutoModel users pass / load their own
    model.save_pretrained(tmp_path / "ckpt")
    ids = torch.tensor([[1, 81, 82, 2, 97, 98, 99, 46, 100, 101, 2, 0, 0], [1, 81, 2, 120, 121, 46, 2, 0, 0, 0, 0, 0, 0]])
    mask = (ids != 0).long()
    want = model(input_ids=ids, attention_mask=mask)
    for auto, klass, key in ((AutoModel, OpenProvenceForSequenceClassification, "ranking_logits"),
                             (AutoModelForTokenClassification, OpenProvenceForTokenClassification, "pruning_logits")):
        loaded = auto.from_pretrained(str(tmp_path / "ckpt"), trust_remote_code=True, tokenizer=CharTokenizer(), device="cuda:0")
        assert type(loaded).__name__ == klass.__name__ and type(loaded).__module__ == "open_provence_amd.modeling"
        got = loaded(input_ids=ids, attention_mask=mask)
        assert torch.equal(got.logits, want[key]) and torch.equal(got.ranking_logits, want.ranking_logits)


def test_segment_means_equal_numpy_mean_bit_for_bit():
    """op_segment_means (process(): per-fragment keep-probability means taken on the device) reproduces
    ``float(values[start:end].mean())`` of numpy on float32 -- pairwise summation order, float64 division -- exactly,
    for every range length up to a long row, and scores an empty range 1.0 (standalone.py:3075-3082)."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import named_dims

    enc = HipEncoder(named_dims("xsmall"), device="cuda:0", precision="bf16x3")
    rng = np.random.default_rng(11)
    values = rng.random(20000, dtype=np.float32)
    lengths = list(range(0, 700)) + [1000, 2047, 2048, 4097, 8192]
    segs, start = [], 3
    for n in lengths:
        if start + n > values.shape[0]:
            start = 1
        segs.append((start, start + n))
        start += max(1, n // 3)
    segs += [(5, 5), (9, 4), (19990, 20000), (-3, 6), (19995, 20010)]
    seg_np = np.asarray(segs, dtype=np.int32)
    out = enc.segment_means(torch.from_numpy(values).to("cuda:0"), torch.from_numpy(seg_np).to("cuda:0")).cpu().numpy()
    for (s, e), got in zip(segs, out):
        s, e = max(0, s), min(e, values.shape[0])
        want = 1.0 if e <= s else float(values[s:e].mean())
        assert float(got) == want, (s, e, float(got), want)
    enc.close()


def test_two_launch_sequences_give_the_single_sequence_bits():
    """``forward_packed_on``: the two halves of a batch enqueued on the two CU-partitioned streams (independent launch
    sequences, own workspaces, nothing ordered between them) produce, pair for pair, exactly the outputs of one
    ``forward_packed`` over the whole batch -- also when both pipelines are re-used back to back."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.packing import pack_rows
    from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict

    dims = named_dims("xsmall")
    state = synth_state_dict(dims, seed=7)
    state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.ndim == 2 and "embeddings" not in k else v) for k, v in state.items()}
    enc = HipEncoder(dims, device="cuda:0", precision="bf16x3")
    enc.load_state_dict(state)
    rows = synth_pair_batch(dims, 24, 160, seed=5) + synth_pair_batch(dims, 8, 512, seed=6)
    ids_np, cu_np, max_len = pack_rows(rows)
    dev = torch.device("cuda:0")
    keep = torch.empty(int(cu_np[-1]), dtype=torch.float32, device=dev)
    prune, rank = enc.forward_packed(torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev), cu_np, max_len, keep_prob=keep)
    torch.cuda.synchronize()
    want = (prune.cpu().numpy(), rank.cpu().numpy(), keep.cpu().numpy())
    half = len(rows) // 2
    for _ in range(2):  # second round: workspaces and streams re-used
        got_p, got_r, got_k = [], [], []
        outs = []
        for part, part_rows in enumerate((rows[:half], rows[half:])):
            p_ids, p_cu, p_max = pack_rows(part_rows)
            p_keep = torch.empty(int(p_cu[-1]), dtype=torch.float32, device=dev)
            ids_d, cu_d = torch.from_numpy(p_ids).to(dev), torch.from_numpy(p_cu).to(dev)
            torch.cuda.synchronize()  # inputs resident before the pipeline's stream touches them
            outs.append((enc.forward_packed_on(part, ids_d, cu_d, p_cu, p_max, keep_prob=p_keep), p_keep, ids_d, cu_d))
        torch.cuda.synchronize()
        for (p, r), k, _, _ in outs:
            got_p.append(p.cpu().numpy()); got_r.append(r.cpu().numpy()); got_k.append(k.cpu().numpy())
        assert np.array_equal(np.concatenate(got_p), want[0])
        assert np.array_equal(np.concatenate(got_r), want[1])
        assert np.array_equal(np.concatenate(got_k), want[2])
    enc.close()
