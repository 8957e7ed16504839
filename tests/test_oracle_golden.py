"""The CPU oracle (oracle/modernbert_oracle.py) must reproduce the outputs of the REAL reference
(OpenProvenceModel.forward + HF ModernBERT) stored under tests/golden/ by tests/golden/make_golden.py.
Tolerance 5e-5 on logits = fp32 re-association noise (the reference itself is ~2e-5 from an fp64 run)."""

import numpy as np
import pytest
import torch

from helpers import dims_from_meta, load_golden, state_from_fixture
from oracle.modernbert_oracle import keep_probabilities, oracle_forward

FORWARD_FIXTURES = ["g0_tiny_hd16", "g0b_hd64_refinit", "g0c_hd64_synth", "g1m_meanpool", "g1_xsmall",
                    "g7_xsmall_refinit", "g8_base_refinit", "g12_prenorm_tf4"]
TOL = 5e-5


@pytest.mark.parametrize("name", FORWARD_FIXTURES)
def test_oracle_matches_reference_outputs(name):
    arrays, meta = load_golden(name)
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    ids = torch.from_numpy(arrays["input_ids"])
    mask = torch.from_numpy(arrays["attention_mask"])
    pre_norm = bool(meta.get("prune_pre_final_norm", False))  # transformers-4.x hidden_states[-1] (see make_golden.py)
    out = oracle_forward(state, dims, ids, mask, return_hidden=True, prune_pre_final_norm=pre_norm)
    m = mask.bool().numpy()
    assert np.abs(out.ranking_logits.numpy() - arrays["ranking_logits"]).max() < TOL
    assert np.abs(out.pruning_logits.numpy() - arrays["pruning_logits"])[m].max() < TOL
    assert len(out.hidden_states) == meta["n_hidden_states"] == dims.num_layers + 1
    if pre_norm:  # the two conventions really differ on this fixture
        other = oracle_forward(state, dims, ids, mask, prune_pre_final_norm=False)
        assert np.abs(other.pruning_logits.numpy() - arrays["pruning_logits"])[m].max() > 0.1
    stride = meta["hidden_stride"]
    if stride:
        for i, h in enumerate(out.hidden_states):
            ref = arrays[f"hidden_{i}"]
            err = np.abs(h[:, ::stride].numpy() - ref)[m[:, ::stride]].max()
            assert err < 1e-4, (i, err)


def test_oracle_sdpa_equals_eager():
    arrays, meta = load_golden("g0c_hd64_synth")
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    ids = torch.from_numpy(arrays["input_ids"])
    mask = torch.from_numpy(arrays["attention_mask"])
    a = oracle_forward(state, dims, ids, mask, attn="eager")
    b = oracle_forward(state, dims, ids, mask, attn="sdpa")
    m = mask.bool()
    assert (a.pruning_logits - b.pruning_logits)[m].abs().max() < 2e-5
    assert (a.ranking_logits - b.ranking_logits).abs().max() < 2e-5


def test_oracle_padding_invariance():
    """Right-padding a batch further must not change the logits at real positions (mask semantics)."""
    arrays, meta = load_golden("g0c_hd64_synth")
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    ids = torch.from_numpy(arrays["input_ids"])
    mask = torch.from_numpy(arrays["attention_mask"])
    wide_ids = torch.cat([ids, torch.full((ids.shape[0], 40), 5, dtype=ids.dtype)], dim=1)
    wide_mask = torch.cat([mask, torch.zeros((ids.shape[0], 40), dtype=mask.dtype)], dim=1)
    a = oracle_forward(state, dims, ids, mask)
    b = oracle_forward(state, dims, wide_ids, wide_mask)
    m = mask.bool()
    assert (a.pruning_logits - b.pruning_logits[:, : ids.shape[1]])[m].abs().max() < 2e-5
    assert (keep_probabilities(a.pruning_logits) - keep_probabilities(b.pruning_logits[:, : ids.shape[1]]))[m].abs().max() < 1e-5


def test_g2_varlen_fixture_against_oracle():
    arrays, meta = load_golden("g2_gte_varlen")
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    # run only the three shortest rows to keep the CPU suite fast; rows are independent
    lengths = np.asarray(meta["lengths"])
    pick = np.argsort(lengths)[:3]
    width = int(lengths[pick].max())
    ids = torch.from_numpy(arrays["input_ids"][pick, :width])
    mask = torch.from_numpy(arrays["attention_mask"][pick, :width])
    out = oracle_forward(state, dims, ids, mask)
    m = mask.bool().numpy()
    assert np.abs(out.pruning_logits.numpy() - arrays["pruning_logits"][pick, :width])[m].max() < TOL
    assert np.abs(out.ranking_logits.numpy() - arrays["ranking_logits"][pick]).max() < TOL


def test_oracle_cache_entry_is_what_the_oracle_computes():
    """tests/golden/oracle_cache/ holds the oracle's outputs for the full-size GPU comparisons (tests/oracle_cache.py: 55 % of
    the GPU suite's time went into recomputing them in every run).  One entry end to end on the CPU: its fingerprint is that of
    the weights and rows the GPU test builds, and rows of it -- first, last, one inside -- are what the oracle computes now."""

    import numpy as np
    import torch

    from oracle_cache import CACHE_DIR, fingerprint
    from open_provence_amd.synthetic import named_dims, pad_rows, refinit_state_dict, synth_pair_batch
    from oracle.modernbert_oracle import oracle_forward

    path = CACHE_DIR / "calibration_xsmall_refinit_fp32_256x512.npz"
    assert path.exists(), "run the GPU suite once with OPEN_PROVENCE_WRITE_ORACLE_CACHE=<dir> and copy <dir>/*.npz to tests/golden/oracle_cache/"
    dims = named_dims("xsmall")
    state = refinit_state_dict(dims, seed=7)
    rows = synth_pair_batch(dims, 256, 512, seed=1234)
    with np.load(path) as data:
        assert str(data["fingerprint"]) == fingerprint(state, rows)
        prune, rank = data["a0"], data["a1"]
    assert prune.shape == (256, 512, 2) and rank.shape == (256, dims.num_labels)
    pick = [0, 131, 255]
    ids, mask = pad_rows([rows[i] for i in pick])
    with torch.no_grad():
        ref = oracle_forward(state, dims, ids, mask, attn="sdpa")
    assert np.abs(ref.pruning_logits.numpy() - prune[pick]).max() < 2e-5  # (batch of 3 here, batches of 32 there: fp32 summation order)
    assert np.abs(ref.ranking_logits.numpy() - rank[pick]).max() < 2e-5
