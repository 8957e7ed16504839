"""Precision policies through the C ABI (GPU).

* A policy evaluated on its curated kernel set (exactly the MFMA passes it needs) is BIT-IDENTICAL to the same
  policy evaluated on the all-terms kernels with the unused lo operands cleared -- the identity the precision sweep
  (scripts/precision_sweep.py) relies on.
* A bf16 checkpoint drops the hi x lo(weight) terms automatically without changing a bit of the result.
"""

import numpy as np
import pytest
import torch

from helpers import CharTokenizer, dims_from_meta, load_golden, period_splitter, rows_from_fixture, state_from_fixture
from parity_utils import run_fixture_on_gpu

pytestmark = pytest.mark.gpu

NO_POLICY_KERNELS = 16  # OP_FLAG_NO_POLICY_KERNELS
NO_LAYER_FUSION = 32    # OP_FLAG_NO_LAYER_FUSION: two fused kernels per layer (the shape the all-terms set has too)
LAYER_M32 = 128         # OP_FLAG_LAYER_M32: the whole-layer kernel on 32x32x16 MFMAs (hidden = 256)
NO_F8 = 512             # OP_FLAG_NO_F8: keep the (hi, lo) bf16 whole-layer kernel (kernel set "bf16-weights")
PANEL_F8 = 2048         # OP_FLAG_PANEL_F8: the fp16 + e4m3 kernel sets on the panel path (hidden 512 / 768) too -- opt-in
PANEL_F8_WI = 4096      # OP_FLAG_PANEL_F8_WI: that format in the Wi GEMM alone (default for fp32-valued weights)


@pytest.mark.parametrize("fixture", ["g1_xsmall", "g2_gte_varlen"])
@pytest.mark.parametrize("precision", ["bf16x2", "bf16"])
def test_curated_kernel_set_equals_cleared_operands(fixture, precision):
    fast = run_fixture_on_gpu(fixture, precision, capture=False, return_outputs=True, flags=NO_LAYER_FUSION, calibrate=False)
    slow = run_fixture_on_gpu(fixture, precision, capture=False, return_outputs=True, flags=NO_POLICY_KERNELS, calibrate=False)
    assert fast["kernel_set"] in ("bf16-weights", "bf16")
    assert slow["kernel_set"].startswith("all-terms")
    assert fast["terms"] == slow["terms"]
    assert np.array_equal(fast["prune"], slow["prune"])
    assert np.array_equal(fast["rank"], slow["rank"])


def test_arbitrary_policy_runs_on_cleared_operands():
    rep = run_fixture_on_gpu("g1_xsmall", {"pv": 2, "wi": 1}, capture=False)
    assert rep["kernel_set"].startswith("all-terms")
    assert rep["terms"]["pv"] == 2 and rep["terms"]["wi"] == 1 and rep["terms"]["qk"] == 3
    assert rep["finite"]


@pytest.mark.parametrize("fixture", ["g1_xsmall", "g2_gte_varlen"])
def test_bf16_checkpoint_drops_weight_lo_terms_bit_identically(fixture):
    """Weights rounded to bf16 (what `torch_dtype=torch.bfloat16` loads, standalone.py:219-233): the default bf16x3
    request resolves to the two-pass weight GEMMs, and the result equals the three-pass evaluation exactly."""

    from open_provence_amd.engine import HipEncoder

    arrays, meta = load_golden(fixture)
    dims = dims_from_meta(meta)
    state = {k: v.to(torch.bfloat16) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v
             for k, v in state_from_fixture(arrays, meta).items()}
    rows = rows_from_fixture(arrays)
    outs = []
    for flags in (NO_LAYER_FUSION, NO_POLICY_KERNELS):
        enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flags)
        enc.load_state_dict(state, calibrate=False)
        policy = enc.effective_policy()
        assert policy["terms"] == {"wqkv": 1, "qk": 3, "pv": 3, "attn_out": 1, "wi": 1, "mlp_out": 1}
        assert policy["kernel_set"] == ("bf16-weights" if flags == NO_LAYER_FUSION else "all-terms kernels, cleared lo operands")
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        outs.append((prune.cpu().numpy(), rank.cpu().numpy()))
        enc.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("fixture", ["g1_xsmall", "g0c_hd64_synth", "g7_xsmall_refinit"])
def test_whole_layer_kernel_matches_two_kernel_path(fixture):
    """hidden <= 256 with single-plane weights runs a layer as ONE kernel (attention output projection + MLP with h kept
    on chip + next q/k/v projection).  Same terms as the two-kernel path, different accumulation order (the MLP output is
    accumulated onto the residual instead of being added at the end): equal to fp32 noise on a bf16 checkpoint, and
    within the 1e-3 bar of the oracle evaluated on the same bf16-valued weights."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import pad_rows
    from oracle.modernbert_oracle import oracle_forward

    arrays, meta = load_golden(fixture)
    dims = dims_from_meta(meta)
    state = {k: v.to(torch.bfloat16).to(torch.float32) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v
             for k, v in state_from_fixture(arrays, meta).items()}
    rows = rows_from_fixture(arrays)
    outs = {}
    # default flags: kernel set "f16-f8" (fp16 hi + e4m3 lo operands in the whole-layer kernel: 1.5 MFMA units per
    # product); NO_F8: the same launch on (hi, lo) bf16 operands; NO_LAYER_FUSION: two kernels per layer
    for label, flags, kernel_set in (("f8", 0, "f16-f8"), ("layer", NO_F8, "bf16-weights"), ("two", NO_LAYER_FUSION, "bf16-weights")):
        enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flags)
        enc.load_state_dict(state, calibrate=False)
        assert enc.effective_policy()["kernel_set"] == kernel_set
        enc.profile_enable(True)
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        kinds = set(enc.profile_read())
        assert ("fused_layer_attnout_mlp_qkv" in kinds) == (label != "two"), kinds
        outs[label] = (prune.cpu().numpy(), rank.cpu().numpy())
        enc.close()
    scale = max(1.0, float(np.abs(outs["two"][0]).max()))
    assert np.abs(outs["layer"][0] - outs["two"][0]).max() < 3e-4 * scale
    assert np.abs(outs["layer"][1] - outs["two"][1]).max() < 3e-4 * scale
    ids, mask = pad_rows(rows)
    ref = oracle_forward(state, dims, ids, mask)
    m = mask.bool().numpy()
    for label in ("f8", "layer"):  # tolerance of the path: 1e-3 on logits against the CPU reference arithmetic
        assert np.abs(outs[label][0] - ref.pruning_logits.numpy()[m]).max() < 1e-3, label
        assert np.abs(outs[label][1] - ref.ranking_logits.numpy()).max() < 1e-3, label


@pytest.mark.parametrize("fixture", ["g1_xsmall", "g7_xsmall_refinit", "g1m_meanpool"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_layer32_kernel_matches_the_default_kernel(fixture, precision):
    """The opt-in 32x32x16 form of the whole-layer kernel (opk_layer32.hip.h: other weight packs, other register layout,
    LayerNorm in packed arithmetic) evaluates the same terms: equal to the default kernel to fp32 noise, and (bf16x3 on a
    bf16 checkpoint) within the 1e-3 bar of the oracle."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import pad_rows
    from oracle.modernbert_oracle import oracle_forward

    arrays, meta = load_golden(fixture)
    dims = dims_from_meta(meta)
    assert dims.hidden_size == 256
    state = {k: v.to(torch.bfloat16).to(torch.float32) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v
             for k, v in state_from_fixture(arrays, meta).items()}
    rows = rows_from_fixture(arrays)
    outs = {}
    for label, flags in (("m16", NO_F8), ("m32", LAYER_M32)):
        enc = HipEncoder(dims, device="cuda:0", precision=precision, flags=flags)
        enc.load_state_dict(state, calibrate=False)
        enc.profile_enable(True)
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        assert "fused_layer_attnout_mlp_qkv" in set(enc.profile_read())
        outs[label] = (prune.cpu().numpy(), rank.cpu().numpy())
        enc.close()
    scale = max(1.0, float(np.abs(outs["m16"][0]).max()))
    tol = 3e-4 if precision == "bf16x3" else 5e-2  # single pass: each kernel's own bf16 rounding noise
    assert np.abs(outs["m32"][0] - outs["m16"][0]).max() < tol * scale
    assert np.abs(outs["m32"][1] - outs["m16"][1]).max() < tol * scale
    if precision == "bf16x3":
        ids, mask = pad_rows(rows)
        ref = oracle_forward(state, dims, ids, mask)
        m = mask.bool().numpy()
        assert np.abs(outs["m32"][0] - ref.pruning_logits.numpy()[m]).max() < 1e-3
        assert np.abs(outs["m32"][1] - ref.ranking_logits.numpy()).max() < 1e-3


NO_HEAD_FUSION = 256  # OP_FLAG_NO_HEAD_FUSION: embedding + LayerNorm and final_norm + pruning head as their own launches


@pytest.mark.parametrize("fixture", ["g0b_hd64_refinit", "g0c_hd64_synth", "g1_xsmall", "g7_xsmall_refinit", "g12_prenorm_tf4"])
def test_embedding_and_head_inside_the_first_and_last_kernels(fixture):
    """Without hidden-state capture the layer-0 q / k / v kernel gathers and normalises the embeddings itself, and (CLS
    pooling) the last whole-layer launch ends with final_norm + the pruning head on the rows in its accumulators (no
    write-back of the residual stream, no embed_ln / final_ln_prune launches; the transformers-4.x pre-norm convention
    of g12 included): same logits as the separate launches to fp32 noise carried through the layers, and within the
    bar of the reference outputs."""

    from open_provence_amd.engine import HipEncoder

    arrays, meta = load_golden(fixture)
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    rows = rows_from_fixture(arrays)
    outs = {}
    for label, flags in (("fused", 0), ("separate", NO_HEAD_FUSION)):
        enc = HipEncoder(dims, device="cuda:0", precision="bf16x2", flags=flags,
                         prune_pre_final_norm=bool(meta.get("prune_pre_final_norm", False)))
        enc.load_state_dict(state, calibrate=False)
        enc.profile_enable(True)
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        kinds = set(enc.profile_read())
        assert "fused_layer_attnout_mlp_qkv" in kinds
        assert ("final_ln_prune" in kinds) == (label == "separate"), kinds
        assert ("embed_ln" in kinds) == (label == "separate"), kinds
        outs[label] = (prune.cpu().numpy(), rank.cpu().numpy())
        enc.close()
    scale = max(1.0, float(np.abs(outs["separate"][0]).max()))
    assert np.abs(outs["fused"][0] - outs["separate"][0]).max() < 3e-4 * scale
    assert np.abs(outs["fused"][1] - outs["separate"][1]).max() < 3e-4 * scale
    # the fixture's own (fp32) weights, default policy, no capture, against the REFERENCE outputs stored in the fixture:
    # the bar, on every fixture (fp32 weights carry their lo planes: all-terms kernel set, two kernels per layer)
    rep = run_fixture_on_gpu(fixture, "bf16x3", capture=False)
    assert rep["prune_max_err"] < 1e-3 and rep["rank_max_err"] < 1e-3, rep


@pytest.mark.parametrize("fixture", ["g2_gte_varlen", "g8_base_refinit"])
@pytest.mark.parametrize("weights", ["bf16", "fp32"])
def test_panel_path_f16_f8_kernel_sets(fixture, weights):
    """hidden 768 (panel GEMMs): OP_FLAG_PANEL_F8 selects the fp16 + e4m3 kernel sets there too -- "f16-f8" for
    bf16-valued weights, "f16-f8-w" (the weights' lo part as a second e4m3 plane) for fp32-valued ones; every panel GEMM
    (q / k / v, attention output, Wi + GeGLU, MLP output) and LayerNorm then runs its fp16 + e4m3 form and attention
    writes o in that format.  Within 1e-3 of the oracle on the same weights (3-layer fixtures), and close to the (hi, lo)
    bf16 sets, which stay the default on this path (the format's error at the published depths: include/open_provence_hip.h)."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import pad_rows
    from oracle.modernbert_oracle import oracle_forward

    arrays, meta = load_golden(fixture)
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    if weights == "bf16":
        state = {k: v.to(torch.bfloat16).to(torch.float32) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v for k, v in state.items()}
    rows = rows_from_fixture(arrays)
    pre = bool(meta.get("prune_pre_final_norm", False))
    # default flags: fp32-valued weights take the format in the Wi GEMM alone (set "bf16x3+wi-f16-f8-w"), bf16-valued ones
    # only with OP_FLAG_PANEL_F8_WI; NO_F8: the (hi, lo) bf16 sets everywhere
    expected = {("bf16", PANEL_F8): "f16-f8", ("bf16", 0): "bf16-weights", ("fp32", PANEL_F8): "f16-f8-w", ("fp32", 0): "bf16x3+wi-f16-f8-w",
                ("bf16", PANEL_F8_WI): "bf16-weights+wi-f16-f8", ("fp32", PANEL_F8_WI): "bf16x3+wi-f16-f8-w",
                ("bf16", NO_F8): "bf16-weights", ("fp32", NO_F8): "bf16x3"}
    outs = {}
    for flags in (PANEL_F8, 0, PANEL_F8_WI, NO_F8):
        enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flags, prune_pre_final_norm=pre)
        enc.load_state_dict(state, calibrate=False)
        assert enc.effective_policy()["kernel_set"] == expected[(weights, flags)]
        enc.profile_enable(True)
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        kinds = set(enc.profile_read())
        assert {"gemm_qkv_rope", "gemm_attn_out", "gemm_wi_geglu", "gemm_mlp_out"} <= kinds, kinds  # the panel path ran
        outs[flags] = (prune.cpu().numpy(), rank.cpu().numpy())
        enc.close()
    ids, mask = pad_rows(rows)
    ref = oracle_forward(state, dims, ids, mask, prune_pre_final_norm=pre)
    m = mask.bool().numpy()
    for flags in (PANEL_F8, 0, PANEL_F8_WI, NO_F8):  # tolerance of the path: 1e-3 on logits against the CPU reference arithmetic
        assert np.isfinite(outs[flags][0]).all() and np.isfinite(outs[flags][1]).all()
        assert np.abs(outs[flags][0] - ref.pruning_logits.numpy()[m]).max() < 1e-3, flags
        assert np.abs(outs[flags][1] - ref.ranking_logits.numpy()).max() < 1e-3, flags
    assert np.abs(outs[NO_F8][0] - outs[PANEL_F8][0]).max() < 5e-4
    assert np.abs(outs[NO_F8][1] - outs[PANEL_F8][1]).max() < 5e-4
    assert np.abs(outs[NO_F8][0] - outs[PANEL_F8_WI][0]).max() < 3e-4  # the Wi GEMM alone: closer to the bf16 sets
    if weights == "bf16":  # bf16-valued weights: the default IS the (hi, lo) bf16 set
        assert np.array_equal(outs[0][0], outs[NO_F8][0])
    else:
        assert np.array_equal(outs[0][0], outs[PANEL_F8_WI][0])


@pytest.mark.parametrize("weights", ["bf16", "fp32"])
def test_tiny_scaled_weight_tensor_keeps_the_bf16_kernel_sets(weights):
    """The fp16 plane of the "f16 + fp8" sets has fp16's exponent range: a weight TENSOR scaled down into fp16's subnormal
    range (here two MLP output projections x 2^-10, their Wi x 32 so that the layer keeps its scale) cannot be
    represented -- run on that format the logits are off by 2e-3 .. 4e-3.  The pack kernels count such weights per tensor
    (note_f16_fit) and the library then keeps the (hi, lo) bf16 sets, whose exponent range is fp32's: within 1e-3."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import pad_rows
    from oracle.modernbert_oracle import oracle_forward

    arrays, meta = load_golden("g1_xsmall")
    dims = dims_from_meta(meta)
    rows = rows_from_fixture(arrays)
    ids, mask = pad_rows(rows)
    m = mask.bool().numpy()
    for scale, expected in ((1.0, {"bf16": "f16-f8", "fp32": "f16-f8-w"}), (32.0, {"bf16": "bf16-weights", "fp32": "bf16x3"})):
        state = state_from_fixture(arrays, meta)
        for layer in (1, 4):
            state[f"ranking_model.model.layers.{layer}.mlp.Wi.weight"] = state[f"ranking_model.model.layers.{layer}.mlp.Wi.weight"] * scale
            state[f"ranking_model.model.layers.{layer}.mlp.Wo.weight"] = state[f"ranking_model.model.layers.{layer}.mlp.Wo.weight"] / (scale * scale)
        if weights == "bf16":
            state = {k: v.to(torch.bfloat16).to(torch.float32) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v for k, v in state.items()}
        enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=0)
        enc.load_state_dict(state, calibrate=False)
        assert enc.effective_policy()["kernel_set"] == expected[weights], scale
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        enc.close()
        ref = oracle_forward(state, dims, ids, mask)
        assert np.abs(prune.cpu().numpy() - ref.pruning_logits.numpy()[m]).max() < 1e-3, scale
        assert np.abs(rank.cpu().numpy() - ref.ranking_logits.numpy()).max() < 1e-3, scale


def _overflowing_state():
    """g1_xsmall with MLP activations h = GeGLU(...) of 1e5 .. 1e6 (Wi x 512 in two layers; their Wo x 2^-18, rounded to
    multiples of 2^-24 so that fp16's subnormal grid holds them exactly and the load-time guard stays quiet)."""

    arrays, meta = load_golden("g1_xsmall")
    dims = dims_from_meta(meta)
    rows = rows_from_fixture(arrays)
    state = state_from_fixture(arrays, meta)
    scale = 512.0
    for layer in (1, 4):
        wi, wo = f"ranking_model.model.layers.{layer}.mlp.Wi.weight", f"ranking_model.model.layers.{layer}.mlp.Wo.weight"
        state[wi] = state[wi] * scale
        # x 2^-18, on fp16's subnormal grid exactly (multiples of 2^-24): representable, so the format is selected
        state[wo] = torch.round(state[wo] / (scale * scale) * 2.0**24) / 2.0**24
    return meta, dims, rows, state


def test_activation_beyond_fp16_range_is_loud_on_the_f8_sets_and_fine_on_the_bf16_sets():
    """The fp16 operand plane cannot hold an h of 1e5 .. 1e6: the whole-layer kernel converts it with IEEE overflow
    (Inf), so the raw outputs come back non-finite instead of being computed from a quietly clamped operand; the
    (hi, lo) bf16 sets (OP_FLAG_NO_F8) have fp32's range and match the oracle."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import pad_rows
    from oracle.modernbert_oracle import oracle_forward

    _meta, dims, rows, state = _overflowing_state()
    ids, mask = pad_rows(rows)
    ref = oracle_forward(state, dims, ids, mask)
    m = mask.bool().numpy()
    outs = {}
    for flags in (0, NO_F8):
        enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flags)
        enc.load_state_dict(state, calibrate=False)
        outs[flags] = enc.effective_policy()["kernel_set"]
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        p, r = prune.cpu().numpy(), rank.cpu().numpy()
        enc.close()
        if flags == 0:
            assert outs[flags] == "f16-f8-w"
            assert not np.isfinite(p).all() or not np.isfinite(r).all()  # Inf / NaN, not a quietly clamped operand
        else:
            assert np.abs(p - ref.pruning_logits.numpy()[m]).max() < 1e-3 and np.abs(r - ref.ranking_logits.numpy()).max() < 1e-3
    assert outs[NO_F8] == "bf16x3"


def test_overflow_of_the_f8_sets_falls_back_to_the_bf16_sets_instead_of_failing():
    """What a caller sees (reference precedent for a silent, correct retry: standalone.py:1631-1642): the range guard
    repeats a non-finite batch on the (hi, lo) bf16 kernel sets -- ``op_set_compact_operands`` through the C ABI, both weight packs
    are resident -- warns once, and the model stays there.  The result is BIT-identical to a model created with
    OP_FLAG_NO_F8, through every entry point: the guarded forward, ``forward()``, the single-block API, ``process()``."""

    import warnings

    from open_provence_amd.config import OpenProvenceConfig
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.modeling import OpenProvenceModel
    from open_provence_amd.packing import pack_rows
    from open_provence_amd.synthetic import pad_rows

    meta, dims, rows, state = _overflowing_state()
    ids_np, cu_np, max_len = pack_rows(rows)

    def run(enc, checked):
        ids, cu = torch.from_numpy(ids_np).to(enc.device), torch.from_numpy(cu_np).to(enc.device)
        fn = enc.forward_packed_checked if checked else enc.forward_packed
        prune, rank = fn(ids, cu, cu_np, max_len)
        torch.cuda.synchronize()
        return prune.cpu().numpy(), rank.cpu().numpy()

    ref_enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=NO_F8)
    ref_enc.load_state_dict(state, calibrate=False)
    ref_p, ref_r = run(ref_enc, False)
    ref_enc.close()
    assert np.isfinite(ref_p).all() and np.isfinite(ref_r).all()

    enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=0)
    enc.load_state_dict(state, calibrate=False)
    assert enc.effective_policy()["kernel_set"] == "f16-f8-w" and enc.f8_active()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        p, r = run(enc, True)
        assert any("fp16 + e4m3" in str(w.message) for w in caught)
    assert enc.effective_policy()["kernel_set"] == "bf16x3" and not enc.f8_active()  # it stays on the bf16 sets
    assert np.array_equal(p, ref_p) and np.array_equal(r, ref_r)
    with warnings.catch_warnings(record=True) as caught:  # nothing to fall back from any more, nothing to warn about
        warnings.simplefilter("always")
        p2, r2 = run(enc, True)
        assert not caught
    assert np.array_equal(p2, ref_p)
    enc.close()

    # the model-level entry points
    cfg = OpenProvenceConfig(base_model_config=meta["base_model_config"], tokenizer_name_or_path="x",
                             pruning_config={"hidden_size": meta["base_model_config"]["hidden_size"]}, max_length=512)
    ids, mask = pad_rows(rows)

    def model_with(env_no_f8):
        import os

        old = os.environ.pop("OPEN_PROVENCE_NO_F8", None)
        if env_no_f8:
            os.environ["OPEN_PROVENCE_NO_F8"] = "1"
        try:
            return OpenProvenceModel(cfg, device="cuda", tokenizer=CharTokenizer(), state_dict=state, calibrate=False)
        finally:
            os.environ.pop("OPEN_PROVENCE_NO_F8", None)
            if old is not None:
                os.environ["OPEN_PROVENCE_NO_F8"] = old

    plain = model_with(True)
    want = plain(input_ids=ids, attention_mask=mask)
    text = "alpha beta gamma. delta epsilon zeta eta. theta iota kappa lambda mu. " * 6
    want_proc = plain.process("what is gamma", [text, text[:120]], sentence_splitter=period_splitter, show_progress=False,
                              return_sentence_metrics=True, return_sentence_texts=True)
    want_raw = plain.get_raw_predictions("what is gamma", [text[:200]])
    for entry in ("forward", "process", "raw"):
        model = model_with(False)
        assert model.encoder.f8_active()
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            if entry == "forward":
                got = model(input_ids=ids, attention_mask=mask)
                assert torch.equal(got.ranking_logits, want.ranking_logits) and torch.equal(got.pruning_logits, want.pruning_logits)
            elif entry == "process":
                got = model.process("what is gamma", [text, text[:120]], sentence_splitter=period_splitter, show_progress=False,
                                    return_sentence_metrics=True, return_sentence_texts=True)
                for key in ("pruned_context", "kept_sentences", "removed_sentences", "reranking_score", "sentence_probabilities"):
                    assert got[key] == want_proc[key], key
            else:
                got = model.get_raw_predictions("what is gamma", [text[:200]])
                assert got.ranking_score == want_raw.ranking_score and np.array_equal(got.pruning_probs, want_raw.pruning_probs)
        assert not model.encoder.f8_active()


NO_LAYER_PAIRS = 8192  # OP_FLAG_NO_LAYER_PAIRS: keep the 8 waves x 16 rows whole-layer kernel on the single-pass sets


@pytest.mark.parametrize("kernel_set", ["f16", "bf16"])
def test_wave_pair_layer_kernel_matches_the_8x16_kernel_and_the_oracle(kernel_set):
    """Round 6: on the single-pass kernel sets a batch (here: more than one 128-row block per CU) runs its whole-layer launches
    on the wave-pair kernel (opk_layer16p.hip.h: two waves per SIMD share a 32-row tile and split the output features; other
    weight packs, LayerNorm statistics combined from two halves, h and LN(x) exchanged through LDS, the residual stream
    tiled between launches).  Same terms as the 8 x 16 kernel it replaces: equal to it up to the summation order (fp16:
    <= 6e-5 on logits of magnitude 7; bf16: its own rounding noise), bit-identical from run to run, and -- fp16 on
    reference-initialised weights -- within the calibration's bound of the fp32 oracle on ragged rows."""

    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.packing import pack_rows
    from open_provence_amd.synthetic import named_dims, pad_rows, refinit_state_dict, synth_pair_batch
    from oracle.modernbert_oracle import oracle_forward

    dims = named_dims("xsmall")
    state = refinit_state_dict(dims, seed=21)
    rows = [r[: 29 + (i * 53) % 480] for i, r in enumerate(synth_pair_batch(dims, 640, 512, seed=9))]  # ~165 k rows: > 256 blocks
    ids_np, cu_np, max_len = pack_rows(rows)
    ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
    outs = {}
    for label, flags in (("pairs", 0), ("pairs again", 0), ("8x16", NO_LAYER_PAIRS)):
        enc = HipEncoder(dims, device="cuda:0", flags=flags)
        enc.load_state_dict(state, calibrate=False, kernel_set=kernel_set)
        prune, rank = enc.forward_packed(ids, cu, cu_np, max_len)
        torch.cuda.synchronize()
        outs[label] = (prune.cpu().numpy(), rank.cpu().numpy())
        enc.close()
    assert np.array_equal(outs["pairs"][0], outs["pairs again"][0]) and np.array_equal(outs["pairs"][1], outs["pairs again"][1])
    tol = 6e-5 if kernel_set == "f16" else 6e-4
    assert np.abs(outs["pairs"][0] - outs["8x16"][0]).max() < tol and np.abs(outs["pairs"][1] - outs["8x16"][1]).max() < tol
    if kernel_set == "f16":
        sample = list(range(0, len(rows), 40))  # 16 ragged rows through the oracle
        pad_ids, mask = pad_rows([rows[i] for i in sample])
        ref = oracle_forward(state, dims, pad_ids, mask)
        rp, rr = ref.pruning_logits.numpy(), ref.ranking_logits.numpy()
        for j, i in enumerate(sample):
            a, b = int(cu_np[i]), int(cu_np[i + 1])
            assert np.abs(outs["pairs"][0][a:b] - rp[j, : b - a]).max() < 3e-4, (i, b - a)
            assert np.abs(outs["pairs"][1][i] - rr[j]).max() < 3e-4


def test_wave_pair_layer_kernel_on_small_ragged_batches():
    """The wave-pair kernel below one 128-row block per CU (OP_FLAG_NO_SMALL_BLOCKS: the other launches on 128-row blocks too): rows of 1 .. 645 tokens on and around every tiling granularity, partly empty last blocks, a one-row batch --
    against the fp32 oracle, and different from the 8 x 16 kernel's bits (it did run).  scripts/forward_fuzz.py --flags
    NO_SMALL_BLOCKS is the long form (profiles/r06_forward_fuzz_pairs.txt)."""

    from open_provence_amd import _lib
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import named_dims, pad_rows, refinit_state_dict, synth_pair_batch
    from oracle.modernbert_oracle import oracle_forward

    dims = named_dims("xsmall")
    state = refinit_state_dict(dims, seed=11)
    full = synth_pair_batch(dims, 12, 700, seed=5)
    batches = [[full[i][:n] for i, n in enumerate((1, 2, 15, 17, 63, 127, 128, 129, 255, 385, 513, 645))], [full[3][:127]], [full[0][:236], full[1][:191], full[2][:384]]]
    outs = {}
    for label, flags in (("pairs", _lib.OP_FLAG_NO_SMALL_BLOCKS), ("8x16", _lib.OP_FLAG_NO_SMALL_BLOCKS | NO_LAYER_PAIRS), ("default", 0)):
        enc = HipEncoder(dims, device="cuda:0", flags=flags)  # default: the wave-pair kernel between 64-row first / last launches
        enc.load_state_dict(state, calibrate=False, kernel_set="f16")
        outs[label] = []
        for rows in batches:
            p, r, _ = enc.forward_rows(rows)
            torch.cuda.synchronize()
            outs[label].append((p.cpu().numpy(), r.cpu().numpy()))
        enc.close()
    for b, rows in enumerate(batches):
        ids, mask = pad_rows(rows)
        with torch.no_grad():
            ref = oracle_forward(state, dims, ids, mask)
        m = mask.bool().numpy()
        for label in ("pairs", "default"):
            p, r = outs[label][b]
            assert np.isfinite(p).all() and np.isfinite(r).all()
            assert np.abs(p - ref.pruning_logits.numpy()[m]).max() < 3e-4 and np.abs(r - ref.ranking_logits.numpy()).max() < 3e-4, (label, b)
            assert np.abs(p - outs["8x16"][b][0]).max() < 6e-5, (label, b)
    assert not np.array_equal(outs["pairs"][0][0], outs["8x16"][0][0])
