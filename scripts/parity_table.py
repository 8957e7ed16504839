#!/usr/bin/env python
"""GPU-box report: every forward fixture of tests/golden through the HIP path (C ABI), default precision, once with
the fixture's fp32 weights and once with the weights rounded to bf16 first (= a bf16 checkpoint: the weight-lo terms
are elided, kernel set "bf16-weights"; the reference values are then the oracle's on the same rounded weights).
Prints one line per case: max |error| of pruning logits, ranking logits, keep probability.
Usage: scripts/parity_table.py > profiles/r02_parity_table.txt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch

from helpers import dims_from_meta, load_golden, rows_from_fixture, state_from_fixture
from parity_utils import run_fixture_on_gpu

FIXTURES = ["g0b_hd64_refinit", "g0c_hd64_synth", "g1m_meanpool", "g1_xsmall", "g2_gte_varlen", "g7_xsmall_refinit",
            "g8_base_refinit", "g12_prenorm_tf4"]


def bf16_checkpoint_case(name):
    """HIP (bf16 checkpoint -> weight-lo elision) vs the oracle evaluated on the same bf16-rounded weights."""
    from open_provence_amd.engine import HipEncoder
    from oracle.modernbert_oracle import oracle_forward

    arrays, meta = load_golden(name)
    dims = dims_from_meta(meta)
    state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.ndim == 2 and "embeddings" not in k else v)
             for k, v in state_from_fixture(arrays, meta).items()}
    rows = rows_from_fixture(arrays)
    pre = bool(meta.get("prune_pre_final_norm", False))
    enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", prune_pre_final_norm=pre)
    enc.load_state_dict(state)
    kernel_set = enc.effective_policy()["kernel_set"]
    prune, rank, cu = enc.forward_rows(rows)
    torch.cuda.synchronize()
    prune, rank = prune.cpu().numpy(), rank.cpu().numpy()
    enc.close()
    ids, mask = torch.from_numpy(arrays["input_ids"]), torch.from_numpy(arrays["attention_mask"])
    pick = np.argsort(mask.sum(1).numpy())[: min(4, len(rows))]  # the oracle is slow: the four shortest rows
    width = int(mask[pick].sum(1).max())
    ref = oracle_forward(state, dims, ids[pick, :width], mask[pick, :width], prune_pre_final_norm=pre)
    worst_p = worst_r = worst_k = 0.0
    for j, b in enumerate(pick):
        n = int(mask[b].sum())
        got = prune[cu[b] : cu[b] + n]
        want = ref.pruning_logits[j, :n].numpy()
        worst_p = max(worst_p, float(np.abs(got - want).max()))
        worst_r = max(worst_r, float(np.abs(rank[b] - ref.ranking_logits[j].numpy()).max()))
        kg = 1 / (1 + np.exp(-(got[:, 1] - got[:, 0]).astype(np.float64)))
        kw = 1 / (1 + np.exp(-(want[:, 1] - want[:, 0]).astype(np.float64)))
        worst_k = max(worst_k, float(np.abs(kg - kw).max()))
    return kernel_set, worst_p, worst_r, worst_k


print(f"{'fixture':22s} {'weights':16s} {'kernel set':14s} {'prune':>9s} {'rank':>9s} {'keep-prob':>9s}")
for name in FIXTURES:
    rep = run_fixture_on_gpu(name, "bf16x3", capture=False)
    print(f"{name:22s} {'fp32 (fixture)':16s} {rep['kernel_set']:14s} {rep['prune_max_err']:9.2e} {rep['rank_max_err']:9.2e} {rep['keep_prob_max_err']:9.2e}")
    ks, p, r, k = bf16_checkpoint_case(name)
    print(f"{name:22s} {'rounded to bf16':16s} {ks:14s} {p:9.2e} {r:9.2e} {k:9.2e}   (vs oracle on the same weights, 4 shortest rows)")
