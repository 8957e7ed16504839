"""Run a golden forward fixture through the HIP path and compare (test infrastructure)."""

from __future__ import annotations

from typing import Any

import numpy as np
import torch

from helpers import dims_from_meta, load_golden, rows_from_fixture, state_from_fixture


def run_fixture_on_gpu(name: str, precision="bf16x3", chunk_rows: int | None = None, capture: bool = True, flags=None, return_outputs: bool = False,
                       calibrate=None, kernel_set=None):
    """Returns dict with max-abs errors of the HIP path vs the reference outputs stored in the fixture."""

    from open_provence_amd.engine import HipEncoder

    arrays, meta = load_golden(name)
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    rows = rows_from_fixture(arrays)
    enc = HipEncoder(dims, device="cuda:0", precision=precision, chunk_rows=chunk_rows, flags=flags,
                     prune_pre_final_norm=bool(meta.get("prune_pre_final_norm", False)))
    enc.load_state_dict(state, calibrate=calibrate, kernel_set=kernel_set)
    policy = enc.effective_policy()
    if capture:
        with enc.capture_hidden():
            prune, rank, cu = enc.forward_rows(rows)
            torch.cuda.synchronize()
        hidden = enc.captured.cpu().numpy()
    else:
        prune, rank, cu = enc.forward_rows(rows)
        torch.cuda.synchronize()
        hidden = None
    prune = prune.cpu().numpy()
    rank = rank.cpu().numpy()
    enc.close()

    mask = arrays["attention_mask"].astype(bool)
    ref_prune = arrays["pruning_logits"][mask]  # row-major over (b, l) == packed order
    ref_rank = arrays["ranking_logits"]
    report: dict[str, Any] = {
        "name": name,
        "precision": precision,
        "kernel_set": policy["kernel_set"],
        "calibration": enc.calibration,
        "terms": policy["terms"],
        "finite": bool(np.isfinite(prune).all() and np.isfinite(rank).all()),
        "prune_max_err": float(np.abs(prune - ref_prune).max()),
        "rank_max_err": float(np.abs(rank - ref_rank).max()),
    }
    keep_ref = 1.0 / (1.0 + np.exp(-(ref_prune[:, 1] - ref_prune[:, 0]).astype(np.float64)))
    keep_hip = 1.0 / (1.0 + np.exp(-(prune[:, 1] - prune[:, 0]).astype(np.float64)))
    report["keep_prob_max_err"] = float(np.abs(keep_hip - keep_ref).max())
    report["sigmoid_rank_max_err"] = float(
        np.abs(1 / (1 + np.exp(-rank[:, 0].astype(np.float64))) - 1 / (1 + np.exp(-ref_rank[:, 0].astype(np.float64)))).max()
    )
    if return_outputs:
        report["prune"] = prune
        report["rank"] = rank
    stride = meta.get("hidden_stride")
    if hidden is not None and stride:
        per_layer = []
        lengths = mask.sum(axis=1)
        for i in range(meta["n_hidden_states"]):
            ref_h = arrays[f"hidden_{i}"]  # [B, ceil(L/stride), H]
            worst = 0.0
            for b, length in enumerate(lengths):
                pos = np.arange(0, int(length), stride)
                got = hidden[i, cu[b] + pos]
                worst = max(worst, float(np.abs(got - ref_h[b, : len(pos)]).max()))
            per_layer.append(worst)
        report["hidden_max_err"] = per_layer
    return report
