#!/usr/bin/env python
"""GPU check of the fp16 + e4m3 kernel sets against the oracle and against the (hi, lo) bf16 sets on the same weights:
bf16-valued weights -> "f16-f8" vs "bf16-weights"; fp32-valued weights -> "f16-f8-w" vs "bf16x3" (two kernels per layer)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import dims_from_meta, load_golden, rows_from_fixture, state_from_fixture  # noqa: E402
from open_provence_amd.engine import HipEncoder  # noqa: E402
from open_provence_amd.synthetic import pad_rows  # noqa: E402
from oracle.modernbert_oracle import oracle_forward  # noqa: E402

NO_F8 = 512
PANEL_F8 = 2048  # the panel path (hidden 512 / 768) takes the fp16 + e4m3 sets on request only


def run(fixture: str, weights: str) -> None:
    arrays, meta = load_golden(fixture)
    dims = dims_from_meta(meta)
    state = state_from_fixture(arrays, meta)
    if weights == "bf16":
        state = {k: v.to(torch.bfloat16).to(torch.float32) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v for k, v in state.items()}
    rows = rows_from_fixture(arrays)
    pre = bool(meta.get("prune_pre_final_norm", False))
    outs = {}
    for label, flags in (("f8", PANEL_F8), ("bf16", NO_F8)):
        enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flags, prune_pre_final_norm=pre)
        enc.load_state_dict(state)
        ks = enc.effective_policy()["kernel_set"]
        prune, rank, _ = enc.forward_rows(rows)
        torch.cuda.synchronize()
        outs[label] = (prune.cpu().numpy(), rank.cpu().numpy(), ks)
        enc.close()
    ids, mask = pad_rows(rows)
    ref = oracle_forward(state, dims, ids, mask, prune_pre_final_norm=pre)
    m = mask.bool().numpy()
    rp, rr = ref.pruning_logits.numpy()[m], ref.ranking_logits.numpy()
    for label in ("f8", "bf16"):
        p, r, ks = outs[label]
        print(f"{fixture:20s} {weights}-valued weights  [{ks:12s}] prune {np.abs(p - rp).max():.2e} rank {np.abs(r - rr).max():.2e} "
              f"finite {bool(np.isfinite(p).all() and np.isfinite(r).all())}", flush=True)


if __name__ == "__main__":
    for fixture in sys.argv[1:] or ["g0c_hd64_synth", "g1_xsmall", "g7_xsmall_refinit", "g0b_hd64_refinit", "g12_prenorm_tf4"]:
        for weights in ("bf16", "fp32"):
            run(fixture, weights)
