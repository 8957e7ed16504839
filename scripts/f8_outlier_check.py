#!/usr/bin/env python
"""Large intermediate activations under the fp16 + e4m3 kernel sets: layer `L`'s Wi scaled by s and its Wo by 1 / s^2, so
that h = gelu(a) * g grows ~s^2-fold while the layer's output keeps its scale.  Error against the oracle on the SAME
weights, fp16 + e4m3 sets vs the (hi, lo) bf16 sets."""
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

fixture = sys.argv[1] if len(sys.argv) > 1 else "g1_xsmall"
arrays, meta = load_golden(fixture)
dims = dims_from_meta(meta)
rows = rows_from_fixture(arrays)
ids, mask = pad_rows(rows)
m = mask.bool().numpy()
for scale in (1.0, 8.0, 32.0, 128.0, 512.0):
    for weights in ("bf16", "fp32"):
        state = state_from_fixture(arrays, meta)
        for layer in (1, 4):
            state[f"ranking_model.model.layers.{layer}.mlp.Wi.weight"] = state[f"ranking_model.model.layers.{layer}.mlp.Wi.weight"] * scale
            state[f"ranking_model.model.layers.{layer}.mlp.Wo.weight"] = state[f"ranking_model.model.layers.{layer}.mlp.Wo.weight"] / (scale * scale)
        if weights == "bf16":
            state = {k: v.to(torch.bfloat16).to(torch.float32) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v for k, v in state.items()}
        ref = oracle_forward(state, dims, ids, mask)
        rp, rr = ref.pruning_logits.numpy()[m], ref.ranking_logits.numpy()
        out = []
        for flags in (0, 512):
            enc = HipEncoder(dims, device="cuda:0", flags=flags)
            enc.load_state_dict(state)
            ks = enc.effective_policy()["kernel_set"]
            prune, rank, _ = enc.forward_rows(rows)
            torch.cuda.synchronize()
            p, r = prune.cpu().numpy(), rank.cpu().numpy()
            out.append(f"[{ks}] prune {np.abs(p - rp).max():.2e} rank {np.abs(r - rr).max():.2e} finite {bool(np.isfinite(p).all())}")
            enc.close()
        print(f"{fixture} Wi x{scale:g} {weights}: max|logit| {np.abs(rp).max():.1f}  " + "   ".join(out), flush=True)
