#!/usr/bin/env python
"""Full-depth published shapes (base / large / en-gte, synthetic weights): error of the fp16 + e4m3 kernel sets and of the
(hi, lo) bf16 sets against the oracle, fp32-valued and bf16-valued weights."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from open_provence_amd.engine import HipEncoder  # noqa: E402
from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict  # noqa: E402
from oracle.modernbert_oracle import oracle_forward  # noqa: E402

CASES = [("base", [512, 300, 77]), ("large", [512, 129]), ("large", [2048]), ("en-gte", [640, 64])]
for model_name, lengths in CASES:
    dims = named_dims(model_name, vocab_size=4096)
    for weights in ("fp32", "bf16"):
        state = synth_state_dict(dims, 17)
        if weights == "bf16":
            state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.ndim == 2 and "embeddings" not in k else v) for k, v in state.items()}
        rows = [synth_pair_batch(dims, 1, n, seed=100 + n)[0] for n in lengths]
        refs = []
        for row in rows:
            ids = torch.tensor([row], dtype=torch.long)
            refs.append(oracle_forward(state, dims, ids, torch.ones_like(ids)))
        for flags in (2048, 4096, 512):  # OP_FLAG_PANEL_F8 / OP_FLAG_PANEL_F8_WI / OP_FLAG_NO_F8
            enc = HipEncoder(dims, device="cuda", flags=flags)
            enc.load_state_dict(state)
            ks = enc.effective_policy()["kernel_set"]
            prune, rank, cu = enc.forward_rows(rows)
            errs = []
            for i, ref in enumerate(refs):
                ep = (prune[cu[i] : cu[i + 1]].cpu() - ref.pruning_logits[0]).abs().max().item()
                er = (rank[i].cpu() - ref.ranking_logits[0]).abs().max().item()
                errs.append(f"{lengths[i]}: {ep:.2e}/{er:.2e}")
            scale = max(float(r.pruning_logits.abs().max()) for r in refs)
            print(f"{model_name:7s} {weights} [{ks:22s}] max|logit| {scale:6.1f}  " + "  ".join(errs), flush=True)
            enc.close()
