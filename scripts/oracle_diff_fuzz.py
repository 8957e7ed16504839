#!/usr/bin/env python
"""Differential fuzz of the ORACLE against the live reference forward (build container only: /root/reference + the installed
transformers ModernBERT): random model shapes -- hidden size, heads (head_dim 16 .. 64), layers, intermediate size, window,
global-attention period, cls / mean pooling, reference-initialised or O(1) synthetic weights -- and random ragged batches,
the reference's ``OpenProvenceModel.forward`` (CPU fp32) against ``oracle/modernbert_oracle.oracle_forward`` (both attention
forms).  The goldens pin the oracle on nine fixed configurations; this looks between them.

    python scripts/oracle_diff_fuzz.py [--trials 40] [--seed 0]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "golden"))

from oracle.modernbert_oracle import oracle_forward  # noqa: E402
from open_provence_amd.synthetic import refinit_state_dict, synth_state_dict  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import make_golden as mg

    ref = mg.load_reference()
    rng = np.random.default_rng(args.seed)
    worst = 0.0
    for trial in range(args.trials):
        head_dim = int(rng.choice([16, 32, 64]))
        heads = int(rng.integers(1, 7))
        cfg = mg.base_cfg(vocab_size=int(rng.choice([256, 1024, 5000])), hidden_size=head_dim * heads,
                          intermediate_size=int(rng.choice([64, 96, 256, 384])), num_hidden_layers=int(rng.integers(2, 7)),
                          num_attention_heads=heads, local_attention=int(rng.choice([8, 16, 32, 64, 128])),
                          global_attn_every_n_layers=int(rng.choice([2, 3, 4])),  # (transformers 5 refuses a stack of one layer type)
                          classifier_pooling=str(rng.choice(["cls", "mean"])))
        init = str(rng.choice(["synth", "refinit"]))
        try:
            model, dims = mg.build_model(ref, cfg, max_length=512, seed=0, weight_seed=int(rng.integers(1, 1000)), weight_init=init)
        except Exception as exc:  # noqa: BLE001 - a shape the installed transformers refuses to configure: not a forward to compare
            print(f"{trial:3d} skipped ({type(exc).__name__}): layers {cfg['num_hidden_layers']} global/{cfg['global_attn_every_n_layers']}", flush=True)
            continue
        lengths = [int(rng.integers(12, 200)) for _ in range(int(rng.integers(1, 6)))]
        if rng.random() < 0.3:
            lengths[0] = int(rng.integers(200, 420))
        ids, mask = mg.make_rows(dims, lengths, int(rng.integers(0, 10_000)))
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=mask)
        state = {k: v for k, v in model.state_dict().items() if "inv_freq" not in k}
        m = mask.bool().numpy()
        scale = float(np.abs(out.pruning_logits.numpy()[m]).max())
        errs = []
        for attn in ("eager", "sdpa"):
            with torch.no_grad():
                got = oracle_forward(state, dims, ids, mask, attn=attn)
            ep = float(np.abs(got.pruning_logits.numpy() - out.pruning_logits.float().numpy())[m].max())
            er = float(np.abs(got.ranking_logits.numpy() - out.ranking_logits.float().numpy()).max())
            errs.append(max(ep, er))
        worst = max(worst, *errs)
        flag = "  <-- ABOVE 1e-4" if max(errs) > 1e-4 else ""
        print(f"{trial:3d} H {cfg['hidden_size']:3d} heads {heads} hd {head_dim} layers {cfg['num_hidden_layers']} I {cfg['intermediate_size']:3d} window {cfg['local_attention']:3d} "
              f"global/{cfg['global_attn_every_n_layers']} {cfg['classifier_pooling']:4s} {init:7s} rows {lengths}  max|logit| {scale:6.2f}  "
              f"eager {errs[0]:.1e} sdpa {errs[1]:.1e}{flag}", flush=True)
    print(f"worst |oracle - reference| over {args.trials} configurations: {worst:.2e}", flush=True)
    sys.exit(1 if worst > 1e-4 else 0)


if __name__ == "__main__":
    main()
