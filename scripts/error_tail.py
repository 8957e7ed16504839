#!/usr/bin/env python
"""GPU: the TAIL of the logit error of the default kernel sets over many rows -- every pair of bench.py's batch (and of
batches of other seeds and lengths) against the oracle, for the fp16 + e4m3 sets and the (hi, lo) bf16 sets side by side.
The fixtures and the 10 spot-checked pairs of tests/test_gpu_timed_path.py say where the error sits on a few rows; this
says how far its maximum moves over hundreds.

    python scripts/error_tail.py [--batches 3] [--pairs 256]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from open_provence_amd import _lib  # noqa: E402
from open_provence_amd.engine import HipEncoder  # noqa: E402
from open_provence_amd.packing import pack_rows  # noqa: E402
from open_provence_amd.synthetic import named_dims, pad_rows, synth_pair_batch, synth_state_dict  # noqa: E402
from oracle.modernbert_oracle import oracle_forward  # noqa: E402  (the checker)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=256)
    args = ap.parse_args()
    torch.set_num_threads(16)
    dims = named_dims("xsmall")
    for weights in ("fp32", "bf16"):
        state = synth_state_dict(dims, seed=7)
        if weights == "bf16":
            state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.ndim == 2 and "embeddings" not in k else v) for k, v in state.items()}
        encs = {}
        for name, flags in (("f8", 0), ("bf16 pairs", _lib.OP_FLAG_NO_F8)):
            enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flags)
            enc.load_state_dict(state)
            encs[name] = enc
        errs = {name: [] for name in encs}
        for b in range(args.batches):
            seq_len = [512, 384, 200][b % 3]
            rows = synth_pair_batch(dims, args.pairs, seq_len, seed=1234 + 1000 * b)
            ids, mask = pad_rows(rows)
            ref_p, ref_r = [], []
            with torch.no_grad():
                for s in range(0, len(rows), 32):
                    ref = oracle_forward(state, dims, ids[s : s + 32], mask[s : s + 32], attn="sdpa")
                    ref_p.append(ref.pruning_logits.numpy())
                    ref_r.append(ref.ranking_logits.numpy())
            ref_p, ref_r = np.concatenate(ref_p), np.concatenate(ref_r)
            ids_np, cu_np, max_len = pack_rows(rows)
            for name, enc in encs.items():
                prune, rank = enc.forward_packed(torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda(), cu_np, max_len)
                prune, rank = prune.cpu().numpy(), rank.cpu().numpy()
                for i, row in enumerate(rows):
                    errs[name].append(max(float(np.abs(prune[cu_np[i] : cu_np[i + 1]] - ref_p[i, : len(row)]).max()),
                                          float(np.abs(rank[i] - ref_r[i]).max())))
        for name, enc in encs.items():
            e = np.sort(np.asarray(errs[name]))
            print(f"xsmall {weights:5s} {enc.effective_policy()['kernel_set']:12s} rows {len(e):4d}  median {np.median(e):.2e}  p90 {e[int(0.9 * len(e))]:.2e}  "
                  f"p99 {e[int(0.99 * len(e))]:.2e}  max {e[-1]:.2e}  rows above 8e-4: {int((e > 8e-4).sum())}  above 1e-3: {int((e > 1e-3).sum())}", flush=True)
            enc.close()


if __name__ == "__main__":
    main()
