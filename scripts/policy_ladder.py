#!/usr/bin/env python
"""Where every kernel set stands on a given kind of weights: max |error| of the logits against the oracle and against
the all-terms (hi, lo) bf16 set on the device, and pairs/s of the timed 256 x 512 batch.

    python scripts/policy_ladder.py [--model xsmall|base] [--weights refinit|synth] [--pairs 256] [--seq-len 512]

The oracle leg runs on a sample of the pairs (--oracle-pairs) so that the script finishes in a minute on the box's host
cores; the device leg compares every row."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from open_provence_amd import _lib  # noqa: E402
from open_provence_amd.engine import HipEncoder  # noqa: E402
from open_provence_amd.packing import pack_rows  # noqa: E402
from open_provence_amd.synthetic import named_dims, pad_rows, refinit_state_dict, synth_pair_batch, synth_state_dict  # noqa: E402
from oracle.modernbert_oracle import oracle_forward  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="xsmall")
    ap.add_argument("--weights", default="refinit", choices=["refinit", "synth"])
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--oracle-pairs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--layers", type=int, default=0)
    args = ap.parse_args()
    dims = named_dims(args.model, **({"num_layers": args.layers} if args.layers else {}))
    state = (refinit_state_dict if args.weights == "refinit" else synth_state_dict)(dims, seed=7)
    rows = synth_pair_batch(dims, args.pairs, args.seq_len, seed=1234)
    # a few ragged rows in front: the sample the oracle sees
    ragged = [rows[i][: n] for i, n in enumerate((args.seq_len, 17, 130, 333, args.seq_len - 1, 64, 257, 96)[: args.oracle_pairs])]
    ids_np, cu_np, max_len = pack_rows(rows)
    dev = torch.device("cuda", 0)
    ids, cu = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)
    r_ids_np, r_cu_np, r_max = pack_rows(ragged)
    r_ids, r_cu = torch.from_numpy(r_ids_np).to(dev), torch.from_numpy(r_cu_np).to(dev)
    o_ids, o_mask = pad_rows(ragged)
    with torch.no_grad():
        ref = oracle_forward(state, dims, o_ids, o_mask)
    m = o_mask.bool().numpy()
    rp, rr = ref.pruning_logits.numpy()[m], ref.ranking_logits.numpy()
    print(f"{args.model} {args.weights}: |prune logit| max {np.abs(rp).max():.3g}, |rank logit| max {np.abs(rr).max():.3g}", flush=True)

    # every kernel set the handle can run, pinned by name on ONE set of fp32-valued weights (op_select_kernel_set), the
    # all-terms (hi, lo) bf16 set first = the reference of the on-device column
    configs = [("pinned", name) for name in ("bf16x3", "bf16x3+wi-f16-f8-w", "f16-f8-w", "bf16-weights", "bf16-weights+wi-f16-f8",
                                             "f16-f8-w+attn-f16", "f16-f8", "f16-f8+attn-f16", "f16+mlp-f16-f8-w", "f16+mlp-f16-f8", "bf16", "f16")] + [("calibrated", None)]
    base = None
    for label, name in configs:
        enc = HipEncoder(dims, device=dev, precision="bf16x3", flags=0)
        try:
            enc.load_state_dict(state, kernel_set=name, calibrate=(1e-4 if name is None else False))
        except _lib.HipLibraryError as exc:
            print(f"{label:10s} [{name}] not available: {str(exc)[:90]}", flush=True)
            enc.close()
            continue
        if name is None:
            print("calibration:", enc.calibration, flush=True)
        ks = enc.effective_policy()["kernel_set"]
        p_r, k_r = enc.forward_packed(r_ids, r_cu, r_cu_np, r_max)
        p, k = enc.forward_packed(ids, cu, cu_np, max_len)
        torch.cuda.synchronize()
        for _ in range(3):
            enc.forward_packed(ids, cu, cu_np, max_len)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            enc.forward_packed(ids, cu, cu_np, max_len)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        enc.profile_enable(True)
        enc.profile_reset()
        enc.forward_packed(ids, cu, cu_np, max_len)
        prof = enc.profile_read()
        enc.profile_enable(False)
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        p_c, k_c = p.cpu().numpy(), k.cpu().numpy()
        if base is None:
            base = (p_c, k_c)
        e_or = max(np.abs(p_r.cpu().numpy() - rp).max(), np.abs(k_r.cpu().numpy() - rr).max())
        e_dev = max(np.abs(p_c - base[0]).max(), np.abs(k_c - base[1]).max())
        print(f"{label:18s} [{ks:22s}] vs oracle {e_or:.2e}  vs all-terms on device {e_dev:.2e}  {args.pairs / dt:9.0f} pairs/s  "
              f"dominant {dom[0]} {dom[1]['avg_ms'] * 1e3:.0f} us x {dom[1]['launches']}", flush=True)
        enc.close()


if __name__ == "__main__":
    main()
