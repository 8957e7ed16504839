#!/usr/bin/env python
"""GPU check + A/B of the LayerNorm tail of the panel path's residual GEMMs (opk_panel.hip.h: panel_ln_tail).

For each model: the same synthetic checkpoint and batch through two encoders -- default flags (LayerNorm done by the last
block of every row block inside the residual GEMM) and OP_FLAG_NO_LN_FUSION (LayerNorm as its own launches).  The
outputs must be bit-identical, on every repetition (the tail is ordered by a ticket and fences: a race would show up
as a run that differs); then both are timed, alternating.

    python scripts/ln_tail_check.py [--models base,en-gte,large] [--weights fp32,bf16] [--reps 6] [--iters 8]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from open_provence_amd import _lib  # noqa: E402
from open_provence_amd.engine import HipEncoder  # noqa: E402
from open_provence_amd.packing import pack_rows  # noqa: E402
from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict, synth_varlen_lengths  # noqa: E402


def batch_for(model: str, dims, small: bool):
    if small:  # a ragged handful: 7 row blocks, the last one partial
        return [synth_pair_batch(dims, 1, n, seed=99 + i)[0] for i, n in enumerate([28, 129, 300, 77, 256, 33])]
    if model == "en-gte":
        lengths = synth_varlen_lengths(256 * 512, seed=1234)
        return [synth_pair_batch(dims, 1, n, seed=1234 + 7 * i)[0] for i, n in enumerate(lengths)]
    if model == "large":
        return synth_pair_batch(dims, 64, 2048, seed=1234)
    return synth_pair_batch(dims, 256, 512, seed=1234)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="base,en-gte,large")
    ap.add_argument("--weights", default="fp32,bf16")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--iters", type=int, default=8)
    args = ap.parse_args()
    bad = 0
    for model in args.models.split(","):
        dims = named_dims(model)
        for weights in args.weights.split(","):
            state = synth_state_dict(dims, seed=7)
            if weights == "bf16":
                state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.ndim == 2 and "embeddings" not in k else v) for k, v in state.items()}
            encs = {}
            for name, flags in (("tail", 0), ("launch", _lib.OP_FLAG_NO_LN_FUSION)):
                enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flags)
                enc.load_state_dict(state)
                encs[name] = enc
            kernel_set = encs["tail"].effective_policy()["kernel_set"]
            for small in (True, False):
                rows = batch_for(model, dims, small)
                ids_np, cu_np, max_len = pack_rows(rows)
                ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
                ref_p, ref_r = encs["launch"].forward_packed(ids, cu, cu_np, max_len)
                torch.cuda.synchronize()
                ref_p, ref_r = ref_p.clone(), ref_r.clone()
                assert torch.isfinite(ref_p).all() and torch.isfinite(ref_r).all()
                diff = 0
                for _ in range(args.reps):
                    p, r = encs["tail"].forward_packed(ids, cu, cu_np, max_len)
                    torch.cuda.synchronize()
                    diff += int((p.view(torch.int32) != ref_p.view(torch.int32)).sum()) + int((r.view(torch.int32) != ref_r.view(torch.int32)).sum())
                bad += diff
                line = f"{model:7s} {weights:5s} {kernel_set:22s} {'ragged 6 rows' if small else f'{len(rows)} rows {int(cu_np[-1])} tokens':26s} differing words over {args.reps} runs: {diff}"
                if not small:
                    ms = {"tail": [], "launch": []}
                    for _ in range(3):
                        for name in ("launch", "tail"):
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            for _ in range(args.iters):
                                encs[name].forward_packed(ids, cu, cu_np, max_len)
                            torch.cuda.synchronize()
                            ms[name].append((time.perf_counter() - t0) / args.iters * 1e3)
                    a, b = min(ms["launch"]), min(ms["tail"])
                    line += f" | ms per forward: own launches {a:.2f}, tail {b:.2f} ({(a / b - 1) * 100:+.1f} % pairs/s)"
                print(line, flush=True)
            for enc in encs.values():
                enc.close()
    print("OK" if bad == 0 else f"MISMATCH: {bad} words", flush=True)
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
