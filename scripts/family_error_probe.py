#!/usr/bin/env python
"""Which contraction family carries the error of a single-pass policy on given weights?  Each family alone (and a few groups)
is evaluated single-pass in bf16 on the all-terms kernels with cleared lo operands (OP_PRECISION_CUSTOM), everything else
with all terms; max |logit difference| to the all-terms result on the same device batch.  fp16 operands err ~8x less per
family, in the same proportions.   python scripts/family_error_probe.py [--model base] [--weights refinit]"""
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
from open_provence_amd.synthetic import named_dims, refinit_state_dict, synth_pair_batch, synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="base")
ap.add_argument("--weights", default="refinit")
ap.add_argument("--pairs", type=int, default=32)
args = ap.parse_args()
dims = named_dims(args.model)
state = (refinit_state_dict if args.weights == "refinit" else synth_state_dict)(dims, seed=7)
rows = synth_pair_batch(dims, args.pairs, 512, seed=1234)
ids_np, cu_np, max_len = pack_rows(rows)
dev = torch.device("cuda", 0)
ids, cu = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)


def run(precision):
    enc = HipEncoder(dims, device=dev, precision=precision, flags=_lib.OP_FLAG_NO_F8)
    enc.load_state_dict(state, calibrate=False)
    p, r = enc.forward_packed(ids, cu, cu_np, max_len)
    torch.cuda.synchronize()
    out = (p.cpu().numpy(), r.cpu().numpy(), enc.effective_policy()["kernel_set"])
    enc.close()
    return out


ref = run("bf16x3")
fams = list(_lib.OP_FAMILIES)
cases = [[f] for f in fams] + [["qk", "pv"], ["wqkv", "qk", "pv"], ["attn_out", "wi", "mlp_out"], ["wi", "mlp_out"], fams]
for case in cases:
    spec = ",".join(f"{f}={0 if f in case else 3}" for f in fams)
    p, r, ks = run(spec)
    print(f"{args.model} {args.weights}: single-pass bf16 in {'+'.join(case):32s} -> max |d logit| {max(np.abs(p - ref[0]).max(), np.abs(r - ref[1]).max()):.3e}   [{ks}]", flush=True)
