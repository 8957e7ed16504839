#!/usr/bin/env python
"""Measurement aid: what every kernel set makes of the trained-like checkpoint proxy (synthetic.trained_like_state_dict) on
Zipf token rows -- max |logit difference| to the fp32 oracle (and the oracle in fp64), per set; the calibration report."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from open_provence_amd.engine import HipEncoder
from open_provence_amd.packing import pack_rows
from open_provence_amd.synthetic import named_dims, pad_rows, trained_like_state_dict, zipf_token_rows
from oracle.modernbert_oracle import oracle_forward

kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = eval(v)
dims = named_dims("xsmall")
state = trained_like_state_dict(dims, seed=7, **kw)
rows = zipf_token_rows(dims, 32, 512, seed=11)
ids, mask = pad_rows(rows)
ref32 = oracle_forward(state, dims, ids, mask)
try:
    ref64 = oracle_forward({k: v.double() for k, v in state.items()}, dims, ids, mask, dtype=torch.float64)
except TypeError:
    ref64 = None
p32, r32 = ref32.pruning_logits.numpy(), ref32.ranking_logits.numpy()
print("proxy", kw, "| |prune| max %.2f, |rank| max %.2f" % (np.abs(p32).max(), np.abs(r32).max()))
if ref64 is not None:
    print("oracle fp32 vs fp64: prune %.2e rank %.2e" % (np.abs(p32 - ref64.pruning_logits.numpy()).max(), np.abs(r32 - ref64.ranking_logits.numpy()).max()))
ids_np, cu_np, max_len = pack_rows(rows)
i_d, c_d = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
enc = HipEncoder(dims, device="cuda:0")
enc.load_state_dict(state)
print("calibration:", {k: enc.calibration[k] for k in ("chosen_set", "default_set", "default_err", "candidates")})
for ks in ("bf16x3", "f16-f8-w", "f16-f8", "f16", "bf16"):
    try:
        enc.select_kernel_set(ks)
    except Exception as e:
        print(ks, "unavailable", str(e)[:60]); continue
    p, r = enc.forward_packed(i_d, c_d, cu_np, max_len)
    p, r = p.cpu().numpy().reshape(32, 512, 2), r.cpu().numpy()
    print(f"{ks:10s} vs fp32 oracle: prune {np.abs(p - p32).max():.2e} rank {np.abs(r - r32).max():.2e}  finite {np.isfinite(p).all()}")
