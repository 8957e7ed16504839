#!/usr/bin/env python
"""Measurement / debugging aid: the wave-pair whole-layer kernel (opk_layer32p.hip.h) against the 8 x 16 kernel it replaces
(OP_FLAG_NO_LAYER_PAIRS) on the same weights and rows -- max |logit difference| and pairs/s of each."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from open_provence_amd import _lib
from open_provence_amd.engine import HipEncoder
from open_provence_amd.packing import pack_rows
from open_provence_amd.synthetic import named_dims, refinit_state_dict, synth_pair_batch

dims = named_dims("xsmall")
state = refinit_state_dict(dims, seed=1234)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = synth_pair_batch(dims, pairs, 512, seed=1234)
ragged = [r[: 37 + (i * 53) % 470] for i, r in enumerate(synth_pair_batch(dims, 600, 512, seed=9))]
for ks in ("f16", "bf16"):
    outs = {}
    for label, flags in (("pairs", 0), ("8x16", _lib.OP_FLAG_NO_LAYER_PAIRS)):
        enc = HipEncoder(dims, device="cuda:0", flags=flags)
        enc.load_state_dict(state, calibrate=False, kernel_set=ks)
        res = []
        for batch in (rows, ragged):
            ids_np, cu_np, max_len = pack_rows(batch)
            ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
            p, r = enc.forward_packed(ids, cu, cu_np, max_len)
            torch.cuda.synchronize()
            res.append((p.cpu().numpy(), r.cpu().numpy()))
        ids_np, cu_np, max_len = pack_rows(rows)
        ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
        for _ in range(5):
            enc.forward_packed(ids, cu, cu_np, max_len)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            enc.forward_packed(ids, cu, cu_np, max_len)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        enc.profile_enable(True); enc.profile_reset()
        enc.forward_packed(ids, cu, cu_np, max_len); torch.cuda.synchronize()
        prof = {k: round(v["total_ms"], 3) for k, v in enc.profile_read().items()}
        outs[label] = res
        print(f"{ks:5s} {label:6s} {pairs / dt:9.0f} pairs/s  {dt * 1e3:.3f} ms  {prof}", flush=True)
        enc.close()
    for b, name in enumerate(("256x512", "ragged")):
        dp = np.abs(outs["pairs"][b][0] - outs["8x16"][b][0]).max()
        dr = np.abs(outs["pairs"][b][1] - outs["8x16"][b][1]).max()
        fin = np.isfinite(outs["pairs"][b][0]).all() and np.isfinite(outs["pairs"][b][1]).all()
        print(f"{ks:5s} {name:8s} max |prune diff| {dp:.3e}  max |rank diff| {dr:.3e}  finite {fin}  (|prune| max {np.abs(outs['8x16'][b][0]).max():.3f})", flush=True)
