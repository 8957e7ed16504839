#!/usr/bin/env python
"""GPU-box diagnostic: per-fixture parity report + a quick throughput probe.  Writes gpurun_out/report.json."""
import json
import os
import sys
import time
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch

from parity_utils import run_fixture_on_gpu


def quick_bench(precision: str, n_pairs: int = 256, seq_len: int = 512, steps: int = 5, chunk_rows=None):
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.packing import pack_rows
    from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict

    dims = named_dims("xsmall")
    enc = HipEncoder(dims, device="cuda:0", precision=precision, chunk_rows=chunk_rows)
    enc.load_state_dict(synth_state_dict(dims, 7))
    rows = synth_pair_batch(dims, n_pairs, seq_len)
    ids_np, cu_np, max_len = pack_rows(rows)
    ids = torch.from_numpy(ids_np).cuda()
    cu = torch.from_numpy(cu_np).cuda()
    for _ in range(2):
        enc.forward_packed(ids, cu, cu_np, max_len)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = enc.forward_packed(ids, cu, cu_np, max_len)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    enc.profile_enable(True)
    enc.profile_reset()
    for _ in range(2):
        enc.forward_packed(ids, cu, cu_np, max_len)
    prof = enc.profile_read()
    enc.profile_enable(False)
    finite = bool(torch.isfinite(out[0]).all().item())
    enc.close()
    return {"precision": precision, "chunk_rows": chunk_rows, "ms_per_step": dt * 1e3, "pairs_per_s": n_pairs / dt, "finite": finite, "profile": prof}


def main():
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    out = {"device": torch.cuda.get_device_name(0), "parity": [], "bench": []}
    names = sys.argv[1:] or ["g0b_hd64_refinit", "g0c_hd64_synth", "g1m_meanpool", "g1_xsmall", "g2_gte_varlen"]
    for name in names:
        for precision in ("bf16x3", "bf16"):
            try:
                rep = run_fixture_on_gpu(name, precision)
            except Exception as exc:  # keep going: one call must report everything
                rep = {"name": name, "precision": precision, "error": repr(exc), "trace": traceback.format_exc()}
            print(json.dumps(rep), flush=True)
            out["parity"].append(rep)
    # chunking must not change results
    try:
        rep = run_fixture_on_gpu("g1_xsmall", "bf16x3", chunk_rows=1024, capture=False)
        rep["note"] = "chunk_rows=1024"
        print(json.dumps(rep), flush=True)
        out["parity"].append(rep)
    except Exception as exc:
        print("chunk test failed", repr(exc), flush=True)
    for precision in ("bf16x3", "bf16"):
        for chunk in (None, 16384, 65536, 140000):
            try:
                b = quick_bench(precision, chunk_rows=chunk)
            except Exception as exc:
                b = {"precision": precision, "chunk_rows": chunk, "error": repr(exc), "trace": traceback.format_exc()}
            print(json.dumps(b), flush=True)
            out["bench"].append(b)
    with open(os.path.join(REPO, "gpurun_out", "report.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
