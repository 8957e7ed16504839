#!/usr/bin/env python
"""GPU probe: what a SMALL forward costs (a query with a handful of contexts) and how much of it is launch overhead.

For n pairs of L tokens: wall time per forward with the stream kept busy (back-to-back launches, one synchronisation at the
end = the enqueue rate), wall time of a lone forward (enqueue + synchronise), and the sum of the kernels' own durations
(library profile: HIP events around every launch).  wall >> kernels = launch-bound.

    python scripts/small_request_probe.py [--model xsmall|base|en-gte|large]
"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from open_provence_amd.engine import HipEncoder  # noqa: E402
from open_provence_amd.packing import pack_rows  # noqa: E402
from open_provence_amd.synthetic import named_dims, refinit_state_dict, synth_pair_batch  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="xsmall")
    ap.add_argument("--flags", default="", help="comma-separated OP_FLAG_* names without the prefix")
    ap.add_argument("--sizes", default="", help="n x L pairs, e.g. 16x256,64x512 (default: the standard ladder)")
    args = ap.parse_args()
    from open_provence_amd import _lib
    flag_bits = 0
    for name in filter(None, args.flags.split(",")):
        flag_bits |= int(getattr(_lib, "OP_FLAG_" + name.strip()))
    dims = named_dims(args.model)
    enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=flag_bits)
    enc.load_state_dict(refinit_state_dict(dims, seed=7))
    print(args.model, "kernel set", enc.effective_policy()["kernel_set"], flush=True)
    sizes = [tuple(int(v) for v in item.split("x")) for item in args.sizes.split(",") if item] or [(1, 64), (1, 512), (4, 128), (8, 128), (16, 256), (32, 256), (64, 512)]
    print("flags", args.flags or "-", flush=True)
    for n, L in sizes:
        rows = synth_pair_batch(dims, n, L, seed=3)
        ids_np, cu_np, max_len = pack_rows(rows)
        ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
        keep = torch.empty(int(cu_np[-1]), dtype=torch.float32, device="cuda")
        for _ in range(5):
            enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
        t_enq = (time.perf_counter() - t0) / reps * 1e3
        torch.cuda.synchronize()
        busy = (time.perf_counter() - t0) / reps * 1e3
        lone = []
        for _ in range(20):
            t1 = time.perf_counter()
            enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
            torch.cuda.synchronize()
            lone.append((time.perf_counter() - t1) * 1e3)
        lone.sort()
        enc.profile_enable(True)
        enc.profile_reset()
        enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
        torch.cuda.synchronize()
        prof = enc.profile_read()
        enc.profile_enable(False)
        kernels = sum(v["total_ms"] for v in prof.values())
        launches = sum(v["launches"] for v in prof.values())
        print(f"{n:3d} x {L:4d} ({int(cu_np[-1]):6d} tokens): back-to-back {busy:6.3f} ms/forward (host enqueue {t_enq:6.3f}), lone median {lone[len(lone) // 2]:6.3f} ms,"
              f" kernels {kernels:6.3f} ms in {launches} launches", flush=True)
    enc.close()


if __name__ == "__main__":
    main()
