#!/usr/bin/env python
"""GPU-box probe: does running two half-batches on two HIP streams (attention of one overlapping the whole-layer
kernel of the other) beat one full batch on one stream?  Usage: scripts/overlap_probe.py [--pairs 256] [--parts 2]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

from open_provence_amd.engine import HipEncoder
from open_provence_amd.packing import pack_rows
from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--model", default="xsmall")
    ap.add_argument("--cu-mask", default="", help="'alt': stream i gets the CUs with index % parts == i; 'half': contiguous "
                    "ranges of CUs (hipExtStreamCreateWithCUMask): the parts then run on disjoint CUs, truly side by side")
    ap.add_argument("--n-cus", type=int, default=256)
    args = ap.parse_args()
    dims = named_dims(args.model)
    state = synth_state_dict(dims, seed=7)
    state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.ndim == 2 and "embeddings" not in k else v) for k, v in state.items()}
    dev = torch.device("cuda", 0)
    rows = synth_pair_batch(dims, args.pairs, args.seq_len, seed=1234)

    def prepare(part_rows):
        enc = HipEncoder(dims, device=dev, precision="bf16x3")
        enc.load_state_dict(state)
        ids_np, cu_np, max_len = pack_rows(part_rows)
        return enc, torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev), cu_np, max_len

    full = prepare(rows)
    n = len(rows) // args.parts
    parts = [prepare(rows[i * n : (i + 1) * n]) for i in range(args.parts)]
    if args.cu_mask:
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        words = (args.n_cus + 31) // 32
        streams, keep = [], []
        for i in range(args.parts):
            if args.cu_mask == "alt":
                bits = [c % args.parts == i for c in range(args.n_cus)]
            elif args.cu_mask == "xcd-alt":  # 32 mask bits per XCD (assumed): every other XCD
                bits = [(c // 32) % args.parts == i for c in range(args.n_cus)]
            else:
                bits = [c * args.parts // args.n_cus == i for c in range(args.n_cus)]
            mask = (ctypes.c_uint32 * words)(*[sum(1 << b for b in range(32) if w * 32 + b < args.n_cus and bits[w * 32 + b]) for w in range(words)])
            handle = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask)
            assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
            keep.append(mask)
            streams.append(torch.cuda.ExternalStream(handle.value, device=dev))
    else:
        streams = [torch.cuda.Stream(dev) for _ in range(args.parts)]

    def run_full():
        enc, ids, cu, cu_np, ml = full
        enc.forward_packed(ids, cu, cu_np, ml)

    def run_parts():
        for (enc, ids, cu, cu_np, ml), st in zip(parts, streams):
            with torch.cuda.stream(st):
                enc.forward_packed(ids, cu, cu_np, ml)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    for rep in range(2):
        a, b = timeit(run_full), timeit(run_parts)
        print(f"one stream, {args.pairs} pairs: {a:.3f} ms ({args.pairs / a * 1e3:.0f} pairs/s)   {args.parts} streams x {n} pairs: {b:.3f} ms "
              f"({args.pairs / b * 1e3:.0f} pairs/s)")


if __name__ == "__main__":
    main()
