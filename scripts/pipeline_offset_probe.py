#!/usr/bin/env python
"""Measurement aid (round 6): N independent launch sequences on disjoint CU partitions with a DELIBERATE start offset.

Every CU of a whole-layer launch is in the same phase at the same time (fetch o / x, MLP from L2 + LDS, write q / k / v^T / x),
so the HBM streams of a launch arrive as bursts from all CUs at once.  `HipEncoder.forward_packed_on` already runs the batch
as two launch sequences on the two halves of the CUs and lets them drift; this probe asks what a chosen offset between the
sequences (a spin kernel in front of sequence j: j x offset) and more than two sequences do to pairs/s.

    python scripts/pipeline_offset_probe.py [--pairs 256] [--seq-len 512] [--steps 30]
"""

from __future__ import annotations

import argparse
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from open_provence_amd.engine import HipEncoder  # noqa: E402
from open_provence_amd.packing import pack_rows  # noqa: E402
from open_provence_amd.synthetic import named_dims, refinit_state_dict, synth_pair_batch  # noqa: E402


def masked_streams(device: torch.device, parts: int, interleave: bool) -> list[torch.cuda.Stream]:
    """`parts` HIP streams on disjoint CU sets: contiguous ranges of the mask's bit index, or bit index mod parts."""

    n_cus = int(torch.cuda.get_device_properties(device).multi_processor_count)
    hip = ctypes.CDLL("libamdhip64.so")
    words = (n_cus + 31) // 32
    out = []
    with torch.cuda.device(device):
        for part in range(parts):
            bits = [((c % parts) if interleave else (c * parts // n_cus)) == part for c in range(n_cus)]
            mask = (ctypes.c_uint32 * words)(*[sum(1 << b for b in range(32) if w * 32 + b < n_cus and bits[w * 32 + b]) for w in range(words)])
            handle = ctypes.c_void_p()
            if hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask) != 0:
                raise OSError("hipExtStreamCreateWithCUMask failed")
            out.append(torch.cuda.ExternalStream(handle.value, device=device))
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--model", default="xsmall")
    ap.add_argument("--kernel-set", default="f16")
    ap.add_argument("--layers", type=int, default=0, help="model depth override (3: ONE wave-pair launch per step... a per-step lock is then a per-layer lock)")
    ap.add_argument("--parts", default="2,4,8", help="numbers of sequences to try (free-running mode)")
    ap.add_argument("--offsets", default="0,10,20,40,60,85,130")
    ap.add_argument("--interleave", action="store_true", help="locked mode: alternate CU-mask bits instead of contiguous halves")
    ap.add_argument("--locked", action="store_true", help="only: two sequences re-locked every step at a chosen offset")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dims = named_dims(args.model, num_layers=args.layers) if args.layers else named_dims(args.model)
    enc = HipEncoder(dims, device=device)
    enc.load_state_dict(refinit_state_dict(dims, seed=1234), calibrate=False, kernel_set=args.kernel_set)
    rows = synth_pair_batch(dims, args.pairs, args.seq_len, seed=1234)

    # spin-kernel calibration: cycles of torch.cuda._sleep per microsecond
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    s0.record()
    torch.cuda._sleep(10_000_000)
    s1.record()
    torch.cuda.synchronize()
    cyc_per_us = 10_000_000 / (s0.elapsed_time(s1) * 1e3)
    print(f"_sleep: {cyc_per_us:.1f} cycles per us", flush=True)

    def run(parts: int, offset_us: float, interleave: bool, unmasked: bool = False) -> float:
        streams = [torch.cuda.Stream(device) for _ in range(parts)] if unmasked else masked_streams(device, parts, interleave)
        per = len(rows) // parts
        work = []
        for j in range(parts):
            ids_np, cu_np, max_len = pack_rows(rows[j * per:(j + 1) * per])
            ids, cu = torch.from_numpy(ids_np).to(device), torch.from_numpy(cu_np).to(device)
            total, n = int(cu_np[-1]), len(cu_np) - 1
            need = int(enc.lib.op_workspace_bytes(enc._handle, n, total, int(max_len)))
            work.append((ids, cu, cu_np, n, total, int(max_len), torch.empty((total, 2), device=device), torch.empty((n, dims.num_labels), device=device),
                         torch.empty(need + 256, dtype=torch.uint8, device=device)))
        torch.cuda.synchronize()

        def step():
            for j, (ids, cu, cu_np, n, total, max_len, prune, rank, ws) in enumerate(work):
                enc._forward_native(ids.data_ptr(), cu.data_ptr(), cu_np, n, total, max_len, prune.data_ptr(), rank.data_ptr(), None, ws, streams[j].cuda_stream)

        for _ in range(6):
            step()
        torch.cuda.synchronize()
        for j in range(1, parts):  # the deliberate offset, once: the sequences then run free
            if offset_us > 0:
                with torch.cuda.stream(streams[j]):
                    torch.cuda._sleep(int(j * offset_us * cyc_per_us))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0 - (parts - 1) * offset_us * 1e-6  # the last sequence starts that much later
        return args.pairs * args.steps / dt

    def run_locked(offset_us: float) -> float:
        """Two sequences on the two CU halves, re-locked EVERY step: sequence 1 may not start a step earlier than `offset` after
        sequence 0 started the same step (event of stream 0 + a spin on stream 1)."""

        streams = masked_streams(device, 2, args.interleave)
        per = len(rows) // 2
        work = []
        for j in range(2):
            ids_np, cu_np, max_len = pack_rows(rows[j * per:(j + 1) * per])
            ids, cu = torch.from_numpy(ids_np).to(device), torch.from_numpy(cu_np).to(device)
            total, n = int(cu_np[-1]), len(cu_np) - 1
            need = int(enc.lib.op_workspace_bytes(enc._handle, n, total, int(max_len)))
            work.append((ids, cu, cu_np, n, total, int(max_len), torch.empty((total, 2), device=device), torch.empty((n, dims.num_labels), device=device),
                         torch.empty(need + 256, dtype=torch.uint8, device=device)))
        torch.cuda.synchronize()
        events = [torch.cuda.Event() for _ in range(args.steps + 8)]

        def step(i):
            for j, (ids, cu, cu_np, n, total, max_len, prune, rank, ws) in enumerate(work):
                if j == 0:
                    events[i].record(streams[0])
                else:
                    streams[1].wait_event(events[i])
                    if offset_us > 0:
                        with torch.cuda.stream(streams[1]):
                            torch.cuda._sleep(int(offset_us * cyc_per_us))
                enc._forward_native(ids.data_ptr(), cu.data_ptr(), cu_np, n, total, max_len, prune.data_ptr(), rank.data_ptr(), None, ws, streams[j].cuda_stream)

        for i in range(6):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(6 + i)
        torch.cuda.synchronize()
        return args.pairs * args.steps / (time.perf_counter() - t0)

    base = run(1, 0.0, False, unmasked=True)
    if args.locked:
        print(f"1 sequence, whole chip: {base:9.0f} pairs/s", flush=True)
        v = run(2, 0.0, args.interleave)
        print(f"2 sequences, {'alternate bits' if args.interleave else 'contiguous'} CU masks, running free: {v:9.0f} pairs/s ({v / base - 1:+.1%})", flush=True)
        for off in (0.0, 10.0, 20.0, 30.0, 40.0, 55.0, 70.0, 100.0, 150.0):
            v = run_locked(off)
            print(f"2 sequences, contiguous CU masks, re-locked every step at offset {off:5.0f} us: {v:9.0f} pairs/s ({v / base - 1:+.1%})", flush=True)
        return
    print(f"1 sequence, whole chip: {base:9.0f} pairs/s", flush=True)
    for parts in [int(v) for v in args.parts.split(',')]:
        for interleave in (False, True):
            for off in [float(v) for v in args.offsets.split(',')]:
                v = run(parts, off, interleave)
                print(f"{parts} sequences, {'interleaved' if interleave else 'contiguous '} CU masks, offset {off:5.0f} us x j: {v:9.0f} pairs/s ({v / base - 1:+.1%})", flush=True)
    v = run(2, 40.0, False, unmasked=True)
    print(f"2 sequences, no CU masks, offset 40 us: {v:9.0f} pairs/s ({v / base - 1:+.1%})", flush=True)


if __name__ == "__main__":
    main()
