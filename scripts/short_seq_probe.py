#!/usr/bin/env python
"""GPU probe: forward time of N sequences of L tokens (xsmall), N swept -- what a process() launch of short contexts costs."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from open_provence_amd.engine import HipEncoder
from open_provence_amd.packing import pack_rows
from open_provence_amd.synthetic import named_dims, synth_pair_batch, synth_state_dict

dims = named_dims("xsmall")
enc = HipEncoder(dims, device="cuda:0", precision="bf16x3", flags=0)
enc.load_state_dict(synth_state_dict(dims, seed=7))
for L in (135, 160, 512):
    for n in (8, 64, 128, 256, 512, 1024):
        rows = synth_pair_batch(dims, n, L, seed=3)
        ids_np, cu_np, max_len = pack_rows(rows)
        ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
        keep = torch.empty(int(cu_np[-1]), dtype=torch.float32, device="cuda")
        for _ in range(3):
            enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"L {L:4d} n {n:5d} tokens {int(cu_np[-1]):7d}  {ms:7.2f} ms  {ms / n * 1e3:6.1f} us/seq  {cu_np[-1] / ms / 1e3:6.2f} M tokens/s", flush=True)
    enc.profile_enable(True)
    enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
    torch.cuda.synchronize()
    print({k: round(v, 3) if isinstance(v, float) else v for k, v in enc.profile_read().items()}, flush=True)
    enc.profile_enable(False)

# the same launch after the GPU has idled for a while (a request's first launch arrives ~10 ms after the last one ended)
rows = synth_pair_batch(dims, 256, 135, seed=3)
ids_np, cu_np, max_len = pack_rows(rows)
ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
keep = torch.empty(int(cu_np[-1]), dtype=torch.float32, device="cuda")
for idle_ms in (0, 1, 3, 10, 30, 100):
    times = []
    for _ in range(8):
        time.sleep(idle_ms / 1e3)
        t0 = time.perf_counter()
        enc.forward_packed(ids, cu, cu_np, max_len, keep_prob=keep)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    print(f"256 x 135 after {idle_ms:3d} ms idle: " + " ".join(f"{t:.2f}" for t in times), flush=True)
