#!/usr/bin/env python
"""GPU-box probe: does replaying the forward as one HIP graph beat launching its ~24 kernels one by one?
(xsmall, 256 pairs x 512 tokens, bf16 checkpoint; same box, alternating)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open_provence_amd.engine import HipEncoder
from open_provence_amd.synthetic import named_dims, synth_state_dict

dims = named_dims("xsmall")
enc = HipEncoder(dims, device="cuda:0", precision="bf16x3")
state = synth_state_dict(dims, 7)
state = {k: (v.to(torch.bfloat16) if any(t in k for t in ("Wqkv", "Wo", "Wi")) else v) for k, v in state.items()}
enc.load_state_dict(state)
B, L = 256, 512
rng = np.random.default_rng(1234)
ids_np = rng.integers(1000, dims.vocab_size - 1000, B * L).astype(np.int32)
cu_np = (np.arange(B + 1) * L).astype(np.int32)
ids = torch.from_numpy(ids_np).cuda(); cu = torch.from_numpy(cu_np).cuda()
keep = torch.empty(B * L, dtype=torch.float32, device="cuda")

def eager(n):
    for _ in range(n):
        enc.forward_packed(ids, cu, cu_np, L, keep_prob=keep)

eager(5); torch.cuda.synchronize()
side = torch.cuda.Stream()
graph = None
try:
    with torch.cuda.stream(side):
        eager(2)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            outs = enc.forward_packed(ids, cu, cu_np, L, keep_prob=keep)
    torch.cuda.synchronize()
except Exception as e:  # noqa: BLE001
    print("capture failed:", repr(e)[:300])
    graph = None

def timed(fn, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for r in range(3):
    print(f"eager : {timed(eager):.3f} ms/step")
    if graph is not None:
        print(f"graph : {timed(lambda n: [graph.replay() for _ in range(n)]):.3f} ms/step")
