#!/usr/bin/env python
"""Same-box A/B of library builds on the headline workload (xsmall, 256 x 512, kernel set f16, ONE launch sequence and two):
    python scripts/ab_pairs.py libA.so libB.so ...   (each in a fresh subprocess, three alternations)"""
import json, os, subprocess, sys
CHILD = r'''
import sys, time, json, torch
sys.path.insert(0, ".")
from open_provence_amd.engine import HipEncoder
from open_provence_amd.packing import pack_rows
from open_provence_amd.synthetic import named_dims, refinit_state_dict, synth_pair_batch
dims = named_dims("xsmall"); state = refinit_state_dict(dims, seed=1234)
rows = synth_pair_batch(dims, 256, 512, seed=1234)
enc = HipEncoder(dims, device="cuda:0"); enc.load_state_dict(state, calibrate=False, kernel_set="f16")
ids_np, cu_np, max_len = pack_rows(rows); ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
halves = []
for part in (rows[:128], rows[128:]):
    i_np, c_np, ml = pack_rows(part); halves.append((torch.from_numpy(i_np).cuda(), torch.from_numpy(c_np).cuda(), c_np, ml))
def one(): enc.forward_packed(ids, cu, cu_np, max_len)
def two():
    for part, (i, c, c_np, ml) in enumerate(halves): enc.forward_packed_on(part, i, c, c_np, ml)
out = {}
for name, fn in (("one", one), ("two", two)):
    for _ in range(15): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): fn()
    torch.cuda.synchronize(); out[name] = round(256 * 40 / (time.perf_counter() - t0))
enc.profile_enable(True); enc.profile_reset(); one(); torch.cuda.synchronize()
out["fused_ms"] = round(enc.profile_read()["fused_layer_attnout_mlp_qkv"]["total_ms"], 3)
print(json.dumps(out))
'''
libs = sys.argv[1:]
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ)
        if lib != "tree":
            env.update(OPEN_PROVENCE_HIP_LIB=lib, OPEN_PROVENCE_HIP_LIB_ANY_ABI="1")
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
