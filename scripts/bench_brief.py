#!/usr/bin/env python
"""Print a one-line digest of a bench.py JSON line read from stdin."""
import json
import sys

d = json.loads([line for line in sys.stdin.read().strip().splitlines() if line.startswith("{")][-1])  # (RCCL prints after it)
r = d.get("roofline") or {}
print(
    f"{d['config']['precision'].split()[0]:7s} {d['value']:9.0f} pairs/s {d['ms_per_step']:7.2f} ms | dominant {r.get('kernel')} "
    f"frac {r.get('frac', 0):.3f} whole {r.get('whole_forward_frac', 0):.3f} | "
    + " ".join(f"{k}={v:.2f}" for k, v in sorted(d["kernel_ms_per_forward"].items(), key=lambda kv: -kv[1]) if v > 0.05)
)
