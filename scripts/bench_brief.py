#!/usr/bin/env python
"""Print a one-line digest of a bench.py JSON line: `bench_brief.py FILE`, or from stdin when no file is given (and stdin is
not a terminal -- an interactive stdin is refused instead of waiting forever)."""
import json
import sys

if len(sys.argv) > 1:
    text = open(sys.argv[1], encoding="utf-8").read()
elif sys.stdin.isatty():
    raise SystemExit("usage: bench_brief.py FILE   (or pipe the bench line in)")
else:
    text = sys.stdin.read()
d = json.loads([line for line in text.strip().splitlines() if line.startswith("{")][-1])  # (RCCL prints after it)
r = d.get("roofline") or {}
print(
    f"{d['config'].get('requested_policy', d['config'].get('precision', '')).split()[0]:7s} {d['value']:9.0f} pairs/s {d['ms_per_step']:7.2f} ms | dominant {r.get('kernel')} "
    f"frac {r.get('frac', 0):.3f} whole {r.get('whole_forward_frac', 0):.3f} | "
    + " ".join(f"{k}={v:.2f}" for k, v in sorted(d["kernel_ms_per_forward"].items(), key=lambda kv: -kv[1]) if v > 0.05)
)
