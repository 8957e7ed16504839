#!/usr/bin/env python
"""Condense rocprofv3 output directories (kernel stats + PMC passes) into a small text/JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name: str) -> str:
    name = name.replace("void ", "")
    for key in ("layer16p_kernel", "rowgemm_kernel", "kstream_gemm_kernel", "attn_fp_kernel", "panel_gemm_kernel", "panel_qkv_kernel", "gemm_kernel", "attn_kernel", "embed_ln_kernel",
                "ln_kernel", "final_ln_prune_kernel", "rank_head_kernel", "row_map_kernel", "seq_offsets_kernel",
                "capture_rows_kernel"):
        if key in name:
            tail = name[name.index(key):]
            return tail[:60]
    return name[:60]


summary = {}
for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(path)))
    print(f"== kernel stats ({os.path.relpath(path, root)})")
    print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    stats = []
    for r in rows:
        name = short(r.get("Name", ""))
        calls = int(r.get("Calls", 0))
        total = float(r.get("TotalDurationNs", 0)) / 1e6
        avg = float(r.get("AverageNs", 0)) / 1e3
        pct = float(r.get("Percentage", 0))
        stats.append({"kernel": name, "calls": calls, "total_ms": total, "avg_us": avg, "pct": pct})
        print(f"{name:62s} {calls:7d} {total:10.3f} {avg:10.2f} {pct:6.2f}")
    summary["kernel_stats"] = stats

for tag in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    for path in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = short(r.get("Kernel_Name", ""))
            acc[k][r.get("Counter_Name", "")] += float(r.get("Counter_Value", 0))
            cnt[k].add(r.get("Dispatch_Id", ""))
        print(f"== {tag} ({os.path.relpath(path, root)}) -- per-dispatch averages")
        out = {}
        for k, counters in sorted(acc.items()):
            n = max(len(cnt[k]), 1)
            out[k] = {c: v / n for c, v in counters.items()}
            out[k]["dispatches"] = n
            print(f"{k:62s} n={n:5d} " + " ".join(f"{c}={v / n:.4g}" for c, v in counters.items()))
        summary[tag] = out

json.dump(summary, open(os.path.join(root, "summary.json"), "w"), indent=1)
