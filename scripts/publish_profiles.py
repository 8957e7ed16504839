#!/usr/bin/env python
"""Copy what scripts/rocprof_pass.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked): the rocprofv3
kernel-stats CSV, the text summary of the stats + PMC passes, the bench line printed under rocprofv3, and the
per-launch HBM traffic merged into profiles/pmc_traffic.json (the file bench.py reads roofline.traffic from).
Usage: scripts/publish_profiles.py <tag> [<tag> ...]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")
traffic_path = os.path.join(PROFILES, "pmc_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for tag in sys.argv[1:]:
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(PROFILES, f"{tag}_kernel_stats.csv"))
    by_grid = os.path.join(src, "trace", "trace_by_grid.csv")
    if os.path.exists(by_grid):
        shutil.copy(by_grid, os.path.join(PROFILES, f"{tag}_kernel_by_grid.csv"))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(PROFILES, f"{tag}_rocprofv3_summary.txt"))
    for line in open(os.path.join(src, "bench_under_rocprof.log")):
        if line.startswith("{") and '"roofline"' in line:
            json.dump(json.loads(line), open(os.path.join(PROFILES, f"{tag}_bench_under_rocprofv3.json"), "w"), indent=1)
    part = os.path.join(src, "pmc_traffic.json")
    if os.path.exists(part):
        traffic.update(json.load(open(part)))
    print("published", tag)
json.dump(traffic, open(traffic_path, "w"), indent=1, sort_keys=True)

# per-launch averages of the dominant kernels in the committed kernel traces: what bench.py prints beside its own in-step
# figure (roofline.rocprofv3_avg_launch_ms) -- key = "<kernel kind>|<model>|<shape>|<kernel set>" as roofline.traffic_key
import csv
import re

avgs_path = os.path.join(PROFILES, "rocprof_launch_avgs.json")
avgs = json.load(open(avgs_path)) if os.path.exists(avgs_path) else {}
KINDS = [(r"(?<!pack_)layer16p_kernel<", "fused_layer_attnout_mlp_qkv"), (r"rowgemm_kernel<\d+, 0, 4,", "fused_layer_attnout_mlp_qkv")]
for tag in sys.argv[1:]:
    bench_json = os.path.join(PROFILES, f"{tag}_bench_under_rocprofv3.json")
    stats_csv = os.path.join(PROFILES, f"{tag}_kernel_stats.csv")
    if not (os.path.exists(bench_json) and os.path.exists(stats_csv)):
        continue
    line = json.load(open(bench_json))
    key_tail = line["roofline"]["traffic_key"].split("|", 1)[1]
    rows = list(csv.DictReader(open(stats_csv)))
    # per-(kernel, grid) averages when the pass kept them (scripts/trace_by_grid.py): the WHOLE-batch launches -- the largest grid of
    # the name -- are what roofline.achieved is quoted on; the half-batch launches of the two-sequence mode are listed beside them
    grid_csv = os.path.join(PROFILES, f"{tag}_kernel_by_grid.csv")
    grid_rows = list(csv.DictReader(open(grid_csv))) if os.path.exists(grid_csv) else []
    for pat, kind in KINDS:
        hit = [r for r in rows if re.search(pat, r.get("Name", ""))]
        ghit = [r for r in grid_rows if re.search(pat, r.get("Name", ""))]
        if ghit:
            widest = max(int(r["GridSize"]) for r in ghit)
            whole = [r for r in ghit if int(r["GridSize"]) == widest]
            half = [r for r in ghit if int(r["GridSize"]) * 2 == widest]
            calls = sum(int(r["Calls"]) for r in whole)
            total = sum(float(r["TotalDurationNs"]) for r in whole)
            entry = {"avg_ms": total / calls / 1e6, "calls": calls, "source": f"profiles/{tag}_kernel_by_grid.csv ({pat}, whole-batch launches: grid {widest})"}
            if half:
                hc = sum(int(r["Calls"]) for r in half)
                entry["half_batch_avg_ms"] = sum(float(r["TotalDurationNs"]) for r in half) / hc / 1e6
                entry["half_batch_calls"] = hc
            avgs[f"{kind}|{key_tail}"] = entry
            break
        if hit:
            calls = sum(int(r["Calls"]) for r in hit)
            total = sum(float(r["TotalDurationNs"]) for r in hit)
            avgs[f"{kind}|{key_tail}"] = {"avg_ms": total / calls / 1e6, "calls": calls, "source": f"profiles/{tag}_kernel_stats.csv ({pat})"}
            break
json.dump(avgs, open(avgs_path, "w"), indent=1, sort_keys=True)
