#!/usr/bin/env python
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate passes, TCC slots) into per-launch HBM
traffic of each kernel kind bench.py names.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-byte requests of wide coalesced streams as 64 bytes, so reads are doubled
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM).  The guide leaves WRITE_SIZE uncalibrated; on this code's 16-byte
per-lane stores it matches known byte counts (the attention kernel writes o as hi + lo planes, 2 x tokens x H x 2 B =
134.2 MB at 256 x 512 tokens of xsmall, H = 256: WRITE_SIZE says 134.2 MB; the q / k / v^T planes of the first-layer
projection: 402.7 MB known, 402.7 MB reported), so writes are taken as reported.  FETCH_SIZE counts L2 misses served by
the Infinity Cache as well as by HBM: re-fetched weight panels show up in it although they never leave the die.
Usage: collect_traffic.py <fetch_dir> <write_dir> <workload key> <out.json>
The workload key is what bench.py prints as roofline.traffic_key minus the kernel kind:
"<model>|<pairs>x<seq_len> or varlen|<kernel set>"."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KIND = [
    (r"(?<!pack_)layer16p_kernel<", "fused_layer_attnout_mlp_qkv"),  # round 6: the wave-pair whole-layer kernel (per-launch average of ITS launches:
    # the last layer's launch stays on rowgemm_kernel and is reported under "fused_layer_last" so that it does not dilute the average)
    (r"rowgemm_kernel<\d+, 1, 4,", "fused_layer_last"),
    (r"rowgemm_kernel<\d+, [01], 4,", "fused_layer_attnout_mlp_qkv"),
    (r"rowgemm_kernel<\d+, 2, 3,", "fused_attnout_ln_wi_geglu"),
    (r"rowgemm_kernel<\d+, 0, 3,", "fused_mlpout_ln_qkv_rope"),
    (r"rowgemm_kernel<\d+, 0, [01],", "rowgemm_ln_qkv_rope"),
    (r"rowgemm_kernel<\d+, 1, 2,", "rowgemm_attn_out"),
    (r"rowgemm_kernel<\d+, 2, 0,", "rowgemm_ln_wi_geglu"),
    (r"kstream_gemm_kernel", "kstream_mlp_out"),
    (r"attn_fp_kernel<\d+, \d+, \w+, \d+, 1,", "attn_local"),
    (r"attn_fp_kernel", "attn_global"),
    (r"panel_qkv_kernel", "gemm_qkv_rope"),
    (r"panel_gemm_kernel<0,", "panel_residual"),  # alternates attention-out / MLP-out projection: split below
    (r"panel_gemm_kernel<1,", "gemm_qk_rope"),
    (r"panel_gemm_kernel<2,", "gemm_v_t"),
    (r"panel_gemm_kernel<3,", "gemm_wi_geglu"),
    (r"attn_kernel", "attn"),
    (r"gemm_kernel<3,", "gemm_wi_geglu"),
    (r"gemm_kernel<2,", "gemm_residual"),
    (r"gemm_kernel<0,", "gemm_qk_rope"),
    (r"gemm_kernel<1,", "gemm_v_t"),
]


def per_kernel(directory, counter):
    acc, cnt = defaultdict(float), defaultdict(set)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        residual_ids = {}  # the residual panel GEMM runs twice per layer, in this order: attention out, MLP out
        # bench.py runs the batch as one launch sequence AND as two half-batch sequences: per-launch figures are those of the
        # whole-batch launches (the largest grid of each kernel name), what roofline.achieved is quoted on
        widest = defaultdict(int)
        for r in rows:
            widest[r["Kernel_Name"]] = max(widest[r["Kernel_Name"]], int(r.get("Grid_Size", 0) or 0))
        for r in rows:
            if int(r.get("Grid_Size", 0) or 0) < widest[r["Kernel_Name"]]:
                continue
            for pat, kind in KIND:
                if re.search(pat, r["Kernel_Name"]):
                    if kind == "panel_residual":
                        slot = residual_ids.setdefault(r["Dispatch_Id"], len(residual_ids))
                        kind = "gemm_attn_out" if slot % 2 == 0 else "gemm_mlp_out"
                    acc[kind] += float(r["Counter_Value"])
                    cnt[kind].add(r["Dispatch_Id"])
                    break
    return {k: acc[k] / len(cnt[k]) for k in acc}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
precision = sys.argv[3]
out_path = sys.argv[4]
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for kind in sorted(set(fetch) | set(write)):
    f_kb, w_kb = fetch.get(kind, 0.0), write.get(kind, 0.0)
    out[f"{kind}|{precision}"] = {
        "fetch_size_kib_raw": f_kb,
        "write_size_kib_raw": w_kb,
        "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; reads x2 (gfx950 correction), writes as reported (they match known store byte counts)",
    }
    print(kind, precision, f"fetch {f_kb/1024:.1f} MiB raw, write {w_kb/1024:.1f} MiB -> {out[f'{kind}|{precision}']['hbm_bytes_per_launch']/1e6:.1f} MB/launch")
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
