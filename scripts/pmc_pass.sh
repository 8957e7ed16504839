#!/bin/bash
# Usage: scripts/pmc_pass.sh <tag> "<counters...>" [bench args]  -- one PMC pass of bench.py, summarised per kernel
set -u
TAG=$1; shift
COUNTERS=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc $COUNTERS -d $OUT -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        for key in ("rowgemm_kernel", "kstream_gemm_kernel", "attn_kernel", "gemm_kernel"):
            if key in k:
                k = k[k.index(key):][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k, c in sorted(acc.items()):
        if not any(s in k for s in ("rowgemm", "kstream", "attn_kernel", "gemm_kernel")):
            continue
        n = len(cnt[k])
        print(f"{k:72s} n={n:4d} " + " ".join(f"{a}={v/n:.4g}" for a, v in sorted(c.items())))
PY
find $OUT -name "*.csv" -size +4M -delete
