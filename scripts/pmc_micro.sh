#!/bin/bash
# PMC passes over one microbenchmark binary (GPU box): scripts/pmc_micro.sh <tag> <binary> -- per-kernel averages
set -u
TAG=$1; BIN=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  N=$1; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/p$N -o pmc -- $GRAFT_REPO_ROOT/$BIN > $OUT/p$N.log 2>&1
}
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass 2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pass 3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_WAVE32_LDS GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k, c in sorted(acc.items()):
        n = len(cnt[k])
        print(f"{k:62s} n={n:4d} " + " ".join(f"{a}={v/n:.5g}" for a, v in sorted(c.items())))
PY
find $OUT -name "*.csv" -size +4M -delete
