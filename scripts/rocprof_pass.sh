#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the bench command, then separate PMC passes
# (rocprofv3 --pmc never combined with sys/hip/hsa traces).  Usage: scripts/rocprof_pass.sh <tag> [bench args...]
set -u
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps ${PROF_STEPS:-30} --warmup ${PROF_WARMUP:-10} --no-cpu-baseline --no-long --no-base "$@" > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-long --no-base "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-long --no-base "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA -d $OUT/pmc_mfma -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-long --no-base "$@" > $OUT/pmc_mfma.log 2>&1
python scripts/trace_by_grid.py $OUT/trace > $OUT/trace_by_grid.log 2>&1
python scripts/summarize_rocprof.py $OUT > $OUT/summary.txt 2>&1
KEY=$(python - $OUT/bench_under_rocprof.log <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{") and '"roofline"' in line:
        print(json.loads(line)["roofline"]["traffic_key"].split("|", 1)[1])
        break
PY
)
echo "workload key: $KEY"
python scripts/collect_traffic.py $OUT/pmc_fetch $OUT/pmc_write "$KEY" $OUT/pmc_traffic.json
head -16 $OUT/summary.txt
grep -E "layer16p|rowgemm|attn_fp|kstream|panel" $OUT/summary.txt | grep -E "MFMA|FETCH|WRITE" | cut -c1-330
find $OUT -name "*.csv" -size +6M -delete
