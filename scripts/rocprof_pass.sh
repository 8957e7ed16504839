#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the bench command, then separate PMC passes.
# Usage: scripts/rocprof_pass.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT/pmc_mfma -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_mfma.log 2>&1
find $OUT -name "*.csv" | head -50
python scripts/summarize_rocprof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -80
# keep the merge small: drop the raw traces above 8 MB
find $OUT -name "*.csv" -size +8M -delete
